"""The builder in REFERENCE ORDER (context option bl_ref_order = 1) against the oracle's one-thread restatement of GraphIndexBuilder
(oracle/jv_oracle.c "GraphIndexBuilder, one thread": addGraphNode :605-659, ConcurrentNeighborMap insertDiverse / backlink / insert /
enforceDegree, NodeArray, VamanaDiversityProvider, java.util.Random(0)): with ONE node per batch the engine performs the reference's
list operations exactly, so the adjacency — ids, their order, and (checked before the final enforceDegree) the scores and the
diverseBefore marks — must equal the oracle's byte for byte.  CPU side: the oracle's own pins and the engine on the mock device."""
import ctypes as C
import os
import platform
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "mock"))

from oracle import oracle as O


def test_java_random_literals():
    """java.util.Random(0).nextDouble(): the first draws of every JVM (the builder's level draws, GraphIndexBuilder.java:337,568)"""
    r = O.JavaRandom(0)
    assert [r.next_double() for _ in range(3)] == [0.730967787376657, 0.24053641567148587, 0.6374174253501083]
    r = O.JavaRandom(42)
    assert r.next_double() == 0.7275636800328681   # new Random(42).nextDouble()
    # getRandomGraphLevel: ml = 1 / ln(32); floor(-ln(u) ml) — level l has probability 32^-l (1 - 1/32)
    r = O.JavaRandom(0)
    lv = np.array([r.graph_level(32) for _ in range(200000)])
    assert lv.max() <= 5 and abs((lv >= 1).mean() - 1 / 32) < 0.002 and abs((lv >= 2).mean() - 1 / 1024) < 0.0005
    assert all(O.JavaRandom(0).graph_level(32, add_hierarchy=False) == 0 for _ in range(3))


def _data(N, D, seed, dup=0):
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((12, D)).astype(np.float32)
    v = (centers[rng.integers(0, 12, N)] + 0.5 * rng.standard_normal((N, D))).astype(np.float32)
    if dup:
        v[N - dup:] = v[:dup]            # identical vectors: identical codes, tied scores everywhere
    return v


def _oracle_pq(N, D, M, seed, v):
    rng = np.random.default_rng(seed + 1)
    sub = D // M
    cb = np.stack([v[rng.choice(N, 256, replace=N < 256)][:, m * sub:(m + 1) * sub] + (0.01 * rng.standard_normal((256, sub))).astype(np.float32)
                   for m in range(M)]).astype(np.float32)
    return cb.reshape(-1)


def test_oracle_builder_contract():
    """the restatement on its own: NodeArray order, marks, degrees, the first node is the entry; with the hierarchy the entry is the
    first node of the top level and an upper level holds exactly the nodes drawn onto it"""
    N, D, M = 600, 32, 8
    v = _data(N, D, 5, dup=20)
    cb = _oracle_pq(N, D, M, 5, v)
    pq = O.OraclePQ(D, M, cb)
    codes = pq.encode_all(v)
    for hier in (False, True):
        b = O.OracleBuilder(pq, codes, v, O.DOT_PRODUCT, 8, 30, add_hierarchy=hier)
        lv = np.array([b.add(i) for i in range(N)])
        r = O.JavaRandom(0)
        assert np.array_equal(lv, [r.graph_level(8, hier) for _ in range(N)])
        info = b.info()
        top = int(lv.max())
        assert info["entry_level"] == top and info["entry_node"] == int(np.argmax(lv == top)) and info["n_levels"] == top + 1
        hard = int(np.float32(1.2) * 8)
        for l in range(top + 1):
            for i in range(N):
                row = b.row(l, i)
                assert (row is None) == (lv[i] < l)
                if row is None:
                    continue
                ids, sc, db = row
                assert ids.size <= hard and 0 <= db <= ids.size and (np.diff(sc) <= 0).all() and i not in ids
        b.cleanup()
        for i in range(N):
            ids, sc, db = b.row(0, i)
            assert ids.size <= 8 and (np.diff(sc) <= 0).all() and i not in ids
        del b


def check_reference_order(J, ctx, dev, N, D, M, max_degree, beam, vsf, dup=0, improve=0, register=None, alpha=1.2, overflow=1.2, seed=7):
    """ONE node per batch == addGraphNode; finish == cleanup's enforceDegree; `improve` passes of improveConnections over EVERY node
    in between (the engine's extension of cleanup(), which refines the upper levels' nodes only: the oracle runs the reference's
    improveConnections with the engine's three stated deviations).  Returns (engine rows, oracle rows) for the caller's report."""
    from jvector_amd.builder import GraphBuilder
    v = _data(N, D, seed + M, dup=dup)
    cb = _oracle_pq(N, D, M, seed, v)
    opq = O.OraclePQ(D, M, cb)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb, None)
    tv = torch.from_numpy(v).to(dev)
    vs = J.VectorSet(ctx, tv)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    codes = opq.encode_all(v)
    ctx.set_option("bl_ref_order", 1)
    try:
        gb = GraphBuilder(ctx, pq, cv, vs, vsf, max_degree, beam, alpha, overflow)
        gb.seed(0)
        ob = O.OracleBuilder(opq, codes, v, int(vsf), max_degree, beam, alpha, overflow, add_hierarchy=False, dedupe_ids=improve > 0,
                             improve_full_vectors=improve > 0, improve_sorted_candidates=improve > 0)
        ob.add(0)
        R = gb.row_width()
        for i in range(1, N):
            gb.insert_batch(np.array([i], np.int32))
            ob.add(i)
            if i in (1, 2, 3, N // 3, N // 2, N - 1):          # the working lists mid-build: ids, scores and marks
                ids, sc, db = gb.working_rows()
                for u in range(i + 1):
                    oi, osc, odb = ob.row(0, u)
                    n = int((ids[u] >= 0).sum())
                    assert n == oi.size and np.array_equal(ids[u, :n], oi), (i, u, ids[u], oi)
                    assert np.array_equal(sc[u, :n].view(np.int32), osc.view(np.int32)) and int(db[u]) == odb, (i, u)
        n_imp = 0
        for _ in range(improve):
            for i in range(N):
                gb.improve_batch(np.array([i], np.int32))
                ob.improve(i)
                n_imp += 1
                if i in (0, 1, N // 2, N - 1):
                    ids, sc, db = gb.working_rows()
                    for u in range(N):
                        oi, osc, odb = ob.row(0, u)
                        n = int((ids[u] >= 0).sum())
                        assert n == oi.size and np.array_equal(ids[u, :n], oi), ("improve", i, u, ids[u], oi)
                        assert np.array_equal(sc[u, :n].view(np.int32), osc.view(np.int32)) and int(db[u]) == odb, ("improve", i, u)
                        assert len(set(oi.tolist())) == oi.size
        out = gb.finish(torch.empty((N, max_degree), dtype=torch.int32, device=dev)).cpu().numpy().copy()
        st = gb.stats()
        gb.close()
    finally:
        ctx.set_option("bl_ref_order", 0)
    ob.cleanup()
    want = ob.rows(0, max_degree)
    assert np.array_equal(out, want), np.argwhere((out != want).any(axis=1))[:5]
    assert 0 <= st["reprunes"] <= ob.info()["reprunes"]          # (the oracle also counts the insertDiverse prunes of the inserts)
    assert (out >= 0).sum(axis=1).max() <= max_degree
    return out, want


def _on_the_mock(fn):
    import build_mock
    import jvector_amd
    import jvector_amd._lib as L
    lib = C.CDLL(build_mock.build())
    for table in (L.SIGNATURES, L.COMPAT_SIGNATURES, L.FORMAT_SIGNATURES):
        for name, (res, args) in table.items():
            f = getattr(lib, name)
            f.restype, f.argtypes = res, args
    saved, L._lib = L._lib, lib
    os.environ["JVECTOR_HIP_HOST_THREADS"] = "1"
    try:
        ctx = jvector_amd.HipContext(0)
        fn(jvector_amd, ctx)
        ctx.close()
    finally:
        L._lib = saved
        os.environ.pop("JVECTOR_HIP_HOST_THREADS", None)


@pytest.mark.skipif(platform.machine() != "x86_64", reason="the mock build needs the x86-64 lane emulator")
@pytest.mark.parametrize("vsf,dup", [(O.COSINE, 0), (O.DOT_PRODUCT, 24), (O.EUCLIDEAN, 0)])
def test_one_node_batches_equal_the_reference_on_the_mock(vsf, dup):
    def run(J, ctx):
        check_reference_order(J, ctx, torch.device("cpu"), 400, 64, 8, 8, 20, J.VectorSimilarityFunction(vsf), dup=dup)
    _on_the_mock(run)


@pytest.mark.skipif(platform.machine() != "x86_64", reason="the mock build needs the x86-64 lane emulator")
@pytest.mark.parametrize("vsf,dup,wgx", [(O.COSINE, 0, 0), (O.EUCLIDEAN, 16, 1)])
def test_improve_pass_in_reference_order_on_the_mock(vsf, dup, wgx):
    """(the improve search excludes its own node — ExcludingBits — in the one-wave form, wgx = 0, and in the workgroup form, 1)"""
    def run(J, ctx):
        ctx.set_option("gs_wgx", wgx)
        ctx.reset_stats()
        check_reference_order(J, ctx, torch.device("cpu"), 300, 64, 8, 8, 20, J.VectorSimilarityFunction(vsf), dup=dup, improve=1)
        assert ctx.stat("gs_calls_host") == 0
    _on_the_mock(run)


def _splitmix_permutation(n, seed):
    """builder.cpp seeded_permutation: Fisher-Yates over splitmix64 draws"""
    mask = (1 << 64) - 1
    st = seed & mask
    p = list(range(n))
    for i in range(n - 1, 0, -1):
        st = (st + 0x9E3779B97F4A7C15) & mask
        z = st
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
        z ^= z >> 31
        j = z % (i + 1)
        p[i], p[j] = p[j], p[i]
    return p


def check_layered_reference_order(J, ctx, dev, N, D, M, max_degree, beam, vsf, improve, seed=11):
    """jv_hip_build_layered with max_batch = 1 in reference order: EVERY level's adjacency equals the oracle's one-thread build of that
    level's nodes (inserted in the engine's seeded order, then `improve` passes, then enforceDegree).  What stays the engine's own
    design is how the levels are composed: each level is a graph of its own (the reference hands entry points from level to level
    inside one insert) and the level draws come from splitmix64, not java.util.Random."""
    from jvector_amd.builder import build_hierarchical
    v = _data(N, D, 21 + M)
    cb = _oracle_pq(N, D, M, 21, v)
    opq = O.OraclePQ(D, M, cb)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb, None)
    tv = torch.from_numpy(v).to(dev)
    vs = J.VectorSet(ctx, tv)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    codes = opq.encode_all(v)
    ctx.set_option("bl_ref_order", 1)
    try:
        levels, entry, entry_level, _nb0, stats = build_hierarchical(ctx, pq, cv, tv, vsf, max_degree=max_degree, beam_width=beam, alpha=1.2, seed=seed,
                                                                     min_top=4, overflow=1.2, max_batch=1, improve=improve, vector_set=vs)
    finally:
        ctx.set_option("bl_ref_order", 0)
    assert len(levels) >= 2 and entry_level == len(levels) - 1
    for l, (ids, rows) in enumerate(levels):
        gids = np.arange(N, dtype=np.int32) if ids is None else np.asarray(ids)
        n_l = gids.size
        sub_v, sub_c = v[gids], codes[gids]
        ob = O.OracleBuilder(opq, sub_c, sub_v, int(vsf), max_degree, beam, 1.2, 1.2, add_hierarchy=False, dedupe_ids=True,
                             improve_full_vectors=True, improve_sorted_candidates=True)
        order = _splitmix_permutation(n_l, seed + l)
        for i in order:
            ob.add(i)
        for _ in range(improve):
            for i in order:
                ob.improve(i)
        ob.cleanup()
        want = ob.rows(0, max_degree)
        want = np.where(want >= 0, gids[np.maximum(want, 0)], -1).astype(np.int32)
        assert np.array_equal(np.asarray(rows), want), (l, np.argwhere((np.asarray(rows) != want).any(axis=1))[:5])
    return stats


@pytest.mark.skipif(platform.machine() != "x86_64", reason="the mock build needs the x86-64 lane emulator")
def test_layered_build_with_one_node_batches_on_the_mock():
    def run(J, ctx):
        check_layered_reference_order(J, ctx, torch.device("cpu"), 260, 64, 8, 6, 16, J.VectorSimilarityFunction.DOT_PRODUCT, improve=1)
    _on_the_mock(run)
