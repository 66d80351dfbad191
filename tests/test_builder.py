"""Batched Vamana construction (jv_hip_builder_* behind jvector_amd/builder.py: BASELINE config 5) — the engine searches the graph it is
building (device traversal over a device-resident, mutable adjacency), prunes with the retain_diverse kernel and backlinks with the
PQ diversity scores, all inside the library.  The reference's builder run by many threads is nondeterministic; what one thread builds IS
pinned — byte for byte, with the builder in reference order and one node per batch — in tests/test_builder_reference_order.py.  Here, for
batches of many nodes and the default list form, what is checked is the contract a Vamana graph has to meet: degrees within maxDegree, no self loops / duplicates / dangling
ids, every node reachable from the entry point, and — the point of the exercise — a search over the built graph finds the
true nearest neighbours (recall against brute force), while every call it is made of is separately parity-tested."""
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


def _data(N, D, seed):
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((40, D)).astype(np.float32)
    v = (centers[rng.integers(0, 40, N)] + 0.6 * rng.standard_normal((N, D))).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    q = (v[rng.integers(0, N, 64)] + 0.1 * rng.standard_normal((64, D))).astype(np.float32)
    return v, q


def check_builder(J, ctx, dev, N, D, M, max_degree, beam, register=None, min_recall=0.85, min_reach=0.995):
    from jvector_amd.builder import build_hierarchical, build_vamana
    VSF = J.VectorSimilarityFunction.COSINE
    v, q = _data(N, D, 3)
    tv = torch.from_numpy(v).to(dev)
    pq = J.ProductQuantization.compute(ctx, tv, M, seed=2)
    vs = J.VectorSet(ctx, tv)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    nbrs, entry, stats = build_vamana(ctx, pq, cv, tv, VSF, max_degree=max_degree, beam_width=beam, alpha=1.2, max_batch=2048)
    assert stats["inserted"] == N and stats["batches"] >= 3 and stats["reprunes"] > 0
    nb = nbrs.cpu().numpy().copy()        # (on the mock the "device" tensor is host memory: detach the view)
    # ---- structural contract ----
    assert nb.shape == (N, max_degree) and nb.min() >= -1 and nb.max() < N
    deg = (nb >= 0).sum(axis=1)
    assert deg.min() >= 1 and deg.max() <= max_degree and stats["avg_degree"] > max_degree / 4
    for i in range(0, N, max(1, N // 200)):
        row = nb[i][nb[i] >= 0]
        assert i not in row and len(set(row.tolist())) == len(row)
        assert (nb[i][:len(row)] >= 0).all()                                    # packed rows: -1 only at the end
    seen = np.zeros(N, bool)
    seen[entry] = True
    frontier = [entry]
    while frontier:
        nxt = nb[frontier].reshape(-1)
        nxt = np.unique(nxt[nxt >= 0])
        nxt = nxt[~seen[nxt]]
        seen[nxt] = True
        frontier = nxt.tolist()
    assert seen.mean() > min_reach, seen.mean()                                    # (almost) everything reachable from the entry
    # ---- the graph serves searches: recall@10 against brute force, through the ordinary (host-adjacency) GraphIndex ----
    graph = J.GraphIndex(ctx, N, [(None, nb)], entry, 0)
    s = J.GraphSearcher(ctx, graph, pq, cv, None, vs, max_queries=64)
    ids, _ = s.search(q, VSF, 10, 4 * beam)
    gt = np.argsort(-(q @ v.T), axis=1)[:, :10]
    recall = np.mean([len(set(a.tolist()) & set(b.tolist())) / 10.0 for a, b in zip(np.asarray(ids), gt)])
    assert recall >= min_recall, recall
    # ---- a second pass (improveConnections for every node: re-insertion against the finished graph) keeps the contract ----
    nb2, entry2, st2p = build_vamana(ctx, pq, cv, tv, VSF, max_degree=max_degree, beam_width=beam, alpha=1.2, max_batch=2048, passes=2, overflow=2.0)
    nb2 = nb2.cpu().numpy()
    assert st2p["inserted"] >= 2 * N - 1 and nb2.shape == (N, max_degree) and nb2.max() < N
    for i in range(0, N, max(1, N // 100)):
        row = nb2[i][nb2[i] >= 0]
        assert i not in row and len(set(row.tolist())) == len(row) and (nb2[i][:len(row)] >= 0).all()
    g2p = J.GraphIndex(ctx, N, [(None, nb2)], entry2, 0)
    ids2p, _ = J.GraphSearcher(ctx, g2p, pq, cv, None, vs, max_queries=64).search(q, VSF, 10, 4 * beam)
    gt2p = np.argsort(-(q @ v.T), axis=1)[:, :10]
    r2p = np.mean([len(set(a.tolist()) & set(b.tolist())) / 10.0 for a, b in zip(np.asarray(ids2p), gt2p)])
    # re-insertion does NOT improve the graph (measured: recall 0.864 vs 0.93 here, rerankK 100 vs 95 at 10M, profiles/r3_k): a
    # re-inserted node's row is rebuilt from one search and loses the back edges the incremental build gave it.  The option stays
    # for experiments; what is pinned is the structural contract and that the graph still serves searches.
    assert r2p >= min_recall - 0.1, r2p
    # ---- improveConnections over every node (jv_hip_builder_improve_batch: search + MERGE with the row + prune + backlink): the
    #      structural contract holds, searches do at least as well as before (a merge cannot lose the edges a node had unless the
    #      prune prefers the new ones), ids are validated
    nb3, entry3, st3 = build_vamana(ctx, pq, cv, tv, VSF, max_degree=max_degree, beam_width=beam, alpha=1.2, max_batch=2048, improve=1, overflow=2.0)
    nb3 = nb3.cpu().numpy()
    assert st3["inserted"] == N and nb3.shape == (N, max_degree) and nb3.max() < N and nb3.min() >= -1
    for i in range(0, N, max(1, N // 100)):
        row = nb3[i][nb3[i] >= 0]
        assert i not in row and len(set(row.tolist())) == len(row) and (nb3[i][:len(row)] >= 0).all() and len(row) >= 1
    g3 = J.GraphIndex(ctx, N, [(None, nb3)], entry3, 0)
    ids3, _ = J.GraphSearcher(ctx, g3, pq, cv, None, vs, max_queries=64).search(q, VSF, 10, 4 * beam)
    r3 = np.mean([len(set(a.tolist()) & set(b.tolist())) / 10.0 for a, b in zip(np.asarray(ids3), gt)])
    assert r3 >= min_recall, r3
    from jvector_amd.builder import GraphBuilder
    gb = GraphBuilder(ctx, pq, cv, vs, VSF, max_degree, beam, 1.2, 2.0)
    gb.seed(0)
    gb.insert_batch(np.array([1], np.int32))
    for bad in (np.array([N], np.int32), np.array([-1], np.int32), np.array([2, 2], np.int32)):
        with pytest.raises(ValueError):
            gb.insert_batch(bad)                       # ids outside the node range / listed twice are refused
        with pytest.raises(ValueError):
            gb.improve_batch(bad)
    gb.close()
    # ---- layered variant (GraphIndexBuilder's hierarchy): nested levels of N / maxDegree^l nodes, searched top-down ----
    if True:
        levels, e2, el2, nb0, st2 = build_hierarchical(ctx, pq, cv, tv, VSF, max_degree=max_degree, beam_width=beam, max_batch=2048, min_top=4)
        assert el2 == len(levels) - 1 >= 1 and st2["levels"][0] == N
        for l in range(1, len(levels)):
            ids_l, nb_l = levels[l]
            assert (np.diff(ids_l) > 0).all() and set(nb_l[nb_l >= 0].tolist()) <= set(ids_l.tolist())
            assert l == 1 or set(ids_l.tolist()) <= set(levels[l - 1][0].tolist())
        assert e2 in set(levels[-1][0].tolist())
        g2 = J.GraphIndex(ctx, N, levels, e2, el2)
        ids2, _ = J.GraphSearcher(ctx, g2, pq, cv, None, vs, max_queries=64).search(q, VSF, 10, 4 * beam)
        r2 = np.mean([len(set(a.tolist()) & set(b.tolist())) / 10.0 for a, b in zip(np.asarray(ids2), gt)])
        assert r2 >= min_recall, r2
        # the whole hierarchy comes from ONE C-ABI call (jv_hip_build_layered: level draws, per-level builders, improve passes, entry
        # point): a second call with the same seed — what any other FFI caller would get — is byte-identical, another seed is not
        again = build_hierarchical(ctx, pq, cv, tv, VSF, max_degree=max_degree, beam_width=beam, max_batch=2048, min_top=4)
        assert again[1] == e2 and again[2] == el2 and len(again[0]) == len(levels)
        for (ia, na), (ib, nb_) in zip(again[0], levels):
            assert (ia is None and ib is None) or np.array_equal(ia, ib)
            assert np.array_equal(na, nb_)
        other = build_hierarchical(ctx, pq, cv, tv, VSF, max_degree=max_degree, beam_width=beam, max_batch=2048, min_top=4, seed=12, improve=1)
        assert not np.array_equal(other[0][0][1], levels[0][1]) and other[4]["levels"][0] == N
        g3l = J.GraphIndex(ctx, N, other[0], other[1], other[2])
        ids3l, _ = J.GraphSearcher(ctx, g3l, pq, cv, None, vs, max_queries=64).search(q, VSF, 10, 4 * beam)
        r3l = np.mean([len(set(a.tolist()) & set(b.tolist())) / 10.0 for a, b in zip(np.asarray(ids3l), gt)])
        assert r3l >= min_recall, r3l
    return stats, recall


def _on_the_mock(fn):
    """run fn(jvector_amd, ctx, register) with the package bound to the mock library"""
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
    registered = []

    def register(ptr):
        registered.append(ptr)
        lib.mock_hip_register_device(C.c_void_p(ptr))

    try:
        ctx = jvector_amd.HipContext(0)
        lib.mock_hip_register_device.argtypes = [C.c_void_p]
        lib.mock_hip_unregister_device.argtypes = [C.c_void_p]
        fn(jvector_amd, ctx, register)
        ctx.close()
    finally:
        # the buffers die with this test: a later test's numpy array that lands on one of these addresses must not be taken
        # for device memory by the mock (seen once as a failure of an unrelated test in a whole-suite run)
        for ptr in registered:
            lib.mock_hip_unregister_device(C.c_void_p(ptr))
        L._lib = saved
        os.environ.pop("JVECTOR_HIP_HOST_THREADS", None)


@pytest.mark.skipif(platform.machine() != "x86_64", reason="the mock build needs the x86-64 lane emulator")
def test_builder_on_the_mock():
    _on_the_mock(lambda J, ctx, register: check_builder(J, ctx, torch.device("cpu"), 500, 128, 16, 16, 24, register=register))


@pytest.mark.skipif(platform.machine() != "x86_64", reason="the mock build needs the x86-64 lane emulator")
def test_builder_on_the_mock_ragged_quantizer():
    """100 dimensions in 12 sub-vectors (9 9 9 9 8 ...): the builder's searches run on the traversal's generic kernels"""
    _on_the_mock(lambda J, ctx, register: check_builder(J, ctx, torch.device("cpu"), 500, 100, 12, 16, 24, register=register))


@pytest.mark.gpu
def test_builder_gpu():
    import jvector_amd as J
    ctx = J.HipContext(0)
    stats, recall = check_builder(J, ctx, torch.device("cuda", 0), 30000, 128, 16, 32, 100, min_recall=0.9)
    print("builder:", dict(stats), "recall@10", recall)
    ctx.close()


@pytest.mark.gpu
def test_builder_gpu_ragged_quantizer():
    """a quantizer outside the specialised traversal builds (100-d, PQ-12, ragged sub-vectors): built and searched on the device
    through the generic kernels"""
    import jvector_amd as J
    ctx = J.HipContext(0)
    ctx.reset_stats()
    # (9-dim sub-vectors are a coarse quantizer for the PQ-scored build: 99.0 % of the nodes reachable at degree 16 / beam 60 in
    # the first hardware run — the structural contract and the device path are what this case pins, not graph quality)
    stats, recall = check_builder(J, ctx, torch.device("cuda", 0), 12000, 100, 12, 32, 100, min_recall=0.75, min_reach=0.97)
    assert ctx.stat("gs_calls_host") == 0
    print("builder (ragged PQ):", dict(stats), "recall@10", recall)
    ctx.close()


@pytest.mark.skipif(platform.machine() != "x86_64", reason="the mock build needs the x86-64 lane emulator")
def test_builder_in_reference_order_batched_on_the_mock():
    """bl_ref_order = 1 with batches of many nodes (the concurrent case; one-node batches are pinned against the oracle in
    test_builder_reference_order.py): the structural contract, recall, re-insertion and improve passes, the layered build"""
    def run(J, ctx, register):
        ctx.set_option("bl_ref_order", 1)
        try:
            check_builder(J, ctx, torch.device("cpu"), 500, 128, 16, 16, 24, register=register)
        finally:
            ctx.set_option("bl_ref_order", 0)
    _on_the_mock(run)


def check_sorted_lists_equal_classic(J, ctx, dev, N, D, M, max_degree, beam, max_batch, vsf=None, overflow=2.0):
    """bl_sorted_lists = 1 stores the symmetric scores the classic path recomputes at every re-prune and keeps the lists sorted: the
    SAME graph must come out (rows as sets — the classic path leaves rows that were never re-pruned in arrival order), insert phase,
    improve pass and layered build alike"""
    from jvector_amd.builder import build_hierarchical, build_vamana
    VSF = vsf if vsf is not None else J.VectorSimilarityFunction.COSINE
    v, _ = _data(N, D, 9)
    tv = torch.from_numpy(v).to(dev)
    pq = J.ProductQuantization.compute(ctx, tv, M, seed=2)
    vs = J.VectorSet(ctx, tv)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    got = {}
    for mode in (0, 1):
        ctx.set_option("bl_sorted_lists", mode)
        try:
            for improve in (0, 1):
                nb, entry, st = build_vamana(ctx, pq, cv, tv, VSF, max_degree=max_degree, beam_width=beam, alpha=1.2, max_batch=max_batch, improve=improve,
                                             overflow=overflow, vector_set=vs)
                got[(mode, "flat", improve)] = (np.sort(nb.cpu().numpy().copy(), axis=1), entry, st["reprunes"])
            levels, e, el, _nb0, _st = build_hierarchical(ctx, pq, cv, tv, VSF, max_degree=max_degree, beam_width=beam, max_batch=max_batch, min_top=4,
                                                          improve=1, overflow=overflow, vector_set=vs)
            got[(mode, "layered", 1)] = ([np.sort(np.asarray(r), axis=1) for _, r in levels], (e, el), 0)
        finally:
            ctx.set_option("bl_sorted_lists", 0)
    for improve in (0, 1):
        a, b = got[(0, "flat", improve)], got[(1, "flat", improve)]
        assert a[1] == b[1] and a[2] == b[2] and np.array_equal(a[0], b[0]), (improve, np.argwhere((a[0] != b[0]).any(axis=1))[:5])
    a, b = got[(0, "layered", 1)], got[(1, "layered", 1)]
    assert a[1] == b[1] and len(a[0]) == len(b[0]) and all(np.array_equal(x, y) for x, y in zip(a[0], b[0]))


@pytest.mark.skipif(platform.machine() != "x86_64", reason="the mock build needs the x86-64 lane emulator")
def test_sorted_lists_build_the_classic_graph_on_the_mock():
    def run(J, ctx, register):
        check_sorted_lists_equal_classic(J, ctx, torch.device("cpu"), 700, 64, 8, 8, 24, 128, overflow=1.5)
        check_sorted_lists_equal_classic(J, ctx, torch.device("cpu"), 400, 64, 8, 8, 24, 64, vsf=J.VectorSimilarityFunction.EUCLIDEAN, overflow=2.0)
    _on_the_mock(run)


@pytest.mark.gpu
def test_sorted_lists_build_the_classic_graph_gpu():
    import jvector_amd as J
    ctx = J.HipContext(0)
    check_sorted_lists_equal_classic(J, ctx, torch.device("cuda", 0), 40000, 128, 16, 32, 100, 4096)
    ctx.close()


def check_builder_with_bound_form(J, ctx, dev, N, max_batch):
    """gs_ubrc = 1: the builder's searches run the register-table bound form over the compacted fresh list (PQ-96, dot product /
    cosine) — every search returns what the plain compacted form returns, so the SAME graph must come out, byte for byte"""
    from jvector_amd.builder import build_vamana
    D, M = 768, 96
    v, _ = _data(N, D, 5)
    tv = torch.from_numpy(v).to(dev)
    pq = J.ProductQuantization.compute(ctx, tv, M, seed=2)
    vs = J.VectorSet(ctx, tv)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    out = {}
    for vsf in (J.VectorSimilarityFunction.COSINE, J.VectorSimilarityFunction.DOT_PRODUCT):
        for mode in (0, 1):
            ctx.set_option("gs_ubrc", mode)
            ctx.set_option("gs_wgx", 0)          # (small batches would otherwise take the workgroup form)
            ctx.reset_stats()
            try:
                nb, entry, st = build_vamana(ctx, pq, cv, tv, vsf, max_degree=32, beam_width=60, alpha=1.2, max_batch=max_batch, improve=1, overflow=2.0, vector_set=vs)
                dropped = ctx.stat("gs_ubr_dropped")
            finally:
                ctx.set_option("gs_ubrc", 0)
                ctx.set_option("gs_wgx", -1)
            out[mode] = (nb.cpu().numpy().copy(), entry, st["visited"], st["expanded"], dropped)
        assert out[0][4] == 0 and out[1][4] > 0.2 * out[1][2], (out[0][4], out[1][4], out[1][2])   # the form ran and dropped neighbours
        assert out[0][1] == out[1][1] and out[0][2] == out[1][2] and out[0][3] == out[1][3] and np.array_equal(out[0][0], out[1][0])


@pytest.mark.skipif(platform.machine() != "x86_64", reason="the mock build needs the x86-64 lane emulator")
def test_builder_searches_in_the_bound_form_on_the_mock():
    _on_the_mock(lambda J, ctx, register: check_builder_with_bound_form(J, ctx, torch.device("cpu"), 260, 64))


@pytest.mark.gpu
def test_builder_searches_in_the_bound_form_gpu():
    import jvector_amd as J
    ctx = J.HipContext(0)
    check_builder_with_bound_form(J, ctx, torch.device("cuda", 0), 40000, 4096)
    ctx.close()
