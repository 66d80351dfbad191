"""Device-resident graph traversal (k_gsearch.hip) on the GPU: identical ids, scores, visitedCount and expandedCount to
the oracle's sequential GraphSearcher restatement, and to the host traversal.

First run on MI355X in round 2 (green) — since then the default traversal wherever the shape is supported
(JV_TRAVERSAL_AUTO); the CPU lane-emulator twin is tests/test_gsearch_emulated.py."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import jvector_amd as J
from jvector_amd import VectorSimilarityFunction as VSF
from oracle import oracle as O
from test_graph_search import build_problem, fused_blocks


@pytest.fixture(scope="module")
def ctx():
    c = J.HipContext(0)
    yield c
    c.close()


def _setup(ctx, seed, N, D, M, levels, use_fused, deg=16):
    v, lv, entry, entry_level, cb, q = build_problem(seed, N=N, D=D, M=M, deg=deg, levels=levels)
    opq = O.OraclePQ(D, M, cb)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    vs = J.VectorSet(ctx, v)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    codes = cv.get(0, N)
    graph = J.GraphIndex(ctx, N, lv, entry, entry_level).set_traversal("device")
    fused = J.FusedPQ(ctx, pq, fused_blocks(codes, lv[0][1]), lv[0][1]) if use_fused else None
    return v, lv, entry, entry_level, opq, pq, vs, cv, codes, graph, fused, q


@pytest.mark.parametrize("levels,use_fused,D,M", [(1, False, 128, 16), (2, True, 128, 16), (3, True, 256, 32),
                                                  (2, False, 384, 48), (2, True, 512, 64), (2, True, 768, 96),
                                                  (1, False, 1024, 128), (2, True, 1536, 192)])  # every built M
def test_device_traversal_matches_oracle(ctx, levels, use_fused, D, M):
    v, lv, entry, entry_level, opq, pq, vs, cv, codes, graph, fused, q = _setup(ctx, 31 * levels + M, 5000, D, M, levels, use_fused)
    og = O.OracleGraph(len(v), lv, entry, entry_level)
    try:
        for vsf in VSF:
            for rerank, top_k, rk in ((True, 10, 60), (False, 5, 20), (True, 1, 1)):
                s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs if rerank else None, max_queries=64)
                wi, ws, wst = og.search(opq, codes, v if rerank else None, q, int(vsf), top_k, rk, fused=use_fused)
                # the one-wave kernels, the workgroup form, and what AUTO picks for a batch this small (the workgroup form)
                # (and the one-wave kernels' four-lanes-per-neighbour path, gs_quad: every expansion of these degree-16 graphs)
                # (gs_quad is compiled into experimental builds only: the default library ignores the option and runs the plain pair form)
                for form in (0, 1, None, "quad"):
                    ctx.set_option("gs_wgx", 0 if form == "quad" else form)
                    ctx.set_option("gs_quad", 1 if form == "quad" else None)
                    ids, sc, stats = s.search(q, vsf, top_k, rk, return_stats=True)
                    assert ctx.stat("gs_last_wgx") == (0 if form in (0, "quad") else 1)
                    assert np.array_equal(stats, wst), (vsf, rerank, form)
                    assert np.array_equal(ids, wi), (vsf, rerank, top_k, form)
                    assert np.array_equal(sc, ws), (vsf, rerank, top_k, form)
    finally:
        ctx.set_option("gs_wgx", None)
        ctx.set_option("gs_quad", None)


def test_device_equals_host_on_a_large_batch_with_spills_and_overflow(ctx, monkeypatch):
    v, lv, entry, entry_level, opq, pq, vs, cv, codes, graph, fused, _ = _setup(ctx, 5, 8000, 128, 16, 2, True, deg=24)
    rng = np.random.default_rng(1)
    q = (v[rng.integers(0, len(v), 1500)] + 0.1 * rng.standard_normal((1500, 128))).astype(np.float32)
    s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=2048)
    host = J.GraphIndex(ctx, len(v), lv, entry, entry_level).set_traversal("host")
    sh = J.GraphSearcher(ctx, host, pq, cv, fused, vs, max_queries=2048)
    want = sh.search(q, VSF.COSINE, 10, 400, return_stats=True)
    monkeypatch.setenv("JVECTOR_HIP_GS_CAND_CAP", "256")       # live candidates >> 256: partitions + spill tier
    got = s.search(q, VSF.COSINE, 10, 400, return_stats=True)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    monkeypatch.delenv("JVECTOR_HIP_GS_CAND_CAP")
    # rerankK = 1: the visited table is sized for tiny searches... still exact; and a batch larger than the worker count
    got1 = s.search(q, VSF.EUCLIDEAN, 1, 1, return_stats=True)
    want1 = sh.search(q, VSF.EUCLIDEAN, 1, 1, return_stats=True)
    for a, b in zip(got1, want1):
        assert np.array_equal(a, b)


def test_overflowed_queries_are_retried_on_the_device(ctx, monkeypatch, capfd):
    """first pass with a 512-slot visited table overflows for most queries; the retry passes (8x, 64x the table) finish them
    on the device — same ids / scores / counters as the oracle, nothing left for the host searcher"""
    v, lv, entry, entry_level, opq, pq, vs, cv, codes, graph, fused, q = _setup(ctx, 11, 5000, 128, 16, 2, True)
    og = O.OracleGraph(len(v), lv, entry, entry_level)
    s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=64)
    monkeypatch.setenv("JVECTOR_HIP_GS_VCAP_LOG2", "9")
    monkeypatch.setenv("JVECTOR_HIP_GS_RETRY", "1")
    monkeypatch.setenv("JVECTOR_HIP_GRAPH_TIMING", "1")
    ids, sc, st = s.search(q, VSF.COSINE, 10, 100, return_stats=True)
    err = capfd.readouterr().err
    wi, ws, wst = og.search(opq, codes, v, q, O.COSINE, 10, 100, fused=True)
    assert np.array_equal(st, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws)
    line = [x for x in err.splitlines() if "graph_search device" in x][-1]
    first, host = int(line.split("overflow=")[1].split()[0]), int(line.rsplit("host ", 1)[1])
    assert first > 0 and host == 0, line


def test_visited_table_grows_inside_the_kernel(ctx, monkeypatch, capfd):
    """a 512-slot base table with the growth pool on: queries that fill it half move to an 8x table inside the kernel
    (re-inserting their visited set, carrying the spill tier over) and finish in the first launch — no retry pass, no host
    fallback, same ids / scores / counters as the oracle; queries that outgrow even that are retried as before"""
    v, lv, entry, entry_level, opq, pq, vs, cv, codes, graph, fused, q = _setup(ctx, 13, 5000, 128, 16, 2, True)
    og = O.OracleGraph(len(v), lv, entry, entry_level)
    s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=64)
    monkeypatch.setenv("JVECTOR_HIP_GS_VCAP_LOG2", "9")
    monkeypatch.setenv("JVECTOR_HIP_GS_GROW", "1")
    monkeypatch.setenv("JVECTOR_HIP_GS_CAND_CAP", "256")
    monkeypatch.setenv("JVECTOR_HIP_GRAPH_TIMING", "1")
    for rk, expect_clean in ((100, True), (600, False)):
        ids, sc, st = s.search(q, VSF.COSINE, 10, rk, return_stats=True)
        err = capfd.readouterr().err
        wi, ws, wst = og.search(opq, codes, v, q, O.COSINE, 10, rk, fused=True)
        assert np.array_equal(st, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws), rk
        assert (wst[:, 0] * 2 > 512).any()                        # the base table really was too small for some queries
        line = [x for x in err.splitlines() if "graph_search device" in x][-1]
        first = int(line.split("overflow=")[1].split()[0])
        assert (first == 0) == expect_clean, line


def test_exact_score_ties_are_resolved_on_the_device(ctx, monkeypatch, capfd):
    """every base vector stored twice: the exact rerank ties at the K-th place for most queries.  The reference keeps whichever
    copy comes first in its result heap's ARRAY (NodeQueue.java:197-214); the traversal's push log lets rerank_tie_kernel rebuild
    that array on the device — nothing goes to the host searcher — and with a log too small to hold the sequence the same
    queries are re-run on the host: both equal the oracle."""
    rng = np.random.default_rng(77)
    D, M, deg = 128, 16, 16
    base = rng.standard_normal((1500, D)).astype(np.float32)
    v = np.repeat(base, 2, axis=0)[rng.permutation(3000)]
    N = len(v)
    nb = np.full((N, deg), -1, np.int32)
    sims = v @ v.T
    np.fill_diagonal(sims, -np.inf)
    order = np.argsort(-sims, axis=1)[:, :deg]
    for i in range(N):
        d = int(rng.integers(deg // 2, deg + 1))
        nb[i, :d] = order[i, :d]
    lv = [(None, nb)]
    sizes, offs = O.subvector_sizes_offsets(D, M)
    pick = rng.choice(N, 256, replace=False)
    cb = np.concatenate([v[pick, offs[m]: offs[m] + sizes[m]].reshape(-1) for m in range(M)])
    q = (v[rng.integers(0, N, 48)] + 0.01 * rng.standard_normal((48, D))).astype(np.float32)
    opq = O.OraclePQ(D, M, cb)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    vs = J.VectorSet(ctx, v)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    codes = cv.get(0, N)
    og = O.OracleGraph(N, lv, 5, 0)
    graph = J.GraphIndex(ctx, N, lv, 5, 0).set_traversal("device")
    s = J.GraphSearcher(ctx, graph, pq, cv, None, vs, max_queries=64)
    monkeypatch.setenv("JVECTOR_HIP_GRAPH_TIMING", "1")
    for top_k in (5, 7, 9):        # odd K: the boundary falls inside a pair of copies for nearly every query
        wi, ws, wst = og.search(opq, codes, v, q, O.DOT_PRODUCT, top_k, 40)
        for cap, on_device in ((None, True), ("8", False)):
            if cap is None:
                monkeypatch.delenv("JVECTOR_HIP_GS_PUSH_LOG_CAP", raising=False)
            else:
                monkeypatch.setenv("JVECTOR_HIP_GS_PUSH_LOG_CAP", cap)
            ids, sc, st = s.search(q, VSF.DOT_PRODUCT, top_k, 40, return_stats=True)
            err = capfd.readouterr().err
            assert np.array_equal(st, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws), (top_k, cap)
            line = [x for x in err.splitlines() if "graph_search device" in x][-1]
            resolved = int(line.split("rerank ties=")[1].split()[0])
            host = int(line.rsplit("host ", 1)[1])
            assert (resolved > 0 and host == 0) if on_device else (resolved == 0 and host > 0), line
    # and the membership really is order dependent: the (score desc, id asc) selection differs from the reference for some query
    monkeypatch.setenv("JVECTOR_HIP_GS_TIE_CHECK", "0")
    differs = 0
    for top_k in (5, 7, 9):
        wi, _, _ = og.search(opq, codes, v, q, O.DOT_PRODUCT, top_k, 40)
        ids, _ = s.search(q, VSF.DOT_PRODUCT, top_k, 40)
        differs += int((ids != wi).any())
    assert differs > 0, "the tie cases never depended on the heap order: the test data lost its teeth"


def test_unsupported_shape_is_refused(ctx):
    """M = 8 has no specialised build: the pinned device traversal serves it through the generic kernels (== the oracle); what it
    refuses is a search whose queues do not fit a wave's LDS"""
    v, lv, entry, entry_level, opq, pq, vs, cv, codes, graph, fused, q = _setup(ctx, 9, 2000, 64, 8, 1, False)  # M = 8
    s = J.GraphSearcher(ctx, graph, pq, cv, None, vs, max_queries=64)
    ids, sc, st = s.search(q, VSF.COSINE, 10, 40, return_stats=True)
    wi, ws, wst = O.OracleGraph(len(v), lv, entry, entry_level).search(opq, codes, v, q, O.COSINE, 10, 40, fused=False)
    assert np.array_equal(st, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws)
    with pytest.raises(J.UnsupportedError):
        s.search(q, VSF.COSINE, 10, 30000)


def test_device_traversal_accept_ords(ctx):
    v, lv, entry, entry_level, opq, pq, vs, cv, codes, graph, fused, q = _setup(ctx, 19, 4000, 128, 16, 2, True)
    og = O.OracleGraph(len(v), lv, entry, entry_level)
    s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=64)
    rng = np.random.default_rng(2)
    per_query = rng.random((len(q), len(v))) < 0.2
    per_query[3] = False
    for accept in (per_query[0], per_query):
        ids, sc, st = s.search(q, VSF.COSINE, 10, 40, return_stats=True, accept=accept)
        wi, ws, wst = og.search(opq, codes, v, q, O.COSINE, 10, 40, fused=True, accept=accept)
        assert np.array_equal(st, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws)



def test_device_traversal_engineered_ties(ctx):
    """duplicated base vectors, grid coordinates, queries that are base vectors: every decision falls to the NodeQueue
    tie order (the CPU twin runs in tests/test_mock_device.py)"""
    import test_graph_search as T
    T.run_ties_cases(J, ctx, "device")


def test_two_tier_visited_set_sizes(ctx):
    """the visited set's LDS tier (gs_visit1) at several pinned sizes, with a tier-2 table small enough that the frozen tier
    hands over to it, grows it inside the kernel and still finishes: ids / scores / counters equal the oracle's, and the
    context's counters say what ran"""
    v, lv, entry, entry_level, opq, pq, vs, cv, codes, graph, fused, q = _setup(ctx, 17, 6000, 128, 16, 2, True, deg=24)
    og = O.OracleGraph(len(v), lv, entry, entry_level)
    wi, ws, wst = og.search(opq, codes, v, q, O.COSINE, 10, 150, fused=True)
    s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=64)
    try:
        ctx.set_option("gs_wgx", 0)   # (the one-wave kernels' tiers; unpinned, a batch this small would go to the workgroup form)
        for v1, vcap in ((None, None), (0, None), (6, None), (9, None), (12, None), (7, 9)):
            ctx.set_option("gs_v1_log2", v1).set_option("gs_vcap_log2", vcap)
            if vcap is not None:
                ctx.set_option("gs_grow", 1).set_option("gs_retry", 1)
            ctx.reset_stats()
            ids, sc, st = s.search(q, VSF.COSINE, 10, 150, return_stats=True)
            assert np.array_equal(st, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws), (v1, vcap)
            assert ctx.stat("gs_calls_device") == 1 and ctx.stat("gs_queries_device") == len(q) and ctx.stat("gs_calls_host") == 0
            assert ctx.stat("gs_queries_host_fallback") == 0
            assert ctx.stat("gs_last_v1_log2") == (v1 if v1 is not None else 12), (v1, ctx.stat("gs_last_v1_log2"))
    finally:
        for k in ("gs_v1_log2", "gs_vcap_log2", "gs_grow", "gs_retry", "gs_wgx"):
            ctx.set_option(k, None)


def test_host_fallback_with_a_device_resident_level0(ctx, register=None):
    """a graph whose level 0 lives in caller-owned device memory has no host adjacency; queries that overflow the
    (pinned, tiny) visited table used to reach the host searcher with an empty row table.  They now walk a temporary copy:
    same answers as the oracle, the fallback counted."""
    import torch
    v, lv, entry, entry_level, opq, pq, vs, cv, codes, graph, fused, q = _setup(ctx, 19, 3000, 128, 16, 1, False)
    og = O.OracleGraph(len(v), lv, entry, entry_level)
    wi, ws, wst = og.search(opq, codes, v, q, O.COSINE, 10, 80, fused=False)
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    nb = torch.from_numpy(np.ascontiguousarray(lv[0][1], np.int32)).to(dev)
    if register is not None:  # the CPU run against the mock device: declare the tensor's bytes "device memory"
        register(nb.data_ptr())
    g = J.GraphIndex.on_device(ctx, nb, entry)
    s = J.GraphSearcher(ctx, g, pq, cv, None, vs, max_queries=64)
    try:
        ctx.set_option("gs_vcap_log2", 8)          # 256 slots, no LDS tier, no growth, no retry: every query overflows
        ctx.reset_stats()
        ids, sc, st = s.search(q, VSF.COSINE, 10, 80, return_stats=True)
        assert ctx.stat("gs_queries_host_fallback") == len(q)
        assert np.array_equal(st, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws)
        ctx.set_option("gs_vcap_log2", None)
        ids, sc, st = s.search(q, VSF.COSINE, 10, 80, return_stats=True)   # and the graph still serves the device traversal
        assert np.array_equal(st, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws)
    finally:
        ctx.set_option("gs_vcap_log2", None)


def test_auto_traversal_reports_the_host_fallback(ctx, capfd):
    """JV_TRAVERSAL_AUTO on a search the device traversal does not cover (a rerankK whose result queue does not fit a wave's LDS
    share) takes the host searcher — and says so: one stderr notice per context, and the gs_calls_host_auto counter"""
    v, lv, entry, entry_level, cb, q = build_problem(23, N=1500, D=96, M=8, deg=12, levels=1)
    RK = 6000
    opq = O.OraclePQ(96, 8, cb)
    pq = J.ProductQuantization.from_codebooks(ctx, 96, 8, cb)
    vs = J.VectorSet(ctx, v)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    graph = J.GraphIndex(ctx, len(v), lv, entry, entry_level).set_traversal("auto")
    s = J.GraphSearcher(ctx, graph, pq, cv, None, vs, max_queries=64)
    ctx.reset_stats()
    capfd.readouterr()
    ids, sc, st = s.search(q, VSF.EUCLIDEAN, 10, RK, return_stats=True)
    ids2, _, _ = s.search(q, VSF.EUCLIDEAN, 10, RK, return_stats=True)
    err = capfd.readouterr().err
    wi, ws, wst = O.OracleGraph(len(v), lv, entry, entry_level).search(opq, cv.get(0, len(v)), v, q, O.EUCLIDEAN, 10, RK, fused=False)
    assert np.array_equal(ids, wi) and np.array_equal(sc, ws) and np.array_equal(st, wst) and np.array_equal(ids2, wi)
    assert ctx.stat("gs_calls_host_auto") == 2 and ctx.stat("gs_calls_device") == 0
    assert err.count("JV_TRAVERSAL_AUTO takes the HOST searcher") == 1


@pytest.mark.parametrize("levels,D,M,deg", [(1, 768, 96, 64), (2, 768, 96, 40), (2, 128, 16, 64), (2, 384, 48, 48), (1, 512, 64, 33),
                                            (1, 1024, 128, 64), (2, 1536, 192, 64), (1, 1536, 192, 17)])
def test_compacted_pair_kernel(ctx, levels, D, M, deg):
    """rows of 33 ... 64 neighbours, codes by ordinal (the builder's working rows): graph_search_pairc_kernel — one lane per neighbour
    probes the visited set, the fresh ones are scored two lanes each.  Same ids / scores / counters as the oracle and as the
    one-lane-per-neighbour kernel (gs_pairc = 0)."""
    v, lv, entry, entry_level, opq, pq, vs, cv, codes, graph, fused, q = _setup(ctx, 13 * levels + M + deg, 6000, D, M, levels, False, deg=deg)
    og = O.OracleGraph(len(v), lv, entry, entry_level)
    s = J.GraphSearcher(ctx, graph, pq, cv, None, vs, max_queries=64)
    try:
        ctx.set_option("gs_wgx", 0)
        for vsf in VSF:
            for top_k, rk in ((10, 100), (1, 1)):
                wi, ws, wst = og.search(opq, codes, v, q, int(vsf), top_k, rk, fused=False)
                for pairc in (1, 0):
                    ctx.set_option("gs_pairc", pairc)
                    ids, sc, st = s.search(q, vsf, top_k, rk, return_stats=True)
                    assert ctx.stat("gs_last_pair") == (2 if pairc else 0)
                    assert np.array_equal(st, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws), (vsf, rk, pairc)
    finally:
        for k in ("gs_wgx", "gs_pairc"):
            ctx.set_option(k, None)


@pytest.mark.parametrize("levels,use_fused,D,M,deg", [(2, True, 768, 96, 32), (3, True, 768, 96, 24), (1, False, 768, 96, 40),
                                                      (2, True, 128, 16, 16), (2, False, 256, 32, 64), (2, True, 384, 48, 32),
                                                      (2, True, 512, 64, 32), (2, True, 1024, 128, 32)])
def test_workgroup_form_kernel(ctx, levels, use_fused, D, M, deg):
    """gs_wgx = 1: the workgroup form (k_gsearch_wgx.hip — one query per workgroup, the ADC table in LDS, expander waves scoring
    adjacency rows ahead of the control wave): ids, scores and both counters equal the oracle's for every similarity function,
    every M it is built for, 2 / 4 / 8 waves, few and many slots, with and without requests ahead"""
    v, lv, entry, entry_level, opq, pq, vs, cv, codes, graph, fused, q = _setup(ctx, 9 * levels + M, 4000, D, M, levels, use_fused, deg=deg)
    og = O.OracleGraph(len(v), lv, entry, entry_level)
    s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=64)
    try:
        ctx.set_option("gs_wgx", 1)
        for vsf in VSF:
            for top_k, rk in ((10, 80), (1, 1)):
                wi, ws, wst = og.search(opq, codes, v, q, int(vsf), top_k, rk, fused=use_fused)
                for waves, slots, depth in ((8, 16, 1), (4, 4, 1), (2, 2, 0)):
                    ctx.set_option("gs_wgx_waves", waves)
                    ctx.set_option("gs_wgx_slots", slots)
                    ctx.set_option("gs_wgx_depth", depth)
                    before = ctx.stat("gs_calls_wgx")
                    ids, sc, st = s.search(q, vsf, top_k, rk, return_stats=True)
                    assert ctx.stat("gs_calls_wgx") == before + 1
                    assert np.array_equal(st, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws), (vsf, rk, waves, slots, depth)
    finally:
        for k in ("gs_wgx", "gs_wgx_waves", "gs_wgx_slots", "gs_wgx_depth"):
            ctx.set_option(k, None)


def test_workgroup_form_large_batch_with_spills_and_overflow(ctx, n_queries=1500):
    """1500 queries over 256 workgroups (every CU serves several queries one after another: the per-query LDS state is rebuilt),
    rerankK 400 with a 256-key candidate tier (partitions + spill tier), and a 512-slot tier-2 table without an LDS tier
    (growth pool / retry launches): equal to the host searcher"""
    v, lv, entry, entry_level, opq, pq, vs, cv, codes, graph, fused, _ = _setup(ctx, 5, 8000, 128, 16, 2, True, deg=24)
    rng = np.random.default_rng(1)
    q = (v[rng.integers(0, len(v), n_queries)] + 0.1 * rng.standard_normal((n_queries, 128))).astype(np.float32)
    s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=2048)
    host = J.GraphIndex(ctx, len(v), lv, entry, entry_level).set_traversal("host")
    sh = J.GraphSearcher(ctx, host, pq, cv, fused, vs, max_queries=2048)
    want = sh.search(q, VSF.COSINE, 10, 400, return_stats=True)
    try:
        ctx.set_option("gs_wgx", 1)
        ctx.set_option("gs_cand_cap", 256)
        got = s.search(q, VSF.COSINE, 10, 400, return_stats=True)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
        ctx.set_option("gs_vcap_log2", 9)
        ctx.set_option("gs_retry", 1)
        ctx.set_option("gs_grow", 1)
        got = s.search(q, VSF.COSINE, 10, 400, return_stats=True)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
        assert ctx.stat("gs_last_wgx") == 1
    finally:
        for k in ("gs_wgx", "gs_cand_cap", "gs_vcap_log2", "gs_retry", "gs_grow"):
            ctx.set_option(k, None)
