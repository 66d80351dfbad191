"""The register-table bound form of the device traversal (gs_body.h "UBR", k_gsearch_ubr.hip; option gs_ubr) on the GPU:
  * ubr_table_kernel's tables == gs_host.h's restatement, byte for byte (and the meta floats bit for bit);
  * searches == the oracle's sequential GraphSearcher: ids, scores, visitedCount and expandedCount — while the form really drops
    neighbours (gs_ubr_dropped) — dot product and cosine, fused blocks and codes by ordinal, trims every 1 / 24 / 200 pushes,
    rerankK from 1 to 150; filtered searches take the plain kernel (euclidean runs the bound form since round 6: lower bucket edges).
The CPU twin (lane emulator) is tests/test_gsearch_emulated.py::test_register_table_bound_form*."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import jvector_amd as J
from jvector_amd import VectorSimilarityFunction as VSF
from oracle import oracle as O
from test_graph_search import build_problem, fused_blocks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    c = J.HipContext(0)
    yield c
    c.close()


def emu_lib():
    import test_gsearch_emulated as E
    if not os.path.exists(E.LIB) or any(os.path.getmtime(s) > os.path.getmtime(E.LIB) for s in E.SRC):
        os.makedirs(os.path.dirname(E.LIB), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", E.SRC[0], "-o", E.LIB])
    return C.CDLL(E.LIB)


@pytest.mark.parametrize("vsf", [VSF.EUCLIDEAN, VSF.DOT_PRODUCT, VSF.COSINE])
def test_bound_tables_equal_the_restatement(ctx, vsf):
    D, M, Q = 768, 96, 21           # (21: a ragged last block of the 8-queries-per-block kernel)
    rng = np.random.default_rng(int(vsf))
    cb = (rng.standard_normal(256 * D) * 0.3).astype(np.float32)
    centroid = (rng.standard_normal(D) * 0.05).astype(np.float32)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb, centroid)
    q = rng.standard_normal((Q, D)).astype(np.float32)
    q[3] = 0.0                      # every entry 0: the scale floor
    q[5, 17] = np.nan               # no usable table
    q[7, 100] = np.inf
    luts = J.QueryTables(ctx, pq, Q).build(q, vsf, J.DecoderKind.FUSED)
    tab, meta = luts.bound_tables()
    L = emu_lib()
    cq = (q - centroid).astype(np.float32)
    for i in range(Q):
        wt = np.empty(M * 64, np.uint32)
        wm = np.empty(4, np.float32)
        L.gs_emu_ubr_table_vsf(cb.ctypes.data_as(C.c_void_p), np.ascontiguousarray(cq[i]).ctypes.data_as(C.c_void_p), M, wt.ctypes.data_as(C.c_void_p),
                               wm.ctypes.data_as(C.c_void_p), 0 if vsf == VSF.EUCLIDEAN else 1)
        if i in (5, 7):
            assert meta[i, 2] == 0.0 and wm[2] == 0.0
            continue
        assert meta[i, 2] == 1.0
        assert np.array_equal(meta[i].view(np.uint32), wm.view(np.uint32)), (i, meta[i], wm)
        assert np.array_equal(tab[i], wt), (i, np.argwhere(tab[i] != wt)[:4])
        # and the table really bounds: for random codes, base + S * sum(b + 1) >= the exact sum of entries (euclidean: the LOWER
        # edges, base + S * sum(b) <= the exact squared distance)
        codes = rng.integers(0, 256, (50, M))
        if vsf == VSF.EUCLIDEAN:
            ent = ((cb.reshape(M, 256, 8).astype(np.float64) - cq[i].reshape(M, 1, 8).astype(np.float64)) ** 2).sum(-1)
        else:
            ent = np.einsum("mcj,mj->mc", cb.reshape(M, 256, 8).astype(np.float64), cq[i].reshape(M, 8).astype(np.float64))
        for c in codes:
            exact = ent[np.arange(M), c].sum()
            k = 2 * (np.arange(M) % (M // 2)) + (c >= 128)
            lane = c & 63
            word = tab[i][((k // 4) * 64 + lane) * 4 + k % 4]
            b = (word >> (8 * (((c >> 6) & 1) + 2 * (np.arange(M) >= M // 2)))) & 0xFF
            if vsf == VSF.EUCLIDEAN:
                assert meta[i, 0] + meta[i, 1] * float(b.sum()) <= exact
            else:
                assert meta[i, 0] + meta[i, 1] * float((b + 1).sum()) >= exact


def _setup(ctx, seed, N, D, M, levels, use_fused, deg):
    v, lv, entry, entry_level, cb, q = build_problem(seed, N=N, D=D, M=M, deg=deg, levels=levels)
    opq = O.OraclePQ(D, M, cb)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    vs = J.VectorSet(ctx, v)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    codes = cv.get(0, N)
    graph = J.GraphIndex(ctx, N, lv, entry, entry_level).set_traversal("device")
    fused = J.FusedPQ(ctx, pq, fused_blocks(codes, lv[0][1]), lv[0][1]) if use_fused else None
    return v, lv, entry, entry_level, opq, pq, vs, cv, codes, graph, fused, q


def test_deferred_scores_above_level_0_on_the_device(ctx):
    """DEFER (round 6, last; gs_body.h, option gs_defer, default on): above level 0 (from gs_defer_min_level) a fresh neighbour whose bound
    lies below the layer's best result is not scored; a level-0 pop it might outrank starts the query over without deferral.  Three
    layers; (1) a proper graph: neighbours ARE deferred and the answer is the oracle's; (2) a RANDOM level-0 graph: most queries start
    over, the answer is still the oracle's, and the context stops deferring on that index (gs_defer_switched_off) unless gs_defer is set
    explicitly — ids, scores, visitedCount and expandedCount == the oracle's GraphSearcher every time"""
    D, M, N = 768, 96, 6000
    for scramble in (False, True):
        v, lv, entry, entry_level, cb, q = build_problem(31, N=N, D=D, M=M, deg=32, top_n=400, top_deg=32, levels=3)
        if scramble:
            lv[0] = (None, np.random.default_rng(5).integers(0, N, lv[0][1].shape).astype(np.int32))
        q = np.concatenate([q, q[::-1] * 0.5 + q * 0.5, -q[:16]]).astype(np.float32)   # 96 queries (>= 64: the switch looks at whole batches)
        opq = O.OraclePQ(D, M, cb)
        pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
        vs = J.VectorSet(ctx, v)
        cv = J.PQVectors.encode_and_build(ctx, pq, vs)
        codes = cv.get(0, N)
        graph = J.GraphIndex(ctx, N, lv, entry, entry_level).set_traversal("device")
        fused = J.FusedPQ(ctx, pq, fused_blocks(codes, lv[0][1]), lv[0][1])
        og = O.OracleGraph(N, lv, entry, entry_level)
        s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=len(q))
        try:
            ctx.set_option("gs_wgx", 0)
            ctx.set_option("gs_defer_min_level", 1)
            if not scramble:
                ctx.set_option("gs_defer", 1)   # (set explicitly: the switch below is not consulted, whatever these toy searches do)
            for vsf in VSF:
                wi, ws, wst = og.search(opq, codes, v, q, int(vsf), 10, 40, fused=True)
                d0, r0, off0 = ctx.stat("gs_deferred"), ctx.stat("gs_defer_restarts"), ctx.stat("gs_defer_switched_off")
                ids, sc, st = s.search(q, vsf, 10, 40, return_stats=True)
                assert ctx.stat("gs_last_ubr") == 1
                assert np.array_equal(st, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws), (scramble, vsf)
                if not scramble:
                    assert ctx.stat("gs_deferred") - d0 > 10 * len(q), (vsf,)
                elif vsf == list(VSF)[0]:
                    # the first batch on the random graph: restarts, and the index is marked
                    assert ctx.stat("gs_defer_restarts") - r0 > len(q) // 10 and ctx.stat("gs_defer_switched_off") == off0 + 1
                else:
                    assert ctx.stat("gs_defer_restarts") == r0 and ctx.stat("gs_deferred") == d0   # no deferral on this index any more
            if scramble:   # set explicitly, the option wins: deferral (and the restarts) are back, the answer stays
                ctx.set_option("gs_defer", 1)
                r0 = ctx.stat("gs_defer_restarts")
                wi, ws, wst = og.search(opq, codes, v, q, O.COSINE, 10, 40, fused=True)
                ids, sc, st = s.search(q, VSF.COSINE, 10, 40, return_stats=True)
                assert np.array_equal(st, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws)
                assert ctx.stat("gs_defer_restarts") > r0
                ctx.set_option("gs_defer", 0)
                d0 = ctx.stat("gs_deferred")
                ids, sc, st = s.search(q, VSF.COSINE, 10, 40, return_stats=True)
                assert np.array_equal(st, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws) and ctx.stat("gs_deferred") == d0
        finally:
            for k in ("gs_wgx", "gs_defer_min_level", "gs_defer"):
                ctx.set_option(k, None)


@pytest.mark.parametrize("levels,use_fused,deg,N", [(2, True, 32, 20000), (2, False, 32, 8000), (3, True, 16, 12000)])
def test_register_table_bound_kernel(ctx, levels, use_fused, deg, N):
    D, M = 768, 96
    v, lv, entry, entry_level, opq, pq, vs, cv, codes, graph, fused, q = _setup(ctx, 17 * levels + deg, N, D, M, levels, use_fused, deg)
    og = O.OracleGraph(len(v), lv, entry, entry_level)
    s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=64)
    try:
        ctx.set_option("gs_ubr", 1)
        ctx.set_option("gs_wgx", 0)
        for vsf in VSF:
            for top_k, rk in ((10, 40), (10, 74), (10, 150), (1, 1)):
                wi, ws, wst = og.search(opq, codes, v, q, int(vsf), top_k, rk, fused=use_fused)
                for trim in (1, 24, 200):
                    ctx.set_option("gs_ubr_trim", trim)
                    before = ctx.stat("gs_ubr_dropped")
                    ids, sc, st = s.search(q, vsf, top_k, rk, return_stats=True)
                    assert ctx.stat("gs_last_ubr") == 1   # (round 6: euclidean too)
                    assert np.array_equal(st, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws), (vsf, rk, trim)
                    if rk >= 40:
                        assert ctx.stat("gs_ubr_dropped") - before > 0.1 * wst[:, 0].sum(), (vsf, rk, trim)
        # a filtered search: no threshold can be proven with rejected nodes around — the plain kernel
        accept = np.ones(len(v), bool)
        accept[::3] = False
        ids, sc, st = s.search(q, VSF.COSINE, 10, 60, return_stats=True, accept=accept)
        wi, ws, wst = og.search(opq, codes, v, q, O.COSINE, 10, 60, fused=use_fused, accept=accept)
        assert ctx.stat("gs_last_ubr") == 0 and np.array_equal(st, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws)
    finally:
        for k in ("gs_ubr", "gs_wgx", "gs_ubr_trim"):
            ctx.set_option(k, None)


def test_register_table_bound_kernel_ties_and_degenerate_queries(ctx):
    """duplicated vectors (equal scores around every threshold and trim pivot), a zero query, a NaN in a query"""
    D, M, N = 768, 96, 6000
    v, lv, entry, entry_level, cb, q = build_problem(5, N=N, D=D, M=M, deg=24, levels=2)
    v = v.copy()
    v[1::2] = v[0:-1:2][: len(v[1::2])]
    opq = O.OraclePQ(D, M, cb)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    vs = J.VectorSet(ctx, v)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    codes = cv.get(0, N)
    graph = J.GraphIndex(ctx, N, lv, entry, entry_level).set_traversal("device")
    fused = J.FusedPQ(ctx, pq, fused_blocks(codes, lv[0][1]), lv[0][1])
    og = O.OracleGraph(N, lv, entry, entry_level)
    s = J.GraphSearcher(ctx, graph, pq, cv, fused, None, max_queries=64)
    q = q.copy()
    q[1] = 0.0
    q[2, 9] = np.nan
    try:
        ctx.set_option("gs_ubr", 1)
        ctx.set_option("gs_wgx", 0)
        for vsf in VSF:
            for trim in (1, 24):
                ctx.set_option("gs_ubr_trim", trim)
                ids, sc, st = s.search(q, vsf, 30, 30, return_stats=True)
                wi, ws, wst = og.search(opq, codes, None, q, int(vsf), 30, 30, fused=True)
                assert ctx.stat("gs_last_ubr") == 1
                assert np.array_equal(st, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws, equal_nan=True), (vsf, trim)
    finally:
        for k in ("gs_ubr", "gs_wgx", "gs_ubr_trim"):
            ctx.set_option(k, None)


def test_fused_rerank_equals_the_rerank_kernel(ctx, N=12000, quick=False):
    """Round 6: the rerank's exact scores inside the traversal wave (gs_body.h gs_rr_round; option gs_fused_rerank, default on).  For
    every similarity and for list lengths on both sides of every rule of exact_fused_rows — one partial round (1, 40), whole rounds (64,
    128), a packed remainder behind whole rounds (74 -> 64 + 10, 150 -> 128 + 22), a partial last round where the remainder does not pack
    (100 -> 100), the cap (256), beyond it (300 -> the kernel of its own) — the results with the option on, with it off and the
    oracle's sequential GraphSearcher + NodeQueue.rerank are the same ids, the same score bits and the same counters; vectors that are
    duplicated force exact-score ties through the tie resolution behind it; codes by ordinal (the plain pair kernel) take the same path."""
    D, M = 768, 96
    for use_fused in (True, False):
        v, lv, entry, entry_level, opq, pq, vs, cv, codes, graph, fused, q = _setup(ctx, 23 + use_fused, N if use_fused else N // 2, D, M, 2, use_fused, 32)
        og = O.OracleGraph(len(v), lv, entry, entry_level)
        s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=64)
        try:
            ctx.set_option("gs_wgx", 0)
            cases = ((1, 1, 1), (10, 40, 40), (10, 64, 64), (10, 74, 64), (10, 100, 100), (20, 150, 128), (10, 256, 256), (10, 300, 0))
            if quick:
                cases = ((10, 40, 40), (10, 74, 64), (10, 100, 100), (10, 300, 0))
            for vsf in VSF:
                for top_k, rk, want_rows in cases:
                    if not use_fused and rk in (64, 256, 300):
                        continue
                    wi, ws, wst = og.search(opq, codes, v, q, int(vsf), top_k, rk, fused=use_fused)
                    ctx.set_option("gs_fused_rerank", 1)
                    ids, sc, st = s.search(q, vsf, top_k, rk, return_stats=True)
                    assert ctx.stat("gs_last_rr_rows") == want_rows, (vsf, rk, ctx.stat("gs_last_rr_rows"))
                    ctx.set_option("gs_fused_rerank", 0)
                    ids0, sc0, st0 = s.search(q, vsf, top_k, rk, return_stats=True)
                    assert ctx.stat("gs_last_rr_rows") == 0
                    assert np.array_equal(ids, ids0) and np.array_equal(sc.view(np.uint32), sc0.view(np.uint32)) and np.array_equal(st, st0), (vsf, rk)
                    assert np.array_equal(st, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws), (vsf, rk)
        finally:
            for k in ("gs_wgx", "gs_fused_rerank"):
                ctx.set_option(k, None)
