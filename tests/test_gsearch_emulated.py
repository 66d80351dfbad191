"""CPU check of the device-resident graph traversal (jvector_amd/csrc/gs_body.h — the body of graph_search_kernel):
the kernel source is compiled unchanged for a 64-lane wave emulator (tests/emu/) and must reproduce the oracle's
sequential GraphSearcher restatement — same kept result set, same approximate scores bit for bit, same visitedCount /
expandedCount — including the paths a GPU run rarely takes (spill-tier partition, popping from the spill tier,
overflow reporting).  The GPU twin of this test is tests/test_zz_device_traversal_gpu.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import platform
import pytest

from oracle import oracle as O
from test_graph_search import build_problem, fused_blocks

pytestmark = pytest.mark.skipif(platform.machine() != "x86_64", reason="the lane emulator's context switch is x86-64 assembly")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "tests", "emu", "gs_emu.cpp"), os.path.join(ROOT, "tests", "emu", "hip_emu.h"),
       os.path.join(ROOT, "jvector_amd", "csrc", "gs_body.h"), os.path.join(ROOT, "jvector_amd", "csrc", "gs_host.h"),
       os.path.join(ROOT, "jvector_amd", "csrc", "gx_body.h"), os.path.join(ROOT, "jvector_amd", "csrc", "gs_params.h")]
LIB = os.path.join(ROOT, "build", "emu", "libgs_emu.so")


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas",
                               SRC[0], "-o", LIB])
    lib = C.CDLL(LIB)
    lib.gs_emu_search.restype = C.c_long
    return lib


def seq_sum_f32(table, codes):
    """sum_m table[m*256 + code[m]] in ascending m, one f32 accumulator (assembleAndSum order), per row."""
    acc = np.zeros(codes.shape[0], np.float32)
    for m in range(codes.shape[1]):
        acc = (acc + table[m * 256 + codes[:, m].astype(np.int64)]).astype(np.float32)
    return acc


def run_emu(emu, lv, entry, entry_level, opq, codes, q, vsf, rerank_k, fused, vcap_log2=14, spill_cap=8192, cand_cap=256,
            workers=2, pair=1, v1_log2=9, v1_idbits=None, evict_cap=0, lutr=0, wgx_waves=0, wgx_slots=4, wgx_depth=1, wgx_lut_m=0, ub8=0):
    """v1_log2: slots of the visited set's LDS tier (default 512: small enough that the toy searches fill it, freeze it and go
    on in tier 2, so both tiers and the hand-over are exercised by every test); 0 = no LDS tier"""
    N, M, D = codes.shape[0], opq.M, opq.D
    Q = q.shape[0]
    deg0 = lv[0][1].shape[1]
    i32p = C.POINTER(C.c_int32)
    L = len(lv)
    nodes = (i32p * L)(*[C.cast(None, i32p) if ids is None else np.ascontiguousarray(ids, np.int32).ctypes.data_as(i32p)
                         for ids, _ in lv])
    keep = [np.ascontiguousarray(nb, np.int32) for _, nb in lv]
    nbrs = (i32p * L)(*[a.ctypes.data_as(i32p) for a in keep])
    count = (C.c_int32 * L)(*[a.shape[0] for a in keep])
    degree = (C.c_int32 * L)(*[a.shape[1] for a in keep])
    cq = np.ascontiguousarray(q if opq.centroid is None else (q - opq.centroid).astype(np.float32), np.float32)
    bmag = np.zeros(Q, np.float32)
    code_norms = fnorms = None
    blocks = fused_blocks(codes, lv[0][1]) if fused else None
    if vsf == O.COSINE:
        amag = None
        for i in range(Q):
            _, amag, bmag[i] = opq.decoder(q[i], vsf, fused)
        code_norms = seq_sum_f32(amag, codes)
        if fused:
            nb0 = lv[0][1]
            fnorms = np.where(nb0 >= 0, code_norms[np.maximum(nb0, 0)], 0).astype(np.float32)
    out_ids = np.empty((Q, rerank_k), np.int32)
    out_sc = np.empty((Q, rerank_k), np.float32)
    stats = np.zeros((Q, 2), np.int64)
    status = np.full(Q, -9, np.int32)
    fp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    cb = np.ascontiguousarray(opq.codebooks, np.float32)
    codes = np.ascontiguousarray(codes, np.uint8)
    if v1_idbits is None:
        v1_idbits = max(1, int(N - 1).bit_length())
    dropped = C.c_longlong(0)
    n = emu.gs_emu_search(L, nodes, nbrs, count, degree, entry, entry_level, fp(cb), fp(cq), fp(bmag), fp(codes),
                          fp(code_norms), fp(blocks), fp(fnorms), D, M, deg0, Q, rerank_k, int(vsf), vcap_log2, spill_cap,
                          cand_cap, workers, pair, fp(out_ids), fp(out_sc), fp(stats), fp(status), v1_log2, v1_idbits, evict_cap, lutr,
                          wgx_waves, wgx_slots, wgx_depth, wgx_lut_m, ub8, C.byref(dropped))
    assert n >= 0, n
    run_emu.last_ub8_dropped = int(dropped.value)
    dst = (C.c_ulonglong * 3)()
    emu.gs_emu_last_defer(dst)
    run_emu.last_defer = tuple(int(x) for x in dst)   # (deferred neighbours, queries started over, unused)
    return out_ids, out_sc, stats, status, n


def check(out_ids, out_sc, stats, status, want_ids, want_sc, want_stats, allow_overflow=False):
    assert allow_overflow or (status == 0).all(), status
    assert np.isin(status, (0, 1)).all()
    for qi in range(out_ids.shape[0]):
        if status[qi] != 0:
            assert (out_ids[qi] == -1).all()
            continue
        assert np.array_equal(stats[qi], want_stats[qi])
        got = sorted(zip(out_sc[qi].tolist(), (-out_ids[qi]).tolist()), reverse=True)
        want = sorted(zip(want_sc[qi].tolist(), (-want_ids[qi]).tolist()), reverse=True)
        assert got == want, qi


def problem(seed, N, D, M, levels, deg=16, nq=10):
    v, lv, entry, entry_level, cb, q = build_problem(seed, N=N, D=D, M=M, deg=deg, levels=levels)
    opq = O.OraclePQ(D, M, cb)
    codes = opq.encode_all(v)
    return lv, entry, entry_level, opq, codes, q[:nq]


def test_level_map_lookup(emu):
    rng = np.random.default_rng(0)
    nodes = np.sort(rng.choice(1 << 20, 5000, replace=False)).astype(np.int32)
    p = nodes.ctypes.data_as(C.POINTER(C.c_int32))
    for i in (0, 1, 77, 4999):
        assert emu.gs_emu_level_lookup(p, 5000, int(nodes[i])) == i
    missing = int(np.setdiff1d(np.arange(100), nodes)[0])
    assert emu.gs_emu_level_lookup(p, 5000, missing) == -1


@pytest.mark.parametrize("levels,fused,M", [(1, False, 16), (2, True, 16), (3, True, 32), (2, False, 48), (2, True, 64),
                                            (2, True, 96), (1, False, 128), (2, True, 192)])  # every CH16 the kernel is built for
def test_emulated_kernel_matches_oracle(emu, levels, fused, M):
    D = 8 * M
    lv, entry, entry_level, opq, codes, q = problem(100 + levels + M, 2500, D, M, levels)
    og = O.OracleGraph(codes.shape[0], lv, entry, entry_level)
    for vsf in (O.EUCLIDEAN, O.DOT_PRODUCT, O.COSINE):
        for rk in (40, 1):
            wi, ws, wst = og.search(opq, codes, None, q, vsf, rk, rk, fused=fused)
            # two lanes per neighbour (degrees <= 32; expansions with <= 16 fresh neighbours: four lanes each — every expansion of
            # these degree-16 graphs), one lane per neighbour, and the pair form with the four-lane path switched off
            for pair, quad in ((1, 1), (0, 1), (1, 0)):
                os.environ["GS_EMU_QUAD"] = str(quad)
                try:
                    ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes, q, vsf, rk, fused, pair=pair)
                finally:
                    os.environ.pop("GS_EMU_QUAD", None)
                check(ids, sc, st, status, wi, ws, wst)


@pytest.mark.parametrize("levels,fused,M,deg", [(1, False, 16, 16), (2, True, 32, 16), (2, False, 48, 40), (2, True, 64, 24), (2, True, 96, 32),
                                                (3, False, 96, 16), (2, True, 128, 64), (2, True, 192, 32)])
def test_workgroup_form_matches_oracle(emu, levels, fused, M, deg):
    """WGX (gx_body.h): one query per workgroup, the ADC table in LDS, a control wave + expander waves that score rows ahead of
    time — every M it is built for, degrees up to 64, all three similarity functions, 2..4 waves, 2..8 slots, with and without
    requests ahead: results, scores and both counters bit-identical to the oracle"""
    D = 8 * M
    lv, entry, entry_level, opq, codes, q = problem(500 + levels + M, 2000, D, M, levels, deg=deg, nq=6)
    og = O.OracleGraph(codes.shape[0], lv, entry, entry_level)
    for vsf in (O.EUCLIDEAN, O.DOT_PRODUCT, O.COSINE):
        for rk in (40, 1):
            wi, ws, wst = og.search(opq, codes, None, q, vsf, rk, rk, fused=fused)
            for waves, slots, depth, lut_m in ((4, 8, 1, 0), (2, 2, 1, 16), (3, 4, 0, max(16, (M // 2) // 16 * 16))):
                ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes, q, vsf, rk, fused, wgx_waves=waves,
                                                 wgx_slots=slots, wgx_depth=depth, wgx_lut_m=lut_m)
                check(ids, sc, st, status, wi, ws, wst)


@pytest.mark.parametrize("order", ["reverse", "random:5", "random:6"])
def test_workgroup_form_under_lane_reordering(emu, monkeypatch, order):
    """the control wave and the expanders hand rows over through LDS flags: any schedule of the lanes must give the same answer
    (also the spill tier, both visited tiers and the eviction of scored rows: 2 slots, rerankK 300)"""
    monkeypatch.setenv("EMU_LANE_ORDER", order)
    lv, entry, entry_level, opq, codes, q = problem(141, 3000, 128, 16, 2, deg=24, nq=5)
    og = O.OracleGraph(codes.shape[0], lv, entry, entry_level)
    for vsf, fused, rk, slots in ((O.COSINE, True, 300, 2), (O.EUCLIDEAN, False, 60, 5)):
        wi, ws, wst = og.search(opq, codes, None, q, vsf, rk, rk, fused=fused)
        ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes, q, vsf, rk, fused, cand_cap=256, wgx_waves=4,
                                         wgx_slots=slots)
        check(ids, sc, st, status, wi, ws, wst)


@pytest.mark.parametrize("levels,fused,M,deg,N", [(2, True, 96, 32, 2500), (2, False, 96, 32, 2000), (3, True, 64, 24, 3000), (1, True, 96, 16, 3000)])
def test_register_table_bound_form(emu, monkeypatch, levels, fused, M, deg, N):
    """UBR (round 5): the bound table prebuilt (gs_ubr_build_ref) and held in registers, survivors compacted and scored eight lanes
    each, the candidate tier trimmed to what can still be popped — ids, scores and BOTH counters equal the oracle's, euclidean (round 6: lower bucket edges), dot
    product and cosine, fused blocks and codes by ordinal, rerankK from 1 to well above the degree, trims every 1 / 8 / 64 pushes"""
    D = 8 * M
    lv, entry, entry_level, opq, codes, q = problem(900 + levels + M, N, D, M, levels, deg=deg, nq=8)
    og = O.OracleGraph(codes.shape[0], lv, entry, entry_level)
    dropped_total = scored_total = 0
    for vsf in (O.EUCLIDEAN, O.DOT_PRODUCT, O.COSINE):
        for rk in (10, 40, 150, 1):
            wi, ws, wst = og.search(opq, codes, None, q, vsf, rk, rk, fused=fused)
            for trim, v1, cc in (("8", 9, 256), ("1", 12, 128), ("64", 9, 256)):
                monkeypatch.setenv("GS_EMU_UBR_TRIM", trim)
                ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes, q, vsf, rk, fused, ub8=2, v1_log2=v1, cand_cap=cc)
                check(ids, sc, st, status, wi, ws, wst)
                scored_total += int(wst[:, 0].sum())
                dropped_total += run_emu.last_ub8_dropped
    assert dropped_total > 0.1 * scored_total, (dropped_total, scored_total)   # the form really drops neighbours in these searches


def test_deferred_scores_above_level_0(emu, monkeypatch):
    """DEFER (round 6, last; gs_body.h): above level 0 a fresh neighbour whose bound lies below the layer's best result is not scored;
    only the largest upper bound U of the deferred scores is kept.  A pop that a deferred node might outrank makes the query start over
    without deferral; once the pop threshold passes U the deferred nodes are forgotten.  Three layers; the level-0 graph of the second
    problem is RANDOM, so that the search runs out of good candidates and the deferred nodes DO matter (restarts, the empty-queue
    rule); deferral from level 1 and from level 2 — ids, scores, visitedCount and expandedCount equal the oracle's GraphSearcher
    (GraphSearcher.java:263-282,324-331,406-457) every time"""
    from test_graph_search import build_problem
    M, D, N = 96, 768, 3000
    tot = [0, 0]
    for scramble in (False, True):
        v, lv, entry, entry_level, cb, q = build_problem(77, N=N, D=D, M=M, deg=32, top_n=400, top_deg=32, levels=3)
        if scramble:
            rng = np.random.default_rng(5)
            lv[0] = (None, rng.integers(0, N, lv[0][1].shape).astype(np.int32))
        opq = O.OraclePQ(D, M, cb)
        codes = opq.encode_all(v)
        q = q[:6]
        og = O.OracleGraph(N, lv, entry, entry_level)
        for vsf in (O.COSINE, O.DOT_PRODUCT, O.EUCLIDEAN):
            for rk in (1, 12, 70):
                wi, ws, wst = og.search(opq, codes, None, q, vsf, rk, rk, fused=True)
                for minl in ("1", "2"):
                    monkeypatch.setenv("GS_EMU_DEFER_MIN_LEVEL", minl)
                    ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes, q, vsf, rk, True, ub8=2)
                    check(ids, sc, st, status, wi, ws, wst)
                    tot[0] += run_emu.last_defer[0]
                    tot[1] += run_emu.last_defer[1]
                monkeypatch.setenv("GS_EMU_DEFER", "0")
                ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes, q, vsf, rk, True, ub8=2)
                monkeypatch.delenv("GS_EMU_DEFER")
                check(ids, sc, st, status, wi, ws, wst)
                assert run_emu.last_defer[:2] == (0, 0)
    # the paths really ran: neighbours were deferred (also in searches that then ran to their end), and some queries had to start over
    assert tot[0] > 1000 and tot[1] > 10, tot


def test_register_table_bound_form_with_equal_and_extreme_scores(emu, monkeypatch):
    """duplicated vectors (equal scores around every threshold and every trim pivot), shuffled lane orders, a NaN in the query
    (no table: nothing may be dropped, the answer is still the oracle's)"""
    lv, entry, entry_level, opq, codes, q = problem(23, 3000, 768, 96, 2, deg=24, nq=6)
    codes = codes.copy()
    codes[1::2] = codes[0:-1:2][: len(codes[1::2])]
    og = O.OracleGraph(codes.shape[0], lv, entry, entry_level)
    for order in ("", "reverse", "random:9"):
        if order:
            monkeypatch.setenv("EMU_LANE_ORDER", order)
        for vsf in (O.EUCLIDEAN, O.DOT_PRODUCT, O.COSINE):
            wi, ws, wst = og.search(opq, codes, None, q, vsf, 30, 30, fused=True)
            ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes, q, vsf, 30, True, ub8=2)
            check(ids, sc, st, status, wi, ws, wst)
    monkeypatch.delenv("EMU_LANE_ORDER", raising=False)
    qn = q.copy()
    qn[0, 5] = np.nan
    qn[1, :] = 0.0
    for vsf in (O.EUCLIDEAN, O.DOT_PRODUCT, O.COSINE):
        wi, ws, wst = og.search(opq, codes, None, qn, vsf, 30, 30, fused=True)
        ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes, qn, vsf, 30, True, ub8=2)
        check(ids, sc, st, status, wi, ws, wst)


def test_workgroup_form_rare_paths(emu):
    """the control wave's own queue code (gx_body.h gx_control): candidate-tier partitions into the spill tier, an exhaustive search
    that drains the LDS tier and refills it from the spill tier again and again, every size class of the visited set's LDS tier
    (absent / tiny / large), overflow reporting"""
    lv, entry, entry_level, opq, codes, q = problem(7, 4000, 128, 16, 2, deg=24, nq=6)
    og = O.OracleGraph(codes.shape[0], lv, entry, entry_level)
    wi, ws, wst = og.search(opq, codes, None, q, O.COSINE, 400, 400, fused=True)
    assert (wst[:, 0] - wst[:, 1]).min() > 2 * 256
    ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes, q, O.COSINE, 400, True, wgx_waves=4, wgx_slots=6)
    check(ids, sc, st, status, wi, ws, wst)
    # exhaustive: rerankK >= N
    lv, entry, entry_level, opq, codes, q = problem(11, 700, 128, 16, 2, deg=12, nq=4)
    og = O.OracleGraph(codes.shape[0], lv, entry, entry_level)
    wi, ws, wst = og.search(opq, codes, None, q, O.EUCLIDEAN, 800, 800, fused=False)
    assert (wst[:, 1] > 600).all()
    ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes, q, O.EUCLIDEAN, 800, False, vcap_log2=12, wgx_waves=3)
    check(ids, sc, st, status, wi, ws, wst)
    # visited tiers
    lv, entry, entry_level, opq, codes, q = problem(29, 3000, 128, 16, 2, deg=24, nq=8)
    og = O.OracleGraph(codes.shape[0], lv, entry, entry_level)
    for v1 in (0, 6, 10, 15):
        for vsf, fused, rk in ((O.COSINE, True, 120), (O.EUCLIDEAN, False, 60), (O.DOT_PRODUCT, True, 1)):
            wi, ws, wst = og.search(opq, codes, None, q, vsf, rk, rk, fused=fused)
            ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes, q, vsf, rk, fused, v1_log2=v1, evict_cap=64,
                                             wgx_waves=4, wgx_slots=8)
            check(ids, sc, st, status, wi, ws, wst)
    # overflow is reported, not hidden
    lv, entry, entry_level, opq, codes, q = problem(13, 3000, 128, 16, 2, deg=24, nq=4)
    og = O.OracleGraph(codes.shape[0], lv, entry, entry_level)
    wi, ws, wst = og.search(opq, codes, None, q, O.DOT_PRODUCT, 120, 120, fused=True)
    for v1 in (0, 6):
        ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes, q, O.DOT_PRODUCT, 120, True, vcap_log2=9, v1_log2=v1, wgx_waves=4)
        assert (status == 1).all() and (ids == -1).all()
    ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes, q, O.DOT_PRODUCT, 120, True, spill_cap=32, wgx_waves=4)
    assert (status == 1).any()
    check(ids, sc, st, status, wi, ws, wst, allow_overflow=True)


def test_workgroup_form_equal_scores(emu):
    """every vector stored three times: equal PQ codes, equal scores — the pop's tie path (node words decide, NodeQueue.java:125-129)"""
    lv, entry, entry_level, opq, codes, q = problem(17, 1500, 128, 16, 2, deg=16, nq=6)
    codes = codes.copy()
    codes[1::3] = codes[0:-1:3][: len(codes[1::3])]
    codes[2::3] = codes[0:-2:3][: len(codes[2::3])]
    og = O.OracleGraph(codes.shape[0], lv, entry, entry_level)
    for vsf, fused in ((O.COSINE, True), (O.EUCLIDEAN, False)):
        wi, ws, wst = og.search(opq, codes, None, q, vsf, 60, 60, fused=fused)
        ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes, q, vsf, 60, fused, wgx_waves=4)
        check(ids, sc, st, status, wi, ws, wst)


@pytest.mark.parametrize("order", ["reverse", "random:7", "random:8"])
def test_result_does_not_depend_on_lane_scheduling(emu, monkeypatch, order):
    """The emulator runs the lanes of a phase one after another; under a reversed / shuffled order every cross-lane hand-over
    that skipped a barrier or collective would read stale data (verified by deleting one barrier: this test fails)."""
    monkeypatch.setenv("EMU_LANE_ORDER", order)
    lv, entry, entry_level, opq, codes, q = problem(131, 3000, 128, 16, 2, deg=24, nq=6)
    og = O.OracleGraph(codes.shape[0], lv, entry, entry_level)
    for vsf, fused, rk, pair in ((O.COSINE, True, 300, 1), (O.EUCLIDEAN, False, 60, 0)):
        wi, ws, wst = og.search(opq, codes, None, q, vsf, rk, rk, fused=fused)
        ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes, q, vsf, rk, fused, cand_cap=256, pair=pair)
        check(ids, sc, st, status, wi, ws, wst)


def test_degree_above_32_uses_one_lane_per_neighbour(emu):
    """maxDegree 40 > 32: the pair-lane scoring does not apply; the one-lane-per-neighbour path must give the same answers"""
    lv, entry, entry_level, opq, codes, q = problem(23, 2000, 128, 16, 2, deg=40, nq=6)
    assert lv[0][1].shape[1] == 40
    og = O.OracleGraph(codes.shape[0], lv, entry, entry_level)
    for vsf, fused in ((O.COSINE, True), (O.EUCLIDEAN, False)):
        wi, ws, wst = og.search(opq, codes, None, q, vsf, 50, 50, fused=fused)
        ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes, q, vsf, 50, fused)
        check(ids, sc, st, status, wi, ws, wst)


@pytest.mark.parametrize("M,deg", [(16, 40), (32, 64), (48, 48), (64, 33), (96, 64), (128, 64), (192, 40)])
def test_compacted_pair_form_matches_oracle(emu, M, deg):
    """rows of 33 ... 64 neighbours with the codes read by ordinal (the builder's working rows): one lane per neighbour probes the
    visited set, the fresh ones are scored two lanes each after a compaction through LDS — up to two passes per expansion.  Same
    ids / scores / counters as the oracle, and as the one-lane-per-neighbour kernel."""
    D = 8 * M
    lv, entry, entry_level, opq, codes, q = problem(300 + M + deg, 2500, D, M, 2, deg=deg, nq=8)
    assert lv[0][1].shape[1] == deg
    og = O.OracleGraph(codes.shape[0], lv, entry, entry_level)
    two_pass = False
    for vsf in (O.EUCLIDEAN, O.DOT_PRODUCT, O.COSINE):
        for rk in (60, 1):
            wi, ws, wst = og.search(opq, codes, None, q, vsf, rk, rk, fused=False)
            ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes, q, vsf, rk, False, pair=2, cand_cap=256)
            check(ids, sc, st, status, wi, ws, wst)
            two_pass = two_pass or (deg > 32)
    # a fused graph has no compacted form: the launch is refused, not silently run in another form
    if M == 16:
        with pytest.raises(Exception):
            run_emu(emu, lv, entry, entry_level, opq, codes, q, O.COSINE, 10, True, pair=2)


@pytest.mark.parametrize("M,deg,N", [(96, 64, 2500), (96, 40, 2000), (64, 48, 2500)])
def test_register_table_bound_form_over_the_compacted_list(emu, monkeypatch, M, deg, N):
    """UBR over the compacted pair form (the builder's searches: rows of 33 ... 64 neighbours, codes by ordinal): a pass's fresh
    neighbours are bounded, the survivors staged and scored eight lanes each, pass by pass — ids, scores and both counters equal the
    oracle's; trims every 1 / 8 / 64 pushes; duplicated vectors; shuffled lane orders"""
    D = 8 * M
    lv, entry, entry_level, opq, codes, q = problem(500 + M + deg, N, D, M, 1, deg=deg, nq=8)
    assert lv[0][1].shape[1] == deg
    og = O.OracleGraph(codes.shape[0], lv, entry, entry_level)
    dropped_total = scored_total = 0
    for vsf in (O.EUCLIDEAN, O.DOT_PRODUCT, O.COSINE):
        for rk in (100, 10, 1):
            wi, ws, wst = og.search(opq, codes, None, q, vsf, rk, rk, fused=False)
            for trim, v1, cc in (("8", 9, 256), ("1", 12, 128), ("64", 9, 256)):
                monkeypatch.setenv("GS_EMU_UBR_TRIM", trim)
                ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes, q, vsf, rk, False, pair=2, ub8=2, v1_log2=v1, cand_cap=cc)
                check(ids, sc, st, status, wi, ws, wst)
                scored_total += int(wst[:, 0].sum())
                dropped_total += run_emu.last_ub8_dropped
    assert dropped_total > 0.1 * scored_total, (dropped_total, scored_total)
    if M == 96 and deg == 64:
        codes2 = codes.copy()
        codes2[1::2] = codes2[0:-1:2][: len(codes2[1::2])]
        og2 = O.OracleGraph(codes2.shape[0], lv, entry, entry_level)
        monkeypatch.setenv("GS_EMU_UBR_TRIM", "8")
        for order in ("", "reverse", "random:5"):
            if order:
                monkeypatch.setenv("EMU_LANE_ORDER", order)
            for vsf in (O.EUCLIDEAN, O.DOT_PRODUCT, O.COSINE):
                wi, ws, wst = og2.search(opq, codes2, None, q, vsf, 40, 40, fused=False)
                ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes2, q, vsf, 40, False, pair=2, ub8=2)
                check(ids, sc, st, status, wi, ws, wst)
        monkeypatch.delenv("EMU_LANE_ORDER", raising=False)


def test_partition_and_spill_paths(emu):
    """rerankK large enough that far more than cand_cap=256 candidates are alive: the LDS tier must spill (several
    partitions per query) and results must not change."""
    lv, entry, entry_level, opq, codes, q = problem(7, 4000, 128, 16, 2, deg=24, nq=6)
    og = O.OracleGraph(codes.shape[0], lv, entry, entry_level)
    wi, ws, wst = og.search(opq, codes, None, q, O.COSINE, 400, 400, fused=True)
    assert (wst[:, 0] - wst[:, 1]).min() > 2 * 256  # live candidates (pushed - popped) >> cand_cap: the tier must spill
    ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes, q, O.COSINE, 400, True, cand_cap=256)
    check(ids, sc, st, status, wi, ws, wst)


def test_exhaustive_search_pops_from_the_spill_tier(emu):
    """rerankK >= N: the search never stops early, visits the whole component and pops every candidate — the LDS tier
    drains and the remaining candidates come back out of the spill tier."""
    lv, entry, entry_level, opq, codes, q = problem(11, 700, 128, 16, 2, deg=12, nq=4)
    og = O.OracleGraph(codes.shape[0], lv, entry, entry_level)
    wi, ws, wst = og.search(opq, codes, None, q, O.EUCLIDEAN, 800, 800, fused=False)
    assert (wst[:, 1] > 600).all()  # expanded (nearly) every node
    ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes, q, O.EUCLIDEAN, 800, False, cand_cap=256,
                                     vcap_log2=12)
    check(ids, sc, st, status, wi, ws, wst)


@pytest.mark.parametrize("v1_log2,idbits_extra", [(0, 0), (6, 0), (8, 0), (10, 0), (12, 0), (12, 12), (9, 4), (15, 0)])
def test_two_tier_visited_set(emu, v1_log2, idbits_extra):
    """the visited set's LDS tier at every size class: absent; tiny (64 / 256 slots: frozen after a few expansions, nearly
    everything lives in tier 2); medium (some queries stay inside it, others freeze it); large (tier 2 is never touched: its
    garbage-filled table must not matter); with more id bits than the graph needs (12 slot bits + 12 remainder bits = the
    headline shape's entry format: 4 displacement bits, so probes run out of displacement long before the tier is full)"""
    lv, entry, entry_level, opq, codes, q = problem(29, 3000, 128, 16, 2, deg=24, nq=8)
    og = O.OracleGraph(codes.shape[0], lv, entry, entry_level)
    idbits = max(1, int(codes.shape[0] - 1).bit_length()) + idbits_extra
    for vsf, fused, rk, pair in ((O.COSINE, True, 120, 1), (O.EUCLIDEAN, False, 60, 0), (O.DOT_PRODUCT, True, 1, 1)):
        wi, ws, wst = og.search(opq, codes, None, q, vsf, rk, rk, fused=fused)
        ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes, q, vsf, rk, fused, pair=pair, v1_log2=v1_log2,
                                         v1_idbits=idbits, cand_cap=128, evict_cap=64)
        check(ids, sc, st, status, wi, ws, wst)


def test_two_tier_visited_set_under_lane_reordering(emu, monkeypatch):
    """the LDS tier's CAS loop (two 16-bit entries share a word) under reversed / shuffled lane schedules"""
    lv, entry, entry_level, opq, codes, q = problem(31, 2500, 128, 16, 2, deg=32, nq=5)
    og = O.OracleGraph(codes.shape[0], lv, entry, entry_level)
    wi, ws, wst = og.search(opq, codes, None, q, O.COSINE, 80, 80, fused=True)
    for order in ("reverse", "random:3", "random:4"):
        monkeypatch.setenv("EMU_LANE_ORDER", order)
        for v1 in (7, 11):
            ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes, q, O.COSINE, 80, True, v1_log2=v1)
            check(ids, sc, st, status, wi, ws, wst)


def test_overflow_is_reported_not_hidden(emu):
    lv, entry, entry_level, opq, codes, q = problem(13, 3000, 128, 16, 2, deg=24, nq=4)
    og = O.OracleGraph(codes.shape[0], lv, entry, entry_level)
    wi, ws, wst = og.search(opq, codes, None, q, O.DOT_PRODUCT, 120, 120, fused=True)
    # visited table too small for these searches: every query must say so
    for v1 in (0, 6):  # (a 64-slot LDS tier holds 48 of them; the rest still overflows tier 2)
        ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes, q, O.DOT_PRODUCT, 120, True, vcap_log2=9, v1_log2=v1)
        assert (status == 1).all() and (ids == -1).all()
    # spill tier too small
    ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes, q, O.DOT_PRODUCT, 120, True, spill_cap=32)
    assert (status == 1).any()
    check(ids, sc, st, status, wi, ws, wst, allow_overflow=True)  # queries that fitted are still exact
    # and with room the same queries are fine
    ids, sc, st, status, _ = run_emu(emu, lv, entry, entry_level, opq, codes, q, O.DOT_PRODUCT, 120, True)
    check(ids, sc, st, status, wi, ws, wst)
