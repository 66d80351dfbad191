"""Batched robust prune (jv_hip_retain_diverse, csrc/rd_body.h) — VamanaDiversityProvider.retainDiverse
(B/graph/diversity/VamanaDiversityProvider.java:43-96) with the PQ diversity score of BuildScoreProvider.pqBuildScoreProvider:
the selected sets, nSelected and the short-edge fraction must equal the oracle's line-by-line restatement of the reference's
sequential loop (oracle.retain_diverse) for every node of the batch.

* CPU: the kernel body compiled unchanged for the 64-lane emulator (tests/emu/rd_emu.cpp), three lane-scheduling orders.
* GPU (-m gpu): the same cases through the C ABI on the MI355X, incl. the C3 / C5 shapes (M = 96 / 192, maxDegree 32).
The reference ships no literal prune fixtures (its tests check graph connectivity / recall only): parity is pinned on the
restatement, like the other PQ rows."""
import ctypes as C
import os
import platform
import subprocess

import numpy as np
import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_case(seed, N, D, M, P, Cn, vsf, max_degree, dup=True):
    """P nodes with Cn candidates each, scored with the node's own PQ diversity score and sorted descending — what
    GraphIndexBuilder hands to retainDiverse; clustered data so that the alpha rule really prunes; engineered duplicates
    (the same node twice in a list: isDiverse's `node == otherNode -> break`) and exact score ties."""
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((12, D)).astype(np.float32)
    v = (centers[rng.integers(0, 12, N)] + 0.35 * rng.standard_normal((N, D))).astype(np.float32)
    v[N // 3] = v[N // 3 + 1]                                      # identical vectors -> identical codes -> score ties
    sizes, offs = O.subvector_sizes_offsets(D, M)
    pick = rng.choice(N, 256, replace=False)
    cb = np.concatenate([v[pick, offs[m]: offs[m] + sizes[m]].reshape(-1) for m in range(M)])
    opq = O.OraclePQ(D, M, cb)
    codes = opq.encode_all(v)
    tri = opq.codebook_partial_sums(vsf)
    base = rng.choice(N, P, replace=False).astype(np.int32)
    cand = np.full((P, Cn), -1, np.int32)
    sc = np.full((P, Cn), -np.inf, np.float32)
    count = rng.integers(max(1, Cn // 3), Cn + 1, P).astype(np.int32)
    count[0] = Cn
    if P > 2:
        count[1] = 1
        count[2] = 0
    for p in range(P):
        n = int(count[p])
        if n == 0:
            continue
        c = rng.choice(np.delete(np.arange(N), base[p]), n, replace=False).astype(np.int32)
        if dup and n >= 6:
            c[5] = c[1]                                             # a node listed twice
        s = np.array([opq.diversity_score(tri, vsf, codes[base[p]], codes[x]) for x in c], np.float32)
        order = np.argsort(-s, kind="stable")
        cand[p, :n], sc[p, :n] = c[order], s[order]
    before = np.zeros(P, np.int32)
    if P > 4:
        before[3], before[4] = 2, min(max_degree + 3, int(count[4]))  # pre-selected prefixes, one longer than maxDegree
    return opq, codes, tri, cand, sc, count, before


def oracle_selection(opq, codes, tri, vsf, cand, sc, count, before, max_degree, alpha):
    P = cand.shape[0]
    sel = np.full((P, max_degree), -1, np.int32)
    nsel = np.zeros(P, np.int32)
    se = np.zeros(P, np.float32)
    for p in range(P):
        n = int(count[p])
        mask, k, short = opq.retain_diverse(tri, vsf, codes, cand[p, :n], sc[p, :n], max_degree, int(before[p]), alpha)
        idx = np.nonzero(mask)[0][:max_degree]
        sel[p, :len(idx)] = idx
        nsel[p], se[p] = k, np.float32(short)
    return sel, nsel, se


CASES = [(1, 600, 64, 8, 9, 40, 16, 1.2), (2, 800, 128, 16, 7, 70, 32, 1.2), (3, 500, 64, 8, 6, 130, 8, 1.4), (4, 400, 96, 12, 5, 20, 64, 1.0),
         (5, 400, 40, 5, 6, 50, 12, 1.2), (6, 500, 152, 19, 5, 60, 32, 1.2),   # (M not a multiple of 4: padded code rows; 16 + 3 subspaces)
         # M a multiple of 32 / 48 / 96: a test spreads a slot's entries over 2 ... 6 lanes while few slots are selected
         (7, 400, 256, 32, 6, 60, 32, 1.2), (8, 300, 768, 96, 5, 80, 32, 1.2), (9, 300, 384, 48, 5, 40, 16, 1.2), (10, 300, 1536, 192, 4, 70, 32, 1.2)]


# ---- CPU: lane emulator ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def emu():
    if platform.machine() != "x86_64":
        pytest.skip("the lane emulator's context switch is x86-64 assembly")
    src = [os.path.join(ROOT, "tests", "emu", "rd_emu.cpp"), os.path.join(ROOT, "tests", "emu", "hip_emu.h"),
           os.path.join(ROOT, "jvector_amd", "csrc", "rd_body.h"), os.path.join(ROOT, "jvector_amd", "csrc", "rd_params.h")]
    lib = os.path.join(ROOT, "build", "emu", "librd_emu.so")
    if not os.path.exists(lib) or any(os.path.getmtime(s) > os.path.getmtime(lib) for s in src):
        os.makedirs(os.path.dirname(lib), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", src[0], "-o", lib])
    L = C.CDLL(lib)
    L.rd_emu_run.restype = C.c_int
    L.rd_emu_run_tf.restype = C.c_int
    return L


@pytest.mark.parametrize("table_free,chunk,split", [(False, 0, 1), (False, 0, 0), (False, 8, 1), (True, 8, 1), (False, 1, 0), (False, 3, 1), (True, 64, 0),
                                                    (False, 0, "square"), (False, 8, "square")])
@pytest.mark.parametrize("seed,N,D,M,P,Cn,max_degree,alpha", CASES)
def test_retain_diverse_emulated(emu, monkeypatch, seed, N, D, M, P, Cn, max_degree, alpha, table_free, chunk, split):
    """table_free: rd_node<true> — the pair-table entries recomputed from the codebook (uniform 8-dimensional sub-vectors: every
    case here), bit for bit what the table holds, so the selections cannot differ"""
    # chunk: the incremental walk of a test over the selected slots (0: every test examines every slot — the rounds 2-3 form;
    # 1: the reference's own slot-by-slot walk; 64: no chunking, only the memory of earlier tests)
    monkeypatch.setenv("RD_EMU_CHUNK", str(chunk))
    # "square": the pair table as [M][k][k] (rd_node<.., SQ>: a test's lanes share one row per subspace), with the split sums on
    monkeypatch.setenv("RD_EMU_SQUARE", "1" if split == "square" else "0")
    split = 1 if split == "square" else split
    monkeypatch.setenv("RD_EMU_WIDE", str(1 - (seed + chunk) % 2 if split else 1))   # row-by-row staging in some of the runs
    monkeypatch.setenv("RD_EMU_SPLIT", str(split))   # idle lanes share a slot's entries (chunk 0 / duplicate-node tests; M % 16 == 0)
    monkeypatch.setenv("EMU_LANE_ORDER", ["", "reverse", "random:3"][seed % 3])
    for vsf in (O.EUCLIDEAN, O.DOT_PRODUCT, O.COSINE):
        opq, codes, tri, cand, sc, count, before = make_case(seed * 10 + vsf, N, D, M, P, Cn, vsf, max_degree)
        want = oracle_selection(opq, codes, tri, vsf, cand, sc, count, before, max_degree, alpha)
        sel = np.full((P, max_degree), -7, np.int32)
        nsel = np.full(P, -7, np.int32)
        se = np.zeros(P, np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        if table_free:
            assert D == 8 * M
            cb = np.ascontiguousarray(opq.codebooks, np.float32)
            emu.rd_emu_run_tf(p(tri), p(cb), p(codes), C.c_int64(N), p(cand), p(sc), p(count), p(before), P, Cn, M, 256, int(vsf), max_degree,
                              C.c_float(alpha), p(sel), p(nsel), p(se))
        else:
            emu.rd_emu_run(p(tri), p(codes), C.c_int64(N), p(cand), p(sc), p(count), p(before), P, Cn, M, 256, int(vsf), max_degree,
                           C.c_float(alpha), p(sel), p(nsel), p(se))
        assert np.array_equal(sel, want[0]), (vsf, sel, want[0])
        assert np.array_equal(nsel, want[1]) and np.array_equal(se, want[2], equal_nan=True), vsf
        assert (nsel > 1).any()                                      # the prune really selects ...
        skipped = [p_ for p_ in range(P) if before[p_] == 0 and nsel[p_] > 0 and
                   not np.array_equal(sel[p_, :min(nsel[p_], max_degree)], np.arange(min(nsel[p_], max_degree)))]
        assert skipped or alpha == 1.0                               # ... and really rejects (not just the first k candidates)


# ---- GPU ------------------------------------------------------------------------------------------------------------
def run_through_cabi(J, ctx, cases):
    for (seed, N, D, M, P, Cn, max_degree, alpha) in cases:
        for vsf in J.VectorSimilarityFunction:
            opq, codes, tri, cand, sc, count, before = make_case(seed * 10 + int(vsf), N, D, M, P, Cn, int(vsf), max_degree)
            want = oracle_selection(opq, codes, tri, int(vsf), cand, sc, count, before, max_degree, alpha)
            pq = J.ProductQuantization.from_codebooks(ctx, D, M, opq.codebooks)
            cv = J.PQVectors(ctx, pq, codes)
            bsp = J.PQBuildScoreProvider(ctx, cv, vsf)
            sel, nsel, se = bsp.retain_diverse(cand, sc, max_degree, alpha, cand_count=count, diverse_before=before)
            assert np.array_equal(sel, want[0]) and np.array_equal(nsel, want[1]), (seed, vsf)
            assert np.array_equal(se, want[2], equal_nan=True), (seed, vsf)
            bsp.close()


@pytest.mark.gpu
def test_retain_diverse_gpu():
    import jvector_amd as J
    ctx = J.HipContext(0)
    run_through_cabi(J, ctx, CASES + [(15, 3000, 768, 96, 64, 100, 32, 1.2), (16, 1500, 1536, 192, 32, 100, 32, 1.2)])
    ctx.set_option("rd_split", 0)        # one lane per selected slot (the default spreads a slot's entries over the test's idle lanes)
    try:
        run_through_cabi(J, ctx, CASES[1:2] + CASES[6:] + [(15, 3000, 768, 96, 64, 100, 32, 1.2), (16, 1500, 1536, 192, 32, 100, 32, 1.2)])
    finally:
        ctx.set_option("rd_split", None)
    # the measured-and-switched-off forms (square pair table, table-free entries, incremental test walk) are compiled into experimental
    # builds only (make EXPERIMENTAL=1); the default library accepts and ignores their options — the selections are the same either way
    for opt, cases in (("rd_square", CASES[1:3] + CASES[6:8] + [(15, 3000, 768, 96, 64, 100, 32, 1.2)]),
                       ("rd_table_free", CASES[:2] + [(15, 3000, 768, 96, 64, 100, 32, 1.2), (16, 1500, 1536, 192, 32, 100, 32, 1.2)]),
                       ("rd_chunk", CASES[1:3])):
        if not ctx.stat("experimental_build") and opt != "rd_square":
            continue
        ctx.set_option(opt, 8 if opt == "rd_chunk" else 1)
        try:
            run_through_cabi(J, ctx, cases)
        finally:
            ctx.set_option(opt, None)
    # unsupported: candidate codes that do not fit LDS (1000 x 192 B > 160 KB) are refused, not truncated
    opq, codes, tri, cand, sc, count, before = make_case(77, 1200, 1536, 192, 2, 1000, 0, 32)
    pq = J.ProductQuantization.from_codebooks(ctx, 1536, 192, opq.codebooks)
    bsp = J.PQBuildScoreProvider(ctx, J.PQVectors(ctx, pq, codes), J.VectorSimilarityFunction.EUCLIDEAN)
    with pytest.raises(J.UnsupportedError):
        bsp.retain_diverse(cand, sc, 32, 1.2)
    ctx.close()
