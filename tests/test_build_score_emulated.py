"""CPU check of the build-time scoring kernels (jvector_amd/csrc/bs_body.h): the kernel bodies are compiled
as plain loops (one iteration per GPU thread, tests/emu/bs_emu.cpp) and compared bit for bit with the oracle's
restatement of ProductQuantization.createCodebookPartialSums, ImmutablePQVectors.diversityFunctionFor,
ProductQuantization.decode and PQVectors.scoreFunctionFor.  GPU twin: tests/test_zz_build_score_gpu.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "tests", "emu", "bs_emu.cpp"), os.path.join(ROOT, "jvector_amd", "csrc", "bs_body.h")]
LIB = os.path.join(ROOT, "build", "emu", "libbs_emu.so")


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", SRC[0], "-o", LIB])
    return C.CDLL(LIB)


def make_pq(D, M, seed, centroid=False):
    rng = np.random.default_rng(seed)
    sizes, offs = O.subvector_sizes_offsets(D, M)
    cb = np.concatenate([rng.standard_normal(256 * s).astype(np.float32) for s in sizes])
    c = rng.standard_normal(D).astype(np.float32) if centroid else None
    pq = O.OraclePQ(D, M, cb, c)
    cb_offsets = np.concatenate([[0], np.cumsum(256 * np.asarray(sizes[:-1], np.int64))]).astype(np.int64)
    return pq, np.asarray(sizes, np.int32), np.asarray(offs, np.int32), cb_offsets


P = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("D,M", [(64, 8), (50, 7), (96, 96), (128, 16)])
def test_pair_table_and_scores(emu, D, M):
    pq, sizes, offs, cbo = make_pq(D, M, D + M)
    rng = np.random.default_rng(1)
    n = 300
    codes = rng.integers(0, 256, (n, M), dtype=np.uint8)
    codes[5] = codes[4]  # identical codes: cosine of a vector with itself
    node1 = rng.integers(0, n, 40).astype(np.int32)
    node2 = rng.integers(0, n, (40, 9)).astype(np.int32)
    node2[3, 2], node2[7, 0], node1[11] = -1, n, -1
    node1[0], node2[0, 0] = 4, 5
    for vsf in (O.EUCLIDEAN, O.DOT_PRODUCT, O.COSINE):
        want_tri = pq.codebook_partial_sums(vsf)
        tri = np.full_like(want_tri, np.nan)
        emu.bs_emu_pair_table(P(pq.codebooks), P(cbo), P(sizes), P(offs), D, M, 256, int(vsf), P(tri))
        assert np.array_equal(tri, want_tri)
        out = np.empty((40, 9), np.float32)
        emu.bs_emu_pair_scores(P(tri), int(vsf), M, 256, P(codes), C.c_int64(n), P(node1), 40, P(node2), 9, P(out))
        for p in range(40):
            for b in range(9):
                bad = node1[p] < 0 or node2[p, b] < 0 or node2[p, b] >= n
                want = -np.inf if bad else np.float32(pq.diversity_score(tri, vsf, codes[node1[p]], codes[node2[p, b]]))
                assert out[p, b] == want, (vsf, p, b)
                if not bad:  # the MutablePQVectors path (straight from the codebooks) agrees
                    assert out[p, b] == np.float32(pq.diversity_score_direct(vsf, codes[node1[p]], codes[node2[p, b]]))


@pytest.mark.parametrize("D,M,centroid", [(64, 8, False), (50, 7, True), (128, 16, True)])
def test_decode_and_direct_scores(emu, D, M, centroid):
    pq, sizes, offs, cbo = make_pq(D, M, 3 * D + M, centroid)
    rng = np.random.default_rng(2)
    n = 200
    codes = rng.integers(0, 256, (n, M), dtype=np.uint8)
    ords = np.array([0, 7, 199, 7, -1, 200], np.int32)
    out = np.full((len(ords), D), np.nan, np.float32)
    emu.bs_emu_decode(P(pq.codebooks), P(cbo), P(sizes), P(offs), P(pq.centroid), D, M, 256, P(codes), C.c_int64(n), P(ords),
                      C.c_int64(0), C.c_int64(len(ords)), P(out))
    for i, o in enumerate(ords):
        want = pq.decode(codes[o]) if 0 <= o < n else np.zeros(D, np.float32)
        assert np.array_equal(out[i], want), i
    rng_out = np.empty((10, D), np.float32)
    emu.bs_emu_decode(P(pq.codebooks), P(cbo), P(sizes), P(offs), P(pq.centroid), D, M, 256, P(codes), C.c_int64(n), None,
                      C.c_int64(50), C.c_int64(10), P(rng_out))
    assert np.array_equal(rng_out, np.stack([pq.decode(codes[50 + i]) for i in range(10)]))

    Q, B = 6, 11
    q = rng.standard_normal((Q, D)).astype(np.float32)
    cq = q if pq.centroid is None else (q - pq.centroid).astype(np.float32)
    cq = np.ascontiguousarray(cq)
    o2 = rng.integers(0, n, (Q, B)).astype(np.int32)
    o2[1, 1] = -1
    for vsf in (O.EUCLIDEAN, O.DOT_PRODUCT, O.COSINE):
        sc = np.empty((Q, B), np.float32)
        qn = np.empty(Q, np.float32)
        emu.bs_emu_direct_scores(P(pq.codebooks), P(cbo), P(sizes), P(offs), D, M, 256, int(vsf), P(codes), C.c_int64(n), P(cq), Q,
                                 P(o2), B, P(qn), P(sc))
        for i in range(Q):
            for b in range(B):
                want = -np.inf if o2[i, b] < 0 else np.float32(pq.direct_score(q[i], vsf, codes[o2[i, b]]))
                assert sc[i, b] == want, (vsf, i, b)


@pytest.mark.parametrize("M,chunk", [(16, 16), (96, 16), (7, 1)])
def test_fused_block_gather(emu, M, chunk):
    """FusedPQ.writeInline as a per-thread gather (bs_fused_gather): neighbour codes in neighbour order, zero padding."""
    rng = np.random.default_rng(M)
    n, deg = 50, 6
    codes = rng.integers(1, 256, (n, M), dtype=np.uint8)
    nb = np.full((n, deg), -1, np.int32)
    for i in range(n):
        d = int(rng.integers(0, deg + 1))
        nb[i, :d] = rng.integers(0, n, d)
    nb[3, 1] = n  # out of range id: treated like padding
    blocks = np.full((n, deg, M), 0xEE, np.uint8)
    emu.bs_emu_fused_gather(P(codes), C.c_int64(n), P(nb), deg, M, chunk, C.c_int64(n), P(blocks))
    want = np.where(((nb >= 0) & (nb < n))[:, :, None], codes[np.clip(nb, 0, n - 1)], 0).astype(np.uint8)
    assert np.array_equal(blocks, want)
