"""CPU check of the MFMA tile form of full-resolution scoring (jvector_amd/csrc/ed_body.h — the body of
exact_dense_kernel): the kernel source is compiled unchanged for the lane emulator (a 4-wavefront workgroup), whose MFMA is the documented
v_mfma_f32_32x32x2_f32 (lane -> operand / accumulator maps, k-ordered f32 fmaf chain), and must reproduce the
k-ascending fmaf-chain specification (oracle.dense_scan) BIT FOR BIT on ragged shapes (Q, N, D not multiples of the
128 x 128 x 32 tile), sit within 1e-5 of the bit-exact scalar-order scores, and pass the identity-times-asymmetric-matrix
probe that catches a transposed accumulator unpack.  The GPU twin is tests/test_zz_exact_dense_gpu.py."""
import ctypes as C
import os
import platform
import subprocess

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.skipif(platform.machine() != "x86_64", reason="the lane emulator's context switch is x86-64 assembly")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "tests", "emu", "ed_emu.cpp"), os.path.join(ROOT, "tests", "emu", "hip_emu.h"),
       os.path.join(ROOT, "jvector_amd", "csrc", "ed_body.h")]
LIB = os.path.join(ROOT, "build", "emu", "libed_emu.so")


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas",
                               SRC[0], "-o", LIB])
    lib = C.CDLL(LIB)
    lib.ed_emu_scan.restype = C.c_long
    lib.ed_emu_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int]
    return lib


def run(emu, vecs, queries, vsf, first=0, count=None):
    vecs, queries = O.f32(vecs), O.f32(queries)
    count = vecs.shape[0] - first if count is None else count
    out = np.full((queries.shape[0], count), np.float32(-7.0))
    n = emu.ed_emu_scan(vecs.ctypes.data, queries.ctypes.data, out.ctypes.data, first, count, vecs.shape[1], queries.shape[0],
                        int(vsf))
    assert n == -(-count // 128) * -(-queries.shape[0] // 128), "block -> tile map must cover every tile exactly once"
    return out


@pytest.mark.parametrize("Q,N,D", [(1, 1, 1), (5, 130, 7), (33, 129, 100), (40, 300, 64), (64, 256, 96), (3, 77, 129), (130, 140, 36), (257, 64, 8)])
def test_dense_tile_equals_the_fma_chain_specification(emu, Q, N, D, monkeypatch):
    monkeypatch.setenv("EMU_LANE_ORDER", ["", "reverse", "random:5"][(Q + N) % 3])
    rng = np.random.default_rng(Q * 1000 + N)
    v = rng.standard_normal((N, D)).astype(np.float32)
    q = rng.standard_normal((Q, D)).astype(np.float32)
    q[0] = v[N // 2]                                               # an exact hit: cosine 1, L2 distance ~0 (the clamp)
    for vsf in (O.EUCLIDEAN, O.DOT_PRODUCT, O.COSINE):
        got = run(emu, v, q, vsf)
        want = O.dense_scan(vsf, q, v)
        assert np.array_equal(got, want), (vsf, np.abs(got - want).max())
        exact = np.array([[O.compare(vsf, q[i], v[j]) for j in range(N)] for i in range(Q)], np.float32)
        # vs the bit-exact scalar-order path: 1e-5 relative (north_star); a dot product that cancels to ~0 is held to the
        # same bound relative to its operands' magnitude |q||v| instead (no summation order can do better)
        scale = float(np.linalg.norm(q, axis=1).max() * np.linalg.norm(v, axis=1).max()) if vsf == O.DOT_PRODUCT else 1.0
        np.testing.assert_allclose(got, exact, rtol=1e-5, atol=1e-6 * scale)


def test_sub_range_and_untouched_output(emu):
    rng = np.random.default_rng(2)
    v = rng.standard_normal((500, 48)).astype(np.float32)
    q = rng.standard_normal((7, 48)).astype(np.float32)
    got = run(emu, v, q, O.DOT_PRODUCT, first=123, count=200)
    assert np.array_equal(got, O.dense_scan(O.DOT_PRODUCT, q, v[123:323]))


def test_identity_times_asymmetric_matrix(emu):
    """queries = rows of the identity, vectors = an asymmetric matrix: out[q][n] must be (1 + B[n][q]) / 2 exactly — a swapped
    row / column in the accumulator unpack or a wrong k pairing cannot pass (cdna_hip_programming.md §3)."""
    D, N = 70, 150
    B = (np.arange(N)[:, None] * 1000 + np.arange(D)[None, :]).astype(np.float32) / 4096.0
    q = np.eye(D, dtype=np.float32)
    got = run(emu, B, q, O.DOT_PRODUCT)
    assert np.array_equal(got, ((1.0 + B.T) / 2.0).astype(np.float32))
