"""The SIMD leg of bench.py's cpu_baseline (oracle/jv_oracle_simd.c — an AVX2 / AVX-512 restatement of the reference's
native kernels) against the scalar checker, at the tolerance the reference's own native tests use for the same
comparison (1e-4 relative, jvector-native/src/main/native/tests/test_similarity.cpp:54-80), on the reference's
known-answer generator make_vec (tests/test_helpers.cpp:78-87) and its 19 lengths (:49-76).  Both ISA tiers are
exercised: the one this CPU selects in-process, the AVX2 one in a child process (JVO_SIMD_TIER caps the tier the way
JVECTOR_MAX_ISA caps the reference's).  The search entry points must give the scalar checker's answers bit for bit
whenever the switch is off — it is the parity checker — and essentially the same top-k when it is on."""
import os
import platform
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.skipif(platform.machine() != "x86_64", reason="x86 intrinsics")

LENGTHS = [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 128, 1021]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def close(a, b, tol=1e-4):
    return abs(a - b) <= tol * max(1.0, abs(b))


def check_kernels():
    """Runs in-process and (capped to AVX2) in a child; returns the tier name that was exercised."""
    L = O.lib()
    tier = L.jvs_tier_name().decode()
    f, u8 = O._f, O._u8
    for n in LENGTHS + [1536]:
        a, b = O.make_vec(n, 1.0), O.make_vec(n, 2.5)
        assert close(L.jvs_dot(f(a), f(b), n), L.jvo_dot(f(a), f(b), n)), ("dot", n, tier)
        assert close(L.jvs_l2(f(a), f(b), n), L.jvo_l2(f(a), f(b), n)), ("l2", n, tier)
        assert close(L.jvs_cosine(f(a), f(b), n), L.jvo_cosine(f(a), f(b), n)), ("cosine", n, tier)
    rng = np.random.default_rng(5)
    k = 256
    for size in (8, 4, 9, 16):          # 8 = every BASELINE config (the packed path); others: per-centroid distances
        cb = rng.standard_normal(k * size).astype(np.float32)
        q = rng.standard_normal(40).astype(np.float32)
        for vsf in (O.DOT_PRODUCT, O.EUCLIDEAN):
            want = np.zeros(3 * k, np.float32)
            got = np.zeros(3 * k, np.float32)
            L.jvo_calculate_partial_sums(f(cb), 2, size, k, f(q), 5, vsf, f(want))
            L.jvs_calculate_partial_sums(f(cb), 2, size, k, f(q), 5, vsf, f(got))
            assert not got[:2 * k].any()                    # only block cbIndex is written
            np.testing.assert_allclose(got[2 * k:], want[2 * k:], rtol=1e-5, atol=1e-5)
    for M in (7, 16, 96, 100, 192):
        lut = rng.standard_normal(M * k).astype(np.float32)
        amag = (rng.random(M * k) + 0.1).astype(np.float32)
        for _ in range(5):
            code = rng.integers(0, 256, M).astype(np.uint8)
            assert close(L.jvs_assemble_and_sum(f(lut), k, u8(code), M), L.jvo_assemble_and_sum(f(lut), k, u8(code), 0, M))
            assert close(L.jvs_pq_decoded_cosine(u8(code), M, k, f(lut), f(amag), 3.0),
                         L.jvo_pq_decoded_cosine(u8(code), 0, M, k, f(lut), f(amag), 3.0))
    return tier


def test_simd_kernels_match_the_scalar_checker():
    tier = check_kernels()
    assert tier in ("avx512", "avx2", "scalar")


def test_avx2_tier_in_a_child_process():
    if O.lib().jvs_tier() < 3:
        pytest.skip("this CPU already runs the AVX2 (or scalar) tier in-process")
    env = dict(os.environ, JVO_SIMD_TIER="avx2", PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "tests"))
    out = subprocess.check_output([sys.executable, "-c", "import test_oracle_simd_cpu as t; print(t.check_kernels())"], env=env,
                                  cwd=ROOT, text=True)
    assert out.strip().endswith("avx2")


@pytest.mark.parametrize("vsf", [O.EUCLIDEAN, O.DOT_PRODUCT, O.COSINE])
def test_search_entry_points_with_the_switch_off_and_on(vsf):
    from test_graph_search import build_problem
    v, lv, entry, entry_level, cb, q = build_problem(21, N=3000, D=64, M=8)
    opq = O.OraclePQ(64, 8, cb)
    codes = opq.encode_all(v)
    og = O.OracleGraph(v.shape[0], lv, entry, entry_level)
    base_g = og.search(opq, codes, v, q, vsf, 10, 40, fused=True)
    base_f = opq.search_flat(codes, v, q, vsf, 10, 40, nthreads=2)
    # the cached partialSquaredMagnitudes table changes nothing (same values, built once instead of per query)
    opq.cache_self_magnitudes()
    for a, b in zip(base_g, og.search(opq, codes, v, q, vsf, 10, 40, fused=True)):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(base_f, opq.search_flat(codes, v, q, vsf, 10, 40, nthreads=2)):
        np.testing.assert_array_equal(a, b)
    try:
        tier = O.set_simd(True)
        simd_g = og.search(opq, codes, v, q, vsf, 10, 40, fused=True)
        simd_f = opq.search_flat(codes, v, q, vsf, 10, 40, nthreads=2)
    finally:
        assert O.set_simd(False) == "scalar"
    if tier == "scalar":
        pytest.skip("no AVX2 on this CPU")
    for base, simd in ((base_g, simd_g), (base_f, simd_f)):
        overlap = np.mean([len(set(a) & set(b)) / 10.0 for a, b in zip(base[0], simd[0])])
        assert overlap >= 0.99, overlap
        same = base[0] == simd[0]
        np.testing.assert_allclose(simd[1][same], base[1][same], rtol=1e-4, atol=1e-6)
    # and the switch really is off again: the checker's answers, bit for bit
    for a, b in zip(base_g[:2], og.search(opq, codes, v, q, vsf, 10, 40, fused=True)[:2]):
        np.testing.assert_array_equal(a, b)
