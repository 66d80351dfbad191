"""Reference-EXECUTED pin of the oracle at the C boundary (VERDICT r4 "Next" #2, SURVEY §8c "parity unpinned").

oracle/_ref/libjvector_ref.so is the reference's OWN jvector_simd_kernels.cpp + jvector_simd.cpp, compiled unmodified by
oracle/ref_build/build.sh over a scalar lane emulation of the Highway ops they use (oracle/ref_build/hwy/highway.h).  Three
lane widths are built, as the reference's meson.build does: avx3 (16 f32 lanes, fma), avx2 (8, fma), sse42 (4, no fma).

  1. the shim is validated with the reference's own native test cases, restated (NC/tests/test_similarity.cpp:92-219,
     test_elementwise.cpp:24-180, test_helpers.cpp:49-87 — make_vec and the 19 lengths), at THEIR tolerances;
  2. the 9 hot symbols + 6 element-wise ones + the 7 NVQ symbols: `_ref` (every tier) vs oracle/jv_oracle.c (the Java scalar
     order) vs csrc/compat_host.cpp (the drop-in SPI) at north_star's 1e-5 relative, on the BASELINE shapes
     (sub-vector size 8 with M = 16 / 96 / 192) and ragged ones (sizes 2, 4, 7, 16, 33/34);
  3. ProductQuantization.encode rebuilt from `_ref`'s euclidean_f32 (what NativeVectorUtilSupport.squareDistance calls under
     ProductQuantization.closetCentroidIndex, ProductQuantization.java:586-600): code bytes equal the oracle's wherever the
     two nearest centroids differ by more than 1e-5 relative.

Bit-exactness of the oracle stays relative to the JAVA scalar order (DefaultVectorUtilSupport); the JVM-side pin remains
scripts/pin_oracle.sh.  What this file adds is that every float the oracle produces at this boundary agrees with what the
reference's native code computes, executed here, to the tolerance north_star states."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as O
from oracle import ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = ref.lib()
pytestmark = pytest.mark.skipif(R is None, reason="oracle/_ref/libjvector_ref.so absent and /root/reference not present to build it")

fp, u8 = ref.fp, ref.u8
TIERS = list(ref.TIERS) + [None]          # None = the library's own CPUID-dispatched export
LENGTHS = [1, 3, 4, 5, 7, 8, 9, 15, 16, 17, 19, 32, 33, 37, 64, 71, 100, 128, 255]   # test_helpers.cpp:49-76
F = C.c_float
f32 = np.float32


@pytest.fixture(scope="module")
def compat():
    import jvector_amd
    return jvector_amd.load()


def seq(terms):
    """float32 left-to-right accumulation (the reference tests' ref_dot / ref_euclidean / ref_cosine, test_similarity.cpp:53-80)"""
    s = f32(0)
    for t in terms:
        s = f32(s + t)
    return s


def near(got, want, tol):
    return abs(float(got) - float(want)) <= tol


# ---------------------------------------------------------------------------------------------------------------------------------
# 1. the reference's own native test cases, on the reference's own kernels over the shim
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tier", TIERS)
def test_reference_similarity_cases(tier):
    dot, l2, cos = (R.fn(tier, n) for n in ("dot_product_f32", "euclidean_f32", "cosine_f32"))
    for n in LENGTHS:
        a, b = O.make_vec(n, 0.7), O.make_vec(n, 1.3)
        want = seq(a * b)
        assert near(dot(fp(a), 0, fp(b), 0, n), want, 1e-4 * abs(want)), ("DotProduct", n)
        ap, bp = np.concatenate([np.full(3, 9e9, f32), a]), np.concatenate([np.full(3, -9e9, f32), b])
        assert near(dot(fp(ap), 3, fp(bp), 3, n), want, 1e-4 * abs(want)), ("DotProductWithOffset", n)
        d = a - b
        want = seq(d * d)
        assert near(l2(fp(a), 0, fp(b), 0, n), want, 1e-4 * abs(want)), ("Euclidean", n)
        s = O.make_vec(n, 0.9)
        assert near(l2(fp(s), 0, fp(s), 0, n), 0.0, 1e-6 * n), ("EuclideanSameVector", n)
        want = f32(seq(a * b) / np.sqrt(f32(seq(a * a) * seq(b * b))))
        assert near(cos(fp(a), 0, fp(b), 0, n), want, 1e-4 * abs(want)), ("Cosine", n)
        p = O.make_vec(n, 1.0)
        p2 = (f32(2) * p).astype(f32)
        assert near(cos(fp(p), 0, fp(p2), 0, n), 1.0, 1e-5), ("CosineParallelVectors", n)
        if n >= 2:
            even = n - n % 2
            x, y = np.zeros(n, f32), np.zeros(n, f32)
            x[:even] = 1
            y[:even] = np.where(np.arange(even) % 2 == 0, 1, -1)
            assert near(cos(fp(x), 0, fp(y), 0, n), 0.0, 1e-4), ("CosineOrthogonalVectors", n)


@pytest.mark.parametrize("tier", TIERS)
def test_reference_elementwise_cases(tier):
    g = lambda name: R.fn(tier, name)
    for n in LENGTHS:
        v1, v2 = O.make_vec(n, 1.1), O.make_vec(n, 0.7)
        x = v1.copy(); g("add_in_place_f32")(fp(x), fp(v2), n); assert np.allclose(x, v1 + v2, rtol=0, atol=1e-5)
        x = v1.copy(); g("add_scalar_in_place_f32")(fp(x), F(3.14), n); assert np.allclose(x, v1 + f32(3.14), rtol=0, atol=1e-5)
        x = v1.copy(); g("sub_in_place_f32")(fp(x), fp(v2), n); assert np.allclose(x, v1 - v2, rtol=0, atol=1e-5)
        x = v1.copy(); g("sub_scalar_in_place_f32")(fp(x), F(2.71), n); assert np.allclose(x, v1 - f32(2.71), rtol=0, atol=1e-5)
        x = v1.copy(); g("min_in_place_f32")(fp(x), fp(v2), n); assert np.allclose(x, np.minimum(v1, v2), rtol=0, atol=1e-5)
        v = O.make_vec(n, 0.9)
        assert g("max_f32")(fp(v), n) == v.max()                                         # MaxF32: EXPECT_FLOAT_EQ
        v = O.make_vec(n, 0.5); v[-1] = 1e6
        assert g("max_f32")(fp(v), n) == f32(1e6)                                        # MaxF32TailElement
        o, d = O.make_vec(n, 1.3), O.make_vec(n, 0.4)
        x = o.copy(); g("add_in_place_f32")(fp(x), fp(d), n); g("sub_in_place_f32")(fp(x), fp(d), n)
        assert np.allclose(x, o, rtol=0, atol=1e-5)                                      # AddSubRoundTrip


def test_reference_dispatch_honours_jvector_max_isa():
    """test_similarity.cpp:237-265 (IsaDispatch.MaxIsaEnvHonoured): the reference's CPUID dispatch, run here"""
    order = ["sse42", "avx2", "avx3", "avx3_dl", "avx3_spr"]
    assert R.active_isa() in order
    code = ("from oracle import ref; R = ref.lib(); e = R.dll.jvector_simd_get_max_isa_env(); "
            "print(R.active_isa(), e.decode() if e else None)")
    for cap in ("avx2", "sse42"):
        out = subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, JVECTOR_MAX_ISA=cap, PYTHONPATH=ROOT),
                                      cwd=ROOT, text=True).split()
        assert out[1] == cap and order.index(out[0]) <= order.index(cap)


# ---------------------------------------------------------------------------------------------------------------------------------
# 2. hot symbols: _ref vs oracle vs compat_host
# ---------------------------------------------------------------------------------------------------------------------------------
def compat_fn(compat, name):
    f = getattr(compat, name)
    f.restype, f.argtypes = ref.SIGNATURES[name]
    return f


@pytest.mark.parametrize("n", [128, 768, 1536, 1021, 100, 8, 7, 3])
def test_dot_l2_cosine_three_ways(compat, n):
    """rerank / exact-build scoring at the BASELINE dimensions and ragged ones: raw values within 1e-5 of
    the sum of the term magnitudes, SCORES (VectorSimilarityFunction.compare's transform) within 1e-5 relative"""
    rng = np.random.default_rng(n)
    for trial in range(4):
        a, b = rng.standard_normal(n + 5).astype(f32), rng.standard_normal(n + 5).astype(f32)
        if trial == 1:          # unit vectors (the ada-002-like case)
            a[:n] /= np.linalg.norm(a[:n]); b[:n] /= np.linalg.norm(b[:n])
        if trial == 2:          # near neighbours: b = a + small noise (top-k candidates look like this)
            b[:n] = a[:n] + f32(0.01) * rng.standard_normal(n).astype(f32)
        a64, b64 = a[:n].astype(np.float64), b[:n].astype(np.float64)
        scale = {"dot_product_f32": np.abs(a64 * b64).sum(), "euclidean_f32": ((a64 - b64) ** 2).sum(), "cosine_f32": 1.0}
        wants = {"dot_product_f32": (O.dot(a[:n], b[:n]), O.DOT_PRODUCT), "euclidean_f32": (O.l2(a[:n], b[:n]), O.EUCLIDEAN),
                 "cosine_f32": (O.cosine(a[:n], b[:n]), O.COSINE)}
        for name, (want, vsf) in wants.items():
            exact = {"dot_product_f32": (a64 * b64).sum(), "euclidean_f32": ((a64 - b64) ** 2).sum(),
                     "cosine_f32": (a64 * b64).sum() / np.sqrt((a64 * a64).sum() * (b64 * b64).sum())}[name]
            assert near(want, exact, 1e-5 * max(scale[name], 1e-30)), (name, "oracle vs float64")
            got_c = compat_fn(compat, name)(fp(a), 0, fp(b), 0, n)
            assert got_c == f32(want), (name, "compat_host == oracle bit for bit")
            for tier in TIERS:
                got = R.fn(tier, name)(fp(a), 0, fp(b), 0, n)
                assert near(got, want, 1e-5 * max(scale[name], 1e-30)), (name, tier, n, got, want)
                if vsf != O.DOT_PRODUCT or trial == 1:   # (1 + dot) / 2 is a score only for unit vectors
                    s_ref, s_or = O.score_from_raw(vsf, got), O.score_from_raw(vsf, want)
                    assert near(s_ref, s_or, 1e-5 * abs(s_or)), (name, tier, n, "score")
                # the offset overloads (NativeVectorUtilSupport passes offsets straight through)
                got_off = R.fn(tier, name)(fp(a), 2, fp(b), 4, n)
                want_off = {"dot_product_f32": O.dot_off, "euclidean_f32": O.l2_off, "cosine_f32": O.cosine_off}[name](a, 2, b, 4, n)
                sc = {"dot_product_f32": np.abs(a[2:2 + n].astype(np.float64) * b[4:4 + n]).sum(),
                      "euclidean_f32": ((a[2:2 + n].astype(np.float64) - b[4:4 + n]) ** 2).sum(), "cosine_f32": 1.0}[name]
                assert near(got_off, want_off, 1e-5 * max(sc, 1e-30)), (name, tier, n, "offset")


# (D, M): C2 128/16, C3 768/96, C5 1536/192 — all sub-vector size 8 — then sizes 2, 4, 16, ragged 8/7, ragged 34/33
PQ_SHAPES = [(128, 16), (768, 96), (1536, 192), (32, 16), (64, 16), (256, 16), (50, 7), (100, 3)]


def make_pq(D, M, seed):
    rng = np.random.default_rng(seed)
    sizes, offs = O.subvector_sizes_offsets(D, M)
    # centroids of a unit-norm-ish data set: sub-vector entries ~ N(0, 1/D) scaled up a little so that sums are O(1)
    cb = np.concatenate([(rng.standard_normal(256 * s) * 0.3).astype(f32) for s in sizes])
    return O.OraclePQ(D, M, cb), sizes, offs, rng


def partial_sums(fn, pq, sizes, offs, q, self_mag=False):
    out = np.zeros(pq.M * 256, f32)
    off = 0
    for m in range(pq.M):
        cb = pq.codebooks[off: off + 256 * int(sizes[m])]
        if self_mag:
            fn(fp(cb), m, int(sizes[m]), 256, fp(out))
        else:
            fn(fp(cb), m, int(sizes[m]), 256, fp(q), int(offs[m]), fp(out))
        off += 256 * int(sizes[m])
    return out


@pytest.mark.parametrize("D,M", PQ_SHAPES)
def test_adc_tables_and_lookups_three_ways(compat, D, M):
    """§8a rows 2, 5, 6: calculate_partial_sums_{dot,euclidean,self_magnitude}_f32 -> assemble_and_sum_f32 /
    pq_decoded_cosine_similarity_f32, table by table and then as the whole chain (each implementation on ITS OWN table)"""
    pq, sizes, offs, rng = make_pq(D, M, D * 1000 + M)
    q = rng.standard_normal(D).astype(f32)
    q /= np.linalg.norm(q)
    codes = rng.integers(0, 256, (8, M), dtype=np.uint8)
    cb64 = [pq.codebook(m).astype(np.float64).reshape(256, int(sizes[m])) for m in range(M)]
    q64 = [q[int(offs[m]): int(offs[m]) + int(sizes[m])].astype(np.float64) for m in range(M)]
    tables = {}
    for vsf, name, scale in ((O.DOT_PRODUCT, "calculate_partial_sums_dot_f32", lambda c, x: np.abs(c * x).sum(1)),
                             (O.EUCLIDEAN, "calculate_partial_sums_euclidean_f32", lambda c, x: ((c - x) ** 2).sum(1))):
        want, _, _ = pq.decoder(q, vsf)
        tol = 1e-5 * np.concatenate([scale(cb64[m], q64[m]) for m in range(M)]) + 1e-30
        got_c = partial_sums(compat_fn(compat, name), pq, sizes, offs, q)
        assert np.array_equal(got_c, want), (name, "compat_host == oracle bit for bit")
        tables[(vsf, "oracle")] = want
        for tier in TIERS:
            got = partial_sums(R.fn(tier, name), pq, sizes, offs, q)
            assert (np.abs(got.astype(np.float64) - want) <= tol).all(), (name, tier, np.abs(got - want).max())
            tables[(vsf, tier)] = got
    lut_or, amag_or, bmag = pq.decoder(q, O.COSINE)
    tol = 1e-5 * np.concatenate([(cb64[m] ** 2).sum(1) for m in range(M)]) + 1e-30
    assert np.array_equal(partial_sums(compat_fn(compat, "calculate_partial_sums_self_magnitude_f32"), pq, sizes, offs, None, True), amag_or)
    amags = {"oracle": amag_or}
    for tier in TIERS:
        amags[tier] = partial_sums(R.fn(tier, "calculate_partial_sums_self_magnitude_f32"), pq, sizes, offs, None, True)
        assert (np.abs(amags[tier].astype(np.float64) - amag_or) <= tol).all(), ("self_magnitude", tier)
    assert np.array_equal(lut_or, tables[(O.DOT_PRODUCT, "oracle")])        # the cosine decoder's table IS the dot table
    for c in codes:
        idx = np.arange(M) * 256 + c
        for vsf in (O.DOT_PRODUCT, O.EUCLIDEAN):
            t_or = tables[(vsf, "oracle")]
            want = O.lib().jvo_assemble_and_sum(O._f(t_or), 256, O._u8(c), 0, M)
            assert compat_fn(compat, "assemble_and_sum_f32")(fp(t_or), 256, u8(c), 0, M) == f32(want)
            sc_or = O.score_from_raw(vsf, want)
            for tier in TIERS:
                asum = R.fn(tier, "assemble_and_sum_f32")
                got_same_table = asum(fp(t_or), 256, u8(c), 0, M)                # the lookup kernel alone, same input
                assert near(got_same_table, want, 1e-5 * np.abs(t_or[idx].astype(np.float64)).sum()), ("assemble_and_sum", tier)
                t = tables[(vsf, tier)]
                got_chain = asum(fp(t), 256, u8(c), 0, M)                        # table build + lookup, all reference code
                assert near(got_chain, want, 2e-5 * np.abs(t_or[idx].astype(np.float64)).sum()), ("chain", tier)
                if vsf == O.EUCLIDEAN:
                    assert near(O.score_from_raw(vsf, got_chain), sc_or, 1e-5 * abs(sc_or)), ("chain score", tier)
                # offset form: the code read from the middle of a longer byte array (PQVectors chunks, PQVectors.java:196-215)
                padded = np.concatenate([np.full(5, 255, np.uint8), c])
                assert asum(fp(t_or), 256, u8(padded), 5, M) == got_same_table
        want = O.lib().jvo_pq_decoded_cosine(O._u8(c), 0, M, 256, O._f(lut_or), O._f(amag_or), F(bmag))
        assert compat_fn(compat, "pq_decoded_cosine_similarity_f32")(u8(c), 0, M, 256, fp(lut_or), fp(amag_or), F(bmag)) == f32(want)
        s_or = O.score_from_raw(O.COSINE, want)
        for tier in TIERS:
            cosf = R.fn(tier, "pq_decoded_cosine_similarity_f32")
            got = cosf(u8(c), 0, M, 256, fp(lut_or), fp(amag_or), F(bmag))
            # cosine of a decoded vector: |numerator| can cancel, the magnitudes cannot: absolute 1e-5 on a value in [-1, 1]
            # relative to sum|terms| / sqrt(aMag * bMag)
            cscale = np.abs(lut_or[idx].astype(np.float64)).sum() / np.sqrt(float(amag_or[idx].astype(np.float64).sum()) * bmag)
            assert near(got, want, 1e-5 * max(cscale, 1.0)), ("pq_decoded_cosine", tier, got, want)
            chain = cosf(u8(c), 0, M, 256, fp(tables[(O.DOT_PRODUCT, tier)]), fp(amags[tier]), F(bmag))
            assert near(O.score_from_raw(O.COSINE, chain), s_or, 2e-5 * max(cscale, 1.0)), ("cosine chain score", tier)


@pytest.mark.parametrize("D,M", [(128, 16), (768, 96), (50, 7), (64, 16)])
def test_pair_table_lookup_three_ways(compat, D, M):
    """§8 f2 (build-time diversity scoring): assemble_and_sum_pq_f32 over ImmutablePQVectors' triangular codebook table
    (jvector_simd_kernels.cpp:729-815) — r == c, r < c and r > c entries, offsets, subspace counts below and above a register"""
    pq, sizes, offs, rng = make_pq(D, M, 77 + D)
    codes = rng.integers(0, 256, (10, M), dtype=np.uint8)
    codes[1] = codes[0]                          # identical codes: the diagonal of every triangle
    codes[2, ::2] = 0; codes[3, ::2] = 255       # extremes of the row index
    for vsf in (O.DOT_PRODUCT, O.EUCLIDEAN):
        tri = pq.codebook_partial_sums(vsf)
        for i in range(9):
            c1, c2 = codes[i], codes[i + 1]
            want = O.lib().jvo_assemble_and_sum_pq(O._f(tri), M, O._u8(c1), 0, O._u8(c2), 0, 256)
            assert compat_fn(compat, "assemble_and_sum_pq_f32")(fp(tri), M, u8(c1), 0, u8(c2), 0, 256) == f32(want)
            r, c = np.minimum(c1, c2).astype(np.int64), np.maximum(c1, c2).astype(np.int64)
            idx = np.arange(M) * (256 * 257 // 2) + r * 256 - r * (r - 1) // 2 + (c - r)
            scale = np.abs(tri[idx].astype(np.float64)).sum()
            p1, p2 = np.concatenate([np.zeros(3, np.uint8), c1]), np.concatenate([np.zeros(7, np.uint8), c2])
            for tier in TIERS:
                f = R.fn(tier, "assemble_and_sum_pq_f32")
                got = f(fp(tri), M, u8(c1), 0, u8(c2), 0, 256)
                assert near(got, want, 1e-5 * max(scale, 1e-30)), (tier, vsf, i, got, want)
                assert f(fp(tri), M, u8(c2), 0, u8(c1), 0, 256) == got           # symmetric by construction
                assert f(fp(tri), M, u8(p1), 3, u8(p2), 7, 256) == got           # offsets


# ---------------------------------------------------------------------------------------------------------------------------------
# 3. ProductQuantization.encode over the reference's euclidean_f32
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D,M", [(128, 16), (768, 96), (50, 7)])
def test_pq_encode_code_bytes_over_the_reference_distance(D, M):
    """closetCentroidIndex (ProductQuantization.java:586-600): argmin over the 256 centroids of
    VectorUtil.squareL2Distance(codebook, j * size, vector, offset, size), first minimum wins.  With `_ref`'s euclidean_f32 in
    that loop the code bytes must equal the oracle's (= the HIP encoder's, test_encode_bit_exact) wherever the nearest two
    centroids are more than 1e-5 (relative) apart — where they are closer, either byte is a correct answer of a conforming
    implementation and the oracle follows the Java scalar order."""
    pq, sizes, offs, rng = make_pq(D, M, 5 + D)
    centroid = (rng.standard_normal(D) * 0.05).astype(f32)
    pq = O.OraclePQ(D, M, pq.codebooks, centroid)
    vecs = rng.standard_normal((6 if M > 16 else 24, D)).astype(f32)
    want = pq.encode_all(vecs)
    for tier in ("avx3", "avx2", "sse42"):
        l2 = R.fn(tier, "euclidean_f32")
        checked = ambiguous = 0
        for v, w in zip(vecs, want):
            x = (v - centroid).astype(f32)        # encodeTo centres first (ProductQuantization.java:561-569)
            off = 0
            for m in range(M):
                s = int(sizes[m])
                cb = pq.codebooks[off: off + 256 * s]
                d = np.array([l2(fp(cb), j * s, fp(x), int(offs[m]), s) for j in range(256)], f32)
                best = int(np.argmin(d))           # numpy's argmin takes the first minimum, like the Java loop's strict <
                two = np.partition(d, 1)[:2]
                if two[1] - two[0] > 1e-5 * two[1]:
                    assert best == int(w[m]), (tier, m, d[best], d[int(w[m])])
                    checked += 1
                else:
                    assert d[int(w[m])] - two[0] <= 1e-5 * two[1]     # the oracle's byte is one of the near-tied centroids
                    ambiguous += 1
                off += 256 * s
        assert checked > 0.99 * (checked + ambiguous)


# ---------------------------------------------------------------------------------------------------------------------------------
# 4. the 7 NVQ symbols
# ---------------------------------------------------------------------------------------------------------------------------------
def shuffled(tier, v):
    """NVQScorer shuffles the query sub-vector (and, for cosine, the mean) once per query, NVQScorer.java:53-57,82-86,111-116"""
    x = v.copy()
    R.fn(tier, "nvq_shuffle_query_in_place_8bit")(fp(x), len(x))
    return x


@pytest.mark.parametrize("n", [768, 384, 96, 64, 100, 13])
def test_nvq_symbols_three_ways(compat, n):
    rng = np.random.default_rng(n)
    L = O.lib()
    for trial in range(3):
        v = (rng.standard_normal(n) * 0.1).astype(f32)
        q = (rng.standard_normal(n) * 0.1).astype(f32)
        cen = (rng.standard_normal(n) * 0.02).astype(f32)
        alpha, x0 = f32([1.3, 4.0, 0.8][trial]), f32([0.1, -0.05, 0.0][trial])
        lo, hi = f32(v.min()), f32(v.max())
        args = (F(alpha), F(x0), F(lo), F(hi))
        want_b = np.empty(n, np.uint8)
        L.jvo_nvq_quantize_8bit(O._f(v), n, *args, O._u8(want_b))
        comp_b = np.empty(n, np.uint8)
        compat_fn(compat, "nvq_quantize_8bit")(fp(v), n, *args, u8(comp_b))
        assert np.array_equal(comp_b, want_b)
        want_loss = L.jvo_nvq_loss(O._f(v), n, *args, 8)
        want_uloss = L.jvo_nvq_uniform_loss(O._f(v), n, F(lo), F(hi), 8)
        want_dot = L.jvo_nvq_dot_8bit(O._f(q), O._u8(want_b), n, *args)
        want_l2 = L.jvo_nvq_l2_8bit(O._f(q), O._u8(want_b), n, *args)
        out2 = np.zeros(2, f32)
        L.jvo_nvq_cosine_8bit(O._f(q), O._u8(want_b), n, *args, O._f(cen), O._f(out2))
        cs, cm = F(float(out2[0])), F(float(out2[1]))
        deq = np.array([L.jvo_nvq_dequantize(int(b), *args) for b in want_b], np.float64)
        for tier in TIERS:
            got_b = np.empty(n, np.uint8)
            R.fn(tier, "nvq_quantize_8bit")(fp(v), n, *args, u8(got_b))
            diff = np.abs(got_b.astype(int) - want_b.astype(int))
            # a byte may land on the other side of a rounding boundary when the logistic is evaluated with / without fma
            assert diff.max() <= 1 and (diff != 0).mean() <= 0.02, (tier, n, diff.max(), (diff != 0).mean())
            assert near(R.fn(tier, "nvq_loss")(fp(v), n, *args, 8), want_loss, 1e-3 * want_loss + 1e-9), (tier, "loss")
            assert near(R.fn(tier, "nvq_uniform_loss")(fp(v), n, F(lo), F(hi), 8), want_uloss, 1e-3 * want_uloss + 1e-9), (tier, "uniform")
            t = tier or {"avx3_spr": "avx3", "avx3_dl": "avx3"}.get(R.active_isa(), R.active_isa())
            qs, cens = shuffled(t, q), shuffled(t, cen)
            # the shuffle is a permutation of whole 4*lanes blocks and the identity on the tail (kernels.cpp:1502-1540)
            step = 4 * ref.LANES[t]
            assert np.array_equal(np.sort(qs), np.sort(q)) and np.array_equal(qs[n - n % step:], q[n - n % step:])
            got_dot = R.fn(tier, "nvq_dot_product_8bit")(fp(qs), u8(want_b), n, *args)
            got_l2 = R.fn(tier, "nvq_square_l2_distance_8bit")(fp(qs), u8(want_b), n, *args)
            assert near(got_dot, want_dot, 1e-5 * np.abs(q * deq).sum()), (tier, "dot", got_dot, want_dot)
            assert near(got_l2, want_l2, 1e-5 * ((q - deq) ** 2).sum()), (tier, "l2", got_l2, want_l2)
            s, mag = ref.unpack_cosine(R.fn(tier, "nvq_cosine_8bit_packed")(fp(qs), u8(want_b), n, *args, fp(cens)))
            assert near(s, cs.value, 1e-5 * np.abs(q * (deq + cen)).sum()) and near(mag, cm.value, 1e-5 * ((deq + cen) ** 2).sum()), (tier, "cosine")
        # compat_host: the SPI's shuffle is the identity (scalar order, like DefaultVectorUtilSupport.java:454), values == the oracle's
        x = q.copy(); compat_fn(compat, "nvq_shuffle_query_in_place_8bit")(fp(x), n); assert np.array_equal(x, q)
        assert compat_fn(compat, "nvq_dot_product_8bit")(fp(q), u8(want_b), n, *args) == f32(want_dot)
        assert compat_fn(compat, "nvq_square_l2_distance_8bit")(fp(q), u8(want_b), n, *args) == f32(want_l2)
        assert compat_fn(compat, "nvq_loss")(fp(v), n, *args, 8) == f32(want_loss)
        assert compat_fn(compat, "nvq_uniform_loss")(fp(v), n, F(lo), F(hi), 8) == f32(want_uloss)
        s, mag = ref.unpack_cosine(compat_fn(compat, "nvq_cosine_8bit_packed")(fp(q), u8(want_b), n, *args, fp(cen)))
        assert s == cs.value and mag == cm.value
