"""The batched HIP kernels (through the C ABI) against the reference's OWN native kernels, executed
(oracle/_ref/libjvector_ref.so: jvector_simd_kernels.cpp compiled unmodified over a scalar Highway stand-in, see
tests/test_ref_native_cpu.py), at north_star's tolerance: float distances within 1e-5 relative, PQ code bytes equal wherever
the nearest two centroids are more than 1e-5 apart.  Shapes: C2 (128 / PQ-16), C3 (768 / PQ-96), C5 (1536 / PQ-192) and a
ragged one.  The library was built in the container that holds /root/reference and travels with the snapshot; when it is
missing these tests SKIP (they never fall back to comparing the oracle with itself)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import jvector_amd as J
from jvector_amd import VectorSimilarityFunction as VSF
from oracle import oracle as O
from oracle import ref

R = ref.lib()
fp, u8, f32, F = ref.fp, ref.u8, np.float32, C.c_float
TIERS = ("avx3", "avx2", "sse42")
REL = 1e-5


@pytest.fixture(scope="module")
def ctx():
    if R is None:
        pytest.skip("oracle/_ref/libjvector_ref.so did not travel with the snapshot")
    c = J.HipContext(0)
    yield c
    c.close()


def unit_rows(rng, n, D):
    c = rng.standard_normal((16, D)).astype(f32)
    v = c[rng.integers(0, 16, n)] + 0.4 * rng.standard_normal((n, D)).astype(f32)
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(f32)


@pytest.mark.parametrize("D", [128, 768, 1536, 1021])
def test_exact_scores_vs_reference_native(ctx, D):
    """§8a row 1 (rerank): exact_gather kernels vs cosine_f32 / dot_product_f32 / euclidean_f32 + VectorSimilarityFunction's
    transform (VectorSimilarityFunction.java:37-69) — scores within 1e-5 relative"""
    rng = np.random.default_rng(D)
    N, Q, B = 400, 6, 40
    v = unit_rows(rng, N, D)
    q = (v[rng.integers(0, N, Q)] + 0.05 * rng.standard_normal((Q, D))).astype(f32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    vs = J.VectorSet(ctx, v)
    ords = rng.integers(0, N, (Q, B)).astype(np.int32)
    names = {VSF.EUCLIDEAN: "euclidean_f32", VSF.DOT_PRODUCT: "dot_product_f32", VSF.COSINE: "cosine_f32"}
    for vsf in VSF:
        got = vs.scores(q, vsf, ords)
        for tier in TIERS:
            fn = R.fn(tier, names[vsf])
            want = np.array([[O.score_from_raw(int(vsf), fn(fp(q[i]), 0, fp(v[o]), 0, D)) for o in ords[i]] for i in range(Q)], f32)
            assert np.allclose(got, want, rtol=REL, atol=0), (vsf, tier, np.abs(got - want).max())


@pytest.mark.parametrize("D,M", [(128, 16), (768, 96), (1536, 192), (50, 7)])
def test_adc_tables_and_scores_vs_reference_native(ctx, D, M):
    """§8a rows 2, 5, 6: lut_build_kernel vs calculate_partial_sums_*_f32 entry by entry; the ADC scan vs the reference's
    chain (its own tables -> assemble_and_sum_f32 / pq_decoded_cosine_similarity_f32 -> score transform)"""
    rng = np.random.default_rng(D + M)
    sizes, offs = O.subvector_sizes_offsets(D, M)
    v = unit_rows(rng, 3000, D)
    cb = np.concatenate([v[:256, offs[m]: offs[m] + sizes[m]].reshape(-1) for m in range(M)])   # centroids = data sub-vectors
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    opq = O.OraclePQ(D, M, cb)
    q = (v[rng.integers(0, 3000, 3)] + 0.05 * rng.standard_normal((3, D))).astype(f32)
    codes = pq.encode_all(v[300:1300])
    cv = J.PQVectors(ctx, pq, codes)
    luts = J.QueryTables(ctx, pq, 4)

    def ref_table(tier, name, query):
        out = np.zeros(M * 256, f32)
        off = 0
        for m in range(M):
            blk = cb[off: off + 256 * int(sizes[m])]
            if query is None:
                R.fn(tier, name)(fp(blk), m, int(sizes[m]), 256, fp(out))
            else:
                R.fn(tier, name)(fp(blk), m, int(sizes[m]), 256, fp(query), int(offs[m]), fp(out))
            off += 256 * int(sizes[m])
        return out

    cb64 = [opq.codebook(m).astype(np.float64).reshape(256, int(sizes[m])) for m in range(M)]
    for vsf, name in ((VSF.DOT_PRODUCT, "calculate_partial_sums_dot_f32"), (VSF.EUCLIDEAN, "calculate_partial_sums_euclidean_f32")):
        luts.build(q, vsf, J.DecoderKind.PQ)
        for i in range(3):
            lut, _ = luts.table(i)
            q64 = [q[i, offs[m]: offs[m] + sizes[m]].astype(np.float64) for m in range(M)]
            scale = np.concatenate([(np.abs(cb64[m] * q64[m]).sum(1) if vsf == VSF.DOT_PRODUCT else ((cb64[m] - q64[m]) ** 2).sum(1))
                                    for m in range(M)])
            for tier in TIERS:
                want = ref_table(tier, name, q[i])
                assert (np.abs(lut.astype(np.float64) - want) <= REL * scale + 1e-30).all(), (vsf, tier, i)
    amag_gpu = pq.self_magnitudes()
    for tier in TIERS:
        want = ref_table(tier, "calculate_partial_sums_self_magnitude_f32", None)
        assert np.allclose(amag_gpu, want, rtol=REL, atol=0), tier
    for vsf in VSF:
        got = cv.precomputed_score_function_for(q, vsf).similarity_to_range(0, len(codes))
        for tier in TIERS:
            for i in range(3):
                if vsf == VSF.COSINE:
                    t = ref_table(tier, "calculate_partial_sums_dot_f32", q[i])
                    am = ref_table(tier, "calculate_partial_sums_self_magnitude_f32", None)
                    bmag = R.fn(tier, "dot_product_f32")(fp(q[i]), 0, fp(q[i]), 0, D)
                    raw = [R.fn(tier, "pq_decoded_cosine_similarity_f32")(u8(c), 0, M, 256, fp(t), fp(am), F(bmag)) for c in codes[:200]]
                else:
                    t = ref_table(tier, "calculate_partial_sums_dot_f32" if vsf == VSF.DOT_PRODUCT else "calculate_partial_sums_euclidean_f32", q[i])
                    raw = [R.fn(tier, "assemble_and_sum_f32")(fp(t), 256, u8(c), 0, M) for c in codes[:200]]
                want = np.array([O.score_from_raw(int(vsf), r) for r in raw], f32)
                assert np.allclose(got[i][:200], want, rtol=2 * REL, atol=0), (vsf, tier, i, np.abs(got[i][:200] - want).max())


@pytest.mark.parametrize("D,M", [(128, 16), (768, 96), (50, 7)])
def test_encode_code_bytes_vs_reference_native(ctx, D, M):
    """§8a row 3: pq_encode kernel's bytes vs the argmin of the reference's euclidean_f32 over the 256 centroids of every
    subspace (ProductQuantization.java:586-600), equal wherever the nearest two centroids differ by more than 1e-5 relative"""
    rng = np.random.default_rng(7 * D + M)
    sizes, offs = O.subvector_sizes_offsets(D, M)
    cb = np.concatenate([(rng.standard_normal(256 * s) * 0.3).astype(f32) for s in sizes])
    centroid = (rng.standard_normal(D) * 0.05).astype(f32)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb, centroid)
    vecs = (rng.standard_normal((8 if M > 16 else 32, D)) * 0.3).astype(f32)
    got = pq.encode_all(vecs)
    l2 = R.fn("avx3", "euclidean_f32")
    checked = ambiguous = 0
    for v, g in zip(vecs, got):
        x = (v - centroid).astype(f32)
        off = 0
        for m in range(M):
            s = int(sizes[m])
            blk = cb[off: off + 256 * s]
            d = np.array([l2(fp(blk), j * s, fp(x), int(offs[m]), s) for j in range(256)], f32)
            two = np.partition(d, 1)[:2]
            if two[1] - two[0] > REL * two[1]:
                assert int(np.argmin(d)) == int(g[m]), (m, d.min(), d[int(g[m])])
                checked += 1
            else:
                assert d[int(g[m])] - two[0] <= REL * two[1]
                ambiguous += 1
            off += 256 * s
    assert checked > 0.99 * (checked + ambiguous)


@pytest.mark.parametrize("D,M", [(128, 16), (768, 96)])
def test_pair_scores_vs_reference_native(ctx, D, M):
    """§8 f2: the build's code-vs-code diversity scores vs assemble_and_sum_pq_f32 over the triangular table (EUCLIDEAN and
    DOT_PRODUCT are the functions the reference's native path serves, ImmutablePQVectors.java:88-118)"""
    rng = np.random.default_rng(D)
    cb = (rng.standard_normal(256 * D) * 0.3).astype(f32)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    n = 2000
    codes = rng.integers(0, 256, (n, M), dtype=np.uint8)
    cv = J.PQVectors(ctx, pq, codes)
    n1 = rng.integers(0, n, 16).astype(np.int32)
    n2 = rng.integers(0, n, (16, 8)).astype(np.int32)
    for vsf in (VSF.EUCLIDEAN, VSF.DOT_PRODUCT):
        bsp = J.PQBuildScoreProvider(ctx, cv, vsf)
        tri = bsp.codebook_partial_sums()
        got = bsp.diversity_scores(n1, n2)
        # random codes of a random codebook: the dot product of two unrelated decoded vectors cancels to ~0, so "1e-5 relative"
        # is taken of what was summed (the magnitudes of the M table entries), as in tests/test_ref_native_cpu.py
        def terms(a, b):
            r, c = np.minimum(codes[a], codes[b]).astype(np.int64), np.maximum(codes[a], codes[b]).astype(np.int64)
            return np.abs(tri[np.arange(M) * (256 * 257 // 2) + r * 256 - r * (r - 1) // 2 + (c - r)].astype(np.float64)).sum()
        scale = np.array([[terms(a, b) for b in row] for a, row in zip(n1, n2)])
        for tier in TIERS:
            f = R.fn(tier, "assemble_and_sum_pq_f32")
            raw = np.array([[f(fp(tri), M, u8(codes[a]), 0, u8(codes[b]), 0, 256) for b in row] for a, row in zip(n1, n2)], f32)
            want = np.array([[O.score_from_raw(int(vsf), x) for x in row] for row in raw], f32)
            if vsf == VSF.EUCLIDEAN:     # 1 / (1 + d): no cancellation, plain relative
                assert np.allclose(got, want, rtol=REL, atol=0), (vsf, tier, np.abs(got - want).max())
            else:                        # (1 + dot) / 2
                assert (np.abs(got.astype(np.float64) - want) <= REL * scale / 2 + 1e-30).all(), (vsf, tier, np.abs(got - want).max())
        bsp.close()


@pytest.mark.parametrize("D,S", [(768, 2), (256, 8), (100, 3)])
def test_nvq_scores_vs_reference_native(ctx, D, S):
    """the NVQ reranker: nvq score kernels vs NVQScorer (NVQScorer.java:46-140) assembled from the reference's
    nvq_dot_product_8bit / nvq_square_l2_distance_8bit / nvq_cosine_8bit_packed on shuffled query sub-vectors"""
    rng = np.random.default_rng(D + S)
    n, Q, B = 300, 3, 24
    X = unit_rows(rng, n, D)
    o = O.OracleNVQ.compute(X, S)
    o.encode_all(X)
    nvq = J.NVQuantization.create(ctx, o.mean, S)
    nv = J.NVQVectors(ctx, nvq, o.bytes, o.params)
    q = (X[rng.integers(0, n, Q)] + 0.05 * rng.standard_normal((Q, D))).astype(f32)
    ords = rng.integers(0, n, (Q, B)).astype(np.int32)
    sizes = o.sizes()
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(int)
    for vsf in VSF:
        got = nv.scores(q, vsf, ords)
        for tier in TIERS:
            shuf = R.fn(tier, "nvq_shuffle_query_in_place_8bit")

            def subs(x):
                out = []
                for s in range(S):
                    p = np.ascontiguousarray(x[offs[s]: offs[s] + sizes[s]]).copy()
                    shuf(fp(p), len(p))
                    out.append(p)
                return out

            want = np.zeros((Q, B), f32)
            for i in range(Q):
                if vsf == VSF.DOT_PRODUCT:
                    bias = R.fn(tier, "dot_product_f32")(fp(q[i]), 0, fp(o.mean), 0, D)
                    qs = subs(q[i])
                elif vsf == VSF.EUCLIDEAN:
                    qs = subs((q[i] - o.mean).astype(f32))
                else:
                    qn = f32(np.sqrt(R.fn(tier, "dot_product_f32")(fp(q[i]), 0, fp(q[i]), 0, D)))
                    qs, ms = subs(q[i]), subs(o.mean)
                for b, node in enumerate(ords[i]):
                    acc, norm = f32(0), f32(0)
                    for s in range(S):
                        by = np.ascontiguousarray(o.bytes[node, offs[s]: offs[s] + sizes[s]])
                        lo, hi, growth, mid = (F(float(x)) for x in o.params[node, s])
                        args = (growth, mid, lo, hi)
                        if vsf == VSF.DOT_PRODUCT:
                            acc = f32(acc + f32(R.fn(tier, "nvq_dot_product_8bit")(fp(qs[s]), u8(by), len(by), *args)))
                        elif vsf == VSF.EUCLIDEAN:
                            acc = f32(acc + f32(R.fn(tier, "nvq_square_l2_distance_8bit")(fp(qs[s]), u8(by), len(by), *args)))
                        else:
                            c, m2 = ref.unpack_cosine(R.fn(tier, "nvq_cosine_8bit_packed")(fp(qs[s]), u8(by), len(by), *args, fp(ms[s])))
                            acc, norm = f32(acc + f32(c)), f32(norm + f32(m2))
                    if vsf == VSF.DOT_PRODUCT:
                        want[i, b] = (f32(1) + acc + f32(bias)) / f32(2)
                    elif vsf == VSF.EUCLIDEAN:
                        want[i, b] = f32(1) / (f32(1) + acc)
                    else:
                        want[i, b] = (f32(1) + (acc / qn) / f32(np.sqrt(norm))) / f32(2)
            assert np.allclose(got, want, rtol=REL, atol=0), (vsf, tier, np.abs(got - want).max())
