"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): bit-exact PQ codes and top-k indices; float distances within 1e-5 relative —
the kernels reproduce the scalar reference's association order, so the tests demand BIT-EXACT floats
(np.array_equal) wherever the oracle defines the order, which is stricter than the stated tolerance.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import jvector_amd as J
from jvector_amd import VectorSimilarityFunction as VSF
from oracle import oracle as O

ALL_VSF = [VSF.EUCLIDEAN, VSF.DOT_PRODUCT, VSF.COSINE]
REL_TOL = 1e-5  # north_star tolerance for float distances (only used where noted)


@pytest.fixture(scope="module")
def ctx():
    c = J.HipContext(0)
    assert c.arch.startswith("gfx950")
    yield c
    c.close()


def make_pq(ctx, rng, D, M, center=False, scale=1.0):
    cb = (rng.standard_normal(256 * D) * scale).astype(np.float32)
    centroid = (rng.standard_normal(D) * 0.1).astype(np.float32) if center else None
    return J.ProductQuantization.from_codebooks(ctx, D, M, cb, centroid), O.OraclePQ(D, M, cb, centroid)


# ------------------------------------------------------------------------------------------------
# row 3: encode
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D,M,center", [(128, 16, False), (768, 96, True), (64, 8, True), (10, 3, False),
                                        (100, 7, True), (32, 32, False), (48, 4, False), (6, 6, True),
                                        (1021, 1, False), (300, 2, True), (250, 2, False)])  # codebook of a subspace beyond LDS / at its edge
def test_encode_bit_exact(ctx, D, M, center):
    rng = np.random.default_rng(D * 1000 + M)
    pq, opq = make_pq(ctx, rng, D, M, center)
    vecs = rng.standard_normal((3000, D)).astype(np.float32)
    got = pq.encode_all(vecs)
    want = opq.encode_all(vecs)
    assert got.dtype == np.uint8 and np.array_equal(got, want)


@pytest.mark.parametrize("D,M,k", [(64, 8, 16), (100, 7, 50), (128, 16, 255), (24, 3, 1)])
def test_cluster_counts_below_256(ctx, D, M, k):
    """ProductQuantization allows 1..256 clusters (the reference's own tests train 16 and 50).  The engine stores such a quantizer
    padded to 256 rows per sub-space with copies of centroid 0 — never chosen, closestCentroidIndex keeps the FIRST minimum — so
    codes, table scores, magnitudes and the wire format must equal the oracle's, which works with the true cluster count"""
    rng = np.random.default_rng(D + k)
    cb = rng.standard_normal(k * D).astype(np.float32)
    centroid = (rng.standard_normal(D) * 0.1).astype(np.float32)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb, centroid, cluster_count=k)
    opq = O.OraclePQ(D, M, cb, centroid, k=k)
    assert pq.get_cluster_count() == k
    vecs = rng.standard_normal((2000, D)).astype(np.float32)
    hit = rng.integers(0, k, (50, M)).astype(np.uint8)
    hit[:, 0] = 0                                        # sub-vector 0 = centroid 0 exactly: ties with every padded copy of it
    vecs[:50] = np.stack([opq.decode(c) for c in hit])
    got = pq.encode_all(vecs)
    assert np.array_equal(got, opq.encode_all(vecs)) and int(got.max()) < k
    codes = rng.integers(0, k, (3000, M)).astype(np.uint8)
    queries = rng.standard_normal((3, D)).astype(np.float32)
    cv = J.PQVectors(ctx, pq, codes)
    for vsf in ALL_VSF:
        sf = cv.precomputed_score_function_for(queries, vsf)
        got_sc = sf.similarity_to_range(0, len(codes))
        for q in range(3):
            assert np.array_equal(got_sc[q], opq.adc_scores(queries[q], int(vsf), codes)), (vsf, q)
    # the wire format carries the caller's cluster count and codebooks, not the padding
    blob = pq.write()
    pq2 = J.ProductQuantization.load(ctx, blob)
    assert pq2.get_cluster_count() == k and np.array_equal(pq2.encode_all(vecs[:300]), got[:300])
    assert np.array_equal(pq.self_magnitudes(), opq.cache_self_magnitudes())
    with pytest.raises(ValueError):           # FusedPQ requires 256 clusters: IllegalArgumentException (FusedPQ.java:57-59)
        J.FusedPQ(ctx, pq, np.zeros((4, 2 * M), np.uint8), np.zeros((4, 2), np.int32))


def test_encode_ties_and_nan(ctx):
    # strict '<' keeps the FIRST minimum; NaN distances never win (ProductQuantization.java:507-520)
    cb = np.full((256, 2), 5.0, np.float32)
    cb[7] = cb[9] = [1.0, 1.0]
    cb[3] = [np.nan, 0.0]
    pq = J.ProductQuantization.from_codebooks(ctx, 2, 1, cb.reshape(-1))
    opq = O.OraclePQ(2, 1, cb.reshape(-1))
    vecs = np.array([[1.0, 1.0], [np.nan, 1.0], [5.0, 5.0], [0.0, 0.0]], np.float32)
    got = pq.encode_all(vecs)
    assert np.array_equal(got, opq.encode_all(vecs))
    assert got[0, 0] == 7 and got[1, 0] == 0


def test_perfect_reconstruction(ctx):  # TestProductQuantization.java:54-80
    rng = np.random.default_rng(0)
    pts = rng.integers(0, 100000, (256, 3)).astype(np.float32)
    cb = np.concatenate([pts[:, 0:2].reshape(-1), pts[:, 2:3].reshape(-1)])
    pq = J.ProductQuantization.from_codebooks(ctx, 3, 2, cb)
    opq = O.OraclePQ(3, 2, cb)
    vecs = np.repeat(pts, 10, axis=0)
    codes = pq.encode_all(vecs)
    for i in range(0, vecs.shape[0], 37):
        assert np.array_equal(opq.decode(codes[i]), vecs[i])


def test_encode_device_resident_path(ctx):
    import torch
    rng = np.random.default_rng(5)
    pq, opq = make_pq(ctx, rng, 128, 16, True)
    vecs = rng.standard_normal((5000, 128)).astype(np.float32)
    tv = torch.from_numpy(vecs).cuda()
    vs = J.VectorSet(ctx, tv)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    ctx.sync()
    assert np.array_equal(cv.get(0, 5000), opq.encode_all(vecs))
    tcodes = pq.encode_all(tv)
    ctx.sync()
    assert tcodes.is_cuda and np.array_equal(tcodes.cpu().numpy(), opq.encode_all(vecs))


# ------------------------------------------------------------------------------------------------
# row 2: LUTs / magnitudes
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D,M,center", [(128, 16, False), (768, 96, True), (10, 3, True), (100, 7, False)])
def test_luts_bit_exact(ctx, D, M, center):
    rng = np.random.default_rng(D + M)
    pq, opq = make_pq(ctx, rng, D, M, center)
    queries = rng.standard_normal((5, D)).astype(np.float32)
    assert np.array_equal(pq.self_magnitudes(), opq.decoder(queries[0], O.COSINE)[1])
    luts = J.QueryTables(ctx, pq, 8)
    for vsf in ALL_VSF:
        for kind, fused in ((J.DecoderKind.PQ, False), (J.DecoderKind.FUSED, True)):
            luts.build(queries, vsf, kind)
            for q in range(5):
                lut, bmag = luts.table(q)
                wl, _, wb = opq.decoder(queries[q], int(vsf), fused=fused)
                assert np.array_equal(lut, wl), (vsf, kind, q)
                if vsf == VSF.COSINE:
                    assert np.float32(bmag) == np.float32(wb), (kind, q)


# ------------------------------------------------------------------------------------------------
# rows 5/6: ADC scan + gather
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D,M,N", [(128, 16, 20000),      # C2 shape
                                   (768, 96, 9000),        # C3 shape (LDS LUT 96 KB)
                                   (1536, 192, 5000),      # C5 shape: two LDS passes
                                   (40, 10, 7000),         # M % 4 != 0 -> byte path
                                   (36, 12, 7000),         # M % 4 == 0, not 16 -> dword path
                                   (9, 3, 3000), (64, 64, 3000), (17, 1, 2500)])
def test_adc_scan_bit_exact(ctx, D, M, N):
    rng = np.random.default_rng(N + M)
    pq, opq = make_pq(ctx, rng, D, M, center=(M % 2 == 0))
    codes = rng.integers(0, 256, (N, M)).astype(np.uint8)
    queries = rng.standard_normal((3, D)).astype(np.float32)
    cv = J.PQVectors(ctx, pq, codes)
    for vsf in ALL_VSF:
        sf = cv.precomputed_score_function_for(queries, vsf)
        got = sf.similarity_to_range(0, N)
        sub = sf.similarity_to_range(1000, 1234)
        for q in range(3):
            want = opq.adc_scores(queries[q], int(vsf), codes)
            assert np.array_equal(got[q], want), (vsf, q)
            assert np.array_equal(sub[q], want[1000:2234])


@pytest.mark.parametrize("B", [1, 32, 100, 5000])  # < 2048: LUT gathered from L2; >= 2048: LUT in LDS
def test_adc_gather_bit_exact(ctx, B):
    rng = np.random.default_rng(B)
    D, M, N = 768, 96, 30000
    pq, opq = make_pq(ctx, rng, D, M, center=True)
    codes = rng.integers(0, 256, (N, M)).astype(np.uint8)
    queries = rng.standard_normal((4, D)).astype(np.float32)
    cv = J.PQVectors(ctx, pq, codes)
    ords = rng.integers(0, N, (4, B)).astype(np.int32)
    if B > 4:
        ords[1, 3] = -1       # skipped slot
        ords[2, 0] = N        # out of range -> treated as invalid
    for vsf in ALL_VSF:
        got = cv.precomputed_score_function_for(queries, vsf).similarity_to(ords)
        for q in range(4):
            valid = (ords[q] >= 0) & (ords[q] < N)
            want = opq.adc_scores(queries[q], int(vsf), codes, np.where(valid, ords[q], 0).astype(np.int32))
            assert np.array_equal(got[q][valid], want[valid]), (vsf, q)
            assert np.all(np.isneginf(got[q][~valid]))


def test_adc_precomputed_equals_direct(ctx):  # TestCompressedVectors.java:230-256, tol 1e-6
    rng = np.random.default_rng(99)
    for D, M in ((64, 8), (130, 13), (2048, 64)):
        pq, opq = make_pq(ctx, rng, D, M, center=True, scale=0.2)
        vecs = rng.uniform(-1, 1, (50, D)).astype(np.float32)
        vecs /= np.linalg.norm(vecs, axis=1, keepdims=True)
        codes = pq.encode_all(vecs)
        cv = J.PQVectors(ctx, pq, codes)
        for vsf in ALL_VSF:
            got = cv.precomputed_score_function_for(vecs[:2], vsf).similarity_to_range(0, 50)
            for q in range(2):
                for i in range(50):
                    assert abs(got[q, i] - opq.direct_score(vecs[q], int(vsf), codes[i])) <= 1e-6


# ------------------------------------------------------------------------------------------------
# row 7: fused blocks
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D,M,deg", [(768, 96, 32), (128, 16, 32), (40, 10, 12)])
def test_fused_equals_unfused(ctx, D, M, deg):  # TestFusedGraphIndex.java:74-114,193-233
    rng = np.random.default_rng(deg + M)
    N = 500
    pq, opq = make_pq(ctx, rng, D, M, center=True)
    codes = rng.integers(0, 256, (N, M)).astype(np.uint8)
    neighbors = np.full((N, deg), -1, np.int32)
    blocks = np.zeros((N, deg * M), np.uint8)  # zero padded (FusedPQ.java:157-160)
    for n in range(N):
        d = int(rng.integers(0, deg + 1))
        nb = rng.choice(N, d, replace=False).astype(np.int32)
        neighbors[n, :d] = nb
        blocks[n, : d * M] = codes[nb].reshape(-1)
    fused = J.FusedPQ(ctx, pq, blocks, neighbors)
    Q = 6
    queries = rng.standard_normal((Q, D)).astype(np.float32)
    origins = rng.integers(0, N, Q).astype(np.int32)
    for vsf in ALL_VSF:
        got, nb = fused.approximate_score_function_for(queries, vsf).similarity_to_neighbors(origins, True)
        for q in range(Q):
            assert np.array_equal(nb[q], neighbors[origins[q]])
            valid = neighbors[origins[q]] >= 0
            want = opq.adc_scores(queries[q], int(vsf), codes, neighbors[origins[q]][valid], fused=True)
            assert np.array_equal(got[q][valid], want), (vsf, q)
            assert np.all(np.isneginf(got[q][~valid]))


# ------------------------------------------------------------------------------------------------
# row 1: exact distances
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D", [128, 768, 1536, 1021, 7, 8, 9, 100])
def test_exact_gather_and_scan_bit_exact(ctx, D):
    rng = np.random.default_rng(D)
    N, Q = 3000, 19
    vecs = rng.standard_normal((N, D)).astype(np.float32)
    queries = rng.standard_normal((Q, D)).astype(np.float32)
    vs = J.VectorSet(ctx, vecs)
    ords = rng.integers(0, N, (Q, 70)).astype(np.int32)
    ords[0, 5] = -1
    for vsf in ALL_VSF:
        g = vs.scores(queries, vsf, ords)
        s = vs.scan(queries, vsf)
        for q in range(Q):
            want_all = O.compare_many(int(vsf), queries[q], vecs)
            assert np.array_equal(s[q], want_all), (vsf, q)
            valid = ords[q] >= 0
            assert np.array_equal(g[q][valid], want_all[ords[q][valid]])
            assert np.all(np.isneginf(g[q][~valid]))


@pytest.mark.parametrize("D,B", [(768, 76), (768, 12), (768, 68), (768, 96), (768, 100), (768, 140), (40, 76), (1536, 74), (128, 5)])
def test_exact_gather_list_lengths_and_packed_remainders(ctx, D, B):
    """the rerank gather over lists of every shape: B = 64 f + rem — full wavefronts by exact_gather_tr_kernel, remainders of 4 .. 32 rows
    PACKED several queries per wavefront (exact_gather_trq_kernel, round 6), longer / shorter remainders in a wavefront of their own;
    invalid ordinals in both parts, a query count that does not fill the last packed wavefront: every score equals the oracle's"""
    rng = np.random.default_rng(D + B)
    N, Q = 2000, 23
    vecs = rng.standard_normal((N, D)).astype(np.float32)
    queries = rng.standard_normal((Q, D)).astype(np.float32)
    vs = J.VectorSet(ctx, vecs)
    ords = rng.integers(0, N, (Q, B)).astype(np.int32)
    ords[0, B - 1] = -1
    ords[Q - 1, 0] = -1
    ords[7, B // 2] = N + 5          # outside the set: -inf like a -1
    for vsf in ALL_VSF:
        g = vs.scores(queries, vsf, ords)
        for q in range(Q):
            want_all = O.compare_many(int(vsf), queries[q], vecs)
            valid = (ords[q] >= 0) & (ords[q] < N)
            assert np.array_equal(g[q][valid], want_all[ords[q][valid]]), (vsf, q)
            assert np.all(np.isneginf(g[q][~valid]))
    # one query: no packing (a single list's remainder has nobody to share a wavefront with)
    g1 = vs.scores(queries[:1], VSF.COSINE, ords[:1])
    want = O.compare_many(int(VSF.COSINE), queries[0], vecs)
    v0 = (ords[0] >= 0) & (ords[0] < N)
    assert np.array_equal(g1[0][v0], want[ords[0][v0]])


def test_wrapped_vectors_edited_in_place_need_invalidate(ctx):
    """a VectorSet that WRAPS caller-owned device memory caches one float per row for the cosine rerank; after the
    caller rewrites rows in place, jv_hip_vectors_invalidate makes the next cosine score see the new rows (== the oracle)"""
    import torch
    rng = np.random.default_rng(5)
    N, D, Q = 500, 128, 7
    vecs = rng.standard_normal((N, D)).astype(np.float32)
    queries = rng.standard_normal((Q, D)).astype(np.float32)
    ords = rng.integers(0, N, (Q, 70)).astype(np.int32)
    t = torch.from_numpy(vecs.copy())
    if torch.cuda.is_available():
        t = t.cuda()
    vs = J.VectorSet(ctx, t)
    g0 = np.asarray(torch.as_tensor(vs.scores(queries, VSF.COSINE, ords)).cpu())
    for q in range(Q):
        assert np.array_equal(g0[q], O.compare_many(int(VSF.COSINE), queries[q], vecs)[ords[q]])
    t.mul_(3.0)                                   # cosine is scale invariant in exact arithmetic, not in f32 bits ...
    t[11] = torch.from_numpy(queries[0]).to(t.device)   # ... and one row changes outright
    vecs2 = np.asarray(t.cpu())
    vs.invalidate()
    ctx.sync()
    g1 = np.asarray(torch.as_tensor(vs.scores(queries, VSF.COSINE, ords)).cpu())
    for q in range(Q):
        assert np.array_equal(g1[q], O.compare_many(int(VSF.COSINE), queries[q], vecs2)[ords[q]])


@pytest.mark.parametrize("D,B", [(768, 150), (1536, 150), (8, 1), (72, 64), (200, 65), (64, 129), (1024, 400)])
def test_exact_gather_transposing_kernel(ctx, D, B, monkeypatch):
    """NodeQueue.rerank's scoring at the benched shapes through the coalesced (LDS-transposing) kernel: ragged candidate
    counts (not a multiple of the 64 rows a wave takes), chunk tails (D not a multiple of the 64-float chunk), -1 and
    out-of-range ordinals, duplicate ordinals, zero rows (cosine NaN) — bit-identical to the oracle AND to the lane-per-row
    kernel it replaces; then an upload invalidates the cosine norm table."""
    rng = np.random.default_rng(D * 1000 + B)
    N, Q = 2500, 23
    vecs = rng.standard_normal((N, D)).astype(np.float32)
    vecs[7] = 0.0
    queries = rng.standard_normal((Q, D)).astype(np.float32)
    vs = J.VectorSet(ctx, vecs)
    ords = rng.integers(0, N, (Q, B)).astype(np.int32)
    ords[0, 0] = -1
    ords[1, B - 1] = N + 5                     # out of range -> -inf like -1
    ords[2, :] = ords[2, 0]                    # one row 150 times
    ords[3, B // 2] = 7                        # the zero vector
    for vsf in ALL_VSF:
        g = vs.scores(queries, vsf, ords)
        monkeypatch.setenv("JVECTOR_HIP_EXACT_LANE_ROWS", "1")
        g_old = vs.scores(queries, vsf, ords)
        monkeypatch.delenv("JVECTOR_HIP_EXACT_LANE_ROWS")
        assert np.array_equal(g, g_old, equal_nan=True), vsf
        for q in range(Q):
            valid = (ords[q] >= 0) & (ords[q] < N)
            want = O.compare_many(int(vsf), queries[q], vecs[ords[q][valid]])
            assert np.array_equal(g[q][valid], want, equal_nan=True), (vsf, q)
            assert np.all(np.isneginf(g[q][~valid]))
    # the cosine norm table follows uploads
    vs2 = J.VectorSet(ctx, np.ascontiguousarray(vecs[:64]))           # host upload -> owned storage
    a = vs2.scores(queries[:2], VSF.COSINE, np.arange(64, dtype=np.int32).reshape(1, 64).repeat(2, 0))
    newrows = (3.0 * vecs[100:164]).astype(np.float32)
    J._lib.check(ctx._lib.jv_hip_vectors_upload(ctx._h, vs2._h, 0, 64, J.engine._ptr(newrows, np.float32)[0]))
    b = vs2.scores(queries[:2], VSF.COSINE, np.arange(64, dtype=np.int32).reshape(1, 64).repeat(2, 0))
    for q in range(2):
        assert np.array_equal(a[q], O.compare_many(int(VSF.COSINE), queries[q], vecs[:64]), equal_nan=True)
        assert np.array_equal(b[q], O.compare_many(int(VSF.COSINE), queries[q], newrows))


def test_exact_known_answers(ctx, golden_dir):
    """the reference's native KATs (test_similarity.cpp:92-219) through the HIP exact kernels"""
    import json
    kat = json.load(open(os.path.join(golden_dir, "oracle_kat.json")))
    for case in kat["cases"]:
        n = case["n"]
        a, b = O.make_vec(n, 0.7), O.make_vec(n, 1.3)
        vs = J.VectorSet(ctx, b.reshape(1, n))
        ords = np.zeros((1, 1), np.int32)
        l2 = 1.0 / vs.scores(a.reshape(1, n), VSF.EUCLIDEAN, ords)[0, 0] - 1.0
        dot = vs.scores(a.reshape(1, n), VSF.DOT_PRODUCT, ords)[0, 0] * 2.0 - 1.0
        cos = vs.scores(a.reshape(1, n), VSF.COSINE, ords)[0, 0] * 2.0 - 1.0
        assert abs(l2 - case["l2"]) <= 1e-4 * abs(case["l2"]) + 1e-6
        assert abs(dot - case["dot"]) <= 1e-4 * abs(case["dot"]) + 1e-6
        assert abs(cos - case["cosine"]) <= 1e-4 * abs(case["cosine"]) + 1e-6
        vs.close()


# ------------------------------------------------------------------------------------------------
# row 9: top-k order
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,k", [(10, 3), (1000, 10), (100000, 100), (100000, 1000), (50, 64), (5000, 4096),
                                 (110, 10), (128, 1), (129, 64), (256, 17), (257, 10), (512, 64), (1024, 64), (1025, 64), (1024, 65),
                                 (2048, 10), (2049, 64), (3200, 10), (4096, 64), (4097, 64)])
def test_topk_matches_nodequeue_order(ctx, n, k):
    """rows of <= 4096 with k <= 64 take the one-wavefront register kernel (2 ... 64 keys per lane), everything else the
    radix select; both must give the NodeQueue order"""
    rng = np.random.default_rng(n + k)
    Q = 5
    scores = rng.standard_normal((Q, n)).astype(np.float32)
    scores[1] = np.round(scores[1] * 4) / 4        # many ties
    scores[2] = 0.5                                # all equal: ids decide
    scores[3, ::7] = np.inf
    scores[4, ::5] = -np.inf
    ids, sc = J.topk(ctx, scores, k)
    for q in range(Q):
        wi, ws = O.topk(None, scores[q], k)
        cnt = len(wi)
        assert np.array_equal(ids[q][:cnt], wi), q
        assert np.array_equal(sc[q][:cnt], ws), q
        assert np.all(ids[q][cnt:] == -1) and np.all(np.isneginf(sc[q][cnt:]))


def test_exact_pair_scores_bit_exact(ctx):
    """the exact build-score provider (BuildScoreProvider.randomAccessScoreProvider :106-160): node-vs-node full-resolution
    similarity for P x B blocks == the scalar compare of the two rows, -inf for ordinals outside the set on either side"""
    rng = np.random.default_rng(21)
    for D in (24, 768):
        N, P, B = 500, 37, 19
        v = rng.standard_normal((N, D)).astype(np.float32)
        vs = J.VectorSet(ctx, v)
        n1 = rng.integers(0, N, P).astype(np.int32)
        n2 = rng.integers(0, N, (P, B)).astype(np.int32)
        n1[3], n1[7] = -1, N + 5
        n2[0, 2], n2[5, 0] = -1, N
        for vsf in VSF:
            got = vs.pair_scores(vsf, n1, n2)
            for p in range(P):
                for b in range(B):
                    bad = not (0 <= n1[p] < N and 0 <= n2[p, b] < N)
                    want = -np.inf if bad else O.compare(int(vsf), v[n1[p]], v[n2[p, b]])
                    assert got[p, b] == want or (np.isnan(got[p, b]) and np.isnan(want)), (D, vsf, p, b)


def test_topk_short_rows_with_ids_and_padding(ctx):
    """the rerank's shape: Q x rerankK candidate lists with -1 padded tails, engineered score ties, many rows"""
    rng = np.random.default_rng(9)
    Q, R, k = 300, 110, 10
    ids = np.stack([rng.permutation(50_000)[:R] for _ in range(Q)]).astype(np.int32)
    scores = (np.round(rng.standard_normal((Q, R)) * 16) / 16).astype(np.float32)
    for q in range(Q):
        ids[q, int(rng.integers(0, R + 1)):] = -1          # anything from an empty list to a full one
    got_i, got_s = J.topk(ctx, scores, k, ids=ids)
    for q in range(Q):
        valid = ids[q] >= 0
        wi, ws = O.topk(ids[q][valid], scores[q][valid], k)
        cnt = len(wi)
        assert np.array_equal(got_i[q][:cnt], wi) and np.array_equal(got_s[q][:cnt], ws), q
        assert np.all(got_i[q][cnt:] == -1) and np.all(np.isneginf(got_s[q][cnt:])), q


def test_topk_explicit_ids_merge(ctx):
    """the sharded merge step: lists with global ids, -1 padding ignored"""
    rng = np.random.default_rng(3)
    Q, P, k = 4, 8, 100
    ids = rng.permutation(1_000_000)[: Q * P * k].reshape(Q, P * k).astype(np.int32)
    scores = np.round(rng.standard_normal((Q, P * k)) * 8).astype(np.float32) / 8
    ids[0, 5:50] = -1
    got_i, got_s = J.topk(ctx, scores, k, ids=ids)
    for q in range(Q):
        valid = ids[q] >= 0
        wi, ws = O.topk(ids[q][valid], scores[q][valid], k)
        assert np.array_equal(got_i[q], wi) and np.array_equal(got_s[q], ws)


# ------------------------------------------------------------------------------------------------
# end to end: two-pass flat search == oracle pipeline
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D,M,N", [(128, 16, 20000), (768, 96, 6000)])
def test_search_flat_matches_oracle(ctx, D, M, N):
    rng = np.random.default_rng(D)
    centers = rng.standard_normal((50, D)).astype(np.float32)
    vecs = (centers[rng.integers(0, 50, N)] + 0.3 * rng.standard_normal((N, D))).astype(np.float32)
    vecs /= np.linalg.norm(vecs, axis=1, keepdims=True)
    queries = vecs[rng.integers(0, N, 7)] + 0.05 * rng.standard_normal((7, D)).astype(np.float32)
    queries = queries.astype(np.float32)
    # codebooks: 256 sampled sub-vectors per subspace (fixed, never retrained in a parity test)
    sizes, offs = O.subvector_sizes_offsets(D, M)
    pick = rng.choice(N, 256, replace=False)
    cb = np.concatenate([vecs[pick, offs[m]: offs[m] + sizes[m]].reshape(-1) for m in range(M)])
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    opq = O.OraclePQ(D, M, cb)
    vs = J.VectorSet(ctx, vecs)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    ctx.sync()
    codes = cv.get(0, N)
    assert np.array_equal(codes, opq.encode_all(vecs))
    searcher = J.FlatSearcher(ctx, pq, cv, vs, max_queries=16, id_base=1000)
    for vsf in ALL_VSF:
        ids, sc = searcher.search(queries, vsf, 10, 100)
        ids_nr, sc_nr = J.FlatSearcher(ctx, pq, cv, None, max_queries=16).search(queries, vsf, 10, 0)
        for q in range(7):
            approx = opq.adc_scores(queries[q], int(vsf), codes)
            cand, cs = O.topk(None, approx, 100)
            assert np.array_equal(ids_nr[q], cand[:10]) and np.array_equal(sc_nr[q], cs[:10])
            exact = O.compare_many(int(vsf), queries[q], vecs[cand])
            wi, ws = O.topk(cand, exact, 10)
            assert np.array_equal(ids[q], wi + 1000), (vsf, q)
            assert np.array_equal(sc[q], ws)


# ------------------------------------------------------------------------------------------------
# fixtures from the reference
# ------------------------------------------------------------------------------------------------
def test_version0_pq_fixture_on_device(ctx, golden_dir):
    data = open(os.path.join(golden_dir, "version0.pq"), "rb").read()
    pq = J.ProductQuantization.load(ctx, data)
    assert pq.bytes_consumed == len(data)
    assert (pq.original_dimension, pq.M, pq.cluster_count, pq.has_global_centroid) == (2, 1, 256, False)
    opq, _, _, _ = O.OraclePQ.parse(data)
    rng = np.random.default_rng(0)
    vecs = rng.uniform(-1, 1, (1000, 2)).astype(np.float32)
    codes = pq.encode_all(vecs)
    assert np.array_equal(codes, opq.encode_all(vecs))
    cv = J.PQVectors(ctx, pq, codes)
    for vsf in ALL_VSF:
        got = cv.precomputed_score_function_for(vecs[:3], vsf).similarity_to_range(0, 1000)
        for q in range(3):
            assert np.array_equal(got[q], opq.adc_scores(vecs[q], int(vsf), codes))


def test_siftsmall_plumbing(ctx, golden_dir):
    """BASELINE config #1: the repo-shipped 100x128 SIFT queries as base and query (self-match)"""
    raw = np.fromfile(os.path.join(golden_dir, "siftsmall_query.fvecs"), dtype=np.int32).reshape(100, 129)
    base = np.ascontiguousarray(raw[:, 1:].view(np.float32))
    vs = J.VectorSet(ctx, base)
    scores = vs.scan(base, VSF.EUCLIDEAN)
    ids, sc = J.topk(ctx, scores, 5)
    for q in range(100):
        want = O.compare_many(O.EUCLIDEAN, base[q], base)
        wi, ws = O.topk(None, want, 5)
        assert np.array_equal(ids[q], wi) and np.array_equal(sc[q], ws)
        assert ids[q, 0] == q and sc[q, 0] == 1.0


# ------------------------------------------------------------------------------------------------
# boundary behaviour
# ------------------------------------------------------------------------------------------------
def test_error_behaviour(ctx):
    rng = np.random.default_rng(1)
    with pytest.raises(ValueError):  # M > D: IllegalArgumentException (ProductQuantization.java:536-538)
        J.ProductQuantization.from_codebooks(ctx, 4, 5, np.zeros(256 * 4, np.float32))
    with pytest.raises(J.UnsupportedError):  # more than one byte's worth of clusters (ProductQuantization.checkClusterCount)
        J.ProductQuantization.from_codebooks(ctx, 8, 2, np.zeros(300 * 8, np.float32), cluster_count=300)
    pq, _ = make_pq(ctx, rng, 16, 4)
    with pytest.raises(ValueError):  # dimension mismatch (VectorUtil.java:46-48)
        pq.encode_all(np.zeros((3, 15), np.float32))
    cv = J.PQVectors(ctx, pq, np.zeros((10, 4), np.uint8))
    sf = cv.precomputed_score_function_for(np.zeros((1, 16), np.float32), VSF.EUCLIDEAN)
    with pytest.raises(ValueError):  # IndexOutOfBoundsException (PQVectors.java:378-381)
        sf.similarity_to_range(5, 10)
    vs = J.VectorSet(ctx, np.zeros((10, 16), np.float32))
    with pytest.raises(ValueError):  # rerankK < topK: IllegalArgumentException (GraphSearcher.java:233)
        J.FlatSearcher(ctx, pq, cv, vs).search(np.zeros((1, 16), np.float32), VSF.EUCLIDEAN, 10, 5)
    # empty batches are no-ops
    assert pq.encode_all(np.zeros((0, 16), np.float32)).shape == (0, 4)


def test_device_pointer_io(ctx):
    import torch
    rng = np.random.default_rng(8)
    pq, opq = make_pq(ctx, rng, 128, 16)
    codes = rng.integers(0, 256, (50000, 16)).astype(np.uint8)
    q = rng.standard_normal((2, 128)).astype(np.float32)
    cv = J.PQVectors(ctx, pq, torch.from_numpy(codes).cuda())
    tq = torch.from_numpy(q).cuda()
    sf = cv.precomputed_score_function_for(tq, VSF.EUCLIDEAN)
    out = sf.similarity_to_range(0, 50000, like=tq)
    ids, sc = J.topk(ctx, out, 10)
    ctx.sync()
    assert out.is_cuda and ids.is_cuda
    for i in range(2):
        want = opq.adc_scores(q[i], O.EUCLIDEAN, codes)
        assert np.array_equal(out[i].cpu().numpy(), want)
        wi, ws = O.topk(None, want, 10)
        assert np.array_equal(ids[i].cpu().numpy(), wi) and np.array_equal(sc[i].cpu().numpy(), ws)


# ------------------------------------------------------------------------------------------------
# BASELINE config 2 at full size (SIFT1M shape: 1M x 128, PQ-16, L2): size-independent properties
# ------------------------------------------------------------------------------------------------
def test_c2_full_size_properties(ctx):
    import torch
    g = torch.Generator(device="cuda").manual_seed(2)
    N, D, M, Q = 1_000_000, 128, 16, 16
    base = torch.clamp(torch.round(torch.randn(N, D, generator=g, device="cuda").abs() * 40), 0, 218)
    queries = base[torch.randint(0, N, (Q,), generator=g, device="cuda")].clone()
    rng = np.random.default_rng(4)
    pick = rng.choice(N, 256, replace=False)
    sub = base[torch.from_numpy(pick).cuda()].cpu().numpy()
    cb = np.concatenate([sub[:, m * 8:(m + 1) * 8].reshape(-1) for m in range(M)])
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    opq = O.OraclePQ(D, M, cb)
    vs = J.VectorSet(ctx, base)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    ctx.sync()
    # (1) the 256 picked vectors are centroids in every subspace: they must encode to a centroid at distance 0
    codes_pick = cv.get(0, N)[pick]
    for j in range(0, 256, 17):
        assert np.array_equal(opq.decode(codes_pick[j]), sub[j])
    # (2) oracle spot check of codes on a strided sample
    sample = np.arange(0, N, 9973)
    assert np.array_equal(cv.get(0, N)[sample], opq.encode_all(base[torch.from_numpy(sample).cuda()].cpu().numpy()))
    # (3) scan == gather on the same ordinals, bit for bit; scores in (0, 1]
    sf = cv.precomputed_score_function_for(queries, VSF.EUCLIDEAN)
    scan = sf.similarity_to_range(0, N, like=queries)
    ords = torch.randint(0, N, (Q, 4096), generator=g, device="cuda", dtype=torch.int32)
    gath = sf.similarity_to(ords)
    ctx.sync()
    assert torch.equal(torch.gather(scan, 1, ords.long()), gath)
    assert float(scan.min()) > 0.0 and float(scan.max()) <= 1.0
    # (4) top-k: sorted best-first, ties by ascending id, and equal to the oracle on one query
    ids, sc = J.topk(ctx, scan, 100)
    ctx.sync()
    ids_h, sc_h = ids.cpu().numpy(), sc.cpu().numpy()
    for q in range(Q):
        keys = [O.nodequeue_encode(int(i), float(s)) for i, s in zip(ids_h[q], sc_h[q])]
        assert keys == sorted(keys, reverse=True)
    wi, ws = O.topk(None, scan[0].cpu().numpy(), 100)
    assert np.array_equal(ids_h[0], wi) and np.array_equal(sc_h[0], ws)
    # (5) a query that IS a base vector finds itself first after the exact rerank (score 1.0)
    s = J.FlatSearcher(ctx, pq, cv, vs, max_queries=Q)
    rid, rsc = s.search(queries, VSF.EUCLIDEAN, 10, 100)
    ctx.sync()
    assert torch.all(rsc[:, 0] == 1.0)


# ------------------------------------------------------------------------------------------------
# multi-query ADC kernel (4 queries per LDS gather) and the threshold-filtered flat search
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D,M,N,Q", [(128, 16, 30000, 2), (768, 96, 20000, 5), (1536, 192, 9000, 3),
                                     (384, 48, 12000, 7), (512, 64, 12000, 4), (256, 32, 10000, 9)])
def test_adc_multi_query_scan_bit_exact(ctx, D, M, N, Q):
    rng = np.random.default_rng(N + Q)
    pq, opq = make_pq(ctx, rng, D, M, center=True)
    codes = rng.integers(0, 256, (N, M)).astype(np.uint8)
    queries = rng.standard_normal((Q, D)).astype(np.float32)
    cv = J.PQVectors(ctx, pq, codes)
    for vsf in ALL_VSF:
        sf = cv.precomputed_score_function_for(queries, vsf)
        got = sf.similarity_to_range(0, N)
        sub = sf.similarity_to_range(777, 8200)
        for q in range(Q):
            want = opq.adc_scores(queries[q], int(vsf), codes)
            assert np.array_equal(got[q], want), (vsf, q)
            assert np.array_equal(sub[q], want[777:777 + 8200])


@pytest.mark.parametrize("vsf", ALL_VSF)
def test_filtered_search_matches_oracle(ctx, vsf):
    """N large enough for the sampled-threshold strategy; quantised scores create many exact ties at the threshold"""
    rng = np.random.default_rng(77)
    N, D, M, Q = 300_000, 64, 16, 6
    base = np.round(rng.standard_normal((N, D)) * 2).astype(np.float32)  # small integer grid -> massive score ties
    base[base == 0] = 1.0
    queries = base[rng.integers(0, N, Q)].copy()
    sizes, offs = O.subvector_sizes_offsets(D, M)
    pick = rng.choice(N, 256, replace=False)
    cb = np.concatenate([base[pick, offs[m]: offs[m] + sizes[m]].reshape(-1) for m in range(M)])
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    opq = O.OraclePQ(D, M, cb)
    vs = J.VectorSet(ctx, base)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    codes = cv.get(0, N)
    s = J.FlatSearcher(ctx, pq, cv, vs, max_queries=8)
    ids, sc = s.search(queries, vsf, 10, 200)
    ids_nr, sc_nr = J.FlatSearcher(ctx, pq, cv, None, max_queries=8).search(queries, vsf, 50, 0)
    wi, ws = opq.search_flat(codes, base, queries, int(vsf), 10, 200, nthreads=4)
    wi_nr, ws_nr = opq.search_flat(codes, None, queries, int(vsf), 50, 0, nthreads=4)
    assert np.array_equal(ids, wi) and np.array_equal(sc, ws)
    assert np.array_equal(ids_nr, wi_nr) and np.array_equal(sc_nr, ws_nr)


def test_filtered_search_overflow_falls_back(ctx):
    """all candidates identical: every score ties with the threshold, the candidate list overflows, the engine
    must fall back to the materialised strategy and return the smallest ids (NodeQueue tie rule)"""
    N, D, M, Q = 300_000, 32, 16, 3
    rng = np.random.default_rng(1)
    base = np.tile(rng.standard_normal((1, D)).astype(np.float32), (N, 1))
    cb = rng.standard_normal(256 * D).astype(np.float32)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    vs = J.VectorSet(ctx, base)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    ids, sc = J.FlatSearcher(ctx, pq, cv, vs, max_queries=4).search(base[:Q], VSF.EUCLIDEAN, 10, 100)
    assert np.array_equal(ids, np.tile(np.arange(10, dtype=np.int32), (Q, 1)))
    assert np.all(sc == 1.0)
