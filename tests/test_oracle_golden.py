"""Pins the CPU oracle against every golden vector / known-answer test the reference holds for
the hot path.  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

LENGTHS = [1, 3, 4, 5, 7, 8, 9, 15, 16, 17, 19, 32, 33, 37, 64, 71, 100, 128, 255]


@pytest.fixture(scope="module")
def kat(golden_dir):
    with open(os.path.join(golden_dir, "oracle_kat.json")) as fh:
        return json.load(fh)


# --- NC/tests/test_similarity.cpp:92-219 restated against the oracle -------------------------
def test_make_vec_matches_fixture(kat):
    assert kat["lengths"] == LENGTHS
    for case in kat["cases"]:
        v = O.make_vec(case["n"], 0.7)
        np.testing.assert_array_equal(v[:4], np.array(case["a_first"], np.float32))


@pytest.mark.parametrize("idx", range(len(LENGTHS)))
def test_similarity_known_answers(kat, idx):
    case = kat["cases"][idx]
    n = case["n"]
    a, b = O.make_vec(n, 0.7), O.make_vec(n, 1.3)
    # tolerance: 1e-4 * |want| exactly as EXPECT_NEAR in test_similarity.cpp:99,123,140,176
    for got, want in ((O.dot(a, b), case["dot"]), (O.l2(a, b), case["l2"]), (O.cosine(a, b), case["cosine"])):
        assert abs(got - want) <= 1e-4 * abs(want)
    # offset forms with a 3-element prefix (DotProductWithOffset :106-124)
    ap, bp = np.concatenate([np.full(3, 9.0, np.float32), a]), np.concatenate([np.full(3, -7.0, np.float32), b])
    assert abs(O.dot_off(ap, 3, bp, 3, n) - case["dot"]) <= 1e-4 * abs(case["dot"])
    assert abs(O.l2_off(ap, 3, bp, 3, n) - case["l2"]) <= 1e-4 * abs(case["l2"])
    assert abs(O.cosine_off(ap, 3, bp, 3, n) - case["cosine"]) <= 1e-4 * abs(case["cosine"])


@pytest.mark.parametrize("n", LENGTHS)
def test_euclidean_same_vector_is_zero(n):  # test_similarity.cpp:146-156
    a = O.make_vec(n, 0.9)
    assert O.l2(a, a) == 0.0


@pytest.mark.parametrize("n", LENGTHS)
def test_cosine_parallel_and_orthogonal(n):  # test_similarity.cpp:182-219
    a = O.make_vec(n, 1.0)
    assert abs(O.cosine(a, (2.0 * a).astype(np.float32)) - 1.0) <= 1e-5
    if n >= 2:
        even = n - n % 2
        x, y = np.zeros(n, np.float32), np.zeros(n, np.float32)
        x[:even] = 1.0
        y[:even] = np.where(np.arange(even) % 2 == 0, 1.0, -1.0)
        assert abs(O.cosine(x, y)) <= 1e-4


# --- TS/vector/TestVectorizationProvider.java:36-91 structure -----------------------------------
def test_full_form_vs_sequential_form_dim_1021():
    rng = np.random.default_rng(1021)
    for _ in range(20):
        # TestUtil.randomVector :126-136 = uniform(-1,1) then L2-normalised
        a = rng.uniform(-1, 1, 1021)
        b = rng.uniform(-1, 1, 1021)
        a = (a / np.linalg.norm(a)).astype(np.float32)
        b = (b / np.linalg.norm(b)).astype(np.float32)
        assert abs(O.dot(a, b) - O.dot_off(a, 0, b, 0, 1021)) <= 1e-4
        assert abs(O.l2(a, b) - O.l2_off(a, 0, b, 0, 1021)) <= 1e-4
        assert abs(O.cosine(a, b) - float(a.astype(np.float64) @ b / np.sqrt((a.astype(np.float64) @ a) * (b.astype(np.float64) @ b)))) <= 1e-4


def test_assemble_and_sum_vs_sum():  # TestVectorizationProvider.java:63-91
    rng = np.random.default_rng(7)
    for _ in range(200):
        table = rng.uniform(-1, 1, 256 * 32).astype(np.float32)
        offs = rng.integers(0, 256, 32).astype(np.uint8)
        want = float(sum(np.float64(table[256 * i + int(offs[i])]) for i in range(32)))
        assert abs(O.assemble_and_sum(table, 256, offs, 0, 32) - want) <= 1e-4


def test_dot_first_elements_are_remainder():
    """DefaultVectorUtilSupport.dotProduct adds the FIRST n%8 products one by one (:50-52) — make
    sure the oracle keeps that order (differs in the last ulp from a naive sequential sum)."""
    rng = np.random.default_rng(3)
    diffs = 0
    for _ in range(200):
        a = rng.standard_normal(77).astype(np.float32)
        b = rng.standard_normal(77).astype(np.float32)
        diffs += O.dot(a, b) != O.dot_off(a, 0, b, 0, 77)
    assert diffs > 0  # the two association orders are genuinely different


# --- version0.pq : TestProductQuantization.java:215-248 -------------------------------------------
def test_load_version0_pq(golden_dir):
    data = open(os.path.join(golden_dir, "version0.pq"), "rb").read()
    assert len(data) == 2064
    pq, version, aniso, consumed = O.OraclePQ.parse(data)
    assert consumed == len(data)
    assert version == 0
    assert pq.D == 2 and pq.M == 1 and pq.k == 256
    assert pq.centroid is None
    assert pq.codebooks.size == int(pq.sizes[0]) * 256
    assert aniso == -1.0  # UNWEIGHTED
    np.testing.assert_allclose(pq.codebooks[:4], [-0.97000760, -0.24307472, 0.83632696, -0.54823089], rtol=0, atol=5e-9)


def test_save_version0_pq_byte_exact(golden_dir):
    data = open(os.path.join(golden_dir, "version0.pq"), "rb").read()
    pq, _, _, _ = O.OraclePQ.parse(data)
    assert pq.serialize(0) == data
    # v3+ round trip carries magic/version/threshold
    blob = pq.serialize(6, aniso=-1.0)
    pq2, version, aniso, _ = O.OraclePQ.parse(blob)
    assert version == 6 and aniso == -1.0
    np.testing.assert_array_equal(pq2.codebooks, pq.codebooks)


# --- PQLayout literal table: TestProductQuantization.java:305-340 ---------------------------------
PQLAYOUT_TABLE = [
    (1, 1, 1, 0, 1, 1, 1, 0), (1, 2, 1, 0, 1, 1, 2, 0), (10, 1, 10, 0, 1, 1, 10, 0), (10, 2, 10, 0, 1, 1, 20, 0),
    (10, 3, 10, 0, 1, 1, 30, 0), (10, 4, 10, 0, 1, 1, 40, 0), (10, 5, 10, 0, 1, 1, 50, 0), (10, 7, 10, 0, 1, 1, 70, 0),
    (10, 8, 10, 0, 1, 1, 80, 0), (10, 9, 10, 0, 1, 1, 90, 0), (10, 15, 10, 0, 1, 1, 150, 0), (10, 16, 10, 0, 1, 1, 160, 0),
    (10, 17, 10, 0, 1, 1, 170, 0), (10, 31, 10, 0, 1, 1, 310, 0), (10, 32, 10, 0, 1, 1, 320, 0), (10, 33, 10, 0, 1, 1, 330, 0),
    (10, 63, 10, 0, 1, 1, 630, 0), (10, 64, 10, 0, 1, 1, 640, 0), (10, 65, 10, 0, 1, 1, 650, 0), (10, 127, 10, 0, 1, 1, 1270, 0),
    (10, 128, 10, 0, 1, 1, 1280, 0), (10, 129, 10, 0, 1, 1, 1290, 0),
    (1073741823, 1, 1073741823, 0, 1, 1, 1073741823, 0), (1073741823, 2, 1073741823, 0, 1, 1, 2147483646, 0),
    (1073741824, 2, 1073741823, 1, 1, 2, 2147483646, 2), (1000, 1024, 1000, 0, 1, 1, 1024000, 0),
    (2000000, 1024, 2000000, 0, 1, 1, 2048000000, 0), (536870911, 4, 536870911, 0, 1, 1, 2147483644, 0),
    (536870912, 4, 536870911, 1, 1, 2, 2147483644, 4), (100, 1073741824, 1, 0, 100, 100, 1073741824, 0),
]


@pytest.mark.parametrize("row", PQLAYOUT_TABLE)
def test_pq_layout_table(row):
    n, dim, fcv, lcv, fsc, tc, fcb, lcb = row
    lay = O.pq_layout(n, dim)
    assert (lay["fullChunkVectors"], lay["lastChunkVectors"], lay["fullSizeChunks"], lay["totalChunks"],
            lay["fullChunkBytes"], lay["lastChunkBytes"]) == (fcv, lcv, fsc, tc, fcb, lcb)


def test_pq_layout_invalid():  # TestProductQuantization.java:293-297
    for n, d in ((-1, 8), (100, -1), (100, 0), (0, 1)):
        with pytest.raises(ValueError):
            O.pq_layout(n, d)
    lay = O.pq_layout(2**31 - 1, 1 << 10)
    assert lay["lastChunkVectors"] <= lay["fullChunkVectors"] and lay["lastChunkBytes"] <= lay["fullChunkBytes"]


# --- subvector split: ProductQuantization.getSubvectorSizesAndOffsets :535-550 ---------------------
def test_subvector_sizes_offsets():
    s, o = O.subvector_sizes_offsets(768, 96)
    assert (s == 8).all() and (o == np.arange(96) * 8).all()
    s, o = O.subvector_sizes_offsets(10, 3)
    assert s.tolist() == [4, 3, 3] and o.tolist() == [0, 4, 7]
    with pytest.raises(ValueError):
        O.subvector_sizes_offsets(4, 5)


# --- perfect reconstruction: TestProductQuantization.java:54-80 -----------------------------------
def test_perfect_reconstruction():
    rng = np.random.default_rng(0)
    pts = rng.integers(0, 100000, (256, 3)).astype(np.float32)
    # every point is a centroid: D=3, M=2 -> sizes [2,1]
    sizes, offs = O.subvector_sizes_offsets(3, 2)
    cb = np.concatenate([pts[:, 0:2].reshape(-1), pts[:, 2:3].reshape(-1)])
    pq = O.OraclePQ(3, 2, cb)
    vecs = np.repeat(pts, 10, axis=0)
    codes = pq.encode_all(vecs, nthreads=2)
    for i in range(vecs.shape[0]):
        np.testing.assert_array_equal(pq.decode(codes[i]), vecs[i])


def test_encode_first_minimum_wins_and_nan_never_wins():
    # two identical centroids: the lower index must be chosen (strict '<', ProductQuantization.java:513)
    cb = np.zeros((256, 2), np.float32)
    cb[:] = 5.0
    cb[7] = [1.0, 1.0]
    cb[9] = [1.0, 1.0]
    pq = O.OraclePQ(2, 1, cb.reshape(-1))
    assert pq.encode(np.array([1.0, 1.0], np.float32))[0] == 7
    # a NaN distance never wins; all-NaN leaves index 0
    assert pq.encode(np.array([np.nan, 1.0], np.float32))[0] == 0
    cb2 = cb.copy()
    cb2[3] = [np.nan, 0.0]
    pq2 = O.OraclePQ(2, 1, cb2.reshape(-1))
    assert pq2.encode(np.array([1.0, 1.0], np.float32))[0] == 7


# --- ADC (precomputed) == direct per-subspace: TestCompressedVectors.java:230-256 -------------------
@pytest.mark.parametrize("center", [False, True])
def test_precomputed_equals_direct(center):
    rng = np.random.default_rng(42 + center)
    for _ in range(6):
        D = int(rng.integers(4, 257))
        M = int(rng.integers(1, D // 2 + 1))
        sizes, _ = O.subvector_sizes_offsets(D, M)
        cb = rng.uniform(-1, 1, 256 * D).astype(np.float32)
        centroid = rng.uniform(-0.1, 0.1, D).astype(np.float32) if center else None
        pq = O.OraclePQ(D, M, cb, centroid)
        vecs = rng.uniform(-1, 1, (20, D)).astype(np.float32)
        vecs /= np.linalg.norm(vecs, axis=1, keepdims=True)
        codes = pq.encode_all(vecs, nthreads=1)
        q = vecs[0]
        for vsf in (O.EUCLIDEAN, O.DOT_PRODUCT, O.COSINE):
            pre = pq.adc_scores(q, vsf, codes)
            for i in range(20):
                assert abs(pre[i] - pq.direct_score(q, vsf, codes[i])) <= 1e-6


# --- fused == unfused ADC: TestFusedGraphIndex.java:74-114,193-233 --------------------------------
def test_fused_equals_unfused():
    rng = np.random.default_rng(5)
    D, M = 64, 8
    pq = O.OraclePQ(D, M, rng.uniform(-1, 1, 256 * D).astype(np.float32))
    vecs = rng.standard_normal((64, D)).astype(np.float32)
    codes = pq.encode_all(vecs, nthreads=1)
    q = rng.standard_normal(D).astype(np.float32)
    for vsf in (O.EUCLIDEAN, O.DOT_PRODUCT, O.COSINE):
        a = pq.adc_scores(q, vsf, codes, fused=False)
        b = pq.adc_scores(q, vsf, codes, fused=True)
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-6)


# --- triangular code-vs-code table vs direct: TestProductQuantization.java:409-435 (1e-6) ---------
def test_codebook_partial_sums_vs_direct():
    rng = np.random.default_rng(11)
    D, M = 32, 4
    pq = O.OraclePQ(D, M, rng.uniform(-1, 1, 256 * D).astype(np.float32))
    codes = rng.integers(0, 256, (30, M)).astype(np.uint8)
    L = O.lib()
    for vsf in (O.EUCLIDEAN, O.DOT_PRODUCT):
        tri = pq.codebook_partial_sums(vsf)
        for i in range(1, 30):
            got = L.jvo_assemble_and_sum_pq(O._f(tri), M, O._u8(codes[0]), 0, O._u8(codes[i]), 0, 256)
            want = 0.0
            for m in range(M):
                cbm = pq.codebook(m)
                s = int(pq.sizes[m])
                f = O.l2_off if vsf == O.EUCLIDEAN else O.dot_off
                want += f(cbm, int(codes[0, m]) * s, cbm, int(codes[i, m]) * s, s)
            assert abs(got - want) <= 1e-5


# --- NodeQueue order: NodeQueue.java:125-129, NumericUtils.java:49-65 -----------------------------
def test_nodequeue_order():
    assert O.float_to_sortable_int(0.0) == 0
    assert O.float_to_sortable_int(-0.0) < O.float_to_sortable_int(0.0)
    vals = [-np.inf, -1.5, -1e-30, -0.0, 0.0, 1e-30, 0.5, 1.0, np.inf]
    enc = [O.float_to_sortable_int(v) for v in vals]
    assert enc == sorted(enc) and len(set(enc)) == len(enc)
    # higher score wins; on ties the smaller node id wins
    assert O.nodequeue_encode(5, 0.9) > O.nodequeue_encode(3, 0.8)
    assert O.nodequeue_encode(3, 0.8) > O.nodequeue_encode(5, 0.8)
    ids, sc = O.topk(np.array([10, 4, 7, 2], np.int32), np.array([0.5, 0.9, 0.5, 0.1], np.float32), 3)
    assert ids.tolist() == [4, 7, 10] and sc.tolist() == [np.float32(0.9), np.float32(0.5), np.float32(0.5)]


# --- score transforms: VectorSimilarityFunction.java:40,54,67 --------------------------------------
def test_score_transforms():
    assert O.score_from_raw(O.EUCLIDEAN, 3.0) == np.float32(0.25)
    assert O.score_from_raw(O.DOT_PRODUCT, 0.5) == np.float32(0.75)
    assert O.score_from_raw(O.COSINE, -1.0) == 0.0
    a = np.array([1, 0, 0, 0], np.float32)
    assert O.compare(O.COSINE, a, a) == 1.0 and O.compare(O.EUCLIDEAN, a, a) == 1.0


def test_siftsmall_query_fixture(golden_dir):
    raw = np.fromfile(os.path.join(golden_dir, "siftsmall_query.fvecs"), dtype=np.int32).reshape(100, 129)
    assert (raw[:, 0] == 128).all()
    q = raw[:, 1:].view(np.float32)
    assert q.min() >= 0 and q.max() <= 255 and np.all(q == np.round(q))
    # self-match sanity under the oracle (C1 plumbing): nearest neighbour of each query among the 100 is itself
    for i in (0, 17, 99):
        sc = O.compare_many(O.EUCLIDEAN, q[i], q)
        ids, _ = O.topk(None, sc, 1)
        assert ids[0] == i


# ---- NodeQueue / BoundedLongHeap: the reference's own literals (TS/graph/TestNodeQueue.java) ----
def test_nodequeue_literals_from_the_reference():
    nn = O.OracleNodeQueue("min", 2)                     # testNeighborsProduct :36-47
    assert nn.push(2, 0.5) and nn.push(1, 0.2) and nn.push(3, 1.0)
    assert nn.top()[1] == 0.5
    nn.pop()
    assert nn.top()[1] == 1.0
    nn = O.OracleNodeQueue("max", 2)                     # testNeighborsMaxHeap :49-58
    assert nn.push(2, 2) and nn.push(1, 1)
    assert not nn.push(3, 3)
    assert nn.top()[1] == 2.0
    nn.pop()
    assert nn.top()[1] == 1.0
    nn = O.OracleNodeQueue("max")                        # testTopMaxHeap :60-68
    nn.push(1, 2), nn.push(2, 1)
    assert nn.top() == (1, 2.0)
    nn = O.OracleNodeQueue("min")                        # testTopMinHeap :70-78
    nn.push(1, 0.5), nn.push(2, -0.5)
    assert nn.top() == (2, -0.5)
    nn = O.OracleNodeQueue("min", 2)                     # testMaxSizeQueue :90-102
    nn.push(1, 1), nn.push(2, 2)
    assert nn.size() == 2 and nn.top()[0] == 1
    nn.push(3, 3)
    assert nn.size() == 2 and nn.top()[0] == 2
    rng = np.random.default_rng(0)                       # testUnboundedQueue :104-120
    nn = O.OracleNodeQueue("max")
    scores = rng.random(256).astype(np.float32)
    for i, s in enumerate(scores):
        nn.push(i, float(s))
    assert nn.top() == (int(np.argmax(scores)), float(scores.max()))
    # equal scores: the smaller node id wins (NodeQueue.encode javadoc :104-124)
    nn = O.OracleNodeQueue("max")
    nn.push(7, 0.25), nn.push(3, 0.25), nn.push(9, 0.25)
    assert [nn.pop() for _ in range(3)] == [3, 7, 9]
