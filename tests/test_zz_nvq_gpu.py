"""NVQ (the reference's compressed rerank codec) on the GPU through the C ABI, bit for bit against the oracle's restatement
of the scalar reference path (oracle/jv_nvq.c): global mean, encoded bytes + parameters, scores for the three similarity
functions, and NVQ rows as the reranker of the flat and graph searchers.  The functions take `ctx` so that
tests/test_nvq_cpu.py re-runs them on the mock device (host logic only there)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import jvector_amd as J
from jvector_amd import VectorSimilarityFunction as VSF
from oracle import jv_writers as W
from oracle import oracle as O


@pytest.fixture(scope="module")
def ctx():
    c = J.HipContext(0)
    yield c
    c.close()


def make_vectors(seed, n, D, kind="gauss"):
    rng = np.random.default_rng(seed)
    if kind == "gauss":
        return rng.standard_normal((n, D)).astype(np.float32)
    if kind == "unit":          # the bench's shape: clustered, unit norm
        c = rng.standard_normal((20, D)).astype(np.float32)
        v = c[rng.integers(0, 20, n)] + 0.3 * rng.standard_normal((n, D)).astype(np.float32)
        return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    if kind == "offset":        # a large mean: what the global-mean subtraction is for
        return (5.0 + 0.01 * rng.standard_normal((n, D))).astype(np.float32)
    raise ValueError(kind)


@pytest.mark.parametrize("n,D", [(1, 8), (33, 7), (1000, 100), (4097, 64)])
def test_global_mean_bit_exact(ctx, n, D):
    X = make_vectors(n + D, n, D, "offset")
    vs = J.VectorSet(ctx, X)
    nvq = J.NVQuantization.compute(ctx, vs, 1)
    want = O.OracleNVQ.compute(X, 1).mean
    assert np.array_equal(nvq.global_mean(), want)


ENCODE_SHAPES = [(64, 1, "gauss"), (64, 2, "unit"), (100, 3, "gauss"), (33, 5, "offset"), (768, 2, "unit"), (256, 8, "gauss"),
                 (17, 17, "gauss"), (1536, 4, "unit")]


@pytest.mark.parametrize("D,S,kind", ENCODE_SHAPES)
@pytest.mark.parametrize("learn", [True, False])
def test_encode_bit_exact(ctx, D, S, kind, learn, n=200):
    X = make_vectors(D * 31 + S, n, D, kind)
    o = O.OracleNVQ.compute(X, S, learn)
    wb, wp = o.encode_all(X)
    vs = J.VectorSet(ctx, X)
    nvq = J.NVQuantization.compute(ctx, vs, S).set_learn(learn)
    assert np.array_equal(nvq.global_mean(), o.mean)
    gb, gp = nvq.encode_all(vs).get()
    assert np.array_equal(gp.view(np.uint32), wp.view(np.uint32)), np.argwhere(gp.view(np.uint32) != wp.view(np.uint32))[:5]
    assert np.array_equal(gb, wb), np.argwhere(gb != wb)[:5]
    if learn and D > S:   # the search did something: not every sub-vector kept the default growth rate
        assert (wp[:, :, 2] != np.float32(1e-2)).any()


def test_encode_edge_values(ctx):
    """constant sub-vectors (max == min: every derived number is inf / NaN), zeros of both signs, huge and tiny magnitudes,
    NaN and infinities: the bytes and parameters still follow the reference's arithmetic bit for bit"""
    D, S = 24, 3
    X = make_vectors(5, 40, D)
    X[0] = 1.5                               # constant vector
    X[1, :8] = 0.0                           # one constant sub-vector (after the mean is subtracted: not constant any more)
    X[2, 3] = np.float32(1e30)
    X[3, 5] = np.float32(-1e-30)
    X[4, 9] = np.inf
    X[5, 10] = np.nan
    X[6] = -0.0
    mean = np.zeros(D, np.float32)           # NVQuantization.create with a zero mean keeps the constants constant
    mean[12] = -0.0
    o = O.OracleNVQ(mean, S)
    wb, wp = o.encode_all(X)
    nvq = J.NVQuantization.create(ctx, mean, S)
    gb, gp = nvq.encode_all(J.VectorSet(ctx, X)).get()
    assert np.array_equal(gp.view(np.uint32), wp.view(np.uint32)), np.argwhere(gp.view(np.uint32) != wp.view(np.uint32))[:5]
    assert np.array_equal(gb, wb), np.argwhere(gb != wb)[:5]


SCORE_SHAPES = [(64, 1, 300, 5, 70), (100, 3, 500, 3, 64), (768, 2, 700, 4, 95), (33, 5, 90, 2, 1), (256, 8, 400, 9, 129),
                (1536, 4, 300, 2, 50)]


@pytest.mark.parametrize("D,S,n,Q,B", SCORE_SHAPES)
def test_scores_bit_exact(ctx, D, S, n, Q, B):
    X = make_vectors(D + S + n, n, D, "unit" if D % 2 == 0 else "gauss")
    rng = np.random.default_rng(D)
    queries = (X[rng.integers(0, n, Q)] + 0.1 * rng.standard_normal((Q, D))).astype(np.float32)
    o = O.OracleNVQ.compute(X, S)
    o.encode_all(X)
    nvq = J.NVQuantization.create(ctx, o.mean, S)
    nv = J.NVQVectors(ctx, nvq, o.bytes, o.params)
    ords = rng.integers(0, n, (Q, B)).astype(np.int32)
    ords[0, 0] = -1
    ords[-1, -1] = n            # outside the set: -inf, as everywhere in this engine
    for vsf in VSF:
        got = nv.scores(queries, vsf, ords)
        want = o.scores(queries, int(vsf), ords)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (vsf, np.argwhere(got != want)[:5])
    # the same through the jv_vectors view every rerank takes
    vs = nv.as_vector_set()
    got = vs.scores(queries, VSF.COSINE, ords)
    assert np.array_equal(got.view(np.uint32), o.scores(queries, 2, ords).view(np.uint32))


def test_scores_after_partial_upload(ctx):
    """rows replaced in place: the derived tables (and the cosine normalisation sums) follow"""
    D, S, n = 64, 2, 300
    X = make_vectors(1, n, D)
    o = O.OracleNVQ.compute(X, S)
    o.encode_all(X)
    nvq = J.NVQuantization.create(ctx, o.mean, S)
    nv = J.NVQVectors(ctx, nvq, o.bytes, o.params)
    q = X[:3].copy()
    ords = np.tile(np.arange(40, dtype=np.int32), (3, 1))
    assert np.array_equal(nv.scores(q, VSF.COSINE, ords), o.scores(q, 2, ords))
    Y = make_vectors(2, 20, D)
    o2 = O.OracleNVQ(o.mean, S)
    b2, p2 = o2.encode_all(Y)
    nv.upload(10, b2, p2)
    o.bytes[10:30], o.params[10:30] = b2, p2
    for vsf in VSF:
        assert np.array_equal(nv.scores(q, vsf, ords), o.scores(q, int(vsf), ords)), vsf


def test_reference_tolerances_hold_on_the_device(ctx):
    """TS/quantization/TestCompressedVectors.testNVQEncodings (:171-228) with the GPU as the implementation: mean score error
    against the full-resolution similarity under the reference's tolerance, every similarity function"""
    d, n = 256, 512
    rng = np.random.default_rng(3)
    X = rng.standard_normal((n, d)).astype(np.float32)
    Qs = rng.standard_normal((10, d)).astype(np.float32)
    Qs /= np.linalg.norm(Qs, axis=1, keepdims=True)       # VectorUtil.l2normalize(q)
    vs = J.VectorSet(ctx, X)
    ords = np.tile(np.arange(n, dtype=np.int32), (10, 1))
    for S in (1, 2, 4, 8):
        for learn in (False, True):
            nvq = J.NVQuantization.compute(ctx, vs, S).set_learn(learn)
            nv = nvq.encode_all(vs)
            for vsf in VSF:
                got = nv.scores(Qs, vsf, ords)
                exact = vs.scores(Qs, vsf, ords)
                if vsf == VSF.DOT_PRODUCT:
                    vv = np.array([O.compare(1, X[j], X[j]) for j in range(n)], np.float32)
                    err = float(np.mean(np.abs(got - exact) / np.abs(vv)[None, :]))
                else:
                    err = float(np.mean(np.abs(got - exact)))
                tol = 0.0005 * (d / 256.0) * (10 if vsf == VSF.COSINE else 4 if vsf == VSF.DOT_PRODUCT else 1)
                assert err <= tol, (S, learn, vsf, err, tol)


def _pq_problem(ctx, seed, N, D, M):
    rng = np.random.default_rng(seed)
    v = make_vectors(seed, N, D, "unit")
    sizes, offs = O.subvector_sizes_offsets(D, M)
    pick = rng.choice(N, 256, replace=False)
    cb = np.concatenate([v[pick, offs[m]: offs[m] + sizes[m]].reshape(-1) for m in range(M)])
    q = (v[rng.integers(0, N, 9)] + 0.05 * rng.standard_normal((9, D))).astype(np.float32)
    return v, cb, q


def test_flat_search_reranks_with_nvq(ctx, N=4000, D=64, M=8, S=2):
    v, cb, q = _pq_problem(ctx, 11, N, D, M)
    opq = O.OraclePQ(D, M, cb)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    vs = J.VectorSet(ctx, v)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    codes = cv.get(0, N)
    nvq = J.NVQuantization.compute(ctx, vs, S)
    nv = nvq.encode_all(vs)
    b, p = nv.get()
    o = O.OracleNVQ(nvq.global_mean(), S).set_rows(b, p)
    searcher = J.FlatSearcher(ctx, pq, cv, nv.as_vector_set(), max_queries=16)
    for vsf in VSF:
        ids, sc = searcher.search(q, vsf, 10, 80)
        with o.as_reranker():
            wi, ws = opq.search_flat(codes, v, q, int(vsf), 10, 80)
        assert np.array_equal(ids, wi), vsf
        assert np.array_equal(sc, ws), vsf
        # and it is not the full-resolution answer in disguise
        fi, fs = J.FlatSearcher(ctx, pq, cv, vs, max_queries=16).search(q, vsf, 10, 80)
        assert not np.array_equal(sc, fs)


@pytest.mark.parametrize("traversal", ["host", "device"])
def test_graph_search_reranks_with_nvq(ctx, traversal):
    import test_graph_search as G
    D, M, S = 128, 16, 2
    v, lv, entry, entry_level, cb, q = G.build_problem(77, N=3000, D=D, M=M, levels=2)
    N = len(v)
    opq = O.OraclePQ(D, M, cb)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    vs = J.VectorSet(ctx, v)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    codes = cv.get(0, N)
    nvq = J.NVQuantization.compute(ctx, vs, S)
    nv = nvq.encode_all(vs)
    b, p = nv.get()
    o = O.OracleNVQ(nvq.global_mean(), S).set_rows(b, p)
    og = O.OracleGraph(N, lv, entry, entry_level)
    graph = J.GraphIndex(ctx, N, lv, entry, entry_level).set_traversal(traversal)
    fused = J.FusedPQ(ctx, pq, G.fused_blocks(codes, lv[0][1]), lv[0][1])
    s = J.GraphSearcher(ctx, graph, pq, cv, fused, nv.as_vector_set(), max_queries=64)
    for vsf in VSF:
        ids, sc, stats = s.search(q, vsf, 10, 40, return_stats=True)
        with o.as_reranker():
            wi, ws, wst = og.search(opq, codes, v, q, int(vsf), 10, 40, fused=True)
        assert np.array_equal(stats, wst), vsf
        assert np.array_equal(ids, wi), vsf
        assert np.array_equal(sc, ws), vsf
    # GraphSearcher objects (threshold / rerankFloor / resume) rerank through the same rows
    s2 = J.GraphSearcher(ctx, graph, pq, cv, fused, nv.as_vector_set(), max_queries=16)
    got = s2.search_ex(q[:6], VSF.COSINE, 10, 40, threshold=0.0, rerank_floor=0.0)
    got1 = s2.resume(5, 20)
    with o.as_reranker():
        for i in range(6):
            osr = og.searcher(opq, codes, v, 2, fused=True)
            G._same(got[i], osr.search(q[i], 10, 40, 0.0, 0.0), ("nvq searcher", i))
            G._same(got1[i], osr.resume(5, 20), ("nvq searcher resume", i))
            osr.close()
    s.close()
    s2.close()


def test_float_entry_points_refuse_nvq_rows(ctx):
    D, S, n = 32, 2, 100
    X = make_vectors(9, n, D)
    vs = J.VectorSet(ctx, X)
    nvq = J.NVQuantization.compute(ctx, vs, S)
    nv = nvq.encode_all(vs)
    view = nv.as_vector_set()
    with pytest.raises(J.UnsupportedError, match="NVQ rows"):
        view.scan(X[:2], VSF.COSINE)
    with pytest.raises(J.UnsupportedError, match="NVQ rows"):
        J.NVQuantization.compute(ctx, view, 2)
    with pytest.raises(J.UnsupportedError, match="NVQ rows"):
        nvq.encode_all(view)
    with pytest.raises(ValueError, match="less than or equal to the vector dimension"):
        J.NVQuantization.create(ctx, np.zeros(4, np.float32), 5)
    other = J.NVQuantization.create(ctx, np.zeros(D, np.float32), S)
    from jvector_amd._lib import check
    with pytest.raises(ValueError, match="another NVQuantization"):
        check(ctx._lib.jv_hip_nvq_encode(ctx._h, other._h, vs._h, 0, n, nv._h, 0))


def test_nvq_formats_round_trip_on_device(ctx):
    """NVQVectors.write of device rows == the oracle writer's bytes; load_nvqvectors / load_index(NVQ_VECTORS, SEPARATED_NVQ)
    bring them back and the loaded index reranks with them"""
    import test_graph_search as G
    from jvector_amd import formats as F
    D, M, S = 64, 8, 3
    v, lv, entry, entry_level, cb, q = G.build_problem(5, N=1500, D=D, M=M, levels=2)
    N = len(v)
    vs = J.VectorSet(ctx, v)
    nvq = J.NVQuantization.compute(ctx, vs, S)
    nv = nvq.encode_all(vs)
    b, p = nv.get()
    blob = nv.write()
    assert blob == W.write_nvqvectors(nvq.global_mean(), S, b, p)
    nvq2, nv2 = F.load_nvqvectors(ctx, blob)
    b2, p2 = nv2.get()
    assert np.array_equal(b2, b) and np.array_equal(p2.view(np.uint32), p.view(np.uint32))
    assert np.array_equal(nvq2.global_mean(), nvq.global_mean())
    # an index whose features are FUSED_PQ + NVQ (inline or separated), as Grid builds for NVQ reranking (EX/Grid.java:508-514)
    opq = O.OraclePQ(D, M, cb)
    codes = opq.encode_all(v)
    nb0 = lv[0][1]
    l0 = [list(r[r >= 0]) for r in nb0]
    upper = [(lv[1][1].shape[1], {int(n): [int(x) for x in row if x >= 0] for n, row in zip(lv[1][0], lv[1][1])})]
    o = O.OracleNVQ(nvq.global_mean(), S).set_rows(b, p)
    og = O.OracleGraph(N, lv, entry, entry_level)
    for separated in (False, True):
        data = W.write_odgi(6, D, l0, nb0.shape[1], entry, upper, codes=codes, pq_block=opq.serialize(6),
                            nvq=(nvq.global_mean(), S, b, p), nvq_separated=separated)
        idx = F.load_index(ctx, data, W.write_pqvectors(opq.serialize(6), codes))
        assert ("SEPARATED_NVQ" if separated else "NVQ_VECTORS") in idx.host.features
        gb, gp = idx.nvq_vectors.get()
        assert np.array_equal(gb, b) and np.array_equal(gp.view(np.uint32), p.view(np.uint32))
        s = idx.searcher(max_queries=64)
        ids, sc = s.search(q, VSF.DOT_PRODUCT, 10, 40)
        with o.as_reranker():
            wi, ws, _ = og.search(opq, codes, v, q, 1, 10, 40, fused=True)
        assert np.array_equal(ids, wi) and np.array_equal(sc, ws)
        s.close()
