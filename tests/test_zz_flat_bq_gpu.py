"""The flat search's two-stage threshold filter (k_adc_bq.hip, option adc_bq, default on): a 7-bit bound scan for sixteen queries per
LDS word drops what cannot reach the query's threshold, the exact ADC score is computed for the survivors only.  At sizes where
jv_hip_search_flat takes the filtered path (N >= 2^18) the results — approximate top-rerankK without a rerank, exact top-K with one —
must equal the single-stage filter's (adc_bq = 0) and the oracle's bit for bit: three similarity functions, PQ-16 and PQ-96, a query
count that is not a multiple of sixteen, duplicated vectors (ties at the threshold), a zero query and a NaN query (no usable bound:
the call falls back to the exact filter)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import jvector_amd as J
from jvector_amd import VectorSimilarityFunction as VSF
from oracle import oracle as O


@pytest.fixture(scope="module")
def ctx():
    c = J.HipContext(0)
    yield c
    c.close()


def problem(seed, N, D, M, Q):
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((200, D)).astype(np.float32)
    vecs = (centers[rng.integers(0, 200, N)] + 0.35 * rng.standard_normal((N, D)).astype(np.float32)).astype(np.float32)
    vecs /= np.linalg.norm(vecs, axis=1, keepdims=True)
    vecs[1::7] = vecs[0:-1:7][: len(vecs[1::7])]            # every seventh vector twice: equal scores around every threshold
    queries = (vecs[rng.integers(0, N, Q)] + 0.05 * rng.standard_normal((Q, D))).astype(np.float32)
    sizes, offs = O.subvector_sizes_offsets(D, M)
    pick = rng.choice(N, 256, replace=False)
    cb = np.concatenate([vecs[pick, offs[m]: offs[m] + sizes[m]].reshape(-1) for m in range(M)])
    return vecs, queries, cb


@pytest.mark.parametrize("D,M,N,Q,k1", [(128, 16, 300_000, 37, 400), (768, 96, 270_000, 20, 50), (256, 32, 262_144, 16, 120)])
def test_two_stage_filter_equals_single_stage_and_oracle(ctx, D, M, N, Q, k1):
    vecs, queries, cb = problem(D + M, N, D, M, Q)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    opq = O.OraclePQ(D, M, cb)
    vs = J.VectorSet(ctx, vecs)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    codes = cv.get(0, N)
    s_rr = J.FlatSearcher(ctx, pq, cv, vs, max_queries=Q)
    s_nr = J.FlatSearcher(ctx, pq, cv, None, max_queries=Q)
    try:
        for vsf in VSF:
            got = {}
            for bq in (1, 0):
                ctx.set_option("adc_bq", bq)
                before = ctx.stat("adc_bq_calls")
                got[bq] = (s_nr.search(queries, vsf, k1, 0), s_rr.search(queries, vsf, 10, k1))
                assert ctx.stat("adc_bq_calls") - before == (2 if bq else 0), (vsf, bq, ctx.stat("adc_bq_fallbacks"))
            for a, b in zip(got[1], got[0]):
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), vsf
            (ids_nr, sc_nr), (ids, sc) = got[1]
            for q in range(0, Q, 5):
                approx = opq.adc_scores(queries[q], int(vsf), codes)
                cand, cs = O.topk(None, approx, k1)
                assert np.array_equal(ids_nr[q], cand) and np.array_equal(sc_nr[q], cs), (vsf, q)
                wi, ws = O.topk(cand, O.compare_many(int(vsf), queries[q], vecs[cand]), 10)
                assert np.array_equal(ids[q], wi) and np.array_equal(sc[q], ws), (vsf, q)
    finally:
        ctx.set_option("adc_bq", None)


def test_queries_without_a_usable_bound_fall_back(ctx):
    D, M, N, Q = 128, 16, 300_000, 5
    vecs, queries, cb = problem(3, N, D, M, Q)
    queries = queries.copy()
    queries[1] = 0.0
    queries[3, 7] = np.nan
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    vs = J.VectorSet(ctx, vecs)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    s = J.FlatSearcher(ctx, pq, cv, None, max_queries=Q)
    try:
        for vsf in VSF:
            ctx.set_option("adc_bq", 0)
            want = s.search(queries, vsf, 100, 0)
            ctx.set_option("adc_bq", 1)
            got = s.search(queries, vsf, 100, 0)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1], equal_nan=True), vsf
    finally:
        ctx.set_option("adc_bq", None)
