"""One-off validation aid: a time-boxed differential sweep of the scoring / quantization kernels against the oracle on random
shapes — any D, any M <= D (ragged sub-vector sizes included), with and without a global centroid, clustered data with duplicated
rows and coarse grids (ties), random ordinal lists with -1 padding.  Checked bit for bit: ProductQuantization.encode, the ADC
tables' scores (PQDecoder and FusedPQDecoder kinds; scan, gather and multi-query scan forms), exact scores (gather, scan, pair
form), top-k under the NodeQueue order, and the two-pass flat search.
usage (GPU box): python scripts/fuzz_kernels.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jvector_amd as J  # noqa: E402
from oracle import oracle as O  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = J.HipContext(0)
VSF = J.VectorSimilarityFunction
t_end = time.time() + budget
cases = checks = 0


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(np.nan_to_num(a, nan=-7.5), np.nan_to_num(b, nan=-7.5))


def fail(what, **kw):
    print("MISMATCH", what, dict(seed=seed, case=cases, **kw))
    sys.exit(1)


while time.time() < t_end:
    D = int(rng.choice([int(rng.integers(1, 40)), int(rng.integers(40, 300)), 128, 768, 1536, 1021]))
    M = int(rng.choice([1, D, max(1, D // 8), int(rng.integers(1, min(D, 200) + 1))]))
    M = min(M, D, 256)
    N = int(rng.integers(50, 4000))
    center = bool(rng.random() < 0.5)
    kind = rng.integers(0, 3)
    if kind == 0:
        v = rng.standard_normal((N, D)).astype(np.float32)
    elif kind == 1:                                           # coarse grid: many equal distances
        v = (np.round(rng.standard_normal((N, D)) * 2) / 2).astype(np.float32)
    else:                                                     # clustered, with duplicated rows
        c = rng.standard_normal((8, D)).astype(np.float32)
        v = (c[rng.integers(0, 8, N)] + 0.1 * rng.standard_normal((N, D))).astype(np.float32)
        v[N // 2:] = v[: N - N // 2]
    sizes, offs = O.subvector_sizes_offsets(D, M)
    pick = rng.integers(0, N, 256)
    cb = np.concatenate([v[pick, offs[m]: offs[m] + sizes[m]].reshape(-1) for m in range(M)]).astype(np.float32)
    centroid = (rng.standard_normal(D) * 0.1).astype(np.float32) if center else None
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb, centroid)
    opq = O.OraclePQ(D, M, cb, centroid)
    tag = dict(D=D, M=M, N=N, center=center, kind=int(kind))
    # --- encode
    vs = J.VectorSet(ctx, v)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    codes = cv.get(0, N)
    want_codes = opq.encode_all(v)
    if not np.array_equal(codes, want_codes):
        fail("encode", **tag)
    checks += 1
    Q = int(rng.integers(1, 6))
    q = (v[rng.integers(0, N, Q)] + rng.choice([0.0, 0.05, 1.0]) * rng.standard_normal((Q, D))).astype(np.float32)
    for vsf in VSF:
        if vsf == VSF.COSINE and kind == 1:
            continue                                          # zero rows on the grid: NaN cosine is covered by the unit tests
        # --- ADC: scan over all codes (single- and multi-query kernels), gather over random ordinals
        sf = cv.precomputed_score_function_for(q, vsf)
        got = sf.similarity_to_range(0, N)
        B = int(rng.integers(1, 70))
        ords = rng.integers(-1, N, (Q, B)).astype(np.int32)
        gg = sf.similarity_to(ords)
        for i in range(Q):
            w = opq.adc_scores(q[i], int(vsf), codes)
            if not same(got[i], w):
                fail("adc scan", vsf=str(vsf), q=i, **tag)
            wg = np.where(ords[i] >= 0, w[np.clip(ords[i], 0, N - 1)], -np.inf).astype(np.float32)
            if not same(gg[i], wg):
                fail("adc gather", vsf=str(vsf), q=i, **tag)
        # --- exact: gather, scan, node pairs
        eg = vs.scores(q, vsf, ords)
        es = vs.scan(q, vsf)
        for i in range(Q):
            w = O.compare_many(int(vsf), q[i], v)
            if not same(es[i], w):
                fail("exact scan", vsf=str(vsf), q=i, **tag)
            wg = np.where(ords[i] >= 0, w[np.clip(ords[i], 0, N - 1)], -np.inf).astype(np.float32)
            if not same(eg[i], wg):
                fail("exact gather", vsf=str(vsf), q=i, **tag)
        n1 = rng.integers(0, N, Q).astype(np.int32)
        ep = vs.pair_scores(vsf, n1, ords)
        for i in range(Q):
            w = O.compare_many(int(vsf), v[n1[i]], v)
            wg = np.where(ords[i] >= 0, w[np.clip(ords[i], 0, N - 1)], -np.inf).astype(np.float32)
            if not same(ep[i], wg):
                fail("exact pairs", vsf=str(vsf), q=i, **tag)
        # --- top-k of the scan rows (ties by node id) and the two-pass flat search
        k = int(rng.integers(1, min(N, 80) + 1))
        ti, ts = J.topk(ctx, np.ascontiguousarray(got), k)
        rk = int(rng.integers(k, min(N, 4 * k) + 1))
        fi, fs = J.FlatSearcher(ctx, pq, cv, vs, max_queries=8).search(q, vsf, k, rk)
        for i in range(Q):
            w = opq.adc_scores(q[i], int(vsf), codes)
            wi, ws = O.topk(None, w, k)
            if not (np.array_equal(ti[i][: len(wi)], wi) and same(ts[i][: len(wi)], ws)):
                fail("topk", vsf=str(vsf), q=i, k=k, **tag)
            cand, _ = O.topk(None, w, rk)
            ex = O.compare_many(int(vsf), q[i], v[cand])
            wi, ws = O.topk(cand, ex, k)
            if not (np.array_equal(fi[i][: len(wi)], wi) and same(fs[i][: len(wi)], ws)):
                fail("flat search", vsf=str(vsf), q=i, k=k, rk=rk, **tag)
        checks += 6
    cases += 1
    for o in (cv, vs, pq):
        o.close()
print(f"fuzz: {cases} random shapes, {checks} kernel checks, all bit-identical to the oracle (seed {seed}, {budget:.0f} s)")
