"""Time-boxed differential sweep of the NVQ kernels against the oracle on the GPU: random dimensions and sub-vector splits (ragged
ones included), data at several scales with constants, zeros of both signs, NaN and infinities mixed in, `learn` on and off;
then scores for the three similarity functions over rows that mix encoder-made parameters with hand-made ones (negative / tiny /
huge growth rates, zero and inverted ranges, NaN) so that wavefronts take the short-division path, the IEEE path, and both within
one launch.  Everything is compared bit for bit.  usage (GPU box): python scripts/fuzz_nvq.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jvector_amd as J  # noqa: E402
from oracle import oracle as O  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = J.HipContext(0)
VSF = J.VectorSimilarityFunction
t_end = time.time() + budget
cases = checks = 0


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_bits(a, b):
    """bit equality, except that two NaNs are equal whatever their sign / payload: which NaN an invalid operation or a NaN operand
    yields is a property of the machine (x86: 0xFFC00000 for inf - inf, AMD: 0x7FC00000), not of the reference's arithmetic"""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    both = np.isnan(a) & np.isnan(b)
    return np.array_equal(np.where(both, 0, bits(a)), np.where(both, 0, bits(b)))


def fail(what, **kw):
    print("MISMATCH", what, dict(seed=seed, case=cases, **kw))
    sys.exit(1)


while time.time() < t_end:
    D = int(rng.choice([int(rng.integers(1, 40)), int(rng.integers(40, 400)), 128, 768, 1021]))
    S = min(D, int(rng.choice([1, 2, 3, int(rng.integers(1, 10))])))
    n = int(rng.integers(1, 300))
    learn = bool(rng.random() < 0.6)
    scale = float(rng.choice([1e-3, 1.0, 1.0, 50.0, 1e6]))
    X = (rng.standard_normal((n, D)) * scale).astype(np.float32)
    if rng.random() < 0.3:
        X[rng.integers(0, n)] = float(rng.choice([0.0, -0.0, 3.25]))            # a constant row
    if rng.random() < 0.2:
        X[rng.integers(0, n), rng.integers(0, D)] = float(rng.choice([np.inf, -np.inf, np.nan, 1e38, -1e-38]))
    tag = dict(D=D, S=S, n=n, learn=learn, scale=scale)
    vs = J.VectorSet(ctx, X)
    o = O.OracleNVQ.compute(X, S, learn)
    nvq = J.NVQuantization.compute(ctx, vs, S).set_learn(learn)
    if not same_bits(nvq.global_mean(), o.mean):
        fail("global mean", **tag)
    wb, wp = o.encode_all(X, nthreads=16)
    nv = nvq.encode_all(vs)
    gb, gp = nv.get()
    if not same_bits(gp, wp):
        fail("encode parameters", where=np.argwhere((bits(gp) != bits(wp)) & ~(np.isnan(gp) & np.isnan(wp)))[:3].tolist(), **tag)
    if not np.array_equal(gb, wb):
        fail("encode bytes", where=np.argwhere(gb != wb)[:3].tolist(), **tag)
    checks += 3
    # ---- scores: some rows get hand-made parameters (the IEEE-division path, and mixtures with the short path inside one wavefront)
    b2, p2 = wb.copy(), wp.copy()
    weird = rng.random(n) < rng.choice([0.0, 0.1, 0.5])
    for i in np.nonzero(weird)[0]:
        s = int(rng.integers(0, S))
        kind = int(rng.integers(0, 6))
        lo, hi = sorted(rng.standard_normal(2).astype(np.float32).tolist())
        gr = float(rng.choice([1e-6, -0.7, 1e-2, 5.0, 19.9, 300.0, 1e-30]))
        if kind == 0:
            p2[i, s] = [lo, hi, gr, 0.0]
        elif kind == 1:
            p2[i, s] = [lo, lo, gr, 0.0]                      # zero range
        elif kind == 2:
            p2[i, s] = [hi, lo, gr, 0.1]                      # inverted range, non-zero midpoint
        elif kind == 3:
            p2[i, s] = [lo, hi, np.nan, 0.0]
        elif kind == 4:
            p2[i, s] = [0.0, 0.0, 0.0, 0.0]                   # QuantizedVector.createEmpty
        else:
            p2[i, s] = [lo * 1e20, hi * 1e20, gr, 0.0]
        b2[i] = rng.integers(0, 256, D, dtype=np.uint8)
    o2 = O.OracleNVQ(o.mean, S).set_rows(b2, p2)
    nv2 = J.NVQVectors(ctx, nvq, b2, p2)
    Q, B = int(rng.integers(1, 6)), int(rng.choice([1, 17, 64, 95, 130]))
    queries = (rng.standard_normal((Q, D)) * scale).astype(np.float32)
    ords = rng.integers(-1, n + 1, (Q, B)).astype(np.int32)
    for vsf in VSF:
        got = nv2.scores(queries, vsf, ords)
        want = o2.scores(queries, int(vsf), ords)
        if not same_bits(got, want):
            g, w = np.asarray(got), np.asarray(want)
            fail("scores", vsf=str(vsf), weird=int(weird.sum()), where=np.argwhere((bits(g) != bits(w)) & ~(np.isnan(g) & np.isnan(w)))[:3].tolist(), **tag)
        checks += 1
    cases += 1
print(f"fuzz_nvq: {cases} cases, {checks} checks, all bit-identical (seed {seed}, {budget:.0f} s)")
