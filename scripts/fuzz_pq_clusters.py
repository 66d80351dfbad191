"""One-off validation aid: product quantizers with 1..256 clusters (kept padded to 256 rows per sub-space on the device side) against the
oracle working with the true cluster count — random D, M (ragged sub-vectors included), k, global centroid: codes (incl. exact hits
on centroid 0, which tie with its padded copies), table scores for the three similarity functions, self-magnitudes, the wire form
and — for small cases — compute / refine.  usage: python scripts/fuzz_pq_clusters.py [seconds] [seed]   (FUZZ_MOCK=1: CPU mock)"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
if os.environ.get("FUZZ_MOCK") == "1":
    sys.path.insert(0, os.path.join(ROOT, "tests", "mock"))
    os.environ.setdefault("JVECTOR_HIP_HOST_THREADS", "1")
    import build_mock
    import jvector_amd._lib as L
    lib = C.CDLL(build_mock.build())
    for table in (L.SIGNATURES, L.COMPAT_SIGNATURES, L.FORMAT_SIGNATURES):
        for name, (res, args) in table.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    L._lib = lib
import jvector_amd as J  # noqa: E402
from oracle import oracle as O  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = J.HipContext(0)
VSF = list(J.VectorSimilarityFunction)
t_end = time.time() + budget
cases = trained = 0
while time.time() < t_end:
    D = int(rng.integers(1, 80))
    M = int(rng.integers(1, min(D, 16) + 1))
    k = int(rng.choice([1, 2, 3, 15, 16, 50, 100, 255, 256]))
    center = bool(rng.random() < 0.5)
    cb = rng.standard_normal(k * D).astype(np.float32)
    cen = (rng.standard_normal(D) * 0.2).astype(np.float32) if center else None
    tag = dict(seed=seed, case=cases, D=D, M=M, k=k, center=center)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb, cen, cluster_count=k)
    opq = O.OraclePQ(D, M, cb, cen, k=k)
    assert pq.get_cluster_count() == k, tag
    v = rng.standard_normal((400, D)).astype(np.float32)
    hit = rng.integers(0, k, (40, M)).astype(np.uint8)
    hit[:, 0] = 0
    v[:40] = np.stack([opq.decode(c) for c in hit])
    codes = pq.encode_all(v)
    assert np.array_equal(codes, opq.encode_all(v)) and int(codes.max()) < k, tag
    rc = rng.integers(0, k, (500, M)).astype(np.uint8)
    q = rng.standard_normal((2, D)).astype(np.float32)
    cv = J.PQVectors(ctx, pq, rc)
    for vsf in VSF:
        got = cv.precomputed_score_function_for(q, vsf).similarity_to_range(0, len(rc))
        for i in range(2):
            assert np.array_equal(got[i], opq.adc_scores(q[i], int(vsf), rc)), (tag, str(vsf))
    assert np.array_equal(pq.self_magnitudes(), opq.cache_self_magnitudes()), tag
    blob = pq.write(6)
    assert blob == opq.serialize(6), tag
    pq2 = J.ProductQuantization.load(ctx, blob)
    assert pq2.get_cluster_count() == k and np.array_equal(pq2.encode_all(v[:50]), codes[:50]), tag
    if D >= M and D // M <= 16 and rng.random() < 0.35 and k >= 2:
        n = int(rng.integers(max(k, 40), max(k, 40) + 600))
        x = (rng.standard_normal((n, D)) + 2.0 * rng.standard_normal((1, D))).astype(np.float32)
        sd = int(rng.integers(1, 1000))
        want, _ = O.pq_train(x, M, k=k, globally_center=center, seed=sd)
        got = J.ProductQuantization.compute(ctx, x, M, cluster_count=k, globally_center=center, seed=sd)
        assert got.write(6) == want.serialize(6), (tag, "train", n, sd)
        y = rng.standard_normal((max(k, 50), D)).astype(np.float32)
        assert got.refine(y, 1, seed=sd + 1).write(6) == want.refine(y, 1, seed=sd + 1).serialize(6), (tag, "refine")
        trained += 1
    cases += 1
print(f"fuzz_pq_clusters: {cases} quantizers ({trained} trained + refined), all identical to the oracle (seed {seed}, {budget:.0f} s)")
