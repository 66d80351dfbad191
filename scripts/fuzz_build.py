"""One-off validation aid: the build-side and training kernels against the oracle on random parameters — the batched robust prune
(jv_hip_retain_diverse), the PQ build-score provider (pair table, code-pair scores, decode, direct scores), anisotropic encode, PQ
training / refinement (byte-identical wire form), and the GraphSearcher objects (threshold / rerankFloor / resume) on more seeds.
Reuses the parity tests' own checkers with random arguments.   usage (GPU box): python scripts/fuzz_build.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import jvector_amd as J  # noqa: E402
import test_retain_diverse as TR  # noqa: E402
import test_zz_build_score_gpu as TB  # noqa: E402
import test_zz_anisotropic_gpu as TA  # noqa: E402
import test_zz_pq_train_gpu as TT  # noqa: E402
import test_graph_search as TG  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 150.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = J.HipContext(0)
t_end = time.time() + budget
n = {"retain_diverse": 0, "build_score": 0, "anisotropic": 0, "pq_train": 0, "searcher_objects": 0}
while time.time() < t_end:
    what = rng.integers(0, 5)
    try:
        if what == 0:
            D = int(rng.choice([64, 128, 256, 768]))
            M = int(rng.choice([D // 8, D // 4, 7]))
            Cn = int(rng.integers(1, 160))
            N = int(rng.integers(max(300, Cn + 50), 2500))
            case = (int(rng.integers(0, 10_000)), N, D, M, int(rng.integers(1, 48)), Cn, int(rng.integers(1, 65)),
                    float(rng.choice([1.0, 1.2, 1.4, 2.0])))
            args = case
            TR.run_through_cabi(J, ctx, [case])
            n["retain_diverse"] += 1
        elif what == 1:
            D = int(rng.integers(8, 600))
            M = int(rng.integers(1, min(D, 120) + 1))
            args = (D, M)
            TB.test_build_score_provider_matches_oracle(ctx, D, M, bool(rng.random() < 0.5))
            n["build_score"] += 1
        elif what == 2:
            D = int(rng.choice([32, 50, 64, 128, 200]))
            M = int(rng.choice([max(1, D // 8), 7, 5]))
            thr = float(rng.choice([0.1, 0.2, 0.5, 0.8]))
            args = (D, M, thr)
            centers = rng.standard_normal((12, D)).astype(np.float32)
            v = (centers[rng.integers(0, 12, 1500)] + 0.6 * rng.standard_normal((1500, D))).astype(np.float32)
            v /= np.linalg.norm(v, axis=1, keepdims=True)
            sizes, offs = TA.O.subvector_sizes_offsets(D, M)
            cen = (0.05 * rng.standard_normal(D)).astype(np.float32) if rng.random() < 0.5 else None
            b = v if cen is None else (v - cen).astype(np.float32)
            pick = rng.choice(1500, 256, replace=False)
            cb = np.concatenate([b[pick, offs[m]: offs[m] + sizes[m]].reshape(-1) for m in range(M)]).astype(np.float32)
            opq = TA.O.OraclePQ(D, M, cb, cen)
            pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb, cen).set_anisotropic_threshold(thr)
            got = pq.encode_all(v[:300])
            want = np.stack([opq.encode_anisotropic(v[i], thr) for i in range(300)])
            assert np.array_equal(got, want)
            n["anisotropic"] += 1
        elif what == 3:
            D = int(rng.choice([16, 26, 32, 48, 64]))
            M = int(rng.choice([2, 3, 4, D // 8 if D >= 16 else 1]))
            args = (D, M)
            TT.test_train_refine_write(ctx, D, max(1, M), bool(rng.random() < 0.5))
            n["pq_train"] += 1
        else:
            args = ()
            TG.run_searcher_object_cases(J, ctx, cases=int(rng.integers(1, 5)))
            n["searcher_objects"] += 1
    except AssertionError as e:
        print("MISMATCH", ["retain_diverse", "build_score", "anisotropic", "pq_train", "searcher_objects"][what], args, e)
        raise
print(f"fuzz: {n} (seed {seed}, {budget:.0f} s): every case bit-identical to the oracle")
