#!/usr/bin/env python3
"""Random configurations of the builder in reference order against the oracle's one-thread GraphIndexBuilder (one node per batch:
adjacency, scores and marks byte for byte — tests/test_builder_reference_order.py::check_reference_order).  Draws: node count,
dimension / subspaces, maxDegree, beam width, alpha, neighborOverflow, similarity function, duplicated vectors (tied scores), improve
passes.  usage: fuzz_builder_reference_order.py [--cases N] [--seed S] [--device mock|gpu]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "mock"))
import test_builder_reference_order as T  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--device", choices=["mock", "gpu"], default="mock")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    cases = []
    for _ in range(args.cases):
        M = int(rng.choice([4, 8, 16]))
        big = args.device == "gpu"
        N = int(rng.integers(40, 2500 if big else 500))
        cases.append(dict(N=N, D=8 * M, M=M, max_degree=int(rng.integers(3, 25)), beam=int(rng.integers(4, 70)), vsf=int(rng.integers(0, 3)),
                          dup=int(rng.integers(0, max(1, N // 8))) if rng.random() < 0.4 else 0, improve=int(rng.random() < 0.4),
                          alpha=float(rng.choice([1.0, 1.2, 1.4])), overflow=float(rng.choice([1.0, 1.2, 1.5, 2.0])), seed=int(rng.integers(1, 1000))))

    def run(J, ctx):
        dev = torch.device("cuda", 0) if args.device == "gpu" else torch.device("cpu")
        t0 = time.time()
        for i, c in enumerate(cases):
            if c["improve"]:
                c["N"] = min(c["N"], 1200 if args.device == "gpu" else 300)
            out, want = T.check_reference_order(J, ctx, dev, c["N"], c["D"], c["M"], c["max_degree"], c["beam"], J.VectorSimilarityFunction(c["vsf"]), dup=c["dup"],
                                                improve=c["improve"], alpha=c["alpha"], overflow=c["overflow"], seed=c["seed"])
            print(f"case {i}: {c} -> identical ({int((out >= 0).sum())} edges, {time.time() - t0:.0f} s)", flush=True)
        print(f"{len(cases)} random reference-order builds identical to the oracle")

    if args.device == "mock":
        T._on_the_mock(run)
    else:
        import jvector_amd as J
        ctx = J.HipContext(0)
        run(J, ctx)
        ctx.close()


if __name__ == "__main__":
    main()
