#!/usr/bin/env python3
"""The register-table bound form for EUCLIDEAN searches (round 6: lower bucket edges, gs_host.h gs_ubr_build_ref(l2 = true)) next to the
plain pair form on an index the engine builds itself: 1M x 768 un-normalised mixture, PQ-96, degree 32, L2.
Prints one JSON object: ms per batch and QPS with gs_ubr = 1 / 0, the dropped share, recall@10 against brute force, identical results.
usage: python scripts/l2_ubr_bench.py [--n 1000000] [--queries 65536] [--rerank 80]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import jvector_amd as J
from benchlib import Mixture
from jvector_amd.builder import build_hierarchical


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--queries", type=int, default=65536)
    ap.add_argument("--rerank", type=int, default=80)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    ctx = J.HipContext(0, stream=torch.cuda.current_stream().cuda_stream)
    VSF = J.VectorSimilarityFunction.EUCLIDEAN
    N, Q, D, M = args.n, args.queries, 768, 96
    mix = Mixture(D, seed=5, device=dev)
    scale = 1.0 + 0.5 * torch.rand(N, 1, generator=torch.Generator(device=dev).manual_seed(9), device=dev)   # norms differ: L2 != cosine order
    base = (mix.sample(N, seed=5) * scale).contiguous()
    q = mix.sample(Q, seed=6) * 1.25
    g = torch.Generator(device=dev).manual_seed(4)
    pq = J.ProductQuantization.compute(ctx, base[torch.randperm(N, generator=g, device=dev)[:128_000]].contiguous(), M, seed=4)
    vs = J.VectorSet(ctx, base)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    t0 = time.perf_counter()
    levels, entry, entry_level, nb0, _ = build_hierarchical(ctx, pq, cv, base, VSF, overflow=2.0)
    ctx.sync()
    build_s = time.perf_counter() - t0
    fused = J.FusedPQ.build(ctx, cv, nb0)
    graph = J.GraphIndex(ctx, N, levels, entry, entry_level)
    s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=Q)
    qq = q[:2048]
    d2 = (qq * qq).sum(1, keepdim=True) - 2.0 * qq @ base.T + (base * base).sum(1)[None, :]
    gt = torch.topk(-d2, 10, dim=1).indices.cpu().numpy()
    out = {"n": N, "queries": Q, "rerankK": args.rerank, "build_s": build_s}
    res = {}
    for ubr in (1, 0):
        ctx.set_option("gs_ubr", ubr)
        ids, sc, st = s.search(q, VSF, 10, args.rerank, return_stats=True)
        ctx.sync()
        d0 = ctx.stat("gs_ubr_dropped")
        t0 = time.perf_counter()
        for _ in range(3):
            ids, sc, st = s.search(q, VSF, 10, args.rerank, return_stats=True)
        ctx.sync()
        dt = (time.perf_counter() - t0) / 3
        ids_h = np.asarray(torch.as_tensor(ids).cpu())
        rec = float(np.mean([len(set(ids_h[i]) & set(gt[i])) / 10.0 for i in range(2048)]))
        st_h = np.asarray(torch.as_tensor(st).cpu())
        res[ubr] = (ids_h, np.asarray(torch.as_tensor(sc).cpu()), st_h)
        out[f"gs_ubr={ubr}"] = {"ms_per_batch": dt * 1e3, "qps": Q / dt, "recall_at_10": rec, "last_ubr": ctx.stat("gs_last_ubr"),
                                "dropped_share_of_visited": (ctx.stat("gs_ubr_dropped") - d0) / 3 / max(float(st_h[:, 0].sum()), 1.0),
                                "avg_visited": float(st_h[:, 0].mean()), "avg_expanded": float(st_h[:, 1].mean())}
    ctx.set_option("gs_ubr", None)
    out["identical_results"] = bool(all(np.array_equal(a, b) for a, b in zip(res[1], res[0])))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
