#!/usr/bin/env python3
"""The device traversal's GENERIC kernels (gs_body.h CH16 = 0: any sub-vector geometry) next to the host searcher and to the
specialised build, on indexes the engine builds itself (the builder searches through the same kernels):
  glove-200-like  200-d, PQ-25 (8-dim sub-vectors, M not a multiple of 16; codebook rows read as 16-byte words)
  glove-100-like  100-d, PQ-12 (ragged sub-vectors 9 9 9 9 8 ...: scalar codebook reads)
  sift-like       128-d, PQ-16 — once through the specialised kernels, once with the generic ones forced (option gs_generic)
Prints one JSON object: QPS (device / host traversal), recall@10 against brute force, build seconds.
usage: python scripts/generic_bench.py [--n 200000] [--queries 16384]"""
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
    ap.add_argument("--n", type=int, default=200_000)
    ap.add_argument("--queries", type=int, default=16384)
    ap.add_argument("--host-queries", type=int, default=2048)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    ctx = J.HipContext(0, stream=torch.cuda.current_stream().cuda_stream)
    VSF = J.VectorSimilarityFunction.COSINE
    N, Q = args.n, args.queries
    out = {"n": N, "queries": Q, "shapes": {}}
    for name, D, M, force in (("200d_pq25", 200, 25, False), ("100d_pq12_ragged", 100, 12, False), ("128d_pq16_specialised", 128, 16, False),
                              ("128d_pq16_generic_forced", 128, 16, True)):
        ctx.set_option("gs_generic", 1 if force else None)
        mix = Mixture(D, seed=5, device=dev)
        base = mix.sample(N, seed=5)
        q = mix.sample(Q, seed=6)
        g = torch.Generator(device=dev).manual_seed(4)
        pq = J.ProductQuantization.compute(ctx, base[torch.randperm(N, generator=g, device=dev)[:min(N, 128_000)]].contiguous(), M, seed=4)
        vs = J.VectorSet(ctx, base)
        cv = J.PQVectors.encode_and_build(ctx, pq, vs)
        t0 = time.perf_counter()
        levels, entry, entry_level, nb0, _ = build_hierarchical(ctx, pq, cv, base, VSF, overflow=2.0)
        ctx.sync()
        build_s = time.perf_counter() - t0
        fused = J.FusedPQ.build(ctx, cv, nb0)
        graph = J.GraphIndex(ctx, N, levels, entry, entry_level)
        s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=Q)
        res = {"D": D, "M": M, "build_s": build_s}
        gt = torch.topk(q[:2048] @ base.T, 10, dim=1).indices.cpu().numpy()
        for rk in (40, 80):
            ctx.reset_stats()
            ids, _ = s.search(q, VSF, 10, rk)
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(3):
                ids, _ = s.search(q, VSF, 10, rk)
            ctx.sync()
            t = (time.perf_counter() - t0) / 3
            ids = np.asarray(ids if not torch.is_tensor(ids) else ids.cpu())
            rec = float(np.mean([len(set(a.tolist()) & set(b.tolist())) / 10.0 for a, b in zip(ids[:2048], gt)]))
            res[f"device_qps_rk{rk}"] = Q / t
            res[f"recall_rk{rk}"] = rec
            res["device_calls"] = ctx.stat("gs_calls_device")
            res["host_calls"] = ctx.stat("gs_calls_host")
        graph.set_traversal("host")
        qh = q[: args.host_queries].contiguous()
        sh = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=args.host_queries)
        ids_h, _ = sh.search(qh, VSF, 10, 80)
        t0 = time.perf_counter()
        ids_h, _ = sh.search(qh, VSF, 10, 80)
        ctx.sync()
        res["host_qps_rk80"] = args.host_queries / (time.perf_counter() - t0)
        ids_h = np.asarray(ids_h if not torch.is_tensor(ids_h) else ids_h.cpu())
        res["host_equals_device"] = bool(np.array_equal(ids_h, ids[: args.host_queries]))
        out["shapes"][name] = res
        graph.close()
        del s, sh, fused, cv, vs, base
    ctx.set_option("gs_generic", None)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
