#!/usr/bin/env python3
"""Tiny driver for PMC collection on the frontier kernel (graph mode's GPU kernel): random codes + a random
regular graph (no torch-heavy index build, so rocprofv3 --pmc stays cheap).  C3 shapes: M=96, maxDegree 32, cosine.
usage: rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python scripts/frontier_pmc.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import jvector_amd as J

N, D, M, DEG, Q = 1_000_000, 768, 96, 32, 4096
rng = np.random.default_rng(0)
dev = torch.device("cuda", 0)
ctx = J.HipContext(0)
cb = rng.standard_normal(256 * D).astype(np.float32) * 0.05
pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
codes = torch.randint(0, 256, (N, M), dtype=torch.uint8, device=dev)
cv = J.PQVectors(ctx, pq, codes)
nbrs = torch.randint(0, N, (N, DEG), dtype=torch.int32, device=dev)
blocks = codes[nbrs.long().reshape(-1)].reshape(N, DEG * M).contiguous()
fused = J.FusedPQ(ctx, pq, blocks, nbrs)
graph = J.GraphIndex(ctx, N, [(None, nbrs.cpu().numpy())], 0, 0)
searcher = J.GraphSearcher(ctx, graph, pq, cv, fused, None, max_queries=Q)
queries = torch.randn(Q, D, device=dev)
for _ in range(2):
    ids, sc, st = searcher.search(queries, J.VectorSimilarityFunction.COSINE, 10, 100, return_stats=True)
ctx.sync()
print("avg expanded", st[:, 1].mean(), "avg visited", st[:, 0].mean())
