#!/usr/bin/env python3
"""Tiny driver for profiling the MFMA tile form of the exact scan (k_exact_dense.hip): Q x 1M x 768 cosine, a few launches,
plus the bit-exact scan of the same shape for comparison.  Checks 64 outputs against the CPU specification first.
usage: rocprofv3 --kernel-trace --stats ... -- python scripts/dense_pmc.py [Q]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import jvector_amd as J
from oracle import oracle as O

Q = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N, D = 1_000_000, 768
dev = torch.device("cuda", 0)
ctx = J.HipContext(0)
g = torch.Generator(device=dev)
g.manual_seed(1)
vecs = torch.randn(N, D, generator=g, device=dev)
queries = torch.randn(Q, D, generator=g, device=dev)
vs = J.VectorSet(ctx, vecs)
VSF = J.VectorSimilarityFunction.COSINE
out = torch.empty(Q, N, dtype=torch.float32, device=dev)
vs.scan(queries, VSF, out=out, dense=True)
ctx.sync()
want = O.dense_scan(O.COSINE, queries[:2].cpu().numpy(), vecs[:32].cpu().numpy())
print("matches the fmaf-chain specification:", bool(np.array_equal(out[:2, :32].cpu().numpy(), want)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for dense in (True, False):
    nq = Q if dense else min(Q, 64)
    e0.record()
    for _ in range(3):
        vs.scan(queries[:nq], VSF, out=out[:nq], dense=dense)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"{'dense' if dense else 'exact'} scan {nq}x{N}x{D}: {ms:.3f} ms, {nq * N * 2 * D / ms / 1e9:.1f} TFLOP/s (2 flop per MAC), "
          f"{N * 4 * D / ms / 1e6:.0f} GB/s rows-once")
