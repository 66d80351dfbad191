"""MFMA tile scan (exact_dense_kernel) on the MI355X: TFLOP/s by similarity function and query count, for the occupancy builds
JVECTOR_HIP_ED_WAVES selects (one process per setting: the choice is read once).  usage: python scripts/dense_bench.py [N] [D]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import jvector_amd as J
from jvector_amd import VectorSimilarityFunction as VSF

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
dev = torch.device("cuda:0")
ctx = J.HipContext(0)
g = torch.Generator(device=dev).manual_seed(1)
base = torch.randn(N, D, device=dev, generator=g)
base = (base / base.norm(dim=1, keepdim=True)).contiguous()
vs = J.VectorSet(ctx, base)
out = {"n": N, "dim": D, "ed_waves": os.environ.get("JVECTOR_HIP_ED_WAVES", "default")}
for nq in (256, 1024, 4096):
    q = torch.randn(nq, D, device=dev, generator=g).contiguous()
    o = torch.empty(nq, N, dtype=torch.float32, device=dev)
    for vsf in VSF:
        vs.scan(q[:128], vsf, out=o[:128], dense=True)
        ctx.profile(True)
        for _ in range(3):
            vs.scan(q, vsf, out=o, dense=True)
        ms, n = ctx.profile_read("exact")
        ctx.profile(False)
        out[f"q{nq}_{vsf.name}"] = {"ms": ms / n, "tflops": nq * N * 2 * D / (ms / n) / 1e9, "frac_mfma_f32": nq * N * 2 * D / (ms / n) / 1e9 / 157.3}
    del o
print(json.dumps(out))
