"""NVQ kernels on the MI355X: global mean, encode rate, and the rerank gather against NVQ rows vs float32 rows at the headline's
shape (Q x rerankK candidates of D = 768).  Kernel times come from the engine's own HIP events on its stream
(HipContext.profile); rates are printed as one JSON object.  usage: python scripts/nvq_bench.py [N] [D] [S] [Q] [B]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import jvector_amd as J
from jvector_amd import VectorSimilarityFunction as VSF

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
S = int(sys.argv[3]) if len(sys.argv) > 3 else 2
Q = int(sys.argv[4]) if len(sys.argv) > 4 else 16384
B = int(sys.argv[5]) if len(sys.argv) > 5 else 95
dev = torch.device("cuda:0")
ctx = J.HipContext(0)
g = torch.Generator(device=dev).manual_seed(1)
base = torch.randn(N, D, device=dev, generator=g)
base = (base / base.norm(dim=1, keepdim=True)).contiguous()
vs = J.VectorSet(ctx, base)
out = {"n": N, "dim": D, "subvectors": S, "queries": Q, "candidates_per_query": B}

t0 = time.perf_counter()
nvq = J.NVQuantization.compute(ctx, vs, S)
ctx.sync()
out["global_mean_s"] = time.perf_counter() - t0
out["global_mean_GBps"] = N * D * 4 / out["global_mean_s"] / 1e9

for learn in (True, False):
    nvq.set_learn(learn)
    ctx.profile(True)
    nv = nvq.encode_all(vs)
    ms, _ = ctx.profile_read("encode")
    ctx.profile(False)
    key = "encode_learn" if learn else "encode_nolearn"
    out[key] = {"ms": ms, "vectors_per_s": N / (ms / 1e3), "dims_per_s": N * D / (ms / 1e3),
                "loss_chain_evaluations_per_s": (N * D * 41 / (ms / 1e3)) if learn else None}
    if not learn:
        nv.close()
nvq.set_learn(True)
nv = nvq.encode_all(vs)
view = nv.as_vector_set()

queries = base[torch.randint(0, N, (Q,), device=dev, generator=g)] + 0.05 * torch.randn(Q, D, device=dev, generator=g)
queries = queries.contiguous()
ords = torch.randint(0, N, (Q, B), device=dev, generator=g, dtype=torch.int32)
for vsf in VSF:
    res = {}
    for name, rows, row_bytes in (("nvq", view, D + 16 * S + 4 + (4 if vsf == VSF.COSINE else 0)), ("float", vs, 4 * D + 4 + (4 if vsf == VSF.COSINE else 0))):
        rows.scores(queries[:256], vsf, ords[:256])          # warm-up (norm tables)
        ctx.profile(True)
        for _ in range(5):
            sc = rows.scores(queries, vsf, ords)
        ms, n = ctx.profile_read("exact")
        ctx.profile(False)
        avg = ms / n
        res[name] = {"avg_ms": avg, "GBps": Q * B * row_bytes / (avg / 1e3) / 1e9, "candidates_per_s": Q * B / (avg / 1e3),
                     "dims_per_s": Q * B * D / (avg / 1e3), "bytes_per_candidate": row_bytes}
    res["speedup_over_float_rows"] = res["float"]["avg_ms"] / res["nvq"]["avg_ms"]
    out["gather_" + vsf.name] = res
# CPU leg (bounded): the oracle's restatement of the scalar reference path on this box's host cores — encode 4096 vectors on 16
# threads, score 256 queries x B candidates single-threaded — with the results compared bit for bit against the GPU's
if os.environ.get("NVQ_BENCH_CPU", "1") != "0":
    from oracle import oracle as O
    nc = 4096
    Xh = base[:nc].cpu().numpy()
    o = O.OracleNVQ(nvq.global_mean(), S)
    t0 = time.perf_counter()
    o.encode_all(Xh, nthreads=16)
    dt = time.perf_counter() - t0
    gb, gp = nv.get(0, nc)
    out["cpu_encode"] = {"vectors_per_s": nc / dt, "threads": 16, "sample": f"{nc} vectors", "kind": "port",
                         "identical_to_gpu": bool(np.array_equal(gb, o.bytes) and np.array_equal(gp.view(np.uint32), o.params.view(np.uint32)))}
    qh = queries[:256].cpu().numpy()
    oh = torch.randint(0, nc, (256, B), generator=torch.Generator().manual_seed(3), dtype=torch.int32).numpy()
    t0 = time.perf_counter()
    want = o.scores(qh, int(VSF.COSINE), oh)
    dt = time.perf_counter() - t0
    got = nv.scores(qh, VSF.COSINE, oh)
    out["cpu_scores"] = {"candidates_per_s": 256 * B / dt, "threads": 1, "sample": f"256 queries x {B} candidates, COSINE", "kind": "port",
                         "identical_to_gpu": bool(np.array_equal(got.view(np.uint32), want.view(np.uint32)))}
print(json.dumps(out))
