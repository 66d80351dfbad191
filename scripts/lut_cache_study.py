"""Analysis aid (not part of the product or the tests): would a small per-query cache of the partial dot products
p(m, code) — kept in LDS by the traversal kernel — absorb most of the codebook gathers?  Builds the bench's index with the
engine, replays 32 queries through the oracle's GraphSearcher restatement with its visit log on, and simulates a direct-mapped
cache of S slots per subspace (slot = code mod S) over the codes of the nodes each query scores, in order.
usage (GPU box): python scripts/lut_cache_study.py [N] [rerankK]"""
import os
import sys
import ctypes as C

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jvector_amd as J  # noqa: E402
from benchlib import Mixture  # noqa: E402
from jvector_amd.builder import build_hierarchical  # noqa: E402
from oracle import oracle as O  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
RK = int(sys.argv[2]) if len(sys.argv) > 2 else 110
D, M = 768, 96
dev = torch.device("cuda", 0)
ctx = J.HipContext(0)
VSF = J.VectorSimilarityFunction.COSINE
mix = Mixture(D, seed=5, device=dev)
base = mix.sample(N, seed=5)
queries = mix.sample(32, seed=6).cpu().numpy()
g = torch.Generator(device=dev).manual_seed(4)
sample = base[torch.randperm(N, generator=g, device=dev)[:128_000]].contiguous()
pq = J.ProductQuantization.compute(ctx, sample, M, seed=4)
vs = J.VectorSet(ctx, base)
cv = J.PQVectors.encode_and_build(ctx, pq, vs)
levels, entry, entry_level, nbrs_dev, bstats = build_hierarchical(ctx, pq, cv, base, VSF, max_degree=32, beam_width=100, alpha=1.2)
codes = cv.get(0, N)
codes = codes.cpu().numpy() if hasattr(codes, "cpu") else np.asarray(codes)
lv = [(None if ids is None else np.asarray(ids), np.asarray(nb)) for ids, nb in levels]
og = O.OracleGraph(N, lv, int(entry), int(entry_level))
opq = O.OraclePQ.parse(pq.write(6))[0]
L = O.lib()
log = np.empty(1 << 16, np.int32)
res = {}
tot = 0
for qi in range(len(queries)):
    L.jvo_set_visit_log(log.ctypes.data_as(C.POINTER(C.c_int32)), len(log))
    og.search(opq, codes, None, queries[qi:qi + 1], int(VSF), 10, RK, fused=True)
    n = int(L.jvo_visit_log_count())
    L.jvo_set_visit_log(None, 0)
    seq = codes[log[:n]]                       # [n, M] codes in scoring order
    tot += n
    for S in (8, 16, 24, 32, 64, 256):
        hits = 0
        for m in range(M):
            col = seq[:, m].astype(np.int32)
            slot = col % S
            order = np.argsort(slot, kind="stable")          # per slot, accesses in time order
            cs, ss = col[order], slot[order]
            hits += int(((ss[1:] == ss[:-1]) & (cs[1:] == cs[:-1])).sum())   # same slot, same code as its previous occupant
        res[S] = res.get(S, 0) + hits
print(f"N={N} rerankK={RK}: {tot / len(queries):.0f} scored nodes per query")
for S in sorted(res):
    print(f"  direct-mapped, {S:3d} slots per subspace ({M * S * 8 / 1024:.1f} KB of {{tag, value}} pairs per query): hit rate {res[S] / (tot * M):.3f}")
