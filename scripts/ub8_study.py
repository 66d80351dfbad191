"""Design study (not part of the product or the tests): how many of the neighbours the traversal scores could be DROPPED without an exact
score if every wave held an 8-bit UPPER-BOUND table of its query's ADC entries (24 KB at PQ-96)?  A neighbour whose bound lies below the
score of the K-th best exactly scored node so far can never be popped (>= rerankK better candidates exist), so it needs no exact score —
96 codebook gathers saved.  Replays 32 queries through the oracle's GraphSearcher with its visit log on (bench index: engine-built,
one improveConnections pass) and simulates the rule in numpy.
usage (GPU box): python scripts/ub8_study.py [N] [rerankK]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jvector_amd as J  # noqa: E402
from benchlib import Mixture  # noqa: E402
from jvector_amd.builder import build_hierarchical  # noqa: E402
from oracle import oracle as O  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
RK = int(sys.argv[2]) if len(sys.argv) > 2 else 76
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
levels, entry, entry_level, nbrs_dev, bstats = build_hierarchical(ctx, pq, cv, base, VSF, max_degree=32, beam_width=100, alpha=1.2, overflow=2.0, improve=1)
codes = cv.get(0, N)
codes = codes.cpu().numpy() if hasattr(codes, "cpu") else np.asarray(codes)
lv = [(None if ids is None else np.asarray(ids), np.asarray(nb)) for ids, nb in levels]
og = O.OracleGraph(N, lv, int(entry), int(entry_level))
opq = O.OraclePQ.parse(pq.write(6))[0]
L = O.lib()
log = np.empty(1 << 16, np.int32)
tot = 0
res = {}
for qi in range(len(queries)):
    L.jvo_set_visit_log(log.ctypes.data_as(C.POINTER(C.c_int32)), len(log))
    og.search(opq, codes, None, queries[qi:qi + 1], int(VSF), 10, RK, fused=True)
    n = int(L.jvo_visit_log_count())
    L.jvo_set_visit_log(None, 0)
    seq = codes[log[:n]].astype(np.int64)                       # [n, M] codes in scoring order
    table, amag, bmag = opq.decoder(queries[qi], int(VSF), True)
    T = np.asarray(table, np.float64).reshape(M, 256)
    A = np.asarray(amag, np.float64).reshape(M, 256)
    rows = np.arange(M)[None, :]
    ent = T[rows, seq]                                           # [n, M] exact entries
    raw = ent.sum(1)
    nm = A[rows, seq].sum(1)
    fin = lambda r: (1.0 + r / np.sqrt(nm * float(bmag))) / 2.0  # noqa: E731
    score = fin(raw)
    lo, hi = T.min(1), T.max(1)
    tot += n
    for name, bits, per_sub in (("8-bit, one scale", 8, False), ("8-bit, scale per subspace", 8, True), ("6-bit, scale per subspace", 6, True),
                                ("4-bit, scale per subspace", 4, True)):
        lv_ = (1 << bits) - 1
        S = (hi - lo) / lv_ if per_sub else np.full(M, (hi - lo).max() / lv_)
        S = np.maximum(S, 1e-12)
        ub = lo[None, :] + S[None, :] * (np.floor((ent - lo[None, :]) / S[None, :]) + 1.0)
        U = fin(ub.sum(1))
        for K in (100, 200):
            kept = []            # exact scores of the nodes that were exactly scored so far
            pruned = 0
            thr = -np.inf
            for i in range(n):
                if len(kept) >= K and U[i] < thr:
                    pruned += 1
                    continue
                kept.append(score[i])
                if len(kept) >= K and (len(kept) % 16 == 0 or len(kept) == K):
                    thr = np.partition(np.asarray(kept), len(kept) - K)[len(kept) - K]
            res[(name, K)] = res.get((name, K), 0) + pruned
    # the ideal (bound = exact score)
    for K in (100, 200):
        kept, pruned, thr = [], 0, -np.inf
        for i in range(n):
            if len(kept) >= K and score[i] < thr:
                pruned += 1
                continue
            kept.append(score[i])
            if len(kept) >= K and (len(kept) % 16 == 0 or len(kept) == K):
                thr = np.partition(np.asarray(kept), len(kept) - K)[len(kept) - K]
        res[("exact score as the bound (ideal)", K)] = res.get(("exact score as the bound (ideal)", K), 0) + pruned
print(f"N={N} rerankK={RK}: {tot / len(queries):.0f} scored nodes per query")
for (name, K), v in sorted(res.items()):
    print(f"  {name:36s} threshold = {K}-th best exact score so far: {v / tot:.3f} of the scored neighbours need no exact score")
