"""Batched Vamana construction driven by the engine's own scoring (BASELINE config 5: "GPU-batched neighbor scoring + PQ
encode"; SURVEY §8 f.2 + Appendix C).

The reference inserts nodes one by one from many threads (GraphIndexBuilder.addGraphNode, B/graph/GraphIndexBuilder.java:
605-659): search the current graph for the new node (beamWidth candidates scored with the BuildScoreProvider's PQ score
function), robust-prune them (VamanaDiversityProvider.retainDiverse), link, and backlink with a re-prune when a
neighbour's list overflows (ConcurrentNeighborMap.insertDiverse / backlink, :104-163).  Its result is nondeterministic
(thread interleaving) and its control flow is host code, which SURVEY §8 keeps out of scope; what is IN scope is the scoring
it calls per node.  This module batches exactly those calls over thousands of concurrent inserts and keeps the control flow
as array plumbing (torch) around four engine entry points:

  * candidates   jv_hip_graph_search on the graph built so far (device traversal over a device-resident, mutable adjacency —
                 GraphIndex.on_device), approximate PQ scores (PQDecoder.similarityTo), topK = rerankK = beam width
  * prune        jv_hip_retain_diverse (the reference's alpha-ramped robust prune, selections identical to its sequential loop)
  * backlinks    jv_hip_code_pair_scores (diversityFunctionFor(s).similarityTo(x): a neighbour s scores its merged list),
                 then jv_hip_retain_diverse again for the lists that overflow maxDegree
  * encode       jv_hip_pq_encode_into (done by the caller: PQVectors.encode_and_build)

Insertion order follows the usual batch-parallel schedule for Vamana (prefix doubling: batch sizes 1, 2, 4, ... capped),
so early nodes are searched against a small graph and later batches see a well-connected one.  Within a batch the inserts do
not see each other — the batch analogue of the reference's concurrent inserts, which also miss nodes in flight except for the
`concurrently inserting` set it scores explicitly (:823-838).
"""
from __future__ import annotations

import time

import numpy as np
import torch

from .engine import GraphIndex, GraphSearcher, PQBuildScoreProvider, VectorSimilarityFunction


class BuildStats(dict):
    pass


def _sync(ctx, t):
    """the engine's stream and torch's may differ: fence both ways around a hand-over of the adjacency"""
    ctx.sync()
    if t.is_cuda:
        torch.cuda.synchronize(t.device)


def build_vamana(ctx, pq, pq_vectors, vectors, vsf, max_degree=32, beam_width=100, alpha=1.2, max_batch=131072, seed=11,
                 search_batch=65536, log=None, out=None):
    """vectors: [N, D] float32 torch tensor on the engine's device (the insert queries); pq_vectors: their PQ codes (PQVectors).
    out: optional preallocated [N, max_degree] int32 tensor for the adjacency.
    Returns (neighbors [N, max_degree] int32 device tensor, -1 padded; entry_node; BuildStats)."""
    dev = vectors.device
    N = int(vectors.shape[0])
    R = int(max_degree)
    g = torch.Generator(device="cpu").manual_seed(seed)
    perm = torch.randperm(N, generator=g).to(dev)
    nbrs = out if out is not None else torch.empty((N, R), dtype=torch.int32, device=dev)
    nbrs.fill_(-1)
    bsp = PQBuildScoreProvider(ctx, pq_vectors, vsf)
    # entry point: the inserted node closest to the mean of a sample (the reference re-centres on the medoid at cleanup)
    entry = int(perm[0])
    graph = GraphIndex.on_device(ctx, nbrs, entry).set_traversal("device")
    searcher = GraphSearcher(ctx, graph, pq, pq_vectors, None, None, max_queries=1024)
    stats = BuildStats(search_s=0.0, prune_s=0.0, backlink_s=0.0, batches=0, reprunes=0)
    t_all = time.perf_counter()
    lo = 1                        # perm[0] is the seed node
    while lo < N:
        hi = min(N, lo + min(max_batch, lo))   # prefix doubling: a batch never exceeds what the graph already holds
        batch = perm[lo:hi]
        B = int(batch.shape[0])
        k = min(beam_width, lo)    # cannot ask for more candidates than inserted nodes
        # ---- 1. candidate search on the graph built so far ----
        t0 = time.perf_counter()
        if searcher.luts.capacity < min(B, search_batch):
            searcher = GraphSearcher(ctx, graph, pq, pq_vectors, None, None, max_queries=min(max(2 * B, 1024), search_batch))
        cand = torch.empty(B, k, dtype=torch.int32, device=dev)
        csc = torch.empty(B, k, dtype=torch.float32, device=dev)
        for s in range(0, B, search_batch):
            q = vectors[batch[s:s + search_batch].long()].contiguous()
            ids, sc = searcher.search(q, vsf, k, k)
            cand[s:s + search_batch], csc[s:s + search_batch] = ids, sc
        _sync(ctx, nbrs)
        stats["search_s"] += time.perf_counter() - t0
        # ---- 2. robust prune of every new node's candidates (sorted best first by the search) ----
        t0 = time.perf_counter()
        count = (cand >= 0).sum(dim=1).to(torch.int32)
        sel, nsel, _ = bsp.retain_diverse(cand, csc, R, alpha, cand_count=count)
        sel = torch.as_tensor(sel)
        chosen = torch.where(sel >= 0, torch.gather(cand, 1, sel.clamp(min=0).long()), torch.full_like(sel, -1))
        nbrs[batch.long()] = chosen
        _sync(ctx, nbrs)
        stats["prune_s"] += time.perf_counter() - t0
        # ---- 3. backlinks: v joins the list of each of its chosen neighbours s; lists that overflow are re-pruned ----
        t0 = time.perf_counter()
        src = batch.view(-1, 1).expand(-1, R).reshape(-1)
        dst = chosen.reshape(-1)
        ok = dst >= 0
        src, dst = src[ok].to(torch.int32), dst[ok].long()
        order = torch.argsort(dst, stable=True)
        src, dst = src[order], dst[order]
        uniq, inv, cnt = torch.unique_consecutive(dst, return_inverse=True, return_counts=True)
        start = torch.cumsum(cnt, 0) - cnt
        pos = torch.arange(dst.shape[0], device=dev) - start[inv]            # rank of the back edge within its target
        K_new = int(min(int(cnt.max()) if cnt.numel() else 0, 2 * R))         # cap the merged list: existing R + up to 2R new
        if uniq.numel():
            keep = pos < K_new
            merged = torch.full((uniq.shape[0], R + K_new), -1, dtype=torch.int32, device=dev)
            merged[:, :R] = nbrs[uniq]
            merged[inv[keep], (R + pos[keep])] = src[keep]
            deg = (merged >= 0).sum(dim=1)
            fits = deg <= R
            # lists that still fit: append (compact the -1 holes to the right)
            comp = torch.sort((merged < 0).to(torch.int8), dim=1, stable=True).indices
            packed = torch.gather(merged, 1, comp)
            nbrs[uniq[fits]] = packed[fits, :R]
            over = (~fits).nonzero().squeeze(1)
            if over.numel():
                tgt = uniq[over].to(torch.int32)
                lst = packed[over]                                           # [P, R + K_new], -1 padded on the right
                sc = torch.as_tensor(bsp.diversity_scores(tgt, lst.contiguous()))   # -inf for the padding
                o2 = torch.argsort(sc, dim=1, descending=True, stable=True)
                lst, sc = torch.gather(lst, 1, o2).contiguous(), torch.gather(sc, 1, o2).contiguous()
                cnt2 = (lst >= 0).sum(dim=1).to(torch.int32)
                sel2, _, _ = bsp.retain_diverse(lst, sc, R, alpha, cand_count=cnt2)
                sel2 = torch.as_tensor(sel2)
                nbrs[tgt.long()] = torch.where(sel2 >= 0, torch.gather(lst, 1, sel2.clamp(min=0).long()), torch.full_like(sel2, -1))
                stats["reprunes"] += int(over.numel())
        _sync(ctx, nbrs)
        stats["backlink_s"] += time.perf_counter() - t0
        stats["batches"] += 1
        if log:
            log(f"[build] inserted {hi}/{N} (batch {B}, beam {k}): search {stats['search_s']:.1f}s prune {stats['prune_s']:.1f}s "
                f"backlink {stats['backlink_s']:.1f}s")
        lo = hi
    stats["total_s"] = time.perf_counter() - t_all
    stats["nodes_per_s"] = N / stats["total_s"]
    stats["avg_degree"] = float((nbrs >= 0).sum().item()) / N
    bsp.close()
    return nbrs, entry, stats
