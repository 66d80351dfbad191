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
                 search_batch=65536, log=None, out=None, overflow=1.25):
    """One graph level.  vectors: [N, D] float32 torch tensor on the engine's device (the insert queries); pq_vectors: their PQ
    codes (PQVectors).  out: optional preallocated [N, W] int32 tensor for the working adjacency, W = floor(overflow *
    max_degree): like the reference's ConcurrentNeighborMap (neighborOverflow 1.2, :298-322) a neighbour list may run over
    max_degree by that factor before a backlink forces a re-prune; a final pass prunes every list back to max_degree.
    Returns (neighbors [N, max_degree] int32 device tensor, -1 padded; entry_node; BuildStats)."""
    dev = vectors.device
    N = int(vectors.shape[0])
    Rf = int(max_degree)                                   # final degree
    R = max(Rf, min(64, int(Rf * overflow)))               # working width
    g = torch.Generator(device="cpu").manual_seed(seed)
    perm = torch.randperm(N, generator=g).to(dev)
    owned = None
    if out is not None:
        nbrs = out
        R = int(out.shape[1])
        if R < Rf or R > 64:
            raise ValueError(f"out has {R} columns; need max_degree {Rf} <= columns <= 64")
    elif vectors.is_cuda:
        nbrs = torch.empty((N, R), dtype=torch.int32, device=dev)
    else:
        # no CUDA tensors (the CPU dry run against the mock device): the adjacency must still be memory the library classifies
        # as device memory -> allocate it through the C ABI and view it as a tensor
        import ctypes as C
        from ._lib import check
        ptr = C.c_void_p()
        check(ctx._lib.jv_hip_device_alloc(ctx._h, N * R * 4, C.byref(ptr)))
        owned = ptr
        nbrs = torch.frombuffer((C.c_byte * (N * R * 4)).from_address(ptr.value), dtype=torch.int32).view(N, R)
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
        sel, nsel, _ = bsp.retain_diverse(cand, csc, Rf, alpha, cand_count=count)
        sel = torch.as_tensor(sel)
        chosen = torch.where(sel >= 0, torch.gather(cand, 1, sel.clamp(min=0).long()), torch.full_like(sel, -1))
        nbrs[batch.long(), :Rf] = chosen
        _sync(ctx, nbrs)
        stats["prune_s"] += time.perf_counter() - t0
        # ---- 3. backlinks: v joins the list of each of its chosen neighbours s; lists that overflow are re-pruned ----
        t0 = time.perf_counter()
        src = batch.view(-1, 1).expand(-1, Rf).reshape(-1)
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
                sel2, _, _ = bsp.retain_diverse(lst, sc, Rf, alpha, cand_count=cnt2)
                sel2 = torch.as_tensor(sel2)
                nbrs[tgt.long()] = -1
                nbrs[tgt.long(), :Rf] = torch.where(sel2 >= 0, torch.gather(lst, 1, sel2.clamp(min=0).long()), torch.full_like(sel2, -1))
                stats["reprunes"] += int(over.numel())
        _sync(ctx, nbrs)
        stats["backlink_s"] += time.perf_counter() - t0
        stats["batches"] += 1
        if log:
            log(f"[build] inserted {hi}/{N} (batch {B}, beam {k}): search {stats['search_s']:.1f}s prune {stats['prune_s']:.1f}s "
                f"backlink {stats['backlink_s']:.1f}s")
        lo = hi
    # ---- final pass (the reference's cleanup: enforceDegree): lists still above max_degree are pruned back ----
    t0 = time.perf_counter()
    if R > Rf:
        over_all = ((nbrs >= 0).sum(dim=1) > Rf).nonzero().squeeze(1)
        for s0 in range(0, int(over_all.numel()), max_batch):
            tgt = over_all[s0:s0 + max_batch].to(torch.int32)
            lst = nbrs[tgt.long()].contiguous()
            sc = torch.as_tensor(bsp.diversity_scores(tgt, lst))
            o2 = torch.argsort(sc, dim=1, descending=True, stable=True)
            lst, sc = torch.gather(lst, 1, o2).contiguous(), torch.gather(sc, 1, o2).contiguous()
            sel2, _, _ = bsp.retain_diverse(lst, sc, Rf, alpha, cand_count=(lst >= 0).sum(dim=1).to(torch.int32))
            sel2 = torch.as_tensor(sel2)
            nbrs[tgt.long()] = -1
            nbrs[tgt.long(), :Rf] = torch.where(sel2 >= 0, torch.gather(lst, 1, sel2.clamp(min=0).long()), torch.full_like(sel2, -1))
            stats["reprunes"] += int(tgt.numel())
        _sync(ctx, nbrs)
    stats["backlink_s"] += time.perf_counter() - t0
    final = nbrs[:, :Rf].contiguous() if R > Rf else nbrs
    stats["total_s"] = time.perf_counter() - t_all
    stats["nodes_per_s"] = N / stats["total_s"]
    stats["avg_degree"] = float((final >= 0).sum().item()) / N
    bsp.close()
    if owned is not None:  # hand back an ordinary tensor and release the library allocation
        graph.close()
        final = final.clone()
        del nbrs
        ctx._lib.jv_hip_device_free(ctx._h, owned)
    return final, entry, stats


def build_hierarchical(ctx, pq, pq_vectors, vectors, vsf, max_degree=32, beam_width=100, alpha=1.2, seed=11, log=None, min_top=8,
                       **kw):
    """The reference's layered graph (GraphIndexBuilder with addHierarchy: a node reaches level >= l with probability
    maxDegree^-l, :562-575 — ml = 1 / ln(degree)): nested random subsets of N / maxDegree^l nodes, each level a Vamana graph over
    its own nodes built by build_vamana.  Returns (levels, entry_node, entry_level, level-0 neighbours on the device, stats) with
    levels[l] = (None | sorted int32 node ids, int32 neighbour rows) as host arrays — what GraphIndex takes."""
    from ._lib import check
    from .engine import PQVectors, VectorSet
    dev = vectors.device
    N = int(vectors.shape[0])
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    perm = torch.randperm(N, generator=g).to(dev)
    nb0, entry, stats = build_vamana(ctx, pq, pq_vectors, vectors, vsf, max_degree, beam_width, alpha, seed=seed, log=log, **kw)
    levels = [(None, nb0.cpu().numpy())]
    entry_level = 0
    n = N // max_degree
    all_stats = {"level0": dict(stats)}
    codes_all = torch.as_tensor(pq_vectors.get(0, N)) if not vectors.is_cuda else None
    mean = vectors[perm[: min(N, 100000)]].mean(0)
    while n >= min_top:
        nodes = torch.sort(perm[:n]).values          # nested: perm[:n_{l+1}] is a subset of perm[:n_l]
        sub_vec = vectors[nodes].contiguous()
        if vectors.is_cuda:
            codes_t = torch.empty((n, pq.M), dtype=torch.uint8, device=dev)
            sub_cv = PQVectors(ctx, pq, codes_t)
            sub_vs = VectorSet(ctx, sub_vec)
            check(ctx._lib.jv_hip_pq_encode_into(ctx._h, pq._h, sub_vs._h, 0, n, sub_cv._h))   # same codes as level 0's rows
        else:
            sub_cv = PQVectors(ctx, pq, codes_all[nodes.cpu()].numpy())
        nbl, _, st = build_vamana(ctx, pq, sub_cv, sub_vec, vsf, max_degree, beam_width, alpha, seed=seed + len(levels), log=log, **kw)
        glob = torch.where(nbl >= 0, nodes[nbl.clamp(min=0).long()].to(torch.int32), nbl)
        levels.append((nodes.to(torch.int32).cpu().numpy(), glob.cpu().numpy()))
        all_stats[f"level{len(levels) - 1}"] = dict(st)
        entry_level = len(levels) - 1
        # entry point: the top level's node closest to the data mean (the reference re-centres on the medoid at cleanup)
        entry = int(nodes[(sub_vec @ mean).argmax()])
        n //= max_degree
    total = BuildStats(stats)
    for k in ("search_s", "prune_s", "backlink_s", "total_s", "reprunes", "batches"):
        total[k] = sum(v[k] for v in all_stats.values())
    total["nodes_per_s"] = N / total["total_s"]
    total["levels"] = [int(N)] + [int(l[0].shape[0]) for l in levels[1:]]
    return levels, entry, entry_level, nb0, total
