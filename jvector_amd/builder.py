"""Batched Vamana construction (BASELINE config 5) — a thin mirror of the C ABI's jv_hip_builder_* (csrc/builder.cpp): the
search -> robust prune -> backlink -> re-prune pipeline of a batch of inserts runs inside the library on the device; what is
left here is the insertion SCHEDULE (a seeded permutation, prefix-doubling batches) and the layering of
GraphIndexBuilder.addHierarchy (B/graph/GraphIndexBuilder.java:562-575).  The reference's builder is concurrent and
nondeterministic; the contract is the structure of the result and the recall of a search over it (tests/test_builder.py)."""
from __future__ import annotations

import ctypes as C
import time

import numpy as np
import torch

from ._lib import check
from .engine import VectorSet, _ptr


class BuildStats(dict):
    pass


class GraphBuilder:
    """jv_builder: one graph level over the nodes of `pq_vectors` / `vectors` (a VectorSet on the engine's device)."""

    def __init__(self, ctx, pq, pq_vectors, vectors: VectorSet, vsf, max_degree=32, beam_width=100, alpha=1.2, overflow=1.25):
        self.ctx, self._lib, self._keep = ctx, ctx._lib, (pq, pq_vectors, vectors)
        self.n, self.max_degree = int(pq_vectors.count()), int(max_degree)
        h = C.c_void_p()
        check(self._lib.jv_hip_builder_create(ctx._h, pq._h, pq_vectors._h, vectors._h, int(vsf), int(max_degree), int(beam_width),
                                              float(alpha), float(overflow), C.byref(h)))
        self._h = h

    def seed(self, node):
        check(self._lib.jv_hip_builder_seed(self.ctx._h, self._h, int(node)))

    def insert_batch(self, nodes):
        """nodes: int32 ordinals (torch tensor on the device, or a numpy array), none inserted before"""
        p, _k = _ptr(nodes, np.int32)
        check(self._lib.jv_hip_builder_insert_batch(self.ctx._h, self._h, p, int(nodes.shape[0])))

    def improve_batch(self, nodes):
        """improveConnections for nodes that are in the graph: search, merge with the node's neighbours, robust prune, backlink"""
        p, _k = _ptr(nodes, np.int32)
        check(self._lib.jv_hip_builder_improve_batch(self.ctx._h, self._h, p, int(nodes.shape[0])))

    def finish(self, out):
        """enforceDegree; `out` [n, max_degree] int32 (torch / numpy) receives the packed, -1 padded rows"""
        p, _k = _ptr(out, np.int32)
        check(self._lib.jv_hip_builder_finish(self.ctx._h, self._h, p))
        return out

    def stats(self):
        s, c = (C.c_double * 3)(), (C.c_int64 * 5)()
        check(self._lib.jv_hip_builder_stats(self._h, s, c))
        return BuildStats(search_s=s[0], prune_s=s[1], backlink_s=s[2], batches=int(c[0]), reprunes=int(c[1]), inserted=int(c[2]),
                          visited=int(c[3]), expanded=int(c[4]))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.jv_hip_builder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def build_vamana(ctx, pq, pq_vectors, vectors, vsf, max_degree=32, beam_width=100, alpha=1.2, max_batch=131072, seed=11, log=None,
                 overflow=1.25, vector_set=None, passes=1, improve=0):
    """One graph level.  vectors: [N, D] float32 tensor on the engine's device (the insert queries).  Prefix-doubling batches: a
    batch never exceeds what the graph already holds.  Returns (neighbors [N, max_degree] int32 tensor, entry_node, BuildStats)."""
    N = int(vectors.shape[0])
    vs = vector_set if vector_set is not None else VectorSet(ctx, vectors)
    perm = torch.randperm(N, generator=torch.Generator(device="cpu").manual_seed(seed)).to(torch.int32).to(vectors.device)
    b = GraphBuilder(ctx, pq, pq_vectors, vs, vsf, max_degree, beam_width, alpha, overflow)
    t0 = time.perf_counter()
    entry = int(perm[0])
    b.seed(entry)
    lo = 1
    while lo < N:
        hi = min(N, lo + min(max_batch, lo))
        b.insert_batch(perm[lo:hi].contiguous())
        if log:
            st = b.stats()
            log(f"[build] inserted {hi}/{N}: search {st['search_s']:.1f}s prune {st['prune_s']:.1f}s backlink {st['backlink_s']:.1f}s")
        lo = hi
    for extra in range(1, passes):   # improveConnections for every node (GraphIndexBuilder.java:510-540): re-insert against the finished graph
        for lo in range(0, N, max_batch):
            b.insert_batch(perm[lo:lo + max_batch].contiguous())
        if log:
            st = b.stats()
            log(f"[build] pass {extra + 1} done: search {st['search_s']:.1f}s prune {st['prune_s']:.1f}s backlink {st['backlink_s']:.1f}s")
    for extra in range(improve):   # cleanup()'s improveConnections, for every node: search + MERGE with the node's row + prune + backlink
        for lo in range(0, N, max_batch):
            b.improve_batch(perm[lo:lo + max_batch].contiguous())
        if log:
            st = b.stats()
            log(f"[build] improve pass {extra + 1} done: search {st['search_s']:.1f}s prune {st['prune_s']:.1f}s backlink {st['backlink_s']:.1f}s")
    out = b.finish(torch.empty((N, max_degree), dtype=torch.int32, device=vectors.device))
    stats = b.stats()
    stats["total_s"] = time.perf_counter() - t0
    stats["nodes_per_s"] = N / stats["total_s"]
    stats["avg_degree"] = float((out >= 0).sum().item()) / N
    b.close()
    return out, entry, stats


def _nearest_to_mean(ctx, vs, vectors, nodes, vsf):
    """the node of `nodes` most similar to the data mean under the index's own similarity function (the reference re-centres
    its entry point on the medoid: GraphIndexBuilder.java updateEntryPoint / approximateCentroid)"""
    mean = vectors[nodes[: min(int(nodes.shape[0]), 100000)].long()].mean(0, keepdim=True).contiguous()
    sc = vs.scores(mean, vsf, nodes.to(torch.int32).view(1, -1).contiguous())
    return int(nodes[int(torch.as_tensor(sc).reshape(-1).argmax())])


def build_hierarchical(ctx, pq, pq_vectors, vectors, vsf, max_degree=32, beam_width=100, alpha=1.2, seed=11, log=None, min_top=8, **kw):
    """The reference's layered graph (a node reaches level >= l with probability maxDegree^-l, ml = 1 / ln(degree)): nested random
    subsets of N / maxDegree^l nodes, each level a Vamana graph over its own nodes.  Returns (levels, entry_node, entry_level,
    level-0 neighbours on the device, stats) with levels[l] = (None | sorted int32 node ids, int32 neighbour rows) as host arrays."""
    from .engine import PQVectors
    dev, N = vectors.device, int(vectors.shape[0])
    perm = torch.randperm(N, generator=torch.Generator(device="cpu").manual_seed(seed + 1)).to(dev)
    vs0 = VectorSet(ctx, vectors)
    nb0, entry, stats = build_vamana(ctx, pq, pq_vectors, vectors, vsf, max_degree, beam_width, alpha, seed=seed, log=log, vector_set=vs0, **kw)
    levels, entry_level, all_stats = [(None, nb0.cpu().numpy())], 0, {"level0": dict(stats)}
    n = N // max_degree
    while n >= min_top:
        nodes = torch.sort(perm[:n]).values          # nested: perm[:n_{l+1}] is a subset of perm[:n_l]
        sub_vec = vectors[nodes].contiguous()
        sub_vs = VectorSet(ctx, sub_vec)
        sub_cv = PQVectors.encode_and_build(ctx, pq, sub_vs)      # same codes as level 0's rows
        nbl, _, st = build_vamana(ctx, pq, sub_cv, sub_vec, vsf, max_degree, beam_width, alpha, seed=seed + len(levels), log=log,
                                  vector_set=sub_vs, **kw)
        glob = torch.where(nbl >= 0, nodes[nbl.clamp(min=0).long()].to(torch.int32), nbl)
        levels.append((nodes.to(torch.int32).cpu().numpy(), glob.cpu().numpy()))
        all_stats[f"level{len(levels) - 1}"] = dict(st)
        entry_level = len(levels) - 1
        entry = _nearest_to_mean(ctx, vs0, vectors, nodes, vsf)   # top level's node closest to the data mean
        n //= max_degree
    total = BuildStats(stats)
    for k in ("search_s", "prune_s", "backlink_s", "total_s", "reprunes", "batches", "visited", "expanded", "inserted"):
        total[k] = sum(v[k] for v in all_stats.values())
    total["nodes_per_s"] = N / total["total_s"]
    total["levels"] = [int(N)] + [int(l[0].shape[0]) for l in levels[1:]]
    return levels, entry, entry_level, nb0, total
