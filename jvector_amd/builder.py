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

from ._lib import JV_ERR_INVALID, check
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

    def row_width(self):
        w = C.c_int()
        self._lib.jv_hip_builder_neighbors_device(self._h, C.byref(w))
        return int(w.value)

    def working_rows(self):
        """the lists as they stand (host arrays): ids [n, row_width]; in reference order also their scores and diverseBefore marks"""
        R = self.row_width()
        ids = np.empty((self.n, R), np.int32)
        sc = np.empty((self.n, R), np.float32)
        db = np.empty(self.n, np.int32)
        rc = self._lib.jv_hip_builder_working_lists(self.ctx._h, self._h, ids.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p),
                                                   db.ctypes.data_as(C.c_void_p))
        if rc == JV_ERR_INVALID:   # a default builder stores no list scores (only bl_ref_order / bl_sorted_lists do): ids alone
            check(self._lib.jv_hip_builder_working_lists(self.ctx._h, self._h, ids.ctypes.data_as(C.c_void_p), None, None))
            return ids, None, None
        check(rc)
        return ids, sc, db

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


def build_hierarchical(ctx, pq, pq_vectors, vectors, vsf, max_degree=32, beam_width=100, alpha=1.2, seed=11, log=None, min_top=8, overflow=1.25,
                       max_batch=131072, improve=0, passes=1, vector_set=None):
    """The reference's layered graph — jv_hip_build_layered: level draws (getRandomGraphLevel, seeded), one Vamana graph per level,
    `improve` passes of improveConnections per level, enforceDegree, entry point, all inside the library; this function only moves
    the result into arrays.  Returns (levels, entry_node, entry_level, level-0 neighbours on the device, stats) with
    levels[l] = (None | ascending int32 node ids, int32 neighbour rows) as host arrays.  (`passes` > 1 — whole-row re-insertion, measured
    worse than an improve pass — is only available through build_vamana.)"""
    if passes != 1:
        raise ValueError("build_hierarchical: re-insertion passes are a build_vamana experiment; use improve=")
    dev, N = vectors.device, int(vectors.shape[0])
    vs = vector_set if vector_set is not None else VectorSet(ctx, vectors)
    lib = ctx._lib
    h = C.c_void_p()
    t0 = time.perf_counter()
    check(lib.jv_hip_build_layered(ctx._h, pq._h, pq_vectors._h, vs._h, int(vsf), int(max_degree), int(beam_width), float(alpha), float(overflow),
                                   int(max_batch), int(improve), int(seed), int(min_top), C.byref(h)))
    try:
        n_lv, entry, entry_level = C.c_int(), C.c_int32(), C.c_int()
        check(lib.jv_hip_layered_info(h, C.byref(n_lv), C.byref(entry), C.byref(entry_level), None))
        counts = (C.c_int64 * n_lv.value)()
        check(lib.jv_hip_layered_info(h, None, None, None, counts))
        nb0 = torch.empty((N, max_degree), dtype=torch.int32, device=dev)
        p0, _k = _ptr(nb0, np.int32)
        check(lib.jv_hip_layered_level(ctx._h, h, 0, None, p0))
        levels = [(None, nb0.cpu().numpy())]
        for l in range(1, n_lv.value):
            ids = np.empty(int(counts[l]), np.int32)
            rows = np.empty((int(counts[l]), max_degree), np.int32)
            check(lib.jv_hip_layered_level(ctx._h, h, l, ids.ctypes.data_as(C.c_void_p), rows.ctypes.data_as(C.c_void_p)))
            levels.append((ids, rows))
        sec, cnt = (C.c_double * 4)(), (C.c_int64 * 5)()
        check(lib.jv_hip_layered_stats(h, sec, cnt))
    finally:
        lib.jv_hip_layered_destroy(h)
    total = BuildStats(search_s=sec[0], prune_s=sec[1], backlink_s=sec[2], batches=int(cnt[0]), reprunes=int(cnt[1]), inserted=int(cnt[2]),
                       visited=int(cnt[3]), expanded=int(cnt[4]), total_s=sec[3], wall_s=time.perf_counter() - t0)
    total["nodes_per_s"] = N / max(total["total_s"], 1e-9)
    total["avg_degree"] = float((nb0 >= 0).sum().item()) / N
    total["levels"] = [int(c) for c in counts]
    if log:
        log(f"[build] layered: {total['levels']} nodes per level in {total['total_s']:.1f}s (search {sec[0]:.1f}s prune {sec[1]:.1f}s backlink {sec[2]:.1f}s)")
    return levels, int(entry.value), int(entry_level.value), nb0, total
