#!/usr/bin/env python3
"""GraphSearcher OBJECTS on the device traversal (round 3): QPS of search_ex — plain, with a rerank floor, with a threshold —
next to the plain batched search, on a 1M x 768 / PQ-96 index the engine builds itself; and the cost of resume() (the session
kernel replays the searcher's earlier calls and goes on: one launch).
Prints one JSON object.   usage: python scripts/searcher_bench.py [--n 1000000] [--queries 4096]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import jvector_amd as J
from benchlib import Mixture
from jvector_amd.builder import build_hierarchical


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--queries", type=int, default=4096)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    ctx = J.HipContext(0, stream=torch.cuda.current_stream().cuda_stream)
    VSF = J.VectorSimilarityFunction.COSINE
    D, M, N, Q = 768, 96, args.n, args.queries
    mix = Mixture(D, seed=5, device=dev)
    base = mix.sample(N, seed=5)
    q = mix.sample(Q, seed=6)
    g = torch.Generator(device=dev).manual_seed(4)
    pq = J.ProductQuantization.compute(ctx, base[torch.randperm(N, generator=g, device=dev)[:128_000]].contiguous(), M, seed=4)
    vs = J.VectorSet(ctx, base)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    levels, entry, entry_level, nb0, _ = build_hierarchical(ctx, pq, cv, base, VSF, overflow=2.0)
    fused = J.FusedPQ.build(ctx, cv, nb0)
    graph = J.GraphIndex(ctx, N, levels, entry, entry_level)
    s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=Q)
    qh = q.cpu().numpy()

    def timed(fn, reps=3):
        fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        ctx.sync()
        return (time.perf_counter() - t0) / reps, r

    out = {"n": N, "queries": Q}
    t, _ = timed(lambda: s.search(q, VSF, 10, 100))
    out["plain_search_qps"] = Q / t
    # a score level ~ the 60th best approximate neighbour of a typical query: the threshold search returns "everything above it"
    import ctypes as C
    ses = s._session()

    def c_call(top_k, rerank_k, threshold=0.0, rerank_floor=0.0):
        """jv_hip_searcher_search straight through ctypes (outputs into preallocated numpy buffers): what a Java caller pays —
        search_ex additionally builds one Python SearchResult object per query"""
        ids, sc = np.empty((Q, top_k), np.int32), np.empty((Q, top_k), np.float32)
        cnt, st, w = np.empty(Q, np.int32), np.empty((Q, 4), np.int64), np.empty(Q, np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        def run():
            J._lib.check(ctx._lib.jv_hip_searcher_search(ctx._h, ses, p(qh), Q, int(VSF), top_k, rerank_k, float(threshold), float(rerank_floor), None, 0,
                                                         p(ids), p(sc), p(cnt), p(st), p(w)))
            return cnt, st
        return run

    # the approximate-score level of the ~60th neighbour (thresholds act on APPROXIMATE scores)
    thr = float(np.median([r.worst_approximate_in_topk for r in s.search_ex(qh[:256], VSF, 60, 60)]))
    if os.environ.get("SEARCHER_BENCH_TIMING"):
        ctx.set_option("graph_timing", 1)
    for name, kw in (("objects_plain", dict(top_k=10, rerank_k=100)), ("objects_floor", dict(top_k=10, rerank_k=100, rerank_floor=thr)),
                     ("objects_threshold", dict(top_k=200, rerank_k=200, threshold=thr)),
                     ("objects_top200_no_threshold", dict(top_k=200, rerank_k=200)),
                     ("objects_threshold_top10", dict(top_k=10, rerank_k=100, threshold=thr))):
        ctx.reset_stats()
        t, (cnt, st) = timed(c_call(**kw))
        out[name] = {"qps": Q / t, "device_calls": ctx.stat("gs_session_calls_device"), "host_overflow_calls": ctx.stat("gs_session_calls_host_overflow"),
                     "queries_retried": ctx.stat("gs_queries_retried"), "workers_per_cu": ctx.stat("gs_last_workers_per_cu"),
                     "avg_results": float(cnt.mean()), "avg_visited": float(st[:, 0].mean())}
    t, _ = timed(lambda: s.search_ex(qh, VSF, top_k=10, rerank_k=100), reps=2)
    out["objects_plain_through_python_search_ex_qps"] = Q / t
    ctx.reset_stats()
    t0 = time.perf_counter()
    s.resume(10, 100)
    out["resume_after_device_search_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    s.resume(10, 100)
    out["second_resume_s"] = time.perf_counter() - t0
    out["resume_device_calls"] = ctx.stat("gs_session_resume_device")
    out["resume_host_replays"] = ctx.stat("gs_session_resume_replays")
    ctx.set_option("graph_traversal", 1)   # the same searches on the host searcher
    t, _ = timed(c_call(top_k=200, rerank_k=200, threshold=thr), reps=1)
    out["objects_threshold_host"] = {"qps": Q / t}
    t, _ = timed(c_call(top_k=10, rerank_k=100), reps=1)
    out["objects_plain_host"] = {"qps": Q / t}
    ctx.set_option("graph_traversal", None)
    out["threshold"] = thr
    print(json.dumps(out))


if __name__ == "__main__":
    main()
