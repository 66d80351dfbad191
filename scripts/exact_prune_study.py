"""exact_prune_study.py — graph-quality experiment (round 6): what would an EXACT-score improve pass buy?

NOT product code: torch does the exact arithmetic here so that the question "is a full-resolution robust prune worth a kernel
family of its own?" has a measured answer (the calibrated rerankK of the headline index) before anything is built.  The engine
builds its default graph (PQ scores, as bench.py does); then level 0 is re-pruned with exact cosine scores
(BuildScoreProvider.randomAccessScoreProvider's arithmetic, B/graph/similarity/BuildScoreProvider.java:106-168;
VamanaDiversityProvider.retainDiverse :45-96) in two stages:
  A  candidates(u) = the engine's search from u's own vector (topK = beam, exact rerank) + u's current row  -> fwd[u] (<= degree)
  B  candidates(v) = fwd[v] + {u : v in fwd[u]} (reverse edges, capped)                                       -> row[v] (<= degree)
and every variant is calibrated by bench.py's own rule (smallest rerankK whose calibration recall clears 0.95 by two standard
errors), then evaluated on the disjoint evaluation set and timed for a few steps.

    python scripts/exact_prune_study.py --n 10000000 [--variants base,AB,B]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import calibrate, evaluate, log, timed_steps  # noqa: E402
from benchgraph import _robust_prune  # noqa: E402
from benchlib import Mixture, ground_truth  # noqa: E402


def exact_prune(base, node_ids, cand, degree, alpha, chunk_rows):
    """node_ids [B], cand [B, K] int64 (-1 padded; duplicates and self allowed) -> [B, degree] int32 rows, -1 padded"""
    out = torch.full((node_ids.shape[0], degree), -1, dtype=torch.int32, device=base.device)
    for s in range(0, node_ids.shape[0], chunk_rows):
        ids, c = node_ids[s:s + chunk_rows], cand[s:s + chunk_rows]
        B, K = c.shape
        # duplicates / self / padding -> -inf
        cs, o = torch.sort(c, dim=1)
        dup = torch.zeros_like(cs, dtype=torch.bool)
        dup[:, 1:] = cs[:, 1:] == cs[:, :-1]
        bad = dup | (cs < 0) | (cs == ids[:, None])
        x = base[ids]
        cv = base[cs.clamp(min=0).reshape(-1)].reshape(B, K, -1)
        sc = torch.einsum("bd,bkd->bk", x, cv)
        sc = torch.where(bad, torch.full_like(sc, float("-inf")), sc)
        o2 = torch.argsort(sc, dim=1, descending=True, stable=True)
        sc, cs = sc.gather(1, o2), cs.gather(1, o2)
        cv = cv.gather(1, o2[:, :, None].expand(-1, -1, cv.shape[2]))
        pair = torch.bmm(cv, cv.transpose(1, 2))
        del cv
        kept = _robust_prune((1 + sc) / 2, (1 + pair) / 2, degree, alpha_max=alpha)
        del pair
        rank = torch.cumsum(kept.long(), 1) - 1
        rows = torch.arange(B, device=base.device)[:, None].expand_as(cs)
        o_rows = out[s:s + chunk_rows]
        o_rows[rows[kept], rank[kept]] = cs[kept].int()
    return out


def reverse_lists(fwd, cap):
    """fwd [N, deg] int32 -> rev [N, cap] int64: the sources of the edges that point at a node (first `cap` in source order)"""
    N, deg = fwd.shape
    dev = fwd.device
    rev = torch.full((N, cap), -1, dtype=torch.int64, device=dev)
    step = 2_000_000
    fill = torch.zeros(N, dtype=torch.int64, device=dev)
    for s in range(0, N, step):   # source chunks in ascending order: positions = running fill + rank within the chunk
        f = fwd[s:s + step].long()
        mask = f >= 0
        src = (torch.arange(s, s + f.shape[0], device=dev)[:, None].expand_as(f))[mask]
        dst = f[mask]
        o = torch.argsort(dst, stable=True)
        src, dst = src[o], dst[o]
        first = torch.searchsorted(dst, dst)            # index of the first edge with the same target
        pos = fill[dst] + (torch.arange(dst.shape[0], device=dev) - first)
        ok = pos < cap
        rev[dst[ok], pos[ok]] = src[ok]
        fill += torch.bincount(dst, minlength=N)
    return rev


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--m", type=int, default=96)
    ap.add_argument("--degree", type=int, default=32)
    ap.add_argument("--latent", type=int, default=32)
    ap.add_argument("--beam", type=int, default=100, help="stage A: topK of the search from the node's own vector")
    ap.add_argument("--alpha", type=float, default=1.2)
    ap.add_argument("--rev-cap", type=int, default=96)
    ap.add_argument("--variants", default="base,B,AB")
    ap.add_argument("--queries", type=int, default=131072)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--cal-queries", type=int, default=16384)
    ap.add_argument("--eval-queries", type=int, default=10240)
    ap.add_argument("--chunk-rows", type=int, default=16384)
    ap.add_argument("--out", default="gpurun_out/exact_prune_study.json")
    args = ap.parse_args()

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import jvector_amd as J
    from jvector_amd.builder import build_hierarchical
    ctx = J.HipContext(0, stream=torch.cuda.current_stream().cuda_stream)
    VSF = J.VectorSimilarityFunction.COSINE
    N, D, M, K, Q = args.n, args.dim, args.m, 10, args.queries
    mix = Mixture(D, seed=5, device=dev, latent=args.latent)
    base = mix.sample(N, seed=5)
    queries = mix.sample(Q * (args.steps + 1), seed=6)
    cal_q, eval_q = mix.sample(args.cal_queries, seed=7), mix.sample(args.eval_queries, seed=8)
    g = torch.Generator(device=dev).manual_seed(4)
    sample = base[torch.randperm(N, generator=g, device=dev)[:min(128_000, N)]].contiguous()
    pq = J.ProductQuantization.compute(ctx, sample, M, seed=4)
    del sample
    vs = J.VectorSet(ctx, base)
    codes_t = torch.empty(N, M, dtype=torch.uint8, device=dev)
    cv = J.PQVectors(ctx, pq, codes_t)
    J._lib.check(ctx._lib.jv_hip_pq_encode_into(ctx._h, pq._h, vs._h, 0, N, cv._h))
    levels, entry, entry_level, nb0, bstats = build_hierarchical(ctx, pq, cv, base, VSF, max_degree=args.degree, beam_width=100, alpha=1.2,
                                                                 log=log, overflow=2.0, improve=1)
    log(f"[build] {dict(bstats)}")
    cal_gt = ground_truth(J, ctx, vs, cal_q, VSF, K, dense=True).cpu().numpy()
    eval_gt = ground_truth(J, ctx, vs, eval_q, VSF, K, dense=True).cpu().numpy()
    ladder = [30, 40, 44, 48, 52, 56, 60, 63, 66, 70, 72, 74, 76, 78, 80, 84, 88, 92, 100, 110, 125, 150, 200]

    def searcher_for(nb):
        lv = [(None, nb.cpu().numpy())] + list(levels[1:])
        fused = J.FusedPQ.build(ctx, cv, nb)
        graph = J.GraphIndex(ctx, N, lv, entry, entry_level).set_traversal("device")
        return J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=Q), (fused, graph)

    results = {}

    def measure(tag, nb, extra=None):
        s, keep = searcher_for(nb)

        def run(qs, rk, stats=False):
            return s.search(qs, VSF, K, rk, return_stats=stats)
        rk, cal_rec = calibrate(run, ctx, ladder, cal_q, cal_gt, Q, tag)
        rec, se = evaluate(run, ctx, eval_q, eval_gt, Q, rk)
        st = run(queries[:Q], rk, stats=True)[2]
        run(queries[:Q], rk)
        el = timed_steps(run, queries[Q:], Q, args.steps, rk, torch.cuda.synchronize)
        r = {"rerankK": rk, "cal_recall": cal_rec, "eval_recall": rec, "eval_se": se, "visited": float(st[:, 0].mean()),
             "expanded": float(st[:, 1].mean()), "qps": Q * args.steps / el, "avg_degree": float((nb >= 0).sum().item()) / N}
        if extra:
            r.update(extra)
        results[tag] = r
        log(f"[study] {tag}: {json.dumps(r)}")
        return s, keep

    variants = args.variants.split(",")
    s0, keep0 = measure("base", nb0, {"build_s": bstats["total_s"]}) if "base" in variants else searcher_for(nb0)
    all_ids = torch.arange(N, device=dev)

    def stage_b(fwd, tag, t_a):
        t0 = time.perf_counter()
        rev = reverse_lists(fwd, args.rev_cap)
        out = torch.empty_like(fwd)
        step = 1_000_000
        for s in range(0, N, step):
            cand = torch.cat([fwd[s:s + step].long(), rev[s:s + step]], 1)
            out[s:s + step] = exact_prune(base, all_ids[s:s + step], cand, args.degree, args.alpha, args.chunk_rows)
        torch.cuda.synchronize()
        return out, {"stage_a_s": t_a, "stage_b_s": time.perf_counter() - t0}

    if "B" in variants:     # exact re-prune of (row + reverse edges) only: no searches
        nbB, ex = stage_b(nb0, "B", 0.0)
        measure("B", nbB, ex)
        del nbB
    if "AB" in variants:
        t0 = time.perf_counter()
        fwd = torch.empty_like(nb0)
        for s in range(0, N, Q):
            ids = all_ids[s:s + Q]
            found = s0.search(base[s:s + Q], VSF, args.beam, args.beam)[0]
            cand = torch.cat([found.long(), nb0[s:s + Q].long()], 1)
            fwd[s:s + Q] = exact_prune(base, ids, cand, args.degree, args.alpha, args.chunk_rows)
        torch.cuda.synchronize()
        t_a = time.perf_counter() - t0
        log(f"[study] stage A {t_a:.1f} s, avg fwd degree {float((fwd >= 0).sum().item()) / N:.2f}")
        if "A" in variants:
            measure("A", fwd, {"stage_a_s": t_a})
        nbAB, ex = stage_b(fwd, "AB", t_a)
        measure("AB", nbAB, ex)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        json.dump({"args": vars(args), "results": results}, f, indent=1)
    print(json.dumps(results))


if __name__ == "__main__":
    main()
