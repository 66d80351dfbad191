#!/usr/bin/env python3
"""Per-kernel micro-benchmarks at the BASELINE shapes (C3: D=768, PQ-96, maxDegree 32, cosine), timed with the
engine's HIP-event regions.  Prints one JSON object; algorithmic bytes per unit follow SURVEY.md §8d.
usage: python scripts/microbench.py [--out profiles/r1_microbench.json]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import jvector_amd as J

HBM = 8000.0


def timed(ctx, region, fn, iters=5):
    fn()  # warm
    ctx.sync()
    ctx.profile(True)
    for _ in range(iters):
        fn()
    ms, n = ctx.profile_read(region)
    ctx.profile(False)
    return ms / max(n, 1) * (n / iters)  # ms per call (a call may hold several regions)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--nodes", type=int, default=1_000_000)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    ctx = J.HipContext(0)
    g = torch.Generator(device=dev).manual_seed(1)
    D, M, DEG, N = 768, 96, 32, args.nodes
    VSF = J.VectorSimilarityFunction.COSINE
    base = torch.randn(N, D, generator=g, device=dev)
    base /= base.norm(dim=1, keepdim=True)
    cb = base[torch.randperm(N, generator=g, device=dev)[:256]].reshape(256, M, 8).permute(1, 0, 2).contiguous().reshape(-1)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb.cpu().numpy())
    vs = J.VectorSet(ctx, base)
    res = {}

    # --- encode (row 3)
    codes_t = torch.empty(N, M, dtype=torch.uint8, device=dev)
    ms = timed(ctx, "encode", lambda: pq.encode_all(base, out=codes_t), 3)
    res["pq_encode"] = {"ms": ms, "vectors_per_s": N / ms * 1e3, "tflops_nonfused": 3 * 256 * D * N / ms / 1e9,
                        "alg_GBps": N * (4 * D + M) / ms / 1e6}
    cv = J.PQVectors(ctx, pq, codes_t)

    # --- LUT build (row 2)
    Q = 4096
    queries = torch.randn(Q, D, generator=g, device=dev)
    luts = J.QueryTables(ctx, pq, Q)
    ms = timed(ctx, "lut", lambda: luts.build(queries, VSF, J.DecoderKind.FUSED))
    res["lut_build"] = {"ms": ms, "queries_per_s": Q / ms * 1e3, "alg_GBps": Q * 256 * D * 4 / ms / 1e6}

    # --- ADC gather: graph-frontier shape, Q queries x 32 random ordinals (rows 5/6)
    sf = J.ApproximateScoreFunction(cv, luts.build(queries, VSF, J.DecoderKind.PQ))
    ords = torch.randint(0, N, (Q, DEG), generator=g, device=dev, dtype=torch.int32)
    ms = timed(ctx, "adc", lambda: sf.similarity_to(ords))
    res["adc_gather_frontier"] = {"ms": ms, "shape": f"{Q}x{DEG}", "lookups_per_s": Q * DEG / ms * 1e3,
                                  "alg_GBps": Q * DEG * (M + 8) / ms / 1e6, "frac_hbm": Q * DEG * (M + 8) / ms / 1e6 / HBM}

    # --- fused blocks (row 7): Q origins, one 3072-byte block each
    nb = torch.randint(0, N, (N, DEG), generator=g, device=dev, dtype=torch.int32)
    blocks = codes_t[nb.long().reshape(-1)].reshape(N, DEG * M).contiguous()
    fused = J.FusedPQ(ctx, pq, blocks, nb)
    fsf = fused.approximate_score_function_for(queries, VSF, luts)
    origins = torch.randint(0, N, (Q,), generator=g, device=dev, dtype=torch.int32)
    ms = timed(ctx, "adc", lambda: fsf.similarity_to_neighbors(origins))
    res["fused_scores"] = {"ms": ms, "shape": f"{Q} origins x {DEG}", "nodes_per_s": Q / ms * 1e3,
                           "lookups_per_s": Q * DEG / ms * 1e3,
                           "alg_GBps": Q * (DEG * M + 4 * DEG) / ms / 1e6, "frac_hbm": Q * (DEG * M + 4 * DEG) / ms / 1e6 / HBM}

    # --- ADC flat scan, store form (Q=64 x N)
    q64 = queries[:64].contiguous()
    sf64 = J.ApproximateScoreFunction(cv, J.QueryTables(ctx, pq, 64).build(q64, VSF))
    out = torch.empty(64, N, dtype=torch.float32, device=dev)
    ms = timed(ctx, "adc", lambda: sf64.similarity_to_range(0, N, out=out))
    res["adc_scan_store_mq"] = {"ms": ms, "shape": f"64x{N}", "lookups_per_s": 64 * N / ms * 1e3,
                                "alg_GBps": 64 * N * (M + 4) / ms / 1e6, "frac_hbm": 64 * N * (M + 4) / ms / 1e6 / HBM}
    q1 = queries[:1].contiguous()
    sf1 = J.ApproximateScoreFunction(cv, J.QueryTables(ctx, pq, 1).build(q1, VSF))
    out1 = torch.empty(1, N, dtype=torch.float32, device=dev)
    ms = timed(ctx, "adc", lambda: sf1.similarity_to_range(0, N, out=out1))
    res["adc_scan_single_query"] = {"ms": ms, "shape": f"1x{N}", "lookups_per_s": N / ms * 1e3,
                                    "alg_GBps": N * (M + 4) / ms / 1e6, "frac_hbm": N * (M + 4) / ms / 1e6 / HBM}

    # --- top-k (row 9)
    ms = timed(ctx, "topk", lambda: J.topk(ctx, out, 400))
    res["topk_400_of_N"] = {"ms": ms, "shape": f"64x{N}", "alg_GBps": 64 * N * 4 / ms / 1e6}

    # --- exact rerank gather (row 1)
    q256 = queries[:256].contiguous()
    ro = torch.randint(0, N, (256, 400), generator=g, device=dev, dtype=torch.int32)
    ms = timed(ctx, "exact", lambda: vs.scores(q256, VSF, ro))
    res["exact_gather_rerank"] = {"ms": ms, "shape": "256x400", "dist_per_s": 256 * 400 / ms * 1e3,
                                  "alg_GBps": 256 * 400 * (4 * D + 4) / ms / 1e6, "frac_hbm": 256 * 400 * (4 * D + 4) / ms / 1e6 / HBM}
    # --- exact scan (row 1, brute force)
    q16 = queries[:16].contiguous()
    o16 = torch.empty(16, N, dtype=torch.float32, device=dev)
    ms = timed(ctx, "exact", lambda: vs.scan(q16, VSF, out=o16), 3)
    res["exact_scan"] = {"ms": ms, "shape": f"16x{N}", "dist_per_s": 16 * N / ms * 1e3,
                         "hbm_GBps_rows_once": N * 4 * D / ms / 1e6, "frac_hbm_rows_once": N * 4 * D / ms / 1e6 / HBM,
                         "tflops_nonfused": 16 * N * 6 * D / ms / 1e9}
    # --- MFMA tile form of the scan (k_exact_dense.hip): f32 MFMA peak 157 TF (155 TF micro-benchmark ceiling)
    if True:
        for nq in (16, 64, 256, 1024):
            qd = queries[:nq].contiguous()
            od = torch.empty(nq, N, dtype=torch.float32, device=dev)
            ms = timed(ctx, "exact", lambda: vs.scan(qd, VSF, out=od, dense=True), 3)
            res[f"exact_scan_dense_q{nq}"] = {"ms": ms, "shape": f"{nq}x{N}", "dist_per_s": nq * N / ms * 1e3,
                                              "tflops_fma": nq * N * 2 * D / ms / 1e9, "frac_mfma_f32": nq * N * 2 * D / ms / 1e9 / 157.3,
                                              "hbm_GBps_rows_once": N * 4 * D / ms / 1e6}
    # --- measured HBM ceilings on this box (record them next to the 8 TB/s vendor peak): device-to-device
    #     copy and a stream triad over 2 GiB operands, torch ops timed with events on the engine's stream
    n_el = 1 << 29
    a = torch.empty(n_el, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a).normal_()
    c = torch.empty_like(a)

    def ev_time(fn, iters=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    ms = ev_time(lambda: c.copy_(a))
    res["hbm_copy_d2d"] = {"ms": ms, "bytes": 2 * 4 * n_el, "GBps": 2 * 4 * n_el / ms / 1e6, "frac_of_8TBps": 2 * 4 * n_el / ms / 1e6 / HBM}
    ms = ev_time(lambda: torch.add(a, b, alpha=3.0, out=c))
    res["hbm_stream_triad"] = {"ms": ms, "bytes": 3 * 4 * n_el, "GBps": 3 * 4 * n_el / ms / 1e6,
                               "frac_of_8TBps": 3 * 4 * n_el / ms / 1e6 / HBM}
    try:
        props = torch.cuda.get_device_properties(0)
        res["device"] = {"name": props.name, "gcn_arch": getattr(props, "gcnArchName", None), "cus": props.multi_processor_count,
                         "hbm_GB": props.total_memory / 2 ** 30}
    except Exception:
        pass
    print(json.dumps(res, indent=1))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
