#!/usr/bin/env python3
"""bench.py — the hot path's headline benchmark on MI355X (contract: see the repo brief).

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): synthetic 10M x 768 cosine
("ada-002-like"), PQ-96 (256 clusters, 8-dim sub-vectors, trained by the engine's own ProductQuantization.compute),
FusedADC graph blocks (maxDegree 32).  One *step* = one batch of Q queries through the hot path, inputs resident in HBM:

  --mode graph (default)  GraphSearcher.search batched: centre the queries -> graph traversal with FusedPQ neighbour-block
                          scoring (device-resident traversal, one wavefront per query; --traversal host = the round-1
                          host batched searcher) -> exact rerank of the kept rerankK (NodeQueue.rerank) -> top-10
  --mode flat             LUT build -> multi-query ADC scan of all N codes (threshold-filtered) -> top-rerankK
                          -> exact rerank -> top-10   (no graph; the brute-force-over-codes path = the per-shard kernel of C4)
  --workload c2           BASELINE.json configs[1]: SIFT-like 1M x 128, L2, PQ-16, flat two-pass ADC search

value = whole-job queries/s at recall@10 >= 0.95.  rerankK is calibrated on one query set (seed 7) and recall is then
REPORTED on a disjoint evaluation set (seed 8, >= 10 000 queries, with its standard error) against exact brute-force
ground truth (MFMA dense scan for candidates, rescored by the bit-exact kernel), all outside the timed region.
`roofline` prices the dominant kernel in algorithmic bytes against the 8 TB/s HBM peak from HIP events
recorded on the engine's stream inside the timed region; `traffic` is filled only from a rocprofv3 PMC summary of THIS
configuration (profiles/traffic_r2.json, written by scripts/summarize_profile.py), else null.  `cpu_baseline` times the CPU
oracle ("port") on the host cores on a bounded sample of the same workload: scalar reference arithmetic (the parity
checker: its top-k must equal the GPU's bit for bit) and the AVX2 / AVX-512 restatement of the reference's native kernels.

N > 1: one process per GPU.  The driver launches the ranks through torch.distributed.run; a bare `python bench.py --gpus N`
re-executes itself under the same launcher (self_spawn_if_needed), and a launcher whose WORLD_SIZE disagrees with --gpus is
refused — the flag can never yield an N = 1 line.  Default workload: every rank holds a full replica of the index and serves
its own query batches (10M x 768 fits one GPU); no data-path collective; scaling = weak.  `--workload c4 --gpus 8` is the
sharded 100M configuration (every rank one 12.5M shard, RCCL all-gather of the partial top-k inside
jv_hip_sharded_search_flat).  Every N > 1 line carries `rccl_ranks` (ncclCommCount of the ENGINE's own RCCL communicator,
created for the run) and `per_rank_qps` (all-gathered through that communicator).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

from benchgraph import build_hier_graph  # noqa: E402
from benchlib import Mixture, ground_truth, recall_per_query, sift_like  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak (MI355X_MICROARCH.md: 8 TB/s spec; measured on this pool: 5.15 TB/s d2d copy,
                             # 6.0 TB/s stream triad — profiles/r2_validation/microbench.json)
LDS_B128_PEAK_GBS = 256 * 256 * 2.4  # ds_read_b128: 256 B/clk/CU x 256 CUs x 2.4 GHz = 157 TB/s (MI355X_MICROARCH.md §LDS)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def effective_cpus():
    """Host cores this process may actually use: min(affinity, cgroup CPU quota), capped at 64."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p)))
        except Exception:
            pass
    return max(1, min(n, 64))


# ------------------------------------------------------------------------------------------------------------------
# cpu_baseline legs (the oracle is the checker and the reported baseline here — never the thing measured as `value`)
# ------------------------------------------------------------------------------------------------------------------
def _simd_leg(run, gpu_ids, top_k):
    """Time `run()` once more with the oracle's SIMD kernels switched on (oracle/jv_oracle_simd.c: AVX2 / AVX-512
    restatement of the reference's native library).  Returns (isa, seconds, mean top-k overlap with the GPU's ids) or
    None when the host CPU has no AVX2."""
    from oracle import oracle as O
    try:
        isa = O.set_simd(True)
        if isa == "scalar":
            return None
        ids, cpu_s = run()
    finally:
        O.set_simd(False)
    overlap = float(np.mean([len(set(a.tolist()) & set(b.tolist())) / float(top_k) for a, b in zip(ids, gpu_ids[:len(ids)])]))
    return isa, cpu_s, overlap


def _baseline_line(nq, scalar_s, simd, threads, sample, matches):
    """cpu_baseline object: `value` is the faster, SIMD leg when the CPU has one (the fairer baseline: it is what the
    reference's native provider would run); the scalar leg is the parity checker and is reported next to it."""
    line = {"value": nq / scalar_s, "unit": "queries/s", "cores": threads, "kind": "port", "isa": "scalar",
            "scalar_value": nq / scalar_s, "sample": sample + f"; scalar oracle {scalar_s:.1f}s wall",
            "matches_gpu_topk": matches}
    if simd is not None:
        isa, simd_s, overlap = simd
        line.update({"value": nq / simd_s, "isa": isa, "simd_topk_overlap_with_gpu": overlap,
                     "sample": line["sample"] + f", {isa} restatement of the reference's native kernels {simd_s:.1f}s wall"})
    return line


def cpu_baseline_flat(cb, D, M, codes_h, base_dev, queries_dev, vsf, top_k, rerank_k, gpu_ids, max_q=None):
    """CPU oracle on a bounded sample of the flat workload: one query per host thread (more when the index is small)."""
    from oracle import oracle as O
    threads = effective_cpus()
    nq = min(max_q or threads, queries_dev.shape[0], gpu_ids.shape[0])
    opq = O.OraclePQ(D, M, cb)
    opq.cache_self_magnitudes()
    q = queries_dev[:nq].cpu().numpy()

    def run():
        t0 = time.perf_counter()
        cand, _ = opq.search_flat(codes_h, None, q, int(vsf), rerank_k, 0, nthreads=threads)
        t1 = time.perf_counter()
        cand_t = torch.from_numpy(cand.astype(np.int64)).to(base_dev.device)
        cand_vecs = base_dev[cand_t.reshape(-1)].reshape(nq, rerank_k, D).cpu().numpy()  # the rows the CPU would fetch
        t2 = time.perf_counter()
        ids, _ = O.rerank(q, cand_vecs, cand, int(vsf), top_k, nthreads=threads)
        t3 = time.perf_counter()
        return ids, (t1 - t0) + (t3 - t2)

    ids, scalar_s = run()
    simd = _simd_leg(run, gpu_ids, top_k)
    return _baseline_line(nq, scalar_s, simd, threads,
                          f"{nq} queries x {codes_h.shape[0]} codes, two-pass flat search (ADC scan + top-{rerank_k} + exact "
                          f"rerank), queries spread over {threads} threads", bool(np.array_equal(ids, gpu_ids[:nq])))


def cpu_baseline_graph(cb, D, M, codes_h, levels, entry, entry_level, base_dev, queries_dev, vsf, top_k, rerank_k,
                       gpu_ids):
    """CPU oracle on a bounded sample of the graph workload: the sequential GraphSearcher restatement (ADC scores
    via the oracle's FusedPQDecoder arithmetic), queries spread over the host cores (ctypes releases the GIL), then
    the exact rerank of the returned candidates."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle as O
    threads = effective_cpus()
    # ~10-30 s of CPU work per leg at the headline shape (~1.6 CPU-ms per query, scalar); gpu_ids covers the first batch only
    nq = min(1024 * threads, queries_dev.shape[0], gpu_ids.shape[0])
    opq = O.OraclePQ(D, M, cb)
    opq.cache_self_magnitudes()
    og = O.OracleGraph(codes_h.shape[0], levels, entry, entry_level)
    q = queries_dev[:nq].cpu().numpy()

    def one(lo):
        return og.search(opq, codes_h, None, q[lo:lo + 16], int(vsf), rerank_k, rerank_k, fused=True)

    def run():
        t0 = time.perf_counter()
        with ThreadPoolExecutor(threads) as ex:
            parts = list(ex.map(one, range(0, nq, 16)))
        t1 = time.perf_counter()
        cand = np.concatenate([p[0] for p in parts])
        # exact rerank in pieces of <= 2048 queries: the rows the CPU would fetch are gathered on the GPU and copied over
        # (untimed), at most 2048 x rerankK x D floats (0.9 GB at the headline shape) of host memory at a time
        ids, rr_s = [], 0.0
        for lo in range(0, nq, 2048):
            c = cand[lo:lo + 2048]
            cand_t = torch.from_numpy(c.astype(np.int64).clip(min=0)).to(base_dev.device)
            cand_vecs = base_dev[cand_t.reshape(-1)].reshape(c.shape[0], rerank_k, D).cpu().numpy()
            t2 = time.perf_counter()
            ids.append(O.rerank(q[lo:lo + 2048], cand_vecs, c, int(vsf), top_k, nthreads=threads)[0])
            rr_s += time.perf_counter() - t2
            del cand_vecs, cand_t
        return np.concatenate(ids), (t1 - t0) + rr_s

    ids, scalar_s = run()
    simd = _simd_leg(run, gpu_ids, top_k)
    return _baseline_line(nq, scalar_s, simd, threads,
                          f"{nq} queries, sequential GraphSearcher restatement over the same graph (fused ADC, rerankK "
                          f"{rerank_k}) + exact rerank, 16 queries per task on {threads} threads",
                          bool(np.array_equal(ids, gpu_ids[:nq])))


def cpu_baseline_build(cb, D, M, codes_h, nbrs_h, entry, base_dev, vsf, max_degree, beam, alpha, per_thread=4096):
    """CPU oracle on a bounded sample of the BUILD workload: what one insert costs the reference's per-node path — a sequential
    GraphSearcher over the (finished) graph for the node's vector, topK = rerankK = beamWidth, PQ scores — and
    VamanaDiversityProvider.retainDiverse of its candidates with the PQ diversity function (oracle restatements, the parity
    checkers of the two kernels), spread over the host cores.  Backlinks are not charged to the CPU side."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle as O
    threads = effective_cpus()
    N = codes_h.shape[0]
    nq = min(per_thread * threads, N)
    opq = O.OraclePQ(D, M, cb)
    opq.cache_self_magnitudes()
    tri = opq.codebook_partial_sums(int(vsf))
    og = O.OracleGraph(N, [(None, nbrs_h)], entry, 0)
    nodes = np.random.default_rng(1).choice(N, nq, replace=False)
    q = base_dev[torch.from_numpy(nodes).to(base_dev.device)].cpu().numpy()

    def one(lo):
        ids, sc, _ = og.search(opq, codes_h, None, q[lo:lo + 4], int(vsf), beam, beam, fused=False)
        kept = 0
        for r in range(ids.shape[0]):
            ok = ids[r] >= 0
            sel = opq.retain_diverse(tri, int(vsf), codes_h, ids[r][ok], sc[r][ok], max_degree, 0, alpha)
            kept += int(sel[1])
        return kept

    def run():
        t0 = time.perf_counter()
        with ThreadPoolExecutor(threads) as ex:
            kept = sum(ex.map(one, range(0, nq, 4)))
        return kept, time.perf_counter() - t0

    kept, scalar_s = run()
    line = {"value": nq / scalar_s, "unit": "nodes/s", "cores": threads, "kind": "port", "isa": "scalar", "scalar_value": nq / scalar_s,
            "sample": f"{nq} inserts replayed against the finished graph: GraphSearcher (beam {beam}, PQ scores) + retainDiverse (alpha {alpha}, "
                      f"maxDegree {max_degree}), 4 nodes per task on {threads} threads; scalar oracle {scalar_s:.1f}s wall; backlinks not charged",
            "avg_selected": kept / float(nq)}
    try:
        isa = O.set_simd(True)
        if isa != "scalar":
            _, simd_s = run()
            line.update({"value": nq / simd_s, "isa": isa, "sample": line["sample"] + f", {isa} restatement of the reference's native kernels {simd_s:.1f}s wall"})
    finally:
        O.set_simd(False)
    return line


# ------------------------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------------------------
def measured_traffic(kernel_key, cfg, want_entry=False):
    """HBM bytes per launch from the rocprofv3 PMC summary (profiles/traffic_r4.json, else traffic_r3.json / traffic_r2.json) — only if it was collected on THIS
    configuration (same kernel, N, D, M, queries per step, rerankK); else None."""
    for name in ("traffic_r6.json", "traffic_r5.json", "traffic_r4.json", "traffic_r3.json", "traffic_r2.json"):   # the newest summary whose configuration matches
        try:
            table = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
        for e in table.get("entries", []):
            ec = e.get("config", {})
            # (rerankK may differ by a rung of the calibration ladder between the profiled run and this one: within 10 % the counters
            #  still describe this kernel on this index; the line names the profiled rerankK next to the number)
            # (the flat filter's bound scan reads the same codes and tables whatever rerankK is: its entries match on the shape alone)
            if e.get("kernel_key") == kernel_key and all(ec.get(k) == v for k, v in cfg.items() if k != "rerankK") and \
                    (kernel_key == "adc_bq" or (ec.get("rerankK") and abs(ec["rerankK"] - cfg.get("rerankK", 0)) <= 0.1 * ec["rerankK"])):
                if want_entry:
                    return e
                return e.get("hbm_bytes_per_launch")
    return None


def search_all(run, queries, Q, rk):
    """ids of `queries` searched in batches of Q (untimed helper for calibration / evaluation)."""
    out = []
    for s in range(0, queries.shape[0], Q):
        out.append(run(queries[s:s + Q].contiguous(), rk)[0].clone())
    return torch.cat(out)


def calibrate(run, ctx, ladder, cal_q, cal_gt, Q, tag):
    """smallest rerankK of the ladder whose calibration recall clears 0.95 by two of its own standard errors — the margin
    that makes the claim hold on the DISJOINT evaluation set too, not just on the queries it was tuned on"""
    rerank_k, rec = ladder[-1], 0.0
    for rk in ladder:
        found = search_all(run, cal_q, Q, rk)
        ctx.sync()
        r = recall_per_query(found.cpu().numpy(), cal_gt)
        rec, se = float(r.mean()), float(r.std(ddof=1) / math.sqrt(len(r))) if len(r) > 1 else 0.0
        rerank_k = rk
        log(f"[calibrate] {tag} rerankK={rk}: recall@{cal_gt.shape[1]} = {rec:.4f} +- {se:.4f} on {cal_q.shape[0]} calibration queries")
        if rec - 2.0 * se >= 0.95:
            break
    return rerank_k, rec


def evaluate(run, ctx, eval_q, eval_gt, Q, rk):
    found = search_all(run, eval_q, Q, rk)
    ctx.sync()
    r = recall_per_query(found.cpu().numpy(), eval_gt)
    return float(r.mean()), float(r.std(ddof=1) / math.sqrt(len(r))) if len(r) > 1 else 0.0


def timed_steps(run, queries, Q, steps, rk, barrier):
    barrier()
    t0 = time.perf_counter()
    for s in range(steps):
        run(queries[s * Q:(s + 1) * Q], rk)
    barrier()
    return time.perf_counter() - t0


def flat_roofline(QF, N, M, f_ms, f_n, cfg, vsf_name="COSINE", bq=False, exact_ms=None):
    """The flat scan's threshold filter.  Round 5 (bq): adc_bq_kernel — one (query, candidate) pair = M look-ups of ONE byte, sixteen
    queries served by one ds_read_b128 from 7-bit bound tables in LDS — followed by the exact ADC gather of the survivors; `f_ms` is the
    `adc` region = bound tables + bound scan (round 6: the survivors' exact stage is a region of its own, `adc_exact`, reported as
    `exact_stage_ms_per_launch`).  Before (adc_mq_kernel): M look-ups of 4 B, four queries per ds_read_b128.  Bound: the LDS gather rate
    either way."""
    f_avg = f_ms / 1e3 / max(f_n, 1)
    bytes_per_lookup = 1.0 if bq else 4.0
    lds_bytes = float(QF) * N * M * bytes_per_lookup   # bytes the ds_read_b128 stream delivers per launch
    compulsory = float(N) * (M + 4)                # codes (+ code norms) read once
    ach = lds_bytes / f_avg / 1e9 if f_avg > 0 else 0.0
    traffic = measured_traffic("adc_bq" if bq else "adc_mq", cfg)
    kernel = (f"adc_bq_kernel<{vsf_name},M={M},R=8> + adc_kernel gather (two-stage threshold filter: 7-bit bound tables of 16 queries per "
              "LDS word drop what cannot reach the threshold, exact ADC sums for the survivors only)") if bq else \
             (f"adc_mq_kernel<{vsf_name},M={M},R=8,FILTER> (threshold-filtered multi-query ADC scan of all N "
              "codes; tables of 4 queries interleaved in LDS, one ds_read_b128 = 4 look-ups)")
    return {"bound": "lds", "kernel": kernel,
            "achieved": ach, "peak": LDS_B128_PEAK_GBS, "unit": "GB/s", "frac": ach / LDS_B128_PEAK_GBS,
            "traffic": traffic, "hbm_compulsory_bytes": compulsory,
            "hbm_traffic_over_compulsory": (traffic / compulsory) if traffic else None,
            "pairs_per_launch": float(QF) * N, "pairs_per_s": float(QF) * N / f_avg if f_avg > 0 else 0.0, "lds_bytes_per_lookup": bytes_per_lookup,
            "avg_launch_ms": f_avg * 1e3, "launches": f_n,
            "bound_scan_ms_per_launch": f_avg * 1e3 if bq else None,
            "exact_stage_ms_per_launch": (exact_ms[0] / max(exact_ms[1], 1)) if (bq and exact_ms) else None,
            "lds_bank_conflict_fraction": (measured_traffic("adc_bq" if bq else "adc_mq", cfg, True) or {}).get("lds_bank_conflict_fraction"),
            "note": "bound = LDS gather rate (ds_read_b128 peak 256 B/clk/CU x 256 CUs x 2.4 GHz; random 16-byte gathers conflict ~2.8x); "
                    "HBM side: counter bytes vs the compulsory one pass over the codes"}


# ------------------------------------------------------------------------------------------------------------------
# multi-GPU launch: one process per GPU.  The driver starts the ranks itself (torch.distributed.run exports RANK /
# WORLD_SIZE / LOCAL_RANK); a bare `python bench.py --gpus N` re-executes itself under the same launcher, so the flag alone
# is enough and can never silently produce an N = 1 line.
# ------------------------------------------------------------------------------------------------------------------
def self_spawn_if_needed(args):
    """--gpus N > 1 without a launcher environment: re-exec this very command line under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` and exit with
    its status (rank 0's JSON line goes to our stdout unchanged)."""
    if args.gpus <= 1 or "RANK" in os.environ or "WORLD_SIZE" in os.environ:
        return
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "--", os.path.abspath(sys.argv[0])] + sys.argv[1:]   # `--`: the launcher's argparse would
    # otherwise try to complete our own flags (`--n` is an ambiguous prefix of its --nnodes / --nproc-per-node)
    log(f"[bench] --gpus {args.gpus}: no launcher environment, starting {args.gpus} ranks: {' '.join(cmd)}")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def launcher_world(args):
    """(rank, world, local_rank) from the launcher's environment; --gpus must agree with it."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks — refusing to print a line whose "
                         f"n_gpus would not be what was asked for")
    return rank, world, local


class RankSet:
    """The ranks of this run as the ENGINE sees them: for world > 1 the library's own RCCL communicator (jv_hip_comm_create,
    the rendezvous id travels over torch.distributed) — `rccl_ranks` is what ncclCommCount says about it, and the per-rank
    timings are all-gathered through it (jv_hip_comm_all_gather), so an N-GPU line is evidence that N RCCL ranks ran."""

    def __init__(self, ctx, rank, world):
        self.rank, self.world, self.comm = rank, world, None
        if world > 1:
            from jvector_amd.sharded import Communicator
            box = [Communicator.unique_id(ctx) if rank == 0 else None]
            torch.distributed.broadcast_object_list(box, src=0)
            self.comm = Communicator(ctx, rank, world, box[0])

    def rccl_ranks(self):
        return self.comm.count() if self.comm is not None else 1

    def gather(self, values):
        """[world, len(values)] float64, every rank gets the same table"""
        if self.comm is None:
            return np.asarray([list(values)], dtype=np.float64)
        return self.comm.all_gather_f64(values)

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None


def aggregate(ranks, elapsed, units_this_rank):
    """contract timing: max over ranks of the barrier-bracketed time; per-rank throughput for the record"""
    table = ranks.gather([elapsed, units_this_rank])
    per_rank = [float(u / t) if t > 0 else 0.0 for t, u in table]
    return float(table[:, 0].max()), float(table[:, 1].sum()), per_rank

# ------------------------------------------------------------------------------------------------------------------
# C2: SIFT-like 1M x 128, L2, PQ-16, flat two-pass ADC search
# ------------------------------------------------------------------------------------------------------------------
def run_c2(args, ctx, J, dev, world, rank, barrier, ranks):
    VSF = J.VectorSimilarityFunction.EUCLIDEAN
    N, D, M, K = (args.n if args.n != 10_000_000 else 1_000_000), 128, 16, args.topk
    QF = args.queries or 1024
    t_setup = time.perf_counter()
    base = sift_like(N, D, seed=2, device=dev)
    queries = sift_like(QF * (args.steps + args.warmup), D, seed=3 + 1000 * rank, device=dev)
    cal_q, eval_q = sift_like(1024, D, seed=13, device=dev), sift_like(max(args.eval_queries, 1024), D, seed=14, device=dev)
    g = torch.Generator(device=dev).manual_seed(4)
    sample = base[torch.randperm(N, generator=g, device=dev)[:min(128_000, N)]].contiguous()
    t0 = time.perf_counter()
    pq = J.ProductQuantization.compute(ctx, sample, M, seed=4)
    ctx.sync()
    train_s = time.perf_counter() - t0
    vs = J.VectorSet(ctx, base)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    flat = J.FlatSearcher(ctx, pq, cv, vs, max_queries=QF)

    def run(qs, rk):
        return flat.search(qs, VSF, K, rk)

    cal_gt = ground_truth(J, ctx, vs, cal_q, VSF, K, dense=True).cpu().numpy()
    eval_gt = ground_truth(J, ctx, vs, eval_q, VSF, K, dense=True).cpu().numpy()
    ladder = [args.rerank] if args.rerank > 0 else [50, 100, 200, 400, 800, 1200, 1600, 2400, 3200, 4096]
    rerank_k, cal_rec = calibrate(run, ctx, ladder, cal_q, cal_gt, QF, "c2 flat")
    rec, rec_se = evaluate(run, ctx, eval_q, eval_gt, QF, rerank_k)
    setup_s = time.perf_counter() - t_setup
    for w in range(args.warmup):
        run(queries[w * QF:(w + 1) * QF], rerank_k)
    ctx.profile(True)
    elapsed = timed_steps(run, queries[args.warmup * QF:], QF, args.steps, rerank_k, barrier)
    prof = {r: ctx.profile_read(r) for r in ("adc", "adc_exact", "sample", "topk", "exact", "lut")}
    ctx.profile(False)
    rccl_ranks = ranks.rccl_ranks()
    elapsed, total_q, per_rank = aggregate(ranks, elapsed, QF * args.steps)
    if rank != 0:
        return None
    cfg = {"n_vectors": N, "dim": D, "pq_subspaces": M, "queries_per_step": QF, "rerankK": rerank_k}
    line = {"metric": "QPS@recall10>=0.95 (SIFT1M-like 1Mx128 L2, PQ-16 ADC search); distances/sec as % roofline",
            "value": total_q / elapsed, "unit": "queries/s", "n_gpus": world, "rccl_ranks": rccl_ranks, "per_rank_qps": per_rank,
            "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE C2: synthetic SIFT-like {N}x{D} (clip(round(|N(0,1)|*40),0,218)), L2, PQ-{M} "
                                   f"(engine-trained, k=256), two-pass flat search: ADC scan of all codes -> top-{rerank_k} -> exact "
                                   f"rerank -> top-{K}", "mode": "flat", **cfg, "topK": K, "similarity": "EUCLIDEAN",
                       "parallelism": "1 GPU" if world == 1 else f"{world} replicas"},
            "recall_at_10": rec, "recall_se": rec_se, "recall_ok": rec >= 0.95, "recall_eval_queries": int(eval_q.shape[0]),
            "recall_calibration": {"queries": int(cal_q.shape[0]), "recall": cal_rec, "disjoint_from_eval": True},
            "roofline": flat_roofline(QF, N, M, prof["adc"][0], prof["adc"][1], cfg, "L2", bq=ctx.stat("adc_bq_calls") > 0, exact_ms=prof["adc_exact"]),
            "kernel_ms_per_step": {r: prof[r][0] / args.steps for r in prof},
            "adc_distances_per_s": float(QF) * N * args.steps * world / elapsed, "pq_train_s": train_s, "setup_s": setup_s}
    if world == 1 and not args.no_cpu_baseline:
        tq = queries[args.warmup * QF:]
        ids_gpu, _ = run(tq[:QF], rerank_k)
        ctx.sync()
        line["cpu_baseline"] = cpu_baseline_flat(pq.codebooks(), D, M, cv.get(0, N), base, tq, VSF, K, rerank_k,
                                                 ids_gpu.cpu().numpy(), max_q=16 * effective_cpus())
    return line


# ------------------------------------------------------------------------------------------------------------------
# C4: sharded index — every rank owns one contiguous ordinal range; jv_hip_sharded_search_flat (RCCL inside the library)
# ------------------------------------------------------------------------------------------------------------------
def sharded_ground_truth(J, ctx, vs, queries, vsf, k, id_base, world):
    """exact global top-k of a sharded index: every rank's exact top-k over its own shard (dense MFMA candidates rescored
    bit-exactly, benchlib.ground_truth), GLOBAL ids, all-gathered over torch.distributed and merged by score (untimed)"""
    ids = ground_truth(J, ctx, vs, queries, vsf, k, dense=True)
    sc = vs.scores(queries, vsf, ids.contiguous())
    ctx.sync()
    ids = ids.to(torch.int64) + int(id_base)
    if world == 1:
        return ids.cpu().numpy()
    all_ids = [torch.empty_like(ids) for _ in range(world)]
    all_sc = [torch.empty_like(sc) for _ in range(world)]
    torch.distributed.all_gather(all_ids, ids.contiguous())
    torch.distributed.all_gather(all_sc, sc.contiguous())
    ids_c, sc_c = torch.cat(all_ids, 1), torch.cat(all_sc, 1)
    top = torch.topk(sc_c, k, dim=1).indices
    return torch.gather(ids_c, 1, top).cpu().numpy()


def run_c4(args, ctx, J, dev, world, rank, barrier, ranks):
    from jvector_amd.sharded import Communicator, CShardedFlatSearcher
    VSF = J.VectorSimilarityFunction.COSINE
    n_shard = args.n if args.n != 10_000_000 else 12_500_000
    D, M, K, QF = args.dim, args.m, args.topk, (args.queries or 256)
    mix = Mixture(D, seed=5, device=dev, latent=args.latent)
    base = mix.sample(n_shard, seed=5 + rank)                      # shard-local block (seeds 5 + i), never on the host
    queries = mix.sample(QF * (args.steps + args.warmup), seed=6)   # the SAME queries on every rank: one sharded index, one answer
    eval_q = mix.sample(min(args.eval_queries, 2048), seed=8)
    g = torch.Generator(device=dev).manual_seed(4)
    pq_bytes = None
    if rank == 0:
        sample = base[torch.randperm(n_shard, generator=g, device=dev)[:128_000]].contiguous()
        pq_bytes = J.ProductQuantization.compute(ctx, sample, M, seed=4).write(6)
    if world > 1:  # codebooks travel over the host's own channel
        box = [pq_bytes]
        torch.distributed.broadcast_object_list(box, src=0)
        pq_bytes = box[0]
    pq = J.ProductQuantization.load(ctx, pq_bytes)
    vs = J.VectorSet(ctx, base)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    # the library's communicator: the N-rank one of this run, or (N = 1) a real RCCL communicator of one rank
    comm = ranks.comm if ranks.comm is not None else Communicator(ctx, 0, 1, Communicator.unique_id(ctx))
    rccl_ranks = comm.count()
    s = CShardedFlatSearcher(ctx, comm, pq, [(cv, vs, rank * n_shard)], max_queries=max(QF, 256))

    def run(qs, rk):
        return s.search(qs, VSF, K, rk)

    # recall of the sharded search against the exact global top-K (every rank computes the same numbers)
    gt = sharded_ground_truth(J, ctx, vs, eval_q, VSF, K, rank * n_shard, world)
    ladder = [args.rerank] if args.rerank > 0 else [30, 40, 50, 60, 80, 100, 150, 200, 400]
    rerank_k, rec, rec_se = ladder[-1], 0.0, 0.0
    for rk in ladder:
        found = torch.cat([run(eval_q[i:i + 256].contiguous(), rk)[0].clone() for i in range(0, eval_q.shape[0], 256)])
        ctx.sync()
        r = recall_per_query(found.cpu().numpy(), gt)
        rerank_k, rec, rec_se = rk, float(r.mean()), float(r.std(ddof=1) / math.sqrt(len(r))) if len(r) > 1 else 0.0
        log(f"[c4] rerankK={rk}: recall@{K} = {rec:.4f} +- {rec_se:.4f} on {len(r)} queries over {n_shard * world} vectors")
        if rec - 2.0 * rec_se >= 0.95:
            break
    for w in range(args.warmup):
        run(queries[w * QF:(w + 1) * QF], rerank_k)
    ctx.profile(True)
    elapsed = timed_steps(run, queries[args.warmup * QF:], QF, args.steps, rerank_k, barrier)
    prof = {r: ctx.profile_read(r) for r in ("adc", "adc_exact", "sample", "topk", "exact", "lut")}
    ctx.profile(False)
    # every rank answers every query (one index, one answer): the job's throughput is queries / max-over-ranks time
    elapsed, _, per_rank = aggregate(ranks, elapsed, QF * args.steps)
    if rank != 0:
        if ranks.comm is None:
            comm.close()
        return None
    N = n_shard * world
    cfg = {"n_vectors": n_shard, "dim": D, "pq_subspaces": M, "queries_per_step": QF, "rerankK": rerank_k}
    gather_bytes = QF * rerank_k * 8 + QF * rerank_k * 4 + 136 * 8   # per rank per step: (ids, scores) + exact scores + the agreement header
    line = {"metric": "QPS@recall10>=0.95, sharded flat search (ADC scan of every shard + RCCL partial-top-k all-gather + owner rerank)",
            "value": QF * args.steps / elapsed, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "rccl_ranks": rccl_ranks, "per_rank_qps": per_rank,
            "config": {"workload": f"BASELINE C4: {N} x {D} cosine mixture (latent {args.latent}) in {world} shard(s) of {n_shard}, PQ-{M}; per query "
                                   f"batch: per-shard ADC scan -> top-{rerank_k}, all-gather, NodeQueue-order merge, exact scores by the owning "
                                   f"shard, all-gather + owner selection, top-{K} (jv_hip_sharded_search_flat)", "n_vectors": N,
                       "shard": n_shard, "dim": D, "pq_subspaces": M, "queries_per_step": QF, "rerankK": rerank_k, "topK": K,
                       "parallelism": f"{world} rank(s), one shard each, RCCL all-gather of Q x rerankK x 8 B per shard",
                       "rccl_bytes_per_rank_per_step": gather_bytes},
            "recall_at_10": rec, "recall_se": rec_se, "recall_ok": rec >= 0.95, "recall_eval_queries": int(eval_q.shape[0]),
            "adc_distances_per_s": float(QF) * N * args.steps / elapsed,
            "kernel_ms_per_step": {r: prof[r][0] / args.steps for r in prof},
            # the per-shard kernel: every rank scans ITS shard for every query, so the kernel roofline is per GPU
            "roofline": flat_roofline(QF, n_shard, M, prof["adc"][0], prof["adc"][1], cfg, bq=ctx.stat("adc_bq_calls") > 0, exact_ms=prof["adc_exact"]), "cpu_baseline": None}
    if world == 1 and not args.no_cpu_baseline:
        tq = queries[args.warmup * QF:]
        ids_gpu, _ = run(tq[:QF], rerank_k)
        ctx.sync()
        line["cpu_baseline"] = cpu_baseline_flat(pq.codebooks(), D, M, cv.get(0, n_shard), base, tq, VSF, K, rerank_k,
                                                 ids_gpu.cpu().numpy())
    if ranks.comm is None:
        comm.close()
    return line


# ------------------------------------------------------------------------------------------------------------------
# C5: index build — PQ training + encode + batched Vamana construction with the engine's scoring, N x 1536, PQ-192
# ------------------------------------------------------------------------------------------------------------------
def run_c5(args, ctx, J, dev, world, rank, barrier, ranks):
    from jvector_amd.builder import build_vamana
    VSF = J.VectorSimilarityFunction.COSINE
    # BASELINE C5 shape unless --dim / --m were given explicitly (the CPU dry run uses a toy shape)
    D, M = (1536, 192) if (args.dim, args.m) == (768, 96) else (args.dim, args.m)
    # default 1M (a few seconds); `--n 10000000` given explicitly runs BASELINE's full 10M x 1536
    N, K = (args.n if any(a == "--n" or a.startswith("--n=") for a in sys.argv) else 1_000_000), args.topk
    mix = Mixture(D, seed=7, device=dev)
    base = mix.sample(N, seed=7 + 1000 * rank)
    eval_q = mix.sample(min(args.eval_queries, 4096), seed=9)
    g = torch.Generator(device=dev).manual_seed(4)
    sample = base[torch.randperm(N, generator=g, device=dev)[:min(128_000, N)]].contiguous()
    barrier()
    t_all = time.perf_counter()
    t0 = time.perf_counter()
    pq = J.ProductQuantization.compute(ctx, sample, M, seed=4)
    ctx.sync()
    train_s = time.perf_counter() - t0
    vs = J.VectorSet(ctx, base)
    t0 = time.perf_counter()
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    ctx.sync()
    encode_s = time.perf_counter() - t0
    ctx.profile(True)
    rd0 = (ctx.stat("rd_tests"), ctx.stat("rd_pairs"))
    nbrs, entry, bstats = build_vamana(ctx, pq, cv, base, VSF, max_degree=args.degree, beam_width=args.build_beam, alpha=args.build_alpha, log=log,
                                       vector_set=vs, overflow=args.build_overflow, max_batch=args.build_max_batch)
    prof = {r: ctx.profile_read(r) for r in ("gsearch", "adc", "prune")}
    rd_tests, rd_pairs = ctx.stat("rd_tests") - rd0[0], ctx.stat("rd_pairs") - rd0[1]   # the robust prune's own work counters
    ctx.profile(False)
    barrier()
    total_s = time.perf_counter() - t_all
    rccl_ranks = ranks.rccl_ranks()
    total_s, _, per_rank = aggregate(ranks, total_s, N)
    # quality of what was built: recall@10 of a search over it (graph + exact rerank) against brute force
    gt = ground_truth(J, ctx, vs, eval_q, VSF, K, dense=True).cpu().numpy()
    nbrs_h = nbrs.cpu().numpy()
    graph = J.GraphIndex(ctx, N, [(None, nbrs_h)], entry, 0)
    fused = J.FusedPQ.build(ctx, cv, nbrs)
    s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=int(eval_q.shape[0]))
    rec = {}
    for rk in (50, 100, 200):
        ids = s.search(eval_q, VSF, K, rk)[0]
        ctx.sync()
        rec[rk] = float(recall_per_query(ids.cpu().numpy(), gt).mean())
    if rank != 0:
        return None
    # dominant kernel: the construction-time graph search (device traversal over the growing adjacency, the neighbours' own
    # codes — no fused blocks exist yet).  SURVEY §8d rows 5/6: M code bytes + 4 B ordinal + 4 B score per scored node.
    g_ms, g_n = prof["gsearch"]
    unit = M + 8
    ach = bstats["visited"] * unit / (g_ms / 1e3) / 1e9 if g_ms > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": f"graph_search_pairc_kernel<COSINE,CH16={M // 16}> over the builder's device-resident adjacency (64-wide working rows: one lane "
                f"per neighbour probes the visited set, the fresh ones are scored {2 if M <= 96 else 4} lanes each; PQDecoder.similarityTo on the neighbours' own codes, table-free)", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                "scored_nodes": bstats["visited"], "expansions": bstats["expanded"], "bytes_per_scored_node": unit, "kernel_ms": g_ms,
                "launches": g_n, "prune_and_pair_score_kernel_ms": prof["adc"][0] + prof["prune"][0],
                "note": "algorithmic bytes = scored nodes x (M + 8) (SURVEY §8d: ADC gather by ordinal); physically the kernel is bound by the L2 "
                        "gather rate of its table-free scoring, like the search kernel of the headline (DESIGN.md §4); the robust-prune / pair-score "
                        "kernels gather M pair-table entries of 4 B per (candidate, selected) pair from L2 / MALL"}
    # the kernel that dominates the rest of the build: the robust prune (VamanaDiversityProvider.retainDiverse over the candidate lists and
    # the overflowed back-link lists).  Unit: one (candidate, selected neighbour) pair of an isDiverse test = M pair-table
    # entries of 4 B, gathered from the M x 256 x 257 / 2 x 4 B triangular table (12.6 MB at PQ-96, 25.3 MB at PQ-192: L2 / MALL resident,
    # not HBM) — priced against the measured 32-byte L2 gather ceiling of this GPU (tools/gather_bench.hip: 396 G rows/s), one row per entry.
    p_ms, p_n = prof["prune"]
    entries_per_s = rd_pairs * M / (p_ms / 1e3) if p_ms > 0 else 0.0
    prune_roofline = {"bound": "l2_gather", "kernel": "retain_diverse_kernel (one wavefront per candidate list; a test's lanes each gather the M "
                      "pair-table entries of one selected slot — 4 B from a 128-byte line of the L2 / Infinity-Cache resident triangular table)",
                      "achieved": entries_per_s / 1e9, "peak": 395.9, "unit": "G gathered entries/s", "frac": entries_per_s / 395.9e9, "traffic": None,
                      "pairs": rd_pairs, "tests": rd_tests, "entries_per_pair": M, "bytes_per_pair": 4 * M, "kernel_ms": p_ms, "launches": p_n,
                      "table_bytes": M * 256 * 257 // 2 * 4, "algorithmic_GBps": rd_pairs * 4.0 * M / (p_ms / 1e3) / 1e9 if p_ms > 0 else 0.0,
                      "note": "pairs / tests are counted by the kernel itself (jv_hip_ctx_get_stat rd_pairs / rd_tests); the ceiling is the divergent-"
                              "gather rate of the vector-memory pipe for L2 HITS (profiles/r2_gather_bench.log) — a ceiling this kernel's access "
                              "pattern does not have: at the C5 shape its L2 hit rate is 0.19 (profiles/traffic_r5.json, retain_diverse), four of five "
                              "look-ups are served by the Infinity Cache; read `frac` as an upper bound of what an L2-resident table would allow",
                      "l2_hit_rate_measured": 0.187}
    line = {"metric": "index build: nodes/s (batched Vamana, PQ scoring) incl. PQ training + encode", "value": N * world / total_s,
            "unit": "nodes/s", "n_gpus": world, "rccl_ranks": rccl_ranks, "per_rank_nodes_per_s": per_rank, "steps": 1, "warmup": 0,
            "ms_per_step": total_s * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE C5: synthetic {N}x{D} cosine mixture, PQ-{M} trained + encoded by the engine, batched Vamana "
                                   f"construction behind jv_hip_builder_* (maxDegree {args.degree}, beamWidth {args.build_beam}, alpha {args.build_alpha}, neighborOverflow "
                                   f"{args.build_overflow}, prefix-doubling batches): candidates from the engine's device-resident graph search over the partial "
                                   "graph, robust prune = the retain_diverse kernel, backlinks + re-prune with PQ diversity scores", "n_vectors": N,
                       "dim": D, "pq_subspaces": M, "max_degree": args.degree, "parallelism": "1 GPU" if world == 1 else f"{world} independent builds"},
            "seconds": {"pq_train": train_s, "encode": encode_s, "search": bstats["search_s"], "prune": bstats["prune_s"],
                        "backlink": bstats["backlink_s"], "total": total_s},
            "build": {"batches": bstats["batches"], "reprunes": bstats["reprunes"], "avg_degree": bstats["avg_degree"]},
            "recall_at_10_by_rerankK": rec, "recall_eval_queries": int(eval_q.shape[0]),
            "roofline": roofline, "prune_roofline": prune_roofline, "prune_roofline_frac": prune_roofline["frac"], "cpu_baseline": None}
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline_build(pq.codebooks(), D, M, cv.get(0, N), nbrs_h, entry, base, VSF, args.degree, args.build_beam, 1.2)
    return line


# ------------------------------------------------------------------------------------------------------------------
def batch_sweep(run, ctx, queries, rerank_k, sizes=(1, 16, 256, 1024, 4096, 131072)):
    """End-to-end search time (traversal + rerank + top-k, inputs resident) as a function of the batch size: the engine picks the
    workgroup form of the traversal (one query per CU, ADC table in LDS) for small batches and the one-wave form (eight queries
    per CU, table-free) for large ones; each small size is also timed with the other form forced, so the line shows what the choice
    buys.  Reference harness: jvector-examples/.../benchmarks/LatencyBenchmark.java."""
    out = []
    for qb in sizes:
        if qb > queries.shape[0]:
            continue
        qs = queries[:qb]
        entry = {"queries": qb}
        for label, opt in (("auto", None), ("one_wave", 0), ("workgroup", 1)):
            if label != "auto" and qb > 4096:
                continue
            ctx.set_option("gs_wgx", opt)
            try:
                run(qs, rerank_k)
                ctx.sync()
                t0 = time.perf_counter()
                run(qs, rerank_k)
                ctx.sync()
                one = max(time.perf_counter() - t0, 1e-6)
                reps = max(2, min(100, int(0.4 / one)))
                t0 = time.perf_counter()
                for _ in range(reps):
                    run(qs, rerank_k)
                ctx.sync()
                dt = (time.perf_counter() - t0) / reps
                entry[label] = {"ms_per_batch": dt * 1e3, "qps": qb / dt, "form": "workgroup" if ctx.stat("gs_last_wgx") else "one-wave"}
            finally:
                ctx.set_option("gs_wgx", None)
        log(f"[batch sweep] Q={qb}: " + ", ".join(f"{k} {v['ms_per_batch']:.3f} ms ({v['form']})" for k, v in entry.items() if isinstance(v, dict)))
        out.append(entry)
    return out


SUB_RUNS = (
    # key, argv, what it is[, extra environment]
    ("reference_order", ["--queries", "131072", "--steps", "3", "--warmup", "1", "--no-flat", "--no-cpu-baseline"],
     "the headline at 10M built in the REFERENCE's order (bl_ref_order = 1: NodeArray lists with their stored insertion scores, diverseBefore, "
     "improveConnections as the reference runs it — the order that equals the oracle's one-thread GraphIndexBuilder byte for byte on one-node "
     "batches): what the reference would build, same calibration rule", {"JVECTOR_HIP_BL_REF_ORDER": "1"}),
    ("hard_case", ["--latent", "64", "--queries", "65536", "--steps", "3", "--warmup", "1", "--no-flat"],
     "the headline pipeline on a HARDER distribution (intrinsic dimension 64 instead of 32): same code, same builder, same calibration rule"),
    ("literal_c3", ["--generator", "literal", "--queries", "16384", "--steps", "3", "--warmup", "1", "--cal-queries", "1024",
                    "--eval-queries", "2048", "--no-flat", "--no-cpu-baseline"],
     "SURVEY §8d's literal C3 generator (sigma 0.1 per coordinate in all 768 dimensions) at the metric's size, 10M vectors: an isotropic cloud"),
    ("c2", ["--workload", "c2"], "BASELINE config 2"),
    ("c5", ["--workload", "c5", "--n", "10000000"], "BASELINE config 5 at its full size (10M x 1536)"),
    ("c4_one_shard", ["--workload", "c4"], "BASELINE config 4, one of its eight 12.5M shards on one GPU"),
)


def run_sub_workloads(keys=None):
    """The other single-GPU BASELINE configurations and the data-sensitivity points, each as its own process of this script (its
    own device memory, its own calibration), timed inside the same driver run.  Returns {key: line or {"error": ...}}."""
    out = {}
    for key, argv, what, *more in SUB_RUNS:
        if keys is not None and key not in keys:
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--sub-line"] + argv
        t0 = time.perf_counter()
        try:
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
            env.update(more[0] if more else {})
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
            lines = [x for x in p.stdout.decode(errors="replace").splitlines() if x.startswith("{")]
            if p.returncode != 0 or not lines:
                out[key] = {"error": f"rc {p.returncode}", "stderr_tail": p.stderr.decode(errors="replace")[-600:]}
            else:
                sub = json.loads(lines[-1])
                sub["what"] = what
                sub["wall_s"] = time.perf_counter() - t0
                out[key] = sub
        except Exception as e:  # a sub-run must never take the headline line with it
            out[key] = {"error": repr(e)}
        log(f"[sub-run] {key}: " + (f"{out[key].get('value')} {out[key].get('unit')} recall {out[key].get('recall_at_10')} in {time.perf_counter() - t0:.0f} s"
                                    if "error" not in out[key] else str(out[key])[:300]))
    return out


COMPACT_LIMIT = 4096         # the driver keeps about 10 KB of stdout: the LAST line must fit with room to spare


def _r(x, nd=5):
    """A float rounded to `nd` significant digits (ints and None pass through) — the compact line's number format."""
    if isinstance(x, bool) or x is None or isinstance(x, int):
        return x
    try:
        x = float(x)
    except (TypeError, ValueError):
        return x
    if x == 0.0 or not math.isfinite(x):
        return x
    return float(f"{x:.{nd}g}")


def _short(s, n):
    s = "" if s is None else str(s)
    return s if len(s) <= n else s[: n - 3] + "..."


def _compact_roofline(r, kernel_chars=72):
    if not isinstance(r, dict):
        return None
    out = {"bound": r.get("bound"), "kernel": _short(r.get("kernel"), kernel_chars), "achieved": _r(r.get("achieved")),
           "peak": _r(r.get("peak")), "unit": r.get("unit"), "frac": _r(r.get("frac"), 4), "traffic": _r(r.get("traffic"), 6),
           "traffic_source": _short(r.get("traffic_source"), 64) if r.get("traffic_source") else None,
           "avg_launch_ms": _r(r.get("avg_launch_ms")), "launches": r.get("launches"),
           "bytes_per_launch": _r(r.get("bytes_per_launch"), 6)}
    if r.get("fused_rerank_rows"):   # round 6: the traversal kernel also reranks; both byte terms and the round-5 definition of frac
        out.update({"fused_rerank_rows": r["fused_rerank_rows"], "traversal_bytes_per_launch": _r(r.get("traversal_bytes_per_launch"), 6),
                    "rerank_bytes_per_launch": _r(r.get("rerank_bytes_per_launch"), 6),
                    "frac_traversal_bytes_only": _r(r.get("frac_traversal_bytes_only"), 4)})
    return out


def _compact_cpu(c):
    if not isinstance(c, dict):
        return None
    return {"value": _r(c.get("value")), "unit": c.get("unit"), "cores": c.get("cores"), "kind": c.get("kind"), "isa": c.get("isa"),
            "matches_gpu_topk": c.get("matches_gpu_topk"), "sample": _short(c.get("sample"), 96)}


def _compact_sub(sub):
    """ONE short object per sub-run: value, unit, recall, rerankK, the dominant kernel's roofline fraction, the CPU leg."""
    if not isinstance(sub, dict):
        return None
    if "error" in sub:
        return {"error": _short(sub["error"], 80)}
    rf = sub.get("roofline") or {}
    cfg = sub.get("config") or {}
    out = {"value": _r(sub.get("value")), "unit": sub.get("unit"), "recall": _r(sub.get("recall_at_10"), 4),
           "rerankK": cfg.get("rerankK"), "roofline_bound": rf.get("bound"), "roofline_frac": _r(rf.get("frac"), 4),
           "cpu_value": _r((sub.get("cpu_baseline") or {}).get("value"))}
    for k in ("prune_roofline_frac",):
        if k in sub:
            out[k] = _r(sub[k], 4)
    if rf.get("hbm_traffic_over_compulsory") is not None:   # the flat scans: counter HBM bytes over one pass over the codes
        out["hbm_over_compulsory"] = _r(rf["hbm_traffic_over_compulsory"], 4)
    if sub.get("graph_build_s") is not None:     # the graph sub-runs: what the build cost and how long a search is
        out["build_s"] = _r(sub["graph_build_s"], 4)
        out["expanded"] = _r(sub.get("avg_expanded"), 4)
    return out


def compact_line(line):
    """The driver-readable form of a full bench line: every key the contract names, the headline `roofline` and `cpu_baseline`
    objects with their numbers (strings shortened), one short object per sub-workload under `workloads`, nothing else.  Everything
    that was cut lives in bench_full.json (path under `full`).  Guaranteed <= COMPACT_LIMIT bytes."""
    cfg = dict(line.get("config") or {})
    cfg["workload"] = _short(cfg.get("workload"), 360)
    out = {k: line.get(k) for k in ("metric", "value", "unit", "n_gpus", "rccl_ranks", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    out["config"] = cfg
    for k in ("recall_at_10", "recall_se"):
        if k in line:
            out[k] = _r(line[k], 4)
    for k in ("recall_ok", "recall_eval_queries"):
        if k in line:
            out[k] = line[k]
    out["roofline"] = _compact_roofline(line.get("roofline"))
    if line.get("cpu_baseline") is not None:      # rank 0 at N = 1 only
        out["cpu_baseline"] = _compact_cpu(line["cpu_baseline"])
    for k in ("avg_visited", "avg_expanded", "adc_distances_per_s", "graph_build_s", "kernel_fraction_of_step"):
        if k in line and line[k] is not None:
            out[k] = _r(line[k])
    if isinstance(line.get("per_rank_qps"), list):
        out["per_rank_qps"] = [_r(x) for x in line["per_rank_qps"]]
    if isinstance(line.get("rerank"), dict):
        out["rerank_roofline_frac"] = _r(line["rerank"].get("frac"), 4)
    if isinstance(line.get("kernel_ms_per_step"), dict):
        out["kernel_ms_per_step"] = {k: _r(v, 4) for k, v in line["kernel_ms_per_step"].items() if v}
    wl = {}
    for key, *_rest in SUB_RUNS:
        if key in line:
            wl[key] = _compact_sub(line[key])
    if isinstance(line.get("flat_mode"), dict):
        wl["flat_mode"] = _compact_sub({**line["flat_mode"], "config": {"rerankK": line["flat_mode"].get("rerankK")}})
    if wl:
        out["workloads"] = wl
    if isinstance(line.get("batch_sweep"), list):
        out["latency_ms"] = {str(e.get("queries")): _r((e.get("auto") or {}).get("ms_per_batch"), 4) for e in line["batch_sweep"]
                             if isinstance(e, dict)}
    out["full"] = line.get("full")
    text = json.dumps(out, separators=(",", ":"))
    # belt and braces: should the line still be too long (a very long workload string, many ranks), drop optional parts in turn
    for drop in ("latency_ms", "kernel_ms_per_step", "per_rank_qps", "workloads"):
        if len(text) <= COMPACT_LIMIT:
            break
        out.pop(drop, None)
        text = json.dumps(out, separators=(",", ":"))
    if len(text) > COMPACT_LIMIT:
        out["config"]["workload"] = _short(out["config"]["workload"], 120)
        text = json.dumps(out, separators=(",", ":"))
    assert len(text) <= COMPACT_LIMIT, len(text)
    return text


def emit(line, args):
    """Rank 0's output.  A sub-run (--sub-line) prints its FULL line for the parent process to parse.  Every other run writes the
    full line to bench_full.json (repo root, and gpurun_out/ when that exists so that it travels back from the GPU box), echoes
    it on stderr, and prints the compact line as the LAST line of stdout."""
    if getattr(args, "sub_line", False):
        print(json.dumps(line), flush=True)
        return
    full = json.dumps(line)
    paths = [os.environ.get("JVECTOR_BENCH_FULL") or os.path.join(ROOT, "bench_full.json")]
    if not os.environ.get("JVECTOR_BENCH_FULL") and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_full.json"))
    written = None
    for path in paths:
        try:
            with open(path, "w") as f:
                f.write(full + "\n")
            written = written or (os.path.relpath(path, ROOT) if path.startswith(ROOT) else path)
        except OSError:
            pass
    line = dict(line)
    line["full"] = written
    log("[full line] " + full)
    sys.stdout.flush()
    print(compact_line(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["c3", "c2", "c4", "c5"], default="c3", help="c3 = the headline 10Mx768 config; c2 = SIFT1M-like; "
                    "c4 = sharded flat search: every rank owns --n vectors (default 12.5M = 100M / 8) and the C ABI's RCCL exchange "
                    "merges the partial top-k; c5 = index build (batched Vamana with the engine's scoring) on Nx1536, PQ-192")
    ap.add_argument("--graph", choices=["engine", "synthetic"], default="engine", help="c3 graph: jvector_amd.builder.build_vamana — "
                    "batched Vamana construction with the engine's own search / robust-prune / backlink scoring (default; BASELINE "
                    "config 5's path) — or benchgraph.py's synthetic kNN+prune graph (round 1's input preparation)")
    ap.add_argument("--mode", choices=["auto", "graph", "flat"], default="auto", help="auto = graph")
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--m", type=int, default=96)
    ap.add_argument("--degree", type=int, default=32)
    ap.add_argument("--queries", type=int, default=0, help="queries per step (0 = 131072 graph / 256 flat / 1024 c2)")
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--rerank", type=int, default=0, help="rerankK; 0 = smallest of the ladder reaching recall>=0.95 on the calibration set")
    ap.add_argument("--cal-queries", type=int, default=16384, help="calibration queries (rerankK ladder).  The rule — the smallest rung whose "
                    "calibration recall clears 0.95 by TWO standard errors — costs a rung or two of headroom at 4096 queries (se 0.0017: the "
                    "recall must reach 0.9534); 16384 halve the standard error")
    ap.add_argument("--eval-queries", type=int, default=10240, help="disjoint evaluation queries the reported recall is measured on")
    ap.add_argument("--gt-exact", action="store_true", help="ground truth by the bit-exact scalar-order scan only (slower; the "
                    "default takes 4k candidates from the MFMA dense scan and rescores them with the bit-exact kernel)")
    ap.add_argument("--traversal", choices=["host", "device"], default=os.environ.get("JVECTOR_BENCH_TRAVERSAL", "device"),
                    help="graph mode: device-resident traversal (default; what JV_TRAVERSAL_AUTO picks at this shape) or the "
                         "host batched searcher")
    ap.add_argument("--build-beam", type=int, default=int(os.environ.get("JVECTOR_BENCH_BUILD_BEAM", "100")),
                    help="engine graph: construction beam width (the reference's efConstruction / beamWidth, default 100)")
    ap.add_argument("--build-alpha", type=float, default=float(os.environ.get("JVECTOR_BENCH_BUILD_ALPHA", "1.2")), help="engine graph: robust-prune alpha")
    ap.add_argument("--build-overflow", type=float, default=float(os.environ.get("JVECTOR_BENCH_BUILD_OVERFLOW", "2.0")),
                    help="engine graph: neighborOverflow (working row width = maxDegree x overflow, <= 64).  Measured at 10M (profiles/r3_c): "
                         "2.0 builds faster (15.6M re-pruned lists instead of 36.6M) AND yields a graph that needs rerankK 95 instead of 105")
    ap.add_argument("--build-max-batch", type=int, default=int(os.environ.get("JVECTOR_BENCH_BUILD_MAX_BATCH", "131072")),
                    help="engine graph: largest insert batch (inserts of one batch do not see each other)")
    ap.add_argument("--build-passes", type=int, default=int(os.environ.get("JVECTOR_BENCH_BUILD_PASSES", "1")),
                    help="engine graph: 2 = re-insert every level-0 node against the finished graph (improveConnections for all nodes)")
    ap.add_argument("--build-improve", type=int, default=int(os.environ.get("JVECTOR_BENCH_BUILD_IMPROVE", "1")),
                    help="engine graph: passes of improveConnections over every node of a level after its last insert (search the finished "
                         "graph, MERGE with the node's neighbours, prune, backlink: jv_hip_builder_improve_batch).  Measured at 10M (profiles/"
                         "r4_i): one pass takes the calibrated rerankK from 95 to 75 (108 -> 87 expansions per query, 19 percent more QPS) for +30 s of "
                         "build; a second pass changes nothing")
    ap.add_argument("--torch-codebooks", action="store_true", help="codebooks from benchlib's torch Lloyd instead of the engine's "
                    "ProductQuantization.compute (round-1 behaviour)")
    ap.add_argument("--index-cache", default=os.environ.get("JVECTOR_BENCH_INDEX_CACHE", ""), help="npz file: synthetic graph + "
                    "codebooks are loaded from it when it exists, else built and saved (lets rocprofv3 wrap search steps only)")
    ap.add_argument("--latent", type=int, default=32, help="intrinsic dimension of the synthetic mixture (benchlib.Mixture: latent-L clusters "
                    "embedded in D dims + isotropic noise); QPS@recall is a strong function of it — 64 / 128 are the sensitivity points")
    ap.add_argument("--reranker", choices=["full", "nvq"], default="full", help="c3: what NodeQueue.rerank scores the kept candidates with — "
                    "full = the float32 rows (INLINE_VECTORS, the headline), nvq = NVQ rows encoded by the engine (the reference's NVQ_VECTORS "
                    "feature: D + 16 S bytes per candidate instead of 4 D); recall is measured against the exact ground truth either way")
    ap.add_argument("--nvq-subvectors", type=int, default=2)
    ap.add_argument("--generator", choices=["mixture", "literal"], default="mixture", help="c3 data: benchlib.Mixture (latent-L clusters, the "
                    "headline) or SURVEY §8d's literal generator (1000 Gaussian clusters, sigma 0.1 per coordinate in all D dimensions)")
    ap.add_argument("--no-sub-workloads", action="store_true", help="default c3 run on one GPU: do not append the other single-GPU BASELINE "
                    "configs (c2, c5, one c4 shard), the latent-64 hard case, the literal-generator point and the batch-size sweep")
    ap.add_argument("--sub-line", action="store_true", help="(internal) this process is one of those sub-runs: a lean line, no sub-runs of its own")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-flat", action="store_true", help="graph mode: skip the secondary flat-scan measurement")
    args = ap.parse_args()

    self_spawn_if_needed(args)
    rank, world, local = launcher_world(args)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import jvector_amd as J
    ctx = J.HipContext(local, stream=torch.cuda.current_stream().cuda_stream)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    ranks = RankSet(ctx, rank, world)
    if args.workload in ("c2", "c4", "c5"):
        line = {"c2": run_c2, "c4": run_c4, "c5": run_c5}[args.workload](args, ctx, J, dev, world, rank, barrier, ranks)
        if rank == 0:
            emit(line, args)
        ranks.close()
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    VSF = J.VectorSimilarityFunction.COSINE
    N, D, M, K = args.n, args.dim, args.m, args.topk
    if args.mode == "auto":
        args.mode = "graph"
    graph_mode = args.mode == "graph"
    # graph mode: one persistent launch serves the whole batch and ends when its LAST query does (a long search takes 2-3 ms),
    # so throughput grows with the batch until that tail is amortised: 775k / 868k / 938k QPS at 16k / 32k / 64k queries (1M run);
    # at 10M: 1.132 M QPS at 65536, 1.183 M at 131072 (profiles/r3_f) — the default
    Q = args.queries or (131072 if graph_mode else 256)
    t_setup = time.perf_counter()
    if args.generator == "literal":
        from benchlib import LiteralMixture
        mix = LiteralMixture(D, seed=5, device=dev)
    else:
        mix = Mixture(D, seed=5, device=dev, latent=args.latent)
    base = mix.sample(N, seed=5)
    queries_all = mix.sample(Q * (args.steps + args.warmup), seed=6 + 1000 * rank)
    cal_q = mix.sample(args.cal_queries, seed=7)
    eval_q = mix.sample(args.eval_queries, seed=8)

    cache = None
    if args.index_cache and os.path.exists(args.index_cache):
        cache = np.load(args.index_cache, allow_pickle=False)
        if int(cache["n"]) != N or int(cache["dim"]) != D or int(cache["m"]) != M or int(cache["degree"]) != args.degree:
            log(f"[bench] index cache {args.index_cache} was built for another shape: ignored")
            cache = None

    # ---- codebooks: the engine's ProductQuantization.compute on <= 128 000 sampled vectors (ProductQuantization.java:63-64,
    #      109-139: k-means++ seeding + 6 Lloyd rounds), deterministic in (sample, seed)
    t0 = time.perf_counter()
    if cache is not None:
        pq = J.ProductQuantization.load(ctx, cache["pq_bytes"].tobytes())
    elif args.torch_codebooks:
        from benchlib import train_codebooks
        pq = J.ProductQuantization.from_codebooks(ctx, D, M, train_codebooks(base, M, seed=4).cpu().numpy())
    else:
        g = torch.Generator(device=dev).manual_seed(4)
        sample = base[torch.randperm(N, generator=g, device=dev)[:min(128_000, N)]].contiguous()
        pq = J.ProductQuantization.compute(ctx, sample, M, seed=4)
        del sample
    ctx.sync()
    train_s = time.perf_counter() - t0
    vs = J.VectorSet(ctx, base)

    # PQ encode (row 3), timed with the engine's own events
    codes_t = torch.empty(N, M, dtype=torch.uint8, device=dev)
    cv = J.PQVectors(ctx, pq, codes_t)
    ctx.profile(True)
    J._lib.check(ctx._lib.jv_hip_pq_encode_into(ctx._h, pq._h, vs._h, 0, N, cv._h))
    enc_ms, _ = ctx.profile_read("encode")
    ctx.profile(False)
    if not codes_t.is_cuda:  # host tensors are copied, not wrapped (CPU dry run of this script against the mock device)
        codes_t.copy_(torch.from_numpy(cv.get(0, N)))

    rerank_vs, nvq_info = vs, None
    if args.reranker == "nvq":   # NVQuantization.compute + encodeAll on the device (EX/Grid.java:508-514 builds the same feature)
        tn = time.perf_counter()
        nvq = J.NVQuantization.compute(ctx, vs, args.nvq_subvectors)
        ctx.sync()
        mean_s = time.perf_counter() - tn
        ctx.profile(True)
        nvq_rows = nvq.encode_all(vs)
        nenc_ms, _ = ctx.profile_read("encode")
        ctx.profile(False)
        rerank_vs = nvq_rows.as_vector_set()
        nvq_info = {"subvectors": args.nvq_subvectors, "global_mean_s": mean_s, "encode_ms": nenc_ms,
                    "encode_vectors_per_s": N / (nenc_ms / 1e3) if nenc_ms > 0 else None,
                    "bytes_per_row": D + 16 * args.nvq_subvectors, "bytes_per_row_full": 4 * D}
        log(f"[nvq] global mean {mean_s:.2f} s, encode {nenc_ms:.1f} ms ({N / max(nenc_ms, 1e-9) / 1e3:.2f} M vectors/s), "
            f"{D + 16 * args.nvq_subvectors} B per row instead of {4 * D}")

    build_s, levels, build_info = None, None, None
    if graph_mode:
        tb = time.perf_counter()
        if cache is not None:
            n_lv = int(cache["n_levels"])
            levels = [(None if l == 0 else cache[f"nodes{l}"], cache[f"nbrs{l}"]) for l in range(n_lv)]
            entry, entry_level = int(cache["entry"]), int(cache["entry_level"])
            nbrs_dev = torch.from_numpy(levels[0][1]).to(dev)
        elif args.graph == "engine":
            from jvector_amd.builder import build_hierarchical
            levels, entry, entry_level, nbrs_dev, bstats = build_hierarchical(ctx, pq, cv, base, VSF, max_degree=args.degree,
                                                                                beam_width=args.build_beam, alpha=args.build_alpha, log=log,
                                                                                overflow=args.build_overflow, max_batch=args.build_max_batch,
                                                                                passes=args.build_passes, improve=args.build_improve)
            log(f"[build] {dict(bstats)}")
            build_info = {k: (float(v) if isinstance(v, float) else v) for k, v in dict(bstats).items()}
            if args.index_cache and rank == 0:
                arrs = {"n": N, "dim": D, "m": M, "degree": args.degree, "n_levels": len(levels), "entry": entry,
                        "entry_level": entry_level, "pq_bytes": np.frombuffer(pq.write(6), dtype=np.uint8)}
                for l, (nodes, nb) in enumerate(levels):
                    arrs[f"nbrs{l}"] = nb
                    if l > 0:
                        arrs[f"nodes{l}"] = nodes
                np.savez(args.index_cache, **arrs)
        else:
            levels, entry, entry_level, nbrs_dev = build_hier_graph(base, max_degree=args.degree)
            if args.index_cache and rank == 0:
                arrs = {"n": N, "dim": D, "m": M, "degree": args.degree, "n_levels": len(levels), "entry": entry,
                        "entry_level": entry_level, "pq_bytes": np.frombuffer(pq.write(6), dtype=np.uint8)}
                for l, (nodes, nb) in enumerate(levels):
                    arrs[f"nbrs{l}"] = nb
                    if l > 0:
                        arrs[f"nodes{l}"] = nodes
                np.savez(args.index_cache, **arrs)
        fused = J.FusedPQ.build(ctx, cv, nbrs_dev)          # FusedPQ.writeInline on the device (jv_hip_fused_build)
        graph = J.GraphIndex(ctx, N, levels, entry, entry_level).set_traversal(args.traversal)
        searcher = J.GraphSearcher(ctx, graph, pq, cv, fused, rerank_vs, max_queries=Q)
        torch.cuda.synchronize()
        build_s = time.perf_counter() - tb

        def run(qs, rk, stats=False):
            return searcher.search(qs, VSF, K, rk, return_stats=stats)
    else:
        flat = J.FlatSearcher(ctx, pq, cv, rerank_vs, max_queries=Q)

        def run(qs, rk, stats=False):
            return flat.search(qs, VSF, K, rk)

    # ---- ground truth (untimed): calibration and evaluation sets are disjoint from each other and from the timed queries
    t0 = time.perf_counter()
    cal_gt = ground_truth(J, ctx, vs, cal_q, VSF, K, dense=not args.gt_exact).cpu().numpy()
    eval_gt = ground_truth(J, ctx, vs, eval_q, VSF, K, dense=not args.gt_exact).cpu().numpy()
    gt_s = time.perf_counter() - t0

    ladder = [args.rerank] if args.rerank > 0 else [20, 30, 40, 50, 60, 66, 70, 72, 74, 76, 78, 80, 82, 84, 86, 88, 90, 92, 95, 100, 105, 110, 115, 120, 125, 135, 150, 175, 200, 250, 300, 400, 600, 800, 1600]
    rerank_k, cal_rec = calibrate(run, ctx, ladder, cal_q, cal_gt, Q, f"mode={args.mode}")
    if world > 1:  # every rank serves with the same (largest calibrated) rerankK
        t_rk = torch.tensor([rerank_k], dtype=torch.int64, device=dev)
        torch.distributed.all_reduce(t_rk, op=torch.distributed.ReduceOp.MAX)
        rerank_k = int(t_rk.item())
    rec, rec_se = evaluate(run, ctx, eval_q, eval_gt, Q, rerank_k)
    log(f"[evaluate] rerankK={rerank_k}: recall@{K} = {rec:.4f} +- {rec_se:.4f} on {eval_q.shape[0]} evaluation queries")
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t_setup

    timed_q = queries_all[args.warmup * Q:]
    for w in range(args.warmup):
        run(queries_all[w * Q:(w + 1) * Q], rerank_k)
    ctx.profile(True)
    elapsed = timed_steps(run, timed_q, Q, args.steps, rerank_k, barrier)
    regions = ("gsearch", "adc", "adc_exact", "sample", "topk", "exact", "lut")
    prof = {r: ctx.profile_read(r) for r in regions}
    ctx.profile(False)

    # developer aid: JVECTOR_BENCH_ENV_SWEEP="A=1,B=2;A=3" re-times the steps under other engine environment settings
    # (stderr only; the reported line is the default configuration above)
    if os.environ.get("JVECTOR_BENCH_ENV_SWEEP"):
        for cfg in os.environ["JVECTOR_BENCH_ENV_SWEEP"].split(";"):
            kv = dict(x.split("=", 1) for x in cfg.split(",") if x)
            os.environ.update(kv)
            run(timed_q[:Q], rerank_k)
            ctx.profile(True)
            d0 = [ctx.stat(k) for k in ("gs_deferred", "gs_defer_restarts")]
            dt = timed_steps(run, timed_q, Q, min(3, args.steps), rerank_k, barrier) / min(3, args.steps)
            pr = {r: ctx.profile_read(r)[0] / min(3, args.steps) for r in regions}
            ctx.profile(False)
            dq_ = [(ctx.stat(k) - v) / (Q * min(3, args.steps)) for k, v in zip(("gs_deferred", "gs_defer_restarts"), d0)]
            log(f"[sweep] {cfg}: {dt * 1e3:.2f} ms/step, {Q / dt:.0f} QPS, kernels " +
                ", ".join(f"{r} {v:.2f}" for r, v in pr.items() if v > 0) +
                f" | per query: deferred {dq_[0]:.1f}, started over {dq_[1]:.4f}")
            for k in kv:
                os.environ.pop(k, None)

    # developer aid: JVECTOR_BENCH_SORT_QUERIES=1 re-times the steps with every batch's queries ordered by the mixture cluster nearest to them
    # (an upper bound of what query-locality ordering could buy the traversal: concurrent waves then walk the same region of the graph);
    # stderr only
    if graph_mode and os.environ.get("JVECTOR_BENCH_SORT_QUERIES") and hasattr(mix, "proj"):
        nst = min(3, args.steps)
        gk = torch.Generator(device=dev).manual_seed(99)
        keyers = {"mixture cluster (generator knowledge)": lambda blk: ((blk @ mix.proj.t()) @ mix.centers.t()).argmax(1)}
        for C_ in (64, 256, 1024):   # nearest of C base vectors taken at a fixed stride: no training, no knowledge of the generator
            piv = base[:: max(1, N // C_)][:C_].contiguous()
            keyers[f"nearest of {C_} strided base vectors"] = (lambda blk, piv=piv: (blk @ piv.t()).argmax(1))
        for bits in (8, 12):
            Rm = torch.randn(D, bits, generator=gk, device=dev)
            w = (2 ** torch.arange(bits, device=dev)).float()
            keyers[f"{bits} random-hyperplane sign bits"] = (lambda blk, Rm=Rm, w=w: ((blk @ Rm > 0).float() @ w).long())
        keyers = {"nothing (arrival order)": None, **keyers}
        for name, keyer in keyers.items():
            sq = timed_q[:nst * Q].clone()
            for s_ in range(nst if keyer else 0):
                blk = sq[s_ * Q:(s_ + 1) * Q]
                sq[s_ * Q:(s_ + 1) * Q] = blk[torch.argsort(keyer(blk), stable=True)]
            run(sq[:Q], rerank_k)
            ctx.profile(True)
            dt = timed_steps(run, sq, Q, nst, rerank_k, barrier) / nst
            pr = {r: ctx.profile_read(r)[0] / nst for r in regions}
            ctx.profile(False)
            log(f"[sorted queries] by {name}: {dt * 1e3:.2f} ms/step, {Q / dt:.0f} QPS, kernels " + ", ".join(f"{r} {v:.2f}" for r, v in pr.items() if v > 0))

    # developer aid: JVECTOR_BENCH_IN_FLIGHT=n re-times the steps with n batches in flight — n host threads, each with its own
    # context (= its own HIP stream and scratch) and searcher over the SAME index, taking the steps round-robin — so that one
    # batch's HBM-bound rerank can overlap another's gather-bound traversal.  stderr only; the reported line stays one batch at a time.
    if graph_mode and int(os.environ.get("JVECTOR_BENCH_IN_FLIGHT", "1")) > 1:
        import threading
        nfl = int(os.environ["JVECTOR_BENCH_IN_FLIGHT"])
        ctxs = [ctx] + [J.HipContext(dev.index or 0) for _ in range(nfl - 1)]
        srs = [searcher] + [J.GraphSearcher(c, graph, pq, cv, fused, rerank_vs, max_queries=Q) for c in ctxs[1:]]
        for sr in srs:
            sr.search(timed_q[:Q], VSF, K, rerank_k)

        def worker(t):
            for s_ in range(t, args.steps, nfl):
                srs[t].search(timed_q[s_ * Q:(s_ + 1) * Q], VSF, K, rerank_k)

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(t,)) for t in range(nfl)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        log(f"[in-flight] {nfl} batches in flight: {dt * 1e3:.2f} ms/step, {Q / dt:.0f} QPS (one at a time: {elapsed / args.steps * 1e3:.2f} ms/step)")
        for sr in srs[1:]:
            sr.close()
        for c in ctxs[1:]:
            c.close()

    rccl_ranks = ranks.rccl_ranks()
    elapsed, total_queries, per_rank = aggregate(ranks, elapsed, Q * args.steps)
    # the driver's default command (the headline configuration itself) also carries the batch-size sweep and the sub-runs
    headline_run = (not args.sub_line and not args.no_sub_workloads and N == 10_000_000 and D == 768 and M == 96 and args.latent == 32 and
                    args.generator == "mixture" and args.rerank == 0 and args.graph == "engine" and args.reranker == "full")

    cfg_key = {"n_vectors": N, "dim": D, "pq_subspaces": M, "queries_per_step": Q, "rerankK": rerank_k}
    graph_stats, extra_roof = None, {}
    if graph_mode:
        # expansions / visited of exactly the timed steps (untimed repeat: the search is deterministic)
        dropped0 = ctx.stat("gs_ubr_dropped")
        defer0 = (ctx.stat("gs_deferred"), ctx.stat("gs_defer_restarts"))
        st = [run(timed_q[s * Q:(s + 1) * Q], rerank_k, stats=True)[2] for s in range(args.steps)]
        ubr_dropped = ctx.stat("gs_ubr_dropped") - dropped0    # neighbours the register-table bound form dropped unscored in those steps
        # round 6 (gs_defer): neighbours met above level 1 whose exact score was put off for good; queries that had to start over
        deferred, defer_restarts = ctx.stat("gs_deferred") - defer0[0], ctx.stat("gs_defer_restarts") - defer0[1]
        ubr_form = ctx.stat("gs_last_ubr") == 1
        graph_stats = np.concatenate(st)
        expansions = float(graph_stats[:, 1].sum())
        unit_bytes = args.degree * M + 4 * args.degree    # fused block (padding is read) + maxDegree scores
        if args.traversal == "device":
            k_ms, k_n = prof["gsearch"]
            kernel_key = "gsearch"
            kernel = (f"graph_search_kernel<COSINE,CH16={M // 16},OCC=2,PAIR> (device-resident GraphSearcher: one wavefront per query, "
                      "candidate / result queues in LDS, visited set in L2, table-free FusedPQ neighbour-block scoring; one "
                      "persistent launch per query batch)")
            note = (f"algorithmic bytes = expansions x {unit_bytes} B (SURVEY §8d row 7: maxDegree*M block bytes + 4*maxDegree "
                    "scores); the kernel additionally reads the 132 B adjacency row per expansion and gathers 32 B of the "
                    "L2-resident codebook per (fresh neighbour, subspace) — it is bound by those L2 gathers and the dependent "
                    "pop -> load -> probe -> score -> push chain, not by HBM (DESIGN.md §4)")
            if ubr_form:
                kernel_key = "gsearch_ubr"
                kernel = (f"graph_search_ubr_kernel<COSINE,CH16={M // 16}> (device-resident GraphSearcher, one wavefront per query; round 5: an "
                          "8-bit upper-bound table of the query's ADC entries, prebuilt per batch by ubr_table_kernel — the `lut` time of "
                          "kernel_ms_per_step — and held in the wave's registers, drops the fresh neighbours that provably can never be "
                          "popped; the others are compacted through LDS and scored table-free, eight lanes each)")
                note = (f"algorithmic bytes = expansions x {unit_bytes} B as before (SURVEY §8d row 7: the fused block is still read whole); "
                        f"{ubr_dropped / max(float(graph_stats[:, 0].sum()), 1.0):.3f} of the visited neighbours are dropped behind their bound "
                        "without the M codebook gathers of an exact score; ids, scores, visitedCount and expandedCount are unchanged "
                        "(DESIGN.md §4 'UBR'); above level 1 "
                        f"{deferred / max(float(graph_stats.shape[0]), 1.0):.1f} neighbours per query have their exact score deferred behind the "
                        f"layer's best result and never computed, {defer_restarts / max(float(graph_stats.shape[0]), 1.0):.4f} of the queries "
                        "start over without deferral (DESIGN.md §4 'DEFER')")
        else:
            k_ms, k_n = prof["adc"]
            kernel_key = "frontier"
            kernel = (f"frontier_direct_kernel<COSINE,CH16={M // 16},two slots per wave> (table-free FusedPQ neighbour-block "
                      "scoring; one launch per traversal round of a slot group; traversal on the host)")
            note = "host traversal: bound by the host cores, not by this kernel"
        bytes_total = expansions * unit_bytes
        # round 6: rows [0, rr_rows) of every query's kept results get their exact rerank score INSIDE the traversal wave (gs_body.h
        # gs_rr_round; counter gs_last_rr_rows): the kernel then also moves SURVEY §8d row 1's 4 D + 4 bytes per reranked candidate,
        # and its time includes that work — both terms are carried separately in the roofline object
        rr_rows = int(ctx.stat("gs_last_rr_rows")) if args.traversal == "device" and args.reranker == "full" else 0
        trav_bytes = bytes_total
        fused_rr_bytes = float(Q) * args.steps * rr_rows * (4 * D + 4)
        if rr_rows > 0:
            bytes_total += fused_rr_bytes
            kernel += (f"; round 6: the wave also computes the exact rerank score of rows [0, {rr_rows}) of its query's {rerank_k} kept results "
                       "(rows transposed through the idle LDS block, the scalar provider's chain)")
            note += (f"; + {rr_rows} reranked candidates per query x {4 * D + 4} B (SURVEY §8d row 1) scored inside this kernel: "
                     "frac = (traversal + rerank bytes) / kernel time, frac_traversal_bytes_only = the round-5 definition over the same time")
    else:
        rr_rows, trav_bytes, fused_rr_bytes = 0, 0.0, 0.0
        k_ms, k_n = prof["adc"]
        kernel_key, kernel, note, bytes_total = "adc_mq", "", "", 0.0
    k_avg_s = k_ms / 1e3 / max(k_n, 1)

    # rerank kernel roofline (row 1): candidates x (4*D + 4) bytes over the exact kernel's time
    e_ms, e_n = prof["exact"]
    if e_n > 0 and e_ms > 0:
        row_bytes = (D + 16 * args.nvq_subvectors + 4 + 4) if args.reranker == "nvq" else (4 * D + 4)
        rr_bytes = float(Q) * (rerank_k - rr_rows) * args.steps * row_bytes   # (fused rerank: only the lists' remainders are left to this region)
        if rr_rows > 0:
            e_n = args.steps   # (two scopes per step: the query norms, the packed remainders)
        ach = rr_bytes / (e_ms / 1e3) / 1e9
        rr_kernel = ("nvq_gather_kernel<COSINE> (NVQ.rerankerFor: NVQ rows of the kept candidates — one byte per dimension + 16 B per "
                     "sub-vector + the row's normalisation sum — de-quantised and chained in the scalar provider's order; VALU-bound "
                     "(~26 instructions per dimension incl. an IEEE divide), priced against HBM for comparability)"
                     if args.reranker == "nvq" else
                     (f"exact_gather_trq_kernel<COSINE> + query_sqnorm_kernel (what is left of NodeQueue.rerank outside the traversal kernel: rows "
                      f"[{rr_rows}, {rerank_k}) of every list, several queries per wavefront, and the queries' norms)" if rr_rows > 0 else
                      "exact_gather_tr_kernel<COSINE> (NodeQueue.rerank: full-resolution cosine of the "
                      "kept candidates; rows gathered by ordinal in coalesced 256-byte pieces and transposed through LDS)"))
        extra_roof["rerank"] = {"bound": "hbm", "kernel": rr_kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": ach / HBM_PEAK_GBS, "traffic": measured_traffic("exact_gather_trq" if rr_rows > 0 else "exact_gather", cfg_key),
                                "bytes_per_launch": rr_bytes / e_n, "avg_launch_ms": e_ms / e_n, "launches": e_n}

    # secondary measurement (single GPU, graph mode): the flat two-pass path on the same index, so that the ADC-scan
    # kernel (the per-shard kernel of the sharded configuration) is priced in the same run.  Not part of `value`.
    flat_info = None
    if world == 1 and not (graph_mode and args.no_flat):
        if graph_mode:
            QF = 256
            flat2 = J.FlatSearcher(ctx, pq, cv, vs, max_queries=QF)

            def frun(qs, rk):
                return flat2.search(qs, VSF, K, rk)
            f_rk, _ = calibrate(frun, ctx, ladder, cal_q[:1024], cal_gt[:1024], QF, "mode=flat")
            f_rec, f_se = evaluate(frun, ctx, eval_q[:2048], eval_gt[:2048], QF, f_rk)
            f_steps = 5
            frun(timed_q[:QF], f_rk)
            ctx.profile(True)
            f_el = timed_steps(frun, timed_q, QF, f_steps, f_rk, barrier)
            f_ms, f_n = ctx.profile_read("adc")
            f_ex = ctx.profile_read("adc_exact")
            ctx.profile(False)
        else:
            f_rk, f_rec, f_se, f_steps, f_el, (f_ms, f_n), QF = rerank_k, rec, rec_se, args.steps, elapsed, prof["adc"], Q
            f_ex = prof.get("adc_exact")
        flat_info = {"value": QF * f_steps / f_el, "unit": "queries/s", "ms_per_step": f_el / f_steps * 1e3,
                     "queries_per_step": QF, "rerankK": f_rk, "recall_at_10": f_rec, "recall_se": f_se,
                     "adc_distances_per_s": float(QF) * N * f_steps / f_el,
                     "roofline": flat_roofline(QF, N, M, f_ms, f_n, {**cfg_key, "queries_per_step": QF, "rerankK": f_rk}, bq=ctx.stat("adc_bq_calls") > 0, exact_ms=f_ex)}

    if rank == 0:
        if graph_mode:
            achieved = bytes_total / (k_ms / 1e3) / 1e9 if k_ms > 0 else 0.0
            roofline = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(kernel_key, cfg_key),
                        "traffic_source": (f"profiles/traffic_r*.json: rocprofv3 PMC passes of this configuration at rerankK "
                                           f"{measured_traffic(kernel_key, cfg_key, True)['config']['rerankK']} collected by the builder "
                                           "(replayed, not measured by this run)") if measured_traffic(kernel_key, cfg_key) is not None else None,
                        "bytes_per_launch": bytes_total / max(k_n, 1), "avg_launch_ms": k_avg_s * 1e3, "launches": k_n,
                        "expansions_per_launch": expansions / max(k_n, 1), "bytes_per_expansion": unit_bytes, "note": note}
            if rr_rows > 0:
                roofline.update({"fused_rerank_rows": rr_rows, "traversal_bytes_per_launch": trav_bytes / max(k_n, 1),
                                 "rerank_bytes_per_launch": fused_rr_bytes / max(k_n, 1),
                                 "frac_traversal_bytes_only": (trav_bytes / (k_ms / 1e3) / 1e9 / HBM_PEAK_GBS) if k_ms > 0 else 0.0})
        elif flat_info is not None:
            roofline = flat_info["roofline"]
        else:
            roofline = flat_roofline(Q, N, M, k_ms, k_n, cfg_key, bq=ctx.stat("adc_bq_calls") > 0)
        line = {
            "metric": "QPS@recall10>=0.95 (10Mx768); distances/sec as % HBM roofline",
            "value": total_queries / elapsed,
            "unit": "queries/s",
            "n_gpus": world, "rccl_ranks": rccl_ranks, "per_rank_qps": per_rank, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"synthetic {N}x{D} cosine (latent-{args.latent} mixture of 1000 clusters, unit norm), PQ-{M} (k=256, " +
                                    ("torch Lloyd x6" if args.torch_codebooks else "engine ProductQuantization.compute: k-means++ + Lloyd x6") +
                                    " on a 128k sample), " +
                                    (f"FusedADC graph search: " + (f"layered Vamana graph built by the engine's batched builder (jv_hip_builder_*: PQ scoring, beamWidth {args.build_beam}, alpha {args.build_alpha}, neighborOverflow {args.build_overflow}, {args.build_improve} improveConnections pass(es) per level; maxDegree {args.degree}, {len(levels)} levels drawn like getRandomGraphLevel), "
                                                                   if args.graph == "engine" else
                                                                   f"synthetic kNN+robust-prune graph (maxDegree {args.degree}, {len(levels)} nested levels), ") +
                                     f"{'device-resident GraphSearcher (one wavefront per query)' if args.traversal == 'device' else 'host batched GraphSearcher, GPU fused-block scoring'}, rerankK {rerank_k} -> exact rerank -> top-{K}"
                                     if graph_mode else
                                     f"two-pass flat search: ADC scan of all codes -> top-{rerank_k} -> exact rerank -> top-{K}")),
                       "mode": args.mode, "traversal": args.traversal if graph_mode else None, "n_vectors": N, "dim": D, "pq_subspaces": M, "queries_per_step": Q, "topK": K,
                       "rerankK": rerank_k, "similarity": "COSINE",
                       "parallelism": "1 GPU" if world == 1 else f"{world} replicas, queries sharded, no collective"},
            "recall_at_10": rec, "recall_se": rec_se, "recall_ok": rec >= 0.95, "recall_eval_queries": int(eval_q.shape[0]),
            "recall_calibration": {"queries": int(cal_q.shape[0]), "recall": cal_rec, "disjoint_from_eval": True},
            "roofline": roofline,
            "kernel_ms_per_step": {r: prof[r][0] / args.steps for r in prof},
            "kernel_fraction_of_step": sum(prof[r][0] for r in prof) / (elapsed * 1e3) if world == 1 else None,
            "encode": {"vectors_per_s": N / (enc_ms / 1e3) if enc_ms > 0 else None, "ms": enc_ms},
            "pq_train_s": train_s, "ground_truth_s": gt_s, "setup_s": setup_s, "graph_build_s": build_s,
            "graph": args.graph, "graph_build": build_info,
            "reranker": args.reranker, "nvq": nvq_info,
        }
        line.update(extra_roof)
        if graph_mode:
            st = graph_stats
            # which traversal served the queries of this process (jv_hip_ctx_get_stat): device-resident, re-run on the device with a
            # bigger visited table, finished by the host searcher, AUTO falling back to the host searcher altogether
            line["traversal_stats"] = {k: ctx.stat(k) for k in ("gs_calls_device", "gs_calls_host", "gs_calls_host_auto", "gs_queries_device",
                                                                "gs_queries_retried", "gs_queries_host_fallback", "gs_ties_resolved_device",
                                                                "gs_ties_to_host", "gs_last_v1_log2", "gs_last_workers_per_cu")}
            line["traversal_stats"]["deferred_per_query"] = deferred / max(float(st.shape[0]), 1.0)
            line["traversal_stats"]["defer_restart_fraction"] = defer_restarts / max(float(st.shape[0]), 1.0)
            line["avg_visited"] = float(st[:, 0].mean())
            line["avg_expanded"] = float(st[:, 1].mean())
            line["adc_distances_per_s"] = float(st[:, 0].mean()) * total_queries / elapsed
            line["expansions_per_s"] = float(st[:, 1].mean()) * total_queries / elapsed
            line["visited_percentiles"] = {k: float(np.percentile(st[:, 0], v)) for k, v in (("p50", 50), ("p99", 99), ("p99.9", 99.9), ("max", 100))}
            # the kernel's physical bound: 32-byte codebook rows gathered from L2, one per (scored neighbour, subspace);
            # ceiling = tools/gather_bench.hip on this GPU (profiles/r2_gather_bench.log: 396 G rows/s, 12.7 TB/s)
            if args.traversal == "device" and k_ms > 0:
                rows = (float(st[:, 0].sum()) - (float(ubr_dropped) if ubr_form else 0.0)) * M
                line["l2_gather"] = {"rows_per_s": rows / (k_ms / 1e3), "ceiling_rows_per_s": 395.9e9,
                                     "frac": rows / (k_ms / 1e3) / 395.9e9, "bytes_per_row": 32,
                                     "note": "scored neighbours x M codebook rows of 32 B (L2-resident 768 KB table) over the traversal "
                                             "kernel's time; ceiling measured by tools/gather_bench.hip (lane per row, 2 x dwordx4)"}
            if flat_info is not None:
                line["flat_mode"] = flat_info
            if world == 1 and args.traversal == "device" and headline_run:
                line["batch_sweep"] = batch_sweep(run, ctx, timed_q, rerank_k)
        else:
            line["adc_distances_per_s"] = float(N) * total_queries / elapsed
        if world == 1 and not args.no_cpu_baseline:
            ids_gpu = run(timed_q[:Q], rerank_k)[0]
            ctx.sync()
            codes_h = codes_t.cpu().numpy()
            cb = pq.codebooks()
            if graph_mode:
                line["cpu_baseline"] = cpu_baseline_graph(cb, D, M, codes_h, levels, entry, entry_level, base,
                                                          timed_q, VSF, K, rerank_k, ids_gpu.cpu().numpy())
            else:
                line["cpu_baseline"] = cpu_baseline_flat(cb, D, M, codes_h, base, timed_q, VSF, K, rerank_k,
                                                         ids_gpu.cpu().numpy())
            if args.reranker == "nvq" and line.get("cpu_baseline"):
                line["cpu_baseline"]["note"] = ("the CPU leg reranks with the float32 rows (the headline's reranker): its top-k differs from the "
                                                "NVQ-reranked GPU top-k by design, so the identical-results flag does not apply to this line")
        if world == 1 and graph_mode and headline_run:
            # the headline's own index is released first: the sub-runs are processes of their own on the same GPU
            for obj in ("searcher", "fused", "graph", "flat2", "rerank_vs", "cv", "vs"):
                o = locals().get(obj)
                try:
                    if o is not None and hasattr(o, "close"):
                        o.close()
                except Exception:
                    pass
            del base, codes_t
            torch.cuda.empty_cache()
            line.update(run_sub_workloads())
        emit(line, args)
    ranks.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
