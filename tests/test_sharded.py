"""Sharded-index path (BASELINE config 4): N-shard merged result == single-index result, bit-identical ids and
scores, including engineered score ties.

* CPU, world_size 2, gloo: exercises the host logic (shard bounds, global ids, all_gather / MAX all_reduce, merge)
  with a CPU checker backend built on the oracle (tests may use the oracle; the product backend is HIP-only).
* GPU: the same equality with the HIP backend, several shards resident on the one available GPU.
"""
import os
import socket

import numpy as np
import pytest
import torch

from jvector_amd.sharded import ShardedFlatSearcher, shard_bounds
from oracle import oracle as O


def make_problem(seed, N=6000, D=64, M=8, Q=5, dup=True):
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((30, D)).astype(np.float32)
    vecs = (centers[rng.integers(0, 30, N)] + 0.4 * rng.standard_normal((N, D))).astype(np.float32)
    if dup:  # engineered ties: identical vectors living in different shards -> identical scores, ids decide
        vecs[N - 7] = vecs[3]
        vecs[N // 2 + 1] = vecs[3]
        vecs[N // 3] = vecs[N - 1]
    queries = (vecs[[3, N - 1, 10, 20, 30][:Q]] + 0.01 * rng.standard_normal((Q, D))).astype(np.float32)
    sizes, offs = O.subvector_sizes_offsets(D, M)
    pick = rng.choice(N, 256, replace=False)
    cb = np.concatenate([vecs[pick, offs[m]: offs[m] + sizes[m]].reshape(-1) for m in range(M)])
    return vecs, queries, cb


class OracleShardBackend:
    """CPU checker backend (test infrastructure): same interface as HipShardBackend, oracle arithmetic."""

    def __init__(self, opq, codes, vecs, lo):
        self.opq, self.codes, self.vecs, self.lo, self.count = opq, codes, vecs, lo, codes.shape[0]

    def adc_topk(self, queries, vsf, k):
        q = queries.numpy()
        ids = np.full((q.shape[0], k), -1, np.int32)
        sc = np.full((q.shape[0], k), -np.inf, np.float32)
        for i in range(q.shape[0]):
            a = self.opq.adc_scores(q[i], int(vsf), self.codes)
            ti, ts = O.topk(None, a, k)
            ids[i, : len(ti)] = ti + self.lo
            sc[i, : len(ti)] = ts
        return torch.from_numpy(ids), torch.from_numpy(sc)

    def exact_scores(self, queries, vsf, global_ids):
        q, g = queries.numpy(), global_ids.numpy()
        out = np.full(g.shape, -np.inf, np.float32)
        for i in range(g.shape[0]):
            for j in range(g.shape[1]):
                loc = g[i, j] - self.lo
                if 0 <= loc < self.count:
                    out[i, j] = O.compare(int(vsf), q[i], self.vecs[loc])
        return torch.from_numpy(out)

    def topk(self, scores, ids, k):
        s, d = scores.numpy(), ids.numpy()
        oi = np.full((s.shape[0], k), -1, np.int32)
        osc = np.full((s.shape[0], k), -np.inf, np.float32)
        for i in range(s.shape[0]):
            valid = d[i] >= 0
            ti, ts = O.topk(d[i][valid], s[i][valid], k)
            oi[i, : len(ti)] = ti
            osc[i, : len(ti)] = ts
        return torch.from_numpy(oi), torch.from_numpy(osc)


def oracle_single(opq, codes, vecs, queries, vsf, top_k, rerank_k):
    return opq.search_flat(codes, vecs, queries, int(vsf), top_k, rerank_k, nthreads=2)


def test_shard_bounds():
    assert shard_bounds(10, 3) == [(0, 4), (4, 8), (8, 10)]
    assert shard_bounds(8, 8) == [(i, i + 1) for i in range(8)]
    b = shard_bounds(100_000_000, 8)
    assert b[0] == (0, 12_500_000) and b[-1] == (87_500_000, 100_000_000)


def _gloo_worker(rank, world, port, seed, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        vecs, queries, cb = make_problem(seed)
        N, D, M = vecs.shape[0], vecs.shape[1], 8
        opq = O.OraclePQ(D, M, cb)
        codes = opq.encode_all(vecs, nthreads=1)
        lo, hi = shard_bounds(N, world)[rank]
        shard = OracleShardBackend(opq, codes[lo:hi], vecs[lo:hi], lo)
        s = ShardedFlatSearcher([shard])
        out = {}
        for vsf in (O.EUCLIDEAN, O.DOT_PRODUCT, O.COSINE):
            ids, sc = s.search(torch.from_numpy(queries), vsf, 10, 40)
            wi, ws = oracle_single(opq, codes, vecs, queries, vsf, 10, 40)
            out[vsf] = bool(np.array_equal(ids.numpy(), wi) and np.array_equal(sc.numpy(), ws))
        results[rank] = out
    finally:
        dist.destroy_process_group()


def test_sharded_two_ranks_gloo_equals_single_index():
    import torch.multiprocessing as mp
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_gloo_worker, args=(2, port, 11, results), nprocs=2, join=True)
    assert dict(results) == {0: {0: True, 1: True, 2: True}, 1: {0: True, 1: True, 2: True}}


def test_sharded_local_shards_cpu_checker():
    """several shards in one process (no collective): merge logic incl. ties, uneven shards, rerankK > shard size"""
    vecs, queries, cb = make_problem(5, N=1000)
    opq = O.OraclePQ(64, 8, cb)
    codes = opq.encode_all(vecs, nthreads=1)
    for world in (1, 3, 7):
        shards = [OracleShardBackend(opq, codes[lo:hi], vecs[lo:hi], lo) for lo, hi in shard_bounds(1000, world)]
        s = ShardedFlatSearcher(shards)
        for vsf in (O.EUCLIDEAN, O.COSINE):
            ids, sc = s.search(torch.from_numpy(queries), vsf, 10, 200)
            wi, ws = oracle_single(opq, codes, vecs, queries, vsf, 10, 200)
            assert np.array_equal(ids.numpy(), wi) and np.array_equal(sc.numpy(), ws)
    with pytest.raises(ValueError):
        s.search(torch.from_numpy(queries), O.COSINE, 10, 5)


class OracleGraphShardBackend(OracleShardBackend):
    """CPU checker for HipGraphShardBackend: the partial top-k comes from the oracle's graph search over the shard's graph."""

    def __init__(self, opq, codes, vecs, lo, graph, fused):
        super().__init__(opq, codes, vecs, lo)
        self.graph, self.fused = graph, fused

    def adc_topk(self, queries, vsf, k):
        ids, sc, _ = self.graph.search(self.opq, self.codes, None, queries.numpy(), int(vsf), k, k, fused=self.fused)
        return torch.from_numpy(np.where(ids >= 0, ids + self.lo, ids).astype(np.int32)), torch.from_numpy(sc)


def _graph_shards(seed, n_shards, N=1500, D=64, M=8):
    from test_graph_search import build_problem
    shards, lo = [], 0
    for s in range(n_shards):
        v, lv, entry, entry_level, cb, q = build_problem(seed + s, N=N, D=D, M=M, levels=2)
        shards.append((v, lv, entry, entry_level, lo))
        lo += N
    return shards, cb, q[:6]


def _manual_merge(opq, shard_results, shard_vecs, q, vsf, top_k, rerank_k):
    """union of the per-shard kept results -> top-rerankK by NodeQueue key -> exact scores -> top-K"""
    out_i, out_s = [], []
    for qi in range(q.shape[0]):
        ids = np.concatenate([r[0][qi] for r in shard_results])
        sc = np.concatenate([r[1][qi] for r in shard_results])
        keep = ids >= 0
        ci, _ = O.topk(ids[keep], sc[keep], rerank_k)
        ex = np.array([O.compare(int(vsf), q[qi], shard_vecs[g]) for g in ci], np.float32)
        ti, ts = O.topk(ci, ex, top_k)
        out_i.append(ti)
        out_s.append(ts)
    return np.stack(out_i), np.stack(out_s)


def test_sharded_graph_backends_cpu_checker():
    """segment indexes: three shards, each with its own graph; sharded result == manual merge of the per-shard searches"""
    shards, cb, q = _graph_shards(40, 3)
    opq = O.OraclePQ(64, 8, cb)
    backends, results, allv = [], {}, {}
    for v, lv, entry, entry_level, lo in shards:
        codes = opq.encode_all(v, nthreads=1)
        og = O.OracleGraph(len(v), lv, entry, entry_level)
        backends.append(OracleGraphShardBackend(opq, codes, v, lo, og, True))
        for i in range(len(v)):
            allv[lo + i] = v[i]
    s = ShardedFlatSearcher(backends)
    for vsf in (O.EUCLIDEAN, O.COSINE):
        per = []
        for b in backends:
            ids, sc = b.adc_topk(torch.from_numpy(q), vsf, 30)
            per.append((ids.numpy(), sc.numpy()))
        wi, ws = _manual_merge(opq, per, allv, q, vsf, 10, 30)
        gi, gs = s.search(torch.from_numpy(q), vsf, 10, 30)
        assert np.array_equal(gi.numpy(), wi) and np.array_equal(gs.numpy(), ws)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_hip_equals_single_gpu(world):
    import jvector_amd as J
    from jvector_amd.sharded import HipShardBackend
    vecs, queries, cb = make_problem(21, N=40000, D=128, M=16, Q=5)
    N, D, M = vecs.shape[0], 128, 16
    ctx = J.HipContext(0)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    tv = torch.from_numpy(vecs).cuda()
    tq = torch.from_numpy(queries).cuda()
    vs_all = J.VectorSet(ctx, tv)
    cv_all = J.PQVectors.encode_and_build(ctx, pq, vs_all)
    single = J.FlatSearcher(ctx, pq, cv_all, vs_all, max_queries=8)
    shards = []
    for lo, hi in shard_bounds(N, world):
        vs = J.VectorSet(ctx, tv[lo:hi].contiguous())
        cv = J.PQVectors.encode_and_build(ctx, pq, vs)
        shards.append(HipShardBackend(ctx, pq, cv, vs, lo, max_queries=8))
    sharded = ShardedFlatSearcher(shards)
    for vsf in J.VectorSimilarityFunction:
        wi, ws = single.search(tq, vsf, 10, 100)
        gi, gs = sharded.search(tq, vsf, 10, 100)
        ctx.sync()
        assert torch.equal(gi, wi) and torch.equal(gs, ws), vsf
    # and the single-GPU result is the oracle's
    opq = O.OraclePQ(D, M, cb)
    wi_o, ws_o = oracle_single(opq, opq.encode_all(vecs), vecs, queries, O.COSINE, 10, 100)
    gi, gs = sharded.search(tq, J.VectorSimilarityFunction.COSINE, 10, 100)
    assert np.array_equal(gi.cpu().numpy(), wi_o) and np.array_equal(gs.cpu().numpy(), ws_o)
