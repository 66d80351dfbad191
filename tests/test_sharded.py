"""Sharded-index path (BASELINE config 4): N-shard merged result == single-index result, bit-identical ids and
scores, including engineered score ties.

* CPU, world_size 2, gloo: two processes drive the library's ONE exchange implementation (csrc/sharded.cpp, on the mock device)
  with torch.distributed carrying its all-gathers (jv_hip_comm_create_external); results equal the oracle's single index.
* GPU: the same equality with the HIP backend, several shards resident on the one available GPU.
"""
import os
import platform
import socket
import sys

import numpy as np
import pytest
import torch

from jvector_amd.sharded import ShardedFlatSearcher, shard_bounds
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def oracle_single(opq, codes, vecs, queries, vsf, top_k, rerank_k):
    return opq.search_flat(codes, vecs, queries, int(vsf), top_k, rerank_k, nthreads=2)


def test_shard_bounds():
    assert shard_bounds(10, 3) == [(0, 4), (4, 8), (8, 10)]
    assert shard_bounds(8, 8) == [(i, i + 1) for i in range(8)]
    b = shard_bounds(100_000_000, 8)
    assert b[0] == (0, 12_500_000) and b[-1] == (87_500_000, 100_000_000)


# The exchange (agreement header, all-gathers, NodeQueue-order merge, owners' exact rerank, owner selection, top-K) has ONE
# implementation — csrc/sharded.cpp — whatever carries its all-gathers.  On the CPU the library's host code runs against the mock
# device (tests/mock/): the shards' ADC scans and rerank kernels are the oracle's arithmetic there, the exchange is the real code.
def _mock_shard(J, ctx, pq, vecs, lo, hi, max_queries=8):
    from jvector_amd.sharded import HipShardBackend
    vs = J.VectorSet(ctx, np.ascontiguousarray(vecs[lo:hi]))
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    return HipShardBackend(ctx, pq, cv, vs, lo, max_queries=max_queries)


def _gloo_worker(rank, world, port, seed, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, os.path.join(ROOT, "tests", "mock"))
    import torch.distributed as dist
    from mockbind import mock_jvector
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with mock_jvector() as J:
            vecs, queries, cb = make_problem(seed)
            N, D, M = vecs.shape[0], vecs.shape[1], 8
            opq = O.OraclePQ(D, M, cb)
            codes = opq.encode_all(vecs, nthreads=1)
            ctx = J.HipContext(0)
            pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
            lo, hi = shard_bounds(N, world)[rank]
            # no communicator given: the initialised process group (gloo) carries the library's all-gathers
            s = ShardedFlatSearcher([_mock_shard(J, ctx, pq, vecs, lo, hi)])
            assert s.comm.world == world and s.comm.count() == world
            out = {}
            for vsf in (O.EUCLIDEAN, O.DOT_PRODUCT, O.COSINE):
                ids, sc = s.search(queries, J.VectorSimilarityFunction(vsf), 10, 40)
                wi, ws = oracle_single(opq, codes, vecs, queries, vsf, 10, 40)
                out[vsf] = bool(np.array_equal(np.asarray(ids), wi) and np.array_equal(np.asarray(sc), ws))
            # ranks that disagree on their arguments all fail (none hangs in a collective the other never issues)
            try:
                s.search(queries, J.VectorSimilarityFunction.COSINE, 10, 40 if rank == 0 else 50)
                out["disagreement refused"] = False
            except Exception:
                out["disagreement refused"] = True
            results[rank] = out
            s.close()
            ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(platform.machine() != "x86_64", reason="the mock build needs the x86-64 lane emulator")
def test_sharded_two_ranks_gloo_equals_single_index():
    """world_size 2 over gloo: two processes, one shard each, the library's exchange (csrc/sharded.cpp) with its all-gathers carried by
    torch.distributed through jv_hip_comm_create_external: identical to the single index, ties included"""
    import torch.multiprocessing as mp
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_gloo_worker, args=(2, port, 11, results), nprocs=2, join=True)
    want = {0: True, 1: True, 2: True, "disagreement refused": True}
    assert dict(results) == {0: want, 1: want}


@pytest.mark.skipif(platform.machine() != "x86_64", reason="the mock build needs the x86-64 lane emulator")
def test_sharded_local_shards_on_the_mock():
    """several shards in one process (local communicator: no collective): merge logic incl. ties, uneven shards, rerankK > shard size"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "mock"))
    from mockbind import mock_jvector
    with mock_jvector() as J:
        vecs, queries, cb = make_problem(5, N=1000)
        opq = O.OraclePQ(64, 8, cb)
        codes = opq.encode_all(vecs, nthreads=1)
        ctx = J.HipContext(0)
        pq = J.ProductQuantization.from_codebooks(ctx, 64, 8, cb)
        for world in (1, 3, 7):
            s = ShardedFlatSearcher([_mock_shard(J, ctx, pq, vecs, lo, hi) for lo, hi in shard_bounds(1000, world)])
            for vsf in (O.EUCLIDEAN, O.COSINE):
                ids, sc = s.search(queries, J.VectorSimilarityFunction(vsf), 10, 200)
                wi, ws = oracle_single(opq, codes, vecs, queries, vsf, 10, 200)
                assert np.array_equal(np.asarray(ids), wi) and np.array_equal(np.asarray(sc), ws)
            with pytest.raises(ValueError):
                s.search(queries, J.VectorSimilarityFunction.COSINE, 10, 5)
            if world == 3:
                # a backend that hands over ids its shard does not own: they leave the exchange before the merge — the
                # answer is the one without them, never an id with no owner
                bad = s.shards[1]
                real = bad.adc_topk

                def lying(queries_, vsf_, k_, real=real):
                    import torch
                    ids_, sc_ = real(queries_, vsf_, k_)
                    ids_ = torch.as_tensor(np.array(ids_)).clone()
                    sc_ = torch.as_tensor(np.array(sc_)).clone()
                    ids_[:, 0] = 999_999          # outside every shard
                    sc_[:, 0] = 1e9
                    ids_[:, 1] = 3                 # a real id — of ANOTHER shard
                    sc_[:, 1] = 1e9
                    return ids_, sc_
                bad.adc_topk = lying
                ids, sc = s.search(queries, J.VectorSimilarityFunction.COSINE, 10, 200)
                assert (np.asarray(ids) < 1000).all() and (np.asarray(ids) >= 0).all() and np.isfinite(np.asarray(sc)).all()
                assert not (np.asarray(sc) > 1.0).any()
                bad.adc_topk = real
            s.close()
        ctx.close()


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
