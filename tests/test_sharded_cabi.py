"""The sharded search behind the C ABI (jv_hip_comm_* / jv_hip_sharded_*, csrc/sharded.cpp) — SURVEY §8b/§8e:
N-shard result == single-index result, bit-identical ids and scores, engineered ties included.

* CPU (mock device): local communicator with 1 / 3 / 7 shards; and a world_size-2 run, one PROCESS per rank, where the C
  library's RCCL calls land in a shared-memory shim (tests/mock/rccl_shim.cpp via JVECTOR_HIP_RCCL_PATH).
* GPU: the same equality on one MI355X with several shards per rank, through a REAL RCCL communicator of world size 1
  (librccl is dlopen'ed, ncclCommInitRank + grouped ncclAllGather run), against jv_hip_search_flat and the oracle.
"""
import ctypes as C
import os
import platform
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "mock"))

from oracle import oracle as O
from test_sharded import make_problem, oracle_single


def shard_bounds(total, world):
    per = (total + world - 1) // world
    return [(min(total, g * per), min(total, (g + 1) * per)) for g in range(world)]


def run_sharded_equals_single(J, ctx, comm, n_shards_local, rank=0, world=1, N=6000, D=64, M=8, rerank_k=40, check_oracle=True,
                              straddling_ties=False):
    """shared by the CPU and GPU tests: this rank holds n_shards_local consecutive pieces of a (world x n_shards_local)-way split"""
    from jvector_amd.sharded import CShardedFlatSearcher
    vecs, queries, cb = make_problem(11, N=N, D=D, M=M)
    vecs[17] = 0.0                                           # a zero row: cosine NaN must survive the exchange
    if straddling_ties:
        # one vector copied into EVERY shard (equal approximate and exact scores on every rank) and the first query aimed at it: the
        # copies fill the top-k, so the merge and the owners' rerank must break the tie by global id exactly like the single index
        bounds = shard_bounds(N, world * n_shards_local)
        for lo, hi in bounds:
            if hi - lo > 9:
                vecs[lo + 9] = vecs[7]
        queries = queries.copy()
        queries[0] = vecs[7] + np.float32(0.01) * queries[0]
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    vs_all = J.VectorSet(ctx, vecs)
    cv_all = J.PQVectors.encode_and_build(ctx, pq, vs_all)
    single = J.FlatSearcher(ctx, pq, cv_all, vs_all, max_queries=8)
    pieces = shard_bounds(N, world * n_shards_local)[rank * n_shards_local:(rank + 1) * n_shards_local]
    shards = []
    for lo, hi in pieces:
        vs = J.VectorSet(ctx, np.ascontiguousarray(vecs[lo:hi]))
        shards.append((J.PQVectors.encode_and_build(ctx, pq, vs), vs, lo))
    s = CShardedFlatSearcher(ctx, comm, pq, shards, max_queries=8)
    s_norerank = CShardedFlatSearcher(ctx, comm, pq, [(c, None, lo) for c, _, lo in shards], max_queries=8)
    opq = O.OraclePQ(D, M, cb)
    codes = cv_all.get(0, N)
    for vsf in J.VectorSimilarityFunction:
        wi, ws = single.search(queries, vsf, 10, rerank_k)
        gi, gs = s.search(queries, vsf, 10, rerank_k)
        assert np.array_equal(gi, wi) and np.array_equal(gs, ws, equal_nan=True), vsf
        if check_oracle:
            oi, osc = oracle_single(opq, codes, vecs, queries, int(vsf), 10, rerank_k)
            assert np.array_equal(gi, oi) and np.array_equal(gs, osc, equal_nan=True), vsf
        wi2, ws2 = J.FlatSearcher(ctx, pq, cv_all, None, max_queries=8).search(queries, vsf, 10, 0)
        gi2, gs2 = s_norerank.search(queries, vsf, 10, 25)
        assert np.array_equal(gi2, wi2) and np.array_equal(gs2, ws2), vsf
    # jv_hip_sharded_topk on hand-made partial lists with ties across ranks: (score, smaller global id first)
    Q, k = 3, 6
    rng = np.random.default_rng(5)
    sc = np.round(rng.random((world, Q, k)).astype(np.float32), 1)           # many equal scores
    ids = (np.arange(world * Q * k, dtype=np.int32).reshape(world, Q, k) * 7919) % 100003
    ids[0, 1, 2] = -1
    oi, osc = s.merge_topk(sc[rank], ids[rank], 8)
    for q in range(Q):
        alli, alls = ids[:, q].reshape(-1), sc[:, q].reshape(-1)
        ti, ts = O.topk(alli[alli >= 0], alls[alli >= 0], 8)
        assert np.array_equal(oi[q][:len(ti)], ti) and np.array_equal(osc[q][:len(ts)], ts)
    with pytest.raises(ValueError):
        s.search(queries, J.VectorSimilarityFunction.COSINE, 10, 5)            # rerankK < topK (GraphSearcher.java:233)


# ---- CPU: mock device -------------------------------------------------------------------------------------------------
def _mock_J():
    import build_mock
    import jvector_amd
    import jvector_amd._lib as L
    lib = C.CDLL(build_mock.build())
    for table in (L.SIGNATURES, L.COMPAT_SIGNATURES, L.FORMAT_SIGNATURES):
        for name, (res, args) in table.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    L._lib = lib
    return jvector_amd


def build_shim():
    out = os.path.join(ROOT, "build", "mock", "librccl_shim.so")
    src = os.path.join(ROOT, "tests", "mock", "rccl_shim.cpp")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", src, "-o", out, "-lpthread", "-lrt"])
    return out


@pytest.mark.skipif(platform.machine() != "x86_64", reason="the mock build needs the x86-64 lane emulator")
@pytest.mark.parametrize("n_local", [1, 3, 7])
def test_sharded_cabi_local_shards_on_the_mock(n_local):
    import jvector_amd._lib as L
    saved = L._lib
    try:
        J = _mock_J()
        from jvector_amd.sharded import Communicator
        ctx = J.HipContext(0)
        comm = Communicator(ctx)                                             # world 1, no id: local, RCCL never loaded
        run_sharded_equals_single(J, ctx, comm, n_local, N=3000)
        comm.close()
        ctx.close()
    finally:
        L._lib = saved


def _rank_main(rank, world, id_path, n_local, N=3000, straddling_ties=False):
    """one rank of the world_size-2 CPU run (its own process: the shim's barrier is process-shared)"""
    os.environ["JVECTOR_HIP_RCCL_PATH"] = build_shim()
    os.environ["JVECTOR_HIP_HOST_THREADS"] = "1"
    J = _mock_J()
    from jvector_amd.sharded import Communicator
    ctx = J.HipContext(0)
    if rank == 0:
        uid = Communicator.unique_id(ctx)
        with open(id_path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(id_path + ".tmp", id_path)
    else:
        import time
        for _ in range(20000):
            if os.path.exists(id_path):
                break
            time.sleep(0.001)
        uid = open(id_path, "rb").read()
    comm = Communicator(ctx, rank, world, uid)
    assert comm.count() == world                                              # what the communicator itself says (ncclCommCount)
    run_sharded_equals_single(J, ctx, comm, n_local, rank=rank, world=world, N=N, straddling_ties=straddling_ties)
    comm.close()
    ctx.close()


@pytest.mark.skipif(platform.machine() != "x86_64", reason="the mock build needs the x86-64 lane emulator")
@pytest.mark.parametrize("n_local", [1, 2])
def test_sharded_cabi_two_ranks_on_the_mock(tmp_path, n_local):
    """world_size 2, one process per rank: unique-id rendezvous, grouped all-gathers, gathered layout [rank][shard][Q][k]"""
    import build_mock
    build_mock.build()
    build_shim()
    id_path = str(tmp_path / "uid")
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_sharded_cabi as T; "
            "T._rank_main(int(sys.argv[1]), 2, %r, %d)") % (ROOT, os.path.join(ROOT, "tests"), id_path, n_local)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"


@pytest.mark.skipif(platform.machine() != "x86_64", reason="the mock build needs the x86-64 lane emulator")
def test_sharded_cabi_eight_ranks_on_the_mock(tmp_path):
    """BASELINE config 4's world size without the hardware: EIGHT processes, one rank each, through the C ABI's
    exchange on the shared-memory RCCL shim — 2995 vectors in shards of 375 with an uneven last one (370), a vector copied into
    every shard so that the top-k of a query is one tie straddling all eight ranks; every rank must return the single index's ids
    and scores (and the oracle's), for the three similarity functions, with and without the owners' exact rerank"""
    import build_mock
    build_mock.build()
    build_shim()
    id_path = str(tmp_path / "uid")
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_sharded_cabi as T; "
            "T._rank_main(int(sys.argv[1]), 8, %r, 1, N=2995, straddling_ties=True)") % (ROOT, os.path.join(ROOT, "tests"), id_path)
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
             for r in range(8)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"


# ---- GPU: real RCCL communicator (world size 1), several shards on the one device ----------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("n_local", [1, 2, 4, 8])
def test_sharded_cabi_rccl_world1_gpu(n_local):
    import jvector_amd as J
    from jvector_amd.sharded import Communicator
    ctx = J.HipContext(0)
    uid = Communicator.unique_id(ctx)                                        # dlopen(librccl) + ncclGetUniqueId
    comm = Communicator(ctx, 0, 1, uid)                                      # ncclCommInitRank, world 1
    run_sharded_equals_single(J, ctx, comm, n_local, N=40000, D=128, M=16, rerank_k=100)
    comm.close()
    ctx.close()
