"""Sharded GRAPH search on the GPU (one segment index per shard, several shards resident on the one available GPU):
HipGraphShardBackend + the same all-gather / merge / owner-rerank as the flat form == manual merge of the oracle's per-shard
graph searches, bit-identical ids and scores."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oracle as O
from test_graph_search import fused_blocks
from test_sharded import _graph_shards, _manual_merge


def test_sharded_graph_hip_equals_manual_merge():
    run_sharded_graph_case(lambda t: t.cuda())


def run_sharded_graph_case(to_device):
    """to_device: where the query tensor lives (cuda on hardware; the CPU mock-device test passes the identity)"""
    import jvector_amd as J
    from jvector_amd.sharded import HipGraphShardBackend, ShardedSearcher
    shards, cb, q = _graph_shards(70, 3, N=3000)
    D, M = 64, 8
    opq = O.OraclePQ(D, M, cb)
    ctx = J.HipContext(0)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    backends, oracle_shards, allv, keep = [], [], {}, []
    for v, lv, entry, entry_level, lo in shards:
        vs = J.VectorSet(ctx, v)
        cv = J.PQVectors.encode_and_build(ctx, pq, vs)
        codes = cv.get(0, len(v))
        graph = J.GraphIndex(ctx, len(v), lv, entry, entry_level)
        fused = J.FusedPQ(ctx, pq, fused_blocks(codes, lv[0][1]), lv[0][1])
        backends.append(HipGraphShardBackend(ctx, graph, pq, cv, fused, vs, lo, max_queries=16))
        oracle_shards.append((O.OracleGraph(len(v), lv, entry, entry_level), codes, lo))
        keep.append((vs, cv, graph, fused))
        for i in range(len(v)):
            allv[lo + i] = v[i]
    s = ShardedSearcher(backends)
    tq = to_device(torch.from_numpy(q))
    for vsf in J.VectorSimilarityFunction:
        per = []
        for og, codes, lo in oracle_shards:
            ids, sc, _ = og.search(opq, codes, None, q, int(vsf), 40, 40, fused=True)
            per.append((np.where(ids >= 0, ids + lo, ids), sc))
        wi, ws = _manual_merge(opq, per, allv, q, vsf, 10, 40)
        gi, gs = s.search(tq, vsf, 10, 40)
        ctx.sync()
        assert np.array_equal(gi.cpu().numpy(), wi) and np.array_equal(gs.cpu().numpy(), ws), vsf
