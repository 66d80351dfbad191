"""End-to-end ingestion on the GPU: an OnDiskGraphIndex v6 file (+ PQVectors blob) written byte-for-byte as the
reference's writers do (oracle/jv_writers.py) -> jvector_amd.formats.load_index -> GraphSearcher; results must equal
the oracle's sequential search over the ORIGINAL in-memory arrays.  (File name sorts last on purpose.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import jvector_amd as J
import jvector_amd.formats as F
from jvector_amd import VectorSimilarityFunction as VSF
from oracle import jv_writers as W
from oracle import oracle as O
from test_graph_search import build_problem


@pytest.fixture(scope="module")
def ctx():
    c = J.HipContext(0)
    yield c
    c.close()


def _files(levels, separated, N=3000, D=64, M=8):
    v, lv, entry, entry_level, cb, q = build_problem(77 + levels, N=N, D=D, M=M, levels=levels)
    opq = O.OraclePQ(D, M, cb)
    codes = opq.encode_all(v)
    nb0 = [[int(x) for x in row if x >= 0] for row in lv[0][1]]
    upper = [(nbrs.shape[1], {int(n): [int(x) for x in r if x >= 0] for n, r in zip(ids, nbrs)}) for ids, nbrs in lv[1:]]
    odgi = W.write_odgi(6, D, nb0, lv[0][1].shape[1], entry, upper_levels=upper, vectors=v, separated=separated,
                        codes=codes, pq_block=opq.serialize(6))
    pqv = W.write_pqvectors(opq.serialize(6), codes)
    return v, lv, entry, entry_level, opq, codes, q, odgi, pqv


@pytest.mark.parametrize("levels,separated,with_pqv", [(2, False, True), (1, True, False), (3, False, False)])
def test_loaded_index_searches_like_the_oracle(ctx, levels, separated, with_pqv):
    v, lv, entry, entry_level, opq, codes, q, odgi, pqv = _files(levels, separated)
    idx = F.load_index(ctx, odgi, pqv if with_pqv else None)
    assert idx.host.entry_node == entry and idx.host.entry_level == entry_level
    if with_pqv:
        assert np.array_equal(idx.pq_vectors.get(0, len(v)), codes)
    s = idx.searcher(max_queries=64)
    og = O.OracleGraph(len(v), lv, entry, entry_level)
    for vsf in VSF:
        ids, sc, stats = s.search(q, vsf, 10, 40, return_stats=True)
        wi, ws, wst = og.search(opq, codes, v, q, int(vsf), 10, 40, fused=True)
        assert np.array_equal(stats, wst), vsf
        assert np.array_equal(ids, wi), vsf
        assert np.array_equal(sc, ws), vsf
