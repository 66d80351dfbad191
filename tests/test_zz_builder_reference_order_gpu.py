"""The builder in reference order on the MI355X: one-node batches == the oracle's one-thread GraphIndexBuilder, byte for byte (flat
graph with and without improve passes, every similarity function, tied scores; the layered build with max_batch = 1) — the checks of
tests/test_builder_reference_order.py at sizes the device is worth asking for; plus the batched build in reference order keeps the
structural contract and serves searches."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import oracle as O
from test_builder_reference_order import check_layered_reference_order, check_reference_order

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import jvector_amd as J
    c = J.HipContext(0)
    yield c
    c.close()


@pytest.mark.parametrize("vsf,dup,improve,wgx", [(O.COSINE, 0, 0, -1), (O.DOT_PRODUCT, 40, 0, -1), (O.EUCLIDEAN, 0, 1, 0), (O.COSINE, 30, 1, 1)])
def test_one_node_batches_equal_the_reference(ctx, vsf, dup, improve, wgx):
    import jvector_amd as J
    if wgx >= 0:
        ctx.set_option("gs_wgx", wgx)
    try:
        ctx.reset_stats()
        N = 2500 if improve == 0 else 1200
        check_reference_order(J, ctx, torch.device("cuda", 0), N, 128, 16, 16, 40, J.VectorSimilarityFunction(vsf), dup=dup, improve=improve)
        assert ctx.stat("gs_calls_host") == 0
    finally:
        if wgx >= 0:
            ctx.set_option("gs_wgx", -1)


def test_layered_build_with_one_node_batches(ctx):
    import jvector_amd as J
    st = check_layered_reference_order(J, ctx, torch.device("cuda", 0), 1500, 128, 16, 8, 24, J.VectorSimilarityFunction.COSINE, improve=1)
    print("layered, reference order, one-node batches:", dict(st))


def test_batched_build_in_reference_order_serves_searches(ctx):
    """batches of thousands (the concurrent case): degrees, packed rows, no self loops / duplicates, recall — and the same build twice
    is byte-identical"""
    import jvector_amd as J
    from jvector_amd.builder import build_vamana
    N, D, M = 30000, 128, 16
    rng = np.random.default_rng(3)
    centers = rng.standard_normal((40, D)).astype(np.float32)
    v = (centers[rng.integers(0, 40, N)] + 0.6 * rng.standard_normal((N, D))).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    q = (v[rng.integers(0, N, 64)] + 0.1 * rng.standard_normal((64, D))).astype(np.float32)
    dev = torch.device("cuda", 0)
    tv = torch.from_numpy(v).to(dev)
    VSF = J.VectorSimilarityFunction.COSINE
    pq = J.ProductQuantization.compute(ctx, tv, M, seed=2)
    vs = J.VectorSet(ctx, tv)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    ctx.set_option("bl_ref_order", 1)
    try:
        outs = []
        for _ in range(2):
            nb, entry, st = build_vamana(ctx, pq, cv, tv, VSF, max_degree=32, beam_width=100, alpha=1.2, max_batch=2048, overflow=1.2, improve=1)
            outs.append(nb.cpu().numpy().copy())
    finally:
        ctx.set_option("bl_ref_order", 0)
    nb = outs[0]
    assert np.array_equal(outs[0], outs[1])
    deg = (nb >= 0).sum(axis=1)
    assert nb.shape == (N, 32) and nb.max() < N and deg.max() <= 32 and deg.mean() > 8
    for i in range(0, N, 150):
        row = nb[i][nb[i] >= 0]
        assert i not in row and len(set(row.tolist())) == len(row) and (nb[i][:len(row)] >= 0).all()
    gt = np.argsort(-(q @ v.T), axis=1)[:, :10]

    def recall_of(rows, e, rk):
        graph = J.GraphIndex(ctx, N, [(None, rows)], e, 0)
        ids, _ = J.GraphSearcher(ctx, graph, pq, cv, None, vs, max_queries=64).search(q, VSF, 10, rk)
        return float(np.mean([len(set(a.tolist()) & set(b.tolist())) / 10.0 for a, b in zip(np.asarray(ids), gt)]))

    # against the default (symmetric re-scoring) build of the same data at the same rerankK: the reference's stored asymmetric scores
    # build the weaker graph (DESIGN.md §7's table: rerankK 150 against 74 at 10M) — stated and bounded here, not hidden
    nb_c, entry_c, _ = build_vamana(ctx, pq, cv, tv, VSF, max_degree=32, beam_width=100, alpha=1.2, max_batch=2048, overflow=1.2, improve=1)
    nb_c = nb_c.cpu().numpy().copy()
    r_ref, r_cls = {rk: recall_of(nb, entry, rk) for rk in (100, 400)}, {rk: recall_of(nb_c, entry_c, rk) for rk in (100, 400)}
    print("reference order, batched:", dict(st), "recall@10 by rerankK", r_ref, "| default build:", r_cls)
    assert r_ref[400] >= 0.5 and r_ref[400] >= 0.6 * r_cls[400] and r_ref[400] >= r_ref[100], (r_ref, r_cls)


@pytest.mark.parametrize("name", ["testDiversity", "testDiversity3d", "testDiversityFallback"])
def test_engine_reproduces_the_reference_literals(ctx, name):
    """the reference's own literal expectations for a one-thread GraphIndexBuilder (TestVectorGraph.java:455-613), on the device"""
    import jvector_amd as J
    from test_builder_reference_goldens import check_engine_reproduces_the_reference_literals
    ctx.reset_stats()
    check_engine_reproduces_the_reference_literals(J, ctx, torch.device("cuda", 0), name)
    assert ctx.stat("gs_calls_host") == 0


def test_engine_aknn_diverse(ctx):
    """TestVectorGraph.testAknnDiverse (:322-347) on the device: built in reference order (== the oracle's adjacency), searched by the engine"""
    import jvector_amd as J
    from test_builder_reference_goldens import check_engine_aknn_diverse
    check_engine_aknn_diverse(J, ctx, torch.device("cuda", 0))
