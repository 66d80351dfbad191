"""The reference's own literal expectations for a single-threaded GraphIndexBuilder — TestVectorGraph.testDiversity /
testDiversityFallback / testDiversity3d (jvector-tests/.../graph/TestVectorGraph.java:455-613: "carefully checked test cases", neighbour
sets asserted after every addGraphNode, with and without the hierarchy) and TestNodeArray's insertSorted / merge literals
(TestNodeArray.java:48-100) — replayed on (1) the oracle's restatement (jvo_builder_*) and (2) the engine's builder in reference order with
one node per batch, on the mock device here and on the MI355X in test_zz_builder_reference_order_gpu.py.

The reference tests score with the EXACT provider; the builder path here scores with PQ.  The bridge is a quantizer whose centroids ARE
the vectors (one sub-space over all D dimensions, cluster count = vector count): every ADC score and every code-to-code diversity score
is then the exact similarity, computed by the same sequential products and sums."""
import os
import platform
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "mock"))

from oracle import oracle as O
from test_builder_reference_order import _on_the_mock


def unit_vector_2d(pi_radians):
    """TestVectorGraph.unitVector2d :768-772"""
    return [np.float32(np.cos(np.pi * pi_radians)), np.float32(np.sin(np.pi * pi_radians))]


# (similarity, maxDegree, vectors, [(nodes added in this step, {node: expected neighbours after it})]) — GraphIndexBuilder(vectors, sf, M,
# beamWidth 10, neighborOverflow 1.0f, alpha 1.0f, addHierarchy)
CASES = {
    "testDiversity": (O.DOT_PRODUCT, 4, [unit_vector_2d(x) for x in (0.5, 0.75, 0.2, 0.9, 0.8, 0.77, 0.6)], [
        ([0, 1, 2], {0: [1, 2], 1: [0], 2: [0]}),
        ([3], {0: [1, 2], 1: [0, 3], 2: [0], 3: [1]}),
        ([4], {0: [1, 2], 1: [0, 3, 4], 2: [0], 3: [1, 4], 4: [1, 3]}),          # "4 is the same distance from 0 that 2 is; we leave the existing node in place"
        ([5], {0: [1, 2], 1: [0, 3, 4, 5], 2: [0], 3: [1, 4], 4: [1, 3, 5], 5: [1, 4]}),
    ]),
    "testDiversityFallback": (O.EUCLIDEAN, 2, [[0, 0, 0], [0, 10, 0], [0, 0, 20], [10, 0, 0], [0, 4, 0]], [
        ([0, 1, 2], {0: [1, 2], 1: [0], 2: [0]}),
        ([3], {0: [1, 3], 1: [0], 2: [0], 3: [0]}),                              # "2 has been displaced by 3"
    ]),
    "testDiversity3d": (O.EUCLIDEAN, 2, [[0, 0, 0], [0, 10, 0], [0, 0, 20], [0, 9, 0]], [
        ([0, 1, 2], {0: [1, 2], 1: [0], 2: [0]}),
        ([3], {0: [2, 3], 1: [0, 3], 2: [0], 3: [0, 1]}),                        # "1 has been displaced by 3"
    ]),
}


def test_rescore_literals_first_half():
    """GraphIndexBuilderTest.testRescore (:82-104), the part before the rescore: three 2-d vectors, EUCLIDEAN, maxDegree 2 — node 0's list
    is [1, 2] IN THAT ORDER with the scores 0.5 and 0.2 (the list carries its scores, sorted)"""
    v, opq, codes = _quantizer([[0, 0], [0, 1], [2, 0]])
    for hierarchy in (False, True):
        b = O.OracleBuilder(opq, codes, v, O.EUCLIDEAN, 2, 10, alpha=1.0, neighbor_overflow=1.0, add_hierarchy=hierarchy)
        for node in range(3):
            b.add(node)
        ids, sc, _ = b.row(0, 0)
        assert ids.tolist() == [1, 2] and abs(sc[0] - 0.5) < 1e-6 and abs(sc[1] - 0.2) < 1e-6


def _quantizer(vectors):
    v = np.asarray(vectors, np.float32)
    n, D = v.shape
    return v, O.OraclePQ(D, 1, v.reshape(-1).copy(), k=n), np.arange(n, dtype=np.uint8).reshape(n, 1)


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("hierarchy", [False, True])
def test_oracle_builder_reproduces_the_reference_literals(name, hierarchy):
    vsf, max_degree, vectors, steps = CASES[name]
    v, opq, codes = _quantizer(vectors)
    for i in range(len(v)):                                  # the bridge: every score of the PQ path is the exact similarity
        lut_scores = [O.lib().jvo_pq_direct_score(opq.ref, O._f(v[i]), int(vsf), O._u8(codes[j])) for j in range(len(v))]
        exact = [O.lib().jvo_compare(int(vsf), O._f(v[i]), O._f(v[j]), v.shape[1]) for j in range(len(v))]
        assert np.array_equal(np.float32(lut_scores).view(np.int32), np.float32(exact).view(np.int32))
    b = O.OracleBuilder(opq, codes, v, vsf, max_degree, 10, alpha=1.0, neighbor_overflow=1.0, add_hierarchy=hierarchy)
    for add, expected in steps:
        for node in add:
            b.add(node)
        for node, want in expected.items():
            ids, sc, _db = b.row(0, node)
            assert sorted(ids.tolist()) == want, (name, hierarchy, add, node, ids.tolist(), want)
            assert (np.diff(sc) <= 0).all()
    if hierarchy:                                            # (Random(0) with degree 4 puts node 1 on level 1: the second draw is 0.2405)
        assert b.info()["n_levels"] >= 2 and b.row(1, 1) is not None


def test_node_array_literals():
    """TestNodeArray.testScoresDescOrder :48-100: insertSorted goes BEHIND equal scores; replayed through the backlink's insert, which is
    where the builder uses it (a list receiving back edges with the scores 1, .8, .9, 1, 1.1, .8, .8, then .9)"""
    st = O.NodeArrayProbe()
    st.add_in_order(0, 1.0)
    st.add_in_order(1, 0.8)
    for node, score, want in ((3, 0.9, [0, 3, 1]), (4, 1.0, [0, 4, 3, 1]), (5, 1.1, [5, 0, 4, 3, 1]), (6, 0.8, [5, 0, 4, 3, 1, 6]),
                              (7, 0.8, [5, 0, 4, 3, 1, 6, 7])):
        st.insert_sorted(node, score)
        assert st.nodes() == want, (node, st.nodes(), want)
    assert st.insert_sorted(7, 0.8) == -1 and st.nodes() == [5, 0, 4, 3, 1, 6, 7]          # duplicateExistsNear: the same (node, score) again
    assert st.insert_sorted(7, 0.85) >= 0 and st.nodes() == [5, 0, 4, 3, 7, 1, 6, 7]       # another score: listed twice (NodeArray.java:212-228)


def _circular(n):
    """TestVectorGraph.CircularFloatVectorValues :735-761"""
    return np.array([unit_vector_2d(i / n) for i in range(n)], np.float32)


def _oracle_graph(b, n, max_degree):
    info = b.info()
    levels = [(None, b.rows(0, max_degree))]
    for l in range(1, info["n_levels"]):
        ids = np.array([i for i in range(n) if b.row(l, i) is not None], np.int32)
        rows = np.full((ids.size, max_degree), -1, np.int32)
        for r, i in enumerate(ids):
            got = b.row(l, int(i))[0]
            rows[r, :got.size] = got
        levels.append((ids, rows))
    return O.OracleGraph(n, levels, info["entry_node"], info["entry_level"])


@pytest.mark.parametrize("hierarchy", [False, True])
def test_aknn_diverse_and_accept_ords_properties(hierarchy):
    """TestVectorGraph.testAknnDiverse (:322-347) and testSearchWithAcceptOrds (:355-382): 100 vectors on the unit circle, DOT_PRODUCT,
    GraphIndexBuilder(M 20, beamWidth 100, overflow 1.0, alpha 1.4), built sequentially + cleanup(); the 10 results for (1, 0) must be
    (nearly) the ten lowest ids — sum < 75 — also when only the even ids are accepted: 10 results, all accepted, sum < 170"""
    n = 100
    v, opq, codes = _quantizer(_circular(n))
    b = O.OracleBuilder(opq, codes, v, O.DOT_PRODUCT, 20, 100, alpha=1.4, neighbor_overflow=1.0, add_hierarchy=hierarchy)
    for i in range(n):
        b.add(i)
    b.cleanup()
    g = _oracle_graph(b, n, 20)
    q = np.array([[1.0, 0.0]], np.float32)
    ids, _sc, _st = g.search(opq, codes, v, q, O.DOT_PRODUCT, 10, 10)
    assert (ids[0] >= 0).sum() == 10 and int(ids[0].sum()) < 75, ids
    accept = np.arange(n) % 2 == 0        # (the reference test draws a random acceptance set and bounds the sum by the accepted ids' own sum)
    ids, _sc, _st = g.search(opq, codes, v, q, O.DOT_PRODUCT, 10, 10, accept=accept)
    assert (ids[0] >= 0).sum() == 10 and accept[ids[0]].all() and int(ids[0].sum()) < 170, ids


def _array(pairs):
    a = O.NodeArrayProbe()
    for node, score in pairs:
        a.add_in_order(node, score)
    return a


def test_node_array_duplicates_and_merge_literals():
    """TestNodeArray.testNoDuplicatesDescOrder / testNoDuplicatesSameScores / testMergeCandidatesSimple (:163-228) literally, and
    testMergeCandidatesRandom's recipe and properties (:230-300) over 5 000 seeded cases"""
    for scores in ((10.0, 9.0, 8.0), (10.0, 10.0, 10.0)):
        a = O.NodeArrayProbe()
        for node, sc in zip((1, 2, 3), scores):
            a.insert_sorted(node, sc)
        assert a.insert_sorted(1, scores[0]) == -1 and a.insert_sorted(3, scores[2]) == -1
        assert a.nodes() == [1, 2, 3] and a.scores() == list(scores)
    assert _array([(1, 1.0)]).merge(_array([(0, 2.0)]))[0] == [0, 1]
    assert _array([(3, 3.0), (2, 2.0), (1, 1.0)]).merge(_array([(4, 4.0), (2, 2.0), (1, 1.0)])) == ([4, 3, 2, 1], [4.0, 3.0, 2.0, 1.0])
    assert _array([(3, 3.0), (2, 2.0)]).merge(_array([(2, 2.0)])) == ([3, 2], [3.0, 2.0])
    rng = np.random.default_rng(4)
    for _ in range(5000):
        max_size = 1 + int(rng.integers(0, 5))
        a1 = O.NodeArrayProbe()
        a1_size = max_size if rng.random() < 0.5 else 1 + int(rng.integers(0, max_size))
        for i in range(a1_size):
            a1.insert_sorted(i, float(np.float32(rng.random())))
        a2 = O.NodeArrayProbe()
        a2_size = max_size if rng.random() < 0.5 else 1 + int(rng.integers(0, max_size))
        for i in range(a2_size):
            if i < a1_size and rng.random() < 0.5:
                j = int(rng.integers(0, a1_size))
                if a1.nodes()[j] not in a2.nodes():
                    a2.insert_sorted(a1.nodes()[j], a1.scores()[j])
            else:
                score = float(np.float32(rng.random())) if rng.random() < 0.5 else a1.scores()[int(rng.integers(0, a1_size))]
                a2.insert_sorted(i + a1_size, score)
        nodes, scores = a1.merge(a2)
        assert max(len(a1.nodes()), len(a2.nodes())) <= len(nodes) <= len(a1.nodes()) + len(a2.nodes())
        assert all(x >= y for x, y in zip(scores, scores[1:])) and len(set(nodes)) == len(nodes)
        assert set(a1.nodes()) | set(a2.nodes()) == set(nodes)


def check_engine_reproduces_the_reference_literals(J, ctx, dev, name):
    from jvector_amd.builder import GraphBuilder
    vsf, max_degree, vectors, steps = CASES[name]
    v, opq, codes = _quantizer(vectors)
    n, D = v.shape
    pq = J.ProductQuantization.from_codebooks(ctx, D, 1, v.reshape(-1).copy(), None, cluster_count=n)
    tv = torch.from_numpy(v).to(dev)
    vs = J.VectorSet(ctx, tv)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    assert np.array_equal(np.asarray(cv.get(0, n)), codes)
    ob = O.OracleBuilder(opq, codes, v, vsf, max_degree, 10, alpha=1.0, neighbor_overflow=1.0, add_hierarchy=False)
    ctx.set_option("bl_ref_order", 1)
    try:
        gb = GraphBuilder(ctx, pq, cv, vs, J.VectorSimilarityFunction(int(vsf)), max_degree, 10, 1.0, 1.0)
        first = True
        for add, expected in steps:
            for node in add:
                ob.add(node)
                if first:
                    gb.seed(node)
                    first = False
                else:
                    gb.insert_batch(np.array([node], np.int32))
            ids, sc, db = gb.working_rows()
            for node, want in expected.items():
                row = ids[node][ids[node] >= 0]
                assert sorted(row.tolist()) == want, (name, add, node, row.tolist(), want)
                oi, osc, odb = ob.row(0, node)
                assert np.array_equal(row, oi) and np.array_equal(sc[node, :len(row)].view(np.int32), osc.view(np.int32)) and int(db[node]) == odb
        gb.close()
    finally:
        ctx.set_option("bl_ref_order", 0)


def check_engine_aknn_diverse(J, ctx, dev):
    """testAknnDiverse on the engine: built in reference order, one node per batch, finish() = cleanup()'s enforceDegree — the adjacency is
    the oracle's, and the engine's own search over it returns the ten lowest ids"""
    from jvector_amd.builder import GraphBuilder
    n = 100
    v, opq, codes = _quantizer(_circular(n))
    pq = J.ProductQuantization.from_codebooks(ctx, 2, 1, v.reshape(-1).copy(), None, cluster_count=n)
    tv = torch.from_numpy(v).to(dev)
    vs = J.VectorSet(ctx, tv)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    assert np.array_equal(np.asarray(cv.get(0, n)), codes)
    VSF = J.VectorSimilarityFunction.DOT_PRODUCT
    ob = O.OracleBuilder(opq, codes, v, O.DOT_PRODUCT, 20, 100, alpha=1.4, neighbor_overflow=1.0)
    ctx.set_option("bl_ref_order", 1)
    try:
        gb = GraphBuilder(ctx, pq, cv, vs, VSF, 20, 100, 1.4, 1.0)
        gb.seed(0)
        ob.add(0)
        for i in range(1, n):
            gb.insert_batch(np.array([i], np.int32))
            ob.add(i)
        rows = gb.finish(torch.empty((n, 20), dtype=torch.int32, device=dev)).cpu().numpy().copy()
        gb.close()
    finally:
        ctx.set_option("bl_ref_order", 0)
    ob.cleanup()
    assert np.array_equal(rows, ob.rows(0, 20))
    graph = J.GraphIndex(ctx, n, [(None, rows)], 0, 0)
    ids, _ = J.GraphSearcher(ctx, graph, pq, cv, None, vs, max_queries=4).search(np.array([[1.0, 0.0]], np.float32), VSF, 10, 10)
    ids = np.asarray(ids)[0]
    assert (ids >= 0).sum() == 10 and int(ids.sum()) < 75, ids


@pytest.mark.skipif(platform.machine() != "x86_64", reason="the mock build needs the x86-64 lane emulator")
def test_engine_aknn_diverse_on_the_mock():
    _on_the_mock(lambda J, ctx: check_engine_aknn_diverse(J, ctx, torch.device("cpu")))


@pytest.mark.skipif(platform.machine() != "x86_64", reason="the mock build needs the x86-64 lane emulator")
@pytest.mark.parametrize("name", sorted(CASES))
def test_engine_reproduces_the_reference_literals_on_the_mock(name):
    _on_the_mock(lambda J, ctx: check_engine_reproduces_the_reference_literals(J, ctx, torch.device("cpu"), name))
