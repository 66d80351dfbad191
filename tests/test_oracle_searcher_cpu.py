"""The oracle's GraphSearcher OBJECT (jvo_searcher_*: threshold > 0, rerankFloor, resume(), rerankedCount,
worstApproximateInTopK) pinned the way the reference pins the same options — its own tests hold no literals for them, only
properties:
  TestVectorGraph.testResume :133-172, testRerankCaching :181-212 (jvector-tests/.../graph/TestVectorGraph.java),
  Test2DThreshold.testThreshold2D :48-92 (visited ratio / recall bounds), TestPruningCompatibility.assertAllAtOrAboveThreshold.
plus: agreement with the one-shot jvo_graph_search_filtered where the two overlap, the heap-array-order rerank of
NodeQueue.rerank (NodeQueue.java:197-214) on engineered exact-score ties, and commons-math's LEGACY percentile on the
examples its documentation works through."""
import numpy as np
import pytest

from oracle import oracle as O

L2, DOT, COS = 0, 1, 2   # oracle vsf codes (jv_oracle.h: JVO_EUCLIDEAN, JVO_DOT_PRODUCT, JVO_COSINE)


def knn_graph(v, deg, vsf, rng, long_edges=2):
    N = len(v)
    if vsf == L2:
        d2 = ((v[:, None, :] - v[None, :, :]) ** 2).sum(-1)
        sims = -d2
    else:
        sims = v @ v.T
    np.fill_diagonal(sims, -np.inf)
    order = np.argsort(-sims, axis=1)
    nb = np.full((N, deg), -1, np.int32)
    for i in range(N):
        row = list(order[i, : deg - long_edges]) + list(rng.choice(N, long_edges, replace=False))
        row = [x for j, x in enumerate(row) if x != i and x not in row[:j]]
        nb[i, : len(row)] = row
    return nb


def problem(seed, N=1000, D=2, M=2, deg=20, vsf=L2, hierarchy=False):
    rng = np.random.default_rng(seed)
    v = rng.random((N, D)).astype(np.float32)
    nb = knn_graph(v, deg, vsf, rng)
    levels = [(None, nb)]
    entry, entry_level = int(rng.integers(0, N)), 0
    if hierarchy:
        top = np.sort(rng.choice(N, 40, replace=False)).astype(np.int32)
        nb2 = top[knn_graph(v[top], 6, vsf, rng, long_edges=1).clip(0)]
        levels.append((top, nb2.astype(np.int32)))
        entry, entry_level = int(top[0]), 1
    sizes, offs = O.subvector_sizes_offsets(D, M)
    pick = rng.choice(N, 256, replace=False)
    cb = np.concatenate([v[pick, offs[m]: offs[m] + sizes[m]].reshape(-1) for m in range(M)])
    pq = O.OraclePQ(D, M, cb)
    codes = pq.encode_all(v)
    g = O.OracleGraph(N, levels, entry, entry_level)
    return rng, v, g, pq, codes


def test_percentile_legacy_documented_examples():
    # Percentile's class documentation (commons-math3 3.6.1): pos = p (n + 1) / 100, linear interpolation between the
    # neighbours of pos in the sorted array; the {1, 2, 3, 4} walk-through gives 1.5 / 1.25 / 3.75 / 2.5
    d = [1.0, 3.0, 2.0, 4.0]
    assert O.percentile_legacy(d, 30) == 1.5
    assert O.percentile_legacy(d, 25) == 1.25
    assert O.percentile_legacy(d, 75) == 3.75
    assert O.percentile_legacy(d, 50) == 2.5
    assert O.percentile_legacy(d, 1) == 1.0       # pos < 1 -> smallest
    assert O.percentile_legacy(d, 100) == 4.0     # pos >= n -> largest
    assert O.percentile_legacy([7.0], 99) == 7.0
    # the tracker's case: n = 500, p = 99 -> pos = 495.99: between the 495th and 496th order statistics
    x = np.arange(500, dtype=np.float64)
    lo, hi = 494.0, 495.0
    dif = 0.99 * 501 - 495.0
    assert O.percentile_legacy(x[::-1].copy(), 99) == lo + dif * (hi - lo)


@pytest.mark.parametrize("hierarchy", [False, True])
@pytest.mark.parametrize("fused", [False, True])
def test_searcher_equals_one_shot_search(hierarchy, fused):
    rng, v, g, pq, codes = problem(3, hierarchy=hierarchy)
    q = rng.random((12, 2)).astype(np.float32)
    accept = rng.random(len(v)) < 0.6
    for acc in (None, accept):
        ids, sc, st = g.search(pq, codes, v, q, L2, 10, 30, fused=fused, accept=acc)
        s = g.searcher(pq, codes, v, L2, fused=fused)
        for i in range(len(q)):
            r = s.search(q[i], 10, 30, accept=acc)
            assert np.array_equal(r.ids, ids[i][: len(r)]) and np.array_equal(r.scores, sc[i][: len(r)])
            assert (ids[i][len(r):] == -1).all()
            assert (r.visited, r.expanded) == tuple(st[i])
            assert r.expanded_base <= r.expanded
            assert r.reranked == min(30, r.reranked)   # every approximate result scored once, none cached yet
        s.close()


@pytest.mark.parametrize("hierarchy", [False, True])
def test_resume_finds_what_a_bigger_search_finds(hierarchy):
    # TestVectorGraph.testResume: search(10) + resume(15) ~ search(25); resumed results are new nodes
    hits = total = 0
    for seed in range(6):
        rng, v, g, pq, codes = problem(10 + seed, hierarchy=hierarchy)
        s = g.searcher(pq, codes, v, L2)
        acc = None if seed % 2 else (rng.random(len(v)) < 0.7)
        q = rng.random(2).astype(np.float32)
        a = s.search(q, 10, 10, accept=acc)
        b = s.resume(15, 15)
        assert len(a) == 10 and len(b) == 15
        assert not set(a.ids) & set(b.ids)
        if acc is not None:
            assert acc[a.ids].all() and acc[b.ids].all()
        s2 = g.searcher(pq, codes, v, L2)
        e = s2.search(q, 25, 25, accept=acc)
        assert len(e) == 25
        assert e.visited * 1.1 > a.visited + b.visited * 0 and a.visited + b.visited <= e.visited * 1.6
        hits += len(set(e.ids) & (set(a.ids) | set(b.ids)))
        total += 25
    assert hits >= 0.9 * total


def test_rerank_caching_and_counts():
    # TestVectorGraph.testRerankCaching: the first search reranks rerankK nodes, a resume fewer (cached exact scores)
    rng, v, g, pq, codes = problem(5, hierarchy=True)
    s = g.searcher(pq, codes, v, L2)
    q = rng.random(2).astype(np.float32)
    a = s.search(q, 10, 30)
    assert len(a) == 10 and a.reranked == 30
    # worstApproximateInTopK = the smallest approximate score among the returned nodes (NodeQueue.java:216-228)
    sc = pq.adc_scores(q, L2, codes, a.ids.astype(np.int32))
    assert a.worst_approximate_in_topk == sc.min()
    b = s.resume(10, 30)
    assert len(b) == 10 and b.reranked < 30
    assert not set(a.ids) & set(b.ids)
    # without a reranker: approximate results, nothing reranked, worst = +inf (:478-487)
    s3 = g.searcher(pq, codes, None, L2)
    c = s3.search(q, 10, 30)
    assert c.reranked == 0 and c.worst_approximate_in_topk == np.inf and len(c) == 10
    d = s3.resume(20, 20)
    assert len(d) == 20 and not set(c.ids) & set(d.ids)
    # the 20 nodes popped off approximateResults on the way down to topK come back first on resume
    ids1, _, _ = g.search(pq, codes, None, q[None], L2, 30, 30)
    assert set(ids1[0]) == set(c.ids) | set(d.ids)


def test_rerank_floor():
    rng, v, g, pq, codes = problem(6)
    s = g.searcher(pq, codes, v, L2)
    q = rng.random(2).astype(np.float32)
    full = s.search(q, 10, 40)
    approx = np.sort(pq.adc_scores(q, L2, codes, None))[::-1]
    floor = float(approx[14])                      # 15 nodes of the whole set reach the floor
    r = s.search(q, 10, 40, rerank_floor=floor)
    assert 1 <= r.reranked <= 15 and r.reranked < full.reranked
    sc = pq.adc_scores(q, L2, codes, r.ids.astype(np.int32))
    assert (sc >= floor).all()
    assert r.worst_approximate_in_topk >= floor if len(r) == 10 else r.worst_approximate_in_topk == np.inf
    # a floor nothing reaches: the best approximate result alone is reranked and returned (:186-191)
    r1 = s.search(q, 10, 40, rerank_floor=2.0)
    assert len(r1) == 1 and r1.reranked == 1 and r1.worst_approximate_in_topk == np.inf
    best = int(np.argmax(pq.adc_scores(q, L2, codes, full_ids := np.arange(len(v), dtype=np.int32))))
    assert r1.ids[0] == best or pq.adc_scores(q, L2, codes, r1.ids.astype(np.int32))[0] == approx[0]
    # the discarded ones resurface on resume
    r2 = s.resume(10, 40)
    assert len(r2) == 10 and r1.ids[0] not in r2.ids


def test_threshold_2d():
    # Test2DThreshold: EUCLIDEAN similarity >= th, topK = all nodes: few nodes visited, most matches found, none below th
    rng = np.random.default_rng(42)
    N = 6000
    v = (rng.random((N, 2)) * rng.choice([-1.0, 1.0], (N, 2))).astype(np.float32)   # TestUtil.randomVector: signed, l2-normalised
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    nb = knn_graph(v, 24, L2, rng, long_edges=2)
    g = O.OracleGraph(N, [(None, nb)], 0, 0)
    sizes, offs = O.subvector_sizes_offsets(2, 2)
    cb = np.concatenate([np.linspace(-1, 1, 256, dtype=np.float32) for _ in range(2)])   # 1-D grid per coordinate: near-exact PQ
    pq = O.OraclePQ(2, 2, cb)
    codes = pq.encode_all(v)
    s = g.searcher(pq, codes, v, L2)
    visited_ratio = recall = 0.0
    nq = 30
    for _ in range(nq):
        q = (rng.random(2) * rng.choice([-1.0, 1.0], 2)).astype(np.float32)
        q /= np.linalg.norm(q)
        th = float(0.3 + 0.45 * rng.random())
        exact = 1.0 / (1.0 + ((v - q) ** 2).sum(1))
        want = int((exact >= th).sum())
        r = s.search(q, N, N, threshold=th)
        sc = pq.adc_scores(q, L2, codes, r.ids.astype(np.int32))
        assert (sc >= th).all()                      # approximate score gates the results (:437)
        visited_ratio += r.visited / N / nq
        recall += (len(r) / max(want, 1)) / nq
    assert visited_ratio < 0.8, visited_ratio      # the reference's bounds (Test2DThreshold.java:46-47)
    assert recall > 0.9, recall
    # threshold = 0 visits everything reachable with topK = N
    r0 = s.search(rng.random(2).astype(np.float32), N, N)
    assert r0.visited > 0.95 * N


def test_exact_tie_membership_follows_heap_array_order():
    # two nodes with the SAME vector (identical exact score) straddle the K-th place; which one is kept is decided by the
    # order approximateResults' heap array is walked in (NodeQueue.java:197-214), not by node id
    found_low = found_high = 0
    for seed in range(40):
        rng, v, g, pq, codes = problem(100 + seed, N=300, deg=12)
        q = rng.random(2).astype(np.float32)
        ex = 1.0 / (1.0 + ((v - q) ** 2).sum(1))
        order = np.argsort(-ex)
        a, b = int(order[4]), int(order[5])           # 5th / 6th best: make them identical vectors
        v[b] = v[a]
        ids, sc, _ = g.search(pq, codes, v, q[None], L2, 5, 30)   # codes (hence traversal) unchanged: only exact scores tie
        s = g.searcher(pq, codes, v, L2)
        r = s.search(q, 5, 30)
        assert np.array_equal(r.ids, ids[0][: len(r)])
        got = set(ids[0])
        if a in got and b in got or (a not in got and b not in got):
            continue
        # replay NodeQueue.rerank by hand from the approximate result heap order
        exact = np.array([O.compare(L2, q, v[i]) for i in range(len(v))], np.float32)
        assert exact[a] == exact[b]
        if (a in got) == (a < b):
            found_low += 1
        else:
            found_high += 1
    assert found_low + found_high >= 5
    assert found_high >= 1, "every tie went to the lower id: the heap-order walk is not being exercised"


def test_illegal_arguments():
    rng, v, g, pq, codes = problem(1, N=300, deg=8)
    s = g.searcher(pq, codes, v, L2)
    with pytest.raises(ValueError):
        s.resume(5, 5)
    with pytest.raises(ValueError):
        s.search(v[0], 10, 5)
