"""Batched graph searcher against the oracle's sequential GraphSearcher restatement: same graph,
same queries -> identical result ids, scores, visitedCount and expandedCount per query.  The host traversal
(graph_search.cpp + frontier kernels) is pinned explicitly here; the device-resident traversal (the default wherever
it applies) has its own file, tests/test_zz_device_traversal_gpu.py; the BASELINE shapes (C3: D=768 / M=96 /
maxDegree 32, C5: D=1536 / M=192) run through BOTH at the bottom of this file."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import jvector_amd as J
from jvector_amd import VectorSimilarityFunction as VSF
from oracle import oracle as O


def build_problem(seed, N=6000, D=64, M=8, deg=16, top_n=60, top_deg=8, levels=2):
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((25, D)).astype(np.float32)
    v = (centers[rng.integers(0, 25, N)] + 0.5 * rng.standard_normal((N, D))).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    sims = v @ v.T
    np.fill_diagonal(sims, -9)
    order = np.argsort(-sims, axis=1)
    nb = np.full((N, deg), -1, np.int32)
    for i in range(N):                      # ragged degrees, a few long-range edges
        d = int(rng.integers(deg // 2, deg + 1))
        row = list(order[i, : d - 2]) + list(rng.choice(N, 2, replace=False))
        row = [x for j, x in enumerate(row) if x != i and x not in row[:j]]
        nb[i, : len(row)] = row
    lv = [(None, nb)]
    entry, entry_level = 0, 0
    if levels > 1:
        top = np.sort(rng.choice(N, top_n, replace=False)).astype(np.int32)
        s2 = v[top] @ v[top].T
        np.fill_diagonal(s2, -9)
        nb2 = top[np.argsort(-s2, axis=1)[:, :top_deg]].astype(np.int32)
        lv.append((top, nb2))
        entry, entry_level = int(top[3]), 1
    if levels > 2:
        top3 = np.sort(rng.choice(lv[1][0], 10, replace=False)).astype(np.int32)
        s3 = v[top3] @ v[top3].T
        np.fill_diagonal(s3, -9)
        nb3 = top3[np.argsort(-s3, axis=1)[:, :4]].astype(np.int32)
        lv.append((top3, nb3))
        entry, entry_level = int(top3[1]), 2
    sizes, offs = O.subvector_sizes_offsets(D, M)
    pick = rng.choice(N, 256, replace=False)
    cb = np.concatenate([v[pick, offs[m]: offs[m] + sizes[m]].reshape(-1) for m in range(M)])
    q = (v[rng.integers(0, N, 40)] + 0.05 * rng.standard_normal((40, D))).astype(np.float32)
    return v, lv, entry, entry_level, cb, q


def fused_blocks(codes, nb):
    N, deg = nb.shape
    M = codes.shape[1]
    blocks = np.zeros((N, deg * M), np.uint8)
    for n in range(N):
        d = int((nb[n] >= 0).sum())
        blocks[n, : d * M] = codes[nb[n, :d]].reshape(-1)
    return blocks


@pytest.fixture(scope="module")
def ctx():
    c = J.HipContext(0)
    yield c
    c.close()


@pytest.mark.parametrize("levels,use_fused,D,M", [(1, False, 64, 8), (1, True, 64, 8), (2, True, 64, 8),
                                                  (3, False, 64, 8), (2, True, 128, 16), (2, False, 96, 12)])
def test_graph_search_matches_oracle(ctx, levels, use_fused, D, M):
    v, lv, entry, entry_level, cb, q = build_problem(levels * 10 + D, D=D, M=M, levels=levels)
    N = v.shape[0]
    opq = O.OraclePQ(D, M, cb)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    vs = J.VectorSet(ctx, v)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    codes = cv.get(0, N)
    assert np.array_equal(codes, opq.encode_all(v))
    og = O.OracleGraph(N, lv, entry, entry_level)
    graph = J.GraphIndex(ctx, N, lv, entry, entry_level).set_traversal("host")
    fused = J.FusedPQ(ctx, pq, fused_blocks(codes, lv[0][1]), lv[0][1]) if use_fused else None
    for vsf in VSF:
        for rerank, top_k, rk in ((True, 10, 40), (False, 5, 20), (True, 1, 1)):
            s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs if rerank else None, max_queries=64)
            ids, sc, stats = s.search(q, vsf, top_k, rk, return_stats=True)
            wi, ws, wst = og.search(opq, codes, v if rerank else None, q, int(vsf), top_k, rk, fused=use_fused)
            assert np.array_equal(stats, wst), (vsf, rerank)           # same visited / expanded counts
            assert np.array_equal(ids, wi), (vsf, rerank, top_k)
            assert np.array_equal(sc, ws), (vsf, rerank, top_k)


def test_graph_search_large_batch_and_errors(ctx):
    v, lv, entry, entry_level, cb, q = build_problem(3, N=3000, levels=2)
    N, D, M = v.shape[0], 64, 8
    rng = np.random.default_rng(9)
    q = (v[rng.integers(0, N, 700)] + 0.1 * rng.standard_normal((700, D))).astype(np.float32)
    opq = O.OraclePQ(D, M, cb)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    vs = J.VectorSet(ctx, v)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    codes = cv.get(0, N)
    graph = J.GraphIndex(ctx, N, lv, entry, entry_level).set_traversal("host")
    fused = J.FusedPQ(ctx, pq, fused_blocks(codes, lv[0][1]), lv[0][1])
    s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=1024)
    ids, sc, stats = s.search(q, VSF.COSINE, 10, 60, return_stats=True)
    wi, ws, wst = O.OracleGraph(N, lv, entry, entry_level).search(opq, codes, v, q, O.COSINE, 10, 60, fused=True)
    assert np.array_equal(ids, wi) and np.array_equal(sc, ws) and np.array_equal(stats, wst)
    with pytest.raises(ValueError):  # rerankK < topK (GraphSearcher.java:233)
        s.search(q[:4], VSF.COSINE, 10, 5)
    with pytest.raises(ValueError):  # more queries than the LUT capacity
        J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=2).search(q[:4], VSF.COSINE, 1, 1)


@pytest.mark.parametrize("slots,groups", [(16, 1), (64, 2), (500, 3)])
def test_graph_search_continuous_batching(ctx, slots, groups, monkeypatch):
    """queries streaming through a few traversal slots (continuous batching) and alternating slot groups
    (host/GPU pipelining) must not change any per-query result"""
    v, lv, entry, entry_level, cb, q = build_problem(5, N=3000, levels=3)
    N, D, M = v.shape[0], 64, 8
    rng = np.random.default_rng(4)
    q = (v[rng.integers(0, N, 333)] + 0.1 * rng.standard_normal((333, D))).astype(np.float32)
    opq = O.OraclePQ(D, M, cb)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    vs = J.VectorSet(ctx, v)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    codes = cv.get(0, N)
    graph = J.GraphIndex(ctx, N, lv, entry, entry_level).set_traversal("host")
    og = O.OracleGraph(N, lv, entry, entry_level)
    monkeypatch.setenv("JVECTOR_HIP_GRAPH_SLOTS", str(slots))
    monkeypatch.setenv("JVECTOR_HIP_GRAPH_GROUPS", str(groups))
    for use_fused in (True, False):
        fused = J.FusedPQ(ctx, pq, fused_blocks(codes, lv[0][1]), lv[0][1]) if use_fused else None
        s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=512)
        ids, sc, stats = s.search(q, VSF.DOT_PRODUCT, 10, 30, return_stats=True)
        wi, ws, wst = og.search(opq, codes, v, q, O.DOT_PRODUCT, 10, 30, fused=use_fused)
        assert np.array_equal(stats, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws)


def test_graph_search_accept_ords(ctx):
    """GraphSearcher.search(..., acceptOrds): filtered-out nodes are traversed but never returned; one mask for the batch
    and one mask per query; identical to the oracle."""
    v, lv, entry, entry_level, cb, q = build_problem(19, N=4000, D=128, M=16, levels=2)
    N = v.shape[0]
    opq = O.OraclePQ(128, 16, cb)
    pq = J.ProductQuantization.from_codebooks(ctx, 128, 16, cb)
    vs = J.VectorSet(ctx, v)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    codes = cv.get(0, N)
    graph = J.GraphIndex(ctx, N, lv, entry, entry_level).set_traversal("host")
    fused = J.FusedPQ(ctx, pq, fused_blocks(codes, lv[0][1]), lv[0][1])
    og = O.OracleGraph(N, lv, entry, entry_level)
    s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=64)
    rng = np.random.default_rng(2)
    shared = rng.random(N) < 0.3
    per_query = rng.random((len(q), N)) < 0.2
    per_query[3] = False
    for accept in (shared, per_query):
        for vsf in VSF:
            ids, sc, st = s.search(q, vsf, 10, 40, return_stats=True, accept=accept)
            wi, ws, wst = og.search(opq, codes, v, q, int(vsf), 10, 40, fused=True, accept=accept)
            assert np.array_equal(st, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws), vsf
    assert (ids[3] == -1).all()


def run_ties_cases(J, ctx, traversal, cases=6):
    """Every base vector stored two or three times under different ordinals, coordinates on a coarse grid, queries that ARE
    base vectors: ADC scores and exact scores tie constantly, so every queue decision falls to the NodeQueue order
    (equal score -> smaller node id first, NodeQueue.java:125-129) — in the candidate queue, the bounded result heap,
    the rerank and the final top-k, with and without an acceptOrds filter.  Shared with tests/test_mock_device.py."""
    VSF = J.VectorSimilarityFunction
    for case in range(cases):
        rng = np.random.default_rng(1000 + case)
        D, M = [(128, 16), (256, 32)][case % 2]
        base = rng.standard_normal((int(rng.integers(150, 500)), D)).astype(np.float32)
        if case % 3 == 0:
            base = np.round(base * 2) / 2
        v = np.repeat(base, int(rng.integers(2, 4)), axis=0)
        v = v[rng.permutation(len(v))]
        N, deg = len(v), int(rng.choice([8, 16, 32]))
        nb = np.full((N, deg), -1, np.int32)
        for i in range(N):
            row = rng.choice(N, int(rng.integers(1, deg + 1)), replace=False)
            row = row[row != i]
            nb[i, :len(row)] = row
        lv, entry, entry_level = [(None, nb)], int(rng.integers(0, N)), 0
        if case % 2:
            top = np.sort(rng.choice(N, 20, replace=False)).astype(np.int32)
            nb2 = np.full((20, 4), -1, np.int32)
            for i in range(20):
                r = rng.choice(top, 3, replace=False)
                r = r[r != top[i]]
                nb2[i, :len(r)] = r
            lv.append((top, nb2))
            entry, entry_level = int(top[0]), 1
        sizes, offs = O.subvector_sizes_offsets(D, M)
        pick = rng.choice(N, 256, replace=True)               # duplicate centroids too: encode ties -> first index wins
        cb = np.concatenate([v[pick, offs[m]: offs[m] + sizes[m]].reshape(-1) for m in range(M)])
        q = np.concatenate([v[rng.integers(0, N, 10)], rng.standard_normal((4, D)).astype(np.float32)]).astype(np.float32)
        opq = O.OraclePQ(D, M, cb)
        pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
        vs = J.VectorSet(ctx, v)
        cv = J.PQVectors.encode_and_build(ctx, pq, vs)
        codes = cv.get(0, N)
        assert np.array_equal(codes, opq.encode_all(v))
        og = O.OracleGraph(N, lv, entry, entry_level)
        graph = J.GraphIndex(ctx, N, lv, entry, entry_level).set_traversal(traversal)
        use_fused = case % 2 == 0
        fused = J.FusedPQ(ctx, pq, fused_blocks(codes, nb), nb) if use_fused else None
        accept = None if case % 3 else (rng.random(N) < 0.6)
        for vsf in VSF:
            if vsf == VSF.COSINE and case % 3 == 0:
                continue                                        # the grid holds zero vectors (NaN cosine; covered elsewhere)
            for rerank, top_k, rk in ((True, 10, 30), (False, 7, 7)):
                s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs if rerank else None, max_queries=16)
                ids, sc, st = s.search(q, vsf, top_k, rk, return_stats=True, accept=accept)
                wi, ws, wst = og.search(opq, codes, v if rerank else None, q, int(vsf), top_k, rk, fused=use_fused,
                                        accept=accept)
                tag = (case, N, deg, vsf, rerank)
                assert np.array_equal(st, wst), tag
                assert np.array_equal(ids, wi) and np.array_equal(sc, ws), tag
                if rerank and accept is None:                   # the ties are really there: a query that is a base vector
                    assert (np.diff(sc[:10], axis=1) == 0).any(), tag  # scores its copies identically


def test_graph_search_engineered_ties(ctx):
    run_ties_cases(J, ctx, "host")


# ---- BASELINE shapes: C3 = 10M x 768 cosine, PQ-96 + FusedADC, maxDegree 32 (the benched kernel instances:
#      frontier_direct_kernel<.,CH16=6,TWO> / graph_search_kernel<.,6,.,PAIR>), C5 = 1536-d / PQ-192 (CH16=12) --------
@pytest.fixture(scope="module")
def baseline_problems():
    cache = {}

    def get(D, M, levels):
        key = (D, M, levels)
        if key not in cache:
            N = 5000 if D <= 768 else 3000
            cache[key] = build_problem(7 * levels + M, N=N, D=D, M=M, deg=32, top_n=80, top_deg=16, levels=levels)
        return cache[key]
    return get


@pytest.mark.parametrize("traversal", ["host", "device", "auto"])
@pytest.mark.parametrize("D,M,levels", [(768, 96, 2), (768, 96, 3), (1536, 192, 2)])
def test_graph_search_baseline_shapes(ctx, baseline_problems, traversal, D, M, levels):
    """ids, scores and visited / expanded counters == the oracle at the benched shapes, for both traversals, fused
    (FusedPQDecoder.similarityToNeighbor) and unfused (PQDecoder.similarityTo), all three similarity functions."""
    v, lv, entry, entry_level, cb, q = baseline_problems(D, M, levels)
    N = v.shape[0]
    assert lv[0][1].shape[1] == 32 and (lv[0][1] >= 0).sum(axis=1).max() == 32      # full-degree rows exist
    opq = O.OraclePQ(D, M, cb)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    vs = J.VectorSet(ctx, v)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    codes = cv.get(0, N)
    assert np.array_equal(codes, opq.encode_all(v))
    og = O.OracleGraph(N, lv, entry, entry_level)
    graph = J.GraphIndex(ctx, N, lv, entry, entry_level).set_traversal(traversal)
    for use_fused in (True, False):
        fused = J.FusedPQ(ctx, pq, fused_blocks(codes, lv[0][1]), lv[0][1]) if use_fused else None
        for vsf in VSF:
            for rerank, top_k, rk in ((True, 10, 150), (False, 10, 50)):
                s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs if rerank else None, max_queries=64)
                ids, sc, stats = s.search(q, vsf, top_k, rk, return_stats=True)
                wi, ws, wst = og.search(opq, codes, v if rerank else None, q, int(vsf), top_k, rk, fused=use_fused)
                tag = (traversal, use_fused, vsf, rerank)
                assert np.array_equal(stats, wst), tag
                assert np.array_equal(ids, wi) and np.array_equal(sc, ws), tag


# ---- GraphSearcher OBJECTS: threshold > 0, rerankFloor, resume() (jv_hip_searcher_*) ------------------------------------
def _same(r, w, tag):
    assert np.array_equal(r.ids, w.ids) and np.array_equal(r.scores, w.scores), tag
    assert (r.visited, r.expanded, r.expanded_base, r.reranked) == (w.visited, w.expanded, w.expanded_base, w.reranked), tag
    assert r.worst_approximate_in_topk == w.worst_approximate_in_topk, tag


def run_searcher_object_cases(J, ctx, cases=4, traversal=None, n_nodes=3000, nq=12):
    """search(topK, rerankK, threshold, rerankFloor, acceptOrds) followed by two resume() calls per query == the oracle's
    jvo_searcher restatement: nodes, scores, the four counters and worstApproximateScoreInTopK.  Shared with the mock.
    traversal: None = the graph's default (AUTO: search() AND resume() run on the DEVICE traversal's session kernels — threshold
    admission, TwoPhaseTracker stop, acceptOrds in the kernel; a resume() replays the earlier calls of the searcher in the same
    launch — the M = 16 cases on the specialised build, the M = 8 cases on the generic one), "host" / "device" pin it."""
    VSF = J.VectorSimilarityFunction
    early = 0
    ctx.reset_stats()
    dev_expected = 0
    for case in range(cases):
        levels = 1 + case % 3
        D, M = [(64, 8), (128, 16)][case % 2]
        v, lv, entry, entry_level, cb, q = build_problem(50 + case, N=n_nodes, D=D, M=M, deg=16, levels=levels)
        N = len(v)
        rng = np.random.default_rng(case)
        if case == 1:                                   # duplicates: exact ties for the heap-order rerank
            v[1::2] = v[0:-1:2][: len(v[1::2])]
        q = q[:nq]
        opq = O.OraclePQ(D, M, cb)
        pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
        vs = J.VectorSet(ctx, v)
        cv = J.PQVectors.encode_and_build(ctx, pq, vs)
        codes = cv.get(0, N)
        og = O.OracleGraph(N, lv, entry, entry_level)
        graph = J.GraphIndex(ctx, N, lv, entry, entry_level)
        if traversal:
            graph.set_traversal(traversal)
        use_fused = case % 2 == 0
        fused = J.FusedPQ(ctx, pq, fused_blocks(codes, lv[0][1]), lv[0][1]) if use_fused else None
        accept = None if case % 2 else (rng.random((len(q), N)) < 0.7)
        if traversal != "host":
            dev_expected += 1
        for vsf in VSF:
            # approximate score levels of this query set, to place thresholds / floors where they bite
            lvl = np.sort(np.stack([opq.adc_scores(q[i], int(vsf), codes, None, fused=use_fused) for i in range(len(q))]), axis=1)
            settings = [(10, 40, 0.0, 0.0), (10, 40, 0.0, float(np.median(lvl[:, -15]))), (5, 25, 0.0, 9.0),
                        (N, N, float(np.median(lvl[:, -120])), 0.0), (300, 300, float(np.median(lvl[:, -40])), float(np.median(lvl[:, -20])))]
            for rerank in (True, False):
                s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs if rerank else None, max_queries=16)
                for top_k, rk, thr, floor in settings:
                    got = s.search_ex(q, vsf, top_k, rk, threshold=thr, rerank_floor=floor, accept=accept)
                    got1 = s.resume(7, 20)
                    got2 = s.resume(30, 30)
                    for i in range(len(q)):
                        o = og.searcher(opq, codes, v if rerank else None, int(vsf), fused=use_fused)
                        acc = None if accept is None else accept[i]
                        tag = (case, vsf, rerank, top_k, rk, thr, floor, i)
                        _same(got[i], o.search(q[i], top_k, rk, thr, floor, accept=acc), tag)
                        _same(got1[i], o.resume(7, 20), tag + ("resume 1",))
                        _same(got2[i], o.resume(30, 30), tag + ("resume 2",))
                        o.close()
                    if thr > 0 and top_k == N:
                        early += sum(r.visited < N * 0.9 for r in got)
                s.close()
        graph.close()
    # TwoPhaseTracker only answers when its observation count sits on a multiple of 100 (ScoreTracker.java:123-126), so an early
    # stop is a matter of luck per search — but over all of these some must have stopped before crawling the whole graph
    assert early > 0, "no threshold search ever stopped early: the tracker path was not exercised"
    if dev_expected:   # the cases really went through the session kernels, their resume() calls included (replayed in-kernel)
        assert ctx.stat("gs_session_calls_device") >= dev_expected * 3 * 2 * 3, ctx.stat("gs_session_calls_device")
        assert ctx.stat("gs_session_resume_device") >= dev_expected * 3 * 2 * 2, ctx.stat("gs_session_resume_device")
    else:
        assert ctx.stat("gs_session_calls_device") == 0


def test_searcher_objects_threshold_floor_resume(ctx):
    run_searcher_object_cases(J, ctx)


def test_searcher_objects_on_the_host_searcher(ctx):
    run_searcher_object_cases(J, ctx, cases=2, traversal="host")


def run_searcher_objects_other_shapes(J, ctx, shapes=((256, 32, 16), (384, 48, 40), (1536, 192, 24)), N=1500, nq=6, vsfs=None):
    """session kernels at the other subspace counts the device traversal is built for (M = 32, 48, 192; degrees <= 32 take the
    pair-lane form, 40 the lane-per-neighbour form): search with a threshold + two resumes == the oracle, all on the device"""
    VSF = J.VectorSimilarityFunction
    for D, M, deg in shapes:
        v, lv, entry, entry_level, cb, q = build_problem(D + deg, N=N, D=D, M=M, deg=deg, top_deg=8, levels=2)
        q = q[:nq]
        N = len(v)
        opq = O.OraclePQ(D, M, cb)
        pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
        vs = J.VectorSet(ctx, v)
        cv = J.PQVectors.encode_and_build(ctx, pq, vs)
        codes = cv.get(0, N)
        og = O.OracleGraph(N, lv, entry, entry_level)
        graph = J.GraphIndex(ctx, N, lv, entry, entry_level)
        fused = J.FusedPQ(ctx, pq, fused_blocks(codes, lv[0][1]), lv[0][1])
        ctx.reset_stats()
        for vsf in (vsfs or (VSF.COSINE, VSF.EUCLIDEAN)):
            lvl = np.sort(np.stack([opq.adc_scores(q[i], int(vsf), codes, None, fused=True) for i in range(len(q))]), axis=1)
            thr = float(np.median(lvl[:, -60]))
            s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=8)
            for top_k, rk, t in ((10, 40, 0.0), (N, N, thr)):
                got = s.search_ex(q, vsf, top_k, rk, threshold=t)
                got1 = s.resume(7, 20)
                got2 = s.resume(20, 30)
                for i in range(len(q)):
                    o = og.searcher(opq, codes, v, int(vsf), fused=True)
                    tag = (D, M, deg, vsf, top_k, i)
                    _same(got[i], o.search(q[i], top_k, rk, t, 0.0), tag)
                    _same(got1[i], o.resume(7, 20), tag + ("resume 1",))
                    _same(got2[i], o.resume(20, 30), tag + ("resume 2",))
                    o.close()
            s.close()
        n_vsf = len(vsfs) if vsfs else 2
        assert ctx.stat("gs_session_calls_device") >= 2 * n_vsf and ctx.stat("gs_session_resume_device") >= 4 * n_vsf, (D, M, ctx.stat("gs_session_calls_device"))
        graph.close()


def test_searcher_objects_other_shapes(ctx):
    run_searcher_objects_other_shapes(J, ctx)


def run_generic_shapes(J, ctx, shapes=((100, 12, 16), (200, 25, 40), (64, 8, 16), (50, 6, 24), (96, 24, 16), (120, 10, 40), (25, 3, 8), (30, 30, 16)),
                       N=1200, nq=10):
    """the device traversal's GENERIC kernels (gs_body.h CH16 = 0): every quantizer outside the specialised builds — ragged
    sub-vectors (100 / 12, 50 / 6, 25 / 3), 8-dim sub-vectors at other M (200 / 25, 64 / 8), 4- and 12-dim ones read as 16-byte words
    (96 / 24, 120 / 10), 1-dim ones (30 / 30), odd D — pinned to the device: plain searches (fused and not, reranked and not) and a
    GraphSearcher object with a threshold + resume, all equal to the oracle bit for bit"""
    VSF = J.VectorSimilarityFunction
    for D, M, deg in shapes:
        levels = 1 + (D + M) % 3
        v, lv, entry, entry_level, cb, q = build_problem(D * 7 + M, N=N, D=D, M=M, deg=deg, top_deg=min(8, deg), levels=levels)
        q = q[:nq]
        n = len(v)
        opq = O.OraclePQ(D, M, cb)
        pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
        vs = J.VectorSet(ctx, v)
        cv = J.PQVectors.encode_and_build(ctx, pq, vs)
        codes = cv.get(0, n)
        og = O.OracleGraph(n, lv, entry, entry_level)
        graph = J.GraphIndex(ctx, n, lv, entry, entry_level).set_traversal("device")
        ctx.reset_stats()
        for use_fused in (True, False):
            fused = J.FusedPQ(ctx, pq, fused_blocks(codes, lv[0][1]), lv[0][1]) if use_fused else None
            for vsf in VSF:
                for rerank, top_k, rk in ((True, 10, 40), (False, 5, 20)):
                    s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs if rerank else None, max_queries=16)
                    ids, sc, stats = s.search(q, vsf, top_k, rk, return_stats=True)
                    wi, ws, wst = og.search(opq, codes, v if rerank else None, q, int(vsf), top_k, rk, fused=use_fused)
                    tag = (D, M, deg, use_fused, vsf, rerank)
                    assert np.array_equal(stats, wst), tag
                    assert np.array_equal(ids, wi) and np.array_equal(sc, ws), tag
                    s.close()
            # GraphSearcher object: threshold search + two resumes through the generic session kernel
            vsf = VSF.COSINE if D % 2 else VSF.EUCLIDEAN
            lvl = np.sort(np.stack([opq.adc_scores(q[i], int(vsf), codes, None, fused=use_fused) for i in range(len(q))]), axis=1)
            thr = float(np.median(lvl[:, -50]))
            s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=16)
            got = s.search_ex(q, vsf, n, n, threshold=thr)
            got1 = s.resume(7, 20)
            got2 = s.resume(20, 30)
            for i in range(len(q)):
                o = og.searcher(opq, codes, v, int(vsf), fused=use_fused)
                tag = (D, M, deg, use_fused, "object", i)
                _same(got[i], o.search(q[i], n, n, thr, 0.0), tag)
                _same(got1[i], o.resume(7, 20), tag + ("resume 1",))
                _same(got2[i], o.resume(20, 30), tag + ("resume 2",))
                o.close()
            s.close()
        assert ctx.stat("gs_calls_device") >= 12 and ctx.stat("gs_calls_host") == 0, (D, M, ctx.stat("gs_calls_device"), ctx.stat("gs_calls_host"))
        assert ctx.stat("gs_session_calls_device") >= 6 and ctx.stat("gs_session_resume_device") >= 4, (D, M)
        graph.close()


def test_generic_pq_shapes_on_the_device_traversal(ctx):
    run_generic_shapes(J, ctx)


def run_wide_rows(J, ctx, shapes=((128, 16, 72), (64, 8, 130), (100, 12, 96)), N=900, nq=8, traversals=("device", "host")):
    """adjacency rows wider than a wavefront (the reference's M grid goes to 128): the traversal kernels and the frontier kernels
    walk such a row 64 neighbours at a time.  Host and device traversal, fused and not, plain searches and a GraphSearcher object
    with a threshold + resume == the oracle"""
    VSF = J.VectorSimilarityFunction
    for D, M, deg in shapes:
        v, lv, entry, entry_level, cb, q = build_problem(D + deg, N=N, D=D, M=M, deg=deg, top_n=max(80, deg + 10), top_deg=min(deg, 70), levels=2)
        q = q[:nq]
        n = len(v)
        opq = O.OraclePQ(D, M, cb)
        pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
        vs = J.VectorSet(ctx, v)
        cv = J.PQVectors.encode_and_build(ctx, pq, vs)
        codes = cv.get(0, n)
        og = O.OracleGraph(n, lv, entry, entry_level)
        for traversal in traversals:
            graph = J.GraphIndex(ctx, n, lv, entry, entry_level).set_traversal(traversal)
            for use_fused in (True, False):
                fused = J.FusedPQ(ctx, pq, fused_blocks(codes, lv[0][1]), lv[0][1]) if use_fused else None
                for vsf in VSF:
                    s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=16)
                    ids, sc, stats = s.search(q, vsf, 10, 30, return_stats=True)
                    wi, ws, wst = og.search(opq, codes, v, q, int(vsf), 10, 30, fused=use_fused)
                    tag = (D, M, deg, traversal, use_fused, vsf)
                    assert np.array_equal(stats, wst), tag
                    assert np.array_equal(ids, wi) and np.array_equal(sc, ws), tag
                    s.close()
                vsf = VSF.COSINE
                lvl = np.sort(np.stack([opq.adc_scores(q[i], int(vsf), codes, None, fused=use_fused) for i in range(len(q))]), axis=1)
                thr = float(np.median(lvl[:, -50]))
                s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=16)
                got = s.search_ex(q, vsf, n, n, threshold=thr)
                got1 = s.resume(9, 25)
                for i in range(len(q)):
                    o = og.searcher(opq, codes, v, int(vsf), fused=use_fused)
                    tag = (D, M, deg, traversal, use_fused, "object", i)
                    _same(got[i], o.search(q[i], n, n, thr, 0.0), tag)
                    _same(got1[i], o.resume(9, 25), tag + ("resume",))
                    o.close()
                s.close()
            graph.close()


def test_rows_wider_than_a_wavefront(ctx):
    run_wide_rows(J, ctx)


def run_small_cluster_count(J, ctx, D=64, M=8, k=40, N=1500, traversals=("host", "device")):
    """a quantizer with fewer than 256 clusters (kept padded on the device side) under the graph searchers: host and device
    traversal, PQVectors codes (FusedPQ needs 256 clusters, as in the reference) == the oracle working with the true count"""
    VSF = J.VectorSimilarityFunction
    v, lv, entry, entry_level, cb256, q = build_problem(k + D, N=N, D=D, M=M, deg=16, levels=2)
    sizes, _ = O.subvector_sizes_offsets(D, M)
    cb = np.concatenate([cb256[256 * int(sizes[:m].sum()): 256 * int(sizes[:m].sum()) + k * int(sizes[m])] for m in range(M)])
    opq = O.OraclePQ(D, M, cb, k=k)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb, cluster_count=k)
    vs = J.VectorSet(ctx, v)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    codes = cv.get(0, len(v))
    assert np.array_equal(codes, opq.encode_all(v)) and int(codes.max()) < k
    og = O.OracleGraph(len(v), lv, entry, entry_level)
    for traversal in traversals:
        graph = J.GraphIndex(ctx, len(v), lv, entry, entry_level).set_traversal(traversal)
        for vsf in VSF:
            s = J.GraphSearcher(ctx, graph, pq, cv, None, vs, max_queries=64)
            ids, sc, stats = s.search(q, vsf, 10, 40, return_stats=True)
            wi, ws, wst = og.search(opq, codes, v, q, int(vsf), 10, 40, fused=False)
            assert np.array_equal(stats, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws), (traversal, vsf)
            s.close()
        graph.close()


def test_graph_search_with_fewer_than_256_clusters(ctx):
    run_small_cluster_count(J, ctx)


def test_searcher_object_errors(ctx):
    v, lv, entry, entry_level, cb, q = build_problem(3, N=500, D=64, M=8, deg=8, levels=1)
    pq = J.ProductQuantization.from_codebooks(ctx, 64, 8, cb)
    vs = J.VectorSet(ctx, v)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    graph = J.GraphIndex(ctx, len(v), lv, entry, entry_level)
    s = J.GraphSearcher(ctx, graph, pq, cv, None, vs, max_queries=8)
    with pytest.raises(ValueError):
        s.resume(5, 5)                                            # resume before search (GraphSearcher.java:533-536)
    with pytest.raises(ValueError):
        s.search_ex(q[:4], VSF.DOT_PRODUCT, 10, 5)                # rerankK < topK (:233)
    with pytest.raises(ValueError):
        s.search_ex(q[:4], VSF.DOT_PRODUCT, 5, 10, threshold=float("nan"))
    r = s.search_ex(q[:4], VSF.DOT_PRODUCT, 5, 10)
    assert all(len(x) == 5 for x in r)
    with pytest.raises(ValueError):
        s.resume(10, 5)
