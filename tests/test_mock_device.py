"""The C ABI's HOST logic on the CPU: the library's host sources (cabi.cpp, graph_search.cpp, build_score.cpp, pq_train.cpp,
formats.cpp) are built against a mock HIP runtime with CPU kernel launchers (tests/mock/) and the GPU test functions are
re-run against that build.  The mock's arithmetic is the oracle's, so these runs say nothing about the kernels (the -m gpu
tests and the lane-emulator tests do that); what they exercise is everything around them: argument validation and error
mapping, host/"device" staging through the pinned buffers, the host batched graph searcher, the device-traversal DRIVER
(scratch sizing, the emulated kernel, status read-back, host fallback for overflowed queries), index ingestion from bytes,
the training launch sequence, and the Python mirror (jvector_amd/engine.py, formats.py) end to end."""
import ctypes as C
import os
import platform
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.skipif(platform.machine() != "x86_64", reason="the lane emulator's context switch is x86-64 assembly")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "mock"))


@pytest.fixture(scope="module")
def J():
    """jvector_amd bound to the mock library for the duration of this module."""
    import build_mock
    import jvector_amd
    import jvector_amd._lib as L
    lib = C.CDLL(build_mock.build())
    for table in (L.SIGNATURES, L.COMPAT_SIGNATURES, L.FORMAT_SIGNATURES):
        for name, (res, args) in table.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    saved, L._lib = L._lib, lib
    # the host searcher's workers spin-wait; on a small shared CPU box that is slower than running the phases inline.
    # One test below turns the pool back on.
    saved_threads = os.environ.get("JVECTOR_HIP_HOST_THREADS")
    os.environ["JVECTOR_HIP_HOST_THREADS"] = "1"
    try:
        assert b"mock" in lib.jv_hip_active_arch(0)
        yield jvector_amd
    finally:
        L._lib = saved
        if saved_threads is None:
            os.environ.pop("JVECTOR_HIP_HOST_THREADS", None)
        else:
            os.environ["JVECTOR_HIP_HOST_THREADS"] = saved_threads


@pytest.fixture()
def ctx(J):
    c = J.HipContext(0)
    yield c
    c.close()


@pytest.fixture()
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# ---- the existing GPU parity tests, on the mock -----------------------------------------------------------------
def test_parity_suite_host_logic(J, ctx, golden_dir):
    import test_gpu_parity as T
    T.test_encode_bit_exact(ctx, 10, 3, False)
    T.test_encode_ties_and_nan(ctx)
    T.test_cluster_counts_below_256(ctx, 64, 8, 16)
    T.test_cluster_counts_below_256(ctx, 100, 7, 50)
    T.test_perfect_reconstruction(ctx)
    T.test_luts_bit_exact(ctx, 10, 3, True)
    T.test_adc_scan_bit_exact(ctx, 128, 16, 20000)
    for B in (1, 100):
        T.test_adc_gather_bit_exact(ctx, B)
    T.test_adc_precomputed_equals_direct(ctx)
    T.test_fused_equals_unfused(ctx, 40, 10, 12)
    for D in (7, 100):
        T.test_exact_gather_and_scan_bit_exact(ctx, D)
    T.test_exact_known_answers(ctx, golden_dir)
    for n, k in ((10, 3), (1000, 10), (50, 64)):
        T.test_topk_matches_nodequeue_order(ctx, n, k)
    T.test_topk_explicit_ids_merge(ctx)
    T.test_topk_short_rows_with_ids_and_padding(ctx)
    T.test_exact_pair_scores_bit_exact(ctx)
    T.test_version0_pq_fixture_on_device(ctx, golden_dir)
    T.test_siftsmall_plumbing(ctx, golden_dir)
    T.test_error_behaviour(ctx)


def test_search_flat_host_logic(J, ctx):
    import test_gpu_parity as T
    T.test_search_flat_matches_oracle(ctx, 128, 16, 20000)


# ---- host batched graph searcher --------------------------------------------------------------------------------
@pytest.mark.parametrize("levels,use_fused,D,M", [(2, True, 64, 8), (3, False, 64, 8)])
def test_host_graph_searcher(J, ctx, levels, use_fused, D, M):
    import test_graph_search as T
    T.test_graph_search_matches_oracle(ctx, levels, use_fused, D, M)


def test_host_graph_searcher_large_batch(J, ctx):
    import test_graph_search as T
    T.test_graph_search_large_batch_and_errors(ctx)
    T.test_graph_search_accept_ords(ctx)


def test_host_graph_searcher_worker_pool(J, monkeypatch):
    """Three pool threads, first search issued right after the context exists: the workers must pick up the very first
    parallel phase even if they are scheduled late (regression: a late worker adopted the pending generation and the caller
    spun forever)."""
    import test_graph_search as T
    monkeypatch.setenv("JVECTOR_HIP_HOST_THREADS", "3")
    for _ in range(3):  # a fresh pool each time
        c = J.HipContext(0)
        try:
            T.test_graph_search_large_batch_and_errors(c)
        finally:
            c.close()


# ---- device-traversal driver (emulated kernel) ------------------------------------------------------------------
def _device_problem(J, ctx, seed, N, D, M, levels, use_fused, deg=16):
    import test_zz_device_traversal_gpu as T
    return T._setup(ctx, seed, N, D, M, levels, use_fused, deg=deg)


@pytest.mark.parametrize("levels,use_fused", [(2, True), (1, False)])
def test_device_traversal_driver(J, ctx, levels, use_fused):
    from oracle import oracle as O
    v, lv, entry, entry_level, opq, pq, vs, cv, codes, graph, fused, q = _device_problem(J, ctx, 31 + levels, 2500, 128, 16, levels,
                                                                                         use_fused)
    q = q[:12]
    og = O.OracleGraph(len(v), lv, entry, entry_level)
    for vsf in J.VectorSimilarityFunction:
        for rerank, top_k, rk in ((True, 10, 40), (False, 5, 20)):
            s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs if rerank else None, max_queries=64)
            ids, sc, stats = s.search(q, vsf, top_k, rk, return_stats=True)
            wi, ws, wst = og.search(opq, codes, v if rerank else None, q, int(vsf), top_k, rk, fused=use_fused)
            assert np.array_equal(stats, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws), (vsf, rerank)


def test_device_traversal_overflow_falls_back_to_the_host(J, ctx, monkeypatch, capfd):
    """A visited table far too small for the search: every query overflows on the device and is re-run by the host
    searcher; the caller sees the same ids, scores and counters."""
    from oracle import oracle as O
    v, lv, entry, entry_level, opq, pq, vs, cv, codes, graph, fused, q = _device_problem(J, ctx, 77, 2500, 128, 16, 2, True)
    q = q[:10]
    og = O.OracleGraph(len(v), lv, entry, entry_level)
    wi, ws, wst = og.search(opq, codes, v, q, O.COSINE, 10, 60, fused=True)
    s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=64)
    monkeypatch.setenv("JVECTOR_HIP_GS_VCAP_LOG2", "8")
    monkeypatch.setenv("JVECTOR_HIP_GRAPH_TIMING", "1")
    ids, sc, stats = s.search(q, J.VectorSimilarityFunction.COSINE, 10, 60, return_stats=True)
    assert "overflow=10" in capfd.readouterr().err          # all ten went through the fallback
    assert np.array_equal(stats, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws)
    monkeypatch.setenv("JVECTOR_HIP_GS_VCAP_LOG2", "10")     # some fit, some do not: results still identical
    ids, sc, stats = s.search(q, J.VectorSimilarityFunction.COSINE, 10, 60, return_stats=True)
    assert np.array_equal(stats, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws)


def test_device_traversal_retry_and_growth_paths(J, ctx, monkeypatch, capfd):
    """the two in-device answers to an outgrown visited table, driven on the mock: the retry pass with 8x / 64x tables and
    the growth pool inside the kernel (the emulated kernel body does the re-insertion and the spill hand-over)"""
    import test_zz_device_traversal_gpu as T
    T.test_overflowed_queries_are_retried_on_the_device(ctx, monkeypatch, capfd)
    T.test_visited_table_grows_inside_the_kernel(ctx, monkeypatch, capfd)
    T.test_exact_score_ties_are_resolved_on_the_device(ctx, monkeypatch, capfd)


def test_device_traversal_tiers_fallbacks_and_counters(J, ctx, capfd):
    """the two-tier visited set at pinned sizes, the host fallback of a graph whose level 0 is device-resident and
    the AUTO-took-the-host notice, driven on the mock through the same functions the GPU suite runs"""
    import test_zz_device_traversal_gpu as T
    T.test_two_tier_visited_set_sizes(ctx)
    lib = ctx._lib
    lib.mock_hip_register_device.argtypes = [C.c_void_p]
    lib.mock_hip_unregister_device.argtypes = [C.c_void_p]
    ptrs = []
    try:
        T.test_host_fallback_with_a_device_resident_level0(ctx, register=lambda p: (ptrs.append(p), lib.mock_hip_register_device(C.c_void_p(p))))
    finally:
        for p in ptrs:
            lib.mock_hip_unregister_device(C.c_void_p(p))
    T.test_auto_traversal_reports_the_host_fallback(ctx, capfd)


def test_workgroup_form_driver_on_the_mock(J, ctx):
    """gs_wgx = 1 through the C ABI on the mock: LDS sizing next to the table, one workgroup per "CU", the emulated control +
    expander waves, candidate-tier spills, the tier-2 table with the growth pool and the retry launches — the functions the GPU
    suite runs"""
    import test_zz_device_traversal_gpu as T
    T.test_workgroup_form_kernel(ctx, 2, True, 128, 16, 16)
    T.test_workgroup_form_kernel(ctx, 1, False, 768, 96, 40)
    T.test_workgroup_form_large_batch_with_spills_and_overflow(ctx, n_queries=60)


def test_device_traversal_refuses_unsupported_shapes(J, ctx):
    import test_zz_device_traversal_gpu as T
    T.test_unsupported_shape_is_refused(ctx)


# ---- ingestion, build-time scoring, anisotropic encode, training: the opt-in GPU tests ----------------------------
@pytest.mark.parametrize("levels,separated,with_pqv", [(2, False, True), (1, True, False), (3, False, False)])
def test_load_index_end_to_end(J, ctx, levels, separated, with_pqv):
    import test_zz_load_index_gpu as T
    T.test_loaded_index_searches_like_the_oracle(ctx, levels, separated, with_pqv)


@pytest.mark.parametrize("D,M,centroid", [(64, 8, False), (50, 7, True)])
def test_build_score_entry_points(J, ctx, D, M, centroid):
    import test_zz_build_score_gpu as T
    T.test_build_score_provider_matches_oracle(ctx, D, M, centroid)


def test_anisotropic_entry_points(J, ctx):
    import test_zz_anisotropic_gpu as T
    T.test_anisotropic_encode_matches_oracle(ctx, 50, 7, True, 0.5)


@pytest.mark.parametrize("D,M,center", [(32, 4, True), (26, 3, False)])
def test_training_entry_points(J, ctx, D, M, center):
    import test_zz_pq_train_gpu as T
    T.test_train_refine_write(ctx, D, M, center)


def test_training_with_fewer_clusters_on_the_mock(J, ctx):
    import test_zz_pq_train_gpu as T
    T.test_train_with_fewer_than_256_clusters(ctx, 32, 4, 16, True)
    T.test_train_with_fewer_than_256_clusters(ctx, 26, 3, 50, False)


def test_anisotropic_training_entry_points(J, ctx):
    import test_zz_pq_train_gpu as T
    T.test_anisotropic_training(ctx)


def test_no_device_memory_leaks(J):
    """every jv_* object created by the tests above was destroyed: the mock runtime has no live device allocations left
    except what lives in still-referenced Python wrappers (collected first)."""
    import gc
    import jvector_amd._lib as L
    gc.collect()
    live = L._lib.mock_hip_live_device_allocations()
    assert live < 50, live


# ---- edge cases of the searchers (cheap here, the same code runs on the GPU) -------------------------------------
@pytest.mark.parametrize("traversal", ["host", "device"])
def test_searcher_edge_cases(J, ctx, traversal):
    """single query, rerankK > number of nodes (exhaustive), nodes without neighbours, an entry node that leads nowhere,
    fewer reachable nodes than topK (padding with -1 / -inf)."""
    from oracle import oracle as O
    import test_graph_search as T
    rng = np.random.default_rng(3)
    D, M, N, deg = 128, 16, 300, 8
    v = rng.standard_normal((N, D)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    sizes, offs = O.subvector_sizes_offsets(D, M)
    cb = np.concatenate([v[rng.choice(N, 256, replace=False), offs[m]: offs[m] + sizes[m]].reshape(-1) for m in range(M)])
    nb = np.full((N, deg), -1, np.int32)
    for i in range(N):
        d = int(rng.integers(0, deg + 1))           # some rows are empty
        row = [int(x) for x in rng.permutation(N)[:d] if x != i]
        nb[i, :len(row)] = row
    nb[7] = -1                                      # node 7: no way out
    nb[0, :3] = [7, 5, 9]
    opq = O.OraclePQ(D, M, cb)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    vs = J.VectorSet(ctx, v)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    codes = cv.get(0, N)
    fused = J.FusedPQ(ctx, pq, T.fused_blocks(codes, nb), nb)
    q = v[[11]] + 0.01
    for entry in (0, 7):
        lv = [(None, nb)]
        graph = J.GraphIndex(ctx, N, lv, entry, 0).set_traversal(traversal)
        og = O.OracleGraph(N, lv, entry, 0)
        for top_k, rk in ((10, 10), (10, 400), (1, 1)):
            s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=4)
            ids, sc, st = s.search(q.astype(np.float32), J.VectorSimilarityFunction.COSINE, top_k, rk, return_stats=True)
            wi, ws, wst = og.search(opq, codes, v, q.astype(np.float32), O.COSINE, top_k, rk, fused=True)
            assert np.array_equal(st, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws), (entry, top_k, rk)
        if entry == 7:
            assert ids[0, 0] == 7 and (ids[0, 1:] == -1).all() if top_k > 1 else ids[0, 0] == 7
    ids, sc = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=4).search(np.zeros((0, D), np.float32),
                                                                                  J.VectorSimilarityFunction.COSINE, 5, 5)
    assert ids.shape == (0, 5)


def test_contexts_on_several_host_threads_share_device_objects(J):
    """the SPI is called concurrently from many host threads.  One context per thread; codebooks, code store,
    fused blocks, vectors and graph are shared; the lazily built cosine-magnitude caches are created under contention."""
    import threading
    from oracle import oracle as O
    import test_graph_search as T
    v, lv, entry, entry_level, cb, q = T.build_problem(91, N=3000, D=64, M=8, levels=2)
    opq = O.OraclePQ(64, 8, cb)
    main = J.HipContext(0)
    pq = J.ProductQuantization.from_codebooks(main, 64, 8, cb)
    vs = J.VectorSet(main, v)
    cv = J.PQVectors.encode_and_build(main, pq, vs)
    codes = cv.get(0, len(v))
    graph = J.GraphIndex(main, len(v), lv, entry, entry_level)
    fused = J.FusedPQ(main, pq, T.fused_blocks(codes, lv[0][1]), lv[0][1])
    want = O.OracleGraph(len(v), lv, entry, entry_level).search(opq, codes, v, q, O.COSINE, 10, 40, fused=True)
    want_adc = np.stack([opq.adc_scores(q[i], O.COSINE, codes[:500]) for i in range(4)])
    errors = []

    def work(tid):
        try:
            c = J.HipContext(0)
            for _ in range(3):
                s = J.GraphSearcher(c, graph, pq, cv, fused, vs, max_queries=64)
                ids, sc, st = s.search(q, J.VectorSimilarityFunction.COSINE, 10, 40, return_stats=True)
                assert np.array_equal(ids, want[0]) and np.array_equal(sc, want[1]) and np.array_equal(st, want[2])
                sf = J.PQVectors(c, pq, codes[:500]).precomputed_score_function_for(q[:4], J.VectorSimilarityFunction.COSINE)
                assert np.array_equal(sf.similarity_to_range(0, 500), want_adc)
            c.close()
        except BaseException as e:  # noqa: BLE001 - reported below
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    main.close()
    assert not errors, errors


def test_graft_entry_smoke_runs_on_the_mock(J, capsys):
    """__graft_entry__.smoke() (the driver's GPU smoke test) end to end on the mock device, and nothing is left allocated."""
    import gc
    import __graft_entry__ as g
    import jvector_amd._lib as L
    gc.collect()
    before = L._lib.mock_hip_live_device_allocations()
    g.smoke()
    gc.collect()
    assert "smoke OK on gfx950:mock" in capsys.readouterr().out
    assert L._lib.mock_hip_live_device_allocations() <= before


def test_sharded_graph_backends_on_the_mock(J):
    """HipGraphShardBackend + ShardedSearcher (three segment graphs in one process) with host tensors on the mock device."""
    import test_zz_sharded_graph_gpu as T
    T.run_sharded_graph_case(lambda t: t)


@pytest.mark.parametrize("traversal", ["host", "device"])
def test_searchers_on_random_irregular_graphs(J, ctx, traversal):
    """Differential sweep against the oracle on graphs no builder would emit: self loops, duplicate neighbours inside a
    row, a -1 in the middle of a row (ends it), rows of every length, 1-3 levels, random topK / rerankK, all three
    similarity functions, with and without FusedPQ and reranking."""
    from oracle import oracle as O
    import test_graph_search as T
    rng = np.random.default_rng(17)
    D, M = 128, 16
    for case in range(14):
        N = int(rng.integers(40, 400))
        deg = int(rng.choice([4, 9, 16, 32, 40]))
        v = rng.standard_normal((N, D)).astype(np.float32)
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        sizes, offs = O.subvector_sizes_offsets(D, M)
        cb = np.concatenate([v[rng.integers(0, N, 256), offs[m]: offs[m] + sizes[m]].reshape(-1) for m in range(M)])
        nb = np.full((N, deg), -1, np.int32)
        for i in range(N):
            d = int(rng.integers(0, deg + 1))
            row = rng.integers(0, N, d)                      # duplicates and self loops allowed
            nb[i, :d] = row
            if d > 2 and rng.random() < 0.15:
                nb[i, int(rng.integers(1, d))] = -1           # everything after it is ignored
        lv = [(None, nb)]
        entry, entry_level = int(rng.integers(0, N)), 0
        n_levels = int(rng.integers(1, 4))
        prev = np.arange(N)
        for _ in range(1, n_levels):
            cnt = max(2, len(prev) // 6)
            ids = np.sort(rng.choice(prev, cnt, replace=False)).astype(np.int32)
            udeg = int(rng.choice([3, 8, 33]))
            un = np.full((cnt, udeg), -1, np.int32)
            for r in range(cnt):
                d = int(rng.integers(0, min(udeg, cnt) + 1))
                un[r, :d] = rng.choice(ids, d, replace=True)
            lv.append((ids, un))
            prev, entry, entry_level = ids, int(ids[rng.integers(0, cnt)]), len(lv) - 1
        use_fused = bool(rng.random() < 0.6)
        rerank = bool(rng.random() < 0.6)
        top_k = int(rng.integers(1, 12))
        rk = top_k + int(rng.integers(0, 60))
        opq = O.OraclePQ(D, M, cb)
        pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
        vs = J.VectorSet(ctx, v)
        cv = J.PQVectors.encode_and_build(ctx, pq, vs)
        codes = cv.get(0, N)
        graph = J.GraphIndex(ctx, N, lv, entry, entry_level).set_traversal(traversal)
        fused = J.FusedPQ(ctx, pq, T.fused_blocks(codes, nb), nb) if use_fused else None
        q = (v[rng.integers(0, N, 5)] + 0.05 * rng.standard_normal((5, D))).astype(np.float32)
        og = O.OracleGraph(N, lv, entry, entry_level)
        s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs if rerank else None, max_queries=8)
        for vsf in J.VectorSimilarityFunction:
            ids, sc, st = s.search(q, vsf, top_k, rk, return_stats=True)
            wi, ws, wst = og.search(opq, codes, v if rerank else None, q, int(vsf), top_k, rk, fused=use_fused)
            tag = (case, N, deg, n_levels, use_fused, rerank, top_k, rk, vsf)
            assert np.array_equal(st, wst), tag
            assert np.array_equal(ids, wi) and np.array_equal(sc, ws), tag


@pytest.mark.parametrize("M", [48, 64, 96, 128, 192])
def test_device_traversal_driver_at_every_supported_M(J, ctx, M):
    """the device-traversal DRIVER (LDS sizing with / without the pair-lane exchange area — pair form only up to M = 96 —
    kernel instantiation per M / 16) through the C ABI, against the oracle, FusedPQ cosine as in the headline configuration"""
    from oracle import oracle as O
    import test_graph_search as T
    D = 8 * M
    v, lv, entry, entry_level, cb, q = T.build_problem(M, N=600, D=D, M=M, deg=32, levels=2)
    N = v.shape[0]
    opq = O.OraclePQ(D, M, cb)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    vs = J.VectorSet(ctx, v)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    codes = cv.get(0, N)
    graph = J.GraphIndex(ctx, N, lv, entry, entry_level).set_traversal("device")
    fused = J.FusedPQ(ctx, pq, T.fused_blocks(codes, lv[0][1]), lv[0][1])
    s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=8)
    ids, sc, st = s.search(q[:6], J.VectorSimilarityFunction.COSINE, 10, 50, return_stats=True)
    wi, ws, wst = O.OracleGraph(N, lv, entry, entry_level).search(opq, codes, v, q[:6], O.COSINE, 10, 50, fused=True)
    assert np.array_equal(st, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws)


def test_device_traversal_refuses_a_rerank_k_that_does_not_fit_lds(J, ctx):
    """the result heap lives in LDS: a rerankK whose queues exceed the per-block limit is refused (not silently truncated),
    and the host traversal serves the same request"""
    from oracle import oracle as O
    import test_graph_search as T
    import jvector_amd._lib as L
    v, lv, entry, entry_level, cb, q = T.build_problem(77, N=400, D=128, M=16, deg=8, levels=1)
    pq = J.ProductQuantization.from_codebooks(ctx, 128, 16, cb)
    vs = J.VectorSet(ctx, v)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    graph = J.GraphIndex(ctx, 400, lv, entry, entry_level).set_traversal("device")
    s = J.GraphSearcher(ctx, graph, pq, cv, None, vs, max_queries=4)
    with pytest.raises(L.UnsupportedError, match="LDS"):
        s.search(q[:2], J.VectorSimilarityFunction.COSINE, 10, 30000)
    graph.set_traversal("host")
    ids, sc = s.search(q[:2], J.VectorSimilarityFunction.COSINE, 10, 30000)
    wi, ws, _ = O.OracleGraph(400, lv, entry, entry_level).search(O.OraclePQ(128, 16, cb), cv.get(0, 400), v, q[:2], O.COSINE, 10, 30000)
    assert np.array_equal(np.asarray(ids), wi) and np.array_equal(np.asarray(sc), ws)


def test_ground_truth_from_dense_candidates_equals_the_exact_one(J, ctx):
    """benchlib.ground_truth(dense=True): MFMA-scan candidates + bit-exact rescoring == the all-bit-exact ground truth"""
    import torch
    import benchlib
    rng = np.random.default_rng(12)
    v = rng.standard_normal((700, 40)).astype(np.float32)
    q = (v[rng.integers(0, 700, 9)] + 0.1 * rng.standard_normal((9, 40))).astype(np.float32)
    vs = J.VectorSet(ctx, v)
    for vsf in J.VectorSimilarityFunction:
        a = benchlib.ground_truth(J, ctx, vs, torch.from_numpy(q), vsf, 10, chunk=256)
        b = benchlib.ground_truth(J, ctx, vs, torch.from_numpy(q), vsf, 10, chunk=256, dense=True)
        assert np.array_equal(np.asarray(a), np.asarray(b)), vsf


def test_dense_mfma_scan_through_the_c_abi(J, ctx):
    """jv_hip_exact_scan_dense end to end on the mock: the shared kernel body on the lane emulator (documented MFMA
    semantics) behind the real host entry point, staging and Python wrapper"""
    import test_zz_exact_dense_gpu as T
    T.run_dense_cases(J, ctx, shapes=((1, 1, 1), (5, 130, 7), (33, 129, 100), (40, 300, 64)))


@pytest.mark.parametrize("traversal", ["host", "device"])
def test_searchers_with_engineered_score_ties(J, ctx, traversal):
    import test_graph_search as T
    T.run_ties_cases(J, ctx, traversal)


def test_searcher_objects_other_shapes_on_the_mock(J, ctx):
    """the session kernels at M = 32 (pair-lane form) and M = 48 at degree 40 (lane-per-neighbour form) on the lane emulator"""
    import test_graph_search as T
    T.run_searcher_objects_other_shapes(J, ctx, shapes=((256, 32, 16), (384, 48, 40)), N=600, nq=3, vsfs=(J.VectorSimilarityFunction.COSINE,))


def test_searcher_objects_on_the_mock(J, ctx):
    """threshold > 0 / rerankFloor / resume() against the oracle's GraphSearcher object: the session kernels on the lane emulator
    (generic build at M = 8, specialised at M = 16) and the host searcher's session path"""
    import test_graph_search as T
    T.run_searcher_object_cases(J, ctx, cases=2, n_nodes=1200, nq=8)
    T.run_searcher_object_cases(J, ctx, cases=1, traversal="host")
    T.test_searcher_object_errors(ctx)


def test_rows_wider_than_a_wavefront_on_the_mock(J, ctx):
    """degree 72 / 96 / 130 graphs: the traversal body's chunk loop on the lane emulator, the host searcher's multi-word masks"""
    import test_graph_search as T
    T.run_wide_rows(J, ctx, N=500, nq=4)


def test_small_cluster_count_on_the_mock(J, ctx):
    import test_graph_search as T
    T.run_small_cluster_count(J, ctx, N=600)


def test_generic_pq_shapes_on_the_mock(J, ctx):
    """the device traversal's generic kernels (any sub-vector geometry) on the lane emulator: searches + GraphSearcher objects"""
    import test_graph_search as T
    T.run_generic_shapes(J, ctx, N=500, nq=4)


@pytest.mark.parametrize("slots,groups", [(64, 1), (96, 3), (700, 2), (1, 1)])
def test_host_searcher_continuous_batching(J, ctx, monkeypatch, slots, groups):
    """more queries than traversal slots: finished queries hand their slot to the next one (and slot groups alternate);
    per-query results and counters must not depend on the slot geometry"""
    import test_graph_search as T
    monkeypatch.setenv("JVECTOR_HIP_GRAPH_SLOTS", str(slots))
    monkeypatch.setenv("JVECTOR_HIP_GRAPH_GROUPS", str(groups))
    T.test_graph_search_large_batch_and_errors(ctx)


@pytest.mark.parametrize("traversal", ["host", "device"])
def test_negative_scores_are_expanded_but_never_results(J, ctx, traversal):
    """DOT_PRODUCT over unnormalised vectors: (1 + dot) / 2 < 0 for strongly opposed pairs.  With threshold 0.0f the reference
    does not add such a candidate to the results (GraphSearcher.java:437) but still expands it."""
    from oracle import oracle as O
    import test_graph_search as T
    rng = np.random.default_rng(8)
    D, M, N, deg = 128, 16, 400, 8
    v = (3.0 * rng.standard_normal((N, D))).astype(np.float32)          # norms ~ 34: dots reach +-hundreds
    sizes, offs = O.subvector_sizes_offsets(D, M)
    cb = np.concatenate([v[rng.integers(0, N, 256), offs[m]: offs[m] + sizes[m]].reshape(-1) for m in range(M)])
    nb = np.stack([rng.permutation(N)[:deg] for _ in range(N)]).astype(np.int32)
    opq = O.OraclePQ(D, M, cb)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    vs = J.VectorSet(ctx, v)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    codes = cv.get(0, N)
    lv = [(None, nb)]
    graph = J.GraphIndex(ctx, N, lv, 0, 0).set_traversal(traversal)
    fused = J.FusedPQ(ctx, pq, T.fused_blocks(codes, nb), nb)
    q = (-v[rng.integers(0, N, 6)]).astype(np.float32)                    # opposed to some node: many negative scores
    og = O.OracleGraph(N, lv, 0, 0)
    s = J.GraphSearcher(ctx, graph, pq, cv, fused, None, max_queries=8)
    ids, sc, st = s.search(q, J.VectorSimilarityFunction.DOT_PRODUCT, 20, 500, return_stats=True)
    wi, ws, wst = og.search(opq, codes, None, q, O.DOT_PRODUCT, 20, 500, fused=True)
    assert np.array_equal(st, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws)
    assert (st[:, 1] > (ids >= 0).sum(axis=1)).all()       # more nodes were expanded than became results
    assert (sc[ids >= 0] >= 0).all()                       # and no negative score was kept


def test_fused_build_entry_point(J, ctx):
    """jv_hip_fused_build == uploading blocks assembled on the host (FusedPQ.writeInline), and it searches the same."""
    from oracle import oracle as O
    import test_graph_search as T
    v, lv, entry, entry_level, cb, q = T.build_problem(5, N=1500, D=128, M=16, levels=2)
    pq = J.ProductQuantization.from_codebooks(ctx, 128, 16, cb)
    vs = J.VectorSet(ctx, v)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    codes = cv.get(0, len(v))
    built = J.FusedPQ.build(ctx, cv, lv[0][1])
    blocks, nbrs = built.get()
    assert np.array_equal(blocks, T.fused_blocks(codes, lv[0][1])) and np.array_equal(nbrs, lv[0][1])
    graph = J.GraphIndex(ctx, len(v), lv, entry, entry_level)
    got = J.GraphSearcher(ctx, graph, pq, cv, built, vs, max_queries=64).search(q, J.VectorSimilarityFunction.COSINE, 10, 40)
    opq = O.OraclePQ(128, 16, cb)
    wi, ws, _ = O.OracleGraph(len(v), lv, entry, entry_level).search(opq, codes, v, q, O.COSINE, 10, 40, fused=True)
    assert np.array_equal(got[0], wi) and np.array_equal(got[1], ws)


@pytest.mark.parametrize("traversal", ["host", "device"])
def test_filtered_search_accept_ords(J, ctx, traversal, monkeypatch):
    """GraphSearcher.search(..., acceptOrds): one filter for the batch, one filter per query, and (device traversal) the
    per-query masks of queries that fall back to the host."""
    from oracle import oracle as O
    import test_graph_search as T
    v, lv, entry, entry_level, cb, q = T.build_problem(19, N=2500, D=128, M=16, levels=2)
    q = q[:12]
    N = len(v)
    opq = O.OraclePQ(128, 16, cb)
    pq = J.ProductQuantization.from_codebooks(ctx, 128, 16, cb)
    vs = J.VectorSet(ctx, v)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    codes = cv.get(0, N)
    graph = J.GraphIndex(ctx, N, lv, entry, entry_level).set_traversal(traversal)
    fused = J.FusedPQ(ctx, pq, T.fused_blocks(codes, lv[0][1]), lv[0][1])
    og = O.OracleGraph(N, lv, entry, entry_level)
    s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs, max_queries=16)
    rng = np.random.default_rng(2)
    shared = rng.random(N) < 0.3
    per_query = rng.random((len(q), N)) < 0.2
    per_query[3] = False                                   # nothing acceptable: empty result, the whole component is searched
    per_query[4] = True
    for accept in (shared, per_query):
        for vsf in J.VectorSimilarityFunction:
            ids, sc, st = s.search(q, vsf, 10, 40, return_stats=True, accept=accept)
            wi, ws, wst = og.search(opq, codes, v, q, int(vsf), 10, 40, fused=True, accept=accept)
            assert np.array_equal(st, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws), vsf
            rows = accept if accept.ndim == 2 else np.broadcast_to(accept, (len(q), N))
            for i in range(len(q)):
                assert rows[i][ids[i][ids[i] >= 0]].all()
    assert (ids[3] == -1).all() and st[3, 1] > 1000
    if traversal == "device":                             # overflow -> host fallback must carry each query's own mask
        monkeypatch.setenv("JVECTOR_HIP_GS_VCAP_LOG2", "9")
        ids, sc, st = s.search(q, J.VectorSimilarityFunction.COSINE, 10, 40, return_stats=True, accept=per_query)
        wi, ws, wst = og.search(opq, codes, v, q, O.COSINE, 10, 40, fused=True, accept=per_query)
        assert np.array_equal(st, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws)
    with pytest.raises(ValueError):
        s.search(q, J.VectorSimilarityFunction.COSINE, 10, 40, accept=np.ones(N - 1, bool))


def test_train_encode_build_write_load_search_pipeline(J, ctx):
    """The whole artefact pipeline through the C ABI: train codebooks -> encode -> gather the FusedPQ blocks -> write an
    OnDiskGraphIndex v6 file and a PQVectors blob -> load both back -> search; every stage equals the oracle's."""
    import jvector_amd.formats as F
    from oracle import oracle as O
    import test_graph_search as T
    v, lv, entry, entry_level, _, q = T.build_problem(123, N=2000, D=64, M=8, levels=2)
    pq = J.ProductQuantization.compute(ctx, v, 8, globally_center=True, seed=11)
    opq, _ = O.pq_train(v, 8, globally_center=True, seed=11)
    vs = J.VectorSet(ctx, v)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    codes = cv.get(0, len(v))
    assert np.array_equal(codes, opq.encode_all(v))
    fused = J.FusedPQ.build(ctx, cv, lv[0][1])
    blocks, _ = fused.get()
    odgi = F.write_odgi(64, lv, entry, vectors=v, fused_blocks=blocks, pq_block=pq.write(6), hierarchy_codes=codes[lv[1][0]])
    pqv = F.write_pqvectors(pq, cv)
    assert pqv[:len(pq.write(6))] == opq.serialize(6)
    idx = F.load_index(ctx, odgi, pqv)
    ids, sc, st = idx.searcher(max_queries=64).search(q, J.VectorSimilarityFunction.COSINE, 10, 50, return_stats=True)
    wi, ws, wst = O.OracleGraph(len(v), lv, entry, entry_level).search(opq, codes, v, q, O.COSINE, 10, 50, fused=True)
    assert np.array_equal(st, wst) and np.array_equal(ids, wi) and np.array_equal(sc, ws)


def test_standalone_canary_against_the_mock_library(J):
    """tools/gs_canary.cpp (the first thing scripts/validate_device_traversal.sh runs on hardware) built against the mock
    library: device traversal == host traversal, exit code 0, for the three similarity functions."""
    import json
    import subprocess
    import build_mock
    lib = build_mock.build()
    exe = os.path.join(ROOT, "build", "mock", "gs_canary_mock")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "tools", "gs_canary.cpp"), "-o", exe,
                           "-L" + os.path.dirname(lib), "-l:" + os.path.basename(lib), "-Wl,-rpath," + os.path.dirname(lib)])
    env = dict(os.environ, JVECTOR_HIP_HOST_THREADS="1")
    for vsf in (0, 1, 2):
        out = subprocess.run([exe, "2500", "16", "32", "50", "1", str(vsf)], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        line = json.loads(out.stdout.strip().splitlines()[-1])
        assert line["identical"] is True and line["avg_expanded"] >= 50


@pytest.mark.parametrize("mode,traversal,graph,n,extra", [("graph", "host", "synthetic", 3000, []), ("graph", "device", "engine", 1200, []),
                                                         ("flat", "host", "synthetic", 6000, []),
                                                         ("graph", "device", "engine", 1200, ["--reranker", "nvq"])])
def test_bench_dry_run_on_the_mock(J, monkeypatch, capsys, tmp_path, mode, traversal, graph, n, extra):
    """bench.py end to end at toy size: torch runs on the CPU (a proxy maps the `cuda` device bench asks for to `cpu` and makes
    the stream / synchronize calls inert) and the engine is the mock device.  Numbers are meaningless; what is checked is the
    control flow — index build, rerankK calibration against exact ground truth, the timed loop, the secondary flat
    measurement, the CPU-baseline leg and its top-K comparison — and the JSON contract of the one output line."""
    import json
    import types
    import torch
    import bench

    class TorchProxy:
        cuda = types.SimpleNamespace(set_device=lambda *_a: None, synchronize=lambda *_a: None,
                                     current_stream=lambda *_a: types.SimpleNamespace(cuda_stream=0))

        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def device(*_a, **_k):
            return torch.device("cpu")

    monkeypatch.setattr(bench, "torch", TorchProxy())
    argv = ["bench.py", "--mode", mode, "--traversal", traversal, "--graph", graph, "--n", str(n), "--dim", "128", "--m", "16", "--degree", "16",
            "--queries", "48", "--steps", "2", "--warmup", "1", "--eval-queries", "48", "--cal-queries", "48"] + extra
    monkeypatch.setattr(sys, "argv", argv)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setenv("JVECTOR_BENCH_FULL", str(tmp_path / "bench_full.json"))
    bench.main()
    out = capsys.readouterr().out.strip().splitlines()
    contract = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline")
    # the LAST stdout line is the compact, driver-readable one (< 4 KB); the full line is in the file it names
    assert len(out[-1]) < 4096
    compact = json.loads(out[-1])
    for key in contract:
        assert key in compact, key
    assert compact["roofline"]["frac"] is not None and compact["cpu_baseline"]["value"] > 0 and "workload" in compact["config"]
    line = json.load(open(compact["full"]))
    assert abs(compact["value"] - line["value"]) <= 1e-5 * line["value"] and compact["config"]["rerankK"] == line["config"]["rerankK"]
    for key in contract:
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["higher_is_better"] is True
    assert line["config"]["mode"] == mode and line["config"]["n_vectors"] == n and "workload" in line["config"]
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    cb = line["cpu_baseline"]
    if "nvq" in extra:   # NVQ rows rerank on the GPU, float rows on the CPU leg: the line says so instead of claiming equality
        assert line["reranker"] == "nvq" and line["nvq"]["bytes_per_row"] == 128 + 32 and "note" in cb
        assert "rerank" not in line or "nvq_gather_kernel" in line["rerank"]["kernel"]   # (the mock's events measure no time)
    else:
        assert cb["matches_gpu_topk"] is True                        # the scalar leg is the parity checker
    assert cb["isa"] in ("scalar", "avx2", "avx512") and cb["scalar_value"] > 0
    if cb["isa"] != "scalar" and "nvq" not in extra:                  # SIMD leg: the reported value, ~the same top-k
        assert cb["value"] != cb["scalar_value"] and cb["simd_topk_overlap_with_gpu"] >= 0.98
    assert 0.0 <= line["recall_at_10"] <= 1.0 and line["recall_at_10"] > 0.5
    if mode == "graph":
        assert line["config"]["traversal"] == traversal and line["avg_expanded"] > 0 and "flat_mode" in line
        assert line["graph"] == graph and (graph != "engine" or line["graph_build"]["nodes_per_s"] > 0)


def test_bench_index_cache_hand_over_on_the_mock(J, monkeypatch, capsys, tmp_path):
    """`bench.py --index-cache FILE` (the hand-over that lets ONE rank build the 10M index and the others load it): the
    first run builds the graph and writes the npz (graph levels + the quantizer's wire bytes), the second finds it, builds nothing and
    serves the same index — same calibrated rerankK, same recall, same adjacency statistics."""
    import json
    import types
    import torch
    import bench

    class TorchProxy:
        cuda = types.SimpleNamespace(set_device=lambda *_a: None, synchronize=lambda *_a: None,
                                     current_stream=lambda *_a: types.SimpleNamespace(cuda_stream=0))

        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def device(*_a, **_k):
            return torch.device("cpu")

    monkeypatch.setattr(bench, "torch", TorchProxy())
    cache = str(tmp_path / "index.npz")
    argv = ["bench.py", "--mode", "graph", "--traversal", "device", "--graph", "engine", "--n", "1200", "--dim", "128", "--m", "16", "--degree", "16",
            "--queries", "48", "--steps", "1", "--warmup", "1", "--eval-queries", "48", "--cal-queries", "48", "--no-flat", "--no-cpu-baseline",
            "--index-cache", cache]
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setenv("JVECTOR_BENCH_FULL", str(tmp_path / "bench_full.json"))
    lines = []
    for _ in range(2):
        monkeypatch.setattr(sys, "argv", argv)
        bench.main()
        compact = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
        lines.append(json.load(open(compact["full"])))
        assert os.path.exists(cache)
    built, loaded = lines
    assert built["graph_build"]["nodes_per_s"] > 0 and not loaded.get("graph_build")      # the second run built nothing
    assert built["config"]["rerankK"] == loaded["config"]["rerankK"] and built["recall_at_10"] == loaded["recall_at_10"]
    assert built["avg_expanded"] == loaded["avg_expanded"] and built["avg_visited"] == loaded["avg_visited"]


@pytest.mark.parametrize("workload,extra", [("c2", ["--n", "3000", "--queries", "32", "--eval-queries", "64", "--rerank", "100"]),
                                            ("c4", ["--n", "3000", "--dim", "128", "--m", "16", "--queries", "16", "--rerank", "40"]),
                                            ("c5", ["--n", "1500", "--dim", "128", "--m", "16", "--degree", "16", "--eval-queries", "32"])])
def test_bench_other_workloads_dry_run(J, monkeypatch, capsys, tmp_path, workload, extra):
    """the secondary workloads of bench.py (C2 flat SIFT-like, C4 sharded through the C ABI's communicator — here a one-rank
    communicator on the shared-memory RCCL shim —, C5 index build) end to end at toy size on the mock: control flow + JSON contract"""
    import json
    import types
    import torch
    import bench
    import test_sharded_cabi as TS
    monkeypatch.setenv("JVECTOR_HIP_RCCL_PATH", TS.build_shim())

    class TorchProxy:
        cuda = types.SimpleNamespace(set_device=lambda *_a: None, synchronize=lambda *_a: None,
                                     current_stream=lambda *_a: types.SimpleNamespace(cuda_stream=0))

        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def device(*_a, **_k):
            return torch.device("cpu")

    monkeypatch.setattr(bench, "torch", TorchProxy())
    monkeypatch.setattr(sys, "argv", ["bench.py", "--workload", workload, "--steps", "2", "--warmup", "1"] + extra)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setenv("JVECTOR_BENCH_FULL", str(tmp_path / "bench_full.json"))
    bench.main()
    last = capsys.readouterr().out.strip().splitlines()[-1]
    assert len(last) < 4096
    compact = json.loads(last)
    line = json.load(open(compact["full"]))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "higher_is_better", "scaling", "dtype", "data", "config"):
        assert key in line and key in compact, key
    assert compact["roofline"]["frac"] is not None
    assert line["value"] > 0 and "workload" in line["config"]
    if workload == "c2":
        assert line["roofline"]["bound"] == "lds" and line["cpu_baseline"]["matches_gpu_topk"] is True
    if workload == "c5":
        assert line["build"]["avg_degree"] > 4 and max(line["recall_at_10_by_rerankK"].values()) > 0.8


def _bench_rank(rank, world, port, mode, out_dir):
    """one rank of a 2-process bench.py dry run: gloo instead of RCCL, CPU tensors, the mock device"""
    import contextlib
    import ctypes as C2
    import types
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), JVECTOR_HIP_HOST_THREADS="1", OMP_NUM_THREADS="2", MKL_NUM_THREADS="2")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "mock"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_sharded_cabi as TS
    os.environ["JVECTOR_HIP_RCCL_PATH"] = TS.build_shim()   # the engine's own communicator (rccl_ranks) lands in the shared-memory shim
    import build_mock
    import jvector_amd._lib as L
    lib = C2.CDLL(build_mock.build())
    for table in (L.SIGNATURES, L.COMPAT_SIGNATURES, L.FORMAT_SIGNATURES):
        for name, (res, args) in table.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    L._lib = lib
    import torch
    import torch.distributed as dist
    torch.set_num_threads(2)  # two ranks share this box: no oversubscription
    import bench
    real_init = dist.init_process_group
    dist.init_process_group = lambda backend=None, **kw: real_init("gloo", rank=rank, world_size=world)

    class TorchProxy:
        cuda = types.SimpleNamespace(set_device=lambda *_a: None, synchronize=lambda *_a: None,
                                     current_stream=lambda *_a: types.SimpleNamespace(cuda_stream=0))

        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def device(*_a, **_k):
            return torch.device("cpu")

    bench.torch = TorchProxy()
    sys.argv = ["bench.py", "--gpus", str(world), "--mode", mode, "--graph", "synthetic", "--n", "2500", "--dim", "128", "--m", "16", "--degree", "16", "--queries",
                "32", "--steps", "2", "--warmup", "1", "--eval-queries", "32", "--cal-queries", "32"]
    with open(os.path.join(out_dir, f"rank{rank}.out"), "w") as f, contextlib.redirect_stdout(f):
        bench.main()


@pytest.mark.parametrize("mode", ["graph", "flat"])
def test_bench_two_rank_dry_run(tmp_path, mode):
    """bench.py --gpus 2 as the driver launches it (one process per rank, RANK / WORLD_SIZE / LOCAL_WORLD_SIZE in the
    environment), with gloo standing in for RCCL: rank 0 prints ONE line with n_gpus = 2, the aggregate over both ranks and the
    max-over-ranks time; rank 1 prints nothing."""
    import json
    import socket
    import torch.multiprocessing as mp
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    mp.spawn(_bench_rank, args=(2, port, mode, str(tmp_path)), nprocs=2, join=True)
    out0 = open(tmp_path / "rank0.out").read().strip().splitlines()
    out1 = open(tmp_path / "rank1.out").read().strip()
    assert out1 == "" and len(out0) == 1
    line = json.loads(out0[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["steps"] == 2 and line["scaling"] == "weak"
    assert line["config"]["mode"] == mode and "2 replicas" in line["config"]["parallelism"]
    assert abs(line["value"] - 2 * 32 * 2 / (line["ms_per_step"] * 2 / 1e3)) < 1e-6 * line["value"]   # total queries / elapsed
    assert "cpu_baseline" not in line                                                                # rank 0 at N = 1 only


def _run_bench_on_mock(argv, timeout=900):
    """`python tests/mock/bench_on_mock.py <argv>` as ONE process, no launcher environment: returns (returncode, JSON lines, stderr)"""
    import json
    import subprocess
    import test_sharded_cabi as TS
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(JVECTOR_HIP_RCCL_PATH=TS.build_shim(), JVECTOR_HIP_HOST_THREADS="1", OMP_NUM_THREADS="2", MKL_NUM_THREADS="2")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mock", "bench_on_mock.py")] + argv, env=env, capture_output=True,
                         text=True, timeout=timeout)
    lines = [json.loads(x) for x in out.stdout.splitlines() if x.startswith("{")]
    return out.returncode, lines, out.stderr


@pytest.mark.parametrize("workload", ["c3", "c4"])
def test_bench_gpus_flag_spawns_its_own_ranks(J, workload):
    """`bench.py --gpus 2` given to ONE process (no RANK / WORLD_SIZE in its environment) must start two ranks itself and print
    ONE line with n_gpus == 2 and rccl_ranks == 2 — the engine's own communicator (here on the shared-memory RCCL shim) joined
    by both.  c3 = replicas (weak scaling line), c4 = the sharded index through jv_hip_sharded_search_flat."""
    if workload == "c3":
        argv = ["--gpus", "2", "--mode", "graph", "--graph", "synthetic", "--n", "2500", "--dim", "128", "--m", "16", "--degree", "16",
                "--queries", "32", "--steps", "2", "--warmup", "1", "--eval-queries", "32", "--cal-queries", "32"]
    else:
        argv = ["--gpus", "2", "--workload", "c4", "--n", "3000", "--dim", "128", "--m", "16", "--queries", "16", "--steps", "2",
                "--warmup", "1", "--eval-queries", "32", "--rerank", "40"]
    rc, lines, err = _run_bench_on_mock(argv)
    assert rc == 0, err[-3000:]
    assert len(lines) == 1, (lines, err[-2000:])
    line = lines[0]
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and len(line["per_rank_qps"]) == 2 and all(v > 0 for v in line["per_rank_qps"])
    assert line["steps"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    if workload == "c3":
        assert "2 replicas" in line["config"]["parallelism"]
        assert abs(line["value"] - 2 * 32 * 2 / (line["ms_per_step"] * 2 / 1e3)) < 1e-6 * line["value"]
    else:
        assert line["config"]["n_vectors"] == 6000 and line["config"]["shard"] == 3000 and line["recall_at_10"] > 0.8
        assert line["roofline"]["bound"] == "lds" and set(line["roofline"]) >= {"achieved", "peak", "unit", "frac", "traffic"}
        assert abs(line["value"] - 16 * 2 / (line["ms_per_step"] * 2 / 1e3)) < 1e-6 * line["value"]   # every rank answers every query


def test_bench_eight_ranks_sharded_c4_on_the_mock(J):
    """`bench.py --gpus 8 --workload c4` as ONE command: eight ranks started by the script itself, the engine's own
    communicator joined by all of them (rccl_ranks == 8, here the shared-memory RCCL shim), eight shards of 1 000 vectors answered
    through jv_hip_sharded_search_flat; recall against the exact ground truth of the whole 8 000-vector index says the exchange merged
    the right lists"""
    argv = ["--gpus", "8", "--workload", "c4", "--n", "1000", "--dim", "128", "--m", "16", "--queries", "16", "--steps", "2",
            "--warmup", "1", "--eval-queries", "32", "--rerank", "40"]
    rc, lines, err = _run_bench_on_mock(argv, timeout=1500)
    assert rc == 0, err[-3000:]
    assert len(lines) == 1, (lines, err[-2000:])
    line = lines[0]
    assert line["n_gpus"] == 8 and line["rccl_ranks"] == 8 and len(line["per_rank_qps"]) == 8 and all(v > 0 for v in line["per_rank_qps"])
    assert line["config"]["n_vectors"] == 8000 and line["config"]["shard"] == 1000 and line["recall_at_10"] > 0.8
    assert line["scaling"] == "weak" and "cpu_baseline" not in line and line["roofline"]["bound"] == "lds"
    assert abs(line["value"] - 16 * 2 / (line["ms_per_step"] * 2 / 1e3)) < 1e-6 * line["value"]   # every rank answers every query


def test_bench_refuses_a_launcher_that_disagrees_with_gpus(J):
    """WORLD_SIZE from the launcher != --gpus: no line at all rather than one with the wrong n_gpus"""
    import subprocess
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mock", "bench_on_mock.py"), "--gpus", "2", "--n", "2000"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "--gpus 2" in out.stderr and not [x for x in out.stdout.splitlines() if x.startswith("{")]


def test_golden_checker_through_the_c_abi_on_the_mock(J):
    """tests/test_reference_goldens.py's HIP-side adapter (every quantity GoldenDump.java records, fetched through the C ABI)
    against the oracle-made golden file, on the mock device: keeps the adapter's API usage compiling until a GPU run does it for real"""
    import test_reference_goldens as T
    g = T.oracle_made_goldens()
    T.check_goldens(g, T.HipSide(g), exact=True)


def test_register_table_bound_form_on_the_mock(J, ctx):
    """tests/test_zz_ubr_gpu.py on the mock device (the traversal on the lane emulator, the tables by gs_host.h's restatement): the
    driver's side of the form — option gs_ubr, table scratch, the drop counter, stats, the fall-backs for euclidean / filtered searches"""
    import test_zz_ubr_gpu as T
    T.test_bound_tables_equal_the_restatement(ctx, J.VectorSimilarityFunction.COSINE)
    T.test_register_table_bound_kernel(ctx, 2, True, 32, 2500)
    T.test_register_table_bound_kernel_ties_and_degenerate_queries(ctx)


def test_fused_rerank_on_the_mock(J, ctx):
    """the rerank inside the traversal wave (gs_body.h gs_rr_round on the lane emulator; the kernel of its own is the oracle's arithmetic
    here): fused == unfused == the oracle, and the driver's shape rule (gs_last_rr_rows)"""
    import test_zz_ubr_gpu as T
    T.test_fused_rerank_equals_the_rerank_kernel(ctx, N=3000, quick=True)
