"""One-off validation aid: a time-boxed differential sweep of GraphSearcher OBJECTS (jv_hip_searcher_*: search with threshold /
rerankFloor / acceptOrds, then a random chain of resume() calls) against the oracle's jvo_searcher restatement.  Random shapes
(every subspace count the session kernels have a specialised build for + arbitrary quantizers, which take the generic build), degrees, level counts, fused /
unfused, reranker / none, similarity functions, call chains long enough to leave the in-kernel replay (GS_MAX_PHASES) for the host
replay, and small candidate / push-log capacities so that spills, refills, retries and host fallbacks all occur.
Every result must agree bit for bit: nodes, scores, the four counters, worstApproximateScoreInTopK.
usage (GPU box): python scripts/fuzz_searcher.py [seconds] [seed]
       (here)  : FUZZ_MOCK=1 python scripts/fuzz_searcher.py 60   — the same sweep on the CPU mock + lane emulator (slow, small)"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
MOCK = os.environ.get("FUZZ_MOCK") == "1"
if MOCK:
    sys.path.insert(0, os.path.join(ROOT, "tests", "mock"))
    os.environ.setdefault("JVECTOR_HIP_HOST_THREADS", "1")
    import build_mock
    import jvector_amd._lib as L
    lib = C.CDLL(build_mock.build())
    for table in (L.SIGNATURES, L.COMPAT_SIGNATURES, L.FORMAT_SIGNATURES):
        for name, (res, args) in table.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    L._lib = lib
import jvector_amd as J  # noqa: E402
from oracle import oracle as O  # noqa: E402
from test_graph_search import build_problem, fused_blocks  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = J.HipContext(0)
ctx.set_option("quiet", 1)
VSF = list(J.VectorSimilarityFunction)
t_end = time.time() + budget
cases = calls = checks = 0
dev_calls = dev_resumes = host_calls = 0
OPTS = ("gs_cand_cap", "gs_push_log_cap", "gs_vcap_log2", "gs_v1_log2", "gs_generic")


def same(r, w, tag):
    ok = (np.array_equal(r.ids, w.ids) and np.array_equal(r.scores, w.scores)
          and (r.visited, r.expanded, r.expanded_base, r.reranked) == (w.visited, w.expanded, w.expanded_base, w.reranked)
          and r.worst_approximate_in_topk == w.worst_approximate_in_topk)
    if not ok:
        print("MISMATCH", tag)
        print("  got ", r.ids[:12], r.scores[:6], r.visited, r.expanded, r.expanded_base, r.reranked, r.worst_approximate_in_topk)
        print("  want", w.ids[:12], w.scores[:6], w.visited, w.expanded, w.expanded_base, w.reranked, w.worst_approximate_in_topk)
        sys.exit(1)


while time.time() < t_end:
    M = int(rng.choice([8, 16, 16, 32, 48, 64, 96, 128, 192] if not MOCK else [8, 16, 16, 32, 48]))
    D = 8 * M
    if rng.random() < 0.3:                                   # any other quantizer: ragged / small / odd geometries (generic kernels)
        D = int(rng.integers(6, 260))
        M = int(rng.integers(1, min(D, 40) + 1))
    N = int(rng.integers(260, 700 if MOCK else 2500))
    deg = int(rng.choice([8, 16, 24, 32, 40, 64, 80, 130]))
    levels = int(rng.integers(1, 4))
    v, lv, entry, entry_level, cb, q = build_problem(int(rng.integers(1 << 30)), N=N, D=D, M=M, deg=deg, top_n=max(12, N // 20), top_deg=min(8, deg),
                                                     levels=levels)
    if rng.random() < 0.25:                                  # duplicates: exact-score ties in the rerank
        v[1::2] = v[0:-1:2][: len(v[1::2])]
    nq = int(rng.integers(1, 5 if MOCK else 13))
    q = q[:nq]
    use_fused = bool(rng.random() < 0.6)
    rerank = bool(rng.random() < 0.7)
    accept = (rng.random((nq, N)) < rng.uniform(0.3, 0.95)) if rng.random() < 0.35 else None
    opts = {}
    if rng.random() < 0.5:
        opts["gs_cand_cap"] = int(rng.choice([64, 128, 256]))
    if rng.random() < 0.3:
        opts["gs_push_log_cap"] = int(rng.choice([64, 256, 1024]))
    if rng.random() < 0.3:
        opts["gs_vcap_log2"] = int(rng.choice([8, 9, 10, 12]))
    if rng.random() < 0.2:
        opts["gs_v1_log2"] = int(rng.choice([0, 8, 10]))
    if rng.random() < 0.3:                                   # the generic kernels on a shape that has a specialised build
        opts["gs_generic"] = 1
    for name in OPTS:
        ctx.set_option(name, opts.get(name))
    opq = O.OraclePQ(D, M, cb)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    vs = J.VectorSet(ctx, v)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    codes = cv.get(0, N)
    og = O.OracleGraph(N, lv, entry, entry_level)
    graph = J.GraphIndex(ctx, N, lv, entry, entry_level)
    fused = J.FusedPQ(ctx, pq, fused_blocks(codes, lv[0][1]), lv[0][1]) if use_fused else None
    ctx.reset_stats()
    for vsf in rng.permutation(len(VSF))[: int(rng.integers(1, 4))]:
        vsf = VSF[int(vsf)]
        lvl = np.sort(np.stack([opq.adc_scores(q[i], int(vsf), codes, None, fused=use_fused) for i in range(nq)]), axis=1)

        def level(rank):
            return float(np.median(lvl[:, -min(N, rank)]))

        s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs if rerank else None, max_queries=16)
        oracles = [og.searcher(opq, codes, v if rerank else None, int(vsf), fused=use_fused) for _ in range(nq)]
        for _ in range(int(rng.integers(1, 4))):            # several search() calls on one searcher object: each restarts the state
            kind = rng.random()
            if kind < 0.35:
                top_k = int(rng.integers(1, 40))
                rk = top_k + int(rng.integers(0, 80))
                thr = 0.0
            elif kind < 0.7:                                # threshold search: everything above a score level
                top_k = rk = N
                thr = level(int(rng.integers(20, 200)))
            else:
                top_k = int(rng.integers(50, 400))
                rk = top_k
                thr = level(int(rng.integers(10, 80)))
            floor = level(int(rng.integers(5, 60))) if rng.random() < 0.4 else (9.0 if rng.random() < 0.1 else 0.0)
            got = s.search_ex(q, vsf, top_k, rk, threshold=thr, rerank_floor=floor, accept=accept)
            calls += 1
            tag = dict(seed=seed, case=cases, D=D, M=M, N=N, deg=deg, levels=levels, fused=use_fused, rerank=rerank, vsf=str(vsf),
                       accept=accept is not None, opts=opts, top_k=top_k, rk=rk, thr=thr, floor=floor)
            for i in range(nq):
                same(got[i], oracles[i].search(q[i], top_k, rk, thr, floor, accept=None if accept is None else accept[i]), (tag, "q", i))
                checks += 1
            chain = int(rng.choice([0, 1, 2, 3, 5, 9]))
            for c in range(chain):
                add_k = int(rng.integers(1, 40))
                rk2 = add_k + int(rng.integers(0, 60))
                gr = s.resume(add_k, rk2)
                calls += 1
                for i in range(nq):
                    same(gr[i], oracles[i].resume(add_k, rk2), (tag, "resume", c, add_k, rk2, "q", i))
                    checks += 1
        for o in oracles:
            o.close()
        s.close()
    dev_calls += ctx.stat("gs_session_calls_device")
    dev_resumes += ctx.stat("gs_session_resume_device")
    host_calls += ctx.stat("gs_calls_host")
    graph.close()
    cases += 1
print(f"fuzz_searcher: {cases} problems, {calls} search/resume calls, {checks} per-query results identical to the oracle "
      f"({dev_calls} session-kernel calls of which {dev_resumes} resumes replayed in the kernel; {host_calls} host-searcher calls)")
