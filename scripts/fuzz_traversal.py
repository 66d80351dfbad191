"""One-off validation aid: a time-boxed differential sweep of the device traversal (and the host searcher) against the oracle's
GraphSearcher restatement on random problems — shapes (specialised and generic kernels), degrees, level counts, similarity functions, fused / unfused, rerank /
no rerank, acceptOrds filters, duplicated vectors (exact-score ties), tiny visited tables (growth / retry / host fallback).
Every case must agree bit for bit on ids, scores and the visited / expanded counters.
usage (GPU box): python scripts/fuzz_traversal.py [seconds] [seed]     (here: FUZZ_MOCK=1 python scripts/fuzz_traversal.py 60)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
MOCK = os.environ.get("FUZZ_MOCK") == "1"       # the same sweep on the CPU mock + lane emulator (slow: small problems)
if MOCK:
    import ctypes as C
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
from test_graph_search import fused_blocks  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = J.HipContext(0)
VSF = J.VectorSimilarityFunction
t_end = time.time() + budget
cases = searches = 0
knobs = ("JVECTOR_HIP_GS_VCAP_LOG2", "JVECTOR_HIP_GS_GROW", "JVECTOR_HIP_GS_RETRY", "JVECTOR_HIP_GS_CAND_CAP", "JVECTOR_HIP_GS_PUSH_LOG_CAP",
         "JVECTOR_HIP_GS_GENERIC", "JVECTOR_HIP_GS_WGX", "JVECTOR_HIP_GS_WGX_WAVES", "JVECTOR_HIP_GS_WGX_SLOTS", "JVECTOR_HIP_GS_WGX_DEPTH",
         "JVECTOR_HIP_GS_WGX_LUT_M", "JVECTOR_HIP_GS_WGX_PER_CU", "JVECTOR_HIP_GS_PAIRC", "JVECTOR_HIP_GS_QUAD", "JVECTOR_HIP_GS_UBR", "JVECTOR_HIP_GS_UBRC",
         "JVECTOR_HIP_GS_UBR_TRIM", "JVECTOR_HIP_GS_DEFER", "JVECTOR_HIP_GS_DEFER_MIN_LEVEL")
wgx_searches = pairc_searches = ubr_searches = 0
defer0 = (ctx.stat("gs_deferred"), ctx.stat("gs_defer_restarts"))
while time.time() < t_end:
    D = int(rng.choice([128, 256, 384, 512, 768, 768, 768] if not MOCK else [128, 256]))   # (768 = PQ-96: the register-table bound forms)
    M = D // 8
    if rng.random() < 0.3:                                   # any other quantizer (ragged / small / odd): the generic kernels
        D = int(rng.integers(6, 260))
        M = int(rng.integers(1, min(D, 40) + 1))
    N = int(rng.integers(200, 900 if MOCK else 6000))
    deg = int(rng.choice([8, 16, 24, 32, 48, 64, 72, 100, 130]))   # (> 64: rows walked 64 neighbours at a time)
    n_levels = int(rng.integers(1, 4))
    base = rng.standard_normal((N, D)).astype(np.float32)
    if rng.random() < 0.3:                                   # duplicates: exact-score ties
        base[N // 2:] = base[: N - N // 2]
    if rng.random() < 0.5:
        base /= np.linalg.norm(base, axis=1, keepdims=True)
    perm = rng.permutation(N)
    v = np.ascontiguousarray(base[perm])
    # neighbour rows: mostly near vectors of a random projection bucket + random long edges, ragged, packed
    key = v @ rng.standard_normal((D, 3)).astype(np.float32)
    order = np.lexsort((key[:, 2], key[:, 1], key[:, 0]))
    pos = np.empty(N, np.int64)
    pos[order] = np.arange(N)
    nb = np.full((N, deg), -1, np.int32)
    for i in range(N):
        d = int(rng.integers(1, deg + 1))
        near = order[np.clip(pos[i] + rng.integers(-deg, deg + 1, d), 0, N - 1)]
        far = rng.integers(0, N, max(1, d // 4))
        row = np.concatenate([near[: d - len(far)], far])
        row = row[row != i]
        _, first = np.unique(row, return_index=True)
        row = row[np.sort(first)]
        nb[i, : len(row)] = row
    lv = [(None, nb)]
    entry, entry_level = int(rng.integers(0, N)), 0
    prev = np.arange(N)
    for _ in range(1, n_levels):
        cnt = max(3, len(prev) // int(rng.integers(4, 20)))
        ids = np.sort(rng.choice(prev, cnt, replace=False)).astype(np.int32)
        udeg = int(rng.choice([4, 8, 16, 32]))
        un = np.full((cnt, udeg), -1, np.int32)
        for i in range(cnt):
            r = rng.choice(ids, int(rng.integers(1, min(udeg, cnt - 1) + 1)), replace=False)
            r = r[r != ids[i]]
            un[i, : len(r)] = r
        lv.append((ids, un))
        entry, entry_level, prev = int(ids[rng.integers(0, cnt)]), len(lv) - 1, ids
    sizes, offs = O.subvector_sizes_offsets(D, M)
    pick = rng.integers(0, N, 256)
    cb = np.concatenate([v[pick, offs[m]: offs[m] + sizes[m]].reshape(-1) for m in range(M)])
    opq = O.OraclePQ(D, M, cb)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    vs = J.VectorSet(ctx, v)
    cv = J.PQVectors.encode_and_build(ctx, pq, vs)
    codes = cv.get(0, N)
    og = O.OracleGraph(N, lv, entry, entry_level)
    use_fused = bool(rng.random() < 0.6)
    fused = J.FusedPQ(ctx, pq, fused_blocks(codes, nb), nb) if use_fused else None
    Q = int(rng.integers(1, 48))
    q = (v[rng.integers(0, N, Q)] + rng.choice([0.0, 0.02, 0.3]) * rng.standard_normal((Q, D))).astype(np.float32)
    for trial in range(3):
        vsf = VSF(int(rng.integers(0, 3)))
        rk = int(rng.choice([1, 5, 20, 60, 150, 300]))
        top_k = int(rng.integers(1, min(rk, 20) + 1))
        rerank = bool(rng.random() < 0.7)
        accept = None if rng.random() < 0.6 else (rng.random(N) < 0.7 if rng.random() < 0.5 else rng.random((Q, N)) < 0.5)
        traversal = "device" if rng.random() < 0.85 else "host"
        env = {}
        if traversal == "device" and rng.random() < 0.4:
            env["JVECTOR_HIP_GS_VCAP_LOG2"] = str(int(rng.integers(8, 11)))
            env["JVECTOR_HIP_GS_GROW"] = str(int(rng.integers(0, 2)))
            env["JVECTOR_HIP_GS_RETRY"] = str(int(rng.integers(0, 2)))
        if traversal == "device" and rng.random() < 0.3:
            env["JVECTOR_HIP_GS_CAND_CAP"] = "256"
        if traversal == "device" and rng.random() < 0.25:   # the generic kernels on a shape that has a specialised build
            env["JVECTOR_HIP_GS_GENERIC"] = "1"
        if traversal == "device" and rng.random() < 0.2:
            env["JVECTOR_HIP_GS_PUSH_LOG_CAP"] = str(int(rng.choice([4, 16, 64])))
        # round 4: the workgroup form (one query per workgroup, ADC table in LDS, control wave + expanders) — forced with random
        # launch shapes / slot counts / request depths / partial tables, or forbidden; left alone, AUTO picks it for these batch sizes
        r = rng.random()
        if traversal == "device" and r < 0.4:
            env["JVECTOR_HIP_GS_WGX"] = "1"
            env["JVECTOR_HIP_GS_WGX_WAVES"] = str(int(rng.choice([2, 3, 4] if MOCK else [2, 3, 4, 6, 8])))
            env["JVECTOR_HIP_GS_WGX_SLOTS"] = str(int(rng.choice([2, 3, 8, 16, 40])))
            env["JVECTOR_HIP_GS_WGX_DEPTH"] = str(int(rng.integers(0, 2)))
            if rng.random() < 0.4 and M % 16 == 0 and M >= 32:
                env["JVECTOR_HIP_GS_WGX_LUT_M"] = str(int(rng.integers(1, M // 16 + 1)) * 16)
                env["JVECTOR_HIP_GS_WGX_PER_CU"] = str(int(rng.integers(1, 4)))
        elif traversal == "device" and r < 0.5:
            env["JVECTOR_HIP_GS_WGX"] = "0"
        elif traversal == "device" and r < 0.85:
            # round 4, the one-wave kernels' lane assignments: the compacted pair form (rows of 33 ... 64 neighbours, codes by ordinal)
            # on / off, four lanes per neighbour in the pair kernels' short expansions on / off
            env["JVECTOR_HIP_GS_WGX"] = "0"
            env["JVECTOR_HIP_GS_PAIRC"] = str(int(rng.integers(0, 2)))
            env["JVECTOR_HIP_GS_QUAD"] = str(int(rng.integers(0, 2)))
            # round 5: the register-table bound forms (PQ-96, dot product / cosine: over the row, and over the compacted fresh list) on / off,
            # trims every 1 ... 64 pushes
            env["JVECTOR_HIP_GS_UBR"] = str(int(rng.random() < 0.8))
            env["JVECTOR_HIP_GS_UBRC"] = str(int(rng.random() < 0.8))
            env["JVECTOR_HIP_GS_UBR_TRIM"] = str(int(rng.choice([1, 8, 48, 64])))
            env["JVECTOR_HIP_GS_DEFER"] = str(int(rng.random() < 0.85))                # deferred scores above level 0 (round 6)
            env["JVECTOR_HIP_GS_DEFER_MIN_LEVEL"] = str(int(rng.choice([1, 1, 2])))    # (from level 1: restarts are common on these graphs)
        for k in knobs:
            os.environ.pop(k, None)
        os.environ.update(env)
        graph = J.GraphIndex(ctx, N, lv, entry, entry_level).set_traversal(traversal)
        s = J.GraphSearcher(ctx, graph, pq, cv, fused, vs if rerank else None, max_queries=64)
        try:
            ids, sc, st = s.search(q, vsf, top_k, rk, return_stats=True, accept=accept)
        except J.UnsupportedError:
            graph.close()
            continue
        wi, ws, wst = og.search(opq, codes, v if rerank else None, q, int(vsf), top_k, rk, fused=use_fused, accept=accept)
        ok = np.array_equal(st, wst) and np.array_equal(ids, wi) and (np.array_equal(sc, ws) or np.array_equal(np.nan_to_num(sc, nan=-7.0), np.nan_to_num(ws, nan=-7.0)))
        if not ok:
            print("MISMATCH", dict(seed=seed, case=cases, D=D, N=N, deg=deg, levels=n_levels, vsf=str(vsf), rk=rk, top_k=top_k, rerank=rerank,
                                   fused=use_fused, traversal=traversal, accept=None if accept is None else accept.shape, env=env))
            bad = np.where((ids != wi).any(axis=1) | (st != wst).any(axis=1))[0]
            print(" queries", bad[:5], "\n got", ids[bad[:2]], "\n want", wi[bad[:2]], "\n stats", st[bad[:2]], wst[bad[:2]])
            sys.exit(1)
        searches += 1
        wgx_searches += ctx.stat("gs_last_wgx") if traversal == "device" else 0
        pairc_searches += int(traversal == "device" and ctx.stat("gs_last_wgx") == 0 and ctx.stat("gs_last_pair") == 2)
        ubr_searches += int(traversal == "device" and ctx.stat("gs_last_wgx") == 0 and ctx.stat("gs_last_ubr") == 1)
        graph.close()
    cases += 1
for k in knobs:
    os.environ.pop(k, None)
print(f"fuzz: deferred neighbours {ctx.stat('gs_deferred') - defer0[0]}, queries started over {ctx.stat('gs_defer_restarts') - defer0[1]}")
print(f"fuzz: {cases} random problems, {searches} searches ({wgx_searches} through the workgroup form, {pairc_searches} through the compacted pair form, {ubr_searches} through a register-table bound form), all bit-identical to the oracle (seed {seed}, {budget:.0f} s)")
