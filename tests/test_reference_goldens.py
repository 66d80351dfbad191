"""Literal goldens from a REAL JVector — the pin the oracle has been missing (VERDICT r2 "parity unpinned").

`jvector-native-hip/src/test/java/.../GoldenDump.java` runs the reference's own classes (scalar provider) on seeded inputs and
writes `tests/golden/ref/jvector_goldens.bin`: PQ wire bytes, PQ code bytes, ADC / direct / diversity scores as raw float bits,
robust-prune selections, a graph the reference built with its search results and counters, and an on-disk index (v6, FusedPQ +
inline vectors) with the fused scores the reference reads back out of it.  This image has no JDK, so the file cannot be produced
here; two commands produce it anywhere a JDK 22 + Maven exist (INTEGRATION.md "Pinning the oracle"):

    mvn -q -pl jvector-native-hip -am test-compile
    mvn -q -pl jvector-native-hip exec:java -Dexec.args="$REPO/tests/golden/ref/jvector_goldens.bin"

With the file present, `test_oracle_matches_reference_goldens` (CPU) and `test_hip_matches_reference_goldens` (-m gpu) check the
oracle AND the HIP path against the reference's literals; absent, they skip with that reason.  So that the checking code itself
cannot rot, `test_golden_checks_run_on_an_oracle_made_file` builds the same container from the ORACLE's outputs (every record
GoldenDump writes, same names and shapes, the graph hand-made instead of reference-built), round-trips it through the container
format and runs the very same checker on it."""
import os
import struct

import numpy as np
import pytest

from oracle import oracle as O
from test_graph_search import build_problem, fused_blocks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_FILE = os.path.join(ROOT, "tests", "golden", "ref", "jvector_goldens.bin")
VSFS = (("EUCLIDEAN", O.EUCLIDEAN), ("DOT_PRODUCT", O.DOT_PRODUCT), ("COSINE", O.COSINE))
DTYPES = {0: np.uint8, 1: np.int32, 2: np.float32, 3: np.int64}
MISSING = ("tests/golden/ref/jvector_goldens.bin is absent: it has to be produced by GoldenDump.java with a real JVector on a box "
           "with a JDK (this image has none) — until then the oracle is pinned only by the reference's own fixtures (DESIGN.md §2)")


# ---- container ---------------------------------------------------------------------------------------------------
def write_goldens(path, records):
    with open(path, "wb") as f:
        f.write(b"JVGOLD01")
        for name, arr in records.items():
            arr = np.ascontiguousarray(arr)
            code = {np.dtype(np.uint8): 0, np.dtype(np.int32): 1, np.dtype(np.float32): 2, np.dtype(np.int64): 3}[arr.dtype]
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)) + nb + struct.pack("<BI", code, arr.ndim) + struct.pack(f"<{arr.ndim}I", *arr.shape))
            f.write(arr.astype(arr.dtype.newbyteorder("<")).tobytes())


def read_goldens(path):
    data = open(path, "rb").read()
    assert data[:8] == b"JVGOLD01", "not a GoldenDump file"
    pos, out = 8, {}
    while pos < len(data):
        (n,) = struct.unpack_from("<I", data, pos)
        name = data[pos + 4:pos + 4 + n].decode()
        pos += 4 + n
        code, ndim = struct.unpack_from("<BI", data, pos)
        pos += 5
        dims = struct.unpack_from(f"<{ndim}I", data, pos)
        pos += 4 * ndim
        dt = np.dtype(DTYPES[code]).newbyteorder("<")
        cnt = int(np.prod(dims)) if ndim else 1
        out[name] = np.frombuffer(data, dt, cnt, pos).reshape(dims).astype(DTYPES[code])
        pos += cnt * dt.itemsize
    return out


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _same_floats(got, want, exact, what):
    """bit-exact under the scalar (Default) provider — the arithmetic the oracle restates; the SIMD providers differ in the last
    ulps by design, so a file dumped under one of them is held to the reference's own 1e-4 instead"""
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    if exact:
        assert np.array_equal(_bits(got), _bits(want)), f"{what}: {int((_bits(got) != _bits(want)).sum())} of {got.size} floats differ in their bits"
    else:
        assert np.allclose(got, want, rtol=1e-4, atol=1e-5, equal_nan=True), what


def _levels(g):
    entry, entry_level, max_level = (int(x) for x in g["graph_entry"])
    levels = []
    for lvl in range(max_level + 1):
        ids, nb = g[f"graph_nodes_l{lvl}"], g[f"graph_nbrs_l{lvl}"]
        if lvl == 0:
            assert np.array_equal(ids, np.arange(len(ids))), "level 0 must hold every node"
        levels.append((None if lvl == 0 else ids.astype(np.int32), nb.astype(np.int32)))
    return levels, entry, entry_level


# ---- the checker: `score` abstracts who computes (the oracle, or the HIP library through the C ABI) ------------------------
class OracleSide:
    def __init__(self, g):
        self.pq, _, _, _ = O.OraclePQ.parse(g["pq_bytes"].tobytes())
        self.pq.cache_self_magnitudes()

    def encode(self, vectors):
        return self.pq.encode_all(vectors)

    def adc(self, q, vsf, codes):
        return np.stack([self.pq.adc_scores(qq, vsf, codes) for qq in q])

    def direct(self, q, vsf, codes):
        return np.array([[self.pq.direct_score(qq, vsf, c) for c in codes] for qq in q], np.float32)

    def diversity(self, vsf, codes, node1):
        tri = self.pq.codebook_partial_sums(vsf)
        return np.array([[self.pq.diversity_score(tri, vsf, codes[a], c) for c in codes] for a in node1], np.float32)

    def retain(self, vsf, codes, cand, scores, deg):
        tri = self.pq.codebook_partial_sums(vsf)
        sel = np.zeros(cand.shape, np.uint8)
        for p in range(cand.shape[0]):
            sel[p] = self.pq.retain_diverse(tri, vsf, codes, cand[p], scores[p], deg, 0, 1.2)[0]
        return sel

    def search(self, levels, entry, entry_level, codes, vectors, q, vsf, top_k, rerank_k, fused):
        og = O.OracleGraph(codes.shape[0], levels, entry, entry_level)
        return og.search(self.pq, codes, vectors, q, vsf, top_k, rerank_k, fused=fused)

    def fused_scores(self, q, vsf, blocks_codes):
        """blocks_codes [n, M]: the codes a fused block holds for its neighbour slots -> FusedPQDecoder arithmetic"""
        return np.stack([self.pq.adc_scores(qq, vsf, blocks_codes, fused=True) for qq in q])

    def nvq_encode(self, vectors, S):
        o = O.OracleNVQ.compute(vectors, S)
        b, p = o.encode_all(vectors)
        return o.mean, b, p

    def nvq_scores(self, mean, S, b, p, q, vsf):
        ords = np.tile(np.arange(len(b), dtype=np.int32), (len(q), 1))
        return O.OracleNVQ(mean, S).set_rows(b, p).scores(q, vsf, ords)


def check_goldens(g, side, exact):
    N, D, M, Q, DEG, BEAM, TOPK, RERANK = (int(x) for x in g["shape"])
    vectors, queries, codes = g["vectors"], g["queries"], g["codes"]
    assert vectors.shape == (N, D) and queries.shape == (Q, D) and codes.shape == (N, M)
    # row 3: ProductQuantization.encode — code bytes (integer work: exact whatever the provider... as long as no distance ties flip)
    got = side.encode(vectors)
    assert np.array_equal(got, codes) or (not exact and (got != codes).mean() < 1e-3), "PQ code bytes"
    node1 = g["div_node1"]
    for name, vsf in VSFS:
        _same_floats(side.adc(queries, vsf, codes), g[f"adc_{name}"], exact, f"PQDecoder scores {name}")          # rows 2, 5, 6
        _same_floats(side.direct(queries[:4], vsf, codes), g[f"direct_{name}"][:4], exact, f"scoreFunctionFor {name}")
        _same_floats(side.diversity(vsf, codes, node1[:8]), g[f"diversity_{name}"][:8], exact, f"diversityFunctionFor {name}")
        sel = side.retain(vsf, codes, g[f"rd_cand_{name}"], g[f"rd_scores_{name}"], DEG)                            # f.2 robust prune
        assert np.array_equal(sel.astype(bool), g[f"rd_selected_{name}"].astype(bool)), f"retainDiverse selections {name}"
    levels, entry, entry_level = _levels(g)
    for name, vsf in VSFS:                                                                                          # f.1 GraphSearcher
        ids, sc, st = side.search(levels, entry, entry_level, codes, vectors, queries, vsf, TOPK, RERANK, fused=False)
        assert np.array_equal(np.asarray(ids), g[f"search_ids_{name}"]), f"search ids {name}"
        _same_floats(sc, g[f"search_scores_{name}"], exact, f"search scores {name}")
        assert np.array_equal(np.asarray(st)[:, :2], g[f"search_counters_{name}"][:, :2]), f"visited / expanded counters {name}"
    # f.4: the on-disk index the reference wrote — adjacency, vectors, fused blocks, and the fused scores it reads back
    from jvector_amd import formats as F
    od = F.read_odgi(g["odgi_bytes"].tobytes())
    assert od.dimension == D and od.entry_node == entry and od.entry_level == entry_level
    assert np.array_equal(od.levels[0][1], levels[0][1]) and np.array_equal(od.vectors, vectors)
    for lvl in range(1, len(levels)):
        assert np.array_equal(od.levels[lvl][0], levels[lvl][0]) and np.array_equal(od.levels[lvl][1], levels[lvl][1])
    assert od.pq_bytes == g["pq_bytes"].tobytes()[:len(od.pq_bytes)]
    nb0 = levels[0][1]
    want_blocks = fused_blocks(codes, nb0).reshape(N, nb0.shape[1], M)
    assert np.array_equal(od.fused_blocks, want_blocks), "FusedPQ.writeInline blocks"
    assert np.array_equal(F.pqvectors_codes(g["pqvectors_bytes"].tobytes()), codes)
    origins = g["fused_origins"]
    for name, vsf in VSFS:
        want = g[f"fused_scores_{name}"]                                  # [Q, origins, DEG], -inf where the slot is empty
        for oi, o in enumerate(origins):
            d = int((nb0[o] >= 0).sum())
            got = side.fused_scores(queries, vsf, od.fused_blocks[o][:d])
            _same_floats(got, want[:, oi, :d], exact, f"FusedPQDecoder.similarityToNeighbor {name} origin {o}")          # row 7
            assert np.all(np.isneginf(want[:, oi, d:]))
    # NVQ (records added in round 3; a file dumped before that simply lacks them): mean, bytes + parameters, NVQScorer scores, blob
    for S in (1, 3):
        if f"nvq_mean_s{S}" not in g:
            continue
        mean, b, p = side.nvq_encode(vectors, S)
        _same_floats(mean, g[f"nvq_mean_s{S}"], exact, f"NVQuantization.compute global mean S={S}")
        if exact:
            assert np.array_equal(b, g[f"nvq_bytes_s{S}"]), f"NVQ bytes S={S}"
        else:
            assert (np.asarray(b) != g[f"nvq_bytes_s{S}"]).mean() < 1e-2, f"NVQ bytes S={S}"
        _same_floats(np.asarray(p).reshape(N, S, 4), g[f"nvq_params_s{S}"], exact, f"NVQ parameters S={S}")
        for name, vsf in VSFS:   # scored from the REFERENCE's rows, so that a byte that differs above does not cascade
            got = side.nvq_scores(g[f"nvq_mean_s{S}"], S, g[f"nvq_bytes_s{S}"], g[f"nvq_params_s{S}"], queries, vsf)
            _same_floats(got, g[f"nvq_scores_{name}_s{S}"], exact, f"NVQScorer {name} S={S}")
        m2, S2, b2, p2 = F.read_nvqvectors(g[f"nvqvectors_bytes_s{S}"].tobytes())
        assert S2 == S and np.array_equal(b2, g[f"nvq_bytes_s{S}"]) and np.array_equal(_bits(p2), _bits(g[f"nvq_params_s{S}"]))
        assert np.array_equal(_bits(m2), _bits(g[f"nvq_mean_s{S}"]))


# ---- an oracle-made file with GoldenDump's records: keeps the checker honest without a JDK ----------------------------------
def oracle_made_goldens(seed=5):
    N, D, M, Q, DEG, BEAM, TOPK, RERANK = 1200, 64, 8, 6, 16, 40, 10, 40
    v, lv, entry, entry_level, cb, q = build_problem(seed, N=N, D=D, M=M, deg=DEG, levels=2)
    q = q[:Q]
    pq = O.OraclePQ(D, M, cb)
    pq.cache_self_magnitudes()
    codes = pq.encode_all(v)
    side = OracleSide.__new__(OracleSide)
    side.pq = pq
    rec = {"shape": np.array([N, D, M, Q, DEG, BEAM, TOPK, RERANK], np.int32), "provider": np.frombuffer(b"oracle", np.uint8),
           "vectors": v, "queries": q, "pq_bytes": np.frombuffer(pq.serialize(6), np.uint8), "codes": codes}
    from jvector_amd import formats as F
    from oracle import jv_writers as W
    rec["pqvectors_bytes"] = np.frombuffer(W.write_pqvectors(pq.serialize(6), codes), np.uint8)
    node1 = np.array([(p * 61) % N for p in range(32)], np.int32)
    rec["div_node1"] = node1
    for name, vsf in VSFS:
        rec[f"adc_{name}"] = side.adc(q, vsf, codes)
        direct = np.zeros((Q, N), np.float32)
        direct[:4] = side.direct(q[:4], vsf, codes)
        rec[f"direct_{name}"] = direct
        div = np.zeros((len(node1), N), np.float32)
        div[:8] = side.diversity(vsf, codes, node1[:8])
        rec[f"diversity_{name}"] = div
        C = 48
        tri = pq.codebook_partial_sums(vsf)
        cand, sc = np.zeros((len(node1), C), np.int32), np.zeros((len(node1), C), np.float32)
        for p, a in enumerate(node1):
            s = np.array([pq.diversity_score(tri, vsf, codes[a], c) for c in codes], np.float32)
            s[a] = -np.inf
            order = np.lexsort((np.arange(N), -s))[:C]        # NodeArray order: score descending, ties by insertion (node) order
            cand[p], sc[p] = order, s[order]
        rec[f"rd_cand_{name}"], rec[f"rd_scores_{name}"] = cand, sc
        rec[f"rd_selected_{name}"] = side.retain(vsf, codes, cand, sc, DEG)
        rec[f"rd_short_edges_{name}"] = np.zeros(len(node1), np.float32)
    rec["graph_entry"] = np.array([entry, entry_level, len(lv) - 1], np.int32)
    for lvl, (ids, nb) in enumerate(lv):
        rec[f"graph_nodes_l{lvl}"] = np.arange(N, dtype=np.int32) if ids is None else ids.astype(np.int32)
        rec[f"graph_nbrs_l{lvl}"] = nb.astype(np.int32)
    for name, vsf in VSFS:
        ids, scs, st = side.search(lv, entry, entry_level, codes, v, q, vsf, TOPK, RERANK, fused=False)
        rec[f"search_ids_{name}"], rec[f"search_scores_{name}"] = np.asarray(ids, np.int32), np.asarray(scs, np.float32)
        rec[f"search_counters_{name}"] = np.concatenate([np.asarray(st, np.int64)[:, :2], np.zeros((Q, 2), np.int64)], 1).astype(np.int32)
    blocks = fused_blocks(codes, lv[0][1]).reshape(N, DEG, M)
    rec["odgi_bytes"] = np.frombuffer(F.write_odgi(D, lv, entry, vectors=v, fused_blocks=blocks.reshape(N, DEG * M), pq_block=pq.serialize(6),
                                                   hierarchy_codes=codes[lv[1][0]]), np.uint8)
    origins = np.array([(i * 83) % N for i in range(24)], np.int32)
    rec["fused_origins"] = origins
    for name, vsf in VSFS:
        fs = np.full((Q, len(origins), DEG), -np.inf, np.float32)
        for oi, o in enumerate(origins):
            d = int((lv[0][1][o] >= 0).sum())
            fs[:, oi, :d] = side.fused_scores(q, vsf, blocks[o][:d])
        rec[f"fused_scores_{name}"] = fs
    for S in (1, 3):
        mean, b, p = side.nvq_encode(v, S)
        rec[f"nvq_mean_s{S}"], rec[f"nvq_bytes_s{S}"], rec[f"nvq_params_s{S}"] = mean, b, np.asarray(p, np.float32).reshape(N, S, 4)
        for name, vsf in VSFS:
            rec[f"nvq_scores_{name}_s{S}"] = side.nvq_scores(mean, S, b, p, q, vsf)
        rec[f"nvqvectors_bytes_s{S}"] = np.frombuffer(W.write_nvqvectors(mean, S, b, np.asarray(p).reshape(N, S, 4)), np.uint8)
    return rec


def test_golden_checks_run_on_an_oracle_made_file(tmp_path):
    rec = oracle_made_goldens()
    path = tmp_path / "oracle_made.bin"
    write_goldens(path, rec)
    g = read_goldens(path)
    assert set(g) == set(rec) and all(np.array_equal(g[k], rec[k], equal_nan=False) or g[k].dtype == np.float32 for k in rec)
    for k in rec:                                # floats: raw bits survive the container (NaN payloads, -inf, -0.0)
        if g[k].dtype == np.float32:
            assert np.array_equal(_bits(g[k]), _bits(rec[k])), k
    check_goldens(g, OracleSide(g), exact=True)
    g["codes"] = g["codes"].copy()
    g["codes"][3, 2] ^= 1                        # and the checker does notice a single flipped code bit
    with pytest.raises(AssertionError):
        check_goldens(g, OracleSide(g), exact=True)


def _load_real():
    if not os.path.exists(GOLDEN_FILE):
        pytest.skip(MISSING)
    g = read_goldens(GOLDEN_FILE)
    provider = g["provider"].tobytes().decode()
    return g, provider.startswith("Default")


def test_oracle_matches_reference_goldens():
    g, exact = _load_real()
    check_goldens(g, OracleSide(g), exact)


class HipSide:
    """the same quantities through the C ABI on the GPU"""

    def __init__(self, g):
        import jvector_amd as J
        self.J, self.ctx = J, J.HipContext(0)
        self.pq = J.ProductQuantization.load(self.ctx, g["pq_bytes"].tobytes())
        self.N = int(g["shape"][0])

    def _cv(self, codes):
        return self.J.PQVectors(self.ctx, self.pq, np.ascontiguousarray(codes))

    def encode(self, vectors):
        vs = self.J.VectorSet(self.ctx, vectors)
        return self.J.PQVectors.encode_and_build(self.ctx, self.pq, vs).get(0, vectors.shape[0])

    def _scan(self, q, vsf, codes, kind):
        J = self.J
        t = J.QueryTables(self.ctx, self.pq, len(q))
        t.build(np.ascontiguousarray(q), J.VectorSimilarityFunction(vsf), kind)
        cv = self._cv(codes)
        return np.asarray(J.ApproximateScoreFunction(cv, t).similarity_to_range(0, codes.shape[0], like=np.empty(0, np.float32)))

    def adc(self, q, vsf, codes):
        return self._scan(q, vsf, codes, self.J.DecoderKind.PQ)

    def direct(self, q, vsf, codes):
        ords = np.tile(np.arange(codes.shape[0], dtype=np.int32), (len(q), 1))
        return np.asarray(self._cv(codes).direct_scores(np.ascontiguousarray(q), self.J.VectorSimilarityFunction(vsf), ords))

    def diversity(self, vsf, codes, node1):
        bsp = self.J.PQBuildScoreProvider(self.ctx, self._cv(codes), self.J.VectorSimilarityFunction(vsf))
        lst = np.tile(np.arange(codes.shape[0], dtype=np.int32), (len(node1), 1))
        return np.asarray(bsp.diversity_scores(np.asarray(node1, np.int32), lst))

    def retain(self, vsf, codes, cand, scores, deg):
        bsp = self.J.PQBuildScoreProvider(self.ctx, self._cv(codes), self.J.VectorSimilarityFunction(vsf))
        sel, _, _ = bsp.retain_diverse(cand, scores, deg, 1.2)
        mask = np.zeros(cand.shape, np.uint8)
        sel = np.asarray(sel)
        for p in range(cand.shape[0]):
            mask[p, sel[p][sel[p] >= 0]] = 1
        return mask

    def search(self, levels, entry, entry_level, codes, vectors, q, vsf, top_k, rerank_k, fused):
        J = self.J
        graph = J.GraphIndex(self.ctx, codes.shape[0], levels, entry, entry_level)
        s = J.GraphSearcher(self.ctx, graph, self.pq, self._cv(codes), None, J.VectorSet(self.ctx, vectors), max_queries=len(q))
        return s.search(q, J.VectorSimilarityFunction(vsf), top_k, rerank_k, return_stats=True)

    def fused_scores(self, q, vsf, blocks_codes):
        return self._scan(q, vsf, blocks_codes, self.J.DecoderKind.FUSED)

    def nvq_encode(self, vectors, S):
        J = self.J
        vs = J.VectorSet(self.ctx, vectors)
        nvq = J.NVQuantization.compute(self.ctx, vs, S)
        b, p = nvq.encode_all(vs).get()
        return nvq.global_mean(), b, p

    def nvq_scores(self, mean, S, b, p, q, vsf):
        J = self.J
        nv = J.NVQVectors(self.ctx, J.NVQuantization.create(self.ctx, np.ascontiguousarray(mean), S), np.ascontiguousarray(b), np.ascontiguousarray(p))
        ords = np.tile(np.arange(len(b), dtype=np.int32), (len(q), 1))
        return np.asarray(nv.scores(np.ascontiguousarray(q), J.VectorSimilarityFunction(vsf), ords))


@pytest.mark.gpu
def test_hip_matches_reference_goldens():
    g, exact = _load_real()
    check_goldens(g, HipSide(g), exact)


@pytest.mark.gpu
def test_hip_passes_the_golden_checker_on_the_oracle_made_file():
    """without the reference's file the HIP path still goes through the same checker, against the oracle-made records"""
    g = oracle_made_goldens()
    check_goldens(g, HipSide(g), exact=True)
