"""ctypes binding for the CPU oracle (oracle/libjv_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under jvector_amd/ may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libjv_oracle.so")

EUCLIDEAN, DOT_PRODUCT, COSINE = 0, 1, 2
VSF_NAMES = {EUCLIDEAN: "EUCLIDEAN", DOT_PRODUCT: "DOT_PRODUCT", COSINE: "COSINE"}


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("jv_oracle.c", "jv_oracle_simd.c", "jv_nvq.c", "jv_oracle.h", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "libjv_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


class _PQ(C.Structure):
    _fields_ = [("D", C.c_int), ("M", C.c_int), ("k", C.c_int),
                ("sizes", C.POINTER(C.c_int)), ("offsets", C.POINTER(C.c_int)),
                ("codebooks", C.POINTER(C.c_float)), ("centroid", C.POINTER(C.c_float)),
                ("self_magnitudes", C.POINTER(C.c_float))]


class _Layout(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("fullChunkVectors", "lastChunkVectors", "fullSizeChunks",
                                       "totalChunks", "fullChunkBytes", "lastChunkBytes")]


class _Graph(C.Structure):
    _fields_ = [("n_nodes", C.c_int64), ("n_levels", C.c_int), ("entry_node", C.c_int32), ("entry_level", C.c_int),
                ("level_count", C.POINTER(C.c_int)), ("level_degree", C.POINTER(C.c_int)),
                ("level_nodes", C.POINTER(C.POINTER(C.c_int32))), ("level_neighbors", C.POINTER(C.POINTER(C.c_int32)))]


def pack_accept_bits(accept):
    """bool [N] or [Q, N] -> uint64 words, bit n of word n // 64 (little-endian bit order), rows padded to whole words"""
    a = np.asarray(accept, bool)
    n = a.shape[-1]
    pad = (-n) % 64
    a = np.pad(a, [(0, 0)] * (a.ndim - 1) + [(0, pad)])
    return np.ascontiguousarray(np.packbits(a, axis=-1, bitorder="little").view(np.uint64))


class OracleGraph:
    """levels: list of (node_ids | None, neighbors[n_l, degree_l] int32 packed, -1 padded); level 0 first."""

    def __init__(self, n_nodes, levels, entry_node, entry_level):
        self.levels = [(None if ids is None else np.ascontiguousarray(ids, np.int32),
                        np.ascontiguousarray(nb, np.int32)) for ids, nb in levels]
        L = len(self.levels)
        self._count = (C.c_int * L)(*[nb.shape[0] for _, nb in self.levels])
        self._deg = (C.c_int * L)(*[nb.shape[1] for _, nb in self.levels])
        i32p = C.POINTER(C.c_int32)
        self._nodes = (i32p * L)(*[C.cast(None, i32p) if ids is None else ids.ctypes.data_as(i32p)
                                   for ids, _ in self.levels])
        self._nbrs = (i32p * L)(*[nb.ctypes.data_as(i32p) for _, nb in self.levels])
        self._s = _Graph(n_nodes, L, entry_node, entry_level, self._count, self._deg, self._nodes, self._nbrs)

    def search(self, pq, codes, vecs, queries, vsf, top_k, rerank_k, fused=False, accept=None):
        """accept: None (Bits.ALL), a bool array [N] shared by all queries, or [Q, N] one filter per query."""
        codes = np.ascontiguousarray(codes, np.uint8)
        queries = f32(queries)
        vecs = None if vecs is None else f32(vecs)
        Q = queries.shape[0]
        ids = np.empty((Q, top_k), np.int32)
        sc = np.empty((Q, top_k), np.float32)
        stats = np.zeros((Q, 2), np.int64)
        L = lib()
        masks = None if accept is None else pack_accept_bits(accept)
        for q in range(Q):
            m = None if masks is None else (masks if masks.ndim == 1 else masks[q]).ctypes.data_as(C.POINTER(C.c_uint64))
            L.jvo_graph_search_filtered(C.byref(self._s), pq.ref, _u8(codes), None if vecs is None else _f(vecs), _f(queries[q]),
                                        vsf, 1 if fused else 0, top_k, rerank_k, m, _i32(ids[q]), _f(sc[q]),
                                        stats[q].ctypes.data_as(C.POINTER(C.c_int64)))
        return ids, sc, stats

    def searcher(self, pq, codes, vecs, vsf, fused=False):
        return OracleSearcher(self, pq, codes, vecs, vsf, fused)


class SearchResult:
    """SearchResult (B/graph/SearchResult.java): nodes best first, the counters, worstApproximateScoreInTopK."""

    def __init__(self, ids, scores, stats, worst):
        self.ids, self.scores = ids, scores
        self.visited, self.expanded, self.expanded_base, self.reranked = (int(x) for x in stats)
        self.worst_approximate_in_topk = float(worst)

    def __len__(self):
        return len(self.ids)


class OracleSearcher:
    """One GraphSearcher instance (jvo_searcher_*): search(...) then any number of resume(...)."""

    def __init__(self, graph, pq, codes, vecs, vsf, fused=False):
        self._keep = (graph, pq, np.ascontiguousarray(codes, np.uint8), None if vecs is None else f32(vecs))
        _, _, c, v = self._keep
        self._top = 0
        self._h = lib().jvo_searcher_new(C.byref(graph._s), pq.ref, _u8(c), None if v is None else _f(v), vsf, 1 if fused else 0)

    def close(self):
        if self._h:
            lib().jvo_searcher_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _result(self, n, top_k, ids, sc, stats, worst):
        if n < 0:
            raise ValueError("illegal argument (rerankK < topK, or resume before search)")
        return SearchResult(ids[:n].copy(), sc[:n].copy(), stats, worst.value)

    def search(self, query, top_k, rerank_k, threshold=0.0, rerank_floor=0.0, accept=None):
        ids, sc = np.empty(max(top_k, 1), np.int32), np.empty(max(top_k, 1), np.float32)
        stats, worst = np.zeros(4, np.int64), C.c_float(0)
        m = None if accept is None else pack_accept_bits(accept)
        n = lib().jvo_searcher_search(self._h, _f(f32(query)), top_k, rerank_k, threshold, rerank_floor,
                                      None if m is None else m.ctypes.data_as(C.POINTER(C.c_uint64)), _i32(ids), _f(sc),
                                      stats.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(worst))
        return self._result(n, top_k, ids, sc, stats, worst)

    def resume(self, additional_k, rerank_k):
        ids, sc = np.empty(max(additional_k, 1), np.int32), np.empty(max(additional_k, 1), np.float32)
        stats, worst = np.zeros(4, np.int64), C.c_float(0)
        n = lib().jvo_searcher_resume(self._h, additional_k, rerank_k, _i32(ids), _f(sc),
                                      stats.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(worst))
        return self._result(n, additional_k, ids, sc, stats, worst)


def percentile_legacy(values, p):
    v = np.ascontiguousarray(values, np.float64)
    return lib().jvo_percentile_legacy(v.ctypes.data_as(C.POINTER(C.c_double)), len(v), float(p))


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        fp, u8p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
        pqp = C.POINTER(_PQ)

        def sig(name, res, *args):
            f = getattr(L, name)
            f.restype = res
            f.argtypes = list(args)

        sig("jvo_dot", C.c_float, fp, fp, C.c_int)
        sig("jvo_dot_off", C.c_float, fp, C.c_int, fp, C.c_int, C.c_int)
        sig("jvo_l2", C.c_float, fp, fp, C.c_int)
        sig("jvo_l2_off", C.c_float, fp, C.c_int, fp, C.c_int, C.c_int)
        sig("jvo_cosine", C.c_float, fp, fp, C.c_int)
        sig("jvo_cosine_off", C.c_float, fp, C.c_int, fp, C.c_int, C.c_int)
        sig("jvo_compare", C.c_float, C.c_int, fp, fp, C.c_int)
        sig("jvo_score_from_raw", C.c_float, C.c_int, C.c_float)
        sig("jvo_assemble_and_sum", C.c_float, fp, C.c_int, u8p, C.c_int, C.c_int)
        sig("jvo_assemble_and_sum_pq", C.c_float, fp, C.c_int, u8p, C.c_int, u8p, C.c_int, C.c_int)
        sig("jvo_calculate_partial_sums", None, fp, C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int, fp)
        sig("jvo_calculate_partial_self_magnitudes", None, fp, C.c_int, C.c_int, C.c_int, fp)
        sig("jvo_pq_decoded_cosine", C.c_float, u8p, C.c_int, C.c_int, C.c_int, fp, fp, C.c_float)
        sig("jvo_subvector_sizes_offsets", None, C.c_int, C.c_int, i32p, i32p)
        sig("jvo_closest_centroid", C.c_int, pqp, fp, C.c_int)
        sig("jvo_pq_encode", None, pqp, fp, u8p)
        sig("jvo_pq_encode_all", None, pqp, fp, C.c_int64, u8p, C.c_int)
        sig("jvo_pq_decode", None, pqp, u8p, fp)
        sig("jvo_pq_codebook_partial_sums", None, pqp, C.c_int, fp)
        sig("jvo_pqdecoder_init", None, pqp, fp, C.c_int, fp, fp, fp)
        sig("jvo_fuseddecoder_init", None, pqp, fp, C.c_int, fp, fp, fp)
        sig("jvo_adc_score", C.c_float, C.c_int, C.c_int, C.c_int, fp, fp, C.c_float, u8p)
        sig("jvo_adc_scores", None, C.c_int, C.c_int, C.c_int, fp, fp, C.c_float, u8p, i32p, C.c_int64, fp)
        sig("jvo_pq_direct_score", C.c_float, pqp, fp, C.c_int, u8p)
        sig("jvo_pq_encode_anisotropic", None, pqp, C.c_float, fp, u8p)
        sig("jvo_pq_train", None, fp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, fp, fp, C.POINTER(C.c_int))
        sig("jvo_pq_refine", None, pqp, fp, C.c_int64, C.c_int, C.c_uint64, fp)
        sig("jvo_pq_train_aniso", None, fp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_int, fp, fp,
            C.POINTER(C.c_int))
        sig("jvo_pq_refine_aniso", None, pqp, C.c_float, fp, C.c_int64, C.c_int, C.c_uint64, fp)
        sig("jvo_parallel_cost_multiplier", C.c_float, C.c_float, C.c_int)
        sig("jvo_pq_diversity_score", C.c_float, fp, C.c_int, C.c_int, C.c_int, u8p, u8p)
        sig("jvo_pq_diversity_score_direct", C.c_float, pqp, C.c_int, u8p, u8p)
        sig("jvo_retain_diverse", C.c_int, fp, C.c_int, C.c_int, C.c_int, u8p, C.POINTER(C.c_int32), fp, C.c_int, C.c_int, C.c_int,
            C.c_float, u8p, C.POINTER(C.c_double))
        sig("jvo_float_to_sortable_int", C.c_int32, C.c_float)
        sig("jvo_sortable_int_to_float", C.c_float, C.c_int32)
        sig("jvo_nodequeue_encode", C.c_int64, C.c_int32, C.c_float)
        sig("jvo_topk", C.c_int, i32p, fp, C.c_int64, C.c_int, i32p, fp)
        sig("jvo_search_flat", None, pqp, u8p, fp, C.c_int64, fp, C.c_int, C.c_int, C.c_int, C.c_int, i32p, fp, C.c_int)
        sig("jvo_graph_search_filtered", None, C.POINTER(_Graph), pqp, u8p, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int,
            C.POINTER(C.c_uint64), i32p, fp, C.POINTER(C.c_int64))
        sig("jvo_graph_search", None, C.POINTER(_Graph), pqp, u8p, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, i32p, fp,
            C.POINTER(C.c_int64))
        sig("jvo_searcher_new", C.c_void_p, C.POINTER(_Graph), pqp, u8p, fp, C.c_int, C.c_int)
        sig("jvo_searcher_free", None, C.c_void_p)
        sig("jvo_searcher_search", C.c_int, C.c_void_p, fp, C.c_int, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_uint64), i32p, fp,
            C.POINTER(C.c_int64), C.POINTER(C.c_float))
        sig("jvo_searcher_resume", C.c_int, C.c_void_p, C.c_int, C.c_int, i32p, fp, C.POINTER(C.c_int64), C.POINTER(C.c_float))
        sig("jvo_percentile_legacy", C.c_double, C.POINTER(C.c_double), C.c_int, C.c_double)
        sig("jvo_set_visit_log", None, i32p, C.c_int64)
        sig("jvo_visit_log_count", C.c_int64)
        sig("jvo_rerank", None, fp, fp, i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i32p, fp, C.c_int)
        sig("jvo_pq_layout_compute", C.c_int, C.c_int, C.c_int, C.POINTER(_Layout))
        sig("jvo_pq_parse", C.c_int, u8p, C.c_size_t, i32p, i32p, i32p, i32p, i32p, fp, i32p, C.c_int,
            fp, fp, C.c_size_t, C.POINTER(C.c_size_t))
        sig("jvo_pq_serialize", C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, i32p, fp, C.c_float, fp,
            u8p, C.c_size_t)
        sig("jvo_make_vec", None, fp, C.c_size_t, C.c_float)
        # cpu_baseline only: SIMD restatement of the reference's native kernels (jv_oracle_simd.c)
        sig("jvs_tier", C.c_int)
        sig("jvs_tier_name", C.c_char_p)
        sig("jvs_dot", C.c_float, fp, fp, C.c_int)
        sig("jvs_l2", C.c_float, fp, fp, C.c_int)
        sig("jvs_cosine", C.c_float, fp, fp, C.c_int)
        sig("jvs_calculate_partial_sums", None, fp, C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int, fp)
        sig("jvs_assemble_and_sum", C.c_float, fp, C.c_int, u8p, C.c_int)
        sig("jvs_pq_decoded_cosine", C.c_float, u8p, C.c_int, C.c_int, fp, fp, C.c_float)
        sig("jvo_set_simd", C.c_int, C.c_int)
        sig("jvo_dense_compare", C.c_float, C.c_int, fp, fp, C.c_int)
        sig("jvo_dense_scan", None, C.c_int, fp, C.c_int, fp, C.c_int64, C.c_int, fp)
        # NVQ (jv_nvq.c)
        F = C.c_float
        sig("jvo_nvq_derive", None, F, F, F, F, F, fp)
        sig("jvo_nvq_min", F, fp, C.c_int)
        sig("jvo_nvq_max", F, fp, C.c_int)
        sig("jvo_nvq_quantize_8bit", None, fp, C.c_int, F, F, F, F, u8p)
        sig("jvo_nvq_loss", F, fp, C.c_int, F, F, F, F, C.c_int)
        sig("jvo_nvq_uniform_loss", F, fp, C.c_int, F, F, C.c_int)
        sig("jvo_nvq_dot_8bit", F, fp, u8p, C.c_int, F, F, F, F)
        sig("jvo_nvq_l2_8bit", F, fp, u8p, C.c_int, F, F, F, F)
        sig("jvo_nvq_cosine_8bit", None, fp, u8p, C.c_int, F, F, F, F, fp, fp)
        sig("jvo_nvq_dequantize", F, C.c_uint8, F, F, F, F)
        sig("jvo_nvq_global_mean", None, fp, C.c_int64, C.c_int, fp)
        sig("jvo_nvq_encode_sub", None, fp, C.c_int, C.c_int, u8p, fp)
        sig("jvo_nvq_growth_grid", C.c_int, fp, fp, i32p, C.c_int)
        sig("jvo_nvq_encode", None, fp, C.c_int, C.c_int, fp, C.c_int, u8p, fp)
        sig("jvo_nvq_encode_all", None, fp, C.c_int, C.c_int, fp, C.c_int64, C.c_int, u8p, fp, C.c_int)
        sig("jvo_nvq_score", F, C.c_int, fp, C.c_int, C.c_int, fp, u8p, fp)
        sig("jvo_nvq_scores", None, C.c_int, fp, C.c_int, C.c_int, fp, C.c_int, u8p, fp, C.c_int64, i32p, C.c_int, fp)
        sig("jvo_nvq_reconstruction_error", C.c_double, fp, C.c_int, C.c_int, fp, C.c_int)
        sig("jvo_set_nvq_reranker", None, u8p, fp, fp, C.c_int, C.c_int)
        sig("jvo_nvq_reranker_active", C.c_int)
        sig("jvo_java_random_seed", None, C.POINTER(C.c_int64), C.c_int64)
        sig("jvo_java_random_next_double", C.c_double, C.POINTER(C.c_int64))
        sig("jvo_random_graph_level", C.c_int, C.POINTER(C.c_int64), C.c_int, C.c_int)
        sig("jvo_builder_new", C.c_void_p, pqp, u8p, fp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int)
        sig("jvo_builder_free", None, C.c_void_p)
        sig("jvo_builder_set_levels", None, C.c_void_p, C.POINTER(C.c_int8))
        sig("jvo_builder_set_deviations", None, C.c_void_p, C.c_int, C.c_int, C.c_int)
        sig("jvo_builder_add", C.c_int, C.c_void_p, C.c_int32)
        sig("jvo_builder_improve", None, C.c_void_p, C.c_int32)
        sig("jvo_builder_enforce_degree", None, C.c_void_p, C.c_int32)
        sig("jvo_builder_cleanup", None, C.c_void_p)
        sig("jvo_builder_row", C.c_int, C.c_void_p, C.c_int, C.c_int32, i32p, fp, C.POINTER(C.c_int))
        sig("jvo_nodearray_merge", C.c_int, i32p, fp, C.c_int, i32p, fp, C.c_int, i32p, fp)
        sig("jvo_nodearray_insert_sorted", C.c_int, i32p, fp, C.POINTER(C.c_int), C.c_int32, C.c_float)
        sig("jvo_builder_info", None, C.c_void_p, i32p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64))
        _lib = L
    return _lib


class OracleNVQ:
    """NVQuantization + NVQVectors + NVQScorer of the scalar reference path (jv_nvq.c).  params[i, s] = {minValue, maxValue,
    growthRate, midpoint}; bytes[i] = the sub-vectors' bytes concatenated."""

    def __init__(self, mean, n_subvectors, learn=True):
        self.mean = f32(mean).copy()
        self.D, self.S, self.learn = int(self.mean.shape[0]), int(n_subvectors), bool(learn)
        self.bytes = self.params = None

    @classmethod
    def compute(cls, vecs, n_subvectors, learn=True):
        """NVQuantization.compute :153-163"""
        vecs = f32(vecs)
        mean = np.empty(vecs.shape[1], np.float32)
        lib().jvo_nvq_global_mean(_f(vecs), vecs.shape[0], vecs.shape[1], _f(mean))
        return cls(mean, n_subvectors, learn)

    def sizes(self):
        return [self.D // self.S + (1 if i < self.D % self.S else 0) for i in range(self.S)]

    def encode_all(self, vecs, nthreads=16):
        vecs = f32(vecs)
        n = vecs.shape[0]
        self.bytes = np.zeros((n, self.D), np.uint8)
        self.params = np.zeros((n, self.S, 4), np.float32)
        lib().jvo_nvq_encode_all(_f(self.mean), self.D, self.S, _f(vecs), n, int(self.learn), _u8(self.bytes), _f(self.params), nthreads)
        return self.bytes, self.params

    def set_rows(self, bytes_, params):
        self.bytes = np.ascontiguousarray(bytes_, np.uint8)
        self.params = np.ascontiguousarray(params, np.float32).reshape(len(self.bytes), self.S, 4)
        return self

    def scores(self, queries, vsf, ordinals):
        queries, ordinals = f32(queries), np.ascontiguousarray(ordinals, np.int32)
        Q, B = ordinals.shape
        out = np.empty((Q, B), np.float32)
        lib().jvo_nvq_scores(vsf, _f(self.mean), self.D, self.S, _f(queries), Q, _u8(self.bytes), _f(self.params), self.bytes.shape[0],
                             _i32(ordinals), B, _f(out))
        return out

    def reconstruction_error(self, vec):
        return float(lib().jvo_nvq_reconstruction_error(_f(self.mean), self.D, self.S, _f(f32(vec)), int(self.learn)))

    def as_reranker(self):
        """context manager: the oracle's search entry points rerank with these rows (NVQ.rerankerFor) while it is active;
        their `vecs` argument must still be non-None (any N x D float array) — it only says "rerank at all" """
        return _NVQReranker(self)


class _NVQReranker:
    def __init__(self, o):
        self.o = o

    def __enter__(self):
        o = self.o
        lib().jvo_set_nvq_reranker(_u8(o.bytes), _f(o.params), _f(o.mean), o.D, o.S)
        return o

    def __exit__(self, *a):
        lib().jvo_set_nvq_reranker(None, None, None, 0, 0)
        return False


def nvq_growth_grid():
    """-> (coarse[20], [fine values of coarse c])"""
    co = np.zeros(32, np.float32)
    fi = np.zeros((32, 32), np.float32)
    fn = np.zeros(32, np.int32)
    nc = lib().jvo_nvq_growth_grid(_f(co), _f(fi), _i32(fn), 32)
    return co[:nc].copy(), [fi[c, :fn[c]].copy() for c in range(nc)]


def dense_scan(vsf, queries, vecs):
    """Specification of the engine's MFMA tile form (jv_hip_exact_scan_dense): [Q, N] similarities, fused chains."""
    queries, vecs = f32(queries), f32(vecs)
    out = np.empty((queries.shape[0], vecs.shape[0]), np.float32)
    lib().jvo_dense_scan(int(vsf), _f(queries), queries.shape[0], _f(vecs), vecs.shape[0], vecs.shape[1], _f(out))
    return out


def set_simd(on):
    """cpu_baseline only: switch the SEARCH entry points (search_flat / rerank / OracleGraph.search) between the scalar
    checker arithmetic (False, the default) and the SIMD restatement of the reference's native kernels.  Returns the
    ISA tier name in effect ("scalar" when off or unsupported).  Process-wide; callers must switch it back."""
    tier = lib().jvo_set_simd(1 if on else 0)
    return lib().jvs_tier_name().decode() if tier else "scalar"


def _f(a):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u8(a):
    assert a.dtype == np.uint8 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _i32(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


# ---- row 1 ----
def dot(a, b):
    a, b = f32(a), f32(b)
    return float(lib().jvo_dot(_f(a), _f(b), a.size))


def l2(a, b):
    a, b = f32(a), f32(b)
    return float(lib().jvo_l2(_f(a), _f(b), a.size))


def cosine(a, b):
    a, b = f32(a), f32(b)
    return float(lib().jvo_cosine(_f(a), _f(b), a.size))


def dot_off(a, ao, b, bo, n):
    a, b = f32(a), f32(b)
    return float(lib().jvo_dot_off(_f(a), ao, _f(b), bo, n))


def l2_off(a, ao, b, bo, n):
    a, b = f32(a), f32(b)
    return float(lib().jvo_l2_off(_f(a), ao, _f(b), bo, n))


def cosine_off(a, ao, b, bo, n):
    a, b = f32(a), f32(b)
    return float(lib().jvo_cosine_off(_f(a), ao, _f(b), bo, n))


def compare(vsf, a, b):
    a, b = f32(a), f32(b)
    return float(lib().jvo_compare(vsf, _f(a), _f(b), a.size))


def compare_many(vsf, q, vecs):
    """scores[i] = VectorSimilarityFunction.compare(q, vecs[i])"""
    q, vecs = f32(q), f32(vecs)
    out = np.empty(vecs.shape[0], np.float32)
    L = lib()
    for i in range(vecs.shape[0]):
        out[i] = L.jvo_compare(vsf, _f(q), _f(vecs[i]), q.size)
    return out


def score_from_raw(vsf, raw):
    return float(lib().jvo_score_from_raw(vsf, C.c_float(raw)))


def make_vec(n, seed):
    v = np.empty(n, np.float32)
    lib().jvo_make_vec(_f(v), n, C.c_float(seed))
    return v


def assemble_and_sum(data, data_base, offs, offs_off, length):
    data, offs = f32(data), np.ascontiguousarray(offs, np.uint8)
    return float(lib().jvo_assemble_and_sum(_f(data), data_base, _u8(offs), offs_off, length))


def float_to_sortable_int(v):
    return int(lib().jvo_float_to_sortable_int(C.c_float(v)))


def nodequeue_encode(node, score):
    return int(lib().jvo_nodequeue_encode(node, C.c_float(score)))


def topk(ids, scores, k):
    scores = f32(scores)
    n = scores.size
    ids_a = None if ids is None else np.ascontiguousarray(ids, np.int32)
    oi = np.empty(k, np.int32)
    os_ = np.empty(k, np.float32)
    cnt = lib().jvo_topk(None if ids_a is None else _i32(ids_a), _f(scores), n, k, _i32(oi), _f(os_))
    return oi[:cnt].copy(), os_[:cnt].copy()


def rerank(queries, cand_vecs, cand_ids, vsf, top_k, nthreads=1):
    """exact rerank of pre-gathered candidates: cand_vecs (Q, R, D), cand_ids (Q, R) -> (ids, scores) (Q, top_k)"""
    queries, cand_vecs = f32(queries), f32(cand_vecs)
    cand_ids = np.ascontiguousarray(cand_ids, np.int32)
    Q, R = cand_ids.shape
    ids = np.empty((Q, top_k), np.int32)
    sc = np.empty((Q, top_k), np.float32)
    lib().jvo_rerank(_f(queries), _f(cand_vecs), _i32(cand_ids), Q, R, queries.shape[1], vsf, top_k, _i32(ids),
                     _f(sc), nthreads)
    return ids, sc


def pq_layout(vector_count, compressed_dim):
    lay = _Layout()
    rc = lib().jvo_pq_layout_compute(vector_count, compressed_dim, C.byref(lay))
    if rc != 0:
        raise ValueError("Invalid vector count" if rc == -1 else "Invalid compressed dimension")
    return {n: getattr(lay, n) for n, _ in _Layout._fields_}


def subvector_sizes_offsets(D, M):
    if M > D:
        raise ValueError("Number of subspaces must be less than or equal to the vector dimension")
    s, o = np.empty(M, np.int32), np.empty(M, np.int32)
    lib().jvo_subvector_sizes_offsets(D, M, _i32(s), _i32(o))
    return s, o


def pq_train(vecs, M, k=256, globally_center=False, seed=1, rounds=6, anisotropic_threshold=-1.0):
    """ProductQuantization.compute with a seeded RNG -> (OraclePQ, rounds_run[M]); anisotropic_threshold > -1 adds the
    anisotropic k-means rounds (cluster(6, 6))."""
    vecs = f32(vecs)
    n, D = vecs.shape
    cb = np.empty(k * D, np.float32)
    cen = np.zeros(D, np.float32)
    rr = (C.c_int * M)()
    lib().jvo_pq_train_aniso(_f(vecs), n, D, M, k, 1 if globally_center else 0, C.c_float(anisotropic_threshold), seed, rounds, _f(cb),
                             _f(cen), rr)
    return OraclePQ(D, M, cb, cen if globally_center else None, k), np.array(list(rr))


class OracleNodeQueue:
    """NodeQueue over a Bounded/GrowableLongHeap (B/graph/NodeQueue.java); order "min" or "max"; max_size None = growable."""

    def __init__(self, order, max_size=None, room=1024):
        self.order = 1 if order == "max" else 0
        self.cap = int(max_size) if max_size else 0
        self.heap = np.zeros(max(room, self.cap), np.int64)
        self.n = C.c_int(0)

    def push(self, node, score):
        return bool(lib().jvo_nodequeue_push(self.heap.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(self.n), self.cap,
                                             self.order, int(node), C.c_float(score)))

    def top(self):
        node, score = C.c_int32(), C.c_float()
        lib().jvo_nodequeue_top(self.heap.ctypes.data_as(C.POINTER(C.c_int64)), self.order, C.byref(node), C.byref(score))
        return node.value, score.value

    def pop(self):
        node, _ = self.top()
        lib().jvo_nodequeue_pop(self.heap.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(self.n))
        return node

    def size(self):
        return self.n.value


class OraclePQ:
    """Flat-array ProductQuantization for the oracle (codebooks concatenated centroid-major)."""

    def __init__(self, D, M, codebooks, centroid=None, k=256, sizes=None):
        self.D, self.M, self.k = int(D), int(M), int(k)
        if sizes is None:
            self.sizes, self.offsets = subvector_sizes_offsets(D, M)
        else:
            self.sizes = np.ascontiguousarray(sizes, np.int32)
            self.offsets = np.concatenate([[0], np.cumsum(self.sizes)[:-1]]).astype(np.int32)
        self.codebooks = f32(np.asarray(codebooks).reshape(-1))
        assert self.codebooks.size == self.k * int(self.sizes.sum())
        self.centroid = None if centroid is None else f32(centroid)
        self._s = _PQ(self.D, self.M, self.k, _i32(self.sizes), _i32(self.offsets), _f(self.codebooks),
                      None if self.centroid is None else _f(self.centroid), None)
        self._self_mag = None

    @property
    def ref(self):
        return C.byref(self._s)

    def cache_self_magnitudes(self):
        """Build partialSquaredMagnitudes once, like the reference's AtomicReference cache (ProductQuantization.java:75,
        238); the search entry points then stop rebuilding it per query.  Same values either way."""
        if self._self_mag is None:
            out = np.empty(self.M * self.k, np.float32)
            off = 0
            for m in range(self.M):
                cb = self.codebooks[off: off + self.k * int(self.sizes[m])]
                lib().jvo_calculate_partial_self_magnitudes(_f(cb), m, int(self.sizes[m]), self.k, _f(out))
                off += self.k * int(self.sizes[m])
            self._self_mag = out
            self._s.self_magnitudes = _f(out)
        return self._self_mag

    def codebook(self, m):
        off = int(self.k * self.sizes[:m].sum())
        return self.codebooks[off: off + self.k * int(self.sizes[m])]

    def encode(self, vec):
        vec = f32(vec)
        out = np.empty(self.M, np.uint8)
        lib().jvo_pq_encode(self.ref, _f(vec), _u8(out))
        return out

    def refine(self, vecs, rounds=1, seed=1, anisotropic_threshold=-1.0):
        """ProductQuantization.refine -> new OraclePQ (anisotropic k-means rounds when anisotropic_threshold > -1)."""
        vecs = f32(vecs)
        cb = np.empty_like(self.codebooks)
        lib().jvo_pq_refine_aniso(self.ref, C.c_float(anisotropic_threshold), _f(vecs), vecs.shape[0], rounds, seed, _f(cb))
        return OraclePQ(self.D, self.M, cb, self.centroid, self.k, sizes=self.sizes)

    def encode_anisotropic(self, vec, threshold):
        """ProductQuantization.encodeTo with anisotropicThreshold = threshold (> -1)."""
        vec = f32(vec)
        out = np.empty(self.M, np.uint8)
        lib().jvo_pq_encode_anisotropic(self.ref, C.c_float(threshold), _f(vec), _u8(out))
        return out

    def encode_all(self, vecs, nthreads=8):
        vecs = f32(vecs)
        n = vecs.shape[0]
        out = np.empty((n, self.M), np.uint8)
        lib().jvo_pq_encode_all(self.ref, _f(vecs), n, _u8(out), nthreads)
        return out

    def decode(self, code):
        code = np.ascontiguousarray(code, np.uint8)
        out = np.empty(self.D, np.float32)
        lib().jvo_pq_decode(self.ref, _u8(code), _f(out))
        return out

    def decoder(self, query, vsf, fused=False):
        """Returns (lut[M*k], amag[M*k] | None, bmag)."""
        query = f32(query)
        lut = np.empty(self.M * self.k, np.float32)
        amag = np.empty(self.M * self.k, np.float32) if vsf == COSINE else None
        bm = C.c_float(0.0)
        fn = lib().jvo_fuseddecoder_init if fused else lib().jvo_pqdecoder_init
        fn(self.ref, _f(query), vsf, _f(lut), None if amag is None else _f(amag), C.byref(bm))
        return lut, amag, float(bm.value)

    def adc_scores(self, query, vsf, codes, ordinals=None, fused=False):
        lut, amag, bm = self.decoder(query, vsf, fused)
        codes = np.ascontiguousarray(codes, np.uint8)
        if ordinals is None:
            n = codes.reshape(-1, self.M).shape[0]
            ordp = None
        else:
            ordinals = np.ascontiguousarray(ordinals, np.int32)
            n = ordinals.size
            ordp = _i32(ordinals)
        out = np.empty(n, np.float32)
        lib().jvo_adc_scores(vsf, self.M, self.k, _f(lut), None if amag is None else _f(amag),
                             C.c_float(bm), _u8(codes), ordp, n, _f(out))
        return out

    def search_flat(self, codes, vecs, queries, vsf, top_k, rerank_k, nthreads=1):
        codes = np.ascontiguousarray(codes, np.uint8)
        queries = f32(queries)
        vecs = None if vecs is None else f32(vecs)
        Q = queries.shape[0]
        ids = np.empty((Q, top_k), np.int32)
        sc = np.empty((Q, top_k), np.float32)
        lib().jvo_search_flat(self.ref, _u8(codes), None if vecs is None else _f(vecs), codes.shape[0], _f(queries), Q,
                              vsf, top_k, rerank_k, _i32(ids), _f(sc), nthreads)
        return ids, sc

    def direct_score(self, query, vsf, code):
        query, code = f32(query), np.ascontiguousarray(code, np.uint8)
        return float(lib().jvo_pq_direct_score(self.ref, _f(query), vsf, _u8(code)))

    def diversity_score(self, tri, vsf, code1, code2):
        """ImmutablePQVectors.diversityFunctionFor(node1, vsf).similarityTo(node2) on the triangular table `tri`."""
        c1, c2 = np.ascontiguousarray(code1, np.uint8), np.ascontiguousarray(code2, np.uint8)
        return float(lib().jvo_pq_diversity_score(_f(tri), self.M, self.k, vsf, _u8(c1), _u8(c2)))

    def retain_diverse(self, tri, vsf, codes, nodes, scores, max_degree, diverse_before=0, alpha=1.2):
        """VamanaDiversityProvider.retainDiverse over one NodeArray (nodes / scores sorted by score descending) with the PQ
        diversity score: returns (selected bool[n], nSelected, shortEdges)."""
        nodes = np.ascontiguousarray(nodes, np.int32)
        scores = f32(scores)
        codes = np.ascontiguousarray(codes, np.uint8)
        sel = np.zeros(max(len(nodes), 1), np.uint8)
        se = C.c_double()
        n = lib().jvo_retain_diverse(_f(tri), self.M, self.k, int(vsf), _u8(codes), nodes.ctypes.data_as(C.POINTER(C.c_int32)),
                                     _f(scores), len(nodes), int(max_degree), int(diverse_before), C.c_float(alpha), _u8(sel), C.byref(se))
        return sel[:len(nodes)].astype(bool), int(n), se.value

    def diversity_score_direct(self, vsf, code1, code2):
        """PQVectors.diversityFunctionFor (MutablePQVectors path): straight from the codebooks."""
        c1, c2 = np.ascontiguousarray(code1, np.uint8), np.ascontiguousarray(code2, np.uint8)
        return float(lib().jvo_pq_diversity_score_direct(self.ref, vsf, _u8(c1), _u8(c2)))

    def codebook_partial_sums(self, vsf):
        out = np.empty(self.M * self.k * (self.k + 1) // 2, np.float32)
        lib().jvo_pq_codebook_partial_sums(self.ref, vsf, _f(out))
        return out

    def serialize(self, version, aniso=-1.0):
        cap = 64 + 4 * (self.D + self.M + self.codebooks.size)
        buf = np.empty(cap, np.uint8)
        n = lib().jvo_pq_serialize(version, self.D, self.M, self.k, _i32(self.sizes),
                                   None if self.centroid is None else _f(self.centroid),
                                   C.c_float(aniso), _f(self.codebooks), _u8(buf), cap)
        return bytes(buf[:n])

    @staticmethod
    def parse(data: bytes):
        buf = np.frombuffer(data, np.uint8).copy()
        ver, D, M, k, cl = (C.c_int32() for _ in range(5))
        aniso = C.c_float()
        max_m = 4096
        sizes = np.empty(max_m, np.int32)
        centroid = np.empty(max(1, len(data) // 4), np.float32)
        cbs = np.empty(max(1, len(data) // 4), np.float32)
        consumed = C.c_size_t()
        rc = lib().jvo_pq_parse(_u8(buf), len(data), C.byref(ver), C.byref(D), C.byref(M), C.byref(k),
                                C.byref(cl), C.byref(aniso), _i32(sizes), max_m, _f(centroid), _f(cbs),
                                cbs.size, C.byref(consumed))
        if rc != 0:
            raise ValueError(f"jvo_pq_parse failed rc={rc}")
        szs = sizes[:M.value].copy()
        total = int(k.value * szs.sum())
        pq = OraclePQ(D.value, M.value, cbs[:total].copy(),
                      centroid[:cl.value].copy() if cl.value > 0 else None, k.value, sizes=szs)
        return pq, ver.value, float(aniso.value), consumed.value


class JavaRandom:
    """java.util.Random (the builder's level draws: new Random(0), GraphIndexBuilder.java:337)"""

    def __init__(self, seed=0):
        self._st = C.c_int64()
        lib().jvo_java_random_seed(C.byref(self._st), int(seed))

    def next_double(self):
        return float(lib().jvo_java_random_next_double(C.byref(self._st)))

    def graph_level(self, degree0, add_hierarchy=True):
        return int(lib().jvo_random_graph_level(C.byref(self._st), int(degree0), int(bool(add_hierarchy))))


class OracleBuilder:
    """GraphIndexBuilder driven by ONE thread over PQ build scores (jv_oracle.c "GraphIndexBuilder, one thread"): add() =
    addGraphNode, cleanup() = cleanup; rows come back in NodeArray order with their scores and diverseBefore mark."""

    def __init__(self, pq, codes, vecs, vsf, max_degree, beam_width, alpha=1.2, neighbor_overflow=1.2, add_hierarchy=False,
                 refine_final_graph=True, levels=None, dedupe_ids=False, improve_full_vectors=False, improve_sorted_candidates=False):
        self.pq = pq
        self.codes = np.ascontiguousarray(codes, np.uint8)
        self.vecs = f32(vecs)
        self.n = int(self.codes.shape[0])
        self.max_degree = int(max_degree)
        self._h = lib().jvo_builder_new(pq.ref, _u8(self.codes), _f(self.vecs), self.n, int(vsf), int(max_degree), int(beam_width),
                                        C.c_float(alpha), C.c_float(neighbor_overflow), int(bool(add_hierarchy)), int(bool(refine_final_graph)))
        self._levels = None
        if levels is not None:
            self._levels = np.ascontiguousarray(levels, np.int8)
            lib().jvo_builder_set_levels(self._h, self._levels.ctypes.data_as(C.POINTER(C.c_int8)))
        if dedupe_ids or improve_full_vectors or improve_sorted_candidates:
            lib().jvo_builder_set_deviations(self._h, int(bool(dedupe_ids)), int(bool(improve_full_vectors)), int(bool(improve_sorted_candidates)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().jvo_builder_free(self._h)
            self._h = None

    def add(self, node):
        return int(lib().jvo_builder_add(self._h, int(node)))

    def improve(self, node):
        lib().jvo_builder_improve(self._h, int(node))

    def enforce_degree(self, node):
        lib().jvo_builder_enforce_degree(self._h, int(node))

    def cleanup(self):
        lib().jvo_builder_cleanup(self._h)

    def info(self):
        e, el, nl, rp = C.c_int32(), C.c_int(), C.c_int(), C.c_int64()
        lib().jvo_builder_info(self._h, C.byref(e), C.byref(el), C.byref(nl), C.byref(rp))
        return {"entry_node": e.value, "entry_level": el.value, "n_levels": nl.value, "reprunes": rp.value}

    def row(self, level, node):
        """(ids, scores, diverseBefore) or None when the node is not on the level"""
        ids = np.empty(4 * self.max_degree + 8, np.int32)
        sc = np.empty(4 * self.max_degree + 8, np.float32)
        db = C.c_int()
        n = lib().jvo_builder_row(self._h, int(level), int(node), ids.ctypes.data_as(C.POINTER(C.c_int32)), _f(sc), C.byref(db))
        if n < 0:
            return None
        return ids[:n].copy(), sc[:n].copy(), db.value

    def rows(self, level, width):
        """[n, width] packed ids, -1 padded (the lists must fit)"""
        out = np.full((self.n, width), -1, np.int32)
        for v in range(self.n):
            r = self.row(level, v)
            if r is not None:
                assert r[0].size <= width
                out[v, :r[0].size] = r[0]
        return out


class NodeArrayProbe:
    """NodeArray's ordered insert on plain arrays (jvo_nodearray_insert_sorted): what the reference's TestNodeArray literals exercise"""

    def __init__(self, capacity=64):
        self._n = np.full(capacity, -1, np.int32)
        self._s = np.zeros(capacity, np.float32)
        self._size = C.c_int(0)

    def add_in_order(self, node, score):
        assert self._size.value == 0 or self._s[self._size.value - 1] >= np.float32(score)      # NodeArray.addInOrder :149-164
        self._n[self._size.value], self._s[self._size.value] = node, score
        self._size.value += 1

    def insert_sorted(self, node, score):
        return int(lib().jvo_nodearray_insert_sorted(self._n.ctypes.data_as(C.POINTER(C.c_int32)), _f(self._s), C.byref(self._size), int(node),
                                                     C.c_float(score)))

    def nodes(self):
        return self._n[:self._size.value].tolist()

    def scores(self):
        return self._s[:self._size.value].tolist()

    def merge(self, other):
        """NodeArray.merge(this, other) -> (nodes, scores)"""
        n1, s1, n2, s2 = self._n[:self._size.value].copy(), self._s[:self._size.value].copy(), other._n[:other._size.value].copy(), other._s[:other._size.value].copy()
        on, osc = np.empty(n1.size + n2.size + 1, np.int32), np.empty(n1.size + n2.size + 1, np.float32)
        p = C.POINTER(C.c_int32)
        m = lib().jvo_nodearray_merge(n1.ctypes.data_as(p), _f(s1), n1.size, n2.ctypes.data_as(p), _f(s2), n2.size, on.ctypes.data_as(p), _f(osc))
        return on[:m].tolist(), osc[:m].tolist()
