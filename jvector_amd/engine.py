"""Host-side mirror of the reference's operator interface for the hot path, over the C ABI.

Names follow the reference: VectorSimilarityFunction, ProductQuantization (encode / encodeAll /
load), PQVectors (precomputedScoreFunctionFor -> batched similarityTo), FusedPQ (similarityToNeighbor), NodeQueue
order top-k, and the two-pass search.  Every method is a thin call into libjvector_hip.so — there is no Python
arithmetic here and no CPU fallback: without the shared library or a gfx950 device the calls raise.

Arrays may be numpy arrays (host) or torch tensors (host or device).  Outputs follow the `like` rule: when the
driving input is a device torch tensor the result is a device torch tensor (zero copy, asynchronous on the
context's stream); otherwise a numpy array (the call synchronises).
"""
from __future__ import annotations

import ctypes as C
import enum

import numpy as np

from . import _lib
from ._lib import check


class VectorSimilarityFunction(enum.IntEnum):
    """B/vector/VectorSimilarityFunction.java:34-69 (ordinal order preserved)."""
    EUCLIDEAN = 0
    DOT_PRODUCT = 1
    COSINE = 2


class DecoderKind(enum.IntEnum):
    PQ = 0      # PQDecoder (B/quantization/PQDecoder.java)
    FUSED = 1   # FusedPQDecoder (B/quantization/FusedPQDecoder.java)


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _ptr(x, dtype=None):
    """(c_void_p, keepalive) of a numpy array / torch tensor; validates dtype + contiguity."""
    if x is None:
        return None, None
    if _is_torch(x):
        import torch
        if dtype is not None:
            want = {np.float32: torch.float32, np.uint8: torch.uint8, np.int32: torch.int32}[dtype]
            if x.dtype != want:
                raise ValueError(f"expected tensor dtype {want}, got {x.dtype}")
        if not x.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return C.c_void_p(x.data_ptr()), x
    a = np.ascontiguousarray(x, dtype=dtype)
    return C.c_void_p(a.ctypes.data), a


def _empty(shape, dtype, like):
    """output buffer of the caller's kind: torch tensor in (any device) -> torch tensor out on that device, else numpy"""
    if like is not None and _is_torch(like):
        import torch
        tdt = {np.float32: torch.float32, np.uint8: torch.uint8, np.int32: torch.int32}[dtype]
        return torch.empty(shape, dtype=tdt, device=like.device)
    return np.empty(shape, dtype=dtype)


def _finalizer(cls):
    """Class decorator: release the device object when the wrapper is garbage collected (close() stays idempotent)."""
    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: the library may already be gone
            pass
    cls.__del__ = __del__
    return cls


def pack_accept_bits(accept, n_nodes):
    """bool [n_nodes] or [Q, n_nodes] -> uint64 words, bit n of word n // 64 (the layout jv_hip_graph_search_filtered takes)"""
    a = np.asarray(accept.cpu().numpy() if _is_torch(accept) else accept, dtype=bool)
    if a.shape[-1] != n_nodes:
        raise ValueError(f"accept covers {a.shape[-1]} nodes, the graph has {n_nodes}")
    pad = (-n_nodes) % 64
    a = np.pad(a, [(0, 0)] * (a.ndim - 1) + [(0, pad)])
    return np.ascontiguousarray(np.packbits(a, axis=-1, bitorder="little").view(np.uint64))


def device_count() -> int:
    return int(_lib.load().jv_hip_device_count())


class HipContext:
    """One per host thread: owns the HIP stream + staging scratch (jv_ctx).

    stream: a hipStream_t handle (int; 0 = the legacy default stream), "private" for a context-owned non-blocking
    stream, or None = the stream torch is currently using on that device (so engine calls are ordered with the
    tensors the caller produces/consumes), falling back to the default stream without torch.
    """

    def __init__(self, device: int = 0, stream=None):
        self._lib = _lib.load()
        h = C.c_void_p()
        if stream == "private":
            sp = C.c_void_p(-1)  # JV_STREAM_PRIVATE
        elif stream is None:
            sp = None
            try:
                import torch
                if torch.cuda.is_available():
                    sp = C.c_void_p(torch.cuda.current_stream(int(device)).cuda_stream or None)
            except ImportError:
                pass
        else:
            sp = C.c_void_p(int(stream) or None)
        check(self._lib.jv_hip_ctx_create(int(device), sp, C.byref(h)))
        self._h = h
        self.device = int(device)

    @property
    def arch(self) -> str:
        return self._lib.jv_hip_active_arch(self.device).decode()

    def sync(self):
        check(self._lib.jv_hip_ctx_sync(self._h))

    def profile(self, enable=True):
        """Start (and reset) / stop HIP-event timing of the kernel regions on this context's stream."""
        check(self._lib.jv_hip_ctx_profile(self._h, 1 if enable else 0))

    def set_option(self, name, value):
        """jv_hip_ctx_set_option: a tuning option of THIS context (wins over the JVECTOR_HIP_<NAME> environment default);
        value None clears it"""
        if value is None:
            check(self._lib.jv_hip_ctx_clear_option(self._h, name.encode()))
        else:
            check(self._lib.jv_hip_ctx_set_option(self._h, name.encode(), int(value)))
        return self

    def stat(self, name):
        """jv_hip_ctx_get_stat: event counter of the searches that ran on this context (0 for names never counted)"""
        v = C.c_int64()
        check(self._lib.jv_hip_ctx_get_stat(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    def reset_stats(self):
        check(self._lib.jv_hip_ctx_reset_stats(self._h))

    def profile_read(self, region):
        """(total_ms, count) of a region: 'adc', 'topk', 'exact', 'lut', 'encode', 'norms'. Synchronises."""
        ms, cnt = C.c_double(), C.c_int64()
        check(self._lib.jv_hip_ctx_profile_read(self._h, region.encode(), C.byref(ms), C.byref(cnt)))
        return float(ms.value), int(cnt.value)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.jv_hip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@_finalizer
class ProductQuantization:
    """Device-resident codebooks (B/quantization/ProductQuantization.java)."""

    DEFAULT_CLUSTERS = 256

    def __init__(self, ctx: HipContext, handle):
        self.ctx = ctx
        self._lib = ctx._lib
        self._h = handle
        D, M, k, hc = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(self._lib.jv_hip_pq_info(handle, C.byref(D), C.byref(M), C.byref(k), C.byref(hc)))
        self.original_dimension, self.M, self.cluster_count = D.value, M.value, k.value
        self.has_global_centroid = bool(hc.value)

    @classmethod
    def from_codebooks(cls, ctx, D, M, codebooks, global_centroid=None, cluster_count=256, sizes=None):
        """codebooks: concatenation over m of k*size_m floats, centroid-major (ProductQuantization.write order)."""
        cb_p, cb_keep = _ptr(codebooks, np.float32)
        ce_p, ce_keep = _ptr(global_centroid, np.float32)
        sz_p, sz_keep = _ptr(sizes, np.int32)
        h = C.c_void_p()
        check(ctx._lib.jv_hip_pq_create(ctx._h, int(D), int(M), int(cluster_count), sz_p, cb_p, ce_p, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def load(cls, ctx, data: bytes):
        """ProductQuantization.load (:649-693): the reference's big-endian wire format."""
        buf = (C.c_ubyte * len(data)).from_buffer_copy(data)
        h, consumed = C.c_void_p(), C.c_size_t()
        check(ctx._lib.jv_hip_pq_load(ctx._h, C.cast(buf, C.c_void_p), len(data), C.byref(consumed), C.byref(h)))
        pq = cls(ctx, h)
        pq.bytes_consumed = consumed.value
        return pq

    @classmethod
    def compute(cls, ctx, vectors, M, cluster_count=256, globally_center=False, seed=1, anisotropic_threshold=-1.0):
        """ProductQuantization.compute (:109-139): k-means++ + 6 Lloyd rounds per subspace on `vectors` (the training sample,
        host or device [n, D]), plus 6 anisotropic rounds when anisotropic_threshold > -1.  Deterministic in (vectors, seed)."""
        n, D = int(vectors.shape[0]), int(vectors.shape[1])
        v_p, keep = _ptr(vectors, np.float32)
        h = C.c_void_p()
        check(ctx._lib.jv_hip_pq_train_anisotropic(ctx._h, v_p, n, D, int(M), int(cluster_count), int(bool(globally_center)),
                                                   C.c_float(anisotropic_threshold), int(seed), C.byref(h)))
        return cls(ctx, h)

    def refine(self, vectors, lloyds_rounds=1, seed=1):
        """ProductQuantization.refine (:194-221): a new PQ fine-tuned on `vectors`."""
        n = int(vectors.shape[0])
        v_p, keep = _ptr(vectors, np.float32)
        h = C.c_void_p()
        check(self._lib.jv_hip_pq_refine(self.ctx._h, self._h, v_p, n, int(lloyds_rounds), int(seed), C.byref(h)))
        return ProductQuantization(self.ctx, h)

    def write(self, version=6) -> bytes:
        """ProductQuantization.write (:560-599): the reference's big-endian wire format."""
        need = C.c_size_t()
        check(self._lib.jv_hip_pq_write(self.ctx._h, self._h, int(version), None, 0, C.byref(need)))
        buf = (C.c_ubyte * need.value)()
        check(self._lib.jv_hip_pq_write(self.ctx._h, self._h, int(version), C.cast(buf, C.c_void_p), need.value, C.byref(need)))
        return bytes(buf)

    def codebooks(self) -> np.ndarray:
        """The codebooks as one host float32 array (concatenation over m of k*size_m floats, centroid-major — what
        from_codebooks takes), read back through the wire format (ProductQuantization.write v6: magic, version, centroid
        length [+ centroid], M, M sizes, anisotropic threshold, k, codebooks; big-endian)."""
        b = self.write(6)
        gcl = int.from_bytes(b[8:12], "big", signed=True)
        off = 12 + 4 * gcl
        M = int.from_bytes(b[off:off + 4], "big", signed=True)
        sizes = np.frombuffer(b, dtype=">i4", count=M, offset=off + 4)
        off += 4 + 4 * M + 4          # sizes, anisotropic threshold
        k = int.from_bytes(b[off:off + 4], "big", signed=True)
        n = int(k) * int(sizes.sum())
        return np.frombuffer(b, dtype=">f4", count=n, offset=off + 4).astype(np.float32)

    @property
    def anisotropic_threshold(self) -> float:
        """ProductQuantization.anisotropicThreshold; -1 = UNWEIGHTED."""
        return float(self._lib.jv_hip_pq_anisotropic_threshold(self._h))

    def set_anisotropic_threshold(self, t: float):
        """t > -1: encode / encode_all / PQVectors.encode_and_build use encodeAnisotropic (:269-306); unit-length input."""
        check(self._lib.jv_hip_pq_set_anisotropic_threshold(self._h, C.c_float(t)))
        return self

    def get_subspace_count(self):
        return self.M

    def get_cluster_count(self):
        return self.cluster_count

    def encode_all(self, vectors, out=None):
        """PQVectors.encodeAndBuild's arithmetic for a batch: (n, D) float32 -> (n, M) uint8."""
        n = int(vectors.shape[0])
        if vectors.shape[1] != self.original_dimension:
            raise ValueError(f"vector dimensions differ: {vectors.shape[1]}!={self.original_dimension}")
        v_p, keep = _ptr(vectors, np.float32)
        if out is None:
            out = _empty((n, self.M), np.uint8, vectors)
        o_p, okeep = _ptr(out, np.uint8)
        check(self._lib.jv_hip_pq_encode(self.ctx._h, self._h, v_p, n, o_p))
        return out

    def encode(self, vector):
        return self.encode_all(np.asarray(vector, np.float32).reshape(1, -1))[0]

    def self_magnitudes(self):
        out = np.empty(self.M * self.cluster_count, np.float32)
        check(self._lib.jv_hip_pq_self_magnitudes(self.ctx._h, self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.jv_hip_pq_destroy(self._h)
            self._h = None


@_finalizer
class VectorSet:
    """Device-resident full-resolution vectors (RandomAccessVectorValues for the reranker)."""

    def __init__(self, ctx, vectors):
        self.ctx, self._lib = ctx, ctx._lib
        n, D = int(vectors.shape[0]), int(vectors.shape[1])
        h = C.c_void_p()
        if _is_torch(vectors) and vectors.is_cuda:
            p, self._keep = _ptr(vectors, np.float32)
            check(self._lib.jv_hip_vectors_wrap(ctx._h, n, D, p, C.byref(h)))
        else:
            check(self._lib.jv_hip_vectors_create(ctx._h, n, D, C.byref(h)))
            p, keep = _ptr(vectors, np.float32)
            check(self._lib.jv_hip_vectors_upload(ctx._h, h, 0, n, p))
        self._h, self.count, self.dimension = h, n, D

    @classmethod
    def _from_nvq(cls, nvq_vectors):
        self = cls.__new__(cls)
        self.ctx, self._lib = nvq_vectors.ctx, nvq_vectors._lib
        h = C.c_void_p()
        check(self._lib.jv_hip_vectors_from_nvq(self.ctx._h, nvq_vectors._h, C.byref(h)))
        self._h, self.count, self.dimension = h, nvq_vectors.count(), nvq_vectors.nvq.dimension
        self._keep = nvq_vectors   # the rows must outlive the handle
        return self

    def size(self):
        return self.count

    def invalidate(self):
        """the wrapped tensor was edited in place: drop the cached per-row norms of the cosine rerank (jv_hip_vectors_invalidate)"""
        check(self._lib.jv_hip_vectors_invalidate(self._h))
        return self

    def scores(self, queries, vsf, ordinals):
        """rerank form: out[q, j] = vsf.compare(queries[q], vectors[ordinals[q, j]])"""
        Q, B = int(ordinals.shape[0]), int(ordinals.shape[1])
        q_p, qk = _ptr(queries, np.float32)
        o_p, ok = _ptr(ordinals, np.int32)
        out = _empty((Q, B), np.float32, ordinals)
        out_p, outk = _ptr(out, np.float32)
        check(self._lib.jv_hip_exact_scores(self.ctx._h, self._h, q_p, Q, int(vsf), o_p, B, out_p))
        return out

    def pair_scores(self, vsf, node1, node2):
        """BuildScoreProvider.randomAccessScoreProvider's diversity function (BuildScoreProvider.java:151-157) for P nodes at once:
        out[p, b] = vsf.compare(vectors[node1[p]], vectors[node2[p, b]]); an ordinal outside the set gives -inf."""
        P, B = int(node2.shape[0]), int(node2.shape[1])
        a_p, ak = _ptr(node1, np.int32)
        b_p, bk = _ptr(node2, np.int32)
        out = _empty((P, B), np.float32, node2)
        out_p, outk = _ptr(out, np.float32)
        check(self._lib.jv_hip_exact_pair_scores(self.ctx._h, self._h, int(vsf), a_p, P, b_p, B, out_p))
        return out

    def scan(self, queries, vsf, first=0, count=None, out=None, dense=False):
        """brute-force form: out[q, i] = vsf.compare(queries[q], vectors[first + i]).
        dense=True: the MFMA tile form (jv_hip_exact_scan_dense) — fused k-ascending chains, within 1e-5 of the default
        bit-exact scalar-order scores, for ground truth / candidate generation over many queries."""
        Q = int(queries.shape[0])
        count = self.count - first if count is None else int(count)
        q_p, qk = _ptr(queries, np.float32)
        if out is None:
            out = _empty((Q, count), np.float32, queries)
        out_p, outk = _ptr(out, np.float32)
        fn = self._lib.jv_hip_exact_scan_dense if dense else self._lib.jv_hip_exact_scan
        check(fn(self.ctx._h, self._h, q_p, Q, int(vsf), int(first), count, out_p))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.jv_hip_vectors_destroy(self._h)
            self._h = None


@_finalizer
class NVQuantization:
    """NVQuantization (B/quantization/NVQuantization.java): the global mean + the split into sub-vectors; encodes on the GPU."""

    def __init__(self, ctx, handle):
        self.ctx, self._lib, self._h = ctx, ctx._lib, handle
        self.dimension = int(self._lib.jv_hip_nvq_dimension(handle))
        self.subvectors = int(self._lib.jv_hip_nvq_subvectors(handle))
        self.learn = True

    @classmethod
    def create(cls, ctx, global_mean, n_subvectors):
        """NVQuantization.create(globalMean, nSubVectors) :170-173"""
        h = C.c_void_p()
        p, keep = _ptr(global_mean, np.float32)
        check(ctx._lib.jv_hip_nvq_create(ctx._h, int(global_mean.shape[0]), int(n_subvectors), p, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def compute(cls, ctx, vectors: "VectorSet", n_subvectors):
        """NVQuantization.compute(ravv, nSubVectors) :153-163 — the mean is accumulated on the device, rows in order"""
        h = C.c_void_p()
        check(ctx._lib.jv_hip_nvq_compute(ctx._h, vectors._h, int(n_subvectors), C.byref(h)))
        return cls(ctx, h)

    def set_learn(self, learn):
        check(self._lib.jv_hip_nvq_set_learn(self._h, int(bool(learn))))
        self.learn = bool(learn)
        return self

    def global_mean(self):
        out = np.empty(self.dimension, np.float32)
        check(self._lib.jv_hip_nvq_global_mean(self.ctx._h, self._h, C.c_void_p(out.ctypes.data)))
        return out

    def subvector_sizes(self):
        """getSubvectorSizesAndOffsets :236-252 -> sizes"""
        base, rem = divmod(self.dimension, self.subvectors)
        return [base + (1 if i < rem else 0) for i in range(self.subvectors)]

    def encode_all(self, vectors: "VectorSet", first=0, count=None) -> "NVQVectors":
        """encodeAll :182-195"""
        count = vectors.count - first if count is None else int(count)
        out = NVQVectors(self.ctx, self, count=count)
        check(self._lib.jv_hip_nvq_encode(self.ctx._h, self._h, vectors._h, int(first), count, out._h, 0))
        return out

    def write(self, version=6) -> bytes:
        """NVQuantization.write :260-277"""
        be = lambda v: np.asarray(v, np.int64).astype(np.uint32).astype(">u4").tobytes()  # noqa: E731
        return (be([version, self.dimension]) + self.global_mean().astype(">f4").tobytes() + be([8, self.subvectors])
                + be(self.subvector_sizes()))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.jv_hip_nvq_destroy(self._h)
            self._h = None


@_finalizer
class NVQVectors:
    """NVQVectors (B/quantization/NVQVectors.java) resident on the device: one byte per dimension + four floats per
    sub-vector.  `as_vector_set()` is the reranker a graph with the NVQ_VECTORS feature uses (NVQ.rerankerFor)."""

    def __init__(self, ctx, nvq: NVQuantization, bytes_=None, params=None, count=None):
        self.ctx, self._lib, self.nvq = ctx, ctx._lib, nvq
        n = int(count if bytes_ is None else bytes_.shape[0])
        h = C.c_void_p()
        check(self._lib.jv_hip_nvq_vectors_create(ctx._h, nvq._h, n, C.byref(h)))
        self._h, self._count = h, n
        if bytes_ is not None:
            self.upload(0, bytes_, params)

    def count(self):
        return self._count

    def upload(self, first, bytes_, params):
        n = int(bytes_.shape[0])
        if tuple(bytes_.shape) != (n, self.nvq.dimension) or int(np.prod(params.shape)) != n * self.nvq.subvectors * 4:
            raise ValueError("NVQ rows: bytes must be count x D and params count x S x 4")
        b_p, bk = _ptr(bytes_, np.uint8)
        p_p, pk = _ptr(params, np.float32)
        check(self._lib.jv_hip_nvq_vectors_upload(self.ctx._h, self._h, int(first), n, b_p, p_p))

    def get(self, first=0, count=None):
        """-> (bytes[count, D] uint8, params[count, S, 4] float32 = {minValue, maxValue, growthRate, midpoint})"""
        count = self._count - first if count is None else int(count)
        b = np.empty((count, self.nvq.dimension), np.uint8)
        p = np.empty((count, self.nvq.subvectors, 4), np.float32)
        check(self._lib.jv_hip_nvq_vectors_download(self.ctx._h, self._h, int(first), count, C.c_void_p(b.ctypes.data),
                                                    C.c_void_p(p.ctypes.data)))
        return b, p

    def scores(self, queries, vsf, ordinals):
        """out[q, j] = scoreFunctionFor(queries[q], vsf).similarityTo(ordinals[q, j]) (NVQVectors.java:110-113)"""
        Q, B = int(ordinals.shape[0]), int(ordinals.shape[1])
        q_p, qk = _ptr(queries, np.float32)
        o_p, ok = _ptr(ordinals, np.int32)
        out = _empty((Q, B), np.float32, ordinals)
        out_p, outk = _ptr(out, np.float32)
        check(self._lib.jv_hip_nvq_scores(self.ctx._h, self._h, q_p, Q, int(vsf), o_p, B, out_p))
        return out

    def as_vector_set(self) -> "VectorSet":
        """a VectorSet whose rerank goes through these rows (jv_hip_vectors_from_nvq); pass it wherever `vectors` goes"""
        return VectorSet._from_nvq(self)

    def write(self, version=6) -> bytes:
        """NVQVectors.write :50-62 (QuantizedVector.write :437-443, QuantizedSubVector.write :577-587)"""
        from .formats import nvq_records
        b, p = self.get()
        n = np.asarray([self._count], np.int64).astype(np.uint32).astype(">u4").tobytes()
        return self.nvq.write(version) + n + nvq_records(self.nvq.subvector_sizes(), b, p).tobytes()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.jv_hip_nvq_vectors_destroy(self._h)
            self._h = None


@_finalizer
class PQVectors:
    """Device-resident code store, ordinal-major (B/quantization/PQVectors.java)."""

    def __init__(self, ctx, pq: ProductQuantization, codes=None, count=None):
        self.ctx, self._lib, self.pq = ctx, ctx._lib, pq
        h = C.c_void_p()
        if codes is not None and _is_torch(codes) and codes.is_cuda:
            n = int(codes.shape[0])
            p, self._keep = _ptr(codes, np.uint8)
            check(self._lib.jv_hip_codes_wrap(ctx._h, pq._h, n, p, C.byref(h)))
        else:
            n = int(count if codes is None else codes.shape[0])
            check(self._lib.jv_hip_codes_create(ctx._h, pq._h, n, C.byref(h)))
            if codes is not None:
                p, keep = _ptr(codes, np.uint8)
                check(self._lib.jv_hip_codes_upload(ctx._h, h, 0, n, p))
        self._h, self._count = h, n

    @classmethod
    def encode_and_build(cls, ctx, pq, vectors: VectorSet):
        """PQVectors.encodeAndBuild (:109-152) on device-resident vectors."""
        self = cls(ctx, pq, count=vectors.count)
        check(ctx._lib.jv_hip_pq_encode_into(ctx._h, pq._h, vectors._h, 0, vectors.count, self._h))
        return self

    def count(self):
        return self._count

    def get(self, first, n=1):
        out = np.empty((n, self.pq.M), np.uint8)
        check(self._lib.jv_hip_codes_download(self.ctx._h, self._h, int(first), int(n), out.ctypes.data_as(C.c_void_p)))
        return out

    def direct_scores(self, queries, vsf, ordinals):
        """PQVectors.scoreFunctionFor(q, vsf).similarityTo(node) (:223-281) for ordinals[Q, B]: no look-up table."""
        Q, B = int(ordinals.shape[0]), int(ordinals.shape[1])
        q_p, kq = _ptr(queries, np.float32)
        o_p, ko = _ptr(ordinals, np.int32)
        out = _empty((Q, B), np.float32, ordinals)
        s_p, ks = _ptr(out, np.float32)
        check(self._lib.jv_hip_direct_scores(self.ctx._h, self._h, q_p, Q, int(vsf), o_p, B, s_p))
        return out

    def precomputed_score_function_for(self, queries, vsf, luts=None):
        """PQVectors.precomputedScoreFunctionFor (:210-221), batched over Q queries."""
        luts = luts or QueryTables(self.ctx, self.pq, int(queries.shape[0]))
        luts.build(queries, vsf, DecoderKind.PQ)
        return ApproximateScoreFunction(self, luts)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.jv_hip_codes_destroy(self._h)
            self._h = None


class PQBuildScoreProvider:
    """Batched BuildScoreProvider.pqBuildScoreProvider (B/graph/similarity/BuildScoreProvider.java:167-212): the PQ-only
    score functions graph construction uses, over device-resident codes."""

    def __init__(self, ctx, pq_vectors: "PQVectors", vsf):
        self.ctx, self._lib, self.cv, self.vsf = ctx, ctx._lib, pq_vectors, vsf
        h = C.c_void_p()
        check(self._lib.jv_hip_pair_table_create(ctx._h, pq_vectors.pq._h, int(vsf), C.byref(h)))
        self._h = h

    def codebook_partial_sums(self):
        """ProductQuantization.createCodebookPartialSums(vsf) (:609-628) as a host array."""
        out = np.empty(int(self._lib.jv_hip_pair_table_size(self._h)), np.float32)
        check(self._lib.jv_hip_pair_table_download(self.ctx._h, self._h, C.c_void_p(out.ctypes.data)))
        return out

    def diversity_scores(self, node1, node2):
        """scores[p, b] = diversityScoreFunctionFor(node1[p]).similarityTo(node2[p, b]); ordinals < 0 give -inf."""
        P, B = int(node2.shape[0]), int(node2.shape[1])
        n1_p, k1 = _ptr(node1, np.int32)
        n2_p, k2 = _ptr(node2, np.int32)
        out = _empty((P, B), np.float32, node2)
        o_p, ko = _ptr(out, np.float32)
        check(self._lib.jv_hip_code_pair_scores(self.ctx._h, self._h, self.cv._h, n1_p, P, n2_p, B, o_p))
        return out

    def retain_diverse(self, cand_nodes, cand_scores, max_degree, alpha=1.2, cand_count=None, diverse_before=None):
        """VamanaDiversityProvider.retainDiverse for P nodes at once (the robust prune of Vamana construction): cand_nodes /
        cand_scores [P, C] sorted by score descending per row.  Returns (selected [P, max_degree] candidate indices ascending,
        -1 padded; n_selected [P]; short_edges [P])."""
        P, Cn = int(cand_nodes.shape[0]), int(cand_nodes.shape[1])
        n_p, kn = _ptr(cand_nodes, np.int32)
        s_p, ks = _ptr(cand_scores, np.float32)
        c_p, kc = _ptr(cand_count, np.int32)
        d_p, kd = _ptr(diverse_before, np.int32)
        sel = _empty((P, max_degree), np.int32, cand_nodes)
        cnt = _empty((P,), np.int32, cand_nodes)
        se = _empty((P,), np.float32, cand_nodes)
        sel_p, k1 = _ptr(sel, np.int32)
        cnt_p, k2 = _ptr(cnt, np.int32)
        se_p, k3 = _ptr(se, np.float32)
        check(self._lib.jv_hip_retain_diverse(self.ctx._h, self._h, self.cv._h, P, Cn, n_p, s_p, c_p, d_p, int(max_degree),
                                              C.c_float(alpha), sel_p, cnt_p, se_p))
        return sel, cnt, se

    def decode(self, ordinals):
        """ProductQuantization.decode of the listed codes (searchProviderFor(node1) searches from this vector)."""
        n = int(ordinals.shape[0])
        o_p, ko = _ptr(ordinals, np.int32)
        out = _empty((n, self.cv.pq.original_dimension), np.float32, ordinals)
        v_p, kv = _ptr(out, np.float32)
        check(self._lib.jv_hip_pq_decode(self.ctx._h, self.cv._h, o_p, 0, n, v_p))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.jv_hip_pair_table_destroy(self._h)
            self._h = None

    __del__ = close


@_finalizer
class QueryTables:
    """ADC look-up tables of a query batch (the state PQDecoder / FusedPQDecoder constructors compute)."""

    def __init__(self, ctx, pq, max_queries):
        self.ctx, self._lib, self.pq = ctx, ctx._lib, pq
        h = C.c_void_p()
        check(self._lib.jv_hip_luts_create(ctx._h, pq._h, int(max_queries), C.byref(h)))
        self._h, self.capacity, self.Q, self.vsf = h, int(max_queries), 0, None

    def build(self, queries, vsf, kind=DecoderKind.PQ):
        Q = int(queries.shape[0])
        if queries.shape[1] != self.pq.original_dimension:
            raise ValueError(f"vector dimensions differ: {queries.shape[1]}!={self.pq.original_dimension}")
        p, keep = _ptr(queries, np.float32)
        check(self._lib.jv_hip_luts_build(self.ctx._h, self._h, p, Q, int(vsf), int(kind)))
        self.Q, self.vsf = Q, VectorSimilarityFunction(int(vsf))
        return self

    def table(self, q):
        lut = np.empty(self.pq.M * 256, np.float32)
        bm = C.c_float()
        check(self._lib.jv_hip_luts_download(self.ctx._h, self._h, int(q), lut.ctypes.data_as(C.c_void_p),
                                             C.cast(C.byref(bm), C.c_void_p)))
        return lut, float(bm.value)

    def bound_tables(self):
        """the 8-bit upper-bound tables of the staged queries as the register-table traversal loads them (diagnostic accessor):
        (tab uint32[Q, M * 64], meta float32[Q, 4])"""
        tab = np.empty((self.Q, self.pq.M * 64), np.uint32)
        meta = np.empty((self.Q, 4), np.float32)
        check(self._lib.jv_hip_luts_bound_tables(self.ctx._h, self._h, tab.ctypes.data_as(C.c_void_p), meta.ctypes.data_as(C.c_void_p)))
        return tab, meta

    def close(self):
        if getattr(self, "_h", None):
            self._lib.jv_hip_luts_destroy(self._h)
            self._h = None


class ApproximateScoreFunction:
    """Batched ScoreFunction.ApproximateScoreFunction (B/graph/similarity/ScoreFunction.java:30-80)."""

    def __init__(self, cv: PQVectors, luts: QueryTables):
        self.cv, self.luts = cv, luts

    def similarity_to(self, ordinals):
        """out[q, j] = similarityTo(ordinals[q, j]) ; negative ordinals -> -inf"""
        Q, B = int(ordinals.shape[0]), int(ordinals.shape[1])
        if Q != self.luts.Q:
            raise ValueError("ordinals must have one row per query")
        o_p, ok = _ptr(ordinals, np.int32)
        out = _empty((Q, B), np.float32, ordinals)
        out_p, outk = _ptr(out, np.float32)
        check(self.cv._lib.jv_hip_adc_scores(self.cv.ctx._h, self.luts._h, self.cv._h, o_p, B, out_p))
        return out

    def similarity_to_range(self, first, count, out=None, like=None):
        """out[q, i] = similarityTo(first + i) — the flat-scan form"""
        if out is None:
            out = _empty((self.luts.Q, int(count)), np.float32, like)
        out_p, outk = _ptr(out, np.float32)
        check(self.cv._lib.jv_hip_adc_scan(self.cv.ctx._h, self.luts._h, self.cv._h, int(first), int(count), out_p))
        return out


@_finalizer
class FusedPQ:
    """Device-resident L0 fused blocks (B/graph/disk/feature/FusedPQ.java:146-161 layout)."""

    def __init__(self, ctx, pq, blocks, neighbors):
        self.ctx, self._lib, self.pq = ctx, ctx._lib, pq
        n, max_degree = int(neighbors.shape[0]), int(neighbors.shape[1])
        h = C.c_void_p()
        check(self._lib.jv_hip_fused_create(ctx._h, pq._h, n, max_degree, C.byref(h)))
        b_p, bk = _ptr(blocks, np.uint8)
        n_p, nk = _ptr(neighbors, np.int32)
        check(self._lib.jv_hip_fused_upload(ctx._h, h, 0, n, b_p, n_p))
        self._h, self.count, self.max_degree = h, n, max_degree

    @classmethod
    def build(cls, ctx, pq_vectors: "PQVectors", neighbors):
        """FusedPQ.writeInline on the device: blocks gathered from `pq_vectors` for the given neighbour rows [n, maxDegree]."""
        self = cls.__new__(cls)
        self.ctx, self._lib, self.pq = ctx, ctx._lib, pq_vectors.pq
        n, max_degree = int(neighbors.shape[0]), int(neighbors.shape[1])
        h = C.c_void_p()
        check(self._lib.jv_hip_fused_create(ctx._h, self.pq._h, n, max_degree, C.byref(h)))
        self._h, self.count, self.max_degree = h, n, max_degree
        n_p, nk = _ptr(neighbors, np.int32)
        check(self._lib.jv_hip_fused_build(ctx._h, h, pq_vectors._h, 0, n, n_p))
        return self

    def get(self, first=0, n=None):
        """(blocks[n, maxDegree*M] uint8, neighbors[n, maxDegree] int32) of nodes [first, first+n) as host arrays."""
        n = self.count - first if n is None else n
        blocks = np.empty((n, self.max_degree * self.pq.M), np.uint8)
        nbrs = np.empty((n, self.max_degree), np.int32)
        check(self._lib.jv_hip_fused_download(self.ctx._h, self._h, int(first), int(n), C.c_void_p(blocks.ctypes.data),
                                              C.c_void_p(nbrs.ctypes.data)))
        return blocks, nbrs

    def approximate_score_function_for(self, queries, vsf, luts=None):
        luts = luts or QueryTables(self.ctx, self.pq, int(queries.shape[0]))
        luts.build(queries, vsf, DecoderKind.FUSED)
        return FusedScoreFunction(self, luts)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.jv_hip_fused_destroy(self._h)
            self._h = None


class FusedScoreFunction:
    """Batched FusedPQDecoder: enableSimilarityToNeighbors(origin) + similarityToNeighbor(origin, i) for all i."""

    def __init__(self, fused: FusedPQ, luts: QueryTables):
        self.fused, self.luts = fused, luts

    def similarity_to_neighbors(self, origins, return_neighbors=False):
        Q = int(origins.shape[0])
        o_p, ok = _ptr(origins, np.int32)
        out = _empty((Q, self.fused.max_degree), np.float32, origins)
        out_p, outk = _ptr(out, np.float32)
        nb = _empty((Q, self.fused.max_degree), np.int32, origins) if return_neighbors else None
        nb_p, nbk = _ptr(nb, np.int32)
        check(self.fused._lib.jv_hip_fused_scores(self.fused.ctx._h, self.luts._h, self.fused._h, o_p, out_p, nb_p))
        return (out, nb) if return_neighbors else out


def topk(ctx, scores, k, ids=None, id_base=0):
    """NodeQueue-order top-k of each row (higher score first, ties -> smaller id). Returns (ids, scores)."""
    Q, n = int(scores.shape[0]), int(scores.shape[1])
    s_p, sk = _ptr(scores, np.float32)
    i_p, ik = _ptr(ids, np.int32)
    oi = _empty((Q, k), np.int32, scores)
    osc = _empty((Q, k), np.float32, scores)
    oi_p, oik = _ptr(oi, np.int32)
    os_p, osk = _ptr(osc, np.float32)
    check(ctx._lib.jv_hip_topk(ctx._h, s_p, i_p, Q, n, n, int(id_base), int(k), oi_p, os_p))
    return oi, osc


class FlatSearcher:
    """Two-pass search over one shard: ADC scan of every code -> top rerankK -> exact rerank -> topK.

    Same per-candidate arithmetic as GraphSearcher.search's scoring calls (GraphSearcher.java:443-450,471-507)
    with the whole shard as the candidate set (the recall-bearing path before a graph exists)."""

    def __init__(self, ctx, pq, pq_vectors: PQVectors, vectors: VectorSet | None, max_queries=256, id_base=0):
        self.ctx, self.pq, self.cv, self.vectors, self.id_base = ctx, pq, pq_vectors, vectors, int(id_base)
        self.luts = QueryTables(ctx, pq, max_queries)

    def search(self, queries, vsf, top_k, rerank_k, out_ids=None, out_scores=None):
        Q = int(queries.shape[0])
        q_p, qk = _ptr(queries, np.float32)
        if out_ids is None:
            out_ids = _empty((Q, top_k), np.int32, queries)
        if out_scores is None:
            out_scores = _empty((Q, top_k), np.float32, queries)
        oi_p, oik = _ptr(out_ids, np.int32)
        os_p, osk = _ptr(out_scores, np.float32)
        check(self.ctx._lib.jv_hip_search_flat(self.ctx._h, self.luts._h, self.cv._h,
                                               self.vectors._h if self.vectors is not None else None, q_p, Q, int(vsf),
                                               int(top_k), int(rerank_k), self.id_base, oi_p, os_p))
        return out_ids, out_scores


@_finalizer
class GraphIndex:
    """Host-resident multi-level adjacency (the reference keeps the graph on the host: OnHeapGraphIndex /
    OnDiskGraphIndex.View).  levels[0] = (None, neighbors[n_nodes, maxDegree]); upper levels = (sorted node ids,
    neighbors[count, degree]); rows packed and padded with -1."""

    def __init__(self, ctx, n_nodes, levels, entry_node, entry_level):
        self.ctx, self._lib = ctx, ctx._lib
        h = C.c_void_p()
        check(self._lib.jv_hip_graph_create(ctx._h, int(n_nodes), len(levels), C.byref(h)))
        self._h = h
        self.n_nodes, self.max_degree = int(n_nodes), int(levels[0][1].shape[1])
        for lv, (ids, nbrs) in enumerate(levels):
            nb = np.ascontiguousarray(nbrs.cpu().numpy() if _is_torch(nbrs) else nbrs, dtype=np.int32)
            idp = None
            if ids is not None:
                ida = np.ascontiguousarray(ids.cpu().numpy() if _is_torch(ids) else ids, dtype=np.int32)
                idp = C.c_void_p(ida.ctypes.data)
            check(self._lib.jv_hip_graph_set_level(ctx._h, h, lv, nb.shape[0], idp, C.c_void_p(nb.ctypes.data),
                                                   nb.shape[1]))
        check(self._lib.jv_hip_graph_set_entry(h, int(entry_node), int(entry_level)))

    @classmethod
    def on_device(cls, ctx, neighbors, entry_node):
        """Single-level graph whose adjacency [n_nodes, degree] int32 (-1 padded) lives in caller-owned DEVICE memory (a
        torch.cuda tensor) and is read in place by the device traversal — the owner may rewrite rows between searches
        (jvector_amd.builder uses this to search the graph it is building)."""
        self = cls.__new__(cls)
        self.ctx, self._lib = ctx, ctx._lib
        n, deg = int(neighbors.shape[0]), int(neighbors.shape[1])
        h = C.c_void_p()
        check(self._lib.jv_hip_graph_create(ctx._h, n, 1, C.byref(h)))
        self._h, self.n_nodes, self.max_degree = h, n, deg
        p, self._keep = _ptr(neighbors, np.int32)
        check(self._lib.jv_hip_graph_set_level0_device(ctx._h, h, p, deg))
        check(self._lib.jv_hip_graph_set_entry(h, int(entry_node), 0))
        return self

    def set_entry(self, node, level=0):
        check(self._lib.jv_hip_graph_set_entry(self._h, int(node), int(level)))
        return self

    TRAVERSAL = {"auto": 0, "host": 1, "device": 2}

    def set_traversal(self, mode: str):
        """Where the traversal state lives: "host" (worker pool + GPU frontier scoring), "device" (one wavefront per
        query runs the whole loop on the GPU) or "auto" (device wherever it applies — 256-cluster codebooks, degree <= 512,
        queues fit LDS; kernels specialised for uniform 8-dim sub-vectors at M in {16,32,48,64,96,128,192}, a generic build for
        every other quantizer — else host).  Results are identical."""
        check(self._lib.jv_hip_graph_set_traversal(self._h, self.TRAVERSAL[mode]))
        return self

    def close(self):
        if getattr(self, "_h", None):
            self._lib.jv_hip_graph_destroy(self._h)
            self._h = None


class GraphSearcher:
    """Batched GraphSearcher (B/graph/GraphSearcher.java): the whole search loop of every query of the batch runs on the GPU
    (one wavefront per query; FusedPQ blocks at layer 0 when `fused` is given, the code store otherwise) or, for shapes the
    kernel does not cover / on request, in the lock-step host searcher with GPU frontier scoring (GraphIndex.set_traversal).
    Results, visitedCount and expandedCount equal the reference's sequential search either way."""

    def __init__(self, ctx, graph: GraphIndex, pq, pq_vectors: PQVectors, fused: FusedPQ | None = None,
                 vectors: VectorSet | None = None, max_queries=4096):
        self.ctx, self.graph, self.cv, self.fused, self.vectors = ctx, graph, pq_vectors, fused, vectors
        self.luts = QueryTables(ctx, pq, max_queries)

    def search(self, queries, vsf, top_k, rerank_k, return_stats=False, accept=None):
        """accept = acceptOrds (GraphSearcher.search's Bits filter): None, a bool array [n_nodes] shared by the batch, or
        [Q, n_nodes] one filter per query; filtered-out nodes are traversed but never returned."""
        Q = int(queries.shape[0])
        q_p, qk = _ptr(queries, np.float32)
        out_ids = _empty((Q, top_k), np.int32, queries)
        out_sc = _empty((Q, top_k), np.float32, queries)
        oi_p, oik = _ptr(out_ids, np.int32)
        os_p, osk = _ptr(out_sc, np.float32)
        stats = np.zeros((Q, 2), np.int64)
        mask_p, stride, mask = None, 0, None
        if accept is not None:
            mask = pack_accept_bits(accept, self.graph.n_nodes)
            if mask.ndim == 2:
                if mask.shape[0] != Q:
                    raise ValueError(f"accept has {mask.shape[0]} rows for {Q} queries")
                stride = int(mask.shape[1])
            mask_p = C.c_void_p(mask.ctypes.data)
        check(self.ctx._lib.jv_hip_graph_search_filtered(
            self.ctx._h, self.graph._h, self.luts._h, self.cv._h, self.fused._h if self.fused is not None else None,
            self.vectors._h if self.vectors is not None else None, q_p, Q, int(vsf), int(top_k), int(rerank_k), mask_p, stride,
            oi_p, os_p, C.c_void_p(stats.ctypes.data) if return_stats else None))
        return (out_ids, out_sc, stats) if return_stats else (out_ids, out_sc)

    def _session(self):
        if getattr(self, "_s", None) is None:
            h = C.c_void_p()
            check(self.ctx._lib.jv_hip_searcher_create(
                self.ctx._h, self.graph._h, self.luts._h, self.cv._h, self.fused._h if self.fused is not None else None,
                self.vectors._h if self.vectors is not None else None, C.byref(h)))
            self._s = h
        return self._s

    def _results(self, Q, top_k, call):
        ids, sc = np.empty((Q, top_k), np.int32), np.empty((Q, top_k), np.float32)
        counts, stats, worst = np.zeros(Q, np.int32), np.zeros((Q, 4), np.int64), np.zeros(Q, np.float32)
        check(call(C.c_void_p(ids.ctypes.data), C.c_void_p(sc.ctypes.data), C.c_void_p(counts.ctypes.data),
                   C.c_void_p(stats.ctypes.data), C.c_void_p(worst.ctypes.data)))
        return [SearchResult(ids[q, : counts[q]].copy(), sc[q, : counts[q]].copy(), stats[q], worst[q]) for q in range(Q)]

    def search_ex(self, queries, vsf, top_k, rerank_k, threshold=0.0, rerank_floor=0.0, accept=None):
        """GraphSearcher.search(scoreProvider, topK, rerankK, threshold, rerankFloor, acceptOrds) (GraphSearcher.java:222-243)
        for every query of the batch, as one GraphSearcher OBJECT per query: the state stays behind for resume().  Returns a
        list of SearchResult (nodes best first, visited / expanded / expanded_base / reranked counts,
        worst_approximate_in_topk).  threshold > 0 and rerank_floor behave as in the reference (TwoPhaseTracker,
        NodeQueue.rerank).  search() runs on the device traversal where its session kernels apply (any 256-cluster quantizer, degree <= 512), else on the host
        batched searcher; resume() likewise (the session kernel replays the searcher's earlier calls and continues)."""
        Q = int(queries.shape[0])
        q_p, qk = _ptr(queries, np.float32)
        mask_p, stride, mask = None, 0, None
        if accept is not None:
            mask = pack_accept_bits(accept, self.graph.n_nodes)
            if mask.ndim == 2:
                if mask.shape[0] != Q:
                    raise ValueError(f"accept has {mask.shape[0]} rows for {Q} queries")
                stride = int(mask.shape[1])
            mask_p = C.c_void_p(mask.ctypes.data)
        s = self._session()
        self._session_q = Q
        return self._results(Q, int(top_k), lambda i, sc, c, st, w: self.ctx._lib.jv_hip_searcher_search(
            self.ctx._h, s, q_p, Q, int(vsf), int(top_k), int(rerank_k), float(threshold), float(rerank_floor), mask_p, stride,
            i, sc, c, st, w))

    def resume(self, additional_k, rerank_k):
        """GraphSearcher.resume(additionalK, rerankK) (:538-547) for every query of the last search_ex: the next
        additional_k results, none of them returned before."""
        s = self._session()
        return self._results(int(getattr(self, "_session_q", 0)), int(additional_k),
                             lambda i, sc, c, st, w: self.ctx._lib.jv_hip_searcher_resume(
                                 self.ctx._h, s, int(additional_k), int(rerank_k), i, sc, c, st, w))

    def close(self):
        if getattr(self, "_s", None) is not None:
            self.ctx._lib.jv_hip_searcher_destroy(self._s)
            self._s = None


class SearchResult:
    """SearchResult (B/graph/SearchResult.java): getNodes() best first + the counters of the call that produced it."""

    def __init__(self, ids, scores, stats, worst):
        self.ids, self.scores = ids, scores
        self.visited, self.expanded, self.expanded_base, self.reranked = (int(x) for x in stats)
        self.worst_approximate_in_topk = float(worst)

    def __len__(self):
        return len(self.ids)
