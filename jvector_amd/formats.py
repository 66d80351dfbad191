"""Ingest what a real JVector writes: ProductQuantization / PQVectors blobs, OnDiskGraphIndex files, fvecs/ivecs.

Thin ctypes mirror of include/jvector_formats.h.  The parsing is host code inside libjvector_hip.so (no device needed:
`describe_*` / `read_*` work on a machine without a GPU); the `load_*` functions then hand the unpacked sections to
the device objects of engine.py.  Reference: PQVectors.load (B/quantization/PQVectors.java:54-75),
OnDiskGraphIndex.load (B/graph/disk/OnDiskGraphIndex.java:235-316), SiftLoader (EX/util/SiftLoader.java:37-83).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from ._lib import OdgiInfo, check, load

FEATURE_NAMES = ("INLINE_VECTORS", "FUSED_PQ", "NVQ_VECTORS", "SEPARATED_VECTORS", "SEPARATED_NVQ")


def _buf(data):
    """bytes / bytearray / memoryview / np.uint8 array (e.g. np.memmap) -> (uint8 view, void*, length); no copy."""
    a = data if isinstance(data, np.ndarray) else np.frombuffer(data, dtype=np.uint8)
    if a.dtype != np.uint8 or a.ndim != 1 or not a.flags.c_contiguous:
        raise ValueError("expected a contiguous byte buffer")
    return a, C.c_void_p(a.ctypes.data if a.size else None), a.size


# ---- fvecs / ivecs ---------------------------------------------------------------------------------------------
def _read_xvecs(data, dtype):
    lib = load()
    a, p, n = _buf(data)
    rows, dim = C.c_int64(), C.c_int()
    check(lib.jv_fmt_xvecs_describe(p, n, C.byref(rows), C.byref(dim)))
    out = np.empty((rows.value, dim.value), dtype=dtype)
    check(lib.jv_fmt_xvecs_read(p, n, C.c_void_p(out.ctypes.data)))
    return out


def read_fvecs(data) -> np.ndarray:
    """SiftLoader.readFvecs (:37-58): rows x dim float32."""
    return _read_xvecs(data, np.float32)


def read_ivecs(data) -> np.ndarray:
    """SiftLoader.readIvecs (:60-83): rows x dim int32 (ground-truth neighbour lists)."""
    return _read_xvecs(data, np.int32)


# ---- ProductQuantization / PQVectors ---------------------------------------------------------------------------
@dataclass
class PQDescription:
    block_len: int
    version: int
    dimension: int
    subspaces: int
    clusters: int
    has_centroid: bool
    anisotropic_threshold: float


def describe_pq(data) -> PQDescription:
    lib = load()
    a, p, n = _buf(data)
    bl, ver, D, M, k, hc, an = C.c_size_t(), C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_float()
    check(lib.jv_fmt_pq_describe(p, n, C.byref(bl), C.byref(ver), C.byref(D), C.byref(M), C.byref(k), C.byref(hc),
                                 C.byref(an)))
    return PQDescription(bl.value, ver.value, D.value, M.value, k.value, bool(hc.value), an.value)


def describe_pqvectors(data):
    """-> (pq_block_len, count, M, codes_off) of a PQVectors.write blob."""
    lib = load()
    a, p, n = _buf(data)
    bl, cnt, M, off = C.c_size_t(), C.c_int64(), C.c_int(), C.c_size_t()
    check(lib.jv_fmt_pqvectors_describe(p, n, C.byref(bl), C.byref(cnt), C.byref(M), C.byref(off)))
    return bl.value, cnt.value, M.value, off.value


def pqvectors_codes(data) -> np.ndarray:
    """The count x M code table of a PQVectors blob as a zero-copy uint8 view."""
    a, _, _ = _buf(data)
    _, cnt, M, off = describe_pqvectors(a)
    return a[off:off + cnt * M].reshape(cnt, M)


def load_pqvectors(ctx, data):
    """PQVectors.load: -> (ProductQuantization, PQVectors) resident on ctx's device."""
    from .engine import PQVectors, ProductQuantization
    a, _, _ = _buf(data)
    bl, cnt, M, off = describe_pqvectors(a)
    pq = ProductQuantization.load(ctx, a[:bl].tobytes())
    return pq, PQVectors(ctx, pq, np.ascontiguousarray(a[off:off + cnt * M].reshape(cnt, M)))


# ---- NVQuantization / NVQVectors -------------------------------------------------------------------------------
def describe_nvq(data):
    """-> (block_len, version, D, S, vector_stride) of an NVQuantization.write block"""
    lib = load()
    a, p, n = _buf(data)
    bl, ver, D, S, mo, st = C.c_size_t(), C.c_int(), C.c_int(), C.c_int(), C.c_size_t(), C.c_int64()
    check(lib.jv_fmt_nvq_describe(p, n, C.byref(bl), C.byref(ver), C.byref(D), C.byref(S), C.byref(mo), C.byref(st)))
    return bl.value, ver.value, D.value, S.value, st.value


def read_nvq_mean(data) -> np.ndarray:
    lib = load()
    a, p, n = _buf(data)
    _, _, D, _, _ = describe_nvq(a)
    mean = np.empty(D, np.float32)
    check(lib.jv_fmt_nvq_read_mean(p, n, C.c_void_p(mean.ctypes.data)))
    return mean


def read_nvqvectors(data):
    """NVQVectors.load (B/quantization/NVQVectors.java:64-84), host only: -> (mean[D], S, bytes[count, D], params[count, S, 4])"""
    lib = load()
    a, p, n = _buf(data)
    bl, cnt, off, st = C.c_size_t(), C.c_int64(), C.c_size_t(), C.c_int64()
    check(lib.jv_fmt_nvqvectors_describe(p, n, C.byref(bl), C.byref(cnt), C.byref(off), C.byref(st)))
    _, _, D, S, _ = describe_nvq(a[:bl.value])
    mean = read_nvq_mean(a[:bl.value])
    b = np.empty((cnt.value, D), np.uint8)
    prm = np.empty((cnt.value, S, 4), np.float32)
    body = a[off.value:]
    check(lib.jv_fmt_nvq_unpack(C.c_void_p(body.ctypes.data if body.size else None), body.size, st.value, cnt.value, D, S,
                                C.c_void_p(b.ctypes.data), C.c_void_p(prm.ctypes.data)))
    return mean, S, b, prm


def load_nvqvectors(ctx, data):
    """NVQVectors.load -> (NVQuantization, NVQVectors) resident on ctx's device"""
    from .engine import NVQuantization, NVQVectors
    mean, S, b, prm = read_nvqvectors(data)
    nvq = NVQuantization.create(ctx, mean, S)
    return nvq, NVQVectors(ctx, nvq, b, prm)


# ---- OnDiskGraphIndex ------------------------------------------------------------------------------------------
@dataclass
class OnDiskGraph:
    """Host-side unpacking of one OnDiskGraphIndex file (numpy arrays in host byte order)."""
    version: int
    dimension: int
    entry_node: int
    entry_level: int
    id_upper_bound: int
    features: tuple
    levels: list                      # [(None, nbrs[N, deg0])] + [(ids[size_l], nbrs[size_l, deg_l]) ...]
    vectors: np.ndarray | None        # N x D float32 (inline or separated), None if the index stores none
    fused_blocks: np.ndarray | None   # N x deg0 x M uint8
    pq_bytes: bytes | None            # FusedPQ's ProductQuantization block
    hierarchy_nodes: np.ndarray | None = None
    hierarchy_codes: np.ndarray | None = None
    info: OdgiInfo = field(default=None, repr=False)
    nvq_block: bytes | None = None          # NVQ_VECTORS / SEPARATED_NVQ header: the NVQuantization block
    nvq_bytes: np.ndarray | None = None     # N x D uint8
    nvq_params: np.ndarray | None = None    # N x S x 4 float32 {minValue, maxValue, growthRate, midpoint}

    def codes_from_fused(self) -> np.ndarray:
        """Rebuild the N x M code table from the fused blocks (+ hierarchy source codes): a node's code sits in the
        block of every node that lists it as a neighbour.  Nodes nobody points at (unreachable from any other node)
        and that are not hierarchy nodes keep an all-zero code.  For upper-level scoring without a PQVectors file
        (the reference scores the hierarchy from the same cached codes, FusedPQDecoder.java:85-111)."""
        if self.fused_blocks is None:
            raise ValueError("the index has no FUSED_PQ feature")
        nbrs = self.levels[0][1]
        N, deg = nbrs.shape
        M = self.fused_blocks.shape[2]
        codes = np.zeros((N, M), dtype=np.uint8)
        valid = nbrs >= 0
        codes[nbrs[valid]] = self.fused_blocks[valid]
        if self.hierarchy_nodes is not None and len(self.hierarchy_nodes):
            codes[self.hierarchy_nodes] = self.hierarchy_codes
        return codes


def describe_odgi(data) -> OdgiInfo:
    lib = load()
    a, p, n = _buf(data)
    info = OdgiInfo()
    check(lib.jv_fmt_odgi_describe(p, n, C.byref(info)))
    return info


def read_odgi(data, want_vectors=True) -> OnDiskGraph:
    """Unpack every section the engine consumes.  Host only (no GPU needed)."""
    lib = load()
    a, p, n = _buf(data)
    info = describe_odgi(a)
    N, D, deg0, M = info.id_upper_bound, info.dimension, info.layer_degree[0], info.pq_M
    nbrs = np.empty((N, deg0), dtype=np.int32)
    has_vec = info.inline_vectors_off >= 0 or info.separated_vectors_off >= 0
    vectors = np.empty((N, D), dtype=np.float32) if (has_vec and want_vectors) else None
    fused = np.empty((N, deg0, M), dtype=np.uint8) if info.fused_off >= 0 else None
    check(lib.jv_fmt_odgi_read_l0(p, n, C.byref(info), C.c_void_p(nbrs.ctypes.data),
                                  C.c_void_p(vectors.ctypes.data) if vectors is not None else None,
                                  C.c_void_p(fused.ctypes.data) if fused is not None else None))
    levels = [(None, nbrs)]
    for lvl in range(1, info.n_layers):
        ids = np.empty(info.layer_size[lvl], dtype=np.int32)
        nb = np.empty((info.layer_size[lvl], info.layer_degree[lvl]), dtype=np.int32)
        check(lib.jv_fmt_odgi_read_level(p, n, C.byref(info), lvl, C.c_void_p(ids.ctypes.data), C.c_void_p(nb.ctypes.data)))
        levels.append((ids, nb))
    h_nodes = h_codes = None
    if info.hierarchy_off >= 0:
        h_nodes = np.empty(info.hierarchy_count, dtype=np.int32)
        h_codes = np.empty((info.hierarchy_count, M), dtype=np.uint8)
        check(lib.jv_fmt_odgi_read_hierarchy_codes(p, n, C.byref(info), C.c_void_p(h_nodes.ctypes.data),
                                                   C.c_void_p(h_codes.ctypes.data)))
    pq_bytes = a[info.pq_off:info.pq_off + info.pq_len].tobytes() if info.pq_off >= 0 else None
    feats = tuple(FEATURE_NAMES[info.feature_id[i]] for i in range(info.n_features))
    nvq_block = nvq_b = nvq_p = None
    if info.nvq_off >= 0:
        nvq_block = a[info.nvq_off:info.nvq_off + info.nvq_len].tobytes()
        nvq_b = np.empty((N, D), dtype=np.uint8)
        nvq_p = np.empty((N, info.nvq_S, 4), dtype=np.float32)
        check(lib.jv_fmt_odgi_read_nvq(p, n, C.byref(info), C.c_void_p(nvq_b.ctypes.data), C.c_void_p(nvq_p.ctypes.data)))
    return OnDiskGraph(info.version, D, info.entry_node, info.entry_level, N, feats, levels, vectors, fused, pq_bytes,
                       h_nodes, h_codes, info, nvq_block, nvq_b, nvq_p)


@dataclass
class LoadedIndex:
    """Device objects built from one OnDiskGraphIndex file (+ optionally a PQVectors file)."""
    graph: object
    pq: object
    pq_vectors: object
    fused: object
    vectors: object
    host: OnDiskGraph
    nvq: object = None
    nvq_vectors: object = None

    def searcher(self, max_queries=4096):
        from .engine import GraphSearcher
        return GraphSearcher(self.graph.ctx, self.graph, self.pq, self.pq_vectors, fused=self.fused, vectors=self.vectors,
                             max_queries=max_queries)


def load_index(ctx, odgi_data, pqvectors_data=None) -> LoadedIndex:
    """OnDiskGraphIndex.load (+ PQVectors.load) -> GraphIndex / FusedPQ / VectorSet / PQVectors on ctx's device.
    Without a PQVectors blob the code table is rebuilt from the fused blocks (OnDiskGraph.codes_from_fused)."""
    from .engine import FusedPQ, GraphIndex, PQVectors, ProductQuantization, VectorSet
    g = read_odgi(odgi_data)
    if g.entry_node < 0:
        raise ValueError("the index is empty (ENTRY_NODE_ABSENT)")
    pq = cv = None
    if pqvectors_data is not None:
        pq, cv = load_pqvectors(ctx, pqvectors_data)
        if cv.count() != g.id_upper_bound:
            raise ValueError(f"PQVectors holds {cv.count()} codes, the index {g.id_upper_bound} nodes")
    elif g.pq_bytes is not None:
        pq = ProductQuantization.load(ctx, g.pq_bytes)
        cv = PQVectors(ctx, pq, g.codes_from_fused())
    else:
        raise ValueError("no PQ codes: the index has no FUSED_PQ feature and no PQVectors blob was given")
    fused = FusedPQ(ctx, pq, g.fused_blocks.reshape(g.id_upper_bound, -1), g.levels[0][1]) if g.fused_blocks is not None else None
    vectors = VectorSet(ctx, g.vectors) if g.vectors is not None else None
    nvq = nvq_vectors = None
    if g.nvq_block is not None:
        # NVQ_VECTORS / SEPARATED_NVQ: the reranker is the NVQ score function; View.rerankerFor (OnDiskGraphIndex.java:705-713)
        # prefers INLINE_VECTORS when the index carries both
        from .engine import NVQuantization, NVQVectors
        nvq = NVQuantization.create(ctx, read_nvq_mean(g.nvq_block), g.nvq_params.shape[1])
        nvq_vectors = NVQVectors(ctx, nvq, g.nvq_bytes, g.nvq_params)
        if vectors is None:
            vectors = nvq_vectors.as_vector_set()
    graph = GraphIndex(ctx, g.id_upper_bound, g.levels, g.entry_node, g.entry_level)
    return LoadedIndex(graph, pq, cv, fused, vectors, g, nvq, nvq_vectors)


# ---- writers: the byte formats JVector reads back ---------------------------------------------------------------
ODGI_MAGIC, FOOTER_MAGIC, V4_MAX_LAYERS = 0xFFFF0D61, 0x4A564244, 32
FEATURE_ID = {name: i for i, name in enumerate(FEATURE_NAMES)}


def _be_i32(values) -> bytes:
    return np.asarray(values, dtype=np.int64).astype(np.uint32).astype(">u4").tobytes()


def write_pqvectors(pq, pq_vectors, version=6) -> bytes:
    """PQVectors.write (B/quantization/PQVectors.java:155-166): PQ block, count, M, then the codes ordinal-major
    (chunk boundaries are invisible on disk).  pq: ProductQuantization, pq_vectors: PQVectors (device resident)."""
    n = pq_vectors.count()
    codes = pq_vectors.get(0, n) if n else np.zeros((0, pq.M), np.uint8)
    return pq.write(version) + _be_i32([n, pq.M]) + np.ascontiguousarray(codes, np.uint8).tobytes()


def nvq_records(sizes, bytes_, params) -> np.ndarray:
    """count x stride uint8: QuantizedVector.write (B/quantization/NVQuantization.java:437-443; QuantizedSubVector.write :577-587)
    for every row.  bytes_: count x D uint8; params: count x S x 4 float32 {minValue, maxValue, growthRate, midpoint}"""
    b = np.ascontiguousarray(bytes_, np.uint8)
    n, S = b.shape[0], len(sizes)
    p = np.ascontiguousarray(params, np.float32).reshape(n, S, 4)
    be32 = lambda v: np.asarray(v, np.int64).astype(np.uint32).astype(">u4").view(np.uint8)  # noqa: E731
    cols, off = [np.broadcast_to(be32([S]), (n, 4))], 0
    for s_, size in enumerate(sizes):
        cols.append(np.broadcast_to(be32([8]), (n, 4)))
        cols.append(np.ascontiguousarray(p[:, s_, :]).astype(">f4").view(np.uint8).reshape(n, 16))
        cols.append(np.broadcast_to(be32([size, size]), (n, 8)))
        cols.append(b[:, off:off + size])
        off += size
    return np.concatenate(cols, axis=1)


def write_odgi(dimension, levels, entry_node, vectors=None, separated=False, fused_blocks=None, pq_block=None,
               hierarchy_codes=None, version=6, nvq=None, nvq_separated=False) -> bytes:
    """An OnDiskGraphIndex file (v6 layout; v4 / v5 without FusedPQ) from plain arrays — what OnDiskGraphIndexWriter /
    OnDiskSequentialGraphIndexWriter emit (header CommonHeader.java:78-112 + Header.java:54-78; L0 records
    OnDiskSequentialGraphIndexWriter.java:106-153; sparse levels and the v6 hierarchy block AbstractGraphIndexWriter.java:
    209-282; separated vectors :284-309; footer :174-187).
      levels          : [(None, nbrs0[N, deg0])] + [(ids_l[size_l], nbrs_l[size_l, deg_l]) ...], rows packed, -1 padded
      vectors         : N x D float32 -> INLINE_VECTORS, or SEPARATED_VECTORS when `separated`
      fused_blocks    : N x (deg0 * M) uint8 + pq_block (ProductQuantization.write bytes) -> FUSED_PQ (v6 only)
      hierarchy_codes : codes of the level-1 nodes in levels[1] order (or of the entry node for a single-layer graph),
                        required with FUSED_PQ
      nvq             : (NVQuantization.write bytes, bytes[N, D] uint8, params[N, S, 4]) -> NVQ_VECTORS, or SEPARATED_NVQ when
                        `nvq_separated` (NVQ.java:64-82, SeparatedNVQ.java:78-95); at most one separated feature per file here
    The writer does not reorder or validate the graph; it is the inverse of read_odgi."""
    if version < 4 or version > 6:
        raise ValueError("write_odgi supports versions 4..6")
    nbrs0 = np.ascontiguousarray(levels[0][1], np.int32)
    N, deg0 = nbrs0.shape
    fused = fused_blocks is not None
    if fused and version < 6:
        raise ValueError("Fused features require version 6 or higher")  # AbstractGraphIndexWriter.java:97-99
    if version < 5 and len(levels) > 1 and version < 4:
        raise ValueError("Multilayer graphs must be written with version 4 or higher")
    feats = []
    if vectors is not None:
        feats.append(FEATURE_ID["SEPARATED_VECTORS" if separated else "INLINE_VECTORS"])
    if fused:
        feats.append(FEATURE_ID["FUSED_PQ"])
    nvq_rows = None
    if nvq is not None:
        if nvq_separated and separated and vectors is not None:
            raise ValueError("write_odgi writes one separated feature per file")
        S_nvq = describe_nvq(nvq[0])[3]
        base_, rem_ = divmod(dimension, S_nvq)
        nvq_rows = nvq_records([base_ + (1 if i < rem_ else 0) for i in range(S_nvq)], nvq[1], nvq[2])
        feats.append(FEATURE_ID["SEPARATED_NVQ" if nvq_separated else "NVQ_VECTORS"])
    feats.sort(key=(lambda f: (f == FEATURE_ID["FUSED_PQ"], f)) if version >= 6 else None)  # AbstractFeature.compareTo
    layer_info = [(N, deg0)] + [(len(ids), nb.shape[1]) for ids, nb in levels[1:]]

    def header(sep_off):
        out = _be_i32([ODGI_MAGIC, version, N, dimension, entry_node, deg0, N, len(layer_info)])
        out += _be_i32([x for li in layer_info for x in li]) + _be_i32([0, 0] * (V4_MAX_LAYERS - len(layer_info)))

        def feature_header(fid):
            if fid == FEATURE_ID["FUSED_PQ"]:
                return pq_block
            if fid == FEATURE_ID["SEPARATED_VECTORS"]:
                return int(sep_off).to_bytes(8, "big")
            if fid == FEATURE_ID["NVQ_VECTORS"]:
                return bytes(nvq[0])
            if fid == FEATURE_ID["SEPARATED_NVQ"]:
                return bytes(nvq[0]) + int(sep_off).to_bytes(8, "big")
            return b""

        if version >= 6:
            out += _be_i32([len(feats)])
            for fid in feats:
                out += _be_i32([fid]) + feature_header(fid)
        else:
            out += _be_i32([sum(1 << f for f in feats)])
            for fid in sorted(feats):
                out += feature_header(fid)
        return out

    # L0 records, vectorised: [ordinal][inline features][degree][neighbours]
    cols = [np.arange(N, dtype=">u4").view(np.uint8).reshape(N, 4)]
    for fid in feats:
        if fid == FEATURE_ID["INLINE_VECTORS"]:
            cols.append(np.ascontiguousarray(vectors, np.float32).astype(">f4").view(np.uint8).reshape(N, 4 * dimension))
        elif fid == FEATURE_ID["FUSED_PQ"]:
            cols.append(np.ascontiguousarray(fused_blocks, np.uint8).reshape(N, -1))
        elif fid == FEATURE_ID["NVQ_VECTORS"]:
            cols.append(nvq_rows)
    degree = (nbrs0 >= 0).sum(axis=1).astype(np.int64)
    cols.append(degree.astype(np.uint32).astype(">u4").view(np.uint8).reshape(N, 4))
    cols.append(nbrs0.astype(np.int64).astype(np.uint32).astype(">u4").view(np.uint8).reshape(N, 4 * deg0))
    out = bytearray(header(0))
    out += np.concatenate(cols, axis=1).tobytes() if N else b""
    for ids, nb in levels[1:]:
        ids = np.asarray(ids, np.int64)
        nb = np.asarray(nb, np.int64)
        cnt = (nb >= 0).sum(axis=1)
        rec = np.concatenate([ids[:, None], cnt[:, None], nb], axis=1)
        out += rec.astype(np.uint32).astype(">u4").tobytes()
    if version == 6 and fused:
        if hierarchy_codes is None:
            raise ValueError("FUSED_PQ needs the hierarchy source codes")
        h_ids = np.asarray(levels[1][0] if len(levels) > 1 else [entry_node], np.int64)
        hc = np.ascontiguousarray(hierarchy_codes, np.uint8).reshape(len(h_ids), -1)
        out += np.concatenate([h_ids.astype(np.uint32).astype(">u4").view(np.uint8).reshape(len(h_ids), 4), hc], axis=1).tobytes()
    sep_off = 0
    if vectors is not None and separated:
        sep_off = len(out)
        out += np.ascontiguousarray(vectors, np.float32).astype(">f4").tobytes()
    if nvq is not None and nvq_separated:
        sep_off = len(out)
        out += nvq_rows.tobytes()
    if version >= 5:
        header_off = len(out)
        out += header(sep_off) + int(header_off).to_bytes(8, "big") + _be_i32([FOOTER_MAGIC])
    elif (separated and vectors is not None) or (nvq is not None and nvq_separated):
        h = header(sep_off)
        out[:len(h)] = h
    return bytes(out)
