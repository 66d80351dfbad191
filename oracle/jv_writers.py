"""TEST INFRASTRUCTURE ONLY (like everything under oracle/): byte-level restatement of the reference's WRITERS, used to
produce inputs for the format readers in jvector_amd/csrc/formats.cpp.  Never imported by the product.

parity unpinned for OnDiskGraphIndex: the reference ships no .odgi fixture (its tests write indexes into temp
directories with a JVM, which is absent here), so these writers are checked only against the reference's source:
  CommonHeader.write            B/graph/disk/CommonHeader.java:78-112
  Header.write                  B/graph/disk/Header.java:54-78
  L0 records                    B/graph/disk/OnDiskSequentialGraphIndexWriter.java:106-153,
                                B/graph/disk/NodeRecordTask.java:119-195 (OMITTED ordinals: zero features, no neighbours)
  sparse levels + v6 hierarchy  B/graph/disk/AbstractGraphIndexWriter.java:209-282
  separated features            B/graph/disk/AbstractGraphIndexWriter.java:284-309
  footer                        B/graph/disk/AbstractGraphIndexWriter.java:174-187
  FusedPQ.writeInline           B/graph/disk/feature/FusedPQ.java:146-161  (neighbour codes in neighbour order, zero padded)
  PQVectors.write               B/quantization/PQVectors.java:155-166
  NVQuantization.write          B/quantization/NVQuantization.java:260-277
  QuantizedVector.write         B/quantization/NVQuantization.java:437-443, QuantizedSubVector.write :577-587
  NVQVectors.write              B/quantization/NVQVectors.java:50-62
  NVQ / SeparatedNVQ headers    B/graph/disk/feature/NVQ.java:64-76, SeparatedNVQ.java:78-95
  fvecs / ivecs                 EX/util/SiftLoader.java:37-83 (little-endian)
The ProductQuantization block itself IS pinned (tests/golden/version0.pq, oracle.OraclePQ.serialize/parse).
All JVector output is big-endian (B/disk/IndexWriter.java:36-42).
"""
import struct

import numpy as np

ODGI_MAGIC = 0xFFFF0D61
FOOTER_MAGIC = 0x4A564244
V4_MAX_LAYERS = 32
INLINE_VECTORS, FUSED_PQ, NVQ_VECTORS, SEPARATED_VECTORS, SEPARATED_NVQ = range(5)


def _i32(*v):
    return struct.pack(">%di" % len(v), *[x if x < 2 ** 31 else x - 2 ** 32 for x in v])


def _be_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32).astype(">f4").tobytes()


def _be_i32(a):
    return np.ascontiguousarray(a, dtype=np.int32).astype(">i4").tobytes()


def write_pqvectors(pq_block: bytes, codes: np.ndarray) -> bytes:
    n, M = codes.shape
    return pq_block + _i32(n, M) + np.ascontiguousarray(codes, dtype=np.uint8).tobytes()


def nvq_sizes(D, S):
    """NVQuantization.getSubvectorSizesAndOffsets :236-252"""
    return [D // S + (1 if i < D % S else 0) for i in range(S)]


def write_nvq_block(mean: np.ndarray, S: int, version=6) -> bytes:
    D = len(mean)
    return _i32(version, D) + _be_f32(mean) + _i32(8, S) + _i32(*nvq_sizes(D, S))


def write_quantized_vector(sizes, row_bytes: np.ndarray, row_params: np.ndarray) -> bytes:
    """row_params: S x {minValue, maxValue, growthRate, midpoint}"""
    out = _i32(len(sizes))
    off = 0
    for s, size in enumerate(sizes):
        out += _i32(8) + _be_f32(row_params[s]) + _i32(size, size) + np.ascontiguousarray(row_bytes[off:off + size], np.uint8).tobytes()
        off += size
    return out


def write_nvqvectors(mean, S, bytes_, params, version=6) -> bytes:
    sizes = nvq_sizes(len(mean), S)
    out = bytearray(write_nvq_block(mean, S, version) + _i32(len(bytes_)))
    for r in range(len(bytes_)):
        out += write_quantized_vector(sizes, bytes_[r], params[r])
    return bytes(out)


def write_xvecs(rows: np.ndarray) -> bytes:
    rows = np.ascontiguousarray(rows)
    assert rows.dtype in (np.float32, np.int32)
    out = bytearray()
    for r in rows:
        out += struct.pack("<i", len(r)) + r.astype("<" + ("f4" if rows.dtype == np.float32 else "i4")).tobytes()
    return bytes(out)


def _common_header(version, dimension, entry_node, layers, id_upper_bound):
    out = b""
    if version >= 3:
        out += _i32(ODGI_MAGIC, version)
    out += _i32(layers[0][0], dimension, entry_node, layers[0][1])
    if version >= 4:
        out += _i32(id_upper_bound, len(layers))
        for size, degree in layers:
            out += _i32(size, degree)
        out += _i32(0, 0) * (V4_MAX_LAYERS - len(layers))
    else:
        assert len(layers) == 1
    return out


def _header(version, dimension, entry_node, layers, id_upper_bound, feature_order, pq_block, sep_offset, nvq_block=None):
    out = _common_header(version, dimension, entry_node, layers, id_upper_bound)

    def feature_header(fid):
        if fid == FUSED_PQ:
            return pq_block
        if fid == SEPARATED_VECTORS:
            return struct.pack(">q", sep_offset)
        if fid == NVQ_VECTORS:
            return nvq_block
        if fid == SEPARATED_NVQ:
            return nvq_block + struct.pack(">q", sep_offset)
        return b""

    if version >= 6:
        out += _i32(len(feature_order))
        for fid in feature_order:
            out += _i32(fid) + feature_header(fid)
    else:
        if version >= 3:
            out += _i32(sum(1 << f for f in feature_order))
        for fid in sorted(feature_order):
            out += feature_header(fid)
    return out


def write_odgi(version, dimension, l0_neighbors, degree0, entry_node, upper_levels=(), vectors=None, separated=False,
               codes=None, pq_block=None, omitted=(), level_file_order=None, sequential_placeholders=False, separated_holes_as_zero_bytes=True,
               placeholder_fill=0, nvq=None, nvq_separated=False) -> bytes:
    """l0_neighbors: list (per ordinal) of neighbour-id lists;  upper_levels: [(degree, {node: [neighbours]}), ...] for
    levels 1..;  vectors: N x D float32 (inline, or separated when `separated`);  codes + pq_block: adds FUSED_PQ (v6);
    omitted: ordinals written as placeholders;  level_file_order: optional {level: [node ids in file order]}.
    sequential_placeholders: write OMITTED ordinals the way the sequential OnDiskGraphIndexWriter does
    (OnDiskGraphIndexWriter.java:101-110: ordinal -1, inline feature bytes seek-skipped — whatever the file held, modelled
    by `placeholder_fill` — count 0, -1 padding) instead of NodeRecordTask's zero-feature record that keeps the ordinal.
    nvq: (mean[D], S, bytes[N, D], params[N, S, 4]) adds NVQ_VECTORS (inline) or, with nvq_separated, SEPARATED_NVQ (an
    omitted ordinal gets QuantizedVector.createEmpty: zero bytes and parameters, SeparatedNVQ.java:86-94)."""
    N = len(l0_neighbors)
    assert not (nvq is not None and nvq_separated and vectors is not None and separated), "one separated feature at a time"
    nvq_block = write_nvq_block(nvq[0], nvq[1], version) if nvq is not None else None
    nsizes = nvq_sizes(dimension, nvq[1]) if nvq is not None else None
    nvq_stride = 4 + sum(28 + z for z in nsizes) if nvq is not None else 0

    def nvq_record(i, empty):
        if empty:
            return write_quantized_vector(nsizes, np.zeros(dimension, np.uint8), np.zeros((nvq[1], 4), np.float32))
        return write_quantized_vector(nsizes, nvq[2][i], nvq[3][i])

    layers = [(N - len(omitted), degree0)] + [(len(nodes), deg) for deg, nodes in upper_levels]
    fused = codes is not None
    assert not fused or version >= 6
    feats = []
    if vectors is not None:
        feats.append(SEPARATED_VECTORS if separated else INLINE_VECTORS)
    if fused:
        feats.append(FUSED_PQ)
    if nvq is not None:
        feats.append(SEPARATED_NVQ if nvq_separated else NVQ_VECTORS)
    # v6 orders features with fused ones last, then by id (AbstractFeature.compareTo :20-25, AbstractGraphIndexWriter
    # :83-90); <= v5 uses FeatureId order.
    feats.sort(key=(lambda f: (f == FUSED_PQ, f)) if version >= 6 else None)
    M = codes.shape[1] if fused else 0

    def hdr(sep_off):
        return _header(version, dimension, entry_node, layers, N, feats, pq_block, sep_off, nvq_block)

    out = bytearray(hdr(0))
    for i in range(N):
        if sequential_placeholders and i in omitted:
            out += _i32(-1)
            for fid in feats:
                if fid == INLINE_VECTORS:
                    out += bytes([placeholder_fill]) * (4 * dimension)
                elif fid == FUSED_PQ:
                    out += bytes([placeholder_fill]) * (degree0 * M)
                elif fid == NVQ_VECTORS:
                    out += bytes([placeholder_fill]) * nvq_stride
            out += _i32(0) + _be_i32([-1] * degree0)
            continue
        out += _i32(i)
        nb = [] if i in omitted else list(l0_neighbors[i])
        assert len(nb) <= degree0
        for fid in feats:
            if fid == INLINE_VECTORS:
                out += _be_f32(np.zeros(dimension, np.float32) if i in omitted else vectors[i])
            elif fid == FUSED_PQ:
                blk = np.zeros((degree0, M), dtype=np.uint8)
                if nb:
                    blk[:len(nb)] = codes[nb]
                out += blk.tobytes()
            elif fid == NVQ_VECTORS:
                out += nvq_record(i, i in omitted)
        out += _i32(len(nb)) + _be_i32(nb + [-1] * (degree0 - len(nb)))
    for lvl, (deg, nodes) in enumerate(upper_levels, start=1):
        order = (level_file_order or {}).get(lvl, list(nodes.keys()))
        assert sorted(order) == sorted(nodes.keys())
        for node in order:
            nb = list(nodes[node])
            out += _i32(node, len(nb)) + _be_i32(nb + [-1] * (deg - len(nb)))
    if version == 6 and fused:
        if upper_levels:
            order = (level_file_order or {}).get(1, list(upper_levels[0][1].keys()))
            for node in order:
                out += _i32(node) + codes[node].tobytes()
        else:
            out += _i32(entry_node) + codes[entry_node].tobytes()
    sep_off = 0
    if vectors is not None and separated:
        sep_off = len(out)
        vv = np.array(vectors, dtype=np.float32, copy=True)
        for i in omitted:
            vv[i] = 0
        out += _be_f32(vv)
    if nvq is not None and nvq_separated:
        sep_off = len(out)
        for i in range(N):
            rec = nvq_record(i, i in omitted)
            # an OMITTED ordinal gets featureSize ZERO bytes (AbstractGraphIndexWriter.writeSeparatedFeatures :298-307), not an empty
            # QuantizedVector (SeparatedNVQ.writeSeparately :86-94 does that for a present ordinal whose state carries no vector)
            out += bytes(len(rec)) if (i in omitted and separated_holes_as_zero_bytes) else rec
    if version >= 5:
        header_off = len(out)
        out += hdr(sep_off) + struct.pack(">q", header_off) + _i32(FOOTER_MAGIC)
    elif (separated and vectors is not None) or (nvq is not None and nvq_separated):
        # pre-footer versions rewrite the leading header in place once the offset is known
        h = hdr(sep_off)
        out[:len(h)] = h
    return bytes(out)
