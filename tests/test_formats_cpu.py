"""CPU tests for the format readers (include/jvector_formats.h, jvector_amd/formats.py): host code, no GPU.

Inputs: the reference's own fixtures where they exist (tests/golden/version0.pq, siftsmall_query.fvecs) and, for
OnDiskGraphIndex, files produced by the test-side writer restatement oracle/jv_writers.py (the reference ships no
.odgi fixture: "parity unpinned" for that format, see the writer's header)."""
import os
import struct

import numpy as np
import pytest

import jvector_amd.formats as F
from oracle import jv_writers as W
from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _pq(D=16, M=4, seed=0, centroid=False):
    rng = np.random.default_rng(seed)
    cbs = rng.standard_normal((M, 256, D // M)).astype(np.float32)
    c = rng.standard_normal(D).astype(np.float32) if centroid else None
    return O.OraclePQ(D, M, cbs, c)


def _graph(N, deg, rng, ragged=True):
    nb = []
    for i in range(N):
        cnt = int(rng.integers(0, deg + 1)) if ragged else deg
        cand = [int(x) for x in rng.permutation(N)[:cnt + 1] if x != i][:cnt]
        nb.append(cand)
    return nb


# ---- fvecs / ivecs ---------------------------------------------------------------------------------------------
def test_fvecs_golden_fixture():
    data = open(os.path.join(GOLDEN, "siftsmall_query.fvecs"), "rb").read()
    v = F.read_fvecs(data)
    assert v.shape == (100, 128) and v.dtype == np.float32
    raw = np.frombuffer(data, dtype="<f4").reshape(100, 129)[:, 1:]
    assert np.array_equal(v, raw)
    assert v[0, 0] == struct.unpack_from("<f", data, 4)[0]


def test_ivecs_golden_fixture():
    """the repo-shipped siftsmall ground truth (SiftLoader.readIvecs, EX/util/SiftLoader.java:63-83): 100 queries x their 100
    nearest base ordinals of the 10 000-vector siftsmall base set"""
    data = open(os.path.join(GOLDEN, "siftsmall_groundtruth.ivecs"), "rb").read()
    g = F.read_ivecs(data)
    assert g.shape == (100, 100) and g.dtype == np.int32
    raw = np.frombuffer(data, dtype="<i4").reshape(100, 101)
    assert (raw[:, 0] == 100).all() and np.array_equal(g, raw[:, 1:])
    assert g.min() >= 0 and g.max() < 10_000                       # ordinals of siftsmall_base (10 000 vectors)
    assert all(len(set(row)) == 100 for row in g.tolist())         # a neighbour list holds no ordinal twice
    assert len(np.unique(g[:, 0])) > 90                            # 100 different queries -> (almost) all different nearest neighbours


def test_xvecs_round_trip_and_errors():
    rng = np.random.default_rng(1)
    f = rng.standard_normal((7, 5)).astype(np.float32)
    i = rng.integers(-2 ** 31, 2 ** 31 - 1, (3, 9), dtype=np.int64).astype(np.int32)
    assert np.array_equal(F.read_fvecs(W.write_xvecs(f)), f)
    assert np.array_equal(F.read_ivecs(W.write_xvecs(i)), i)
    assert F.read_fvecs(b"").shape == (0, 0)
    with pytest.raises(ValueError, match="whole number"):
        F.read_fvecs(W.write_xvecs(f)[:-3])
    bad = bytearray(W.write_xvecs(f))
    struct.pack_into("<i", bad, 24, 4)  # second row claims another dimension
    with pytest.raises(ValueError, match="row 1 has dimension 4"):
        F.read_fvecs(bytes(bad))


# ---- ProductQuantization / PQVectors ---------------------------------------------------------------------------
def test_pq_describe_golden_version0():
    data = open(os.path.join(GOLDEN, "version0.pq"), "rb").read()
    d = F.describe_pq(data)
    ref, ver, aniso, consumed = O.OraclePQ.parse(data)
    assert d.block_len == len(data) == consumed and d.version == ver == 0
    assert (d.dimension, d.subspaces, d.clusters) == (ref.D, ref.M, ref.k)
    assert d.anisotropic_threshold == -1.0


@pytest.mark.parametrize("version", [0, 2, 3, 6])
@pytest.mark.parametrize("centroid", [False, True])
def test_pq_describe_all_versions(version, centroid):
    pq = _pq(centroid=centroid)
    blob = pq.serialize(version)
    d = F.describe_pq(blob + b"trailing")
    assert d.block_len == len(blob)
    assert (d.version, d.dimension, d.subspaces, d.clusters, d.has_centroid) == (version if version >= 3 else 0, 16, 4, 256, centroid)
    with pytest.raises(ValueError, match="truncated"):
        F.describe_pq(blob[:-1])


def test_pqvectors_describe_and_codes():
    pq = _pq()
    rng = np.random.default_rng(2)
    codes = rng.integers(0, 256, (1500, 4), dtype=np.uint8)  # > 1 chunk of 1024 in MutablePQVectors: invisible on disk
    blob = W.write_pqvectors(pq.serialize(6), codes)
    bl, cnt, M, off = F.describe_pqvectors(blob)
    assert (bl, cnt, M, off) == (len(pq.serialize(6)), 1500, 4, len(pq.serialize(6)) + 8)
    assert np.array_equal(F.pqvectors_codes(blob), codes)
    empty = W.write_pqvectors(pq.serialize(0), np.zeros((0, 4), np.uint8))
    assert F.describe_pqvectors(empty)[1] == 0
    with pytest.raises(ValueError, match="truncated"):
        F.describe_pqvectors(blob[:-1])
    bad = bytearray(blob)
    struct.pack_into(">i", bad, bl + 4, 5)
    with pytest.raises(ValueError, match="compressed dimension 5"):
        F.describe_pqvectors(bytes(bad))


# ---- OnDiskGraphIndex ------------------------------------------------------------------------------------------
def _packed(nb, deg):
    out = np.full((len(nb), deg), -1, np.int32)
    for i, r in enumerate(nb):
        out[i, :len(r)] = r
    return out


@pytest.mark.parametrize("version", [2, 3, 4, 5, 6])
def test_odgi_inline_vectors_single_layer(version):
    rng = np.random.default_rng(version)
    N, D, deg = 37, 6, 5
    nb = _graph(N, deg, rng)
    vec = rng.standard_normal((N, D)).astype(np.float32)
    blob = W.write_odgi(version, D, nb, deg, entry_node=11, vectors=vec)
    g = F.read_odgi(blob)
    assert (g.version, g.dimension, g.entry_node, g.entry_level, g.id_upper_bound) == (version, D, 11, 0, N)
    assert g.features == ("INLINE_VECTORS",)
    assert np.array_equal(g.levels[0][1], _packed(nb, deg)) and len(g.levels) == 1
    assert np.array_equal(g.vectors, vec)
    assert g.fused_blocks is None and g.pq_bytes is None and g.hierarchy_nodes is None
    assert g.info.record_stride == 4 + 4 * D + 4 * (1 + deg)


@pytest.mark.parametrize("separated", [False, True])
def test_odgi_v6_fused_multilayer(separated):
    rng = np.random.default_rng(7)
    N, D, M, deg = 60, 16, 4, 6
    pq = _pq(D, M, centroid=True)
    nb = _graph(N, deg, rng)
    vec = rng.standard_normal((N, D)).astype(np.float32)
    codes = rng.integers(0, 256, (N, M), dtype=np.uint8)
    l1 = {int(n): [int(x) for x in rng.choice([3, 9, 20, 41, 55], 2, replace=False) if x != n][:2] for n in [41, 3, 55, 9, 20]}
    l2 = {9: [41], 41: [9]}
    blob = W.write_odgi(6, D, nb, deg, entry_node=41, upper_levels=[(3, l1), (2, l2)], vectors=vec, separated=separated,
                        codes=codes, pq_block=pq.serialize(6), level_file_order={1: [41, 3, 55, 9, 20], 2: [41, 9]})
    g = F.read_odgi(blob)
    assert g.features == (("SEPARATED_VECTORS", "FUSED_PQ") if separated else ("INLINE_VECTORS", "FUSED_PQ"))
    assert (g.entry_node, g.entry_level) == (41, 2)
    assert np.array_equal(g.levels[0][1], _packed(nb, deg))
    assert np.array_equal(g.vectors, vec)
    # upper levels come back sorted by node id, rows permuted alike
    ids1, nb1 = g.levels[1]
    assert ids1.tolist() == [3, 9, 20, 41, 55] and nb1.shape == (5, 3)
    for n, row in zip(ids1, nb1):
        assert [x for x in row if x >= 0] == l1[int(n)]
    ids2, nb2 = g.levels[2]
    assert ids2.tolist() == [9, 41] and nb2.tolist() == [[41, -1], [9, -1]]
    # fused blocks: neighbour codes in neighbour order, zero padded (FusedPQ.writeInline)
    assert g.fused_blocks.shape == (N, deg, M)
    for i in (0, 17, N - 1):
        for j in range(deg):
            exp = codes[nb[i][j]] if j < len(nb[i]) else np.zeros(M, np.uint8)
            assert np.array_equal(g.fused_blocks[i, j], exp)
    assert g.hierarchy_nodes.tolist() == [41, 3, 55, 9, 20]
    assert np.array_equal(g.hierarchy_codes, codes[[41, 3, 55, 9, 20]])
    rt = O.OraclePQ.parse(g.pq_bytes)[0]
    assert (rt.D, rt.M) == (D, M) and g.pq_bytes == pq.serialize(6)
    # code table rebuilt from the blocks: exact for every node somebody points at or that sits in the hierarchy
    rebuilt = g.codes_from_fused()
    referenced = np.zeros(N, bool)
    referenced[[x for r in nb for x in r]] = True
    referenced[[41, 3, 55, 9, 20]] = True
    assert np.array_equal(rebuilt[referenced], codes[referenced])
    assert not rebuilt[~referenced].any()


def test_odgi_v6_fused_single_layer_entry_code_and_omitted():
    rng = np.random.default_rng(9)
    N, D, M, deg = 20, 8, 2, 4
    pq = _pq(D, M)
    nb = _graph(N, deg, rng, ragged=False)
    codes = rng.integers(0, 256, (N, M), dtype=np.uint8)
    vec = rng.standard_normal((N, D)).astype(np.float32)
    blob = W.write_odgi(6, D, nb, deg, entry_node=5, vectors=vec, codes=codes, pq_block=pq.serialize(6), omitted={7})
    g = F.read_odgi(blob)
    assert g.hierarchy_nodes.tolist() == [5] and np.array_equal(g.hierarchy_codes[0], codes[5])
    assert g.info.layer_size[0] == N - 1 and g.id_upper_bound == N
    assert (g.levels[0][1][7] == -1).all() and not g.vectors[7].any() and not g.fused_blocks[7].any()
    assert F.read_odgi(blob, want_vectors=False).vectors is None


@pytest.mark.parametrize("separated", [False, True])
def test_odgi_sequential_writer_placeholders(separated):
    """OnDiskGraphIndexWriter.writeL0Records :97-110 — an OMITTED ordinal is written as ordinal -1, its inline feature
    bytes are seek-skipped (whatever the file held: modelled as 0xAB garbage), count 0, -1 padding.  The reference's
    reader never looks at the stored ordinal; the record must load as a hole."""
    rng = np.random.default_rng(10)
    N, D, M, deg = 24, 8, 2, 4
    pq = _pq(D, M)
    holes = {3, 17, 23}
    nb = [[int(x) for x in rng.choice([j for j in range(N) if j not in holes and j != i], deg, replace=False)] for i in range(N)]
    codes = rng.integers(0, 256, (N, M), dtype=np.uint8)
    vec = rng.standard_normal((N, D)).astype(np.float32)
    blob = W.write_odgi(6, D, nb, deg, entry_node=5, vectors=vec, separated=separated, codes=codes, pq_block=pq.serialize(6),
                        omitted=holes, sequential_placeholders=True, placeholder_fill=0xAB)
    info = F.describe_odgi(blob)
    assert struct.unpack_from(">i", blob, info.l0_off + 17 * info.record_stride)[0] == -1
    g = F.read_odgi(blob)
    assert g.info.layer_size[0] == N - len(holes) and g.id_upper_bound == N
    for h in holes:
        assert (g.levels[0][1][h] == -1).all() and not g.fused_blocks[h].any() and not g.vectors[h].any()
    live = [i for i in range(N) if i not in holes]
    assert np.array_equal(g.levels[0][1][live], _packed(nb, deg)[live])
    assert np.array_equal(g.vectors[live], vec[live])
    for i in live[:5]:
        assert np.array_equal(g.fused_blocks[i], codes[nb[i]])
    # same graph through NodeRecordTask's variant (ordinal kept, zero features): identical arrays
    g2 = F.read_odgi(W.write_odgi(6, D, nb, deg, entry_node=5, vectors=vec, separated=separated, codes=codes,
                                  pq_block=pq.serialize(6), omitted=holes))
    assert np.array_equal(g.levels[0][1], g2.levels[0][1]) and np.array_equal(g.fused_blocks, g2.fused_blocks)
    assert np.array_equal(g.vectors, g2.vectors)


def test_odgi_no_vectors_and_empty():
    nb = [[1], [0]]
    g = F.read_odgi(W.write_odgi(6, 4, nb, 2, entry_node=0))
    assert g.features == () and g.vectors is None and g.levels[0][1].tolist() == [[1, -1], [0, -1]]
    e = F.read_odgi(W.write_odgi(6, 4, [], 2, entry_node=-1))
    assert e.entry_node == -1 and e.levels[0][1].shape == (0, 2)


def test_odgi_rejects_corruption():
    rng = np.random.default_rng(3)
    N, D, deg = 10, 4, 3
    nb = _graph(N, deg, rng, ragged=False)
    vec = rng.standard_normal((N, D)).astype(np.float32)
    blob = W.write_odgi(6, D, nb, deg, entry_node=0, vectors=vec)
    info = F.describe_odgi(blob)
    with pytest.raises(ValueError, match="footer magic"):
        F.describe_odgi(blob[:-1] + b"\x00")
    with pytest.raises(ValueError):
        F.describe_odgi(blob[:200])
    bad = bytearray(blob)
    struct.pack_into(">i", bad, info.l0_off + 3 * info.record_stride, 4)  # record 3 claims ordinal 4
    with pytest.raises(ValueError, match="record 3 carries ordinal 4"):
        F.read_odgi(bytes(bad))
    bad = bytearray(blob)
    struct.pack_into(">i", bad, info.l0_off + info.neighbors_off, deg + 1)
    with pytest.raises(ValueError, match="max degree"):
        F.read_odgi(bytes(bad))
    bad = bytearray(blob)
    struct.pack_into(">i", bad, info.l0_off + info.neighbors_off + 4, N)  # neighbour id out of range
    with pytest.raises(ValueError, match="out of range"):
        F.read_odgi(bytes(bad))
    bad = bytearray(blob)
    struct.pack_into(">i", bad, 4, 7)  # version 7
    with pytest.raises(ValueError, match="unsupported version 7"):
        F.describe_odgi(bytes(bad))


def test_odgi_nvq_header_is_validated():
    """NVQ features are decoded since round 3 (tests/test_nvq_cpu.py); what stays refused: a feature header that is not an
    NVQuantization block, and bit depths the reference's own loader rejects (BitsPerDimension.load)"""
    import jvector_amd as J
    # v6 header listing NVQ_VECTORS (feature ordinal 2) followed by zeros instead of an NVQuantization block
    hdr = W._common_header(6, 4, 0, [(1, 2)], 1) + W._i32(1, W.NVQ_VECTORS)
    with pytest.raises(ValueError, match="nvq"):
        F.describe_odgi(hdr + b"\0" * 64)
    four_bits = W._i32(6, 4) + W._be_f32(np.zeros(4, np.float32)) + W._i32(4, 1) + W._i32(4)
    with pytest.raises(J.UnsupportedError, match="Unsupported BitsPerDimension 4"):
        F.describe_odgi(hdr + four_bits + b"\0" * 64)


def test_readers_survive_corrupted_input():
    """The readers take bytes from files: random byte flips, planted extreme ints and truncations must end in a parse or
    in ValueError / UnsupportedError — never in a crash or an out-of-bounds read (sizes are validated against the buffer)."""
    import jvector_amd as J
    rng = np.random.default_rng(0)
    pq = _pq()
    N, D, M, deg = 40, 16, 4, 5
    nb = _graph(N, deg, rng)
    vec = rng.standard_normal((N, D)).astype(np.float32)
    codes = rng.integers(0, 256, (N, M), dtype=np.uint8)
    l1 = {3: [9], 9: [3, 20], 20: [9]}
    blobs = [W.write_odgi(6, D, nb, deg, 9, upper_levels=[(2, l1)], vectors=vec, codes=codes, pq_block=pq.serialize(6)),
             W.write_odgi(6, D, nb, deg, 9, vectors=vec, separated=True, codes=codes, pq_block=pq.serialize(6)),
             W.write_odgi(4, D, nb, deg, 9, upper_levels=[(2, l1)], vectors=vec), W.write_odgi(2, D, nb, deg, 9, vectors=vec)]
    pqv = W.write_pqvectors(pq.serialize(6), codes)
    parsed = rejected = 0
    for it in range(1500):
        b = bytearray(blobs[it % 4]) if it % 5 else bytearray(pqv)
        for _ in range(int(rng.integers(1, 4))):
            mode = rng.integers(0, 3)
            pos = int(rng.integers(0, min(len(b), 700))) if rng.random() < 0.7 else int(rng.integers(0, len(b)))
            if mode == 0:
                b[pos] = int(rng.integers(0, 256))
            elif mode == 1 and pos + 4 <= len(b):
                struct.pack_into(">i", b, pos - pos % 4, int(rng.choice([-1, 0, 1, 2 ** 31 - 1, -2 ** 31, 255, 65536, 10 ** 6])))
            else:
                b = b[:max(1, pos)]
        try:
            if it % 5:
                F.read_odgi(bytes(b))
            else:
                F.pqvectors_codes(bytes(b))
            parsed += 1
        except (ValueError, J.UnsupportedError):
            rejected += 1
    assert parsed > 100 and rejected > 100


@pytest.mark.parametrize("version,fused,separated", [(6, True, False), (6, True, True), (6, False, False), (5, False, True),
                                                     (4, False, False)])
def test_product_writer_matches_the_writer_restatement_and_round_trips(version, fused, separated):
    """jvector_amd.formats.write_odgi (vectorised) == oracle/jv_writers.py (the line-by-line restatement of the reference's
    writers) byte for byte, and read_odgi(write_odgi(x)) == x."""
    rng = np.random.default_rng(version * 7 + fused)
    N, D, M, deg = 60, 16, 4, 6
    pq = _pq(D, M, centroid=True)
    nb = _graph(N, deg, rng)
    vec = rng.standard_normal((N, D)).astype(np.float32)
    codes = rng.integers(0, 256, (N, M), dtype=np.uint8)
    l1 = {int(n): [int(x) for x in rng.choice([3, 9, 20, 41, 55], 2, replace=False) if x != n][:2] for n in [3, 9, 20, 41, 55]}
    l2 = {9: [41], 41: [9]}
    want = W.write_odgi(version, D, nb, deg, entry_node=41, upper_levels=[(3, l1), (2, l2)], vectors=vec, separated=separated,
                        codes=codes if fused else None, pq_block=pq.serialize(6) if fused else None,
                        level_file_order={1: [3, 9, 20, 41, 55], 2: [9, 41]})
    nb0 = _packed(nb, deg)
    lv1 = (np.array([3, 9, 20, 41, 55], np.int32), _packed([l1[k] for k in (3, 9, 20, 41, 55)], 3))
    lv2 = (np.array([9, 41], np.int32), _packed([l2[9], l2[41]], 2))
    blocks = np.where((nb0 >= 0)[:, :, None], codes[np.clip(nb0, 0, N - 1)], 0).astype(np.uint8).reshape(N, deg * M) if fused else None
    got = F.write_odgi(D, [(None, nb0), lv1, lv2], 41, vectors=vec, separated=separated, fused_blocks=blocks,
                       pq_block=pq.serialize(6) if fused else None, hierarchy_codes=codes[[3, 9, 20, 41, 55]] if fused else None,
                       version=version)
    assert got == want
    g = F.read_odgi(got)
    assert np.array_equal(g.levels[0][1], nb0) and np.array_equal(g.vectors, vec)
    assert np.array_equal(g.levels[1][0], lv1[0]) and np.array_equal(g.levels[1][1], lv1[1])
    if fused:
        assert np.array_equal(g.fused_blocks.reshape(N, -1), blocks)
