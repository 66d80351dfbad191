"""NVQ without a GPU: (1) the oracle's restatement (oracle/jv_nvq.c) against the reference's own test of this path
(TS/quantization/TestCompressedVectors.testNVQEncodings, TestReconstructionError), (2) against the second, independent
restatement the product ships for the reference's per-pair SPI (the nvq_* symbols of include/jvector_simd_compat.h, host code in
compat_host.cpp), (3) the byte formats (NVQuantization / NVQVectors / NVQ_VECTORS / SEPARATED_NVQ) from the oracle's writers
through the product's readers, (4) the C ABI's host logic of every NVQ entry point on the mock device."""
import ctypes as C
import os
import platform
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "mock"))

from oracle import jv_writers as W
from oracle import oracle as O


# ---- (1) the reference's own tests, restated on the oracle ------------------------------------------------------
@pytest.mark.parametrize("d", [256, 512])
def test_oracle_meets_the_reference_tolerances(d):
    """TestCompressedVectors.testNVQEncodings :171-228 (d = 256..2048 there; two sizes here keep the CPU suite short)"""
    rng = np.random.default_rng(d)
    X = rng.standard_normal((512, d)).astype(np.float32)          # createNormalRandomVectors(512, d)
    Qs = rng.standard_normal((10, d)).astype(np.float32)
    Qs /= np.linalg.norm(Qs, axis=1, keepdims=True)               # VectorUtil.l2normalize(q)
    ords = np.tile(np.arange(512, dtype=np.int32), (10, 1))
    vv = np.array([O.compare(1, X[j], X[j]) for j in range(512)], np.float32)
    for S in (1, 2, 4, 8):
        for learn in (False, True):
            o = O.OracleNVQ.compute(X, S, learn)
            o.encode_all(X)
            for vsf in (O.EUCLIDEAN, O.DOT_PRODUCT, O.COSINE):
                got = o.scores(Qs, vsf, ords)
                exact = np.stack([O.compare_many(vsf, Qs[i], X) for i in range(10)])
                err = float(np.mean(np.abs(got - exact) / np.abs(vv)[None, :])) if vsf == O.DOT_PRODUCT else float(np.mean(np.abs(got - exact)))
                tol = 0.0005 * (d / 256.0) * (10 if vsf == O.COSINE else 4 if vsf == O.DOT_PRODUCT else 1)
                assert err <= tol, (d, S, learn, vsf, err, tol)
            if learn:   # the optimisation is worth something: NVQ beats the learn = false setting on reconstruction
                e1 = np.mean([o.reconstruction_error(X[i]) for i in range(0, 512, 16)])
                o0 = O.OracleNVQ(o.mean, S, learn=False)
                e0 = np.mean([o0.reconstruction_error(X[i]) for i in range(0, 512, 16)])
                assert e1 <= e0


def test_oracle_reconstruction_error_is_stable_across_samples():
    """TestReconstructionError.testReconstructionError_withNVQuantization (:91-98, compareErrors :101-116): the error statistics of
    a second sample from the same distribution match the first's (relative tolerances of the reference's 1 000-vector case)"""
    rng = np.random.default_rng(4)
    a = rng.uniform(-1, 1, (1000, 32)).astype(np.float32)
    b = rng.uniform(-1, 1, (1000, 32)).astype(np.float32)
    o = O.OracleNVQ.compute(a, 2)
    e1 = np.array([o.reconstruction_error(v) for v in a])
    e2 = np.array([o.reconstruction_error(v) for v in b])
    assert abs(e2.mean() / e1.mean() - 1) <= 4e-2
    assert abs(e2.var() / e1.var() - 1) <= 0.25


def test_growth_rate_grid_fits_the_kernel():
    """QuantizedSubVector.quantizeTo's loops (:523-541) visit 20 coarse and at most 21 fine growth rates — what
    nvq_encode_kernel's 21 lanes per unit rely on (nvq.cpp refuses to start otherwise)"""
    coarse, fine = O.nvq_growth_grid()
    assert len(coarse) == 20 and coarse[0] == np.float32(1e-6)
    assert max(len(f) for f in fine) <= 21 and min(len(f) for f in fine) >= 20


# ---- (2) two independent restatements agree -----------------------------------------------------------------------
def test_oracle_equals_the_compat_spi_restatement():
    import jvector_amd._lib as L
    if not os.path.exists(L.LIB_PATH):
        pytest.skip("libjvector_hip.so not built")
    lib = C.CDLL(L.LIB_PATH)
    for name, (res, args) in L.COMPAT_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))      # noqa: E731
    u8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_ubyte))      # noqa: E731
    F = C.c_float
    rng = np.random.default_rng(8)
    ol = O.lib()
    for n in (1, 7, 64, 383):
        for trial in range(6):
            v = (rng.standard_normal(n) * rng.choice([0.01, 1.0, 30.0])).astype(np.float32)
            q = rng.standard_normal(n).astype(np.float32)
            cen = rng.standard_normal(n).astype(np.float32)
            lo, hi = float(v.min()), float(v.max())
            gr = float(rng.choice([1e-6, 1e-2, 0.7, 5.2, 19.0, -0.3]))
            mid = float(rng.choice([0.0, 0.1]))
            a = np.zeros(n, np.uint8)
            b = np.zeros(n, np.uint8)
            lib.nvq_quantize_8bit(fp(v), n, F(gr), F(mid), F(lo), F(hi), u8(a))
            ol.jvo_nvq_quantize_8bit(fp(v), n, F(gr), F(mid), F(lo), F(hi), b.ctypes.data_as(C.POINTER(C.c_uint8)))
            assert np.array_equal(a, b), (n, gr)
            same = lambda x, y: np.float32(x).view(np.uint32) == np.float32(y).view(np.uint32) or (x != x and y != y)  # noqa: E731
            assert same(lib.nvq_loss(fp(v), n, F(gr), F(mid), F(lo), F(hi), 8), ol.jvo_nvq_loss(fp(v), n, F(gr), F(mid), F(lo), F(hi), 8))
            assert same(lib.nvq_uniform_loss(fp(v), n, F(lo), F(hi), 8), ol.jvo_nvq_uniform_loss(fp(v), n, F(lo), F(hi), 8))
            bp = b.ctypes.data_as(C.POINTER(C.c_uint8))
            assert same(lib.nvq_dot_product_8bit(fp(q), u8(a), n, F(gr), F(mid), F(lo), F(hi)), ol.jvo_nvq_dot_8bit(fp(q), bp, n, F(gr), F(mid), F(lo), F(hi)))
            assert same(lib.nvq_square_l2_distance_8bit(fp(q), u8(a), n, F(gr), F(mid), F(lo), F(hi)),
                        ol.jvo_nvq_l2_8bit(fp(q), bp, n, F(gr), F(mid), F(lo), F(hi)))
            packed = lib.nvq_cosine_8bit_packed(fp(q), u8(a), n, F(gr), F(mid), F(lo), F(hi), fp(cen))
            out2 = np.zeros(2, np.float32)
            ol.jvo_nvq_cosine_8bit(fp(q), bp, n, F(gr), F(mid), F(lo), F(hi), fp(cen), fp(out2))
            got2 = np.array([packed & 0xFFFFFFFF, (packed >> 32) & 0xFFFFFFFF], np.uint32).view(np.float32)
            assert same(got2[0], out2[0]) and same(got2[1], out2[1])


# ---- (3) formats ---------------------------------------------------------------------------------------------------
def _rows(seed, n, D, S):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, D)).astype(np.float32)
    o = O.OracleNVQ.compute(X, S)
    o.encode_all(X, nthreads=4)
    return X, o


def test_nvqvectors_reader_round_trip():
    from jvector_amd import formats as F
    X, o = _rows(1, 37, 50, 3)
    blob = W.write_nvqvectors(o.mean, 3, o.bytes, o.params)
    bl, ver, D, S, stride = F.describe_nvq(blob)
    assert (ver, D, S) == (6, 50, 3) and bl == 4 + 4 + 4 * 50 + 4 + 4 + 4 * 3
    assert stride == 4 + sum(28 + z for z in W.nvq_sizes(50, 3))           # NVQuantization.compressedVectorSize :357-363
    mean, S2, b, p = F.read_nvqvectors(blob)
    assert S2 == 3 and np.array_equal(mean, o.mean) and np.array_equal(b, o.bytes)
    assert np.array_equal(p.view(np.uint32), o.params.view(np.uint32))
    with pytest.raises(ValueError, match="truncated|run past"):
        F.read_nvqvectors(blob[:-3])
    bad = bytearray(blob)
    bad[8 + 4 * 50: 8 + 4 * 50 + 4] = (4).to_bytes(4, "big")              # bitsPerDimension = 4: BitsPerDimension.load throws
    from jvector_amd import UnsupportedError
    with pytest.raises(UnsupportedError, match="Unsupported BitsPerDimension 4"):
        F.describe_nvq(bytes(bad))
    odd = bytearray(blob)
    odd[8 + 4 * 50 + 8: 8 + 4 * 50 + 12] = (16).to_bytes(4, "big")        # sizes that NVQuantization.create would not make
    with pytest.raises(UnsupportedError, match="NVQuantization.create"):
        F.describe_nvq(bytes(odd))


@pytest.mark.parametrize("separated,version", [(False, 6), (True, 6), (False, 4), (True, 5)])
def test_odgi_with_nvq_features(separated, version):
    from jvector_amd import formats as F
    N, D, S, deg = 23, 12, 2, 4
    X, o = _rows(2, N, D, S)
    rng = np.random.default_rng(3)
    l0 = [list(rng.choice(N, int(rng.integers(0, deg + 1)), replace=False)) for _ in range(N)]
    omitted = (5,)
    data = W.write_odgi(version, D, l0, deg, 0, vectors=None, nvq=(o.mean, S, o.bytes, o.params), nvq_separated=separated,
                        omitted=omitted)
    info = F.describe_odgi(data)
    assert info.nvq_S == S and info.nvq_off >= 0
    assert (info.separated_nvq_off >= 0) == separated and (info.nvq_inline_off >= 0) == (not separated)
    g = F.read_odgi(data)
    assert g.features == (("SEPARATED_NVQ",) if separated else ("NVQ_VECTORS",))
    assert np.array_equal(F.read_nvq_mean(g.nvq_block), o.mean)
    want_b, want_p = o.bytes.copy(), o.params.copy()
    want_b[5], want_p[5] = 0, 0                                            # the omitted ordinal: QuantizedVector.createEmpty
    assert np.array_equal(g.nvq_bytes, want_b) and np.array_equal(g.nvq_params.view(np.uint32), want_p.view(np.uint32))
    for i in range(N):
        assert list(g.levels[0][1][i][: len(l0[i])]) == ([] if i in omitted else l0[i])
    if separated:
        # both shapes of a hole in the separated block: featureSize zero bytes (an OMITTED ordinal, AbstractGraphIndexWriter
        # :298-307 — ADVICE r3: the reader used to refuse it) and an empty QuantizedVector (a present ordinal without a vector,
        # SeparatedNVQ.java:86-94); and a zero record whose level-0 record is NOT a placeholder
        for kw in (dict(separated_holes_as_zero_bytes=False), dict(sequential_placeholders=True)):
            gk = F.read_odgi(W.write_odgi(version, D, l0, deg, 0, nvq=(o.mean, S, o.bytes, o.params), nvq_separated=True, omitted=omitted, **kw))
            assert np.array_equal(gk.nvq_bytes, want_b) and np.array_equal(gk.nvq_params.view(np.uint32), want_p.view(np.uint32))
        blob = bytearray(W.write_odgi(version, D, l0, deg, 0, nvq=(o.mean, S, o.bytes, o.params), nvq_separated=True))
        stride = info.nvq_stride
        blob[info.separated_nvq_off + 7 * stride: info.separated_nvq_off + 8 * stride] = bytes(stride)
        g7 = F.read_odgi(bytes(blob))
        w7b, w7p = o.bytes.copy(), o.params.copy()
        w7b[7], w7p[7] = 0, 0
        assert np.array_equal(g7.nvq_bytes, w7b) and np.array_equal(g7.nvq_params.view(np.uint32), w7p.view(np.uint32))
    # sequential-writer placeholders carry unspecified inline bytes: the reader hands back a zeroed row
    if not separated:
        data2 = W.write_odgi(version, D, l0, deg, 0, nvq=(o.mean, S, o.bytes, o.params), omitted=omitted, sequential_placeholders=True,
                             placeholder_fill=0xAB)
        g2 = F.read_odgi(data2)
        assert np.array_equal(g2.nvq_bytes, want_b) and np.array_equal(g2.nvq_params.view(np.uint32), want_p.view(np.uint32))


def test_product_odgi_writer_with_nvq_matches_the_oracle_writer():
    """formats.write_odgi(nvq=...) (the product's writer) emits the bytes the oracle's restatement of the reference's writers does"""
    from jvector_amd import formats as F
    N, D, S, deg = 23, 12, 2, 4
    X, o = _rows(6, N, D, S)
    rng = np.random.default_rng(7)
    l0 = [list(rng.choice(N, int(rng.integers(1, deg + 1)), replace=False)) for _ in range(N)]
    nb = np.full((N, deg), -1, np.int32)
    for i, r in enumerate(l0):
        nb[i, :len(r)] = r
    blk = W.write_nvq_block(o.mean, S)
    for sep in (False, True):
        want = W.write_odgi(6, D, l0, deg, 0, nvq=(o.mean, S, o.bytes, o.params), nvq_separated=sep)
        assert F.write_odgi(D, [(None, nb)], 0, nvq=(blk, o.bytes, o.params), nvq_separated=sep) == want
        g = F.read_odgi(want)
        assert np.array_equal(g.nvq_bytes, o.bytes) and g.nvq_block == blk


# ---- (4) host logic on the mock device -------------------------------------------------------------------------------
pytest_mock = pytest.mark.skipif(platform.machine() != "x86_64", reason="the lane emulator's context switch is x86-64 assembly")


@pytest.fixture(scope="module")
def J():
    import build_mock
    import jvector_amd
    import jvector_amd._lib as L
    lib = C.CDLL(build_mock.build())
    for table in (L.SIGNATURES, L.COMPAT_SIGNATURES, L.FORMAT_SIGNATURES):
        for name, (res, args) in table.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    saved, L._lib = L._lib, lib
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


@pytest_mock
def test_nvq_entry_points_on_the_mock(J, ctx):
    import test_zz_nvq_gpu as T
    T.test_global_mean_bit_exact(ctx, 33, 7)
    T.test_encode_bit_exact(ctx, 33, 5, "offset", True, n=20)
    T.test_encode_bit_exact(ctx, 64, 2, "unit", False, n=20)
    T.test_encode_edge_values(ctx)
    T.test_scores_bit_exact(ctx, 100, 3, 120, 3, 64)
    T.test_scores_after_partial_upload(ctx)
    T.test_float_entry_points_refuse_nvq_rows(ctx)


@pytest_mock
def test_nvq_rerank_through_the_searchers_on_the_mock(J, ctx):
    import test_zz_nvq_gpu as T
    T.test_flat_search_reranks_with_nvq(ctx, N=1500)
    T.test_graph_search_reranks_with_nvq(ctx, "host")
    T.test_graph_search_reranks_with_nvq(ctx, "device")
    T.test_nvq_formats_round_trip_on_device(ctx)
