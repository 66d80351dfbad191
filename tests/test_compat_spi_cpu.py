"""The reference's per-pair native SPI (include/jvector_simd_compat.h, host code in csrc/compat_host.cpp) on the CPU:
every symbol against the oracle's restatement of DefaultVectorUtilSupport, bit for bit, on the reference's own KAT
generator and lengths (NC/tests/test_helpers.cpp:49-87), plus the NVQ symbols against their defining relations.
No GPU is involved: these functions are host code by design (one pair per call cannot amortise a kernel launch)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O

LENGTHS = [1, 2, 3, 4, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 1021]


@pytest.fixture(scope="module")
def lib():
    import jvector_amd
    return jvector_amd.load()


def fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_ubyte))


@pytest.mark.parametrize("n", LENGTHS)
def test_dot_l2_cosine(lib, n):
    a, b = O.make_vec(n + 5, 1.0), O.make_vec(n + 5, 2.0)
    # full-vector forms (offsets 0): 8-block association order
    assert lib.dot_product_f32(fp(a), 0, fp(b), 0, n) == np.float32(O.dot(a[:n], b[:n]))
    assert lib.euclidean_f32(fp(a), 0, fp(b), 0, n) == np.float32(O.l2(a[:n], b[:n]))
    assert lib.cosine_f32(fp(a), 0, fp(b), 0, n) == np.float32(O.cosine(a[:n], b[:n]))
    # offset forms: sequential order
    assert lib.dot_product_f32(fp(a), 3, fp(b), 5, n) == np.float32(O.dot_off(a, 3, b, 5, n))
    assert lib.euclidean_f32(fp(a), 3, fp(b), 5, n) == np.float32(O.l2_off(a, 3, b, 5, n))
    assert lib.cosine_f32(fp(a), 3, fp(b), 5, n) == np.float32(O.cosine_off(a, 3, b, 5, n))


def test_elementwise_and_reductions(lib):
    rng = np.random.default_rng(0)
    a, b = rng.standard_normal(37).astype(np.float32), rng.standard_normal(37).astype(np.float32)
    x = a.copy(); lib.add_in_place_f32(fp(x), fp(b), 37); assert np.array_equal(x, a + b)
    x = a.copy(); lib.sub_in_place_f32(fp(x), fp(b), 37); assert np.array_equal(x, a - b)
    x = a.copy(); lib.add_scalar_in_place_f32(fp(x), C.c_float(0.5), 37); assert np.array_equal(x, a + np.float32(0.5))
    x = a.copy(); lib.sub_scalar_in_place_f32(fp(x), C.c_float(0.5), 37); assert np.array_equal(x, a - np.float32(0.5))
    x = a.copy(); lib.min_in_place_f32(fp(x), fp(b), 37); assert np.array_equal(x, np.minimum(a, b))
    assert lib.max_f32(fp(a), 37) == a.max()


@pytest.mark.parametrize("D,M", [(64, 8), (50, 7), (768, 96)])
def test_pq_symbols(lib, D, M):
    rng = np.random.default_rng(D)
    sizes, offs = O.subvector_sizes_offsets(D, M)
    cb = np.concatenate([rng.standard_normal(256 * s).astype(np.float32) for s in sizes])
    pq = O.OraclePQ(D, M, cb)
    q = rng.standard_normal(D).astype(np.float32)
    codes = rng.integers(0, 256, (6, M), dtype=np.uint8)
    for vsf, fn in ((O.DOT_PRODUCT, lib.calculate_partial_sums_dot_f32), (O.EUCLIDEAN, lib.calculate_partial_sums_euclidean_f32)):
        want, _, _ = pq.decoder(q, vsf)
        got = np.empty(M * 256, np.float32)
        off = 0
        for m in range(M):
            fn(fp(cb[off:]), m, sizes[m], 256, fp(q), int(offs[m]), fp(got))
            off += 256 * sizes[m]
        assert np.array_equal(got, want)
        for c in codes:  # assembleAndSum over the table == the decoder's raw sum
            raw = lib.assemble_and_sum_f32(fp(got), 256, u8(c), 0, M)
            assert O.score_from_raw(vsf, raw) == np.float32(pq.adc_scores(q, vsf, c[None, :])[0])
    lut, amag, bmag = pq.decoder(q, O.COSINE)
    self_mag = np.empty(M * 256, np.float32)
    off = 0
    for m in range(M):
        lib.calculate_partial_sums_self_magnitude_f32(fp(cb[off:]), m, sizes[m], 256, fp(self_mag))
        off += 256 * sizes[m]
    assert np.array_equal(self_mag, amag)
    for c in codes:
        cos = lib.pq_decoded_cosine_similarity_f32(u8(c), 0, M, 256, fp(lut), fp(amag), C.c_float(bmag))
        assert O.score_from_raw(O.COSINE, cos) == np.float32(pq.adc_scores(q, O.COSINE, c[None, :])[0])
    for vsf in (O.DOT_PRODUCT, O.EUCLIDEAN):
        tri = pq.codebook_partial_sums(vsf)
        for i in range(5):
            got = lib.assemble_and_sum_pq_f32(fp(tri), M, u8(codes[i]), 0, u8(codes[i + 1]), 0, 256)
            assert got == np.float32(O.lib().jvo_assemble_and_sum_pq(fp(tri), M, u8(codes[i]), 0, u8(codes[i + 1]), 0, 256))


def test_nvq_symbols_are_consistent(lib):
    """No oracle for NVQ (outside the GPU scope): check the defining relations of DefaultVectorUtilSupport.java:376-548 —
    the distances computed against the quantized bytes equal the float distances to the DEquantized vector, and the loss
    is the squared reconstruction error."""
    rng = np.random.default_rng(4)
    n = 96
    v = rng.standard_normal(n).astype(np.float32)
    q = rng.standard_normal(n).astype(np.float32)
    alpha, x0 = np.float32(1.3), np.float32(0.1)
    lo, hi = np.float32(v.min()), np.float32(v.max())
    qz = np.empty(n, np.uint8)
    lib.nvq_quantize_8bit(fp(v), n, C.c_float(alpha), C.c_float(x0), C.c_float(lo), C.c_float(hi), u8(qz))
    assert qz.min() == 0 and qz.max() == 255            # the extremes map to the ends of the code range
    loss = lib.nvq_loss(fp(v), n, C.c_float(alpha), C.c_float(x0), C.c_float(lo), C.c_float(hi), 8)
    uloss = lib.nvq_uniform_loss(fp(v), n, C.c_float(lo), C.c_float(hi), 8)
    assert 0 <= loss < 1e-2 * n and 0 <= uloss < 1e-2 * n
    l2 = lib.nvq_square_l2_distance_8bit(fp(v), u8(qz), n, C.c_float(alpha), C.c_float(x0), C.c_float(lo), C.c_float(hi))
    assert abs(l2 - loss) <= 1e-4 * max(1.0, loss)      # distance of v to its own quantization == the loss
    dq = lib.nvq_dot_product_8bit(fp(q), u8(qz), n, C.c_float(alpha), C.c_float(x0), C.c_float(lo), C.c_float(hi))
    assert abs(dq - float(q @ v)) < 0.05 * np.linalg.norm(q) * np.linalg.norm(v)
    cen = np.zeros(n, np.float32)
    packed = lib.nvq_cosine_8bit_packed(fp(q), u8(qz), n, C.c_float(alpha), C.c_float(x0), C.c_float(lo), C.c_float(hi), fp(cen))
    packed &= (1 << 64) - 1
    sum_, bmag = np.array([packed & 0xFFFFFFFF, packed >> 32], np.uint32).view(np.float32)
    # two floats packed lo/hi (NativeVectorUtilSupport.java:289-297): sum = <q, v~>, bMagnitude = <v~, v~> with zero centroid
    assert abs(sum_ - dq) <= 1e-4 * max(1.0, abs(dq))
    assert abs(bmag - float(v @ v)) < 0.05 * float(v @ v)
    assert lib.jvector_simd_get_active_isa() == b"gfx950-host"
