"""Anisotropic PQ encode on the GPU through the C ABI: codes byte-identical to the oracle's restatement
of ProductQuantization.encodeAnisotropic.  First run on MI355X in round 2 (green); the CPU lane-emulator twin is
tests/test_anisotropic_emulated.py."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import jvector_amd as J
from oracle import oracle as O


@pytest.fixture(scope="module")
def ctx():
    c = J.HipContext(0)
    yield c
    c.close()


@pytest.mark.parametrize("D,M,centroid,threshold", [(64, 8, False, 0.2), (50, 7, True, 0.5), (768, 96, False, 0.2)])
def test_anisotropic_encode_matches_oracle(ctx, D, M, centroid, threshold):
    rng = np.random.default_rng(D + M)
    centers = rng.standard_normal((12, D)).astype(np.float32)
    v = (centers[rng.integers(0, 12, 4000)] + 0.6 * rng.standard_normal((4000, D))).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    sizes, offs = O.subvector_sizes_offsets(D, M)
    cen = (0.05 * rng.standard_normal(D)).astype(np.float32) if centroid else None
    base = v if cen is None else (v - cen).astype(np.float32)
    pick = rng.choice(4000, 256, replace=False)
    cb = np.concatenate([base[pick, offs[m]: offs[m] + sizes[m]].reshape(-1) for m in range(M)]).astype(np.float32)
    opq = O.OraclePQ(D, M, cb, cen)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb, cen).set_anisotropic_threshold(threshold)
    assert pq.anisotropic_threshold == np.float32(threshold)
    n = 600
    got = pq.encode_all(v[:n])
    want = np.stack([opq.encode_anisotropic(v[i], threshold) for i in range(n)])
    assert np.array_equal(got, want)
    assert (want != opq.encode_all(v[:n])).any()
    # serialized form carries the threshold (version >= 3) and load restores it
    pq2 = J.ProductQuantization.load(ctx, opq.serialize(6, aniso=threshold))
    assert pq2.anisotropic_threshold == np.float32(threshold)
    assert np.array_equal(pq2.encode_all(v[:50]), want[:50])
    with pytest.raises(ValueError):
        pq.set_anisotropic_threshold(1.0)
    pq.set_anisotropic_threshold(-1.0)
    assert np.array_equal(pq.encode_all(v[:50]), opq.encode_all(v[:50]))
