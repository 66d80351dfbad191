"""PQ training on the GPU through the C ABI: ProductQuantization.compute / refine / write, bit-identical to
the oracle's sequential restatement for the same seed.  First run on MI355X in round 2 (green); the CPU twin of the same
kernel bodies is tests/test_pq_train_emulated.py."""
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


def data(n, D, seed):
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((40, D)).astype(np.float32)
    return (centers[rng.integers(0, 40, n)] + 0.3 * rng.standard_normal((n, D))).astype(np.float32)


@pytest.mark.parametrize("D,M,center", [(32, 4, True), (26, 3, False), (128, 16, True)])
def test_train_refine_write(ctx, D, M, center):
    v = data(5000, D, D)
    want, _ = O.pq_train(v, M, globally_center=center, seed=7)
    pq = J.ProductQuantization.compute(ctx, v, M, globally_center=center, seed=7)
    blob = pq.write(6)
    got, ver, aniso, used = O.OraclePQ.parse(blob)
    assert used == len(blob) and ver == 6 and aniso == -1.0
    assert np.array_equal(got.codebooks, want.codebooks)
    assert (got.centroid is None) == (want.centroid is None)
    if center:
        assert np.array_equal(got.centroid, want.centroid)
    assert blob == want.serialize(6)                       # byte-identical wire form
    assert np.array_equal(pq.encode_all(v[:200]), want.encode_all(v[:200]))
    x = data(3000, D, D + 1)
    want2 = want.refine(x, 2, seed=3)
    got2, _, _, _ = O.OraclePQ.parse(pq.refine(x, 2, seed=3).write(0 if not center else 2))
    assert np.array_equal(got2.codebooks, want2.codebooks)
    with pytest.raises(ValueError):
        J.ProductQuantization.compute(ctx, v[:100], M)   # fewer points than clusters


@pytest.mark.parametrize("D,M,k,center", [(32, 4, 16, True), (26, 3, 50, False), (64, 8, 255, False)])
def test_train_with_fewer_than_256_clusters(ctx, D, M, k, center):
    """ProductQuantization.compute / refine with clusterCount < 256 (the reference's own tests train 16 and 50 clusters): the
    training kernels work on a k-row layout, the finished quantizer is padded on the device side; codebooks, wire form and codes ==
    the oracle trained with the same count"""
    v = data(4000, D, D + k)
    want, _ = O.pq_train(v, M, k=k, globally_center=center, seed=5)
    pq = J.ProductQuantization.compute(ctx, v, M, cluster_count=k, globally_center=center, seed=5)
    assert pq.get_cluster_count() == k
    blob = pq.write(6)
    assert blob == want.serialize(6)
    codes = pq.encode_all(v[:300])
    assert np.array_equal(codes, want.encode_all(v[:300])) and int(codes.max()) < k
    x = data(2500, D, D + k + 1)
    want2 = want.refine(x, 2, seed=3)
    pq2 = pq.refine(x, 2, seed=3)
    assert pq2.get_cluster_count() == k and pq2.write(6) == want2.serialize(6)
    with pytest.raises(ValueError):
        J.ProductQuantization.compute(ctx, v[: k - 1], M, cluster_count=k)   # fewer points than clusters (:119-121)
    with pytest.raises(J.UnsupportedError):
        J.ProductQuantization.compute(ctx, v, M, cluster_count=k, anisotropic_threshold=0.2)


def test_anisotropic_training(ctx):
    """compute / refine with an anisotropic threshold: unweighted + anisotropic k-means rounds == the oracle, bit for bit"""
    v = data(4000, 32, 77)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    want, _ = O.pq_train(v, 4, seed=9, anisotropic_threshold=0.2)
    pq = J.ProductQuantization.compute(ctx, v, 4, seed=9, anisotropic_threshold=0.2)
    got, ver, aniso, _ = O.OraclePQ.parse(pq.write(6))
    assert aniso == np.float32(0.2) and np.array_equal(got.codebooks, want.codebooks)
    x = data(3000, 32, 78)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    want2 = want.refine(x, 2, seed=4, anisotropic_threshold=0.2)
    got2, _, _, _ = O.OraclePQ.parse(pq.refine(x, 2, seed=4).write(6))
    assert np.array_equal(got2.codebooks, want2.codebooks)
    codes = pq.encode_all(v[:100])                      # the trained PQ encodes anisotropically
    assert np.array_equal(codes, np.stack([want.encode_anisotropic(v[i], 0.2) for i in range(100)]))

