"""CPU check of the PQ training kernel bodies (jvector_amd/csrc/km_body.h): per-thread bodies run as loops,
the k-means++ seeding on the 64-lane wave emulator, in the launch order of pq_train.cpp.  With the same seeded RNG the
trained / refined codebooks must equal the oracle's sequential restatement of KMeansPlusPlusClusterer +
ProductQuantization.compute / refine BIT FOR BIT (the per-cluster replay keeps the reference's accumulation order)."""
import ctypes as C
import os
import subprocess

import numpy as np
import platform
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.skipif(platform.machine() != "x86_64", reason="the lane emulator's context switch is x86-64 assembly")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "tests", "emu", "km_emu.cpp"), os.path.join(ROOT, "tests", "emu", "hip_emu.h"),
       os.path.join(ROOT, "jvector_amd", "csrc", "km_body.h")]
LIB = os.path.join(ROOT, "build", "emu", "libkm_emu.so")
P = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", SRC[0], "-o", LIB])
    return C.CDLL(LIB)


def data(n, D, seed, n_centers=40, spread=0.3):
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((n_centers, D)).astype(np.float32)
    return (centers[rng.integers(0, n_centers, n)] + spread * rng.standard_normal((n, D))).astype(np.float32)


def layout(D, M, k=256):
    sizes, offs = O.subvector_sizes_offsets(D, M)
    cbo = np.concatenate([[0], np.cumsum(k * np.asarray(sizes[:-1], np.int64))]).astype(np.int64)
    return np.asarray(sizes, np.int32), np.asarray(offs, np.int32), cbo


@pytest.mark.parametrize("D,M,center,seed", [(32, 4, True, 3), (26, 3, False, 9), (64, 8, True, 1)])
def test_train_matches_oracle_bit_for_bit(emu, D, M, center, seed, monkeypatch):
    monkeypatch.setenv("EMU_LANE_ORDER", ["reverse", "random:3", ""][seed % 3])  # k-means++ wave: lane scheduling must not matter
    v = data(3000, D, seed)
    sizes, offs, cbo = layout(D, M)
    want, rounds = O.pq_train(v, M, globally_center=center, seed=seed)
    got = np.full(256 * D, np.nan, np.float32)
    cen = np.zeros(D, np.float32)
    emu.km_emu_train(P(v), C.c_int64(len(v)), D, M, 256, P(cbo), P(sizes), P(offs), int(center), C.c_uint64(seed), 6, 0, None,
                     P(got), P(cen), 0, None)
    assert np.array_equal(got, want.codebooks)
    if center:
        assert np.array_equal(cen, want.centroid)


def test_early_stop_per_subspace_and_empty_clusters(emu):
    """Subspace 0 has 300 distinct sub-vectors (converges after one round: the 1 % rule stops it early), subspace 1 only
    100 (< k: duplicate seeds, 156 clusters run empty every round and are re-seeded from the RNG stream), subspace 2 is
    noise (all six rounds) — still bit-identical."""
    D, M = 24, 3
    rng = np.random.default_rng(5)
    v = np.empty((4000, D), np.float32)
    v[:, :8] = rng.standard_normal((300, 8)).astype(np.float32)[rng.integers(0, 300, 4000)]
    v[:, 8:16] = rng.standard_normal((100, 8)).astype(np.float32)[rng.integers(0, 100, 4000)]
    v[:, 16:] = rng.standard_normal((4000, 8)).astype(np.float32)
    sizes, offs, cbo = layout(D, M)
    want, rounds = O.pq_train(v, M, globally_center=False, seed=2)
    assert rounds[0] < rounds[2] == 6
    assert len(np.unique(want.encode_all(v)[:, 1])) <= 100
    got = np.empty(256 * D, np.float32)
    emu.km_emu_train(P(v), C.c_int64(len(v)), D, M, 256, P(cbo), P(sizes), P(offs), 0, C.c_uint64(2), 6, 0, None, P(got), None, 0, None)
    assert np.array_equal(got, want.codebooks)


@pytest.mark.parametrize("rounds,centroid", [(1, False), (2, True)])
def test_refine_matches_oracle_bit_for_bit(emu, rounds, centroid):
    D, M = 32, 4
    v = data(2500, D, 11)
    sizes, offs, cbo = layout(D, M)
    rng = np.random.default_rng(0)
    cen = (0.1 * rng.standard_normal(D)).astype(np.float32) if centroid else None
    base = v if cen is None else (v - cen).astype(np.float32)
    pick = rng.choice(len(v), 256, replace=False)
    cb = np.concatenate([base[pick, offs[m]: offs[m] + sizes[m]].reshape(-1) for m in range(M)]).astype(np.float32)
    pq = O.OraclePQ(D, M, cb, cen)
    want = pq.refine(data(2000, D, 12), rounds, seed=4)
    got = cb.copy()
    x = data(2000, D, 12)
    emu.km_emu_train(P(x), C.c_int64(len(x)), D, M, 256, P(cbo), P(sizes), P(offs), 0, C.c_uint64(4), rounds, 1, P(cen), P(got), None, 0, None)
    assert np.array_equal(got, want.codebooks)
    assert not np.array_equal(got, cb)


def test_perfect_reconstruction_like_the_reference_test(emu):
    """TestProductQuantization.testPerfectReconstruction (TS/quantization/TestProductQuantization.java:54-80): as many
    distinct 3-d integer-valued vectors as clusters (then each repeated 10x): every vector must decode to itself."""
    rng = np.random.default_rng(42)
    v1 = rng.integers(0, 100000, (256, 3)).astype(np.float32)
    for v in (v1, np.repeat(v1, 10, axis=0)):
        sizes, offs, cbo = layout(3, 2)
        want, _ = O.pq_train(v, 2, globally_center=False, seed=5)
        codes = want.encode_all(v)
        assert np.array_equal(np.stack([want.decode(c) for c in codes]), v)
        got = np.empty(256 * 3, np.float32)
        emu.km_emu_train(P(v), C.c_int64(len(v)), 3, 2, 256, P(cbo), P(sizes), P(offs), 0, C.c_uint64(5), 6, 0, None, P(got), None, 0, None)
        assert np.array_equal(got, want.codebooks)


def _loss(pq, v):
    codes = pq.encode_all(v)
    rec = np.stack([pq.decode(c) for c in codes])
    return float(((v - rec) ** 2).sum())


def test_reference_training_properties_hold_for_the_restatement():
    """The reference's statistical tests of the clusterer, applied to the oracle restatement the device code is checked
    against: one Lloyd round improves on the k-means++ seeds (testIterativeImprovementOnce, TestProductQuantization.java:
    91-104) and refining on fresh data from the same distribution lowers the loss on that data (testRefine :107-131)."""
    rng = np.random.default_rng(1)
    for trial in range(3):
        n = 256 + int(rng.integers(0, 2560))
        D = 2 + int(rng.integers(0, 10))
        centers = rng.standard_normal((20, D)).astype(np.float32)
        v = (centers[rng.integers(0, 20, n)] + 0.3 * rng.standard_normal((n, D))).astype(np.float32)
        seeds, _ = O.pq_train(v, 1, seed=trial, rounds=0)
        once, _ = O.pq_train(v, 1, seed=trial, rounds=1)
        assert _loss(once, v) < _loss(seeds, v)
        half1, half2 = v[: n // 2], v[n // 2:]
        if len(half1) >= 256:
            pq1, _ = O.pq_train(half1, 1, seed=trial)
            assert _loss(pq1.refine(half2, 1, seed=trial), half2) < _loss(pq1, half2)


@pytest.mark.parametrize("D,M,threshold,mode", [(32, 4, 0.2, "train"), (26, 3, 0.5, "train"), (32, 4, 0.3, "refine")])
def test_anisotropic_kmeans_matches_oracle_bit_for_bit(emu, D, M, threshold, mode):
    """cluster(6, 6) / refine with an anisotropic threshold: the anisotropic rounds (per-cluster outer-product sums,
    8x8 Gauss-Jordan inverse, weighted reassignment) replayed per cluster in point order == the oracle's restatement."""
    rng = np.random.default_rng(D)
    v = data(2500, D, D + 3)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    sizes, offs, cbo = layout(D, M)
    pcm = np.array([O.lib().jvo_parallel_cost_multiplier(C.c_float(threshold), int(s)) for s in sizes], np.float32)
    if mode == "train":
        want, _ = O.pq_train(v, M, seed=5, anisotropic_threshold=threshold)
        plain, _ = O.pq_train(v, M, seed=5)
        got = np.empty(256 * D, np.float32)
        emu.km_emu_train(P(v), C.c_int64(len(v)), D, M, 256, P(cbo), P(sizes), P(offs), 0, C.c_uint64(5), 6, 0, None, P(got), None, 6,
                         P(pcm))
        assert not np.array_equal(want.codebooks, plain.codebooks)
    else:
        pick = rng.choice(len(v), 256, replace=False)
        cb = np.concatenate([v[pick, offs[m]: offs[m] + sizes[m]].reshape(-1) for m in range(M)]).astype(np.float32)
        want = O.OraclePQ(D, M, cb).refine(v, 2, seed=5, anisotropic_threshold=threshold)
        got = cb.copy()
        emu.km_emu_train(P(v), C.c_int64(len(v)), D, M, 256, P(cbo), P(sizes), P(offs), 0, C.c_uint64(5), 0, 1, None, P(got), None, 2,
                         P(pcm))
    assert np.isfinite(got).all()
    assert np.array_equal(got, want.codebooks)

