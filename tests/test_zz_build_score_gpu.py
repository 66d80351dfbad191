"""Build-time scoring on the GPU through the C ABI: pair table, code-vs-code diversity scores, decode and
table-free query-vs-code scores, bit-exact against the oracle.

First run on MI355X in round 2 (green); the CPU twin of the same kernel bodies is tests/test_build_score_emulated.py."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import jvector_amd as J
from jvector_amd import VectorSimilarityFunction as VSF
from oracle import oracle as O


@pytest.fixture(scope="module")
def ctx():
    c = J.HipContext(0)
    yield c
    c.close()


@pytest.mark.parametrize("D,M,centroid", [(64, 8, False), (50, 7, True), (768, 96, False)])
def test_build_score_provider_matches_oracle(ctx, D, M, centroid):
    rng = np.random.default_rng(D + M)
    sizes, offs = O.subvector_sizes_offsets(D, M)
    cb = np.concatenate([rng.standard_normal(256 * s).astype(np.float32) for s in sizes])
    cen = rng.standard_normal(D).astype(np.float32) if centroid else None
    opq = O.OraclePQ(D, M, cb, cen)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb, cen)
    n = 5000
    codes = rng.integers(0, 256, (n, M), dtype=np.uint8)
    cv = J.PQVectors(ctx, pq, codes)
    node1 = rng.integers(0, n, 64).astype(np.int32)
    node2 = rng.integers(0, n, (64, 32)).astype(np.int32)
    node2[3, 2], node1[11] = -1, -1
    q = rng.standard_normal((8, D)).astype(np.float32)
    ords = rng.integers(0, n, (8, 24)).astype(np.int32)
    for vsf in VSF:
        bsp = J.PQBuildScoreProvider(ctx, cv, vsf)
        tri = opq.codebook_partial_sums(int(vsf))
        assert np.array_equal(bsp.codebook_partial_sums(), tri)
        got = bsp.diversity_scores(node1, node2)
        for p in range(0, 64, 3):
            for b in range(0, 32, 5):
                bad = node1[p] < 0 or node2[p, b] < 0
                want = -np.inf if bad else np.float32(opq.diversity_score(tri, int(vsf), codes[node1[p]], codes[node2[p, b]]))
                assert got[p, b] == want, (vsf, p, b)
        dec = bsp.decode(node2[0])
        assert np.array_equal(dec, np.stack([opq.decode(codes[o]) for o in node2[0]]))
        sc = cv.direct_scores(q, vsf, ords)
        for i in range(8):
            for b in range(0, 24, 5):
                assert sc[i, b] == np.float32(opq.direct_score(q[i], int(vsf), codes[ords[i, b]])), (vsf, i, b)
        bsp.close()


def test_fused_build_on_device(ctx):
    """jv_hip_fused_build (FusedPQ.writeInline as a device gather) == blocks assembled on the host"""
    rng = np.random.default_rng(3)
    D, M, n, deg = 768, 96, 4000, 32
    cb = rng.standard_normal(256 * D).astype(np.float32)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb)
    codes = rng.integers(0, 256, (n, M), dtype=np.uint8)
    cv = J.PQVectors(ctx, pq, codes)
    nb = np.full((n, deg), -1, np.int32)
    for i in range(n):
        d = int(rng.integers(0, deg + 1))
        nb[i, :d] = rng.integers(0, n, d)
    blocks, nbrs = J.FusedPQ.build(ctx, cv, nb).get()
    want = np.where((nb >= 0)[:, :, None], codes[np.clip(nb, 0, n - 1)], 0).astype(np.uint8).reshape(n, deg * M)
    assert np.array_equal(blocks, want) and np.array_equal(nbrs, nb)

