"""MFMA tile form of full-resolution scoring (k_exact_dense.hip, jv_hip_exact_scan_dense) on the GPU, through the C ABI:
bit-equal to its k-ascending fmaf-chain specification (oracle.dense_scan), within 1e-5 of the bit-exact scalar-order scan,
identity-times-asymmetric-matrix probe, sub-ranges, device-resident inputs / outputs.

First run on MI355X in round 2 (green); the CPU lane-emulator twin with the documented semantics of
v_mfma_f32_32x32x2_f32 is tests/test_exact_dense_emulated.py."""
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


def run_dense_cases(J, ctx, shapes=((1, 1, 1), (5, 130, 7), (33, 129, 100), (64, 1000, 96), (100, 3000, 768), (3, 77, 129))):
    """shared with tests/test_mock_device.py (smaller shapes there)"""
    for Q, N, D in shapes:
        rng = np.random.default_rng(Q * 1000 + N)
        v = rng.standard_normal((N, D)).astype(np.float32)
        q = rng.standard_normal((Q, D)).astype(np.float32)
        q[0] = v[N // 2]
        vs = J.VectorSet(ctx, v)
        for vsf in VSF:
            got = np.asarray(vs.scan(q, vsf, dense=True))
            assert np.array_equal(got, O.dense_scan(int(vsf), q, v)), (Q, N, D, vsf)
            exact = np.asarray(vs.scan(q, vsf))
            scale = float(np.linalg.norm(q, axis=1).max() * np.linalg.norm(v, axis=1).max()) if vsf == VSF.DOT_PRODUCT else 1.0
            np.testing.assert_allclose(got, exact, rtol=1e-5, atol=1e-6 * scale)
        if N > 40:
            got = np.asarray(vs.scan(q, VSF.COSINE, first=17, count=N - 30, dense=True))
            assert np.array_equal(got, O.dense_scan(O.COSINE, q, v[17:N - 13]))
        vs.close()
    # identity x asymmetric matrix (a transposed accumulator unpack cannot pass)
    D, N = 70, 150
    B = (np.arange(N)[:, None] * 1000 + np.arange(D)[None, :]).astype(np.float32) / 4096.0
    vs = J.VectorSet(ctx, B)
    got = np.asarray(vs.scan(np.eye(D, dtype=np.float32), VSF.DOT_PRODUCT, dense=True))
    assert np.array_equal(got, ((1.0 + B.T) / 2.0).astype(np.float32))
    # argument checks
    with pytest.raises(ValueError):
        vs.scan(np.eye(D, dtype=np.float32), VSF.DOT_PRODUCT, first=100, count=100, dense=True)
    vs.close()


def test_dense_scan_matches_specification(ctx):
    run_dense_cases(J, ctx)


def test_dense_scan_device_resident(ctx):
    import torch
    rng = np.random.default_rng(8)
    v = rng.standard_normal((5000, 256)).astype(np.float32)
    q = rng.standard_normal((96, 256)).astype(np.float32)
    vt, qt = torch.from_numpy(v).cuda(), torch.from_numpy(q).cuda()
    vs = J.VectorSet(ctx, vt)
    out = vs.scan(qt, VSF.COSINE, dense=True)
    assert out.is_cuda
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), O.dense_scan(O.COSINE, q, v))
    # top-10 by the dense scores, rescored by the bit-exact kernel, is the bit-exact top-10 whenever the 10th / 11th gap
    # exceeds the forms' disagreement (always, on this data): the intended use for ground truth
    exact = vs.scan(qt, VSF.COSINE).cpu().numpy()
    top_d = np.argsort(-out.cpu().numpy(), axis=1, kind="stable")[:, :10]
    top_e = np.argsort(-exact, axis=1, kind="stable")[:, :10]
    assert np.array_equal(np.sort(top_d, axis=1), np.sort(top_e, axis=1))
    vs.close()
