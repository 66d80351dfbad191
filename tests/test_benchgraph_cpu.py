"""CPU checks of bench.py's synthetic input preparation (benchlib / benchgraph): structural invariants of the
hierarchical graph and that the oracle's GraphSearcher restatement reaches a sane recall on it.  Not product code —
this guards the bench's inputs."""
import os

import numpy as np
import torch

from benchgraph import build_hier_graph
from benchlib import Mixture, recall_at_k, train_codebooks
from oracle import oracle as O


def test_synthetic_graph_structure_and_recall():
    torch.manual_seed(0)
    N, D, M = 12000, 64, 16
    mix = Mixture(D, seed=5, device="cpu", n_clusters=12, latent=16)
    base = mix.sample(N, seed=5)
    q = mix.sample(30, seed=6)
    assert torch.allclose(base.norm(dim=1), torch.ones(N), atol=1e-5)
    levels, entry, entry_level, nb0 = build_hier_graph(base, max_degree=32)
    assert levels[0][0] is None and levels[0][1].shape == (N, 32) and entry_level == len(levels) - 1
    prev = None
    for ids, nb in levels:
        n = nb.shape[0]
        assert nb.dtype == np.int32 and nb.min() >= -1 and nb.max() < N
        valid = nb >= 0
        # packed rows: no hole before a valid entry; no self loops; no duplicates
        assert np.all(valid[:, 1:] <= valid[:, :-1])
        rows = np.arange(n) if ids is None else ids
        assert not np.any(nb == rows[:, None])
        for r in range(0, n, max(1, n // 50)):
            v = nb[r][nb[r] >= 0]
            assert len(set(v.tolist())) == len(v)
        if ids is not None:
            assert np.all(np.diff(ids) > 0)                      # ascending ids
            assert np.isin(nb[valid], ids).all()                 # edges stay inside the level
            if prev is not None:
                assert np.isin(ids, prev).all()                  # nested levels
            prev = ids
    assert entry in set(levels[-1][0].tolist())
    cb = train_codebooks(base, M, seed=4)
    opq = O.OraclePQ(D, M, cb.numpy())
    codes = opq.encode_all(base.numpy(), nthreads=4)
    gt = (q @ base.t()).topk(10, dim=1).indices.numpy()
    og = O.OracleGraph(N, levels, entry, entry_level)
    ids, _, stats = og.search(opq, codes, base.numpy(), q.numpy(), O.COSINE, 10, 200, fused=True)
    assert recall_at_k(ids, gt) >= 0.9
    assert stats[:, 0].mean() < N / 2          # a graph search, not a scan


def test_bench_module_imports_and_parses_its_flags():
    """bench.py cannot run without a GPU, but a syntax / import error or a broken flag table must not wait for the GPU box"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "--traversal" in out.stdout and "--gpus" in out.stdout
    sys.path.insert(0, root)
    import bench
    assert bench.effective_cpus() >= 1
