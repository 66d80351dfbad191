"""benchlib.py — synthetic INPUT PREPARATION for bench.py (data, codebooks, ground truth, a kNN+prune graph).

Nothing here is the measured hot path and nothing here is product code: torch is used as plumbing to create
seeded synthetic inputs on the device (BASELINE.json: no datasets/checkpoints are available offline).
"""
from __future__ import annotations

import math

import numpy as np
import torch


# ------------------------------------------------------------------------------------------------------
# synthetic "ada-002-like" data (seeded; generated on the device, never materialised on the host)
# Embedding-like structure: a low intrinsic dimension (latent L-dim mixture of clusters) linearly embedded in
# D dims plus small isotropic noise, then L2-normalised.  Pure input preparation (torch is plumbing here).
# ------------------------------------------------------------------------------------------------------
class Mixture:
    def __init__(self, D, seed, device, n_clusters=1000, latent=32, spread=0.7, noise=0.08):
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.D, self.L, self.device = D, latent, device
        self.spread, self.noise = spread, noise
        self.centers = torch.randn(n_clusters, latent, generator=g).to(device)
        self.centers /= self.centers.norm(dim=1, keepdim=True)
        proj = torch.randn(latent, D, generator=g)
        q, _ = torch.linalg.qr(proj.t())          # D x L with orthonormal columns
        self.proj = q.t().contiguous().to(device)  # L x D

    def sample(self, n, seed, chunk=1_000_000, out=None):
        g = torch.Generator(device=self.device).manual_seed(seed)
        if out is None:
            out = torch.empty(n, self.D, dtype=torch.float32, device=self.device)
        for s in range(0, n, chunk):
            e = min(n, s + chunk)
            cid = torch.randint(0, self.centers.shape[0], (e - s,), generator=g, device=self.device)
            z = self.centers[cid] + self.spread * torch.randn(e - s, self.L, generator=g, device=self.device) / math.sqrt(self.L)
            x = z @ self.proj
            x += self.noise * torch.randn(e - s, self.D, generator=g, device=self.device) / math.sqrt(self.D)
            x /= x.norm(dim=1, keepdim=True)
            out[s:e] = x
        return out


class LiteralMixture:
    """SURVEY §8d's C3 generator taken literally: 1 000 Gaussian clusters, sigma = 0.1 PER COORDINATE around unit-norm centres in all D
    dimensions, then L2-normalised.  In 768 dimensions the noise vector has norm 0.1 * sqrt(768) = 2.8 against a centre of norm 1: the
    result is an isotropic 768-dimensional cloud in which nearest neighbours are barely closer than random pairs — bench.py measures
    it as `literal_c3` so that the record shows why the headline uses the low-intrinsic-dimension Mixture instead."""

    def __init__(self, D, seed, device, n_clusters=1000, sigma=0.1):
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.D, self.device, self.sigma = D, device, sigma
        self.centers = torch.randn(n_clusters, D, generator=g).to(device)
        self.centers /= self.centers.norm(dim=1, keepdim=True)

    def sample(self, n, seed, chunk=1_000_000, out=None):
        g = torch.Generator(device=self.device).manual_seed(seed)
        if out is None:
            out = torch.empty(n, self.D, dtype=torch.float32, device=self.device)
        for s in range(0, n, chunk):
            e = min(n, s + chunk)
            cid = torch.randint(0, self.centers.shape[0], (e - s,), generator=g, device=self.device)
            x = self.centers[cid] + self.sigma * torch.randn(e - s, self.D, generator=g, device=self.device)
            x /= x.norm(dim=1, keepdim=True)
            out[s:e] = x
        return out


def train_codebooks(base, M, seed, iters=6, sample=128_000, k=256):
    """Fixed synthetic codebooks: Lloyd iterations per subspace on a sample (the reference trains on <= 128k
    vectors for 6 iterations, ProductQuantization.java:63-64).  Input preparation only — PQ training is a
    'next' row, not part of the measured path.  Returns float32 [M*k*size] centroid-major."""
    n, D = base.shape
    size = D // M
    g = torch.Generator(device=base.device).manual_seed(seed)
    idx = torch.randperm(n, generator=g, device=base.device)[: min(sample, n)]
    X = base[idx].reshape(-1, M, size).permute(1, 0, 2).contiguous()      # M x S x size
    S = X.shape[1]
    cent = X[:, torch.randperm(S, generator=g, device=base.device)[:k], :].clone()  # M x k x size
    for _ in range(iters):
        assign = torch.empty(M, S, dtype=torch.long, device=base.device)
        for s in range(0, S, 16384):
            xs = X[:, s:s + 16384]
            d = (xs * xs).sum(-1, keepdim=True) - 2 * torch.bmm(xs, cent.transpose(1, 2)) + (cent * cent).sum(-1).unsqueeze(1)
            assign[:, s:s + 16384] = d.argmin(-1)
        sums = torch.zeros_like(cent)
        sums.scatter_add_(1, assign.unsqueeze(-1).expand(-1, -1, size), X)
        cnt = torch.zeros(M, k, device=base.device).scatter_add_(1, assign, torch.ones(M, S, device=base.device))
        cent = torch.where(cnt.unsqueeze(-1) > 0, sums / cnt.clamp(min=1).unsqueeze(-1), cent)
    return cent.reshape(-1).contiguous()


def sift_like(n, D, seed, device, chunk=1_000_000):
    """SURVEY §8d C1/C2 generator: non-negative integers-as-float, clip(round(|N(0,1)| * 40), 0, 218) (SIFT descriptors are
    byte-valued, mostly small)."""
    g = torch.Generator(device=device).manual_seed(seed)
    out = torch.empty(n, D, dtype=torch.float32, device=device)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        out[s:e] = (torch.randn(e - s, D, generator=g, device=device).abs() * 40.0).round().clamp_(0, 218)
    return out


def recall_per_query(found, truth):
    """per-query |top-k ∩ gt-k| / k (AccuracyMetrics.recallFromSearchResults, EX/util/AccuracyMetrics.java:38-90, before the
    average): the mean is the reported recall, std / sqrt(n) its standard error."""
    k = truth.shape[1]
    out = np.empty(truth.shape[0], np.float64)
    for i, (f, t) in enumerate(zip(found, truth)):
        out[i] = len(set(int(x) for x in f if x >= 0) & set(int(x) for x in t)) / float(k)
    return out


def recall_at_k(found, truth):
    """AccuracyMetrics.recallFromSearchResults (EX/util/AccuracyMetrics.java:38-90): |top-k ∩ gt-k| / k averaged."""
    hits = 0
    for f, t in zip(found, truth):
        hits += len(set(int(x) for x in f if x >= 0) & set(int(x) for x in t))
    return hits / float(truth.shape[0] * truth.shape[1])


def ground_truth(J, ctx, vs, queries, vsf, k, chunk=1_000_000, dense=False, q_group=2048):
    """Exact top-k by brute force with the engine's exact-scan kernels + NodeQueue-order top-k, query groups of <= q_group
    (the score buffer is q_group x chunk floats).
    dense=True: candidates from the MFMA tile form of the scan (4k per query, fused-chain scores within 1e-5 of the exact
    ones), then the bit-exact kernel rescores just those and picks the top k — the same ids as the bit-exact scan whenever
    the k-th / 4k-th score gap exceeds the two forms' disagreement (~1e-7), at a fraction of the VALU work."""
    Qall, N = queries.shape[0], vs.count
    kc = min(4 * k, N) if dense else k
    buf = torch.empty(min(q_group, Qall), min(chunk, N), dtype=torch.float32, device=queries.device)
    result = []
    for q0 in range(0, Qall, q_group):
        qs = queries[q0:q0 + q_group].contiguous()
        Q = qs.shape[0]
        part_ids, part_sc = [], []
        for s in range(0, N, chunk):
            c = min(chunk, N - s)
            out = buf[:Q, :c] if (c == buf.shape[1] and Q == buf.shape[0]) else torch.empty(Q, c, dtype=torch.float32, device=queries.device)
            vs.scan(qs, vsf, first=s, count=c, out=out, dense=dense)
            ids, sc = J.topk(ctx, out, min(kc, c), id_base=s)
            part_ids.append(ids)
            part_sc.append(sc)
        ids, sc = J.topk(ctx, torch.cat(part_sc, 1).contiguous(), kc, ids=torch.cat(part_ids, 1).contiguous())
        if dense:
            exact = vs.scores(qs, vsf, ids.contiguous())
            ids, sc = J.topk(ctx, exact, k, ids=ids.contiguous())
        ctx.sync()
        result.append(ids.clone())
    return torch.cat(result)


def fused_blocks_from(codes, nbrs):
    """FusedPQ.writeInline layout on the device: neighbour i's code at bytes [i*M, (i+1)*M), zero padded."""
    N, deg = nbrs.shape
    M = codes.shape[1]
    out = torch.zeros(N, deg * M, dtype=torch.uint8, device=codes.device)
    for s in range(0, N, 1_000_000):
        nb = nbrs[s:s + 1_000_000].long()
        blk = codes[nb.clamp(min=0).reshape(-1)].reshape(nb.shape[0], deg, M)
        blk[nb < 0] = 0
        out[s:s + 1_000_000] = blk.reshape(nb.shape[0], deg * M)
    return out
