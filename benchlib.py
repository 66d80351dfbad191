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


def recall_at_k(found, truth):
    """AccuracyMetrics.recallFromSearchResults (EX/util/AccuracyMetrics.java:38-90): |top-k ∩ gt-k| / k averaged."""
    hits = 0
    for f, t in zip(found, truth):
        hits += len(set(int(x) for x in f if x >= 0) & set(int(x) for x in t))
    return hits / float(truth.shape[0] * truth.shape[1])


def ground_truth(J, ctx, vs, queries, vsf, k, chunk=1_000_000):
    """Exact top-k by brute force with the engine's bit-exact exact-scan kernel + NodeQueue-order top-k."""
    Q, N = queries.shape[0], vs.count
    part_ids, part_sc = [], []
    buf = torch.empty(Q, min(chunk, N), dtype=torch.float32, device=queries.device)
    for s in range(0, N, chunk):
        c = min(chunk, N - s)
        out = buf[:, :c] if c == buf.shape[1] else torch.empty(Q, c, dtype=torch.float32, device=queries.device)
        vs.scan(queries, vsf, first=s, count=c, out=out)
        ids, sc = J.topk(ctx, out, k, id_base=s)
        part_ids.append(ids)
        part_sc.append(sc)
    ids, sc = J.topk(ctx, torch.cat(part_sc, 1).contiguous(), k, ids=torch.cat(part_ids, 1).contiguous())
    ctx.sync()
    return ids




# ------------------------------------------------------------------------------------------------------
# Synthetic graph index (INPUT PREPARATION — not the reference's GraphIndexBuilder, which is host-side and out of
# scope; SURVEY §8f ranks GPU-assisted construction as a later row).  Produces a Vamana-shaped structure the
# searcher can traverse: layer 0 = every node with <= max_degree diverse neighbours, layer 1 = one medoid per
# coarse cluster, entry = the medoid nearest the global mean.
#   1. coarse k-means (C ~ N/2500 clusters, torch matmul)
#   2. per cluster: exact similarities against the pool {own cluster + `n_probe` nearest clusters}, top-`n_cand`
#   3. robust prune (VamanaDiversityProvider.retainDiverse's rule, B/graph/diversity/VamanaDiversityProvider.java:
#      45-96: keep i iff for every kept j: sim(i,j) <= score(i) * alpha, alpha ramp 1.0 -> 1.2), scores in the
#      similarity domain (1 + cos) / 2
#   4. reverse edges fill the remaining slots (a light-weight stand-in for backlink + prune)
# ------------------------------------------------------------------------------------------------------
def _kmeans(x, C, seed, iters=8, sample_per=128):
    g = torch.Generator(device=x.device).manual_seed(seed)
    n = x.shape[0]
    idx = torch.randperm(n, generator=g, device=x.device)[: min(n, C * sample_per)]
    xs = x[idx]
    cent = xs[torch.randperm(xs.shape[0], generator=g, device=x.device)[:C]].clone()
    for _ in range(iters):
        a = torch.cat([(xs[s:s + 262144] @ cent.t()).argmax(1) for s in range(0, xs.shape[0], 262144)])
        sums = torch.zeros_like(cent).index_add_(0, a, xs)
        cnt = torch.zeros(C, device=x.device).index_add_(0, a, torch.ones_like(a, dtype=torch.float32))
        cent = torch.where(cnt[:, None] > 0, sums / cnt.clamp(min=1)[:, None], cent)
        cent = cent / cent.norm(dim=1, keepdim=True).clamp(min=1e-12)
    return cent


def _robust_prune(score, pair, keep_n, alpha_max=1.2):
    """score [B, K] candidate similarities to the node (descending); pair [B, K, K] candidate-candidate
    similarities.  Returns a bool mask [B, K] of kept candidates (<= keep_n per row)."""
    B, K = score.shape
    kept = torch.zeros(B, K, dtype=torch.bool, device=score.device)
    n_kept = torch.zeros(B, dtype=torch.long, device=score.device)
    for alpha in (1.0, alpha_max):
        for i in range(K):
            # candidate i is occluded if some kept j has pair[i, j] > score[i] * alpha
            occl = ((pair[:, i, :] > (score[:, i] * alpha)[:, None]) & kept).any(1)
            take = (~occl) & (~kept[:, i]) & (n_kept < keep_n) & torch.isfinite(score[:, i])
            kept[:, i] |= take
            n_kept += take.long()
    return kept


def build_graph(base, max_degree=32, seed=11, n_cand=64, n_probe=2, fwd_degree=24, top_degree=16):
    dev = base.device
    N, D = base.shape
    C = max(8, min(8192, N // 2500))
    cent = _kmeans(base, C, seed)
    assign = torch.empty(N, dtype=torch.long, device=dev)
    for s in range(0, N, 1_000_000):
        assign[s:s + 1_000_000] = (base[s:s + 1_000_000] @ cent.t()).argmax(1)
    order = torch.argsort(assign, stable=True)
    counts = torch.bincount(assign, minlength=C)
    offs = torch.zeros(C + 1, dtype=torch.long, device=dev)
    offs[1:] = torch.cumsum(counts, 0)
    offs_h = offs.cpu().tolist()
    csim = cent @ cent.t()
    csim.fill_diagonal_(-2.0)
    near = csim.topk(min(n_probe, C - 1), dim=1).indices.cpu().tolist()

    nbrs = torch.full((N, max_degree), -1, dtype=torch.int32, device=dev)
    medoids = torch.empty(C, dtype=torch.long, device=dev)
    for c in range(C):
        lo, hi = offs_h[c], offs_h[c + 1]
        if hi == lo:
            medoids[c] = -1
            continue
        own = order[lo:hi]
        pool = torch.cat([own] + [order[offs_h[p]:offs_h[p + 1]] for p in near[c]])
        xp = base[pool]
        n_c, n_p = own.shape[0], pool.shape[0]
        medoids[c] = own[(xp[:n_c] @ cent[c]).argmax()]
        k = min(n_cand, n_p - 1)
        for r0 in range(0, n_c, 4096):
            r1 = min(n_c, r0 + 4096)
            s = xp[r0:r1] @ xp.t()                                   # [b, n_p] cosine (unit vectors)
            s[torch.arange(r1 - r0, device=dev), torch.arange(r0, r1, device=dev)] = -2.0   # no self edge
            sc, ci = s.topk(k, dim=1)                                # descending
            cv = xp[ci]                                              # [b, k, D]
            pair = torch.bmm(cv, cv.transpose(1, 2))                 # [b, k, k]
            kept = _robust_prune((1 + sc) / 2, (1 + pair) / 2, fwd_degree)
            # compact kept candidates to the front, in score order
            rank = torch.cumsum(kept.long(), 1) - 1
            sel = torch.full((r1 - r0, max_degree), -1, dtype=torch.int32, device=dev)
            rows = torch.arange(r1 - r0, device=dev)[:, None].expand_as(ci)
            sel[rows[kept], rank[kept]] = pool[ci[kept]].int()
            nbrs[own[r0:r1]] = sel
    # reverse edges into the free slots (closest-first is not tracked; deterministic by source id order)
    deg = (nbrs >= 0).sum(1)
    src = torch.arange(N, device=dev)[:, None].expand(N, max_degree)[nbrs >= 0]
    dst = nbrs[nbrs >= 0].long()
    o = torch.argsort(dst, stable=True)
    src, dst = src[o], dst[o]
    first = torch.searchsorted(dst, torch.arange(N, device=dev))
    pos_in_dst = torch.arange(dst.shape[0], device=dev) - first[dst]
    slot = deg[dst] + pos_in_dst
    ok = slot < max_degree
    # drop reverse edges that already exist as forward edges
    exists = (nbrs[dst[ok]] == src[ok].int()[:, None]).any(1)
    d2, s2, sl2 = dst[ok][~exists], src[ok][~exists], slot[ok][~exists]
    nbrs[d2, sl2] = s2.int()
    # re-pack rows (holes left by dropped duplicates)
    valid = nbrs >= 0
    rank = torch.cumsum(valid.long(), 1) - 1
    packed = torch.full_like(nbrs, -1)
    rows = torch.arange(N, device=dev)[:, None].expand_as(nbrs)
    packed[rows[valid], rank[valid]] = nbrs[valid]

    # layer 1: the medoids, kNN + prune among themselves
    med = torch.sort(medoids[medoids >= 0]).values
    xm = base[med]
    sm = xm @ xm.t()
    sm.fill_diagonal_(-2.0)
    k1 = min(4 * top_degree, med.shape[0] - 1)
    sc, ci = sm.topk(k1, dim=1)
    pair = torch.stack([sm[ci[i]][:, ci[i]] for i in range(med.shape[0])]) if med.shape[0] <= 512 else None
    if pair is None:
        pair = torch.empty(med.shape[0], k1, k1, device=dev)
        for s in range(0, med.shape[0], 256):
            cvm = xm[ci[s:s + 256]]
            pair[s:s + 256] = torch.bmm(cvm, cvm.transpose(1, 2))
    kept = _robust_prune((1 + sc) / 2, (1 + pair) / 2, top_degree)
    rank = torch.cumsum(kept.long(), 1) - 1
    top_nbrs = torch.full((med.shape[0], top_degree), -1, dtype=torch.int32, device=dev)
    rows = torch.arange(med.shape[0], device=dev)[:, None].expand_as(ci)
    top_nbrs[rows[kept], rank[kept]] = med[ci[kept]].int()
    mean = base[torch.randperm(N, device=dev, generator=torch.Generator(device=dev).manual_seed(seed))[:100000]].mean(0)
    entry = int(med[(xm @ mean).argmax()])
    levels = [(None, packed.cpu().numpy()), (med.int().cpu().numpy(), top_nbrs.cpu().numpy())]
    return levels, entry, 1, packed


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
