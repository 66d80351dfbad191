"""benchgraph.py — synthetic hierarchical graph index for bench.py (INPUT PREPARATION, torch on the device).

NOT the reference's GraphIndexBuilder (host-side, concurrent, out of scope — SURVEY §2.1 #9; GPU-assisted
construction is a later 'next' row).  It produces an HNSW/Vamana-shaped structure with the reference's layout so the
host batched searcher has something realistic to traverse at 10M nodes:

  * nested levels like GraphIndexBuilder's level draw (level l holds ~N / 32^l nodes, B/graph/GraphIndexBuilder.java:
    562-575 with ml = 1/ln(degree)): sparse upper levels give the long jumps, level 0 holds every node;
  * per level: exact cosine k-NN candidates (brute force for small levels, coarse-cluster pools for large ones),
    then the robust-prune rule of VamanaDiversityProvider.retainDiverse (B/graph/diversity/VamanaDiversityProvider.java:
    45-96: keep i iff for every kept j  sim(i,j) <= score(i)*alpha, alpha 1.0 then 1.2) in the (1+cos)/2 domain;
  * reverse edges fill the free slots (a light stand-in for backlink + prune);
  * entry = the top-level node nearest the data mean.
"""
from __future__ import annotations

import torch


def _kmeans(x, C, seed, iters=8, sample_per=128):
    g = torch.Generator(device=x.device).manual_seed(seed)
    n = x.shape[0]
    xs = x[torch.randperm(n, generator=g, device=x.device)[: min(n, C * sample_per)]]
    cent = xs[torch.randperm(xs.shape[0], generator=g, device=x.device)[:C]].clone()
    for _ in range(iters):
        a = torch.cat([(xs[s:s + 262144] @ cent.t()).argmax(1) for s in range(0, xs.shape[0], 262144)])
        sums = torch.zeros_like(cent).index_add_(0, a, xs)
        cnt = torch.zeros(C, device=x.device).index_add_(0, a, torch.ones_like(a, dtype=torch.float32))
        cent = torch.where(cnt[:, None] > 0, sums / cnt.clamp(min=1)[:, None], cent)
        cent = cent / cent.norm(dim=1, keepdim=True).clamp(min=1e-12)
    return cent


def _robust_prune(score, pair, keep_n, alpha_max=1.2):
    """score [B, K] similarities of the candidates to the node, descending; pair [B, K, K] candidate-candidate
    similarities.  Returns a bool mask [B, K] of kept candidates (<= keep_n per row)."""
    B, K = score.shape
    kept = torch.zeros(B, K, dtype=torch.bool, device=score.device)
    n_kept = torch.zeros(B, dtype=torch.long, device=score.device)
    for alpha in (1.0, alpha_max):
        for i in range(K):
            occl = ((pair[:, i, :] > (score[:, i] * alpha)[:, None]) & kept).any(1)
            take = (~occl) & (~kept[:, i]) & (n_kept < keep_n) & torch.isfinite(score[:, i])
            kept[:, i] |= take
            n_kept += take.long()
    return kept


def _select(x_rows, row_ids, x_pool, pool_ids, x_all, n_cand, n_rand, fwd_degree, max_degree, gen, segs=None, per_seg=8):
    """Forward neighbours of `x_rows` (level-local ids `row_ids`): candidates = exact top-`n_cand` of the pool
    (level-local ids `pool_ids`) + `n_rand` uniformly random level nodes (Vamana's random initial edges — they are
    what survives occlusion as long-range links), robust-pruned.  Returns [b, max_degree] level-local ids, -1 padded."""
    dev = x_rows.device
    b = x_rows.shape[0]
    k = min(n_cand, x_pool.shape[0] - 1)
    s = x_rows @ x_pool.t()
    s[pool_ids[None, :] == row_ids[:, None]] = -2.0                       # no self edge
    sc, ci = s.topk(k, dim=1)
    if segs is not None and len(segs) > 1:
        # diversified candidates: besides the global top-k (dominated by the node's own dense cluster), the best
        # `per_seg` nodes of EVERY neighbouring coarse cell — the medium-range edges a search-based builder finds
        # along its paths, which robust-prune keeps because own-cluster neighbours do not occlude them
        extra_sc, extra_ci = [], []
        for lo, hi in segs[1:]:
            if hi - lo <= 0:
                continue
            kk = min(per_seg, hi - lo)
            e_sc, e_ci = s[:, lo:hi].topk(kk, dim=1)
            extra_sc.append(e_sc)
            extra_ci.append(e_ci + lo)
        if extra_sc:
            e_sc, e_ci = torch.cat(extra_sc, 1), torch.cat(extra_ci, 1)
            dup = (e_ci[:, :, None] == ci[:, None, :]).any(2)
            e_sc = torch.where(dup, torch.full_like(e_sc, float("-inf")), e_sc)
            sc, ci = torch.cat([sc, e_sc], 1), torch.cat([ci, e_ci], 1)
            o = torch.argsort(sc, dim=1, descending=True, stable=True)
            sc, ci = sc.gather(1, o), ci.gather(1, o)
    cid = pool_ids[ci]                                                    # [b, k] level-local ids
    if n_rand > 0:
        rid = torch.randint(0, x_all.shape[0], (b, n_rand), generator=gen, device=dev)
        xr = x_all[rid]                                                   # [b, R, D]
        sr = torch.einsum("bd,brd->br", x_rows, xr)
        dup = (rid[:, :, None] == cid[:, None, :]).any(2) | (rid == row_ids[:, None])
        dup |= torch.triu((rid[:, :, None] == rid[:, None, :]), diagonal=1).any(1)   # repeated random picks
        sr = torch.where(dup, torch.full_like(sr, float("-inf")), sr)
        sc = torch.cat([sc, sr], 1)
        cid = torch.cat([cid, rid], 1)
        cv = torch.cat([x_pool[ci], xr], 1)
        o = torch.argsort(sc, dim=1, descending=True, stable=True)
        sc = sc.gather(1, o)
        cid = cid.gather(1, o)
        cv = cv.gather(1, o[:, :, None].expand(-1, -1, cv.shape[2]))
    else:
        cv = x_pool[ci]
    pair = torch.bmm(cv, cv.transpose(1, 2))
    kept = _robust_prune((1 + sc) / 2, (1 + pair) / 2, fwd_degree)
    rank = torch.cumsum(kept.long(), 1) - 1
    sel = torch.full((b, max_degree), -1, dtype=torch.long, device=dev)
    rows = torch.arange(b, device=dev)[:, None].expand_as(cid)
    sel[rows[kept], rank[kept]] = cid[kept]
    return sel


def _add_reverse_and_pack(nbrs, max_degree):
    """nbrs [n, max_degree] LOCAL ids (-1 padded): add reverse edges into free slots, drop duplicates, re-pack."""
    dev = nbrs.device
    n = nbrs.shape[0]
    deg = (nbrs >= 0).sum(1)
    mask = nbrs >= 0
    src = torch.arange(n, device=dev)[:, None].expand(n, max_degree)[mask]
    dst = nbrs[mask]
    o = torch.argsort(dst, stable=True)
    src, dst = src[o], dst[o]
    first = torch.searchsorted(dst, torch.arange(n, device=dev))
    slot = deg[dst] + (torch.arange(dst.shape[0], device=dev) - first[dst])
    ok = slot < max_degree
    dst, src, slot = dst[ok], src[ok], slot[ok]
    exists = (nbrs[dst] == src[:, None]).any(1)
    nbrs[dst[~exists], slot[~exists]] = src[~exists]
    valid = nbrs >= 0
    rank = torch.cumsum(valid.long(), 1) - 1
    packed = torch.full_like(nbrs, -1)
    rows = torch.arange(n, device=dev)[:, None].expand_as(nbrs)
    packed[rows[valid], rank[valid]] = nbrs[valid]
    return packed


def _level_knn(x, max_degree, fwd_degree, n_cand, seed, n_probe=5, row_chunk=4096, n_rand=32):
    """x [n, D] unit vectors of ONE level -> [n, max_degree] local neighbour ids (packed, -1 padded)."""
    dev = x.device
    n = x.shape[0]
    nbrs = torch.full((n, max_degree), -1, dtype=torch.long, device=dev)
    gen = torch.Generator(device=dev).manual_seed(seed + 7919)
    if n <= 40000:
        all_ids = torch.arange(n, device=dev)
        # small (upper) levels do the routing between clusters: give robust-prune a WIDE candidate list (up to 512
        # nearest) so the kept edges cover all directions (MRNG-like; greedy descent with beam 1 then rarely stalls)
        k_here = min(n - 1, 512 if n <= 12000 else 256)
        chunk = 512 if n <= 12000 else 1024
        for r0 in range(0, n, chunk):
            r1 = min(n, r0 + chunk)
            nbrs[r0:r1] = _select(x[r0:r1], all_ids[r0:r1], x, all_ids, x, k_here, n_rand, fwd_degree, max_degree, gen)
        return _add_reverse_and_pack(nbrs, max_degree)
    C = max(8, min(8192, n // 2500))
    cent = _kmeans(x, C, seed)
    assign = torch.empty(n, dtype=torch.long, device=dev)
    for s in range(0, n, 1_000_000):
        assign[s:s + 1_000_000] = (x[s:s + 1_000_000] @ cent.t()).argmax(1)
    order = torch.argsort(assign, stable=True)
    offs = torch.zeros(C + 1, dtype=torch.long, device=dev)
    offs[1:] = torch.cumsum(torch.bincount(assign, minlength=C), 0)
    offs_h = offs.cpu().tolist()
    csim = cent @ cent.t()
    csim.fill_diagonal_(-2.0)
    near = csim.topk(min(n_probe, C - 1), dim=1).indices.cpu().tolist()
    for c in range(C):
        lo, hi = offs_h[c], offs_h[c + 1]
        if hi == lo:
            continue
        own = order[lo:hi]
        parts = [own] + [order[offs_h[p]:offs_h[p + 1]] for p in near[c]]
        pool = torch.cat(parts)
        segs, acc = [], 0
        for part in parts:
            segs.append((acc, acc + part.shape[0]))
            acc += part.shape[0]
        xp = x[pool]
        for r0 in range(0, hi - lo, row_chunk):
            r1 = min(hi - lo, r0 + row_chunk)
            nbrs[own[r0:r1]] = _select(xp[r0:r1], own[r0:r1], xp, pool, x, n_cand, n_rand, fwd_degree, max_degree, gen,
                                       segs=segs)
    return _add_reverse_and_pack(nbrs, max_degree)


def build_hier_graph(base, max_degree=32, upper_degree=32, fanout=8, seed=11, n_cand=64, min_top=16):
    """Returns (levels, entry_node, entry_level, nbrs0_dev):
    levels[l] = (None | sorted int32 node ids, int32 neighbours [count, degree]) on the HOST (numpy);
    nbrs0_dev = level-0 neighbours on the device (for building the FusedPQ blocks)."""
    dev = base.device
    N = base.shape[0]
    g = torch.Generator(device=dev).manual_seed(seed)
    perm = torch.randperm(N, generator=g, device=dev)
    sizes = []
    n = N // fanout
    while n >= min_top:
        sizes.append(n)
        n //= fanout
    levels = []
    nb0 = _level_knn(base, max_degree, (3 * max_degree) // 4, n_cand, seed)
    levels.append((None, nb0.int().cpu().numpy()))
    last_nodes = None
    for li, sz in enumerate(sizes):
        nodes = torch.sort(perm[:sz]).values            # nested: perm[:sz_l+1] is a subset of perm[:sz_l]
        xl = base[nodes]
        nbl = _level_knn(xl, upper_degree, (3 * upper_degree) // 4, n_cand, seed + 1 + li)
        glob = torch.where(nbl >= 0, nodes[nbl.clamp(min=0)], nbl)
        levels.append((nodes.int().cpu().numpy(), glob.int().cpu().numpy()))
        last_nodes = nodes
    if last_nodes is None:
        entry, entry_level = int(perm[0]), 0
    else:
        mean = base[perm[: min(N, 100000)]].mean(0)
        entry = int(last_nodes[(base[last_nodes] @ mean).argmax()])
        entry_level = len(levels) - 1
    return levels, entry, entry_level, nb0.int()
