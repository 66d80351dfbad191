// rt_body.h — exact-score ties at the K-th place of the rerank, resolved on the device (body of rerank_tie_kernel,
// k_topk.hip; compiled unchanged for the CPU lane emulator by tests/mock/mock_kernels.cpp).
//
// NodeQueue.rerank (B/graph/NodeQueue.java:160-230) walks approximateResults in HEAP ARRAY order and keeps a reranked entry
// only while the bounded queue has room or its exact score is STRICTLY better than the worst kept one (:204-211).  When more
// candidates carry the K-th exact score than the selection returned, the survivors therefore depend on that array order —
// which is a function of the whole sequence of addTopCandidate calls (GraphSearcher.java:515-530) on the reference's binary
// heap (AbstractLongHeap.java:77-85,158-187), evicted entries included.  The traversal kernel logs that sequence (every
// layer-0 candidate that passed acceptOrds and `score >= 0`, gs_body.h); here one wavefront per query
//   1. detects the tie (float equality on purpose: -0.0 and 0.0 tie under the reference's `>` although their keys differ),
//   2. replays the log into the reference's heap (lane 0; keys are unique, so the array equals the reference's),
//   3. walks the array through NodeQueue.rerank's loop with the exact scores the rerank kernel already produced
//      (the candidate list is searched lane-parallel), and
//   4. rewrites the query's top-K, best first (the order rerankedResults pops in, :497-502).
// A log that overflowed its capacity leaves the query marked GS_RERANK_TIE for the host searcher.
#pragma once
#include <cstdint>

#include "gs_params.h"

namespace jv {

GS_FN float rt_key_score(long long k)
{
    const int32_t e = (int32_t)(k >> 32);
    const int32_t bits = e ^ ((e >> 31) & 0x7fffffff);
    union { int32_t i; float f; } u;
    u.i = bits;
    return u.f;
}
GS_FN int32_t rt_key_node(long long k) { return (int32_t)~(uint32_t)(k & 0xFFFFFFFFll); }
GS_FN long long rt_key(int32_t node, float score)
{
    union { int32_t i; float f; } u;
    u.f = score;
    int32_t bits = (score != score) ? 0x7fc00000 : u.i;
    const int32_t s = bits ^ ((bits >> 31) & 0x7fffffff);
    return (long long)(((unsigned long long)(uint32_t)s) << 32) | (long long)(0xFFFFFFFFull & (unsigned long long)(uint32_t)(~node));
}
GS_FN void rt_up(long long *h, int i)  // AbstractLongHeap.upHeap :158-168 (0-based)
{
    const long long v = h[i];
    while (i > 0) {
        const int p = (i - 1) >> 1;
        if (!(v < h[p])) break;
        h[i] = h[p];
        i = p;
    }
    h[i] = v;
}
GS_FN void rt_down(long long *h, int n, int i)  // downHeap :170-187
{
    const long long v = h[i];
    for (;;) {
        int j = 2 * i + 1;
        if (j >= n) break;
        if (j + 1 < n && h[j + 1] < h[j]) ++j;
        if (!(h[j] < v)) break;
        h[i] = h[j];
        i = j;
    }
    h[i] = v;
}

// one wavefront; lds: rt_lds_bytes(rerankK, K)
GS_FN void rt_query(const RtParams &p, int q, char *lds_raw)
{
    const int lane = gs_lane();
    const int K = p.K, R = p.R;
    const int32_t *cid = p.cand_ids + (int64_t)q * R;
    const float *csc = p.cand_sc + (int64_t)q * R;
    float *osc = p.out_sc + (int64_t)q * K;
    int32_t *oid = p.out_ids + (int64_t)q * K;
    if (p.status[q] != GS_OK) return;  // already on its way to the host searcher
    if (oid[K - 1] < 0) return;         // fewer than K results: everything was kept
    const float sk = osc[K - 1];
    int all = 0, sel = 0;
    for (int i = lane; i < R; i += 64)
        if (cid[i] >= 0 && csc[i] == sk) ++all;
    for (int i = lane; i < K; i += 64)
        if (osc[i] == sk) ++sel;
    for (int o = 32; o > 0; o >>= 1) {
        all += (int)gs_shfl_xor((long long)all, o);
        sel += (int)gs_shfl_xor((long long)sel, o);
    }
    if (all <= sel) return;
    const int n_log = p.push_log ? p.push_log_n[q] : -1;
    if (n_log < 0 || n_log > p.log_cap) {  // no usable log: the host searcher replays the query
        if (lane == 0) {
            p.status[q] = GS_RERANK_TIE;
            (void)gs_fetch_add(p.count, 1u);
        }
        return;
    }
    long long *heap = reinterpret_cast<long long *>(lds_raw);  // approximateResults, then rerankedResults behind it
    long long *rer = heap + p.rerankK;
    long long *xch = rer + K;                                   // lane-parallel search result
    const long long *log = p.push_log + (int64_t)q * p.log_cap;
    int size = 0;
    if (lane == 0) {  // addTopCandidate :515-530 for every logged candidate
        for (int t = 0; t < n_log; ++t) {
            const long long key = log[t];
            if (size < p.rerankK) {
                heap[size] = key;
                rt_up(heap, size);
                ++size;
            } else if (rt_key_score(key) > rt_key_score(heap[0])) {
                heap[0] = key;
                rt_down(heap, size, 0);
            }
        }
        xch[0] = size;
    }
    gs_barrier();
    size = (int)xch[0];
    gs_barrier();
    int rn = 0;
    for (int i = 0; i < size; ++i) {  // NodeQueue.rerank :197-214 in array order
        const int32_t node = rt_key_node(heap[i]);
        float ex = 0.0f;
        bool hit = false;
        for (int j = lane; j < R; j += 64)
            if (cid[j] == node) {
                ex = csc[j];
                hit = true;
            }
        const uint64_t hm = gs_ballot(hit);
        if (hm == 0) continue;  // cannot happen: the log's survivors ARE the candidate list
        int src = 0;
        while (!((hm >> src) & 1ull)) ++src;
        union { float f; int32_t i; } u;
        u.f = ex;
        u.i = (int32_t)gs_shfl((long long)u.i, src);
        ex = u.f;
        if (lane == 0) {
            if (rn < K) {
                rer[rn] = rt_key(node, ex);
                rt_up(rer, rn);
            } else if (ex > rt_key_score(rer[0])) {
                rer[0] = rt_key(node, ex);
                rt_down(rer, K, 0);
            }
        }
        if (rn < K) ++rn;
    }
    gs_barrier();
    if (lane == 0) {  // :497-502: pop the worst first, fill from the back
        (void)gs_fetch_add(p.count + 1, 1u);
        int n = rn;
        for (int i = rn - 1; i >= 0; --i) {
            const long long k = rer[0];
            rer[0] = rer[--n];
            if (n > 0) rt_down(rer, n, 0);
            oid[i] = rt_key_node(k);
            osc[i] = rt_key_score(k);
        }
    }
}

}  // namespace jv
