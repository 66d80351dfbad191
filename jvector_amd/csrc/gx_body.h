// gx_body.h — the WORKGROUP form of the device-resident graph traversal: one query per workgroup, the query's ADC table in LDS.
//
// Why (DESIGN.md §4 "the workgroup form"): the one-wave-per-query kernels (gs_body.h) score table-free — each scored neighbour
// gathers M codebook rows of 32 bytes from L2 — because eight 96 KB tables do not fit a CU's LDS, and they are bound by the rate at
// which the vector-memory pipe takes divergent 16-byte requests (TD busy 0.96).  The reference builds ONE table per query
// (PQDecoder.java:41-54, FusedPQDecoder.java:66-76) and a scored neighbour costs M look-ups of 4 bytes.  This form does the same
// on a CU: the table (M x 256 f32 = 96 KB at PQ-96) is built once per query into LDS by the whole workgroup and every score is M
// ds_read_b32 + M adds in ascending m (assembleAndSum, DefaultVectorUtilSupport.java:302-309,323-330).  One query per CU leaves
// nothing to hide the HBM latency of an expansion's adjacency row + FusedPQ block behind, so the work is split by ROLE:
//   wave 0           the control wave: GraphSearcher's loop (gs_search_one<..., WGX = true>) — candidate / result queues, visited set,
//                    stop rule; it never touches an adjacency row or a code byte
//   waves 1 .. E     expanders: take (node, level) requests from a ring in LDS, read the node's adjacency row + the neighbours' code
//                    bytes (the FusedPQ block at level 0) from HBM, score EVERY neighbour of the row against the table and leave one
//                    NodeQueue key per neighbour in the request's slot
// The control wave asks for a row as soon as its node can be foreseen to be popped (the best remaining candidate at every pop, the best
// fresh neighbour of every expansion), so that most pops find their row already scored.  A scored row depends on nothing but
// (query, node, level): speculation cannot change a result, a visit count or an expansion count — the control wave applies
// visited.add / candidates.push in exactly the reference's order (GraphSearcher.java:406-457).
//
// Included after gs_body.h, same wave API (the includer makes gs_barrier() a WAVE-scope sync point for this form).
#pragma once

#include "gs_body.h"

namespace jv {

// the whole workgroup: stage the centred query, then write table entry (m, code) for every code of every subspace —
// calculatePartialSums (DefaultVectorUtilSupport.java:351-365) entry by entry, the arithmetic of gs_lut_entry / lut_build_kernel
template <int VSF>
GS_FN void gx_lut_build(const GsParams &p, int q, float *qs, float *lut, int tid, int nthreads)
{
    {
        const gs_f4 *src = reinterpret_cast<const gs_f4 *>(p.cq + (int64_t)q * p.D);
        gs_f4 *dst = reinterpret_cast<gs_f4 *>(qs);
        for (int i = tid; i < p.D / 4; i += nthreads) dst[i] = src[i];
    }
    gs_block_barrier();
    // eight entries per thread and pass, their codebook rows (32 B each, consecutive threads -> consecutive rows) loaded before the
    // first one is used: the build is bound by the L2 -> CU stream of the 768 KB codebook, not by one load's latency
    const int total = p.M * 256;
    for (int base = tid; base < total; base += 8 * nthreads) {
        gs_f4 c0[8], c1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * nthreads;
            if (i < total) {
                const gs_f4 *cp = reinterpret_cast<const gs_f4 *>(p.codebooks + (int64_t)i * 8);
                c0[u] = cp[0];
                c1[u] = cp[1];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * nthreads;
            if (i < total) lut[i] = gs_lut_entry_from<VSF>(c0[u], c1[u], qs + (i >> 8) * 8);
        }
    }
}

// An expander wave's service loop for one query.  lane = 0..63 inside the wave.
template <int VSF, int CH16>
GS_FN void gx_expander(const GsParams &p, int q, char *sh, int lane)
{
    int32_t *hdr = reinterpret_cast<int32_t *>(sh);
    int32_t *ring = reinterpret_cast<int32_t *>(sh + gx_off_ring());
    int32_t *slot_node = reinterpret_cast<int32_t *>(sh + gx_off_slot_node());
    int32_t *slot_lvl = reinterpret_cast<int32_t *>(sh + gx_off_slot_lvl());
    int32_t *slot_state = reinterpret_cast<int32_t *>(sh + gx_off_slot_state());
    long long *keys = reinterpret_cast<long long *>(sh + gx_off_keys());
    const float *lut = reinterpret_cast<const float *>(sh + gx_off_lut(p.wgx_slots, p.wgx_kps, p.wgx_log));
    const float query_mag = (VSF == 2) ? p.bmag[q] : 0.0f;
    for (;;) {
        // take a ticket; wait until the control wave has posted that many requests (or the query is over)
        int32_t tv = 0;
        if (lane == 0) tv = gs_lds_add(hdr + GX_REQ_HEAD, 1);
        const int32_t ticket = gs_shfl32(tv, 0);
        for (;;) {
            int32_t st = 0;
            if (lane == 0) st = gs_lds_load(hdr + GX_REQ_TAIL) > ticket ? 1 : (gs_lds_load(hdr + GX_QUIT) ? -1 : 0);
            st = gs_shfl32(st, 0);
            if (st > 0) break;
            if (st < 0) return;
            gs_spin_pause();
        }
        const int slot = ring[ticket & (GX_RING - 1)];
        const int32_t node = slot_node[slot];
        const int lvl = slot_lvl[slot];
        const GsLevel &L = p.lv[lvl];
        const int32_t *row = gs_level_row(L, node);
        const int deg = L.degree;
        const bool fused0 = lvl == 0 && p.blocks != nullptr;
        const int32_t nb = (row && lane < deg) ? row[lane] : -1;
        gs_u4 w[CH16];
        float node_mag = 0.0f;
        if (fused0 && row && lane < deg) {   // FusedPQDecoder.similarityToNeighbor: the origin's packed block (zero padded)
            const int64_t r = (int64_t)node * p.deg0 + lane;
            gs_load_row<CH16>(p.blocks + r * p.M, w);
            if (VSF == 2) node_mag = p.fused_norms[r];
        }
        const int first_neg = gs_first(gs_ballot(nb < 0));   // rows are packed: the first -1 ends the row
        const bool valid = lane < first_neg;
        if (!fused0 && valid) {               // PQDecoder.similarityTo: the neighbour's own code
            gs_load_row<CH16>(p.codes + (int64_t)nb * p.M, w);
            if (VSF == 2) node_mag = p.code_norms[nb];
        }
        long long key = GX_KEY_NONE;
        if (valid) key = gs_key(nb, gs_finish<VSF>(gx_row_sum<CH16>(lut, w), node_mag, query_mag));
        if (lane < p.wgx_kps) keys[slot * p.wgx_kps + lane] = key;
        gs_barrier();
        if (lane == 0) gs_lds_store(slot_state + slot, GX_READY);   // release: whoever sees READY sees the keys
    }
}

// Persistent workgroup: pulls queries off the shared counter until none are left.  Every wave of the workgroup calls it.
template <int VSF, int CH16, bool PROF = false>
GS_FN void gx_worker(const GsParams &p, int worker, char *lds)
{
    const int tid = gs_tid(), nthreads = gs_block_threads();
    const int evict_cap = p.evict_cap > 0 ? p.evict_cap : GS_EVICT_CAP;
    char *sh = lds + gx_ctl_bytes(p.D, p.rerankK, p.cand_cap, evict_cap, p.v1_log2);
    int32_t *hdr = reinterpret_cast<int32_t *>(sh);
    float *lut = reinterpret_cast<float *>(sh + gx_off_lut(p.wgx_slots, p.wgx_kps, p.wgx_log));
    for (;;) {
        if (tid == 0) {
            hdr[GX_ITEM] = (int32_t)gs_fetch_add(p.next_query, 1u);
            hdr[GX_REQ_HEAD] = 0;
            hdr[GX_REQ_TAIL] = 0;
            hdr[GX_QUIT] = 0;
        }
        gs_block_barrier();
        const int item = hdr[GX_ITEM];
        if (item >= p.Q) break;
        const int q = p.qmap ? p.qmap[item] : item;
        unsigned long long t0 = 0;
        if (PROF) t0 = GS_CLOCK();
        gx_lut_build<VSF>(p, q, reinterpret_cast<float *>(lds), lut, tid, nthreads);
        gs_block_barrier();
        if (PROF && p.prof && tid == 0) gs_fetch_add64(p.prof + 15, GS_CLOCK() - t0);
        if (tid < 64) gs_search_one<VSF, CH16, false, PROF, false, false, true>(p, q, worker, lds);
        else gx_expander<VSF, CH16>(p, q, sh, tid & 63);
        gs_block_barrier();   // nobody is inside this query's LDS state any more
    }
}

}  // namespace jv
