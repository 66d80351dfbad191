// gx_body.h — the WORKGROUP form of the device-resident graph traversal: one query per workgroup, the query's ADC table in LDS.
//
// Why (DESIGN.md §4 "the workgroup form"): the one-wave-per-query kernels (gs_body.h) score table-free — each scored neighbour
// gathers M codebook rows of 32 bytes from L2 — because eight 96 KB tables do not fit a CU's LDS, and they are bound by the rate at
// which the vector-memory pipe takes divergent 16-byte requests (TD busy 0.96).  The reference builds ONE table per query
// (PQDecoder.java:41-54, FusedPQDecoder.java:66-76) and a scored neighbour costs M look-ups of 4 bytes.  This form does the same
// on a CU: the table (M x 256 f32 = 96 KB at PQ-96) is built once per query into LDS by the whole workgroup and every score is M
// ds_read_b32 + M adds in ascending m (assembleAndSum, DefaultVectorUtilSupport.java:302-309,323-330).  One query per CU leaves
// nothing to hide the HBM latency of an expansion's adjacency row + FusedPQ block behind, so the work is split by ROLE:
//   wave 0           the control wave: GraphSearcher's loop (gx_control below) — candidate / result queues, visited set,
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
    const int total = p.wgx_lut_m * 256;
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
template <int VSF, int CH16, bool FULL>
GS_FN void gx_expander(const GsParams &p, int q, char *sh, int lane, const char *qs_lds)
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
        if (valid) key = gs_key(nb, gs_finish<VSF>(gx_row_sum<VSF, CH16, FULL>(lut, p.wgx_lut_m, p.codebooks, reinterpret_cast<const float *>(qs_lds), w), node_mag, query_mag));
        if (lane < p.wgx_kps) keys[slot * p.wgx_kps + lane] = key;
        gs_barrier();
        if (lane == 0) gs_lds_store(slot_state + slot, GX_READY);   // release: whoever sees READY sees the keys
    }
}

// ---- 32-bit wave reductions (the includer may supply DPP forms: GS_HAVE_WAVE_REDUCE32) ----
#ifndef GS_HAVE_WAVE_REDUCE32
GS_FN int32_t gs_wave_max_i32(int32_t v) { return (int32_t)gs_wave_max((long long)v); }
GS_FN int32_t gs_wave_min_i32(int32_t v) { return (int32_t)gs_wave_min((long long)v); }
GS_FN uint32_t gs_wave_max_u32(uint32_t v) { return (uint32_t)gs_wave_max((long long)(unsigned long long)v); }
// a value every lane holds identically, as a wave-uniform scalar (v_readfirstlane_b32 on the GPU)
GS_FN int32_t gs_uniform(int32_t v) { return v; }
#endif

constexpr int GX_HOT = 256;                        // candidates' LDS tier of the workgroup form (4 keys per lane: they are scanned from registers)
constexpr int32_t GX_HI_NONE = (int32_t)0x80000000;  // "no key" in a score word: floatToSortableInt never yields it (NaN is canonical)

GS_FN long long gx_join(int32_t hi, uint32_t lo) { return (long long)(((unsigned long long)(uint32_t)hi << 32) | (unsigned long long)lo); }
GS_FN float gx_hi_score(int32_t hi) { return gs_bits_float(hi ^ ((hi >> 31) & 0x7fffffff)); }

// The control wave of the workgroup form: GraphSearcher's loop for one query (what gs_search_one does, the same reference lines:
// GraphSearcher.java:263-282, 334-369, 406-457, 324-331, 515-530) written for a wave that runs ALONE on its SIMD — a lone wave
// issues one vector instruction per 4-5 clocks, so what counts is the number of instructions per expansion, not memory latency:
//   * NodeQueue keys are handled as two 32-bit words (score word hi = floatToSortableInt, node word lo = ~node): the candidates'
//     LDS tier is stored as two arrays, a pop scans score words only (4 per lane, from registers), reduces them with 32-bit DPP
//     steps, and looks at node words only when two candidates carry the same score;
//   * it never touches an adjacency row or a code byte: rows arrive scored in slots (see the head of this file);
//   * the push log waits in LDS until the query ends.
// candidates = LDS tier (GX_HOT keys) + the global spill tier of gs_body.h (same invariant: every LDS key > spill_max >= every
// spilled key); results / evicted = 64-bit key arrays in LDS as in gs_body.h; visited = the two-tier set of gs_body.h.
template <int VSF, int CH16, bool PROF, bool FULL>
GS_FN void gx_control(const GsParams &p, int q, int worker, char *lds)
{
    const int lane = gs_lane();
    const uint64_t lt = (1ull << lane) - 1ull;
    const int evict_cap = p.evict_cap > 0 ? p.evict_cap : GS_EVICT_CAP;
    // ---- LDS: [query][results rerankK x 8][candidates: score words 256 x 4, node words 256 x 4][evicted][64 x 8 scratch][visited tier 1]
    long long *res = reinterpret_cast<long long *>(lds + gs_q_bytes(p.D));
    int32_t *hot_hi = reinterpret_cast<int32_t *>(res + p.rerankK);
    uint32_t *hot_lo = reinterpret_cast<uint32_t *>(hot_hi + GX_HOT);
    long long *evicted = res + p.rerankK + p.cand_cap;
    GsVis1 v1;
    v1.w = reinterpret_cast<uint32_t *>(lds + ((gs_lds_bytes(p.D, p.rerankK, p.cand_cap, 0, evict_cap, 0) + 15) & ~(size_t)15));
    v1.bmask = p.v1_log2 > 0 ? (1u << (p.v1_log2 - 2)) - 1u : 0u;
    v1.idmask = (p.v1_idbits >= 32) ? 0xFFFFFFFFu : ((1u << p.v1_idbits) - 1u);
    v1.rbits = p.v1_idbits > p.v1_log2 - 2 ? p.v1_idbits - (p.v1_log2 - 2) : 0;
    const bool has_v1 = p.v1_log2 > 0;
    char *sh = lds + gx_ctl_bytes(p.D, p.rerankK, p.cand_cap, evict_cap, p.v1_log2);
    int32_t *hdr = reinterpret_cast<int32_t *>(sh);
    int32_t *ring = reinterpret_cast<int32_t *>(sh + gx_off_ring());
    int32_t *slot_node = reinterpret_cast<int32_t *>(sh + gx_off_slot_node());
    int32_t *slot_lvl = reinterpret_cast<int32_t *>(sh + gx_off_slot_lvl());
    int32_t *slot_state = reinterpret_cast<int32_t *>(sh + gx_off_slot_state());
    const long long *slot_keys = reinterpret_cast<const long long *>(sh + gx_off_keys());
    long long *log_lds = reinterpret_cast<long long *>(sh + gx_off_log(p.wgx_slots, p.wgx_kps));
    const float *lut = reinterpret_cast<const float *>(sh + gx_off_lut(p.wgx_slots, p.wgx_kps, p.wgx_log));

    unsigned long long pf[5] = {0, 0, 0, 0, 0}, fh[4] = {0, 0, 0, 0}, px[3] = {0, 0, 0}, pt = 0, pq0 = 0;
    if (PROF) pq0 = GS_CLOCK();
#define GX_PHASE(i)                                      \
    do {                                                 \
        if (PROF) {                                      \
            const unsigned long long now_ = GS_CLOCK();  \
            pf[i] += now_ - pt;                          \
            pt = now_;                                   \
        }                                                \
    } while (0)

    // ---- wave-uniform state ----
    int hot_n = 0, spill_n = 0, res_n = 0, ev_n = 0, res_min_idx = -1, log_n = 0, tail = 0;
    long long spill_max = GS_KEY_MIN, res_min = GS_KEY_MAX;
    long long *spill = p.spill + (int64_t)worker * p.spill_cap;
    int32_t status = GS_OK;
    int n_visited = 0, n_expanded = 0;
    const int vcap = 1 << p.vcap_log2;
    const uint32_t vmask = (uint32_t)vcap - 1u;
    const int vshift = 32 - p.vcap_log2;
    int32_t *vis = p.visited + (int64_t)worker * vcap;
    int n2 = 0;                  // nodes in tier 2 of the visited set
    bool t2_ready = !has_v1;
    // per-lane: the (node, level) slot `lane` holds (-1 = free) and the candidate key it was requested for
    int32_t sl_node = -1, sl_lvl = 0;
    long long sl_ckey = 0;

    // ---- the scored-row slots.  Slot t is managed by lane t: sl_node / sl_lvl / sl_ckey are PER-LANE registers (the (node, level) the
    //      slot holds, -1 = free; the candidate key it was requested for, which decides evictions).  slot_find: the slot holding
    //      (node, level) or -1.  slot_post: ask the expanders for a row; `must` (the popped node itself) evicts the READY slot whose
    //      candidate is the worst when every slot is taken (its row is simply requested again should that candidate ever be popped);
    //      a speculative request just gives up (-1). ----
    auto slot_find = [&](int32_t node, int lvl) -> int {
        const uint64_t m = gs_ballot(lane < p.wgx_slots && sl_node == node && sl_lvl == lvl);
        return m ? gs_first(m) : -1;
    };
    auto slot_post = [&](int32_t node, int lvl, long long ckey, bool must) -> int {
        const uint64_t fm = gs_ballot(lane < p.wgx_slots && sl_node == -1);
        int slot;
        if (fm) {
            slot = gs_first(fm);
        } else if (!must) {
            return -1;
        } else {   // every slot is taken: evict the READY row whose candidate is the worst (it is requested again if ever popped)
            uint64_t rm;
            for (;;) {
                rm = gs_ballot(lane < p.wgx_slots && gs_lds_load(slot_state + lane) == GX_READY);
                if (rm) break;
                gs_spin_pause();
            }
            const bool mine = ((rm >> lane) & 1ull) != 0;
            const long long mn = gs_wave_min(mine ? sl_ckey : GS_KEY_MAX);
            slot = gs_first(gs_ballot(mine && sl_ckey == mn));
        }
        if (lane == slot) {
            sl_node = node;
            sl_lvl = lvl;
            sl_ckey = ckey;
        }
        if (lane == 0) {
            slot_node[slot] = node;
            slot_lvl[slot] = lvl;
            slot_state[slot] = GX_REQUESTED;
            ring[tail & (GX_RING - 1)] = slot;
        }
        gs_barrier();
        tail++;
        if (lane == 0) gs_lds_store(hdr + GX_REQ_TAIL, tail);
        return slot;
    };

    // ---- visited.add for one node per participating lane (the two-tier set of gs_body.h; wave-uniform call) ----
    auto visit = [&](bool act, int32_t nb) -> bool {
        int r1 = act ? 2 : 0;
        if (has_v1 && act) r1 = gs_visit1(v1, nb);
        bool fr = r1 == 1;
        if (gs_ballot(r1 == 2)) {
            if (!t2_ready) {   // tier 2 is cleared by the first probe that needs it
                gs_u4 *v4 = reinterpret_cast<gs_u4 *>(vis);
                const gs_u4 ones = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
                for (int i = lane; i < vcap / 4; i += 64) v4[i] = ones;
                gs_fence();
                gs_barrier();
                t2_ready = true;
            }
            const bool f2 = r1 == 2 && gs_visit(vis, vmask, vshift, nb);
            fr = fr || f2;
            n2 += gs_popc(gs_ballot(f2));
            if ((n2 + 1) * 2 > vcap) status = GS_OVERFLOW;   // (no growth pool in this form: the retry launch takes the query)
        }
        return fr;
    };

    // ---- candidates.push for up to one key per lane ----
    // Move every LDS-tier key <= a pivot to the spill tier.  The pivot is one of the keys: a sample with 20..44 of the 64 samples
    // above it, found by counting (no ranking pass: 2 instructions per try, ~3 tries).
    auto partition = [&]() {
        const int32_t shi = hot_hi[(int)(((long long)lane * hot_n) >> 6)];
        const uint32_t slo = hot_lo[(int)(((long long)lane * hot_n) >> 6)];
        int32_t phi = 0;
        uint32_t plo = 0;
        for (int t = 0; t < 64; ++t) {
            phi = (int32_t)gs_shfl((long long)shi, t);
            plo = (uint32_t)gs_shfl((long long)slo, t);
            const int above = gs_popc(gs_ballot(shi > phi || (shi == phi && slo > plo)));
            if (above >= 20 && above <= 44) break;
        }
        int new_n = 0, moved = 0;
        for (int base = 0; base < hot_n; base += 64) {
            const int i = base + lane;
            const bool in = i < hot_n;
            const int32_t khi = in ? hot_hi[i] : 0;
            const uint32_t klo = in ? hot_lo[i] : 0u;
            const bool hi = in && (khi > phi || (khi == phi && klo > plo));
            const bool lo = in && !hi;
            const uint64_t mh = gs_ballot(hi), ml = gs_ballot(lo);   // every lane has read its key before any lane writes
            if (hi) {                                                 // in place: target index <= i
                hot_hi[new_n + gs_popc(mh & lt)] = khi;
                hot_lo[new_n + gs_popc(mh & lt)] = klo;
            }
            if (lo) {
                const int pos = spill_n + moved + gs_popc(ml & lt);
                if (pos < p.spill_cap) spill[pos] = gx_join(khi, klo);
            }
            new_n += gs_popc(mh);
            moved += gs_popc(ml);
            gs_barrier();
        }
        if (spill_n + moved > p.spill_cap) status = GS_OVERFLOW;
        spill_n += moved;
        hot_n = new_n;
        spill_max = gx_join(phi, plo);   // the pivot itself moved, everything that stayed is larger
    };
    auto push = [&](long long key, bool has) {
        const int32_t khi = (int32_t)(key >> 32);
        const uint32_t klo = (uint32_t)(unsigned long long)key;
        bool to_lds;
        uint64_t ml;
        for (;;) {
            to_lds = has && (spill_n == 0 || key > spill_max);
            ml = gs_ballot(to_lds);
            if (hot_n + gs_popc(ml) <= GX_HOT) break;
            partition();
            if (status != GS_OK) return;
        }
        if (to_lds) {
            hot_hi[hot_n + gs_popc(ml & lt)] = khi;
            hot_lo[hot_n + gs_popc(ml & lt)] = klo;
        }
        hot_n += gs_popc(ml);
        const bool to_sp = has && !to_lds;
        const uint64_t ms = gs_ballot(to_sp);
        if (ms) {
            if (spill_n + gs_popc(ms) > p.spill_cap) {
                status = GS_OVERFLOW;
                return;
            }
            if (to_sp) spill[spill_n + gs_popc(ms & lt)] = key;
            spill_n += gs_popc(ms);
        }
        gs_barrier();
    };
    // The LDS tier ran dry while keys wait in the spill tier: bring the best of them back (all of them if they fit in half the
    // tier, else those above a pivot aimed at a quarter of it) — gs_refill's rule; the two tiers' invariant holds afterwards.
    auto refill = [&]() {
        gs_fence();
        const int n = spill_n;
        if (n <= GX_HOT / 2) {
            for (int base = 0; base < n; base += 64)
                if (base + lane < n) {
                    const long long k = spill[base + lane];
                    hot_hi[base + lane] = (int32_t)(k >> 32);
                    hot_lo[base + lane] = (uint32_t)(unsigned long long)k;
                }
            hot_n = n;
            spill_n = 0;
            spill_max = GS_KEY_MIN;
            gs_barrier();
            return;
        }
        const long long mine = spill[(int)(((long long)lane * n) >> 6)];
        int rank = 0;
        for (int j = 0; j < 64; ++j) rank += (gs_shfl(mine, j) < mine) ? 1 : 0;   // keys are unique: the ranks are 0..63, each once
        int r = 63 - (int)(((long long)(GX_HOT / 4) * 64) / n);
        r = r < 1 ? 1 : (r > 62 ? 62 : r);
        long long pivot = 0;
        int above = 0;
        for (int attempt = 0; attempt < 8; ++attempt) {
            pivot = gs_shfl(mine, gs_first(gs_ballot(rank == r)));
            above = 0;
            for (int base = 0; base < n; base += 64) above += gs_popc(gs_ballot(base + lane < n && spill[base + lane] > pivot));
            if (above <= GX_HOT - 64 || r >= 62) break;
            r += (64 - r) / 2;
            if (r > 62) r = 62;
        }
        if (above == 0 || above > GX_HOT - 64) {   // no pivot separates a usable share: the single best key comes back
            long long best = GS_KEY_MIN;
            int bi = -1;
            for (int i = lane; i < n; i += 64) {
                const long long k = spill[i];
                if (k > best) {
                    best = k;
                    bi = i;
                }
            }
            const long long m = gs_wave_max(best);
            const int idx = (int)gs_shfl((long long)bi, gs_first(gs_ballot(bi >= 0 && best == m)));
            if (lane == 0) {
                hot_hi[0] = (int32_t)(m >> 32);
                hot_lo[0] = (uint32_t)(unsigned long long)m;
                spill[idx] = spill[n - 1];
            }
            hot_n = 1;
            spill_n = n - 1;
            spill_max = m;   // still an upper bound of what is left
            gs_fence();
            gs_barrier();
            return;
        }
        int nc = 0, ns = 0;
        for (int base = 0; base < n; base += 64) {
            const int i = base + lane;
            const bool in = i < n;
            const long long k = in ? spill[i] : 0;
            const bool hi = in && k > pivot;
            const bool lo = in && !hi;
            const uint64_t mh = gs_ballot(hi), ml = gs_ballot(lo);
            if (hi) {
                hot_hi[nc + gs_popc(mh & lt)] = (int32_t)(k >> 32);
                hot_lo[nc + gs_popc(mh & lt)] = (uint32_t)(unsigned long long)k;
            }
            if (lo) spill[ns + gs_popc(ml & lt)] = k;   // in place: target index <= i
            nc += gs_popc(mh);
            ns += gs_popc(ml);
            gs_barrier();
        }
        hot_n = nc;
        spill_n = ns;
        spill_max = pivot;
        gs_fence();
    };

    // ---- per-query setup: clear the visited set ----
    {
        const gs_u4 ones = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        if (has_v1) {
            gs_u4 *t4 = reinterpret_cast<gs_u4 *>(v1.w);
            for (int i = lane; i < (int)((2u << p.v1_log2) / 16u); i += 64) t4[i] = ones;
        } else {
            gs_u4 *v4 = reinterpret_cast<gs_u4 *>(vis);
            for (int i = lane; i < vcap / 4; i += 64) v4[i] = ones;
        }
    }
    gs_fence();
    gs_barrier();
    const float query_mag = (VSF == 2) ? p.bmag[q] : 0.0f;
    const unsigned long long *acc = p.accept ? p.accept + (long long)q * p.accept_stride : nullptr;
    const int32_t excl = p.exclude ? p.exclude[q] : -1;

    // ---- initializeInternal :334-353: mark and score the entry node ----
    {
        const int32_t e = p.entry_node;
        if (has_v1) {
            if (lane == 0) (void)gs_visit1(v1, e);
        } else {
            if (lane == 0) (void)gs_visit(vis, vmask, vshift, e);
            n2 = 1;
        }
        gs_u4 we[CH16];
        gs_load_row<CH16>(p.codes + (int64_t)e * p.M, we);
        const float sc = gs_finish<VSF>(gx_row_sum<VSF, CH16, FULL>(lut, p.wgx_lut_m, p.codebooks, reinterpret_cast<const float *>(lds), we),
                                        (VSF == 2) ? p.code_norms[e] : 0.0f, query_mag);
        const long long k = gs_key(e, sc);
        if (lane == 0) {
            hot_hi[0] = (int32_t)(k >> 32);
            hot_lo[0] = (uint32_t)(unsigned long long)k;
        }
        hot_n = 1;
        gs_barrier();
    }
    if (PROF) px[0] = GS_CLOCK() - pq0;

    for (int lvl = p.entry_level; lvl >= 0 && status == GS_OK; --lvl) {
        const int rk = lvl > 0 ? 1 : p.rerankK;
        if (lvl == 0) log_n = 0;
        // ---- searchOneLayer :406-457 ----
        for (;;) {
            if (hot_n == 0 && spill_n == 0) break;
            if (hot_n == 0) refill();
            if (PROF) pt = GS_CLOCK();
            // ---- candidates.top(): score words of the LDS tier, four per lane ----
            int32_t h0 = lane < hot_n ? hot_hi[lane] : GX_HI_NONE, h1 = lane + 64 < hot_n ? hot_hi[lane + 64] : GX_HI_NONE;
            int32_t h2 = lane + 128 < hot_n ? hot_hi[lane + 128] : GX_HI_NONE, h3 = lane + 192 < hot_n ? hot_hi[lane + 192] : GX_HI_NONE;
            const int32_t m01 = h0 > h1 ? h0 : h1, m23 = h2 > h3 ? h2 : h3;
            const int32_t H = gs_wave_max_i32(m01 > m23 ? m01 : m23);
            const uint64_t e0 = gs_ballot(h0 == H), e1 = gs_ballot(h1 == H), e2 = gs_ballot(h2 == H), e3 = gs_ballot(h3 == H);
            int idx;
            uint32_t L;
            if (gs_popc(e0) + gs_popc(e1) + gs_popc(e2) + gs_popc(e3) == 1) {
                idx = e0 ? gs_first(e0) : (e1 ? 64 + gs_first(e1) : (e2 ? 128 + gs_first(e2) : 192 + gs_first(e3)));
                L = (uint32_t)gs_uniform((int32_t)hot_lo[idx]);
            } else {   // several candidates with this score: the NodeQueue order goes on with the node word (the smaller node id wins)
                uint32_t l = 0;
                int li = -1;
                if (h0 == H) { l = hot_lo[lane]; li = lane; }
                if (h1 == H) { const uint32_t x = hot_lo[lane + 64]; if (li < 0 || x > l) { l = x; li = lane + 64; } }
                if (h2 == H) { const uint32_t x = hot_lo[lane + 128]; if (li < 0 || x > l) { l = x; li = lane + 128; } }
                if (h3 == H) { const uint32_t x = hot_lo[lane + 192]; if (li < 0 || x > l) { l = x; li = lane + 192; } }
                L = gs_wave_max_u32(li >= 0 ? l : 0u);
                idx = (int)gs_shfl((long long)li, gs_first(gs_ballot(li >= 0 && l == L)));
            }
            const long long top = gx_join(H, L);
            const float top_score = gx_hi_score(H);
            if (res_n >= rk && top_score < gs_key_score(res_min)) break;   // stopSearch :355-369
            // the best of what stays queued (by score word; which of several equal ones does not matter): the row to ask for next
            {
                const int pu = idx >> 6;
                if (lane == (idx & 63)) {
                    if (pu == 0) h0 = GX_HI_NONE;
                    else if (pu == 1) h1 = GX_HI_NONE;
                    else if (pu == 2) h2 = GX_HI_NONE;
                    else h3 = GX_HI_NONE;
                }
            }
            const int32_t n01 = h0 > h1 ? h0 : h1, n23 = h2 > h3 ? h2 : h3;
            const int32_t R = gs_wave_max_i32(n01 > n23 ? n01 : n23);
            long long runner_up = GS_KEY_MIN;
            if (R != GX_HI_NONE) {
                const uint64_t r0 = gs_ballot(h0 == R), r1 = gs_ballot(h1 == R), r2 = gs_ballot(h2 == R);
                const int ridx = r0 ? gs_first(r0) : (r1 ? 64 + gs_first(r1) : (r2 ? 128 + gs_first(r2) : 192 + gs_first(gs_ballot(h3 == R))));
                runner_up = gx_join(R, (uint32_t)gs_uniform((int32_t)hot_lo[ridx]));
            }
            // candidates.pop(): the last key takes the popped one's place
            if (lane == 0) {
                hot_hi[idx] = hot_hi[hot_n - 1];
                hot_lo[idx] = hot_lo[hot_n - 1];
            }
            hot_n--;
            const int32_t node = gs_key_node(top);
            int slot = slot_find(node, lvl);
            if (PROF) fh[slot < 0 ? 2 : 0] += 1;
            if (slot < 0) slot = slot_post(node, lvl, top, true);
            if (p.wgx_depth > 0 && runner_up != GS_KEY_MIN && !(res_n >= rk && gs_key_score(runner_up) < gs_key_score(res_min))) {
                const int32_t rn = gs_key_node(runner_up);
                if (slot_find(rn, lvl) < 0 && slot_post(rn, lvl, runner_up, false) >= 0 && PROF) fh[3] += 1;
            }
            GX_PHASE(0);
            // `topCandidateScore >= threshold` (:437, threshold 0: negative / NaN scores are expanded but never results), acceptOrds at
            // layer 0 only (:276); then addTopCandidate :515-530
            bool result = top_score >= 0.0f;
            if (result && lvl == 0 && acc) result = ((acc[node >> 6] >> (node & 63)) & 1ull) != 0;
            if (result && lvl == 0 && excl >= 0) result = node != excl;
            if (result && lvl == 0 && p.push_log) {   // the addTopCandidate sequence, for rt_body.h's tie resolution
                if (lane == 0) {
                    if (log_n < p.wgx_log) log_lds[log_n] = top;
                    else if (log_n < p.push_log_cap) p.push_log[(int64_t)q * p.push_log_cap + log_n] = top;
                }
                log_n++;
            }
            if (!result) {
            } else if (res_n < rk) {
                if (lane == 0) res[res_n] = top;
                if (top < res_min) {
                    res_min = top;
                    res_min_idx = res_n;
                }
                res_n++;
            } else if (top_score > gs_key_score(res_min)) {
                if (lvl > 0) {
                    if (ev_n >= evict_cap) {
                        status = GS_OVERFLOW;
                        break;
                    }
                    if (lane == 0) evicted[ev_n] = res_min;
                    ev_n++;
                }
                if (lane == 0) res[res_min_idx] = top;
                gs_barrier();
                res_min = gs_scan_extreme<false>(res, res_n, &res_min_idx);
            }
            n_expanded++;
            GX_PHASE(1);

            // ---- expand: the popped node's row, scored by an expander ----
            if (PROF && gs_ballot(lane == 0 && gs_lds_load(slot_state + slot) != GX_READY)) fh[1] += 1;
            while (!gs_ballot(lane == 0 && gs_lds_load(slot_state + slot) == GX_READY)) gs_spin_pause();
            const long long key = lane < p.wgx_kps ? slot_keys[slot * p.wgx_kps + lane] : GX_KEY_NONE;
            if (lane == slot) sl_node = -1;   // the slot is free again (its keys are in registers now)
            const bool fresh = visit(key != GX_KEY_NONE, gs_key_node(key));
            if (status != GS_OK) break;
            const uint64_t fm = gs_ballot(fresh);
            if (fm == 0) continue;
            n_visited += gs_popc(fm);
            GX_PHASE(2);
            // a fresh neighbour that beats everything queued is popped next: its row cannot be asked for any earlier than now
            if (p.wgx_depth > 0) {
                const int32_t fhi = gs_wave_max_i32(fresh ? (int32_t)(key >> 32) : GX_HI_NONE);
                if (fhi > (int32_t)(runner_up >> 32) && !(res_n >= rk && gx_hi_score(fhi) < gs_key_score(res_min))) {
                    const long long bf = gs_shfl(key, gs_first(gs_ballot(fresh && (int32_t)(key >> 32) == fhi)));
                    if (slot_post(gs_key_node(bf), lvl, bf, false) >= 0 && PROF) fh[3] += 1;
                }
            }
            GX_PHASE(3);
            push(key, fresh);
            GX_PHASE(4);
            if (status != GS_OK) break;
        }
        if (status != GS_OK) break;
        unsigned long long ptr0 = 0;
        if (PROF) ptr0 = GS_CLOCK();
        if (lvl > 0) {
            // rows requested for this level are of no use on the next one: let the expanders finish them, then free every slot
            while (gs_ballot(lane < p.wgx_slots && sl_node != -1 && gs_lds_load(slot_state + lane) != GX_READY)) gs_spin_pause();
            sl_node = -1;
            // setEntryPointsFromPreviousLayer :324-331
            for (int base = 0; base < res_n && status == GS_OK; base += 64) {
                const bool has = base + lane < res_n;
                push(has ? res[base + lane] : 0, has);
            }
            for (int base = 0; base < ev_n && status == GS_OK; base += 64) {
                const bool has = base + lane < ev_n;
                push(has ? evicted[base + lane] : 0, has);
            }
            res_n = 0;
            ev_n = 0;
            res_min = GS_KEY_MAX;
            res_min_idx = -1;
        }
        if (PROF) px[1] += GS_CLOCK() - ptr0;
    }
    unsigned long long pep0 = 0;
    if (PROF) pep0 = GS_CLOCK();
    if (lane == 0) gs_lds_store(hdr + GX_QUIT, 1);   // the expanders leave their service loop (gx_worker's barrier waits for them)

    // ---- hand the kept approximate results to the rerank stage ----
    gs_barrier();
    if (p.push_log && status == GS_OK) {
        int n = log_n < p.wgx_log ? log_n : p.wgx_log;
        n = n < p.push_log_cap ? n : p.push_log_cap;
        for (int i = lane; i < n; i += 64) p.push_log[(int64_t)q * p.push_log_cap + i] = log_lds[i];
    }
    for (int i = lane; i < p.rerankK; i += 64) {
        const bool have = status == GS_OK && i < res_n;
        const long long k = have ? res[i] : 0;
        p.out_ids[(int64_t)q * p.rerankK + i] = have ? gs_key_node(k) : -1;
        p.out_scores[(int64_t)q * p.rerankK + i] = have ? gs_key_score(k) : -__builtin_inff();
    }
    if (lane == 0) {
        p.out_stats[2 * (int64_t)q] = n_visited;
        p.out_stats[2 * (int64_t)q + 1] = n_expanded;
        p.out_status[q] = status;
        if (p.push_log) p.push_log_n[q] = status == GS_OK ? log_n : -1;
    }
    gs_barrier();
    if (PROF && p.prof && lane == 0) {
        unsigned long long in_loop = 0;
        for (int i = 0; i < 5; ++i) {
            gs_fetch_add64(p.prof + i, pf[i]);
            in_loop += pf[i];
        }
        gs_fetch_add64(p.prof + 5, (unsigned long long)n_expanded);
        gs_fetch_add64(p.prof + 6, 1ull);
        gs_fetch_add64(p.prof + 7, (GS_CLOCK() - pq0) - in_loop);
        for (int i = 0; i < 4; ++i) gs_fetch_add64(p.prof + 8 + i, fh[i]);
        gs_fetch_add64(p.prof + 12, px[0]);
        gs_fetch_add64(p.prof + 13, px[1]);
        gs_fetch_add64(p.prof + 14, GS_CLOCK() - pep0);
    }
#undef GX_PHASE
}

// Persistent workgroup: pulls queries off the shared counter until none are left.  Every wave of the workgroup calls it.
template <int VSF, int CH16, bool PROF = false, bool FULL = false>
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
        if (tid < 64) gx_control<VSF, CH16, PROF, FULL>(p, q, worker, lds);
        else gx_expander<VSF, CH16, FULL>(p, q, sh, tid & 63, lds);
        gs_block_barrier();   // nobody is inside this query's LDS state any more
    }
}

}  // namespace jv
