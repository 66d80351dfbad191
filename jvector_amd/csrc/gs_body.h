// gs_body.h — device-resident graph traversal: one 64-lane wavefront runs one query's whole GraphSearcher loop.
//
// This is the body of graph_search_kernel (k_gsearch.hip).  It is written against a tiny wave API so that the very
// same source also compiles as plain C++ for the lane-level emulator the CPU tests use (tests/emu/): the includer
// defines
//     GS_FN                                   function qualifier (__device__ __forceinline__ / inline)
//     gs_lane()                               0..63
//     gs_barrier()                            block barrier (the block IS one wavefront)
//     gs_ballot(bool) -> uint64_t             wave vote
//     gs_shfl(long long v, int src)           read lane src's v
//     gs_shfl32(int32_t v, int src)           the same for 32 bits (one ds_bpermute_b32)
//     gs_perm(uint32_t hi, uint32_t lo, uint32_t sel)   v_perm_b32: result byte i = byte sel.byte[i] of the eight bytes {hi : lo}
//                                             (0..3 = lo, 4..7 = hi), 0x0c = the constant 0
//     gs_shfl_xor(long long v, int laneMask)  butterfly exchange
//     GS_OPAQUE_I32(x)                        optimisation barrier on an int (no-op on the emulator)
//     gs_cas(int32_t *p, int32_t expect, int32_t desired) -> old      (device-scope atomic)
//     gs_fetch_add(uint32_t *p, uint32_t v) -> old                    (device-scope atomic)
//     gs_fetch_add64(unsigned long long *p, unsigned long long v)     (device-scope atomic; profiling aid only)
//     gs_lds_cas(uint32_t *p, uint32_t expect, uint32_t desired) -> old    (LDS atomic, workgroup scope)
//     gs_prefetch_lds(const void *g, void *lds)  speculative touch: every active lane starts a 4-byte load of ITS address g whose
//                                             result lands (whenever) at lds + 4 * lane and is never read — no register waits for it
//     gs_fence()                              device-scope memory fence
//     gs_sqrt(double)
//     gs_rsq_approx(float)                    1 / sqrt(x) to a few ulps (v_rsq_f32); only ever used with a margin around it
//     GS_SCHED_FENCE()                        instruction-scheduling fence (may be empty)
// and, for the workgroup form (WGX, gx_body.h — several wavefronts per query; gs_barrier() is then a WAVE-scope sync point):
//     gs_tid()                                thread index inside the workgroup
//     gs_block_barrier()                      workgroup barrier
//     gs_lds_load(const int32_t *p)           LDS flag read, acquire at workgroup scope
//     gs_lds_store(int32_t *p, int32_t v)     LDS flag write, release at workgroup scope
//     gs_lds_add(int32_t *p, int32_t v) -> old   LDS atomic add
//     gs_spin_pause()                         inside a spin-wait on an LDS flag (s_sleep / a scheduling point of the emulator)
//
// What it computes is GraphSearcher.search for threshold 0 / acceptOrds ALL (B/graph/GraphSearcher.java:222-243,
// 263-282 internalSearch, 334-353 initializeInternal, 355-369 stopSearch, 406-457 searchOneLayer, 324-331
// setEntryPointsFromPreviousLayer, 515-530 addTopCandidate) with the approximate score function of PQDecoder /
// FusedPQDecoder (B/quantization/PQDecoder.java:65-80,124-135, FusedPQDecoder.java:104-111,206-213): the same
// per-query state machine the host searcher (graph_search.cpp) runs, so results, visitedCount and expandedCount
// are identical.  The three queues only ever need "max / min under the NodeQueue key order" and set membership, and
// NodeQueue keys (sortableInt(score) << 32 | ~node, NodeQueue.java:125-129) are unique per query, so they are
// kept as UNSORTED arrays that the wave scans in parallel instead of heaps a single lane would have to sift:
//   candidates  (NodeQueue MAX_HEAP, unbounded)   LDS tier of cand_cap keys + a global spill tier.  Invariant:
//               every LDS key > spill_max >= every spilled key, so the LDS maximum is the global maximum.  When
//               the LDS tier would overflow, keys <= the median of 64 samples move to the spill tier.  A search
//               stops long before it has to pop from the spill tier; that path exists but is a plain scan.
//   results     (BoundedLongHeap(rerankK), MIN_HEAP)  LDS array + cached minimum; replace-min = rescan.
//   evicted     (upper layers, rk = 1)               LDS array.
//   visited     two-tier open-addressing set, the <= 64 neighbours of an expansion inserted by their lanes concurrently
//               with CAS.  Tier 1: 16-bit entries in LDS (gs_visit1).  Tier 2: node ids in global memory (one table per
//               worker), touched — and only then cleared — by the queries whose tier 1 fills up (or by all of them when
//               the launch has no LDS tier, GsParams::v1_log2 == 0).
// Scores: table-free ADC (k_frontier.hip's direct variant): lane i recomputes the look-up entries its neighbour's
// code selects from the L2-resident codebook, summing in ascending m into one f32 — bit-identical to
// assembleAndSum / pqDecodedCosineSimilarity on the decoder's tables (uniform 8-dim sub-vectors only).
// Anything that does not fit the fixed-size structures (hash table half full, spill tier or evicted list full)
// marks the query GS_OVERFLOW; the host re-runs such queries through the host searcher.
#pragma once

#include <cstdint>

#include "gs_params.h"

#ifndef GS_UBR_VAR_LPS
#define GS_UBR_VAR_LPS 1   // 32 / 16 / 8 lanes per survivor by survivor count (0: always 8; bit-identical; 47.8 vs 48.8 ms, profiles/r6_h)
#endif

namespace jv {

struct alignas(16) gs_f4 { float x, y, z, w; };
struct alignas(16) gs_u4 { uint32_t x, y, z, w; };

GS_FN int32_t gs_float_bits(float f)
{
    int32_t b;
    __builtin_memcpy(&b, &f, 4);
    return b;
}
GS_FN float gs_bits_float(int32_t b)
{
    float f;
    __builtin_memcpy(&f, &b, 4);
    return f;
}
// NodeQueue.encode (NodeQueue.java:125-129) with NumericUtils.floatToSortableInt (:49-65)
GS_FN long long gs_key(int32_t node, float score)
{
    const int32_t bits = (score != score) ? 0x7fc00000 : gs_float_bits(score);
    const int32_t s = bits ^ ((bits >> 31) & 0x7fffffff);
    return (long long)((((unsigned long long)(uint32_t)s) << 32) | (unsigned long long)(uint32_t)(~node));
}
GS_FN int32_t gs_key_node(long long k) { return (int32_t)~(uint32_t)((unsigned long long)k & 0xFFFFFFFFull); }
GS_FN float gs_key_score(long long k)
{
    const int32_t e = (int32_t)(k >> 32);
    return gs_bits_float(e ^ ((e >> 31) & 0x7fffffff));
}

constexpr long long GS_KEY_MIN = (long long)0x8000000000000000ull;
constexpr long long GS_KEY_MAX = (long long)0x7fffffffffffffffull;
// "no neighbour in this lane" in a scored row of the workgroup form: a real key's low word is ~node with node >= 0, never 0
constexpr long long GX_KEY_NONE = 0;

#ifndef GS_HAVE_WAVE_REDUCE   // (gs_wave_hip.h supplies DPP forms under GS_UNIFORM_SHFL)
GS_FN long long gs_wave_max(long long v)
{
    for (int o = 32; o > 0; o >>= 1) {
        const long long t = gs_shfl_xor(v, o);
        v = t > v ? t : v;
    }
    return v;
}
GS_FN long long gs_wave_min(long long v)
{
    for (int o = 32; o > 0; o >>= 1) {
        const long long t = gs_shfl_xor(v, o);
        v = t < v ? t : v;
    }
    return v;
}
#endif
GS_FN int gs_popc(uint64_t m) { return __builtin_popcountll(m); }
GS_FN int gs_first(uint64_t m) { return m ? __builtin_ctzll(m) : 64; }

// max (or min) key of a[0..n) and its index; n > 0; result uniform across the wave
template <bool MAX>
GS_FN long long gs_scan_extreme(const long long *a, int n, int *idx_out)
{
    const int lane = gs_lane();
    long long best = MAX ? GS_KEY_MIN : GS_KEY_MAX;
    int bi = -1;
    for (int i = lane; i < n; i += 256) {   // four keys per lane and pass, read before any of them is compared
        long long k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) k[u] = (i + 64 * u < n) ? a[i + 64 * u] : (MAX ? GS_KEY_MIN : GS_KEY_MAX);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MAX ? (k[u] > best) : (k[u] < best)) {
                best = k[u];
                bi = i + 64 * u;
            }
        }
    }
    const long long m = MAX ? gs_wave_max(best) : gs_wave_min(best);
    const uint64_t who = gs_ballot(bi >= 0 && best == m);
    *idx_out = (int)gs_shfl((long long)bi, gs_first(who));
    return m;
}

// max key of a[0..n), its index, and the runner-up key (GS_KEY_MIN when n < 2); results uniform across the wave
GS_FN long long gs_scan_top2(const long long *a, int n, int *idx_out, long long *second_out)
{
    const int lane = gs_lane();
    long long best = GS_KEY_MIN, second = GS_KEY_MIN;
    int bi = -1;
    // four keys per lane and pass, read before any of them is compared (one LDS round trip per 256 keys)
    for (int i = lane; i < n; i += 256) {
        long long k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) k[u] = (i + 64 * u < n) ? a[i + 64 * u] : GS_KEY_MIN;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (k[u] > best) {
                second = best;
                best = k[u];
                bi = i + 64 * u;
            } else if (k[u] > second) {
                second = k[u];
            }
        }
    }
    const long long m = gs_wave_max(best);
    const int w = gs_first(gs_ballot(bi >= 0 && best == m));  // keys are unique: exactly one lane holds the maximum
    *idx_out = (int)gs_shfl((long long)bi, w);
    *second_out = gs_wave_max(lane == w ? second : best);
    return m;
}

// ---- table-free ADC row score: DefaultVectorUtilSupport.calculatePartialSums (:351-365) entry by entry, summed in
//      ascending m (assembleAndSum :323-330).  qs = the centred query in LDS.  Same arithmetic as
//      frontier_direct_kernel; -ffp-contract=off keeps every mul and add separate.
// the M code bytes of one row as CH16 16-byte words (issued early: their latency hides behind the visited-set probes)
template <int CH16>
GS_FN void gs_load_row(const uint8_t *rp, gs_u4 (&w)[CH16])
{
    const gs_u4 *r4 = reinterpret_cast<const gs_u4 *>(rp);
#pragma unroll
    for (int c = 0; c < CH16; ++c) w[c] = r4[c];
}

template <int VSF, int CH16>
GS_FN float gs_row_sum(const float *codebooks, const float *qs, const gs_u4 (&w)[CH16])
{
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < CH16; ++c) {
        const uint32_t d[4] = {w[c].x, w[c].y, w[c].z, w[c].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int m = c * 16 + e * 4 + b;
                const uint32_t code = (d[e] >> (8 * b)) & 0xFFu;
                const gs_f4 *cp = reinterpret_cast<const gs_f4 *>(codebooks + ((int64_t)(m * 256) + code) * 8);
                const gs_f4 c0 = cp[0], c1 = cp[1];
                const float *q = qs + m * 8;
                float ent = 0.0f;
                if (VSF == 0 /* L2 */) {
                    float t;
                    t = c0.x - q[0]; ent += t * t;
                    t = c0.y - q[1]; ent += t * t;
                    t = c0.z - q[2]; ent += t * t;
                    t = c0.w - q[3]; ent += t * t;
                    t = c1.x - q[4]; ent += t * t;
                    t = c1.y - q[5]; ent += t * t;
                    t = c1.z - q[6]; ent += t * t;
                    t = c1.w - q[7]; ent += t * t;
                } else {
                    ent += c0.x * q[0];
                    ent += c0.y * q[1];
                    ent += c0.z * q[2];
                    ent += c0.w * q[3];
                    ent += c1.x * q[4];
                    ent += c1.y * q[5];
                    ent += c1.z * q[6];
                    ent += c1.w * q[7];
                }
                sum += ent;
            }
        }
    }
    return sum;
}

// ---- the same score for ANY product quantizer (CH16 = 0 kernels: ragged sub-vectors, sizes other than 8, M not a multiple of 16):
//      entry (m, code) = calculatePartialSums' chain over the sub-vector's own length (j ascending, mul and add separate), entries
//      summed in ascending m — what lut_build_kernel (k_pq.hip) writes into the tables of the flat path, so the bits agree with
//      every other scoring form.  The code bytes are read one at a time straight from the row (no alignment to rely on).
template <int VSF>
GS_FN float gs_row_sum_any(const GsParams &p, const float *qs, const uint8_t *rp)
{
    float sum = 0.0f;
    const int M = p.M;
    if (p.sub_uniform4 > 0) {  // uniform sizes, a multiple of 4: codebook rows are 16-byte aligned
        const int S = p.sub_uniform4;
#pragma unroll 4
        for (int m = 0; m < M; ++m) {
            const gs_f4 *cp = reinterpret_cast<const gs_f4 *>(p.codebooks + ((int64_t)m * 256 + rp[m]) * S);
            const float *q = qs + m * S;
            float ent = 0.0f;
            for (int j = 0; j < S; j += 4) {
                const gs_f4 c = cp[j >> 2];
                if (VSF == 0 /* L2 */) {
                    float t;
                    t = c.x - q[j]; ent += t * t;
                    t = c.y - q[j + 1]; ent += t * t;
                    t = c.z - q[j + 2]; ent += t * t;
                    t = c.w - q[j + 3]; ent += t * t;
                } else {
                    ent += c.x * q[j];
                    ent += c.y * q[j + 1];
                    ent += c.z * q[j + 2];
                    ent += c.w * q[j + 3];
                }
            }
            sum += ent;
        }
        return sum;
    }
#pragma unroll 2
    for (int m = 0; m < M; ++m) {
        const int S = p.sub_sizes[m];
        const float *cb = p.codebooks + p.cb_offsets[m] + (int64_t)rp[m] * S;
        const float *q = qs + p.sub_offsets[m];
        float ent = 0.0f;
        for (int j = 0; j < S; ++j) {
            if (VSF == 0 /* L2 */) {
                const float t = cb[j] - q[j];
                ent += t * t;
            } else {
                ent += cb[j] * q[j];
            }
        }
        sum += ent;
    }
    return sum;
}

template <int VSF>
GS_FN float gs_lut_entry_from(const gs_f4 c0, const gs_f4 c1, const float *q);
template <int VSF>
GS_FN float gs_lut_entry(const float *codebooks, const float *qs, int m, int code)
{
    const gs_f4 *cp = reinterpret_cast<const gs_f4 *>(codebooks + ((int64_t)(m * 256) + code) * 8);
    return gs_lut_entry_from<VSF>(cp[0], cp[1], qs + m * 8);
}
// one table entry from its codebook row (c0, c1) and the query's sub-vector q: calculatePartialSums' chain
template <int VSF>
GS_FN float gs_lut_entry_from(const gs_f4 c0, const gs_f4 c1, const float *q)
{
    float ent = 0.0f;
    if (VSF == 0 /* L2 */) {
        float t;
        t = c0.x - q[0]; ent += t * t;
        t = c0.y - q[1]; ent += t * t;
        t = c0.z - q[2]; ent += t * t;
        t = c0.w - q[3]; ent += t * t;
        t = c1.x - q[4]; ent += t * t;
        t = c1.y - q[5]; ent += t * t;
        t = c1.z - q[6]; ent += t * t;
        t = c1.w - q[7]; ent += t * t;
    } else {
        ent += c0.x * q[0];
        ent += c0.y * q[1];
        ent += c0.z * q[2];
        ent += c0.w * q[3];
        ent += c1.x * q[4];
        ent += c1.y * q[5];
        ent += c1.z * q[6];
        ent += c1.w * q[7];
    }
    return ent;
}

// one table entry with its eight products formed two at a time: v_pk_mul_f32 on the register pairs the 16-byte loads delivered (the
// row's (x, y) / (z, w) against the query's — no operand has to be moved), then the reference's chain of eight additions in ascending
// dimension.  Same IEEE operations, same order, same bits as gs_lut_entry_from; 12 instructions instead of 16.  (Running TWO entries'
// chains in the halves of packed operations costs more than it saves: the operand pairs would have to be assembled with moves — 144
// v_mov for 96 packed operations in the first version of the UBR scoring loop.)
template <int VSF>
GS_FN float gs_lut_entry_pk(const gs_f4 c0, const gs_f4 c1, const float *q)
{
#ifdef GS_HAVE_PK_F32
    if (VSF != 0) {
        const gs_f4 q0 = reinterpret_cast<const gs_f4 *>(q)[0], q1 = reinterpret_cast<const gs_f4 *>(q)[1];
        const gs_pk2 p01 = gs_pk2{c0.x, c0.y} * gs_pk2{q0.x, q0.y}, p23 = gs_pk2{c0.z, c0.w} * gs_pk2{q0.z, q0.w};
        const gs_pk2 p45 = gs_pk2{c1.x, c1.y} * gs_pk2{q1.x, q1.y}, p67 = gs_pk2{c1.z, c1.w} * gs_pk2{q1.z, q1.w};
        float ent = 0.0f;
        ent += p01.x;
        ent += p01.y;
        ent += p23.x;
        ent += p23.y;
        ent += p45.x;
        ent += p45.y;
        ent += p67.x;
        ent += p67.y;
        return ent;
    }
#endif
    return gs_lut_entry_from<VSF>(c0, c1, q);
}

// the same with the query's sub-vector already in registers (q0, q1): callers that batch their LDS reads (UBR's scoring loop)
template <int VSF>
GS_FN float gs_lut_entry_pk_q(const gs_f4 c0, const gs_f4 c1, const gs_f4 q0, const gs_f4 q1)
{
#ifdef GS_HAVE_PK_F32
    if (VSF != 0) {
        const gs_pk2 p01 = gs_pk2{c0.x, c0.y} * gs_pk2{q0.x, q0.y}, p23 = gs_pk2{c0.z, c0.w} * gs_pk2{q0.z, q0.w};
        const gs_pk2 p45 = gs_pk2{c1.x, c1.y} * gs_pk2{q1.x, q1.y}, p67 = gs_pk2{c1.z, c1.w} * gs_pk2{q1.z, q1.w};
        float ent = 0.0f;
        ent += p01.x;
        ent += p01.y;
        ent += p23.x;
        ent += p23.y;
        ent += p45.x;
        ent += p45.y;
        ent += p67.x;
        ent += p67.y;
        return ent;
    }
#endif
    const float q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
    return gs_lut_entry_from<VSF>(c0, c1, q);
}

// ---- the workgroup form's score: the query's whole ADC table [M][256] f32 sits in LDS (gx_body.h gx_lut_build wrote it with
//      gs_lut_entry's arithmetic: calculatePartialSums entry by entry); a row's score is assembleAndSum (:323-330): the entries
//      its code bytes select, added in ascending m into one f32.
//      The table may cover only the subspaces [0, lut_m) (a multiple of 16): the entries of the others are recomputed from the
//      codebook like gs_row_sum does — the same bits, in the same ascending-m sum.
template <int VSF, int CH16, bool FULL = false>   // FULL: the table covers every subspace (lut_m == M, no table-free code at all)
GS_FN float gx_row_sum(const float *lut, int lut_m, const float *codebooks, const float *qs, const gs_u4 (&w)[CH16])
{
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < CH16; ++c) {
        const uint32_t d[4] = {w[c].x, w[c].y, w[c].z, w[c].w};
        if (FULL || c * 16 < lut_m) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int m = c * 16 + e * 4 + b;
                    const uint32_t code = (d[e] >> (8 * b)) & 0xFFu;
                    sum += lut[m * 256 + (int)code];
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int m = c * 16 + e * 4 + b;
                    const uint32_t code = (d[e] >> (8 * b)) & 0xFFu;
                    sum += gs_lut_entry<VSF>(codebooks, qs, m, (int)code);
                }
            }
        }
    }
    return sum;
}

// ---- pair-lane scoring (maxDegree <= 32 leaves half the wave idle): lanes i and i + 32 share neighbour i.  Each
//      computes the table entries of HALF the subspaces (entries are independent of one another); the low lane then forms
//      the running sum in ascending m — its own entries first, the partner's after — so the f32 result is unchanged.
struct alignas(8) gs_u2 { uint32_t x, y; };

template <int HW>  // HW 8-byte words = M / 2 code bytes
GS_FN void gs_load_half(const uint8_t *rp, gs_u2 (&w)[HW])
{
    const gs_u2 *r2 = reinterpret_cast<const gs_u2 *>(rp);
#pragma unroll
    for (int c = 0; c < HW; ++c) w[c] = r2[c];
}

// Entries of this lane's half of the subspaces.  Low lane (xw == nullptr): returns their running sum (ascending m).
// High lane: writes entry j to xw[j * XS] (LDS exchange area, column = neighbour) for its partner and returns 0.
template <int VSF, int HW, int XS = 32>
GS_FN float gs_half_entries(const float *codebooks, const float *qs, const gs_u2 (&w)[HW], int m_base, float *xw)
{
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < HW; ++c) {
        const uint32_t d[2] = {w[c].x, w[c].y};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int j = c * 8 + e * 4 + b;
                const int m = m_base + j;
                const uint32_t code = (d[e] >> (8 * b)) & 0xFFu;
                const gs_f4 *cp = reinterpret_cast<const gs_f4 *>(codebooks + ((int64_t)m * 256 + code) * 8);
                const gs_f4 c0 = cp[0], c1 = cp[1];
                const float *q = qs + m * 8;
                float v = 0.0f;
                if (VSF == 0 /* L2 */) {
                    float t;
                    t = c0.x - q[0]; v += t * t;
                    t = c0.y - q[1]; v += t * t;
                    t = c0.z - q[2]; v += t * t;
                    t = c0.w - q[3]; v += t * t;
                    t = c1.x - q[4]; v += t * t;
                    t = c1.y - q[5]; v += t * t;
                    t = c1.z - q[6]; v += t * t;
                    t = c1.w - q[7]; v += t * t;
                } else {
                    v += c0.x * q[0];
                    v += c0.y * q[1];
                    v += c0.z * q[2];
                    v += c0.w * q[3];
                    v += c1.x * q[4];
                    v += c1.y * q[5];
                    v += c1.z * q[6];
                    v += c1.w * q[7];
                }
                if (xw) xw[j * XS] = v;
                else sum += v;
            }
        }
    }
    return sum;
}

// ---- why a bound may drop a neighbour (the argument UBR rests on; round 4's UB8 form introduced it) ------------------------------
// Most neighbours a search scores are never popped: a node is only expanded if, at its turn, fewer than rerankK nodes with a
// strictly greater score have been popped.  So once >= rerankK nodes with exact score >= T are known (queued or popped, all of them
// able to become results: scores >= 0, no NaN, no acceptOrds filter), a neighbour whose score is < T can never be expanded — the
// reference would mark it visited, score it, push it, and never look at it again.  Its exact score (M codebook gathers) is therefore
// not needed IF a cheap rigorous upper bound already shows score < T: the table holds, per (subspace, code), the entry's upper
// bucket edge in 8 bits (ub = lo_m + S * (b + 1) >= entry), a row's bound is base + S * sum_m (b_m + 1) (base carries a slack that
// covers every rounding of the reference's f32 chain and of this sum), and the finishing transform is monotone.  Dropped neighbours
// are counted in visitedCount (they were marked) and are simply not pushed: results, scores, visitedCount and expandedCount are
// unchanged.  (UB8 — the table built by the traversal wave itself, 24 KB of LDS per wave, survivors scored in place — was 2.3x
// slower than the plain pair form and left the source in round 6; its measurements: profiles/r4_o ... r4_zz, scripts/ub8_study.py.)

// ---- UBR: the upper-bound form (round 5; it finishes what round 4's UB8 started) -------------------------------------------------------------
// UB8 was sound and dropped 60 % of the scored neighbours, and was 2.3x slower: its table was built by the traversal wave itself
// (355 k clocks per query), took 24 KB of LDS per wave (4 waves per CU, no room for the visited set's LDS tier) and the neighbours it
// kept were scored in place (dead lanes free no gather instructions).  UBR removes the three:
//   * the tables of the whole batch are PREBUILT by a dense kernel (k_gsearch_ubr.hip ubr_table_kernel; the same arithmetic as
//     gs_ubr_build_ref in gs_host.h) with ONE scale per query: entry (m, c) -> b = bucket of (entry - lo_m) / S, upper edge
//     lo_m + S (b + 1) >= entry checked in f32, so a row's bound is  base + S * sum_m (b_m + 1)  — an integer sum;
//   * a query's table lives in the wave's REGISTERS, M dwords per lane: for step r < M/2 register 2r of lane s holds the bytes
//     {t[r][s], t[r][s+64], t[r+M/2][s], t[r+M/2][s+64]}, register 2r+1 the same for codes s+128 / s+192.  In the pair-lane
//     arrangement (low lane: subspaces [0, M/2), high lane: [M/2, M) of the same neighbour) both lanes look up step r at once: two
//     ds_bpermute_b32 (source lane = code & 63) + a byte select.  No LDS bytes: 8 waves per CU and the LDS tier stay;
//   * the neighbours the bound cannot drop are COMPACTED: their code bytes go to LDS and eight lanes score each of them (lane t of a
//     group: subspaces [t M/8, (t + 1) M/8); lane 0 adds all M entries in ascending m, the others hand theirs over through LDS) —
//     a pass of 8 survivors costs 2 M/8 gather instructions instead of the pair form's M;
//   * the threshold: T = the score of a candidate such that >= rerankK known nodes (queued or kept results) score strictly higher —
//     found every `ubr_trim` pushes among 64 samples of the candidate tier by bisection on exact counts — and the candidates at or
//     below it are DISCARDED with it (they can never be popped either), so the tier stays around rerankK keys and pops scan little.
// Soundness: see above; every parity test of the pair-lane traversal runs with this form as well.
template <int CH16>
GS_FN void gs_ubr_load(const uint32_t *tab_q, uint32_t (&tab)[CH16 * 16])
{
    const gs_u4 *src = reinterpret_cast<const gs_u4 *>(tab_q) + gs_lane();
#pragma unroll
    for (int k = 0; k < CH16 * 4; ++k) {
        const gs_u4 v = src[k * 64];
        tab[4 * k + 0] = v.x;
        tab[4 * k + 1] = v.y;
        tab[4 * k + 2] = v.z;
        tab[4 * k + 3] = v.w;
    }
}

// this lane's half of sum_m (b_m + 1): the low lane of a pair (hi = 0) holds the codes of subspaces [0, M/2) in w, the high lane
// (hi = 1) those of [M/2, M).  EVERY lane of the wave must execute it (a ds_bpermute reads the registers of active lanes only).
template <int CH16>
GS_FN int32_t gs_ubr_half(const uint32_t (&tab)[CH16 * 16], const gs_u2 (&w)[CH16], int hi, bool lower = false)
{
    int32_t acc = lower ? 0 : CH16 * 8;   // the "+ 1" of every bucket (upper edges; the euclidean form sums LOWER edges)
    // byte selector of gs_perm: the looked-up pair {upper word (codes >= 128) : lower word} holds the entry's byte at index
    // (code >> 6 & 1) + 4 (code >> 7) + 2 hi; the other three result bytes are the constant 0
    const uint32_t selbase = 0x0c0c0c00u + 2u * (uint32_t)hi;
#pragma unroll
    for (int c = 0; c < CH16; ++c) {
        // the eight look-ups of one code word: all sixteen cross-lane reads are issued before the first result is used
        const uint32_t d[2] = {w[c].x, w[c].y};
        uint32_t lo[8], up[8], sel[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = c * 8 + i;
            const uint32_t code = (d[i >> 2] >> (8 * (i & 3))) & 0xFFu;
            const int src = (int)(code & 63u);
            lo[i] = (uint32_t)gs_shfl32((int32_t)tab[2 * r], src);
            up[i] = (uint32_t)gs_shfl32((int32_t)tab[2 * r + 1], src);
            const uint32_t t2 = code >> 6;
            sel[i] = selbase + t2 + (t2 & 2u);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += (int32_t)gs_perm(up[i], lo[i], sel[i]);
    }
    return acc;
}

// raw table sum -> similarity (jv_device.h score_from_raw / cosine_finish; PQDecoder.java:68,79,126)
template <int VSF>
GS_FN float gs_finish(float sum, float node_mag, float query_mag)
{
    if (VSF == 0) return 1.0f / (1.0f + sum);
    if (VSF == 2) {
        const float prod = node_mag * query_mag;
        sum = (float)((double)sum / gs_sqrt((double)prod));
    }
    return (1.0f + sum) / 2.0f;
}

// UBR's drop test: is gs_finish(bound_raw, ...) CERTAINLY below the pop threshold T?  For the dot product that is the finish itself.
// For cosine the finish divides by a double-precision square root (45 f64 instructions executed by all 64 lanes per expansion);
// the test needs no such care — a neighbour wrongly KEPT is only scored exactly, never lost: x = raw * rsq(|a||b|) in f32 (a few
// ulps off the real quotient, < 1e-6 relative) against 2 T - 1 with a margin of 1e-5 (1 + |x|), twenty times those errors and the
// finish's own three roundings together; a product that is not a positive finite number keeps the neighbour.  The bound moves by
// 1e-5 of a score: the drop rate does not (tests/test_zz_ubr_gpu.py counts it).
template <int VSF>
GS_FN bool gs_bound_below(float bound_raw, float node_mag, float query_mag, float T)
{
    if (VSF == 0) return bound_raw > 0.0f && gs_finish<VSF>(bound_raw, node_mag, query_mag) < T;   // (a LOWER bound of the distance: 1 / (1 + d) falls)
    if (VSF != 2) return gs_finish<VSF>(bound_raw, node_mag, query_mag) < T;
    const float prod = node_mag * query_mag;
    if (!(prod > 0.0f) || !(prod < 3.0e38f)) return false;
    const float x = bound_raw * gs_rsq_approx(prod);
    const float ax = x < 0.0f ? -x : x;
    return x + 1e-5f * (1.0f + ax) < 2.0f * T - 1.0f;
}

// An explicit upper bound U of the finished score behind a raw bound (DEFER): gs_bound_below(bound_raw, ..., U) is re-checked by the
// caller, so that "exact score < U" rests on the very predicate the level-0 drops rest on.  +inf = no bound.
template <int VSF>
GS_FN float gs_bound_score(float bound_raw, float node_mag, float query_mag)
{
    float u;
    if (VSF == 2) {
        const float prod = node_mag * query_mag;
        if (!(prod > 0.0f) || !(prod < 3.0e38f)) return __builtin_inff();
        const float x = bound_raw * gs_rsq_approx(prod);
        const float ax = x < 0.0f ? -x : x;
        u = 0.5f * (1.0f + (x + 1e-5f * (1.0f + ax)));
    } else {
        u = gs_finish<VSF>(bound_raw, node_mag, query_mag);
    }
    const float au = u < 0.0f ? -u : u;
    return u + 1e-6f * (1.0f + au);
}

// UBR: one scoring pass over the survivors [base, base + 64 / LPS) of an expansion, LPS lanes each (round 6: 8, 16 or 32 — picked by
// how many survived; two thirds of the headline's expansions keep at most FOUR of their ~21 fresh neighbours, and eight lanes each
// left half the wave idle while every lane walked twelve subspaces in two rounds of codebook requests).  Lane LPS g + t takes the
// subspaces [t SUBS, (t + 1) SUBS) of survivor base + g, SUBS = M / LPS: its code bytes from the staging area, its codebook rows in
// rounds of at most six (requested before the first entry is formed), the query's sub-vectors from LDS up to three entries at a time;
// lane t = 0 (the owner) adds its own entries, then — behind the barrier — the other lanes' in ascending m from its column of the
// hand-over area: assembleAndSum's order, the same bits whatever LPS is.  Returns the owner lanes' keys (fresh = the lane holds one
// that is to be pushed).  Wave-uniform call; xf: [64 / LPS columns][column stride] floats, columns 16-byte aligned.
template <int VSF, int M_, int LPS>
GS_FN void gs_ubr_pass(const float *codebooks, const float *qs, float *xf, const int32_t *st_nb, const float *st_mag, const uint8_t *st_code,
                       int base, int ns, float query_mag, bool ub_active, float ub_T, bool &fresh, long long &key)
{
    static_assert(M_ % LPS == 0 && (LPS == 8 || LPS == 16 || LPS == 32), "lanes per survivor");
    constexpr int SUBS = M_ / LPS;                      // subspaces per lane
    constexpr int RH = SUBS % 6 == 0 ? 6 : (SUBS % 4 == 0 ? 4 : SUBS);   // codebook rows per round (PQ-96: 6, 6, 3)
    constexpr int QB = RH % 3 == 0 ? 3 : (RH % 2 == 0 ? 2 : 1);          // entries per batch of query reads
    static_assert(SUBS % RH == 0 && RH % QB == 0 && RH <= 6, "whole rounds of whole batches");
    constexpr int NB = (LPS - 1) * SUBS;                // entries the owner takes over
    constexpr int CS = (NB + 3) & ~3;                   // column stride (floats)
    const int lane = gs_lane();
    const int g = lane / LPS;
    int t = lane % LPS;
    GS_OPAQUE_I32(t);   // (or the per-lane subspace pointers are hoisted out of the search loop and spilled)
    const int j = base + g;
    const bool work = j < ns;
    float sum = 0.0f;
    if (work) {
        // the lane's SUBS code bytes: an aligned 8-byte window (12-byte for SUBS = 12) of the survivor's staged row, shifted into place
        const int off = t * SUBS;
        const uint32_t *cw = reinterpret_cast<const uint32_t *>(st_code + j * M_ + (off & ~3));
        uint32_t d[SUBS / 4 > 3 ? SUBS / 4 : 3];
        if (SUBS % 4 == 0) {
#pragma unroll
            for (int i = 0; i < SUBS / 4; ++i) d[i] = cw[i];
        } else {
            const unsigned long long win = (unsigned long long)cw[0] | ((unsigned long long)cw[1] << 32);
            const unsigned long long sh = win >> (8 * (off & 3));
            d[0] = (uint32_t)sh;
            d[1] = (uint32_t)(sh >> 32);
            d[2] = 0u;
        }
        float v[SUBS];
#pragma unroll
        for (int h2 = 0; h2 < SUBS / RH; ++h2) {
            gs_f4 c0[RH], c1[RH];
#pragma unroll
            for (int kk = 0; kk < RH; ++kk) {
                const int k = h2 * RH + kk;
                const uint32_t code = (d[k >> 2] >> (8 * (k & 3))) & 0xFFu;
                const gs_f4 *cp = reinterpret_cast<const gs_f4 *>(codebooks + ((int64_t)((t * SUBS + k) * 256) + code) * 8);
                c0[kk] = cp[0];
                c1[kk] = cp[1];
            }
            // the query's sub-vectors are read from LDS three entries at a time behind the codebook requests: left to itself the
            // compiler — at its 255-register limit — read them entry by entry and waited for each pair in turn (all six of a round at
            // once cost 34 spilled table registers, reloaded in front of every bound phase)
#pragma unroll
            for (int b3 = 0; b3 < RH / QB; ++b3) {
                gs_f4 q0[QB], q1[QB];
#pragma unroll
                for (int kk = 0; kk < QB; ++kk) {
                    const gs_f4 *qp = reinterpret_cast<const gs_f4 *>(qs + (t * SUBS + h2 * RH + b3 * QB + kk) * 8);
                    q0[kk] = qp[0];
                    q1[kk] = qp[1];
                }
                GS_SCHED_FENCE();
#pragma unroll
                for (int kk = 0; kk < QB; ++kk) {
                    const int k = h2 * RH + b3 * QB + kk;
                    v[k] = gs_lut_entry_pk_q<VSF>(c0[b3 * QB + kk], c1[b3 * QB + kk], q0[kk], q1[kk]);
                }
                GS_SCHED_FENCE();
            }
        }
        if (t == 0) {
#pragma unroll
            for (int k = 0; k < SUBS; ++k) sum += v[k];
        } else {   // column g of the hand-over area: the owner reads its entries as contiguous 16-byte words
#pragma unroll
            for (int k = 0; k < SUBS; ++k) xf[g * CS + (t - 1) * SUBS + k] = v[k];
        }
    }
    gs_barrier();
    fresh = work && t == 0;
    key = 0;
    if (fresh) {   // the owner: its own subspaces are summed; now the other lanes' in ascending m — the column's 16-byte words are
        // requested a batch at a time before the first of the batch is added (the compiler had issued four and then one at a time,
        // each waited for: 17 exposed LDS latencies in front of the finish)
        const gs_f4 *col = reinterpret_cast<const gs_f4 *>(xf + g * CS);
        constexpr int NW = CS / 4;                          // 16-byte words of the column (the last may hold 1 .. 4 entries)
        constexpr int EB = NW % 7 == 0 ? 7 : 8;             // words per batch (28 / 32 registers; the last batch may be short)
#pragma unroll
        for (int b7 = 0; b7 < (NW + EB - 1) / EB; ++b7) {
            gs_f4 e4[EB];
#pragma unroll
            for (int i = 0; i < EB; ++i)
                if (b7 * EB + i < NW) e4[i] = col[b7 * EB + i];
            GS_SCHED_FENCE();
#pragma unroll
            for (int i = 0; i < EB; ++i) {
                const int w0 = (b7 * EB + i) * 4;            // (entries past NB are the column's padding: never added)
                if (w0 + 0 < NB) sum += e4[i].x;
                if (w0 + 1 < NB) sum += e4[i].y;
                if (w0 + 2 < NB) sum += e4[i].z;
                if (w0 + 3 < NB) sum += e4[i].w;
            }
            GS_SCHED_FENCE();
        }
        const float sc = gs_finish<VSF>(sum, st_mag[j], query_mag);
        key = gs_key(st_nb[j], sc);
        if (ub_active && sc < ub_T) fresh = false;   // exactly scored and still below the threshold: never popped
    }
}

// visited.add: true iff the node was not in the set.  Linear probing; callers keep the load <= 1/2.
GS_FN bool gs_visit(int32_t *tab, uint32_t mask, int shift, int32_t node)
{
    uint32_t h = ((uint32_t)node * 0x9E3779B1u) >> shift;
    for (;;) {
        const int32_t old = gs_cas(tab + h, -1, node);
        if (old == -1) return true;
        if (old == node) return false;
        h = (h + 1) & mask;
    }
}

// ---- two-tier visited set, tier 1 (LDS) ----
// A two-choice bucketed hash set of 16-bit entries: slots / 4 buckets of four entries (8 bytes, one ds_read_b64).  Node ids
// are < 2^idbits; h_c = node * odd_c mod 2^idbits (c = 0, 1) are two bijections on them.  The top bits of h_c pick bucket
// b_c, the remaining rbits = idbits - log2(buckets) bits are the remainder, and the entry stored in bucket b_c is
// (c << rbits | remainder): (bucket, entry) give h_c back and therefore the node EXACTLY, at half the bytes of a node id —
// 4096 slots are 8 KB, what 8 waves per CU leave free in LDS at the headline shape, and hold the visited set of all but the
// longest searches.  0xFFFF = empty (rbits <= 14 keeps real entries away from it).  A probe reads bucket b_0 and, only if
// that is full, bucket b_1: one or two LDS reads whatever the load (linear probing, the first version, cost the wave the
// LONGEST of its 32 lanes' probe sequences: slower than the global-memory CAS it replaced once the table filled up).
// Returns 0: already in the set; 1: inserted (fresh); 2: both buckets are full of other nodes.  Buckets never lose entries,
// so a node answered 2 once is answered 2 on every later probe and tier 2 alone decides about it; and a node is inserted into
// b_1 only while b_0 is full, i.e. it can never sit in b_1 while b_0 has room: a node lives in exactly one place.
struct GsVis1 {
    uint32_t *w;       // LDS words, two entries each; bucket b = words 2b, 2b + 1
    uint32_t bmask;    // buckets - 1
    uint32_t idmask;   // 2^idbits - 1
    int rbits;
};

// 0 found / 1 inserted / 2 bucket full of other entries
GS_FN int gs_v1_bucket(uint32_t *bw, uint32_t mine)
{
    for (;;) {
        const uint32_t w0 = bw[0], w1 = bw[1];
        const uint32_t e0 = w0 & 0xFFFFu, e1 = w0 >> 16, e2 = w1 & 0xFFFFu, e3 = w1 >> 16;
        if (e0 == mine || e1 == mine || e2 == mine || e3 == mine) return 0;
        const int slot = e0 == 0xFFFFu ? 0 : (e1 == 0xFFFFu ? 1 : (e2 == 0xFFFFu ? 2 : (e3 == 0xFFFFu ? 3 : -1)));
        if (slot < 0) return 2;
        const uint32_t w = slot < 2 ? w0 : w1;
        const int sh = (slot & 1) * 16;
        const uint32_t want = (w & ~(0xFFFFu << sh)) | (mine << sh);
        if (gs_lds_cas(bw + (slot >> 1), w, want) == w) return 1;
        // a neighbouring lane changed the word first: look again (its entry may have taken the slot)
    }
}

GS_FN int gs_visit1(const GsVis1 &t, int32_t node)
{
    const uint32_t rmask = (1u << t.rbits) - 1u;
    const uint32_t h0 = ((uint32_t)node * 0x9E3779B1u) & t.idmask;
    const int r = gs_v1_bucket(t.w + 2u * ((h0 >> t.rbits) & t.bmask), h0 & rmask);
    if (r != 2) return r;
    const uint32_t h1 = ((uint32_t)node * 0x85EBCA6Bu) & t.idmask;
    return gs_v1_bucket(t.w + 2u * ((h1 >> t.rbits) & t.bmask), (1u << t.rbits) | (h1 & rmask));
}

// upper-level adjacency row of `node`, or nullptr (uniform: every lane probes the same slots)
GS_FN const int32_t *gs_level_row(const GsLevel &L, int32_t node)
{
    if (!L.hkeys) return (node >= 0 && node < L.count) ? L.nbrs + (int64_t)node * L.degree : nullptr;
    uint32_t h = ((uint32_t)node * 0x9E3779B1u) >> L.hshift;
    for (;;) {
        const int32_t k = L.hkeys[h];
        if (k == node) return L.nbrs + (int64_t)L.hvals[h] * L.degree;
        if (k == -1) return nullptr;
        h = (h + 1) & L.hmask;
    }
}

// Per-query traversal state.  Every member is wave-uniform (all 64 lanes hold the same value).
struct GsState {
    long long *cand, *res, *evicted, *samp;  // LDS
    long long *spill;                        // global
    int cand_n, spill_n, res_n, ev_n, res_min_idx, spill_cap;
    long long spill_max, res_min;
    int32_t status;
};

// Move every LDS-tier key <= the median of 64 samples to the spill tier (cand_n >= 64).
GS_FN void gs_partition(GsState &s, const GsParams &p)
{
    const int lane = gs_lane();
    const uint64_t lt = (1ull << lane) - 1ull;
    const long long mine = s.cand[(int)(((long long)lane * s.cand_n) >> 6)];
    s.samp[lane] = mine;
    gs_barrier();
    int rank = 0;
    for (int j = 0; j < 64; ++j) rank += (s.samp[j] < mine) ? 1 : 0;
    const long long pivot = gs_shfl(mine, gs_first(gs_ballot(rank == 31)));
    int new_n = 0, moved = 0;
    for (int base = 0; base < s.cand_n; base += 64) {
        const int i = base + lane;
        const bool in = i < s.cand_n;
        const long long k = in ? s.cand[i] : 0;
        const bool hi = in && k > pivot;
        const bool lo = in && !hi;
        const uint64_t mh = gs_ballot(hi), ml = gs_ballot(lo);  // every lane has read its key before any lane writes
        if (hi) s.cand[new_n + gs_popc(mh & lt)] = k;          // in place: target index <= i
        if (lo) {
            const int pos = s.spill_n + moved + gs_popc(ml & lt);
            if (pos < s.spill_cap) s.spill[pos] = k;
        }
        new_n += gs_popc(mh);
        moved += gs_popc(ml);
        gs_barrier();
    }
    if (s.spill_n + moved > s.spill_cap) s.status = GS_OVERFLOW;
    s.spill_n += moved;
    s.cand_n = new_n;
    s.spill_max = pivot;  // the pivot itself moved, everything that stayed is larger
}

// The LDS tier ran dry while keys wait in the spill tier: bring the BEST of them back (all of them if they fit in half the
// tier; else those above a pivot taken from 64 samples, aimed at a quarter of the tier), so that the pops that follow scan LDS
// again instead of the whole spill tier in global memory — one pass over the tier per ~cand_cap / 4 pops instead of one per pop
// (a threshold search that waits for the TwoPhaseTracker drains hundreds of candidates this way: 12x the time of a plain
// search before this existed).  The invariant of the two tiers holds afterwards (every LDS key > spill_max >= every spilled
// key), so the pop order — always the global maximum — is unchanged.  Leaves cand_n == 0 only if no pivot separates anything
// (the caller then pops from the spill tier directly, as before).
GS_FN void gs_refill(GsState &s, const GsParams &p)
{
    const int lane = gs_lane();
    const uint64_t lt = (1ull << lane) - 1ull;
    gs_fence();
    const int n = s.spill_n;
    if (n <= p.cand_cap / 2) {
        for (int base = 0; base < n; base += 64)
            if (base + lane < n) s.cand[base + lane] = s.spill[base + lane];
        s.cand_n = n;
        s.spill_n = 0;
        s.spill_max = GS_KEY_MIN;
        gs_barrier();
        return;
    }
    const long long mine = s.spill[(int)(((long long)lane * n) >> 6)];
    s.samp[lane] = mine;
    gs_barrier();
    int rank = 0;
    for (int j = 0; j < 64; ++j) rank += (s.samp[j] < mine) ? 1 : 0;   // keys are unique: the ranks are 0..63, each once
    int r = 63 - (int)(((long long)(p.cand_cap / 4) * 64) / n);           // ~cand_cap / 4 keys expected above the rank-r sample
    r = r < 1 ? 1 : (r > 62 ? 62 : r);
    long long pivot = 0;
    int above = 0;
    for (int attempt = 0; attempt < 8; ++attempt) {
        pivot = gs_shfl(mine, gs_first(gs_ballot(rank == r)));
        above = 0;
        for (int base = 0; base < n; base += 64) above += gs_popc(gs_ballot(base + lane < n && s.spill[base + lane] > pivot));
        if (above <= p.cand_cap - 64 || r >= 62) break;   // (r <= 62: at least the largest sample lies above the pivot)
        r += (64 - r) / 2;                                  // too many for the tier: a higher pivot
        if (r > 62) r = 62;
    }
    gs_barrier();
    if (above == 0 || above > p.cand_cap - 64) return;
    int nc = 0, ns = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const bool in = i < n;
        const long long k = in ? s.spill[i] : 0;
        const bool hi = in && k > pivot;
        const bool lo = in && !hi;
        const uint64_t mh = gs_ballot(hi), ml = gs_ballot(lo);  // every lane has read its key before any lane writes
        if (hi) s.cand[nc + gs_popc(mh & lt)] = k;
        if (lo) s.spill[ns + gs_popc(ml & lt)] = k;             // in place: target index <= i
        nc += gs_popc(mh);
        ns += gs_popc(ml);
        gs_barrier();
    }
    s.cand_n = nc;
    s.spill_n = ns;
    s.spill_max = pivot;   // the pivot itself stayed behind; everything that moved is larger
    gs_fence();
}

// candidates.push for up to one key per lane
GS_FN void gs_push(GsState &s, const GsParams &p, long long key, bool has)
{
    const int lane = gs_lane();
    const uint64_t lt = (1ull << lane) - 1ull;
    bool to_lds;
    uint64_t ml;
    for (;;) {
        to_lds = has && (s.spill_n == 0 || key > s.spill_max);
        ml = gs_ballot(to_lds);
        if (s.cand_n + gs_popc(ml) <= p.cand_cap) break;
        gs_partition(s, p);
        if (s.status != GS_OK) return;
    }
    if (to_lds) s.cand[s.cand_n + gs_popc(ml & lt)] = key;
    s.cand_n += gs_popc(ml);
    const bool to_sp = has && !to_lds;
    const uint64_t ms = gs_ballot(to_sp);
    if (ms) {
        if (s.spill_n + gs_popc(ms) > s.spill_cap) {
            s.status = GS_OVERFLOW;
            return;
        }
        if (to_sp) s.spill[s.spill_n + gs_popc(ms & lt)] = key;
        s.spill_n += gs_popc(ms);
    }
    gs_barrier();
}

// UBR: raise the pop threshold T and drop what it rules out.  Among 64 samples of the LDS candidate tier, find — by bisection over
// the sample ranks, each step an exact count over the tier and the kept results — the HIGHEST sample such that at least rk known
// keys are strictly greater: its score becomes T, and every candidate whose SCORE is below T can never be popped (rk nodes with a
// strictly higher score are popped or kept first, then stopSearch — a comparison of scores — ends the layer), so it is discarded.  Keys of the spill tier (all below every LDS key)
// are not counted: the count can only be too small.  Wave-uniform; leaves T alone when no sample qualifies.
GS_FN void gs_ubr_trim_lds(GsState &s, int rk, float &T)
{
    const int lane = gs_lane();
    const uint64_t lt = (1ull << lane) - 1ull;
    const int n = s.cand_n;
    // (a key's high word is never 0x80000000 — gs_key canonicalises NaN — so these sentinels are unique and below every real key)
    const long long mine = n >= 64 ? s.cand[(int)(((long long)lane * n) >> 6)] : (lane < n ? s.cand[lane] : GS_KEY_MIN + 1 + lane);
    s.samp[lane] = mine;
    gs_barrier();
    int rank = 0;
    for (int j = 0; j < 64; ++j) rank += (s.samp[j] < mine) ? 1 : 0;   // keys are unique: the ranks are 0..63, each once
    auto above = [&](long long pv) -> int {
        int c = 0;
        for (int b = 0; b < n; b += 64) c += gs_popc(gs_ballot(b + lane < n && s.cand[b + lane] > pv));
        for (int b = 0; b < s.res_n; b += 64) c += gs_popc(gs_ballot(b + lane < s.res_n && s.res[b + lane] > pv));
        return c;
    };
    auto sample = [&](int r) -> long long { return gs_shfl(mine, gs_first(gs_ballot(rank == r))); };
    long long pv = sample(0);
    if (above(pv) < rk) {
        gs_barrier();
        return;
    }
    int lo = 0, hi = 63;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        const long long pm = sample(mid);
        if (above(pm) >= rk) {
            lo = mid;
            pv = pm;
        } else {
            hi = mid - 1;
        }
    }
    const float ps = gs_key_score(pv);
    if (!(ps >= 0.0f) || !(ps > T)) {   // (a sentinel decodes to NaN; a negative score never becomes a result: no threshold from it)
        gs_barrier();
        return;
    }
    int new_n = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const bool in = i < n;
        const long long k = in ? s.cand[i] : 0;
        // (by SCORE, not by key: stopSearch compares scores, so a candidate that ties with the threshold may still be popped)
        const bool keep = in && (int32_t)(k >> 32) >= (int32_t)(pv >> 32);
        const uint64_t mk = gs_ballot(keep);   // every lane has read its key before any lane writes
        if (keep) s.cand[new_n + gs_popc(mk & lt)] = k;   // in place: target index <= i
        new_n += gs_popc(mk);
        gs_barrier();
    }
    s.cand_n = new_n;
    T = ps;
}

// Round 6: the same trim with the keys held in REGISTERS.  The LDS form above re-read the candidate tier and the result queue for
// every step of the bisection (7 steps x 4-6 passes of ds_read + compare + ballot, each waited for in turn) and found the sample
// ranks through 64 broadcast reads: 8.7 k clocks per trim, 1.2 k per expansion of the headline.  Here every lane loads its <= 4
// candidate keys and <= 2 result keys ONCE, the sample ranks come from 64 v_readlane pairs, a bisection step is six compares +
// s_bcnt1, and the compaction writes straight from the registers.  Same pivot, same survivors, same T as gs_ubr_trim_lds (the
// emulator test runs both); tiers of more than 256 keys / result queues of more than 128 take the LDS form.
GS_FN void gs_ubr_trim(GsState &s, int rk, float &T)
{
    const int n = s.cand_n;
    if (n > 256 || s.res_n > 128) {
        gs_ubr_trim_lds(s, rk, T);
        return;
    }
    const int lane = gs_lane();
    const uint64_t lt = (1ull << lane) - 1ull;
    long long ck[4], rr[2];
#pragma unroll
    for (int u = 0; u < 4; ++u) ck[u] = lane + 64 * u < n ? s.cand[lane + 64 * u] : GS_KEY_MIN;   // (GS_KEY_MIN is above no pivot)
#pragma unroll
    for (int u = 0; u < 2; ++u) rr[u] = lane + 64 * u < s.res_n ? s.res[lane + 64 * u] : GS_KEY_MIN;
    // (a key's high word is never 0x80000000 — gs_key canonicalises NaN — so these sentinels are unique and below every real key)
    const long long mine = n >= 64 ? s.cand[(int)(((long long)lane * n) >> 6)] : (lane < n ? ck[0] : GS_KEY_MIN + 1 + lane);
    int rank = 0;
#pragma unroll
    for (int j = 0; j < 64; ++j) rank += (gs_shfl(mine, j) < mine) ? 1 : 0;   // keys are unique: the ranks are 0..63, each once
    auto above = [&](long long pv) -> int {
        int c = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) c += gs_popc(gs_ballot(ck[u] > pv));
#pragma unroll
        for (int u = 0; u < 2; ++u) c += gs_popc(gs_ballot(rr[u] > pv));
        return c;
    };
    auto sample = [&](int r) -> long long { return gs_shfl(mine, gs_first(gs_ballot(rank == r))); };
    long long pv = sample(0);
    if (above(pv) < rk) {
        gs_barrier();
        return;
    }
    int lo = 0, hi = 63;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        const long long pm = sample(mid);
        if (above(pm) >= rk) {
            lo = mid;
            pv = pm;
        } else {
            hi = mid - 1;
        }
    }
    const float ps = gs_key_score(pv);
    if (!(ps >= 0.0f) || !(ps > T)) {   // (a sentinel decodes to NaN; a negative score never becomes a result: no threshold from it)
        gs_barrier();
        return;
    }
    int new_n = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        // (by SCORE, not by key: stopSearch compares scores, so a candidate that ties with the threshold may still be popped)
        const bool keep = lane + 64 * u < n && (int32_t)(ck[u] >> 32) >= (int32_t)(pv >> 32);
        const uint64_t mk = gs_ballot(keep);
        if (keep) s.cand[new_n + gs_popc(mk & lt)] = ck[u];   // (every key was read into registers before the first write)
        new_n += gs_popc(mk);
    }
    gs_barrier();
    s.cand_n = new_n;
    T = ps;
}

// ---- round 6: the exact rerank inside the traversal wave (GraphSearcher.reranking :471-507 / NodeQueue.rerank :160-230: the exact
//      similarity of every kept approximate result; the selection stays with topk / rerank_tie_kernel) --------------------------------
// One round = up to 64 of the query's kept results, lane = row.  The rows travel exactly as in exact_gather_tr_kernel (k_exact.hip
// tr_rows, the 64 x 64 shape): load instruction k fetches 64-float chunks of rows 4 k + sub, 16 bytes per lane, the chunk is
// transposed through an LDS tile (row stride 68 dwords), and lane r walks row r's chunk in index order with the reference's
// association per block of eight (DefaultVectorUtilSupport.java:38-105 dot, :158-193 L2, :121-139 cosine; dot8 / l28 / the cosine
// chain of k_exact.hip) — non-fused, the same bits.  The query's chunk sits one float per lane in a register and reaches the chain
// through gs_shfl from a wave-uniform lane (v_readlane_b32: the scalar operand of the multiply).  Why here: the separate kernel cost
// 5.4 ms of a 57.6 ms step behind a traversal that leaves HBM at 13 % and the SIMDs' issue slots at 38 %; inside the wave the same
// work fills those gaps (the search has ended: its 96 table registers and its LDS block are free).  Every lane takes part in every
// step (the shuffles are wave collectives): a lane without a row walks zeros.
typedef float gs_v4f __attribute__((vector_size(16)));   // (a builtin vector: loadable through an address-space-qualified pointer)

template <int VSF>
GS_FN float gs_rr_block8(float acc, const float (&a)[8], const gs_v4f v0, const gs_v4f v1)
{
    if (VSF == 1) {
        float t = v0[0] * a[0] + v0[1] * a[1];
        t = t + v0[2] * a[2];
        t = t + v0[3] * a[3];
        t = t + v1[0] * a[4];
        t = t + v1[1] * a[5];
        t = t + v1[2] * a[6];
        t = t + v1[3] * a[7];
        return acc + t;
    } else if (VSF == 0) {
        const float d0 = a[0] - v0[0], d1 = a[1] - v0[1], d2 = a[2] - v0[2], d3 = a[3] - v0[3];
        const float d4 = a[4] - v1[0], d5 = a[5] - v1[1], d6 = a[6] - v1[2], d7 = a[7] - v1[3];
        float t = d0 * d0 + d1 * d1;
        t = t + d2 * d2;
        t = t + d3 * d3;
        t = t + d4 * d4;
        t = t + d5 * d5;
        t = t + d6 * d6;
        t = t + d7 * d7;
        return acc + t;
    } else {
        float s = acc;
        s += a[0] * v0[0];
        s += a[1] * v0[1];
        s += a[2] * v0[2];
        s += a[3] * v0[3];
        s += a[4] * v1[0];
        s += a[5] * v1[1];
        s += a[6] * v1[2];
        s += a[7] * v1[3];
        return s;
    }
}

// my_row: the ordinal of this lane's row, -1 = none; qraw: the query's raw vector (wave-uniform); tile: gs_rr_lds_bytes() of LDS.
// First version (profiles/r6_s): inlined, one chunk requested ahead, the chain a runtime loop of eight blocks, each behind its own pair
// of LDS reads and sixteen lane broadcasts — 12.5 k clocks per chunk, 150 k per query: the fused rerank cost the traversal what the
// kernel of its own had cost the step.  Now: TWO chunks in flight, one v_readlane_b32 per query element (gs_bcast32), the chain
// unrolled so that its LDS reads are issued together — and a CALL (GS_NOINLINE): inlined, the 130 staging registers pushed three
// per-query values of the expansion loop into scratch (two scratch loads per expansion in front of the drop test).
template <int VSF>
GS_NOINLINE float gs_rr_round(const float *vecs_generic, int D, const float *qraw_generic, int32_t my_row, float *tile_generic)
{
    GS_LDS_AS float *tile = (GS_LDS_AS float *)tile_generic;
    GS_GLOBAL_AS const float *vecs = (GS_GLOBAL_AS const float *)vecs_generic;
    GS_GLOBAL_AS const float *qraw = (GS_GLOBAL_AS const float *)qraw_generic;
    const int lane = gs_lane();
    const int seg = (lane & 15) * 4, sub = lane >> 4;
    int32_t ro[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) ro[k] = gs_shfl32(my_row, 4 * k + sub);
    const int nc = (D + GS_RR_CH - 1) / GS_RR_CH;
    gs_v4f rA[16], rB[16];
    float qA = 0.0f, qB = 0.0f;
    // (requests are unconditional — sixteen loads back to back, no branch around each: a lane without a row reads row 0, a piece past
    //  the end of a ragged last chunk reads the row's last 16 bytes; neither value is ever used — the chain skips the blocks past D
    //  and the caller discards the score of a lane without a row)
    auto issue = [&](int c, gs_v4f (&r)[16], float &qv) {
        const int off = c * GS_RR_CH + seg < D ? c * GS_RR_CH + seg : D - 4;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            r[k] = *reinterpret_cast<GS_GLOBAL_AS const gs_v4f *>(vecs + (int64_t)(ro[k] >= 0 ? ro[k] : 0) * D + off);
        qv = qraw[c * GS_RR_CH + lane < D ? c * GS_RR_CH + lane : D - 1];
    };
    float acc = 0.0f;
    // chunk c: the staged rows leave their registers, the registers take chunk c + 2, the lane walks its row's 64 floats
    auto step = [&](int c, gs_v4f (&r)[16], float &qv) {
#pragma unroll
        for (int k = 0; k < 16; ++k) *reinterpret_cast<GS_LDS_AS gs_v4f *>(tile + (4 * k + sub) * GS_RR_LS + seg) = r[k];
        const uint32_t qc = __builtin_bit_cast(uint32_t, qv);
        gs_barrier();
        if (c + 2 < nc) issue(c + 2, r, qv);
        const int len = (D - c * GS_RR_CH < GS_RR_CH) ? (D - c * GS_RR_CH) : GS_RR_CH;   // (a multiple of 8: exact_tr_supported)
        GS_LDS_AS const float *row = tile + lane * GS_RR_LS;
        auto block = [&](int i, const gs_v4f v0, const gs_v4f v1) {
            float a8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a8[j] = __builtin_bit_cast(float, gs_bcast32(qc, i + j));
            acc = gs_rr_block8<VSF>(acc, a8, v0, v1);
        };
        if (len == GS_RR_CH) {   // the LDS reads of half a chunk issued together (the fence keeps them in front of the chain); all sixteen
                                 // at once made the callee touch every register and the CALLER spill inside its expansion loop
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                gs_v4f t[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) t[i] = *reinterpret_cast<GS_LDS_AS const gs_v4f *>(row + 32 * h + 4 * i);
                GS_SCHED_FENCE();
#pragma unroll
                for (int i = 0; i < 32; i += 8) block(32 * h + i, t[i / 4], t[i / 4 + 1]);
            }
        } else {
            for (int i = 0; i < len; i += 8)
                block(i, *reinterpret_cast<GS_LDS_AS const gs_v4f *>(row + i), *reinterpret_cast<GS_LDS_AS const gs_v4f *>(row + i + 4));
        }
        gs_barrier();
    };
    issue(0, rA, qA);
    if (nc > 1) issue(1, rB, qB);
    // (three chunks in flight — 192 staging registers — measured slower: 52.3 vs 51.1 ms, profiles/r6_z)
    for (int c = 0; c < nc; c += 2) {
        step(c, rA, qA);
        if (c + 1 < nc) step(c + 1, rB, qB);
    }
    return acc;
}

#ifndef GS_CLOCK
#define GS_CLOCK() 0ull  // the GPU build maps it to the shader clock (gs_wave_hip.h); the emulator has no clock
#endif

// One query, start to finish.  lds: gs_lds_bytes() bytes, 16-byte aligned.
// PAIR: every level's degree is <= 32 -> pair-lane scoring (decided by the host at launch)
// PROF: developer aid — per-phase shader-clock totals of the expansion loop are added to p.prof[0..7]
//       (pop, result insert, row + block + visited probes, scoring, push, expansions, queries, setup + epilogue)
// SES:  GraphSearcher OBJECTS (jv_hip_searcher_*): layer 0 admits `score >= p.threshold` (:437) and, for threshold > 0, stops
//       through ScoreTracker.TwoPhaseTracker (ScoreTracker.java:80-140: a 500-score window + the 100 best scores, both in LDS);
//       expandedCountBaseLayer is reported.  What reranking / resume need beyond that is rebuilt by the host from the
//       addTopCandidate log (graph_search.cpp searcher_search_device).
// PAIRC: pair-lane scoring of the FRESH neighbours of rows up to 64 wide whose codes are read by ordinal (the builder's working
//       rows, maxDegree x neighborOverflow): one lane per neighbour probes the visited set, the unvisited ids are compacted
//       through LDS and scored two lanes each like PAIR (<= 32 per pass, a second pass for the rest; M > 96: four lanes each,
//       16 per pass).  LDS layout = PAIR's.
template <int VSF, int CH16, bool PAIR, bool PROF = false, bool SES = false, bool PAIRC = false, bool UBR = false>
GS_FN bool gs_search_one(const GsParams &p, int q, int worker, char *lds, bool nodefer)
{
    static_assert(!UBR || ((PAIR != PAIRC) && !SES && CH16 % 2 == 0 && 60 * CH16 * 16 + 256 <= 64 * CH16 * 16),
                  "the register-table bound form serves the pair-lane kernels (over the row, or over the compacted fresh list), M a multiple of 32 and >= 64");
    static_assert(!(UBR && PAIRC) || CH16 <= 6, "the bound form of the compacted pair kernel: two lanes per neighbour (M <= 96)");
    static_assert(!PAIRC || (!PAIR && CH16 > 0), "the compacted pair form is a variant of the plain one-lane-per-neighbour kernel");
    constexpr bool XA = PAIR || PAIRC;   // the worker's LDS block has the [M/2][32] exchange area
    static_assert(CH16 > 0 || !PAIR, "the generic form (CH16 = 0) is one lane per neighbour, table-free");
    constexpr int CW = CH16 > 0 ? CH16 : 1;  // code words a lane holds (the generic form reads its row from memory instead)
    unsigned long long pf[5] = {0, 0, 0, 0, 0};
    unsigned long long fh[4] = {0, 0, 0, 0};  // PROF: scored neighbours in expansions with <= 8 / <= 16 / <= 24 / <= 32 fresh ones
    unsigned long long pt = 0, pq0 = 0;
    unsigned long long px[3] = {0, 0, 0};   // PROF: setup (entry -> first pop), level transitions, epilogue
    unsigned long long pz[2] = {0, 0}, pzf[5] = {0, 0, 0, 0, 0}, pzp[2] = {0, 0};   // (pzf: the five phases at the levels above 0; pzp: their scoring passes' clocks, passes)
    bool prof_upper = false;
    unsigned long long py[7] = {0, 0, 0, 0, 0, 0, 0};   // PROF (UBR): wait for the row, scoring rounds, owner sum + finish, passes, expansions with <= 4 / <= 8 / <= 16 survivors
    if (PROF) pq0 = GS_CLOCK();
#define GS_PHASE(i)                          \
    do {                                     \
        if (PROF) {                          \
            const unsigned long long now_ = GS_CLOCK(); \
            pf[i] += now_ - pt;              \
            if (prof_upper) pzf[i] += now_ - pt; \
            pt = now_;                       \
        }                                    \
    } while (0)
    const int lane = gs_lane();
    float *qs = reinterpret_cast<float *>(lds);
    GsState s;
    s.res = reinterpret_cast<long long *>(lds + gs_q_bytes(p.D));
    s.cand = s.res + p.rerankK;
    s.evicted = s.cand + p.cand_cap;
    const int evict_cap = p.evict_cap > 0 ? p.evict_cap : GS_EVICT_CAP;
    // PAIR: [M/2][32] entries handed from high to low lanes; the partition step's 64-key sample buffer lives in the same
    // bytes (it is only touched inside gs_push, after every lane has consumed the exchange area)
    // (with an exchange area: at the next 16-byte boundary — its hand-over columns are read as 16-byte words; gs_lds_bytes has the 8 bytes)
    // (an OFFSET from lds rounded up, not the pointer's integer value: a pointer that went through an integer is a flat pointer to
    // the compiler — every access of the exchange area became a flat_load / flat_store instead of a ds_ operation: 53.5 vs 49.0 ms for
    // the headline and 40 s instead of 17.6 s of builder searches at C5, profiles/r6_g)
    float *xchg = reinterpret_cast<float *>(lds + (XA ? (((size_t)(reinterpret_cast<char *>(s.evicted + evict_cap) - lds) + 15) & ~(size_t)15)
                                                      : (size_t)(reinterpret_cast<char *>(s.evicted + evict_cap) - lds)));
    s.samp = s.evicted + evict_cap;
    (void)xchg;
    s.spill = p.spill + (int64_t)worker * p.spill_cap;
    s.spill_cap = p.spill_cap;
    s.cand_n = s.spill_n = s.res_n = s.ev_n = 0;
    s.res_min_idx = -1;
    s.spill_max = GS_KEY_MIN;
    s.res_min = GS_KEY_MAX;
    s.status = GS_OK;
    int vcap = 1 << p.vcap_log2;
    uint32_t vmask = (uint32_t)vcap - 1u;
    int vshift = 32 - p.vcap_log2;
    int32_t *vis = p.visited + (int64_t)worker * vcap;
    bool grown = false;
    // tier 1 of the visited set (LDS), behind everything else in the worker's LDS block.  Its geometry is re-derived from the
    // launch parameters at every use (gs_v1_of) instead of living in registers across the scoring loop.
    const bool has_v1 = p.v1_log2 > 0;
    auto gs_v1_of = [&]() -> GsVis1 {
        GsVis1 t;
        const size_t base = ((size_t)((char *)(xchg + (XA ? gs_xchg_floats(p.M) : 0)) - lds) + (XA ? 0 : sizeof(long long) * 64) + 15) & ~(size_t)15;
        t.w = reinterpret_cast<uint32_t *>(lds + base);
        t.bmask = (1u << (p.v1_log2 - 2)) - 1u;
        t.idmask = (p.v1_idbits >= 32) ? 0xFFFFFFFFu : ((1u << p.v1_idbits) - 1u);
        t.rbits = p.v1_idbits > p.v1_log2 - 2 ? p.v1_idbits - (p.v1_log2 - 2) : 0;
        return t;
    };
    long long n2 = 0;            // nodes in tier 2
    bool t2_ready = !has_v1;     // tier 2 cleared for this query (eagerly below when there is no LDS tier)
    // The visited table is half full: move to a table of the growth pool (once per query), or give up with GS_OVERFLOW.
    // Wave-uniform.  The old table is read back with atomics (a CAS that can never succeed), like every other access to it.
    auto grow = [&]() -> bool {
        if (grown || !p.big_visited) return false;
        long long sv = 0;
        if (lane == 0) sv = (long long)gs_fetch_add(p.big_next, 1u);
        const long long slot = gs_shfl(sv, 0);
        if (slot >= p.big_count) return false;
        const int bcap = 1 << p.big_log2;
        int32_t *nvis = p.big_visited + slot * (long long)bcap;
        {
            gs_u4 *v4 = reinterpret_cast<gs_u4 *>(nvis);
            const gs_u4 ones = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            for (int i = lane; i < bcap / 4; i += 64) v4[i] = ones;
        }
        gs_fence();
        gs_barrier();
        const uint32_t bmask = (uint32_t)bcap - 1u;
        const int bshift = 32 - p.big_log2;
        for (int i = lane; i < vcap; i += 64) {
            const int32_t v = gs_cas(vis + i, -2, -2);
            if (v >= 0) (void)gs_visit(nvis, bmask, bshift, v);
        }
        long long *nspill = p.big_spill + slot * (long long)p.big_spill_cap;
        gs_fence();
        for (int i = lane; i < s.spill_n; i += 64) nspill[i] = s.spill[i];
        gs_fence();
        gs_barrier();
        vis = nvis;
        vcap = bcap;
        vmask = bmask;
        vshift = bshift;
        s.spill = nspill;
        s.spill_cap = p.big_spill_cap;
        grown = true;
        return true;
    };
    long long n_visited = 0, n_expanded = 0;
    int n_log = 0;   // push-log entries offered at layer 0 (wave-uniform)

    // tier 2 is cleared by the first probe that needs it (wave-uniform call)
    auto t2_init = [&]() {
        gs_u4 *v4 = reinterpret_cast<gs_u4 *>(vis);
        const gs_u4 ones = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        for (int i = lane; i < vcap / 4; i += 64) v4[i] = ones;
        gs_fence();
        gs_barrier();
        t2_ready = true;
    };
    // visited.add for one node per participating lane; true = the node was not in the set.  Wave-uniform call (every lane
    // enters, `act` says whether it carries a node).  Sets s.status on table overflow.
    auto visit = [&](bool act, int32_t nb) -> bool {
        int r1 = act ? 2 : 0;
        if (has_v1 && act) r1 = gs_visit1(gs_v1_of(), nb);
        bool fr = r1 == 1;
        if (gs_ballot(r1 == 2)) {
            if (!t2_ready) t2_init();
            const bool f2 = r1 == 2 && gs_visit(vis, vmask, vshift, nb);
            fr = fr || f2;
            n2 += gs_popc(gs_ballot(f2));
            if ((n2 + 1) * 2 > vcap && !((n2 + 1) * 2 <= (1ll << p.big_log2) && grow())) s.status = GS_OVERFLOW;
        }
        return fr;
    };

    // ---- per-query setup: clear the visited set, stage the centred query ----
    {
        const gs_u4 ones = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        if (has_v1) {
            gs_u4 *t4 = reinterpret_cast<gs_u4 *>(gs_v1_of().w);
            for (int i = lane; i < (int)((2u << p.v1_log2) / 16u); i += 64) t4[i] = ones;
        } else {
            gs_u4 *v4 = reinterpret_cast<gs_u4 *>(vis);
            for (int i = lane; i < vcap / 4; i += 64) v4[i] = ones;
        }
        if constexpr (CH16 == 0) {  // any D: rows of the query matrix need not be 16-byte aligned
            const float *src = p.cq + (int64_t)q * p.D;
            for (int i = lane; i < p.D; i += 64) qs[i] = src[i];
        } else {
            const gs_f4 *src = reinterpret_cast<const gs_f4 *>(p.cq + (int64_t)q * p.D);
            gs_f4 *dst = reinterpret_cast<gs_f4 *>(qs);
            for (int i = lane; i < p.D / 4; i += 64) dst[i] = src[i];
        }
    }
    gs_fence();
    gs_barrier();
    const float query_mag = (VSF == 2) ? p.bmag[q] : 0.0f;
    const unsigned long long *acc = p.accept ? p.accept + (long long)q * p.accept_stride : nullptr;
    const int32_t excl = p.exclude ? p.exclude[q] : -1;
    // ---- the bound form's pop threshold (wave-uniform): >= rerankK nodes with exact score >= ub_T are known (queued or popped) ----
    float ub_T = -__builtin_inff();
    bool ub_on = false;
    unsigned long long ub_dropped = 0;
    (void)ub_T;
    (void)ub_dropped;
    // ---- UBR: the query's prebuilt bound table in registers; base / scale of its bounds; pushes since the last trim ----
    uint32_t ubtab[UBR ? CH16 * 16 : 1];
    float ubr_base = 0.0f, ubr_scale = 0.0f;
    int ubr_since = 0;
    if constexpr (UBR) {
        gs_ubr_load<CH16>(p.ubr_tab + (int64_t)q * (CH16 * 16) * 64, ubtab);
        const float *meta = p.ubr_meta + (int64_t)q * 4;
        ubr_base = meta[0];
        ubr_scale = meta[1];
        ub_on = meta[2] != 0.0f && acc == nullptr && excl < 0;   // (acceptOrds: rejected nodes never become results — no threshold)
    }
    // ---- DEFER (GsParams::defer): exact scores put off above level 0; all that is kept of them is the largest upper bound ----
#ifndef GS_DEFER_ENABLE
#define GS_DEFER_ENABLE 1
#endif
    constexpr bool DEFER = GS_DEFER_ENABLE && UBR && PAIR && !PAIRC && !SES;
    bool d_on = false;
    long long d_maxk = GS_KEY_MIN;     // the largest (U, node) key among the deferred neighbours; GS_KEY_MIN = none (wave-uniform)
    uint32_t d_deferred = 0u;
    if constexpr (DEFER) d_on = !nodefer && p.defer != 0;
    (void)d_on;
    (void)d_maxk;
    (void)d_deferred;
    (void)ubtab;
    (void)ubr_base;
    (void)ubr_scale;
    (void)ubr_since;

    // ---- initializeInternal :334-353: mark and score the entry node ----
    {
        const int32_t e = p.entry_node;
        // first insertion into an empty set: no probing conflicts, no growth check
        if (has_v1) {
            if (lane == 0) (void)gs_visit1(gs_v1_of(), e);
        } else {
            if (lane == 0) (void)gs_visit(vis, vmask, vshift, e);
            n2 = 1;
        }
        float sc;
        if constexpr (CH16 == 0) {
            sc = gs_row_sum_any<VSF>(p, qs, p.codes + (int64_t)e * p.M);
        } else {
            {
                // The entry row's M table entries are formed by the lanes side by side (lane l: subspaces l, l + 64, ...) and parked in
                // the still empty candidate tier; every lane then adds them in ascending m — assembleAndSum's order, the same bits.
                // (One lane walking the row waited for M dependent-in-practice L2 round trips: ~55 k of a query's ~2 M clocks, round 5.)
                constexpr int M_ = CH16 * 16;
                static_assert(M_ * 4 <= 128 * 8, "the candidate tier (>= 128 keys) is the scratch of the entry row");
                float *ent = reinterpret_cast<float *>(s.cand);
                const uint8_t *erow = p.codes + (int64_t)e * p.M;
                for (int m = lane; m < M_; m += 64) ent[m] = gs_lut_entry<VSF>(p.codebooks, qs, m, (int)erow[m]);
                gs_barrier();
                sc = 0.0f;
#pragma unroll 8
                for (int m = 0; m < M_; ++m) sc += ent[m];
                gs_barrier();
            }
        }
        sc = gs_finish<VSF>(sc, (VSF == 2) ? p.code_norms[e] : 0.0f, query_mag);
        if (lane == 0) s.cand[0] = gs_key(e, sc);
        s.cand_n = 1;
        if (UBR && sc != sc) ub_on = false;
        gs_barrier();
    }

    // ---- SES: ScoreTracker.TwoPhaseTracker (ScoreTracker.java:80-140), wave-uniform state + two LDS arrays behind the worker's block
    constexpr int TRK_RECENT = 500, TRK_BEST = 100;
    float *trk_recent = nullptr;
    int32_t *trk_best = nullptr;
    int trk_obs = 0, trk_ridx = 0, trk_nbest = 0, trk_min_idx = -1;
    int32_t trk_min = 0x7fffffff;
    bool thr_on = false;
    long long n_expanded_base = 0;
    float cur_thr = p.threshold;   // the running phase's threshold (SES with n_phases > 1: ph_threshold[phase])
    if constexpr (SES) {
        trk_recent = reinterpret_cast<float *>(lds + gs_lds_bytes(p.D, p.rerankK, p.cand_cap, XA ? p.M : 0, evict_cap, p.v1_log2));
        trk_best = reinterpret_cast<int32_t *>(trk_recent + TRK_RECENT);
    }
    auto trk_sortable = [](float f) -> int32_t {   // NumericUtils.floatToSortableInt
        const int32_t b = (f != f) ? 0x7fc00000 : gs_float_bits(f);
        return b ^ ((b >> 31) & 0x7fffffff);
    };
    // track(score) for every lane with `has` (order inside one expansion is immaterial: the window and the best-100 set are sets)
    auto trk_track = [&](bool has, float sc) {
        const uint64_t fm = gs_ballot(has);
        if (fm == 0) return;
        const uint64_t lt = (1ull << lane) - 1ull;
        if (has) trk_recent[(trk_ridx + gs_popc(fm & lt)) % TRK_RECENT] = sc;
        trk_ridx = (trk_ridx + gs_popc(fm)) % TRK_RECENT;
        const long long mine = (long long)trk_sortable(sc);
        // a FULL heap rejects a score below its minimum, and the minimum only grows: such lanes need no turn
        const uint64_t turns = gs_ballot(has && (trk_nbest < TRK_BEST || !((int32_t)mine < trk_min)));
        for (uint64_t rest = turns; rest; rest &= rest - 1) {   // BoundedLongHeap(100).push :59-69, one score at a time
            const int32_t v = (int32_t)gs_shfl(mine, gs_first(rest));
            if (trk_nbest < TRK_BEST) {
                if (lane == 0) trk_best[trk_nbest] = v;
                if (v < trk_min) {
                    trk_min = v;
                    trk_min_idx = trk_nbest;
                }
                trk_nbest++;
            } else if (!(v < trk_min)) {                     // rejects value < top; an equal value replaces it
                if (lane == 0) trk_best[trk_min_idx] = v;
                gs_barrier();
                long long best = GS_KEY_MAX;
                for (int i = lane; i < TRK_BEST; i += 64) {
                    const long long k = ((long long)trk_best[i] << 32) | (long long)(uint32_t)i;
                    best = k < best ? k : best;
                }
                best = gs_wave_min(best);
                trk_min = (int32_t)(best >> 32);
                trk_min_idx = (int)(best & 0xFFFFFFFFll);
            }
        }
        trk_obs += gs_popc(fm);
        gs_barrier();
    };
    // shouldStop(): only looked at when 500 scores have been seen and the count is a multiple of 100; the 99th percentile of the
    // window is commons-math3's LEGACY estimate: pos = 0.99 * 501 = 495.99 -> sorted[494] + 0.99 * (sorted[495] - sorted[494]),
    // i.e. the 6th and the 5th largest of the 500 (in double, like the reference)
    auto trk_should_stop = [&]() -> bool {
        if (!thr_on || trk_obs < TRK_RECENT || trk_obs % 100 != 0) return false;
        long long loc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = lane + 64 * j;
            loc[j] = i < TRK_RECENT ? (((long long)trk_sortable(trk_recent[i]) << 32) | (long long)(uint32_t)i) : GS_KEY_MIN;
        }
        long long fifth = GS_KEY_MIN, sixth = GS_KEY_MIN;
        for (int round = 0; round < 6; ++round) {
            long long m = GS_KEY_MIN;
#pragma unroll
            for (int j = 0; j < 8; ++j) m = loc[j] > m ? loc[j] : m;
            const long long w = gs_wave_max(m);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (loc[j] == w) loc[j] = GS_KEY_MIN;   // keys are unique (index in the low word): exactly one lane drops it
            if (round == 4) fifth = w;
            if (round == 5) sixth = w;
        }
        auto val = [](long long k) -> double {
            const int32_t e = (int32_t)(k >> 32);
            return (double)gs_bits_float(e ^ ((e >> 31) & 0x7fffffff));
        };
        const double pos = (99.0 / 100.0) * (double)(TRK_RECENT + 1);
        const double lower = val(sixth), upper = val(fifth);
        const double window = lower + (pos - (double)(int)pos) * (upper - lower);
        const int32_t e = trk_min;
        const double worst_best = (double)gs_bits_float(e ^ ((e >> 31) & 0x7fffffff));
        return window < worst_best && window < (double)cur_thr;
    };

    if (PROF) px[0] = GS_CLOCK() - pq0;
    for (int lvl = p.entry_level; lvl >= 0 && s.status == GS_OK; --lvl) {
        int rk = lvl > 0 ? 1 : p.rerankK;
        // (Round 6, measured and not kept: the descriptor copied into registers of its own behind an optimisation barrier removes 14 of
        // the loop's 16 kernel-argument re-reads — and costs 188 bytes of scratch, 40 table registers reloaded in front of every
        // bound phase: 59.2 vs 50.0 ms, profiles/r6_c.)
        const GsLevel &L = p.lv[lvl];
        int phase = 0;
        unsigned long long plv0 = 0;
        const long long plv_e0 = n_expanded;
        if (PROF) plv0 = GS_CLOCK();
        if (PROF) prof_upper = lvl > 0;
        if (SES && lvl == 0 && p.n_phases > 1) {
            rk = p.ph_rerankK[0];
            cur_thr = p.ph_threshold[0];
        }
        float thr = (SES && lvl == 0) ? cur_thr : 0.0f;
        if constexpr (SES) {   // getScoreTracker (ScoreTracker.java:38-58): layer 0 of a threshold search, reset per layer entry
            thr_on = lvl == 0 && cur_thr > 0.0f;
            trk_obs = 0;
            trk_nbest = 0;
            trk_min = 0x7fffffff;
            trk_min_idx = -1;
        }
        // the push log's entry count: wave-uniform, in a scalar register (round 6; it sat in the evicted area's first LDS word, read
        // and written by lane 0 at every addTopCandidate — two LDS round trips on the expansion's chain)
        if (lvl == 0) n_log = 0;
        // ---- searchOneLayer :406-457 (layer 0 of a session with a history: once per phase) ----
        for (;;) {
        for (;;) {
            if (s.cand_n == 0 && s.spill_n == 0) {
                if constexpr (DEFER) {
                    // the reference's queue would still hold the deferred nodes: they are popped unless the stop rule fires on them
                    if (d_maxk != GS_KEY_MIN && !(s.res_n >= rk && gs_key_score(d_maxk) < gs_key_score(s.res_min)) &&
                        !(ub_on && lvl == 0 && ub_T > -__builtin_inff() && gs_key_score(d_maxk) < ub_T))
                        s.status = GS_RESTART;
                }
                break;
            }
            if (s.cand_n == 0) gs_refill(s, p);
            if (PROF) pt = GS_CLOCK();
            int idx;
            long long top, runner_up = GS_KEY_MIN;
            const bool from_lds = s.cand_n > 0;
            if (from_lds) {
                if (p.prefetch && lvl == 0) top = gs_scan_top2(s.cand, s.cand_n, &idx, &runner_up);
                else top = gs_scan_extreme<true>(s.cand, s.cand_n, &idx);
            } else {  // the LDS tier ran dry: the best candidate is somewhere in the spill tier
                gs_fence();
                top = gs_scan_extreme<true>(s.spill, s.spill_n, &idx);
            }
            const float top_score = gs_key_score(top);
            if constexpr (DEFER) {
                if (d_maxk != GS_KEY_MIN) {
                    const float dmax = gs_key_score(d_maxk);
                    if (ub_on && lvl == 0 && ub_T > -__builtin_inff() && dmax < ub_T) {   // no deferred node can ever be popped
                        d_maxk = GS_KEY_MIN;
                    } else if (!(top_score >= dmax) && !(s.res_n >= rk && dmax < gs_key_score(s.res_min))) {
                        // a deferred node may outrank the top of the queue (exact < U <= dmax is all that is known): the query starts over
                        // with every neighbour scored when it is met
                        s.status = GS_RESTART;
                        break;
                    }
                }
            }
            if (s.res_n >= rk && top_score < gs_key_score(s.res_min)) break;  // stopSearch :355-369
            if constexpr (SES) {
                if (trk_should_stop()) break;                                 // "preserve legacy threshold early termination"
            }
            // candidates.pop()
            if (from_lds) {
                if (lane == 0) s.cand[idx] = s.cand[s.cand_n - 1];
                s.cand_n--;
            } else {
                if (lane == 0) s.spill[idx] = s.spill[s.spill_n - 1];
                s.spill_n--;
                s.spill_max = top;  // still an upper bound of what is left
                gs_fence();
            }
            // Speculative touch of the NEXT expansion's data: unless one of the neighbours scored below beats it, the runner-up
            // is popped next, and its adjacency row + FusedPQ block (3.2 KB of HBM, the head of that expansion's dependent chain)
            // can be on their way into L2 while this expansion is scored.  Lanes 0..n-1 each touch one 128-byte line; the loaded
            // words land in the evicted list's (unused at layer 0) LDS bytes and are never read.  Results cannot change.
            if (p.prefetch && lvl == 0 && runner_up != GS_KEY_MIN &&
                !(s.res_n >= rk && gs_key_score(runner_up) < gs_key_score(s.res_min))) {
                const int32_t rn = gs_key_node(runner_up);
                const char *row_b = reinterpret_cast<const char *>(L.nbrs + (int64_t)rn * L.degree);
                const int row_lines = (L.degree * 4 + 127) / 128;
                const int blk_lines = p.blocks ? (p.deg0 * p.M + 127) / 128 : 0;
                const char *addr = nullptr;
                if (lane < row_lines) addr = row_b + lane * 128;
                else if (lane < row_lines + blk_lines) addr = reinterpret_cast<const char *>(p.blocks + (int64_t)rn * p.deg0 * p.M) + (lane - row_lines) * 128;
                else if (VSF == 2 && p.blocks && lane == row_lines + blk_lines) addr = reinterpret_cast<const char *>(p.fused_norms + (int64_t)rn * p.deg0);
                if (addr) gs_prefetch_lds(addr, reinterpret_cast<char *>(s.evicted) + 64);
            }
            // ---- round 6: the popped node's adjacency row, FusedPQ block and magnitudes are REQUESTED here, before the result queue is
            //      touched — the HBM round trip at the head of the expansion's dependent chain (~900 clocks unloaded, more under load)
            //      runs while addTopCandidate, the rescan of the result minimum and (UBR) the trim of the candidate tier work in LDS.
            //      Pair-lane forms at level 0 (the row is found without a map look-up); nothing is consumed before the expansion
            //      below, and a search that stops or skips the expansion simply drops the registers.
            int32_t pre_nb = -1;
            gs_u2 pre_w[PAIR ? (CH16 > 0 ? CH16 : 1) : 1];
            float pre_mag = 0.0f;
            bool pre_loaded = false;
            if constexpr (PAIR) {
#pragma unroll
                for (int c = 0; c < CH16; ++c) pre_w[c] = gs_u2{0u, 0u};
                if (lvl == 0 && !L.hkeys) {
                    const int32_t pn = gs_key_node(top);
                    if (pn >= 0 && pn < L.count) {
                        const int pni = lane & 31;
                        const bool phi = lane >= 32;
                        if (pni < L.degree) pre_nb = (L.nbrs + (int64_t)pn * L.degree)[pni];
                        if (p.blocks != nullptr && pni < L.degree) {
                            const int64_t r = (int64_t)pn * p.deg0 + pni;
                            gs_load_half<CH16>(p.blocks + r * p.M + (phi ? p.M / 2 : 0), pre_w);
                            if (VSF == 2 && !phi) pre_mag = p.fused_norms[r];
                        }
                        pre_loaded = true;
                    }
                }
            }
            (void)pre_nb;
            (void)pre_w;
            (void)pre_mag;
            (void)pre_loaded;
            GS_PHASE(0);
            // threshold 0.0f: `topCandidateScore >= threshold` (:437) keeps negative / NaN scores out of the results (the
            // node is expanded all the same); then addTopCandidate :515-530 (BoundedLongHeap.push / updateTop)
            bool result = top_score >= thr;
            if (result && lvl == 0 && acc) {  // acceptOrds: layer 0 only (upper layers run with Bits.ALL, :276)
                const int32_t tn = gs_key_node(top);
                result = ((acc[tn >> 6] >> (tn & 63)) & 1ull) != 0;
            }
            if (result && lvl == 0 && excl >= 0) result = gs_key_node(top) != excl;
            if (result && lvl == 0 && p.push_log) {  // the addTopCandidate sequence, for rt_body.h's tie resolution
                if (lane == 0 && n_log < p.push_log_cap) p.push_log[(int64_t)q * p.push_log_cap + n_log] = top;
                n_log++;
            }
            if (!result) {
                gs_barrier();
            } else if (s.res_n < rk) {
                if (lane == 0) s.res[s.res_n] = top;
                if (top < s.res_min) {
                    s.res_min = top;
                    s.res_min_idx = s.res_n;
                }
                s.res_n++;
                gs_barrier();
            } else if (top_score > gs_key_score(s.res_min)) {
                if (lvl > 0) {
                    if (s.ev_n >= evict_cap) {
                        s.status = GS_OVERFLOW;
                        break;
                    }
                    if (lane == 0) s.evicted[s.ev_n] = s.res_min;
                    s.ev_n++;
                }
                if (lane == 0) s.res[s.res_min_idx] = top;
                gs_barrier();
                s.res_min = gs_scan_extreme<false>(s.res, s.res_n, &s.res_min_idx);
            } else {
                gs_barrier();
            }
            if constexpr (UBR) {
                // ---- the pop threshold: the worst kept result once the result queue is full; every ubr_trim pushes the candidate
                //      tier is searched for a higher one and loses what it rules out (gs_ubr_trim).  Round 6: here, between the
                //      request for the popped node's row + block and their first use, instead of behind the push — the trim works in
                //      LDS while the HBM round trip runs.  (The popped key already sits in the result queue or was turned away: the
                //      trim's count of known better nodes can only be smaller than behind the push — still a proof.) ----
                if (ub_on && lvl == 0 && s.status == GS_OK) {
                    if (s.res_n >= rk) {
                        const float t = gs_key_score(s.res_min);
                        if (t > ub_T) ub_T = t;
                    }
                    // (also before the LDS tier would have to spill: a partition costs more than a trim and frees less)
                    if ((ubr_since >= p.ubr_trim || s.cand_n + 40 > p.cand_cap) && s.cand_n > 0 && s.cand_n + s.res_n >= rk) {
                        unsigned long long tc0 = 0;
                        if (PROF) tc0 = GS_CLOCK();
                        gs_ubr_trim(s, rk, ub_T);
                        ubr_since = 0;
                        if (PROF) fh[2] += GS_CLOCK() - tc0;
                    }
                }
            }
            if constexpr (SES) {
                // "skip edge loading if we've found a local maximum and we have enough results" (:441-444)
                if (trk_should_stop() && (long long)s.cand_n + s.spill_n >= (long long)rk - s.res_n) continue;
                if (lvl == 0) n_expanded_base++;
            }
            n_expanded++;
            GS_PHASE(1);

            // ---- expand: visited.mark + score + candidates.push for every unvisited neighbour ----
            const int32_t node = gs_key_node(top);
            const int32_t *row = gs_level_row(L, node);
            if (!row) continue;
            const int deg = L.degree;
            const bool fused0 = lvl == 0 && p.blocks != nullptr;
            long long key = 0;
            bool fresh = false;
            if constexpr (PAIR) {
                // ---- pair-lane form: neighbour i is handled by lanes i (low) and i + 32 (high) ----
                const int ni = lane & 31;
                const bool hi = lane >= 32;
                const int m_base = hi ? p.M / 2 : 0;
                int32_t nb;
                gs_u2 w[CH16];
                float node_mag = 0.0f;
                if (pre_loaded) {   // (requested right after the pop, see above; uniform)
                    nb = pre_nb;
#pragma unroll
                    for (int c = 0; c < CH16; ++c) w[c] = pre_w[c];
                    node_mag = pre_mag;
                } else {
                    nb = ni < deg ? row[ni] : -1;
                    if constexpr (UBR) {   // every lane takes part in the bound's cross-lane reads: no uninitialised code words
#pragma unroll
                        for (int c = 0; c < CH16; ++c) w[c] = gs_u2{0u, 0u};
                    }
                    if (fused0 && ni < deg) {  // FusedPQDecoder.similarityToNeighbor: the origin's packed block (zero padded)
                        const int64_t r = (int64_t)node * p.deg0 + ni;
                        gs_load_half<CH16>(p.blocks + r * p.M + m_base, w);
                        if (VSF == 2 && !hi) node_mag = p.fused_norms[r];
                    }
                }
                const int first_neg = gs_first(gs_ballot(!hi && nb < 0));  // rows are packed: the first -1 ends the row
                if (PROF && UBR) py[0] += GS_CLOCK() - pt;   // (the row's ids have arrived)
                const bool valid = ni < first_neg;
                if (!fused0 && valid) {    // PQDecoder.similarityTo: the neighbour's own code
                    gs_load_half<CH16>(p.codes + (int64_t)nb * p.M + m_base, w);
                    if (VSF == 2 && !hi) node_mag = p.code_norms[nb];
                }
                fresh = visit(!hi && valid, nb);  // one probe per neighbour: the low lane's
                if (s.status != GS_OK) break;
                const uint64_t fm = gs_ballot(fresh);
                if (fm == 0) continue;
                n_visited += gs_popc(fm);
                GS_PHASE(2);
                if (PROF && !UBR) {
                    const int f = gs_popc(fm);
                    fh[f <= 8 ? 0 : (f <= 16 ? 1 : (f <= 24 ? 2 : 3))] += (unsigned long long)f;
                }
                if constexpr (UBR) {
                    constexpr int M_ = CH16 * 16;
                    unsigned long long ubc0 = 0;
                    if (PROF) ubc0 = GS_CLOCK();
                    // ---- the bound of every fresh neighbour against the pop threshold ----
                    uint64_t sm = fm;   // the fresh neighbours that need their exact score (bits of the low lanes)
                    const bool ub_active = ub_on && lvl == 0 && ub_T > -__builtin_inff();
                    // DEFER: above level 0 the layer's best result (topK = 1) is the threshold, and what lies below it is DEFERRED — no exact
                    // score now, only the largest upper bound U of the deferred scores is kept (GsParams::defer has the argument)
                    bool ub_up = false;
                    if constexpr (DEFER) ub_up = ub_on && d_on && lvl > 0 && lvl >= p.defer_min_level && s.res_n >= rk;
                    if (ub_active || ub_up) {
                        const float thrT = ub_active ? ub_T : gs_key_score(s.res_min);
                        const int32_t part = gs_ubr_half<CH16>(ubtab, w, hi ? 1 : 0, VSF == 0);   // all 64 lanes
                        const int32_t other = gs_shfl32(part, lane ^ 32);
                        const float braw = ubr_base + ubr_scale * (float)(part + other);
                        bool drop = false;
                        if (fresh) drop = gs_bound_below<VSF>(braw, node_mag, query_mag, thrT);
                        if constexpr (DEFER) {
                            if (ub_up) {
                                float U = 0.0f;
                                if (drop) {
                                    U = gs_bound_score<VSF>(braw, node_mag, query_mag);
                                    drop = U < thrT && gs_bound_below<VSF>(braw, node_mag, query_mag, U);
                                }
                                const long long mk = gs_wave_max(drop ? gs_key(nb, U) : GS_KEY_MIN);
                                if (mk > d_maxk) d_maxk = mk;
                            }
                        }
                        const uint64_t dm = gs_ballot(drop);
                        sm = fm & ~dm;
                        if (ub_active) ub_dropped += (unsigned long long)gs_popc(dm);
                        else d_deferred += (uint32_t)gs_popc(dm);
                    }
                    const int ns = gs_popc(sm);
                    ubr_since += ns;
                    if (PROF) {
                        py[4] += ns <= 4 ? 1 : 0;
                        py[5] += ns <= 8 ? 1 : 0;
                        py[6] += ns <= 16 ? 1 : 0;
                    }
                    // ---- compact the survivors: code bytes, node id and magnitude of survivor j (row order) into LDS ----
                    float *xf = xchg;                                              // [8 owners][7 SUBS] entries handed to the owner lanes
                    int32_t *st_nb = reinterpret_cast<int32_t *>(xchg + 7 * M_);   // [32]
                    float *st_mag = reinterpret_cast<float *>(st_nb + 32);          // [32]
                    uint8_t *st_code = reinterpret_cast<uint8_t *>(st_mag + 32);    // [32][M]
                    if ((sm >> ni) & 1ull) {
                        const int j = gs_popc(sm & ((1ull << ni) - 1ull));
                        gs_u2 *dst = reinterpret_cast<gs_u2 *>(st_code + j * M_ + (hi ? M_ / 2 : 0));
#pragma unroll
                        for (int c = 0; c < CH16; ++c) dst[c] = w[c];
                        if (!hi) {
                            st_nb[j] = nb;
                            st_mag[j] = node_mag;
                        }
                    }
                    gs_barrier();
                    if (PROF) {
                        const unsigned long long now_ = GS_CLOCK();
                        fh[0] += now_ - ubc0;   // (the UBR build reuses the fresh-count histogram's slots: bound + staging clocks,
                        fh[3] += (unsigned long long)ns;   // ... survivors)
                    }
                    // ---- score them: 8, 16 or 32 lanes per survivor (gs_ubr_pass), 8 / 4 / 2 survivors per pass ----
                    fresh = false;
                    key = 0;
                    bool give_up = false;
                    // (32 / 16 lanes per survivor when at most 2 / 4 survive — two thirds of the headline's expansions — is one round of
                    // codebook requests instead of two: 47.8 vs 48.8 ms per 131 072 queries, profiles/r6_h.  GS_UBR_VAR_LPS = 0: always 8.)
#if GS_UBR_VAR_LPS
                    const int per = ns <= 2 ? 2 : (ns <= 4 ? 4 : 8);
#else
                    const int per = 8;
#endif
#pragma unroll 1
                    for (int base = 0; base < ns; base += per) {
                        unsigned long long pyc = 0;
                        if (PROF) {
                            pyc = GS_CLOCK();
                            py[3]++;
                            if (prof_upper) pzp[1]++;
                        }
#if GS_UBR_VAR_LPS
                        if (ns <= 2) gs_ubr_pass<VSF, M_, 32>(p.codebooks, qs, xf, st_nb, st_mag, st_code, base, ns, query_mag, ub_active, ub_T, fresh, key);
                        else if (ns <= 4) gs_ubr_pass<VSF, M_, 16>(p.codebooks, qs, xf, st_nb, st_mag, st_code, base, ns, query_mag, ub_active, ub_T, fresh, key);
                        else
#endif
                        gs_ubr_pass<VSF, M_, 8>(p.codebooks, qs, xf, st_nb, st_mag, st_code, base, ns, query_mag, ub_active, ub_T, fresh, key);
                        if (PROF) {
                            const uint64_t done_ = gs_ballot(fresh && key != 0);   // (the owners' keys exist before the clock is read)
                            if (done_ == 0xdeadbeefdeadbeefull) py[3]++;
                            py[1] += GS_CLOCK() - pyc;
                            if (prof_upper) pzp[0] += GS_CLOCK() - pyc;
                        }
                        if (base + per >= ns) break;   // the shared tail below pushes the last pass
                        gs_barrier();   // every owner lane has read its column before the push's sample buffer reuses the bytes
                        if (ub_on && gs_ballot(fresh && (int32_t)(key >> 32) == 0x7fc00000)) ub_on = false;
                        gs_push(s, p, key, fresh);
                        fresh = false;
                        if (s.status != GS_OK) {
                            give_up = true;
                            break;
                        }
                    }
                    if (give_up) break;
                    if (PROF) fh[1] += GS_CLOCK() - ubc0;
                } else {
                uint64_t fm_score = fm;   // fresh neighbours that get an exact score
                // ---- GsParams::quad (off by default): at most 16 fresh neighbours -> FOUR lanes each, spread over all 64 lanes: lane
                //      16 t + g takes subspaces [t M/4, (t + 1) M/4) of the g-th fresh neighbour (row order), its code words come
                //      from the pair lanes that loaded them (ds_bpermute), lanes 0 ... 15 add all M entries in ascending m.  Half
                //      the gather INSTRUCTIONS for 61 % of the headline's expansions, the same number of lane addresses — measured
                //      6 % slower (graph_search.cpp gs_quad): what one lane per neighbour loses against pair lanes (33.7 vs 19.0
                //      ms) is the length of the dependent per-lane chain, not a per-instruction charge of the memory path.
#ifdef JV_EXPERIMENTAL   // (gs_quad is a measured-and-switched-off variant: experimental builds and the CPU test harnesses only)
                constexpr bool QUAD_OK = CH16 % 2 == 0;   // (a lane's M/4 code bytes are whole 8-byte words)
#else
                constexpr bool QUAD_OK = false;
#endif
                bool quad = false;
                if constexpr (QUAD_OK) quad = p.quad != 0 && gs_popc(fm) <= 16;
                if (quad) {
                    if constexpr (QUAD_OK) {
                        constexpr int QW = CH16 / 2;   // 8-byte code words per lane
                        constexpr int QS = QW * 8;     // subspaces per lane
                        const int g = lane & 15;
                        int t = lane >> 4;
                        GS_OPAQUE_I32(t);   // (or 24 loop-invariant subspace pointers are hoisted out of the search loop and spilled)
                        const int nfq = gs_popc(fm);
                        int32_t *cmp = reinterpret_cast<int32_t *>(xchg);   // (read back before the exchange area is written)
                        if (fresh) cmp[gs_popc(fm & ((1ull << lane) - 1ull))] = lane;
                        gs_barrier();
                        const int src = g < nfq ? cmp[g] : 0;               // the low lane that holds the g-th fresh neighbour
                        gs_barrier();
                        const int from = src + 32 * (t >> 1);               // ... and the lane that holds this quarter's code bytes
                        gs_u2 qw[QW];
#pragma unroll
                        for (int k = 0; k < QW; ++k) {
                            const uint32_t ax = (uint32_t)gs_shfl32((int32_t)w[k].x, from), ay = (uint32_t)gs_shfl32((int32_t)w[k].y, from);
                            const uint32_t bx = (uint32_t)gs_shfl32((int32_t)w[k + QW].x, from), by = (uint32_t)gs_shfl32((int32_t)w[k + QW].y, from);
                            qw[k].x = (t & 1) ? bx : ax;
                            qw[k].y = (t & 1) ? by : ay;
                        }
                        const int32_t qnb = gs_shfl32(nb, src);
                        const float qmag = gs_bits_float(gs_shfl32(gs_float_bits(node_mag), src));
                        const bool qwork = g < nfq;
                        float sum = 0.0f;
                        if (qwork) sum = gs_half_entries<VSF, QW, 16>(p.codebooks, qs, qw, t * QS, t ? xchg + (t - 1) * QS * 16 + g : nullptr);
                        gs_barrier();
                        fresh = qwork && t == 0;
                        if (fresh) {
#pragma unroll
                            for (int j = 0; j < 3 * QS; ++j) sum += xchg[j * 16 + g];
                            key = gs_key(qnb, gs_finish<VSF>(sum, qmag, query_mag));
                        }
                    }
                } else {
                const bool work = ((fm_score >> ni) & 1ull) != 0;  // this lane's pair has a fresh neighbour that needs its score
                float sum = 0.0f;
                if (work) sum = gs_half_entries<VSF, CH16>(p.codebooks, qs, w, m_base, hi ? xchg + ni : nullptr);
                gs_barrier();
                if (fresh) {  // low lane: its own subspaces [0, M/2) are summed; now the partner's [M/2, M) in order
#pragma unroll
                    for (int j = 0; j < CH16 * 8; ++j) sum += xchg[j * 32 + ni];
                    key = gs_key(nb, gs_finish<VSF>(sum, node_mag, query_mag));
                }
                }
                }   // (!UBR)
            } else if constexpr (PAIRC) {
                // ---- rows of up to 64 neighbours, codes by ordinal: one lane per neighbour for the visited probe, then the fresh
                //      ones — compacted in row order — LPN lanes each (lane t of a group: subspaces [t M/LPN, (t+1) M/LPN); lane 0
                //      adds all M entries in ascending m, the others hand theirs over through LDS).  LPN = 2 up to M = 96 (32
                //      neighbours per pass), 4 above (a lane's share of the row must fit its registers: 16 per pass).  The order of
                //      the pushes inside an expansion is immaterial (see below).
                constexpr int LPN = CH16 > 6 ? 4 : 2;
                constexpr int PER = 64 / LPN;            // neighbours per pass
                constexpr int HW = CH16 * 2 / LPN;       // 8-byte code words per lane
                constexpr int SUBS = HW * 8;             // subspaces per lane
                const int32_t nb0 = lane < deg ? row[lane] : -1;
                const int first_neg = gs_first(gs_ballot(nb0 < 0));  // rows are packed: the first -1 ends the row
                const bool fr0 = visit(lane < first_neg, nb0);
                if (s.status != GS_OK) break;
                const uint64_t fm = gs_ballot(fr0);
                if (fm == 0) continue;
                const int nf = gs_popc(fm);
                n_visited += nf;
                GS_PHASE(2);
                int32_t *cmp = reinterpret_cast<int32_t *>(xchg);   // (consumed into registers before the exchange area is written)
                if (fr0) cmp[gs_popc(fm & ((1ull << lane) - 1ull))] = nb0;
                gs_barrier();
                const int ni = lane & (PER - 1);
                const int sub = lane / PER;              // 0: the lane that owns the neighbour's sum
                const int m_base = sub * SUBS;
                int32_t cn_[LPN];
#pragma unroll
                for (int t = 0; t < LPN; ++t) cn_[t] = t * PER + ni < nf ? cmp[t * PER + ni] : -1;
                gs_barrier();
                bool give_up = false;
#pragma unroll 1
                for (int pass = 0; pass < LPN; ++pass) {
                    int32_t cn = cn_[0];
#pragma unroll
                    for (int t = 1; t < LPN; ++t) cn = pass == t ? cn_[t] : cn;
                    const bool work = cn >= 0;
                    if constexpr (UBR) {
                        // ---- UBR over the compacted list (the builder's searches): a pass's <= 32 fresh neighbours are laid out like the
                        //      pair form's row — lane ni low half, lane ni + 32 high half of the code — so the bound, the staging of
                        //      the survivors and their eight-lane scoring are the pair form's (above), per pass ----
                        constexpr int M_ = CH16 * 16;
                        const bool hi = sub != 0;
                        const bool last_pass = (pass + 1) * PER >= nf;
                        gs_u2 w[CH16];
#pragma unroll
                        for (int c = 0; c < CH16; ++c) w[c] = gs_u2{0u, 0u};
                        float node_mag = 0.0f;
                        if (work) {
                            gs_load_half<CH16>(p.codes + (int64_t)cn * p.M + m_base, w);
                            if (VSF == 2 && !hi) node_mag = p.code_norms[cn];
                        }
                        const bool lowf = work && !hi;
                        const uint64_t fmp = gs_ballot(lowf);
                        uint64_t sm = fmp;
                        const bool ub_active = ub_on && lvl == 0 && ub_T > -__builtin_inff();
                        if (ub_active) {
                            const int32_t part = gs_ubr_half<CH16>(ubtab, w, hi ? 1 : 0, VSF == 0);   // all 64 lanes
                            const int32_t other = gs_shfl32(part, lane ^ 32);
                            bool drop = false;
                            if (lowf) drop = gs_bound_below<VSF>(ubr_base + ubr_scale * (float)(part + other), node_mag, query_mag, ub_T);
                            const uint64_t dm = gs_ballot(drop);
                            sm = fmp & ~dm;
                            ub_dropped += (unsigned long long)gs_popc(dm);
                        }
                        const int ns = gs_popc(sm);
                        ubr_since += ns;
                        float *xf = xchg;
                        int32_t *st_nb = reinterpret_cast<int32_t *>(xchg + 7 * M_);
                        float *st_mag = reinterpret_cast<float *>(st_nb + 32);
                        uint8_t *st_code = reinterpret_cast<uint8_t *>(st_mag + 32);
                        if ((sm >> ni) & 1ull) {
                            const int j = gs_popc(sm & ((1ull << ni) - 1ull));
                            gs_u2 *dst = reinterpret_cast<gs_u2 *>(st_code + j * M_ + (hi ? M_ / 2 : 0));
#pragma unroll
                            for (int c = 0; c < CH16; ++c) dst[c] = w[c];
                            if (!hi) {
                                st_nb[j] = cn;
                                st_mag[j] = node_mag;
                            }
                        }
                        gs_barrier();
                        // (round 6: the scoring passes are the pair form's — gs_ubr_pass: batched LDS reads, 32 / 16 / 8 lanes per survivor)
                        fresh = false;
                        key = 0;
#if GS_UBR_VAR_LPS
                        const int per = ns <= 2 ? 2 : (ns <= 4 ? 4 : 8);
#else
                        const int per = 8;
#endif
#pragma unroll 1
                        for (int base = 0; base < ns; base += per) {
#if GS_UBR_VAR_LPS
                            if (ns <= 2) gs_ubr_pass<VSF, M_, 32>(p.codebooks, qs, xf, st_nb, st_mag, st_code, base, ns, query_mag, ub_active, ub_T, fresh, key);
                            else if (ns <= 4) gs_ubr_pass<VSF, M_, 16>(p.codebooks, qs, xf, st_nb, st_mag, st_code, base, ns, query_mag, ub_active, ub_T, fresh, key);
                            else
#endif
                            gs_ubr_pass<VSF, M_, 8>(p.codebooks, qs, xf, st_nb, st_mag, st_code, base, ns, query_mag, ub_active, ub_T, fresh, key);
                            if (base + per >= ns && last_pass) break;   // the shared tail below pushes the expansion's last scores
                            gs_barrier();
                            if (ub_on && gs_ballot(fresh && (int32_t)(key >> 32) == 0x7fc00000)) ub_on = false;
                            gs_push(s, p, key, fresh);
                            fresh = false;
                            if (s.status != GS_OK) {
                                give_up = true;
                                break;
                            }
                        }
                        if (give_up || last_pass) break;
                        continue;
                    }
                    gs_u2 w[HW];
                    float node_mag = 0.0f, sum = 0.0f;
                    if (work) {   // PQDecoder.similarityTo: the neighbour's own code
                        gs_load_half<HW>(p.codes + (int64_t)cn * p.M + m_base, w);
                        if (VSF == 2 && sub == 0) node_mag = p.code_norms[cn];
                        sum = gs_half_entries<VSF, HW, PER>(p.codebooks, qs, w, m_base, sub ? xchg + (sub - 1) * SUBS * PER + ni : nullptr);
                    }
                    gs_barrier();
                    fresh = work && sub == 0;
                    key = 0;
                    if (fresh) {
#pragma unroll
                        for (int j = 0; j < (LPN - 1) * SUBS; ++j) sum += xchg[j * PER + ni];
                        key = gs_key(cn, gs_finish<VSF>(sum, node_mag, query_mag));
                    }
                    if ((pass + 1) * PER >= nf) break;   // the shared tail below pushes the last pass
                    gs_barrier();   // every owner lane has read its column before the push's sample buffer reuses the bytes
                    if constexpr (SES) {
                        if (thr_on) trk_track(fresh, gs_key_score(key));
                    }
                    gs_push(s, p, key, fresh);
                    fresh = false;
                    if (s.status != GS_OK) {
                        give_up = true;
                        break;
                    }
                }
                if (give_up) break;
            } else {
                // ---- one lane per neighbour, 64 neighbours at a time (degrees above 64: the next chunk of the row; inside an
                //      expansion the order of visited.mark / push / track calls is immaterial — sets and a priority queue) ----
                bool give_up = false;
                for (int c0 = 0; c0 < deg; c0 += 64) {
                    const int li = c0 + lane;
                    const int32_t nb = li < deg ? row[li] : -1;
                    // code bytes first, then the visited probes: the loads do not depend on the probes' outcome
                    gs_u4 w[CW];
                    const uint8_t *rp = nullptr;  // generic form: where the lane's code row lives
                    (void)w;
                    (void)rp;
                    float node_mag = 0.0f;
                    if (fused0 && li < deg) {  // FusedPQDecoder.similarityToNeighbor: the origin's packed block (zero padded)
                        const int64_t r = (int64_t)node * p.deg0 + li;
                        if constexpr (CH16 == 0) rp = p.blocks + r * p.M;
                        else gs_load_row<CW>(p.blocks + r * p.M, w);
                        if (VSF == 2) node_mag = p.fused_norms[r];
                    }
                    const int first_neg = gs_first(gs_ballot(nb < 0));  // rows are packed: the first -1 ends the row
                    const bool valid = lane < first_neg;
                    if (!fused0 && valid) {      // PQDecoder.similarityTo: the neighbour's own code
                        if constexpr (CH16 == 0) rp = p.codes + (int64_t)nb * p.M;
                        else gs_load_row<CW>(p.codes + (int64_t)nb * p.M, w);
                        if (VSF == 2) node_mag = p.code_norms[nb];
                    }
                    fresh = visit(valid, nb);
                    if (s.status != GS_OK) {
                        give_up = true;
                        break;
                    }
                    const uint64_t fm = gs_ballot(fresh);
                    const bool last = first_neg < 64 || c0 + 64 >= deg;  // the row ends inside this chunk
                    if (fm == 0) {
                        if (last) break;
                        continue;
                    }
                    n_visited += gs_popc(fm);
                    GS_PHASE(2);
                    key = 0;
                    if constexpr (CH16 == 0) {
                        if (fresh) key = gs_key(nb, gs_finish<VSF>(gs_row_sum_any<VSF>(p, qs, rp), node_mag, query_mag));
                    } else {
                        if (fresh) key = gs_key(nb, gs_finish<VSF>(gs_row_sum<VSF, CW>(p.codebooks, qs, w), node_mag, query_mag));
                    }
                    if (last) break;  // (the common case, every degree <= 64: the shared tail below pushes this chunk)
                    if constexpr (SES) {
                        if (thr_on) trk_track(fresh, gs_key_score(key));
                    }
                    gs_push(s, p, key, fresh);
                    fresh = false;
                    if (s.status != GS_OK) {
                        give_up = true;
                        break;
                    }
                }
                if (give_up) break;
                if (gs_ballot(fresh) == 0) continue;  // nothing left for the tail (no fresh neighbour in the last chunk)
            }
            if (PROF) {  // the scores must have arrived before the phase is closed
                const uint64_t done_ = gs_ballot(fresh && key != 0);
                if (done_ == 0xdeadbeefdeadbeefull) n_visited++;
            }
            GS_PHASE(3);
            if constexpr (SES) {
                if (thr_on) trk_track(fresh, gs_key_score(key));
            }
            if constexpr (UBR) {   // a NaN score sorts above everything but never becomes a result: no threshold can be proven with one around
                if (ub_on && gs_ballot(fresh && (int32_t)(key >> 32) == 0x7fc00000)) ub_on = false;
            }
            gs_push(s, p, key, fresh);
            GS_PHASE(4);
            if (s.status != GS_OK) break;
        }
        if (!(SES && lvl == 0 && s.status == GS_OK && phase + 1 < p.n_phases)) break;
        if constexpr (SES) {
            // ---- the call returned and resume() was called: reranking() drained approximateResults (:471-507), searchLayer0 puts
            //      evictedResults back into the candidates (:459-469), the counters of SearchResult restart (:541-545) ----
            s.res_n = 0;
            s.res_min = GS_KEY_MAX;
            s.res_min_idx = -1;
            const int32_t *off = p.ph_extra_off + (int64_t)phase * p.ph_Q + q;
            const int e0 = off[0], e1 = off[1];
            for (int base = e0; base < e1 && s.status == GS_OK; base += 64) {
                const bool has = base + lane < e1;
                gs_push(s, p, has ? p.ph_extra[base + lane] : 0, has);
            }
            phase++;
            rk = p.ph_rerankK[phase];
            cur_thr = p.ph_threshold[phase];
            thr = cur_thr;
            thr_on = cur_thr > 0.0f;
            trk_obs = 0;
            trk_nbest = 0;
            trk_min = 0x7fffffff;
            trk_min_idx = -1;
            n_visited = 0;
            n_expanded = 0;
            n_expanded_base = 0;
            n_log = 0;
            gs_barrier();
        }
        }
        if (s.status != GS_OK) break;
        unsigned long long ptr0 = 0;
        if (PROF) ptr0 = GS_CLOCK();
        if (lvl > 0) {  // setEntryPointsFromPreviousLayer :324-331
            for (int base = 0; base < s.res_n && s.status == GS_OK; base += 64) {
                const bool has = base + lane < s.res_n;
                gs_push(s, p, has ? s.res[base + lane] : 0, has);
            }
            for (int base = 0; base < s.ev_n && s.status == GS_OK; base += 64) {
                const bool has = base + lane < s.ev_n;
                gs_push(s, p, has ? s.evicted[base + lane] : 0, has);
            }
            s.res_n = 0;
            s.ev_n = 0;
            s.res_min = GS_KEY_MAX;
            s.res_min_idx = -1;
        }
        if (PROF) px[1] += GS_CLOCK() - ptr0;
        if (PROF && lvl > 0) {
            pz[0] += (unsigned long long)(n_expanded - plv_e0);
            pz[1] += GS_CLOCK() - plv0;
        }
    }
    if constexpr (DEFER) {
        if (s.status == GS_RESTART) {   // nothing has left the wave yet (results, counters and status are written below)
            gs_barrier();
            if (p.defer_count && lane == 0) gs_fetch_add64(p.defer_count + 1, 1ull);
            return true;
        }
    }
    unsigned long long pep0 = 0;
    if (PROF) pep0 = GS_CLOCK();

    // ---- hand the kept approximate results to the rerank stage ----
    gs_barrier();
    // (round 6: rows [0, rr_rows) get their EXACT similarity right here, below — their ordinals leave LDS first: the tile overwrites it)
    // (compiled into the register-table bound form over the row — the headline's kernel — ONLY: in every other instantiation the mere
    //  presence of the call cost the expansion loop registers — the compacted pair kernels went from 0 to 95 spilled VGPRs and C5's
    //  builder searches from 17.6 to 30 s, profiles/r6_final — and those forms never asked for it)
    constexpr bool RR = UBR && PAIR && !PAIRC && !SES;
    const bool rr = RR && p.rr_vecs != nullptr;
    int32_t rr_node[GS_RR_MAX_ROUNDS];
#pragma unroll
    for (int r = 0; r < GS_RR_MAX_ROUNDS; ++r) {
        const int i = 64 * r + lane;
        rr_node[r] = (rr && s.status == GS_OK && i < s.res_n && i < p.rr_rows) ? gs_key_node(s.res[i]) : -1;
    }
    for (int i = lane; i < p.rerankK; i += 64) {
        const bool have = s.status == GS_OK && i < s.res_n;
        const long long k = have ? s.res[i] : 0;
        p.out_ids[(int64_t)q * p.rerankK + i] = have ? gs_key_node(k) : -1;
        if (!(rr && i < p.rr_rows)) p.out_scores[(int64_t)q * p.rerankK + i] = have ? gs_key_score(k) : -__builtin_inff();
    }
    if constexpr (RR) if (rr) {
        gs_barrier();   // every lane has read its keys: the LDS block is the tile's now
        const float *qraw = p.rr_queries + (int64_t)q * p.D;
        for (int r = 0; 64 * r < p.rr_rows; ++r) {
            int32_t o = rr_node[0];   // (a select chain, not an indexed read: the array stays in registers)
#pragma unroll
            for (int t = 1; t < GS_RR_MAX_ROUNDS; ++t) o = r == t ? rr_node[t] : o;
            if ((long long)o >= p.rr_n) o = -1;
            const float raw = gs_rr_round<VSF>(p.rr_vecs, p.D, qraw, o, reinterpret_cast<float *>(lds));
            const int i = 64 * r + lane;
            if (i < p.rr_rows)
                p.out_scores[(int64_t)q * p.rerankK + i] =
                    o < 0 ? -__builtin_inff() : gs_finish<VSF>(raw, VSF == 2 ? p.rr_vnorm[o] : 0.0f, VSF == 2 ? p.rr_qnorm[q] : 0.0f);
        }
    }
    if (lane == 0) {
        p.out_stats[2 * (int64_t)q] = n_visited;
        p.out_stats[2 * (int64_t)q + 1] = n_expanded;
        p.out_status[q] = s.status;
        if (SES && p.out_base) p.out_base[q] = (int32_t)n_expanded_base;
        if (p.push_log) p.push_log_n[q] = s.status == GS_OK ? n_log : -1;
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
        if (UBR) for (int i = 0; i < 7; ++i) gs_fetch_add64(p.prof + 16 + i, py[i]);
        gs_fetch_add64(p.prof + 24, pz[0]);
        gs_fetch_add64(p.prof + 25, pz[1]);
        for (int i = 0; i < 5; ++i) gs_fetch_add64(p.prof + 26 + i, pzf[i]);
        gs_fetch_add64(p.prof + 31, pzp[0]);
        gs_fetch_add64(p.prof + 32, pzp[1]);
    }
    if (UBR && p.ubr_count && lane == 0) gs_fetch_add64(p.ubr_count, ub_dropped);
    if constexpr (DEFER) {
        if (p.defer_count && lane == 0) gs_fetch_add64(p.defer_count, (unsigned long long)d_deferred);
    }
#undef GS_PHASE
    return false;
}

// Persistent worker: pulls queries off the shared counter until none are left.
template <int VSF, int CH16, bool PAIR, bool PROF = false, bool SES = false, bool PAIRC = false, bool UBR = false>
GS_FN void gs_worker(const GsParams &p, int worker, char *lds)
{
    // (measured and not kept, profiles/r6_z2: work items dealt per XCD — eight ranges, stolen when one runs dry — so that a batch ordered
    //  by locality keeps neighbouring searches under one L2: no gain, 0.2 - 0.5 ms worse; what ordering buys comes from the whole chip
    //  walking one region at a time, not from one L2)
    for (;;) {
        long long qv = 0;
        if (gs_lane() == 0) qv = (long long)gs_fetch_add(p.next_query, 1u);
        const int item = (int)gs_shfl(qv, 0);
        if (item >= p.Q) break;
        // (DEFER: a query whose deferred neighbours turned out to matter starts over with every neighbour scored when it is met)
        bool again = false;
        do {
            again = gs_search_one<VSF, CH16, PAIR, PROF, SES, PAIRC, UBR>(p, p.qmap ? p.qmap[item] : item, worker, lds, again);
        } while (again);
    }
}

}  // namespace jv
