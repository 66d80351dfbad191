// rd_body.h — batched robust prune: VamanaDiversityProvider.retainDiverse (B/graph/diversity/VamanaDiversityProvider.java:
// 43-96) for P nodes at once, with the PQ diversity score of BuildScoreProvider.pqBuildScoreProvider
// (ImmutablePQVectors.diversityFunctionFor, B/quantization/ImmutablePQVectors.java:61-104: assembleAndSumPQ on the
// triangular centroid-pair table).  SURVEY §8 f.2 / BASELINE config 5 ("GPU-batched neighbor scoring").
//
// One 64-lane wavefront per node.  The reference walks the candidates (sorted by score, descending) in order, once per alpha
// step (1.0, 1.2, ... <= alpha + 1e-6), and keeps candidate c iff no ALREADY selected neighbour s has
// similarity(c, s) > score(c) * alpha.  That outer walk is sequential by definition; what runs in parallel is the test of one
// candidate against all selected neighbours — lane j owns selected slot j (maxDegree <= 64) and sums its M table entries in
// ascending m into one f32, the association of assembleAndSumPQ (DefaultVectorUtilSupport.java:312-335) — followed by one
// vote.  isDiverse (:82-96) walks the selected set in ascending candidate INDEX and stops at the first event (the candidate
// itself -> diverse, a violation -> not diverse); the vote reproduces that with a minimum over the event lanes' indices.
// LDS: the candidates' code rows (C x M bytes, staged once), the selected neighbours' codes transposed in 4-byte words ([m / 4][slot],
// so the 64 lanes read 64 consecutive words), cosine self-magnitudes.  Shared source: compiled for gfx950 through gs_wave_hip.h and for
// the CPU lane emulator (tests/emu/rd_emu.cpp).  Wave API: gs_lane, gs_barrier, gs_ballot, gs_shfl, gs_shfl32, gs_shfl_xor, gs_sqrt.
#pragma once

#include <cstdint>

#include "rd_params.h"

namespace jv {

// the incremental (chunked) test walk is a measured-and-switched-off variant (rd_chunk): compiled only into
// experimental builds (make EXPERIMENTAL=1) and into the CPU test harnesses
#ifdef JV_EXPERIMENTAL
constexpr bool RD_HAVE_CHUNK = true;
#else
constexpr bool RD_HAVE_CHUNK = false;
#endif

GS_FN int64_t rd_tri_row(int r, int k) { return (int64_t)r * k - ((int64_t)r * (r - 1)) / 2; }

// assembleAndSumPQ of (candidate row in LDS, this lane's selected slot column in LDS).  The M table entries are independent
// loads (L2 / Infinity Cache latency each) feeding one sequential f32 sum: they are fetched 16 at a time and only then added,
// in ascending m, so the latency is paid once per 16 entries instead of once per entry.  Codes come four to an LDS word (crow4:
// the candidate's row, the same word for every lane; scol4: this lane's slot, word w at scol4[w * 64]) and the entry index is
// 32-bit arithmetic (M k (k + 1) / 2 < 2^31): round 4 — the byte-wise LDS reads with a wait behind each and the 64-bit index
// arithmetic of rounds 2 - 3 were ~21 instructions per entry, and the kernel was bound by the issue of exactly those.
GS_FN uint32_t rd_tri_index(uint32_t m_base, uint32_t k, uint32_t c1, uint32_t c2)
{
    const uint32_t r = c1 < c2 ? c1 : c2, c = c1 < c2 ? c2 : c1;
    return m_base + r * k - ((r * (r - 1u)) >> 1) + (c - r);   // (r = 0: 0 * 0xFFFFFFFF = 0)
}
// SQ: the square table [M][k][k], row = the candidate's code (wave-uniform inside a test), column = the slot's (M k k < 2^31)
template <bool SQ>
GS_FN uint32_t rd_index(uint32_t m, uint32_t block, uint32_t k, uint32_t c1, uint32_t c2)
{
    if (SQ) return (m * k + c1) * k + c2;
    return rd_tri_index(m * block, k, c1, c2);
}

template <bool SQ = false>
GS_FN float rd_pair_sum(const float *tri, int M, int k, const uint32_t *crow4, const uint32_t *scol4 /* stride 64 words */)
{
    const uint32_t block = (uint32_t)k * ((uint32_t)k + 1u) / 2u;
    float res = 0.0f;
    int m = 0;
    for (; m + 16 <= M; m += 16) {
        uint32_t cw[4], sw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            cw[q] = crow4[(m >> 2) + q];
            sw[q] = scol4[(size_t)((m >> 2) + q) * 64];
        }
        float e[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t c1 = (cw[j >> 2] >> (8 * (j & 3))) & 0xFFu, c2 = (sw[j >> 2] >> (8 * (j & 3))) & 0xFFu;
            e[j] = tri[rd_index<SQ>((uint32_t)(m + j), block, (uint32_t)k, c1, c2)];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) res += e[j];
    }
    for (; m < M; ++m) {
        const uint32_t c1 = (crow4[m >> 2] >> (8 * (m & 3))) & 0xFFu, c2 = (scol4[(size_t)(m >> 2) * 64] >> (8 * (m & 3))) & 0xFFu;
        res += tri[rd_index<SQ>((uint32_t)m, block, (uint32_t)k, c1, c2)];
    }
    return res;
}

// The same sum with the LANES THE TEST LEAVES IDLE (RdParams::split).  A prune is a chain of ~200 dependent tests run by one wave, a
// test keeps one lane per selected slot busy (a dozen on average, 32 at most) with M entries each: the length of that per-lane
// chain is the kernel's time.  With `parts` lanes per slot (lane q S + j: 16-entry blocks [q bpl, (q + 1) bpl) of
// slot j, S = 64 / parts) the chain is M / parts entries; the sum stays the one sequential f32 chain in ascending m: lane j takes its
// own entries, then lane S + j's, ... through ds_bpermute (no LDS storage: the entries wait in registers).  M a multiple of 16, one
// to three blocks per lane.  Called by ALL lanes; the result is valid on lanes < nSlots.
GS_FN int rd_split_parts(int M, int nSlots)
{
    if (M % 16 != 0 || nSlots <= 0) return 1;
    const int nb = M / 16;
    for (int parts = 6; parts >= 2; --parts)
        if (parts != 5 && nb % parts == 0 && nb / parts <= 3 && nSlots * parts <= 64) return parts;
    return 1;
}

template <bool SQ = false>
GS_FN float rd_pair_sum_split(const float *tri, int M, int k, const uint32_t *crow4, const uint32_t *st4, int lane, int nSlots, int parts)
{
    const int S = 64 / parts;
    const int part = lane / S, slot = lane - part * S;
    const bool act = part < parts && slot < nSlots;
    const int bpl = (M / 16) / parts;   // blocks per lane: 1 ... 3
    const uint32_t block = (uint32_t)k * ((uint32_t)k + 1u) / 2u;
    const uint32_t *scol4 = st4 + (act ? slot : 0);
    float e[3][16];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
#pragma unroll
        for (int j = 0; j < 16; ++j) e[b][j] = 0.0f;
        if (b < bpl && act) {
            const int m = (part * bpl + b) * 16;
            uint32_t cw[4], sw[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                cw[q] = crow4[(m >> 2) + q];
                sw[q] = scol4[(size_t)((m >> 2) + q) * 64];
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t c1 = (cw[j >> 2] >> (8 * (j & 3))) & 0xFFu, c2 = (sw[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                e[b][j] = tri[rd_index<SQ>((uint32_t)(m + j), block, (uint32_t)k, c1, c2)];
            }
        }
    }
    float res = 0.0f;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        if (b < bpl) {
#pragma unroll
            for (int j = 0; j < 16; ++j) res += e[b][j];
        }
    }
    for (int q = 1; q < parts; ++q) {
        const int src = (slot + q * S) & 63;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            if (b < bpl) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    int32_t bits;
                    __builtin_memcpy(&bits, &e[b][j], 4);
                    bits = gs_shfl32(bits, src);
                    float v;
                    __builtin_memcpy(&v, &bits, 4);
                    res += v;
                }
            }
        }
    }
    return res;
}

// The same sum TABLE-FREE (uniform 8-dimensional sub-vectors): entry (m, c1, c2) is recomputed with the arithmetic that filled the
// table (bs_body.h bs_pair_table_row: one f32 chain over the 8 dimensions, a[d] * b[d] or (a[d] - b[d])^2 — both symmetric in the two
// centroids, so which of them has the smaller index does not matter) from this lane's selected neighbour's centroid (two 16-byte
// gathers into the L2-resident codebook, 768 KB at PQ-96 / 1.5 MB at PQ-192) and the candidate's, decoded once per test into LDS
// (cvec, broadcast reads).  The pair table (12.6 / 25 MB) misses L2 four times out of five; the codebook does not —
// and the form is still 1.4 - 1.7x SLOWER on the MI355X (profiles/r4_t): selectable (rd_table_free = 1), off by default.
struct alignas(16) rd_f4 { float x, y, z, w; };
struct alignas(16) rd_u4 { uint32_t x, y, z, w; };

template <bool L2>
GS_FN float rd_entry_tf(const rd_f4 &a0, const rd_f4 &a1, const rd_f4 &c0, const rd_f4 &c1)
{
    float v = 0.0f;
    if (L2) {
        float t;
        t = a0.x - c0.x; v += t * t;
        t = a0.y - c0.y; v += t * t;
        t = a0.z - c0.z; v += t * t;
        t = a0.w - c0.w; v += t * t;
        t = a1.x - c1.x; v += t * t;
        t = a1.y - c1.y; v += t * t;
        t = a1.z - c1.z; v += t * t;
        t = a1.w - c1.w; v += t * t;
    } else {
        v += a0.x * c0.x;
        v += a0.y * c0.y;
        v += a0.z * c0.z;
        v += a0.w * c0.w;
        v += a1.x * c1.x;
        v += a1.y * c1.y;
        v += a1.z * c1.z;
        v += a1.w * c1.w;
    }
    return v;
}

template <bool L2>
GS_FN float rd_pair_sum_tf(const float *cb, int M, int k, const float *cvec, const uint32_t *scol4 /* stride 64 words */)
{
    float res = 0.0f;
    int m = 0;
    for (; m + 8 <= M; m += 8) {   // 16 gathers in flight, then the 8 entries in ascending m
        rd_f4 a0[8], a1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t code = (scol4[(size_t)((m + j) >> 2) * 64] >> (8 * ((m + j) & 3))) & 0xFFu;
            const rd_f4 *r = reinterpret_cast<const rd_f4 *>(cb + ((int64_t)(m + j) * k + code) * 8);
            a0[j] = r[0];
            a1[j] = r[1];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const rd_f4 *c = reinterpret_cast<const rd_f4 *>(cvec + (size_t)(m + j) * 8);
            res += rd_entry_tf<L2>(a0[j], a1[j], c[0], c[1]);
        }
    }
    for (; m < M; ++m) {
        const uint32_t code = (scol4[(size_t)(m >> 2) * 64] >> (8 * (m & 3))) & 0xFFu;
        const rd_f4 *r = reinterpret_cast<const rd_f4 *>(cb + ((int64_t)m * k + code) * 8);
        const rd_f4 *c = reinterpret_cast<const rd_f4 *>(cvec + (size_t)m * 8);
        res += rd_entry_tf<L2>(r[0], r[1], c[0], c[1]);
    }
    return res;
}

template <bool SQ = false>
GS_FN float rd_self_sum(const float *tri, int M, int k, const uint8_t *crow)
{
    const uint32_t block = (uint32_t)k * ((uint32_t)k + 1u) / 2u;
    float res = 0.0f;
    int m = 0;
    for (; m + 16 <= M; m += 16) {   // 16 independent loads, then their sum in ascending m
        float e[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t c = crow[m + j];
            e[j] = tri[rd_index<SQ>((uint32_t)(m + j), block, (uint32_t)k, c, c)];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) res += e[j];
    }
    for (; m < M; ++m) {
        const uint32_t c = crow[m];
        res += tri[rd_index<SQ>((uint32_t)m, block, (uint32_t)k, c, c)];
    }
    return res;
}

// the largest lane value; NaN-free input (callers pass -inf for "nothing")
GS_FN float rd_wave_max_f(float v)
{
    for (int o = 32; o > 0; o >>= 1) {
        int32_t b;
        __builtin_memcpy(&b, &v, 4);
        b = (int32_t)gs_shfl_xor((long long)b, o);
        float t;
        __builtin_memcpy(&t, &b, 4);
        v = t > v ? t : v;
    }
    return v;
}

GS_FN int rd_wave_min(int v)
{
    for (int o = 32; o > 0; o >>= 1) {
        const int t = (int)gs_shfl_xor((long long)v, o);
        v = t < v ? t : v;
    }
    return v;
}

// lds: rd_lds_bytes(C, M, TF) bytes, 16-byte aligned.  TF: the table-free form (p.codebooks; the cosine self magnitudes still come
// from the table's diagonal: M x k entries, L2-resident)
// PROF (developer aid, rd_prof = 1): shader-clock totals per phase, added to p.prof at the end of every node —
//   [0] staging of ids / scores / code rows   [1] self magnitudes + init   [2] pre-selected prefix   [3] a test's sums (table
//   entries + ordered sum)   [4] a test's decision (wave minimum / ballot)   [5] take()   [6] loop bookkeeping   [7] output
//   [8] tests   [9] selected slots examined by them   [10] nodes   [11] candidates   [12] tests that ran split over idle lanes
#ifndef GS_CLOCK
#define GS_CLOCK() 0ull   // (the CPU lane emulator has no clock)
#endif
template <bool TF = false, bool PROF = false, bool SQ = false>
GS_FN void rd_node(const RdParams &p, int node_idx, char *lds, unsigned long long *work2 = nullptr /* += {tests, pairs} (wave-uniform) */)
{
    const float *table = SQ ? p.sq : p.tri;   // (SQ: the square form of the same entries)
    unsigned long long pf[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long pt = 0;
    if (PROF) pt = GS_CLOCK();
#define RD_PHASE(k)                                   \
    do {                                              \
        if (PROF) {                                   \
            const unsigned long long now_ = GS_CLOCK(); \
            pf[k] += now_ - pt;                       \
            pt = now_;                                \
        }                                             \
    } while (0)
    const int lane = gs_lane();
    const int M = p.M, C = p.C;
    uint8_t *cc = reinterpret_cast<uint8_t *>(lds);               // [C][M] candidate codes
    const int Mp = rd_row_bytes(M);                                // row stride: whole 4-byte words
    uint32_t *st = reinterpret_cast<uint32_t *>(cc + (size_t)C * Mp);   // [Mp/4][64] selected codes: word w of slot j at st[w * 64 + j]
    float *cnorm = reinterpret_cast<float *>(lds + (((size_t)C * Mp + (size_t)Mp * 64 + 15) & ~(size_t)15));  // [C]
    float *snorm = cnorm + C;                                      // [64]
    int32_t *sidx = reinterpret_cast<int32_t *>(snorm + 64);       // [64] candidate index of slot j
    int32_t *snode = sidx + 64;                                    // [64]
    int32_t *tested = reinterpret_cast<int32_t *>(lds + rd_off_tested(C, M));   // [C] leading selected slots candidate i has been tested against
    float *best = reinterpret_cast<float *>(tested + C);                        // [C] the largest similarity among them (-inf: none)
    int32_t *cid = reinterpret_cast<int32_t *>(best + C);                       // [C] the candidates' node ids   } copies of the global rows: a test
    float *csc = reinterpret_cast<float *>(cid + C);                            // [C] ... and scores             } starts without a global load
    float *cvec = reinterpret_cast<float *>(lds + rd_off_cvec(C, M));   // TF: [M][8] the candidate under test, decoded
    (void)cvec;
    const int32_t *nodes = p.cand_nodes + (int64_t)node_idx * C;
    const float *scores = p.cand_scores + (int64_t)node_idx * C;
    int n = p.cand_count ? p.cand_count[node_idx] : C;
    if (n > C) n = C;
    if (n < 0) n = 0;
    const int maxDegree = p.maxDegree;
    int diverseBefore = p.diverse_before ? p.diverse_before[node_idx] : 0;
    if (diverseBefore < 0) diverseBefore = 0;

    // ---- stage the candidates' ids, scores and code rows (and, for cosine, their self magnitudes) ----
    for (int i = lane; i < n; i += 64) {
        cid[i] = nodes[i];
        csc[i] = scores[i];
    }
    gs_barrier();
    if (p.wide_stage && (M & 15) == 0 && (reinterpret_cast<uintptr_t>(p.codes) & 15) == 0) {
        // (row, 16-byte piece) items over the lanes: every load independent of every other — the row-by-row loop below is n
        // dependent round trips (node id -> code bytes), which for 100 - 190 candidates was a large part of a prune's time
        const int cpr = M >> 4, items = n * cpr;
        for (int w = lane; w < items; w += 64) {
            const int i = w / cpr, c = w - i * cpr;
            const int32_t nd = cid[i];
            const bool ok = nd >= 0 && nd < p.n;   // (an id outside the store reads row 0 and is zeroed: no struct temporary — it went to scratch)
            const rd_u4 v = *reinterpret_cast<const rd_u4 *>(p.codes + (ok ? (int64_t)nd : 0) * M + 16 * c);
            uint32_t *dst = reinterpret_cast<uint32_t *>(cc + (size_t)i * Mp + 16 * c);
            dst[0] = ok ? v.x : 0u;
            dst[1] = ok ? v.y : 0u;
            dst[2] = ok ? v.z : 0u;
            dst[3] = ok ? v.w : 0u;
        }
    } else {
        for (int i = 0; i < n; ++i) {
            const int32_t nd = cid[i];
            const bool ok = nd >= 0 && nd < p.n;
            for (int b = lane; b < Mp; b += 64) cc[(size_t)i * Mp + b] = (ok && b < M) ? p.codes[(int64_t)nd * M + b] : (uint8_t)0;
        }
    }
    gs_barrier();
    RD_PHASE(0);
    if (p.vsf == 2)
        for (int i = lane; i < n; i += 64) cnorm[i] = rd_self_sum<SQ>(table, M, p.k, cc + (size_t)i * Mp);
    for (int i = lane; i < n; i += 64) {
        tested[i] = 0;
        best[i] = -__builtin_inff();
    }
    gs_barrier();
    RD_PHASE(1);

    // selected BitSet as two 64-bit words per 128 candidates would not cover C up to 1024: keep it as one bit per lane-chunk:
    // lane l holds the bits of candidates l, l + 64, l + 128, ... in `mine`
    unsigned long long mine = 0;  // bit t: candidate t * 64 + lane is selected
    int nSlots = 0;               // selected neighbours held in the slot arrays (== number of selected bits)
    auto take = [&](int i) {      // wave-uniform i
        if (lane == (i & 63)) mine |= 1ull << (i >> 6);
        if (lane == 0) {
            sidx[nSlots] = i;
            snode[nSlots] = cid[i];
            if (p.vsf == 2) snorm[nSlots] = cnorm[i];
        }
        for (int b = lane; b < Mp / 4; b += 64) st[(size_t)b * 64 + nSlots] = reinterpret_cast<const uint32_t *>(cc + (size_t)i * Mp)[b];
        nSlots++;
        gs_barrier();
    };
    {
        const int pre = diverseBefore < maxDegree ? diverseBefore : maxDegree;
        for (int i = 0; i < pre && i < n; ++i) take(i);
    }
    RD_PHASE(2);
    int nSelected = diverseBefore;
    float shortEdges = __builtin_nanf("");
    float currentAlpha = 1.0f;
    while ((double)currentAlpha <= (double)p.alpha + 1E-6 && nSelected < maxDegree) {
        for (int i = diverseBefore; i < n && nSelected < maxDegree; ++i) {
            const unsigned long long owner_bits = (unsigned long long)gs_shfl((long long)mine, i & 63);
            if ((owner_bits >> (i >> 6)) & 1ull) continue;
            const int32_t cNode = cid[i];
            const float cScore = csc[i];
            RD_PHASE(6);
            if (PROF) {
                pf[8] += 1;
                pf[9] += (unsigned long long)nSlots;
            }
            if (work2) {
                work2[0] += 1;
                work2[1] += (unsigned long long)nSlots;
            }
            // ---- isDiverse.  Events of the reference's walk over the selected set (ascending candidate index): the candidate itself
            //      -> diverse, a violation -> not diverse; the first event decides.  Lane j owns selected slot j.
            auto stage = [&]() {   // TF: the candidate's sub-vectors, decoded into LDS for this test
                if constexpr (TF) {
                    for (int idx = lane; idx < 2 * M; idx += 64) {
                        const int m = idx >> 1;
                        const rd_f4 *r = reinterpret_cast<const rd_f4 *>(p.codebooks + ((int64_t)m * p.k + cc[(size_t)i * Mp + m]) * 8);
                        reinterpret_cast<rd_f4 *>(cvec)[idx] = r[idx & 1];
                    }
                    gs_barrier();
                }
            };
            auto sim_from = [&](float sum) -> float {
                if (p.vsf == 0) return 1.0f / (1.0f + sum);
                if (p.vsf == 1) return (1.0f + sum) / 2.0f;
                const float prod = cnorm[i] * snorm[lane];
                const float cosine = sum / (float)gs_sqrt((double)prod);
                return (1.0f + cosine) / 2.0f;
            };
            auto sim_of = [&]() -> float {   // this lane's slot against candidate i
                float sum;
                if constexpr (TF) sum = p.vsf == 0 ? rd_pair_sum_tf<true>(p.codebooks, M, p.k, cvec, st + lane)
                                                   : rd_pair_sum_tf<false>(p.codebooks, M, p.k, cvec, st + lane);
                else sum = rd_pair_sum<SQ>(table, M, p.k, reinterpret_cast<const uint32_t *>(cc + (size_t)i * Mp), st + lane);
                return sim_from(sum);
            };
            bool not_diverse;
            const bool dup = gs_ballot(lane < nSlots && snode[lane] == cNode) != 0;
            if (dup || !RD_HAVE_CHUNK || p.chunk <= 0) {
                // every selected slot in parallel, then the first event in ascending candidate index (the candidate's node is in the
                // selected set — a node listed twice — or incremental tests are off)
                int ev_idx = 0x7fffffff;  // this lane's event index (none: INT_MAX)
                bool ev_fail = false;
                if (nSlots > 0) stage();   // (the previous test's readers are past their last wave-wide step)
                int parts = 1;
                if constexpr (!TF) parts = p.split ? rd_split_parts(M, nSlots) : 1;
                float split_sum = 0.0f;
                if (parts > 1)   // every lane takes part; lanes < nSlots end up with their slot's sum
                    split_sum = rd_pair_sum_split<SQ>(table, M, p.k, reinterpret_cast<const uint32_t *>(cc + (size_t)i * Mp), st, lane, nSlots, parts);
                float simv = 0.0f;
                const bool have = lane < nSlots && snode[lane] != cNode;
                if (have) simv = parts > 1 ? sim_from(split_sum) : sim_of();
                if (PROF) {   // the similarities must have arrived before the phase is closed
                    if (gs_ballot(have && simv == 123456.7890625f) == 0xdeadbeefdeadbeefull) pf[12] += 1000000;
                    if (parts > 1) pf[12] += 1;
                }
                RD_PHASE(3);
                if (lane < nSlots) {
                    if (!have) {
                        ev_idx = sidx[lane];
                    } else if (simv > cScore * currentAlpha) {
                        ev_idx = sidx[lane];
                        ev_fail = true;
                    }
                }
                const int first = rd_wave_min(ev_idx);
                not_diverse = gs_ballot(ev_fail && ev_idx == first && first != 0x7fffffff) != 0;
                RD_PHASE(4);
            } else {
                // No selected slot holds the candidate's node: it is diverse iff NO selected slot violates, whatever the order.  Slots
                // are only ever appended and a similarity does not depend on alpha, so what earlier tests of this candidate saw is
                // still true: `tested[i]` leading slots with the largest similarity `best[i]` among them (NaN similarities never
                // violate and never become the maximum).  Only the slots behind them are examined, `chunk` at a time, and the walk
                // stops at the first violation — the reference's own early exit; the second alpha pass re-tests nothing it knows.
                const float thr = cScore * currentAlpha;
                int t = tested[i];
                float mx = best[i];
                bool viol = mx > thr;
                bool staged = false;
                while (!viol && t < nSlots) {
                    if (!staged) {
                        stage();
                        staged = true;
                    }
                    const int hi = t + p.chunk < nSlots ? t + p.chunk : nSlots;
                    float sv = -__builtin_inff();
                    if (lane >= t && lane < hi) {
                        const float sim = sim_of();
                        if (sim == sim) sv = sim;
                    }
                    const float cm = rd_wave_max_f(sv);
                    if (cm > mx) mx = cm;
                    t = hi;
                    viol = mx > thr;
                }
                if (lane == 0) {
                    tested[i] = t;
                    best[i] = mx;
                }
                gs_barrier();
                not_diverse = viol;
                RD_PHASE(3);
            }
            if (!not_diverse) {
                if (nSlots < 64) take(i);
                nSelected++;
            }
            RD_PHASE(5);
        }
        if (currentAlpha == 1.0f) shortEdges = nSelected / (float)maxDegree;
        currentAlpha += 0.2f;
    }

    // ---- results: the selected candidate indices in ascending order (what `selected.nextSetBit` iterates) ----
    gs_barrier();
    {
        int out_n = 0;
        for (int base = 0; base < n; base += 64) {
            const bool sel = ((mine >> (base >> 6)) & 1ull) != 0 && base + lane < n;
            const unsigned long long m = gs_ballot(sel);
            const int pos = out_n + __builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (sel && pos < maxDegree) p.selected_out[(int64_t)node_idx * maxDegree + pos] = base + lane;
            out_n += __builtin_popcountll(m);
        }
        for (int j = (out_n < maxDegree ? out_n : maxDegree) + lane; j < maxDegree; j += 64) p.selected_out[(int64_t)node_idx * maxDegree + j] = -1;
    }
    if (lane == 0) {
        p.n_selected_out[node_idx] = nSelected;
        if (p.short_edges_out) p.short_edges_out[node_idx] = shortEdges;
    }
    gs_barrier();
    RD_PHASE(7);
    if (PROF) {
        pf[10] = 1;
        pf[11] = (unsigned long long)n;
        if (lane == 0 && p.prof)
            for (int k = 0; k < 13; ++k) gs_fetch_add64(p.prof + k, pf[k]);
    }
#undef RD_PHASE
}

}  // namespace jv
