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
// LDS: the candidates' code rows (C x M bytes, staged once), the selected neighbours' codes transposed ([m][slot], so the 64
// lanes read 64 consecutive bytes), cosine self-magnitudes.  Shared source: compiled for gfx950 through gs_wave_hip.h and for
// the CPU lane emulator (tests/emu/rd_emu.cpp).  Wave API: gs_lane, gs_barrier, gs_ballot, gs_shfl, gs_shfl_xor, gs_sqrt.
#pragma once

#include <cstdint>

#include "rd_params.h"

namespace jv {

GS_FN int64_t rd_tri_row(int r, int k) { return (int64_t)r * k - ((int64_t)r * (r - 1)) / 2; }

// assembleAndSumPQ of (candidate row in LDS, this lane's selected slot column in LDS).  The M table entries are independent
// loads (L2 / Infinity Cache latency each) feeding one sequential f32 sum: they are fetched 16 at a time and only then added,
// in ascending m, so the latency is paid once per 16 entries instead of once per entry.
GS_FN float rd_pair_sum(const float *tri, int M, int k, const uint8_t *crow, const uint8_t *scol /* stride 64 */)
{
    const int64_t block = (int64_t)k * (k + 1) / 2;
    float res = 0.0f;
    int m = 0;
    for (; m + 16 <= M; m += 16) {
        float e[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int c1 = crow[m + j], c2 = scol[(size_t)(m + j) * 64];
            const int r = c1 < c2 ? c1 : c2, c = c1 < c2 ? c2 : c1;
            e[j] = tri[(int64_t)(m + j) * block + rd_tri_row(r, k) + (c - r)];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) res += e[j];
    }
    for (; m < M; ++m) {
        const int c1 = crow[m], c2 = scol[(size_t)m * 64];
        const int r = c1 < c2 ? c1 : c2, c = c1 < c2 ? c2 : c1;
        res += tri[(int64_t)m * block + rd_tri_row(r, k) + (c - r)];
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
GS_FN float rd_pair_sum_tf(const float *cb, int M, int k, const float *cvec, const uint8_t *scol /* stride 64 */)
{
    float res = 0.0f;
    int m = 0;
    for (; m + 8 <= M; m += 8) {   // 16 gathers in flight, then the 8 entries in ascending m
        rd_f4 a0[8], a1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const rd_f4 *r = reinterpret_cast<const rd_f4 *>(cb + ((int64_t)(m + j) * k + scol[(size_t)(m + j) * 64]) * 8);
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
        const rd_f4 *r = reinterpret_cast<const rd_f4 *>(cb + ((int64_t)m * k + scol[(size_t)m * 64]) * 8);
        const rd_f4 *c = reinterpret_cast<const rd_f4 *>(cvec + (size_t)m * 8);
        res += rd_entry_tf<L2>(r[0], r[1], c[0], c[1]);
    }
    return res;
}

GS_FN float rd_self_sum(const float *tri, int M, int k, const uint8_t *crow)
{
    const int64_t block = (int64_t)k * (k + 1) / 2;
    float res = 0.0f;
    for (int m = 0; m < M; ++m) {
        const int c = crow[m];
        res += tri[(int64_t)m * block + rd_tri_row(c, k)];
    }
    return res;
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
template <bool TF = false>
GS_FN void rd_node(const RdParams &p, int node_idx, char *lds)
{
    const int lane = gs_lane();
    const int M = p.M, C = p.C;
    uint8_t *cc = reinterpret_cast<uint8_t *>(lds);               // [C][M] candidate codes
    uint8_t *st = cc + (size_t)C * M;                              // [M][64] selected codes, transposed
    float *cnorm = reinterpret_cast<float *>(lds + (((size_t)C * M + (size_t)M * 64 + 15) & ~(size_t)15));  // [C]
    float *snorm = cnorm + C;                                      // [64]
    int32_t *sidx = reinterpret_cast<int32_t *>(snorm + 64);       // [64] candidate index of slot j
    int32_t *snode = sidx + 64;                                    // [64]
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

    // ---- stage the candidates' code rows (and, for cosine, their self magnitudes) ----
    for (int i = 0; i < n; ++i) {
        const int32_t nd = nodes[i];
        const bool ok = nd >= 0 && nd < p.n;
        for (int b = lane; b < M; b += 64) cc[(size_t)i * M + b] = ok ? p.codes[(int64_t)nd * M + b] : (uint8_t)0;
    }
    gs_barrier();
    if (p.vsf == 2)
        for (int i = lane; i < n; i += 64) cnorm[i] = rd_self_sum(p.tri, M, p.k, cc + (size_t)i * M);
    gs_barrier();

    // selected BitSet as two 64-bit words per 128 candidates would not cover C up to 1024: keep it as one bit per lane-chunk:
    // lane l holds the bits of candidates l, l + 64, l + 128, ... in `mine`
    unsigned long long mine = 0;  // bit t: candidate t * 64 + lane is selected
    int nSlots = 0;               // selected neighbours held in the slot arrays (== number of selected bits)
    auto take = [&](int i) {      // wave-uniform i
        if (lane == (i & 63)) mine |= 1ull << (i >> 6);
        if (lane == 0) {
            sidx[nSlots] = i;
            snode[nSlots] = nodes[i];
            if (p.vsf == 2) snorm[nSlots] = cnorm[i];
        }
        for (int b = lane; b < M; b += 64) st[(size_t)b * 64 + nSlots] = cc[(size_t)i * M + b];
        nSlots++;
        gs_barrier();
    };
    {
        const int pre = diverseBefore < maxDegree ? diverseBefore : maxDegree;
        for (int i = 0; i < pre && i < n; ++i) take(i);
    }
    int nSelected = diverseBefore;
    float shortEdges = __builtin_nanf("");
    float currentAlpha = 1.0f;
    while ((double)currentAlpha <= (double)p.alpha + 1E-6 && nSelected < maxDegree) {
        for (int i = diverseBefore; i < n && nSelected < maxDegree; ++i) {
            const unsigned long long owner_bits = (unsigned long long)gs_shfl((long long)mine, i & 63);
            if ((owner_bits >> (i >> 6)) & 1ull) continue;
            const int32_t cNode = nodes[i];
            const float cScore = scores[i];
            // ---- isDiverse: every selected slot in parallel, then the first event in ascending candidate index ----
            int ev_idx = 0x7fffffff;  // this lane's event index (none: INT_MAX)
            bool ev_fail = false;
            if constexpr (TF) {
                if (nSlots > 0) {   // (the previous test's readers are past their last wave-wide step: rd_wave_min / the ballot)
                    for (int idx = lane; idx < 2 * M; idx += 64) {
                        const int m = idx >> 1;
                        const rd_f4 *r = reinterpret_cast<const rd_f4 *>(p.codebooks + ((int64_t)m * p.k + cc[(size_t)i * M + m]) * 8);
                        reinterpret_cast<rd_f4 *>(cvec)[idx] = r[idx & 1];
                    }
                    gs_barrier();
                }
            }
            if (lane < nSlots) {
                if (snode[lane] == cNode) {
                    ev_idx = sidx[lane];
                } else {
                    float sum;
                    if constexpr (TF) sum = p.vsf == 0 ? rd_pair_sum_tf<true>(p.codebooks, M, p.k, cvec, st + lane)
                                                       : rd_pair_sum_tf<false>(p.codebooks, M, p.k, cvec, st + lane);
                    else sum = rd_pair_sum(p.tri, M, p.k, cc + (size_t)i * M, st + lane);
                    float sim;
                    if (p.vsf == 0) sim = 1.0f / (1.0f + sum);
                    else if (p.vsf == 1) sim = (1.0f + sum) / 2.0f;
                    else {
                        const float prod = cnorm[i] * snorm[lane];
                        const float cosine = sum / (float)gs_sqrt((double)prod);
                        sim = (1.0f + cosine) / 2.0f;
                    }
                    if (sim > cScore * currentAlpha) {
                        ev_idx = sidx[lane];
                        ev_fail = true;
                    }
                }
            }
            const int first = rd_wave_min(ev_idx);
            const bool not_diverse = gs_ballot(ev_fail && ev_idx == first && first != 0x7fffffff) != 0;
            if (!not_diverse) {
                if (nSlots < 64) take(i);
                nSelected++;
            }
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
}

}  // namespace jv
