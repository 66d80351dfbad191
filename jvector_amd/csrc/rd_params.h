// rd_params.h — launch parameters of the batched robust prune (rd_body.h / k_retain_diverse.hip), shared with the host
// driver in build_score.cpp.  Plain data only.
#pragma once

#include <cstddef>
#include <cstdint>

namespace jv {

struct RdParams {
    const float *tri;          // pair table: M x k(k+1)/2 floats
    const float *sq;           // the same entries as a SQUARE table [M][k][k] (rd_node<.., SQ = true>: a test's lanes read ONE row per
                               // subspace — the candidate's code picks it — instead of up to one cache line each), else nullptr
    const float *codebooks;    // table-free form (rd_node<true>): [M][k][8] centroids (uniform 8-dimensional sub-vectors), else nullptr
    const uint8_t *codes;      // [n][M]
    int64_t n;
    const int32_t *cand_nodes; // [P][C] sorted by score descending; entries >= count are ignored
    const float *cand_scores;  // [P][C]
    const int32_t *cand_count; // [P] or nullptr (= C)
    const int32_t *diverse_before;  // [P] or nullptr (= 0): the first diverse_before candidates are taken as already diverse
    int32_t P, C, M, k, vsf, maxDegree;
    float alpha;
    int32_t wide_stage;        // != 0: candidate code rows are staged 16 bytes per lane with every load independent (M % 16 == 0)
    int32_t split;             // != 0: a test spreads each selected slot's M entries over up to six lanes (rd_pair_sum_split)
    int32_t chunk;             // > 0: incremental tests (rd_body.h): a candidate remembers how many leading selected slots it has been tested
                               // against and the largest similarity among them; a test examines only the slots behind that, `chunk` at a
                               // time, and stops at the first violation.  0: every test examines every selected slot.
    int32_t *selected_out;     // [P][maxDegree] selected candidate INDICES in ascending order, -1 padded
    int32_t *n_selected_out;   // [P]
    unsigned long long *prof;  // rd_node<.., PROF = true>: 16 counters (rd_body.h RD_PHASE), else nullptr
    float *short_edges_out;    // [P] or nullptr: nSelected after the alpha = 1.0 pass / maxDegree (NaN if the loop never ran)
    unsigned long long *counts; // [2] or nullptr: += {isDiverse tests, selected slots examined by them = (candidate, selected) pairs summed}
};

// LDS bytes one wavefront needs: candidate code rows, transposed selected codes, self magnitudes, slot bookkeeping
// (table_free: + the current candidate's decoded sub-vectors, M x 8 floats, at rd_off_cvec)
// layout: [C][Mp] candidate codes | [Mp/4][64] words of selected codes | (16-byte aligned) cnorm [C] | snorm [64] | sidx [64] | snode [64] |
//         tested [C] (int) | best [C] (float) | cid [C] (int) | csc [C] (float) | (16-byte aligned, table-free only) cvec [M][8]
// (code rows are padded to whole 4-byte words: the kernel reads four codes per LDS word)
constexpr int rd_row_bytes(int M) { return (M + 3) & ~3; }
constexpr size_t rd_off_tested(int C, int M)
{
    return (((size_t)C * rd_row_bytes(M) + (size_t)rd_row_bytes(M) * 64 + 15) & ~(size_t)15) + sizeof(float) * ((size_t)C + 64) + sizeof(int32_t) * 64 * 2;
}
constexpr size_t rd_off_cvec(int C, int M) { return (rd_off_tested(C, M) + 16 * (size_t)C + 15) & ~(size_t)15; }
inline size_t rd_lds_bytes(int C, int M, bool table_free = false)
{
    return rd_off_cvec(C, M) + (table_free ? sizeof(float) * 8 * (size_t)M : 0);
}

}  // namespace jv
