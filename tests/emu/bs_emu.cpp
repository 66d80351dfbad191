// bs_emu.cpp — the build-time scoring kernel bodies (jvector_amd/csrc/bs_body.h) as plain host loops: one loop
// iteration per GPU thread.  TEST HARNESS (g++ -O2 -ffp-contract=off), never linked into the product.
#include <cmath>
#include <cstdint>

#define BS_FN static inline
static inline double bs_sqrt(double x) { return std::sqrt(x); }
#include "../../jvector_amd/csrc/bs_body.h"

extern "C" {

void bs_emu_pair_table(const float *codebooks, const int64_t *cb_offsets, const int32_t *sizes, const int32_t *offsets, int D,
                       int M, int k, int vsf, float *out)
{
    jv::BsPq pq{codebooks, cb_offsets, sizes, offsets, nullptr, D, M, k};
    for (int64_t t = 0; t < (int64_t)M * k; ++t) jv::bs_pair_table_row(pq, vsf, t, out);
}

void bs_emu_pair_scores(const float *tri, int vsf, int M, int k, const uint8_t *codes, int64_t n, const int32_t *node1, int P,
                        const int32_t *node2, int B, float *out)
{
    for (int64_t t = 0; t < (int64_t)P * B; ++t) jv::bs_pair_score(tri, vsf, M, k, codes, n, node1, node2, B, t, out);
}

void bs_emu_fused_gather(const uint8_t *codes, int64_t n_codes, const int32_t *neighbors, int maxDegree, int M, int chunk, int64_t count,
                         uint8_t *blocks)
{
    for (int64_t t = 0; t < count * maxDegree * (M / chunk); ++t) jv::bs_fused_gather(codes, n_codes, neighbors, maxDegree, M, chunk, t, blocks);
}

void bs_emu_decode(const float *codebooks, const int64_t *cb_offsets, const int32_t *sizes, const int32_t *offsets,
                   const float *centroid, int D, int M, int k, const uint8_t *codes, int64_t n, const int32_t *ordinals,
                   int64_t first, int64_t count, float *out)
{
    jv::BsPq pq{codebooks, cb_offsets, sizes, offsets, centroid, D, M, k};
    for (int64_t t = 0; t < count * D; ++t) jv::bs_decode(pq, codes, n, ordinals, first, t, out);
}

void bs_emu_direct_scores(const float *codebooks, const int64_t *cb_offsets, const int32_t *sizes, const int32_t *offsets, int D,
                          int M, int k, int vsf, const uint8_t *codes, int64_t n, const float *cq, int Q, const int32_t *ordinals,
                          int B, float *qnorm_scratch, float *out)
{
    jv::BsPq pq{codebooks, cb_offsets, sizes, offsets, nullptr, D, M, k};
    for (int64_t q = 0; q < Q; ++q) jv::bs_query_norm(cq, D, q, qnorm_scratch);
    for (int64_t t = 0; t < (int64_t)Q * B; ++t) jv::bs_direct_score(pq, vsf, codes, n, cq, qnorm_scratch, ordinals, B, t, out);
}
}
