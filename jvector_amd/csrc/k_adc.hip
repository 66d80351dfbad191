// k_adc.hip — ADC scoring kernels: assembleAndSum / pqDecodedCosineSimilarity
// over candidate sets, with the look-up table of the block's query staged in LDS.
//
// Design:
//   * one workgroup = one (query, candidate segment); the query's LUT slice (m_count x 256 f32, 96 KB at
//     M=96) is copied L2 -> LDS once per workgroup and amortised over >= 32k candidates;
//   * one lane = one candidate: the lane streams its M code bytes with 16-byte loads (a wave covers 64
//     consecutive rows = one contiguous 64*M-byte span, every fetched line fully used) and performs the
//     M dependent-free LDS gathers, accumulating IN ORDER m = 0..M-1 into one f32 — the exact association
//     of DefaultVectorUtilSupport.assembleAndSum (:302-309), so sums are bit-identical to the scalar
//     reference;
//   * cosine: the candidate-side magnitude sum_m aMag[m*256+code[m]] is query independent, so it is
//     precomputed once per code row (same kernel, table = self-magnitudes, VSF_RAW) and streamed as 4
//     extra bytes per candidate — this halves the LDS gathers and lets PQ-96 fit LDS (two 96 KB tables do
//     not fit 160 KB).  The per-candidate f32 sum is formed in the same m-ascending order, hence identical
//     bits to pqDecodedCosineSimilarity's aMag accumulator (VectorUtilSupport.java:152-165);
//   * M*1 KB > LDS budget (e.g. PQ-192): the row is processed in several passes over m-ranges, carrying
//     the partial f32 sum through the output buffer — still one sequential chain, still bit-exact;
//   * small candidate sets per query (graph frontiers, fused neighbour blocks): the LUT stays in L2 and is
//     gathered through the vector-memory path (lut_in_lds = false) because staging 96 KB for 32 rows
//     would cost 30x the useful traffic.
#include "jv_device.h"
#include "jv_internal.h"

namespace jv {

struct AdcParams {
    const float *luts;      // [Q][M_total*256]
    const float *bmag;      // [Q] (cosine)
    const uint8_t *codes;   // rows of M_total bytes
    const float *norms;     // per-row decoded magnitude (cosine)
    const int32_t *ordinals;  // gather: [Q][count]; nullptr: contiguous scan
    const int32_t *origins;   // fused: [Q] origin node; row = origin*maxDegree + j
    const int32_t *neighbors; // fused: [n_nodes][maxDegree] neighbour ids (validity)
    int32_t *neighbors_out;   // fused optional: [Q][maxDegree]
    float *out;             // [Q][count]
    int64_t n_rows;         // number of valid rows in codes (bounds for gather)
    int64_t first;          // scan: first row
    int64_t count;          // candidates per query
    int64_t ld;             // gather: rows of `ordinals` and `out` are ld entries apart (0 = count)
    int64_t seg_len;        // candidates per workgroup
    int M_total;            // bytes per row
    int m_begin, m_count;   // subspace range handled by this pass
    int init_from_out;      // continue a partial sum stored in out
    int finalize;           // apply the score transform
    int maxDegree;          // fused
};

// row: pointer to the candidate's code bytes at m_begin.  Accumulates table[(m)*256 + code[m]] for the
// pass's subspaces, in ascending m.
template <int CH16>
__device__ __forceinline__ float adc_row_sum16(const float *lut, const uint8_t *row, float sum)
{
    const uint4 *r4 = reinterpret_cast<const uint4 *>(row);
    uint4 w[CH16];
#pragma unroll
    for (int c = 0; c < CH16; ++c) w[c] = r4[c];
#pragma unroll
    for (int c = 0; c < CH16; ++c) {
        const uint32_t d[4] = {w[c].x, w[c].y, w[c].z, w[c].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int m = c * 16 + e * 4 + b;
                const uint32_t code = (d[e] >> (8 * b)) & 0xFFu;
                sum += lut[m * kClusters + code];
            }
        }
    }
    return sum;
}

__device__ __forceinline__ float adc_row_sum_generic(const float *lut, const uint8_t *row, int m_count, float sum)
{
    int m = 0;
    if (((reinterpret_cast<uintptr_t>(row)) & 3) == 0) {
        for (; m + 4 <= m_count; m += 4) {
            const uint32_t d = *reinterpret_cast<const uint32_t *>(row + m);
#pragma unroll
            for (int b = 0; b < 4; ++b) sum += lut[(m + b) * kClusters + ((d >> (8 * b)) & 0xFFu)];
        }
    }
    for (; m < m_count; ++m) sum += lut[m * kClusters + row[m]];
    return sum;
}

// grid (Q, nseg), block THREADS.  CH16 > 0: m_count == 16*CH16 and rows 16-byte aligned.
template <int VSF, int CH16, bool LUT_IN_LDS, int THREADS>
__global__ __launch_bounds__(THREADS) void adc_kernel(AdcParams p)
{
    extern __shared__ __attribute__((aligned(16))) float lds_lut[];
    const int q = blockIdx.x;
    const float *glut = p.luts + ((int64_t)q * p.M_total + p.m_begin) * kClusters;
    const float *lut;
    if (LUT_IN_LDS) {
        const float4 *src = reinterpret_cast<const float4 *>(glut);
        float4 *dst = reinterpret_cast<float4 *>(lds_lut);
        const int n4 = p.m_count * (kClusters / 4);
        for (int i = threadIdx.x; i < n4; i += THREADS) dst[i] = src[i];
        __syncthreads();
        lut = lds_lut;
    } else {
        lut = glut;
    }

    const int64_t seg_begin = (int64_t)blockIdx.y * p.seg_len;
    int64_t seg_end = seg_begin + p.seg_len;
    if (seg_end > p.count) seg_end = p.count;
    const float bmag = (VSF == VSF_COS) ? p.bmag[q] : 0.0f;

    for (int64_t i = seg_begin + threadIdx.x; i < seg_end; i += THREADS) {
        int64_t row;
        bool valid = true;
        if (p.origins) {
            const int64_t origin = p.origins[q];
            valid = origin >= 0 && origin < p.n_rows / p.maxDegree;
            row = origin * p.maxDegree + i;
            if (valid) {
                const int32_t nb = p.neighbors[row];
                if (p.neighbors_out && p.m_begin == 0) p.neighbors_out[(int64_t)q * p.count + i] = nb;
                valid = nb >= 0;
            } else if (p.neighbors_out && p.m_begin == 0) {
                p.neighbors_out[(int64_t)q * p.count + i] = -1;
            }
        } else if (p.ordinals) {
            row = p.ordinals[(int64_t)q * (p.ld ? p.ld : p.count) + i];
            valid = row >= 0 && row < p.n_rows;
        } else {
            row = p.first + i;
        }
        float *o = p.out + (int64_t)q * ((p.ordinals && p.ld) ? p.ld : p.count) + i;
        if (!valid) {
            *o = -INFINITY;
            continue;
        }
        const uint8_t *rp = p.codes + row * p.M_total + p.m_begin;
        float sum = p.init_from_out ? *o : 0.0f;
        if (CH16 > 0) sum = adc_row_sum16<(CH16 > 0 ? CH16 : 1)>(lut, rp, sum);
        else sum = adc_row_sum_generic(lut, rp, p.m_count, sum);
        if (p.finalize) {
            if (VSF == VSF_COS) sum = score_from_raw(VSF_COS, cosine_finish(sum, p.norms[row], bmag));
            else sum = score_from_raw(VSF, sum);
        }
        *o = sum;
    }
}

template <int VSF, bool LUT_IN_LDS, int THREADS>
static int launch_adc_pass(hipStream_t s, const AdcParams &p, dim3 grid, size_t lds)
{
#define JV_ADC(CH)                                                                                           \
    do {                                                                                                     \
        auto kfn = adc_kernel<VSF, CH, LUT_IN_LDS, THREADS>;                                                 \
        if (lds > 64 * 1024)                                                                                 \
            JV_HIP_CHECK(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                             (int)lds));                                                     \
        hipLaunchKernelGGL(kfn, grid, dim3(THREADS), lds, s, p);                                             \
    } while (0)
    const bool aligned16 = (p.M_total % 16 == 0) && (p.m_begin % 16 == 0) && (p.m_count % 16 == 0) &&
                           ((reinterpret_cast<uintptr_t>(p.codes) & 15) == 0);
    int ch = aligned16 ? p.m_count / 16 : 0;
    switch (ch) {
    case 1: JV_ADC(1); break;
    case 2: JV_ADC(2); break;
    case 3: JV_ADC(3); break;
    case 4: JV_ADC(4); break;
    case 6: JV_ADC(6); break;
    case 8: JV_ADC(8); break;
    default: JV_ADC(0); break;
    }
#undef JV_ADC
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

template <bool LUT_IN_LDS, int THREADS>
static int launch_adc_vsf(hipStream_t s, int vsf, const AdcParams &p, dim3 grid, size_t lds)
{
    switch (vsf) {
    case VSF_L2: return launch_adc_pass<VSF_L2, LUT_IN_LDS, THREADS>(s, p, grid, lds);
    case VSF_DOT: return launch_adc_pass<VSF_DOT, LUT_IN_LDS, THREADS>(s, p, grid, lds);
    case VSF_COS: return launch_adc_pass<VSF_COS, LUT_IN_LDS, THREADS>(s, p, grid, lds);
    default: return launch_adc_pass<VSF_RAW, LUT_IN_LDS, THREADS>(s, p, grid, lds);
    }
}

// Largest number of subspaces whose table fits the LDS budget, rounded to a multiple of 16 when possible.
static int lds_subspaces(const jv_ctx *ctx, int M)
{
    int budget = (int)((ctx->lds_per_block - 1024) / (kClusters * sizeof(float)));  // keep 1 KB slack
    if (budget >= M) return M;
    if (budget >= 16) budget -= budget % 16;
    // balance the passes (192 -> 96 + 96 rather than 144 + 48)
    int passes = (M + budget - 1) / budget;
    int per = (M + passes - 1) / passes;
    if (per % 16) per += 16 - per % 16;
    return per <= budget ? per : budget;
}

static int run_adc(hipStream_t s, const jv_ctx *ctx, AdcParams p, int Q, int vsf)
{
    if (Q == 0 || p.count == 0) return JV_OK;
    const int M = p.M_total;
    // Small per-query candidate sets: LUT stays in L2 (gathered through the vector memory path).
    const bool small = p.count < 2048;
    if (small) {
        p.m_begin = 0;
        p.m_count = M;
        p.init_from_out = 0;
        p.finalize = 1;
        p.seg_len = 64;
        dim3 grid(Q, (unsigned)((p.count + 63) / 64));
        return launch_adc_vsf<false, 64>(s, vsf, p, grid, 0);
    }
    const int per_pass = lds_subspaces(ctx, M);
    // segment length: amortise the LUT fill (m_count KB) over >= 32k candidates, but keep enough
    // workgroups to fill the chip when Q is small.
    int64_t seg = 32768;
    while (seg > 4096 && (int64_t)Q * ((p.count + seg - 1) / seg) < 2 * (int64_t)ctx->num_cus) seg /= 2;
    p.seg_len = seg;
    dim3 grid(Q, (unsigned)((p.count + seg - 1) / seg));
    for (int mb = 0; mb < M; mb += per_pass) {
        p.m_begin = mb;
        p.m_count = (mb + per_pass <= M) ? per_pass : (M - mb);
        p.init_from_out = mb > 0;
        p.finalize = (mb + p.m_count >= M);
        size_t lds = (size_t)p.m_count * kClusters * sizeof(float);
        JV_TRY((launch_adc_vsf<true, 1024>(s, vsf, p, grid, lds)));
    }
    return JV_OK;
}

int launch_adc(hipStream_t s, const jv_ctx *ctx, const float *d_luts, const float *d_bmag, int Q, int M, int vsf,
               const uint8_t *d_codes, const float *d_norms, int64_t n_codes, int64_t first, int64_t count,
               const int32_t *d_ordinals, float *d_out)
{
    AdcParams p{};
    p.luts = d_luts;
    p.bmag = d_bmag;
    p.codes = d_codes;
    p.norms = d_norms;
    p.ordinals = d_ordinals;
    p.out = d_out;
    p.n_rows = n_codes;
    p.first = first;
    p.count = count;
    p.M_total = M;
    return run_adc(s, ctx, p, Q, vsf);
}

// gather over the first `count` entries of ordinal rows that are `ld` entries apart (scores land at the same pitch)
int launch_adc_pitched(hipStream_t s, const jv_ctx *ctx, const float *d_luts, const float *d_bmag, int Q, int M, int vsf, const uint8_t *d_codes,
                       const float *d_norms, int64_t n_codes, int64_t count, int64_t ld, const int32_t *d_ordinals, float *d_out)
{
    AdcParams p{};
    p.luts = d_luts;
    p.bmag = d_bmag;
    p.codes = d_codes;
    p.norms = d_norms;
    p.ordinals = d_ordinals;
    p.out = d_out;
    p.n_rows = n_codes;
    p.first = 0;
    p.count = count;
    p.ld = ld;
    p.M_total = M;
    return run_adc(s, ctx, p, Q, vsf);
}

int launch_code_norms(hipStream_t s, const jv_ctx *ctx, const float *d_table, int M, const uint8_t *d_codes,
                      int64_t count, float *d_out)
{
    AdcParams p{};
    p.luts = d_table;
    p.codes = d_codes;
    p.out = d_out;
    p.n_rows = count;
    p.first = 0;
    p.count = count;
    p.M_total = M;
    return run_adc(s, ctx, p, 1, VSF_RAW);
}

int launch_fused(hipStream_t s, const jv_ctx *ctx, const float *d_luts, const float *d_bmag, int Q, int M, int vsf,
                 const uint8_t *d_blocks, const int32_t *d_neighbors, const float *d_norms, int maxDegree,
                 int64_t n_nodes, const int32_t *d_origins, float *d_out, int32_t *d_neighbors_out)
{
    AdcParams p{};
    p.luts = d_luts;
    p.bmag = d_bmag;
    p.codes = d_blocks;
    p.norms = d_norms;
    p.origins = d_origins;
    p.neighbors = d_neighbors;
    p.neighbors_out = d_neighbors_out;
    p.out = d_out;
    p.n_rows = n_nodes * maxDegree;
    p.count = maxDegree;
    p.M_total = M;
    p.maxDegree = maxDegree;
    return run_adc(s, ctx, p, Q, vsf);
}

}  // namespace jv
