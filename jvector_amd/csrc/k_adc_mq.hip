// k_adc_mq.hip — multi-query ADC scan: the LDS-efficient form of k_adc.hip's scan for the flat-scan case
// where every query of the batch scores the SAME contiguous candidate range.
//
// Why: the single-query kernel is bound by LDS bank conflicts, not HBM (rocprof r1: 56.6 G lookups/s = the
// ds_read_b32 rate under ~3.5-way random conflicts; HBM traffic 8x below algorithmic).  Here P = 4 queries'
// tables are interleaved in LDS — entry (m, code) is one float4 {q0,q1,q2,q3} — so ONE ds_read_b128 per
// (candidate, m) serves four (query, candidate) lookups, and the candidate's code bytes are loaded once for
// the four queries.  4 x M x 1 KB does not fit 160 KB, so the table is cycled through LDS in slices of
// SL = 16*SLCH subspaces while each lane keeps the running sums of its R candidates x P queries in registers:
// per (query, candidate) the additions still happen in ascending m into one f32 — the exact association of
// DefaultVectorUtilSupport.assembleAndSum (:302-309) — so results stay bit-identical to the scalar reference.
//
// Epilogues:
//   * store   : out[q][i] = score                                  (jv_hip_adc_scan, sampling pass)
//   * filter  : append (id, score) to the query's candidate list when score >= tau[q]
//               (threshold-filtered scan of jv_hip_search_flat: no Q x N score round trip through HBM)
#include "jv_device.h"
#include "jv_internal.h"

namespace jv {

struct AdcMqParams {
    const float *luts;     // [Q][M_total*256]
    const float *bmag;     // [Q] (cosine)
    const uint8_t *codes;  // rows of M_total bytes, 16-byte aligned
    const float *norms;    // per-row decoded magnitude (cosine)
    float *out;            // store mode: [Q][count]
    const float *tau;      // filter mode: tau[q * tau_stride]
    int tau_stride;
    int32_t *cand_ids;     // filter mode: [Q][cap]
    float *cand_scores;    // filter mode: [Q][cap]
    unsigned int *cand_count;  // filter mode: [Q]
    int cap;
    int staged;            // filter mode: collect survivors per workgroup in LDS first (pays when a workgroup keeps dozens per query)
    int64_t first, count, row_stride;
    int Q, M_total;
};

template <int VSF, int SLCH, int R, bool FILTER>
__global__ __launch_bounds__(1024) void adc_mq_kernel(AdcMqParams p)
{
    constexpr int P = 4;
    constexpr int SL = 16 * SLCH;  // subspaces per LDS slice
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];  // [SL*256]

    const int q0 = blockIdx.x * P;
    const int64_t tile_base = (int64_t)blockIdx.y * (1024 * R);
    const int tid = threadIdx.x;
    const int nslices = p.M_total / SL;

    float acc[R][P];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < P; ++j) acc[r][j] = 0.0f;

    // per-query table bases (queries past Q read query Q-1's table; their results are discarded)
    const float *lq[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        int q = q0 + j < p.Q ? q0 + j : p.Q - 1;
        lq[j] = p.luts + (int64_t)q * p.M_total * kClusters;
    }

    for (int s = 0; s < nslices; ++s) {
        __syncthreads();  // readers of the previous slice are done
        const int mb = s * SL;
        for (int idx = tid; idx < SL * kClusters; idx += 1024) {
            const int off = mb * kClusters + idx;
            lds4[idx] = make_float4(lq[0][off], lq[1][off], lq[2][off], lq[3][off]);
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t i = tile_base + (int64_t)r * 1024 + tid;
            if (i >= p.count) continue;
            const int64_t row = p.first + i * p.row_stride;
            const uint4 *rp = reinterpret_cast<const uint4 *>(p.codes + row * p.M_total + mb);
            uint4 w[SLCH];
#pragma unroll
            for (int c = 0; c < SLCH; ++c) w[c] = rp[c];
#pragma unroll
            for (int c = 0; c < SLCH; ++c) {
                const uint32_t d[4] = {w[c].x, w[c].y, w[c].z, w[c].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int m = c * 16 + e * 4 + b;
                        const uint32_t code = (d[e] >> (8 * b)) & 0xFFu;
                        const float4 t = lds4[m * kClusters + code];
                        acc[r][0] += t.x;
                        acc[r][1] += t.y;
                        acc[r][2] += t.z;
                        acc[r][3] += t.w;
                    }
                }
            }
        }
    }

    // epilogue
    // FILTER: survivors are collected per workgroup — in the LDS the table slices no longer need — and handed to the query's
    // global list with ONE reservation per (workgroup, query) instead of one global atomic per survivor (at rerankK 3200 over 1M
    // codes 2.5 % of all pairs survive: the per-survivor atomics cost 1.6 of the scan's 4.7 ms).  A workgroup's share that does
    // not fit the staging area falls back to the per-survivor form; the list's order is immaterial (it feeds a top-k).  With few
    // survivors per workgroup (C3's rerankK 50 over 10M codes: ~3 per query) the extra barriers cost more than the atomics:
    // the host turns staging on from the expected count (AdcMqParams::staged).
    constexpr int STAGE = (SL * kClusters * (int)sizeof(float4)) / (P * 8);  // (id, score) pairs per query
    __shared__ unsigned int s_cnt[P], s_base[P];
    int32_t *st_ids = reinterpret_cast<int32_t *>(lds4);
    float *st_sc = reinterpret_cast<float *>(lds4) + (size_t)P * STAGE;
    const bool staged = FILTER && p.staged != 0;
    if (staged) {
        __syncthreads();  // every lane is done with the last table slice
        if (tid < P) s_cnt[tid] = 0u;
        __syncthreads();
    }
    // the sure-reject bounds of the four queries (see the loop below), once per lane
    float cut[P], bm[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        cut[j] = (VSF == VSF_L2) ? __builtin_inff() : -__builtin_inff();   // never rejects
        bm[j] = 0.0f;
        if (FILTER && q0 + j < p.Q) {
            const float tq = p.tau[(int64_t)(q0 + j) * p.tau_stride];
            if (VSF == VSF_L2) {
                if (tq > 1e-6f) cut[j] = (1.0f / tq - 1.0f) * 1.000002f + 4e-6f;
            } else if (VSF == VSF_DOT) {
                cut[j] = (2.0f * tq - 1.0f) - 4e-6f * (fabsf(2.0f * tq - 1.0f) + 1.0f);
            } else {
                cut[j] = 2.0f * tq - 1.0f;
                bm[j] = p.bmag[q0 + j];
            }
            if (!(tq == tq)) cut[j] = (VSF == VSF_L2) ? __builtin_inff() : -__builtin_inff();
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t i = tile_base + (int64_t)r * 1024 + tid;
        if (i >= p.count) continue;
        const int64_t row = p.first + i * p.row_stride;
        const float nrm = (VSF == VSF_COS) ? p.norms[row] : 0.0f;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int q = q0 + j;
            if (q >= p.Q) continue;
            if (FILTER) {
                // Cheap sure-reject before the similarity transform (an IEEE divide for EUCLIDEAN, a double sqrt + divide for COSINE:
                // a third of this kernel's arithmetic at M = 16, and all but a few pairs in a thousand fail the threshold anyway).
                // The test is conservative by a relative 2e-6 / absolute 4e-6 — orders of magnitude above the transform's rounding —
                // so it only drops pairs the exact comparison below would drop too; the survivor set is unchanged.
                const float raw = acc[r][j];
                bool sure_fail;
                if (VSF == VSF_L2) sure_fail = raw > cut[j];                                          // 1 / (1 + raw) < tau
                else if (VSF == VSF_DOT) sure_fail = raw < cut[j];                                     // (1 + raw) / 2 < tau
                else {
                    const float prod = nrm * bm[j];
                    const float c = raw * __builtin_amdgcn_rsqf(prod);                                  // ~cosine, few ulps
                    sure_fail = prod > 1e-30f && prod < 1e30f && c < cut[j] - 8e-6f * (fabsf(c) + 1.0f);
                }
                if (sure_fail) continue;
            }
            float sc;
            if (VSF == VSF_COS) sc = score_from_raw(VSF_COS, cosine_finish(acc[r][j], nrm, p.bmag[q]));
            else sc = score_from_raw(VSF, acc[r][j]);
            if (FILTER) {
                if (sc >= p.tau[(int64_t)q * p.tau_stride]) {
                    const unsigned int sp = staged ? atomicAdd(&s_cnt[j], 1u) : (unsigned int)STAGE;
                    if (sp < (unsigned int)STAGE) {
                        st_ids[j * STAGE + sp] = (int32_t)row;
                        st_sc[j * STAGE + sp] = sc;
                    } else {  // staging area full: straight to the global list
                        const unsigned int pos = atomicAdd(&p.cand_count[q], 1u);
                        if (pos < (unsigned int)p.cap) {
                            p.cand_ids[(int64_t)q * p.cap + pos] = (int32_t)row;
                            p.cand_scores[(int64_t)q * p.cap + pos] = sc;
                        }
                    }
                }
            } else {
                p.out[(int64_t)q * p.count + i] = sc;
            }
        }
    }
    if (staged) {
        __syncthreads();
        if (tid < P && q0 + tid < p.Q) {
            const unsigned int n = s_cnt[tid] < (unsigned int)STAGE ? s_cnt[tid] : (unsigned int)STAGE;
            s_base[tid] = n ? atomicAdd(&p.cand_count[q0 + tid], n) : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int q = q0 + j;
            if (q >= p.Q) continue;
            const unsigned int n = s_cnt[j] < (unsigned int)STAGE ? s_cnt[j] : (unsigned int)STAGE, base = s_base[j];
            for (unsigned int t = tid; t < n; t += 1024) {
                const unsigned int pos = base + t;
                if (pos < (unsigned int)p.cap) {
                    p.cand_ids[(int64_t)q * p.cap + pos] = st_ids[j * STAGE + t];
                    p.cand_scores[(int64_t)q * p.cap + pos] = st_sc[j * STAGE + t];
                }
            }
        }
    }
}

template <int VSF, int SLCH, bool FILTER>
static int launch_mq_r(hipStream_t s, const AdcMqParams &p, int R)
{
    const size_t lds = (size_t)16 * SLCH * kClusters * sizeof(float4);
#define JV_MQ(RR)                                                                                              \
    do {                                                                                                       \
        auto kfn = adc_mq_kernel<VSF, SLCH, RR, FILTER>;                                                       \
        JV_HIP_CHECK(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        dim3 grid((p.Q + 3) / 4, (unsigned)((p.count + 1024 * RR - 1) / (1024 * RR)));                         \
        hipLaunchKernelGGL(kfn, grid, dim3(1024), lds, s, p);                                                  \
    } while (0)
    if (R >= 8) JV_MQ(8);
    else JV_MQ(4);
#undef JV_MQ
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

template <int VSF, bool FILTER>
static int launch_mq_slch(hipStream_t s, const AdcMqParams &p, int slch, int R)
{
    if (slch == 2) return launch_mq_r<VSF, 2, FILTER>(s, p, R);
    return launch_mq_r<VSF, 1, FILTER>(s, p, R);
}

// true when the multi-query kernel can run this shape (else callers use k_adc.hip)
bool adc_mq_supported(int M, const uint8_t *d_codes)
{
    return M % 16 == 0 && (reinterpret_cast<uintptr_t>(d_codes) & 15) == 0;
}

static int mq_slch(int M) { return (M % 32 == 0) ? 2 : 1; }

static int mq_R(const jv_ctx *ctx, int Q, int64_t count)
{
    // candidates per lane: amortise the per-slice table refill, but keep >= ~2 workgroups per CU in the grid
    const int64_t groups = (Q + 3) / 4;
    // (R = 16 would need 64 accumulators + staging > the 128 VGPRs a 1024-thread workgroup allows: it spills)
    if (groups * ((count + 1024 * 8 - 1) / (1024 * 8)) >= 2 * (int64_t)ctx->num_cus) return 8;
    return 4;
}

int launch_adc_mq_store(hipStream_t s, const jv_ctx *ctx, const float *d_luts, const float *d_bmag, int Q, int M,
                        int vsf, const uint8_t *d_codes, const float *d_norms, int64_t first, int64_t count,
                        int64_t row_stride, float *d_out)
{
    if (Q == 0 || count == 0) return JV_OK;
    AdcMqParams p{};
    p.luts = d_luts; p.bmag = d_bmag; p.codes = d_codes; p.norms = d_norms; p.out = d_out;
    p.first = first; p.count = count; p.row_stride = row_stride; p.Q = Q; p.M_total = M;
    const int slch = mq_slch(M), R = mq_R(ctx, Q, count);
    switch (vsf) {
    case VSF_L2: return launch_mq_slch<VSF_L2, false>(s, p, slch, R);
    case VSF_DOT: return launch_mq_slch<VSF_DOT, false>(s, p, slch, R);
    case VSF_COS: return launch_mq_slch<VSF_COS, false>(s, p, slch, R);
    default: return launch_mq_slch<VSF_RAW, false>(s, p, slch, R);
    }
}

int launch_adc_mq_filter(hipStream_t s, const jv_ctx *ctx, const float *d_luts, const float *d_bmag, int Q, int M,
                         int vsf, const uint8_t *d_codes, const float *d_norms, int64_t first, int64_t count,
                         const float *d_tau, int tau_stride, int32_t *d_cand_ids, float *d_cand_scores,
                         unsigned int *d_cand_count, int cap)
{
    if (Q == 0 || count == 0) return JV_OK;
    AdcMqParams p{};
    p.luts = d_luts; p.bmag = d_bmag; p.codes = d_codes; p.norms = d_norms;
    p.tau = d_tau; p.tau_stride = tau_stride; p.cand_ids = d_cand_ids; p.cand_scores = d_cand_scores; p.cand_count = d_cand_count; p.cap = cap;
    {   // expected survivors per (workgroup, query): the filter aims at cap / 4 per query over `count` candidates
        const int R0 = mq_R(ctx, Q, count);
        const double per_wg = (double)cap / 4.0 * (1024.0 * R0) / (double)count;
        p.staged = (per_wg >= 16.0 && !getenv("JVECTOR_HIP_ADC_NO_STAGING")) ? 1 : 0;
    }
    p.first = first; p.count = count; p.row_stride = 1; p.Q = Q; p.M_total = M;
    const int slch = mq_slch(M), R = mq_R(ctx, Q, count);
    switch (vsf) {
    case VSF_L2: return launch_mq_slch<VSF_L2, true>(s, p, slch, R);
    case VSF_DOT: return launch_mq_slch<VSF_DOT, true>(s, p, slch, R);
    default: return launch_mq_slch<VSF_COS, true>(s, p, slch, R);
    }
}

}  // namespace jv
