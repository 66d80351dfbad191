// k_adc_bq.hip — the flat scan's threshold filter in two stages (round 5):
//   1. a BOUND scan over all N codes with 7-bit bound tables, SIXTEEN queries per 16-byte LDS word (adc_bq_kernel): per (query,
//      candidate) an integer sum of M buckets — a quarter of the multi-query float kernel's LDS traffic per pair — and the candidates
//      whose bound cannot reach the query's threshold are dropped;
//   2. the EXACT ADC score (k_adc.hip's gather kernel: float tables, ascending m, bit-exact) of the survivors only.
// The candidate lists a search takes its top-rerankK from hold a superset of {score >= tau} with exact scores, so results are the
// multi-query kernel's (k_adc_mq.hip) bit for bit.  Why: that kernel is bound by LDS bank conflicts of random 16-byte gathers (0.16 /
// 0.27 of the ds_read_b128 peak at M = 16 / 96, conflict fraction 0.64) and sums all M entries of every pair although a threshold at
// the top 4e-6 of the candidates rejects nearly all of them; a wave-uniform early abandon does not work (scripts/flat_abandon_study.py:
// 85 % of the pairs but 0 % of the waves are rejectable after a third of the subspaces).
// Bound tables (adc_bq_table_kernel, from the queries' float tables): per subspace lo_m / hi_m, ONE scale S = max range / 127 per
// query, bucket b in [0, 127].  Dot product / cosine need an UPPER bound of the raw sum: lo_m + S (b + 1) >= entry; euclidean a LOWER
// bound: lo_m + S b <= entry; both verified in f32 per entry, plus a slack for the roundings of the exact chain.  Two look-ups are
// added as packed bytes before they are widened to 16-bit lanes (127 + 127 < 256): 0.75 VALU instructions per (query, candidate,
// subspace) instead of 1.25.
#include "jv_device.h"
#include "jv_internal.h"

namespace jv {

constexpr int BQ_P = 16;   // queries per LDS word

struct AdcBqParams {
    const uint8_t *tab;      // [G][M][256][16] buckets, G = ceil(Q / 16)
    const float *meta;       // [Q][4]: base (incl. slack), S, usable, unused
    const float *bmag;       // [Q] (cosine)
    const uint8_t *codes;    // rows of M bytes, 16-byte aligned
    const float *norms;      // per-row decoded magnitude (cosine)
    const float *tau;        // tau[q * tau_stride]
    int tau_stride;
    int32_t *surv_ids;       // [Q][cap2], preset to -1
    unsigned int *surv_cnt;  // [Q]
    int cap2;
    int64_t first, count;
    int Q, M_total;
    int xcd_order;           // 1: tiles dealt to XCDs, a tile's query groups back to back (adc_bq_kernel)
};

__device__ __forceinline__ float bq_wave_min(float v)
{
    for (int o = 32; o > 0; o >>= 1) {
        const float t = __shfl_xor(v, o, 64);
        v = t < v ? t : v;
    }
    return v;
}
__device__ __forceinline__ float bq_wave_max(float v)
{
    for (int o = 32; o > 0; o >>= 1) {
        const float t = __shfl_xor(v, o, 64);
        v = t > v ? t : v;
    }
    return v;
}

// one block per query, thread = code
template <int VSF>
__global__ __launch_bounds__(256) void adc_bq_table_kernel(const float *__restrict__ luts, int Q, int M, uint8_t *__restrict__ tab, float *__restrict__ meta)
{
    __shared__ float w_lo[4][192], w_hi[4][192];
    __shared__ float s_lo[192], s_hi[192];
    __shared__ float s_S, s_inv;
    __shared__ int s_bad;
    const int q = (int)blockIdx.x, c = (int)threadIdx.x, lane = c & 63, wave = c >> 6;
    const float *lut = luts + (int64_t)q * M * kClusters;
    if (c == 0) s_bad = 0;
    __syncthreads();
    bool bad = false;
    for (int m = 0; m < M; ++m) {
        const float e = lut[m * kClusters + c];
        bad = bad || !(e - e == 0.0f);
        const float mn = bq_wave_min(e), mx = bq_wave_max(e);
        if (lane == 0) {
            w_lo[wave][m] = mn;
            w_hi[wave][m] = mx;
        }
    }
    if (bad) s_bad = 1;   // (benign race)
    __syncthreads();
    for (int m = c; m < M; m += 256) {
        float mn = w_lo[0][m], mx = w_hi[0][m];
        for (int w = 1; w < 4; ++w) {
            mn = w_lo[w][m] < mn ? w_lo[w][m] : mn;
            mx = w_hi[w][m] > mx ? w_hi[w][m] : mx;
        }
        s_lo[m] = mn;
        s_hi[m] = mx;
    }
    __syncthreads();
    if (c == 0) {
        float range = 0.0f;
        for (int m = 0; m < M; ++m) {
            const float r = s_hi[m] - s_lo[m];
            if (r > range) range = r;
        }
        float S = range / 127.0f;
        if (!(S > 1e-30f)) S = 1e-30f;
        float sum_lo = 0.0f, sum_abs = 0.0f, max_abs = 0.0f;
        for (int m = 0; m < M; ++m) {
            const float l = s_lo[m], h = s_hi[m];
            sum_lo += l;
            const float a = fmaxf(fabsf(l), fabsf(h));
            sum_abs += a + 128.0f * S;
            if (a > max_abs) max_abs = a;
        }
        const bool ok = s_bad == 0 && (sum_abs - sum_abs == 0.0f) && S * 1e6f >= max_abs;
        const float slack = 4e-5f * sum_abs;
        s_S = S;
        s_inv = 1.0f / S;
        float *mq = meta + (int64_t)q * 4;
        mq[0] = (VSF == VSF_L2) ? sum_lo - slack : sum_lo + slack;
        mq[1] = S;
        mq[2] = ok ? 1.0f : 0.0f;
        mq[3] = 0.0f;
    }
    __syncthreads();
    const float S = s_S, inv = s_inv;
    uint8_t *out = tab + ((int64_t)(q / BQ_P) * M * kClusters + c) * BQ_P + (q % BQ_P);
    for (int m = 0; m < M; ++m) {
        const float e = lut[m * kClusters + c], l = s_lo[m];
        int b = (int)((e - l) * inv);
        b = b < 0 ? 0 : (b > 127 ? 127 : b);
        if (VSF == VSF_L2) {   // lower edge l + S b <= e
            if (b > 0 && l + S * (float)b > e) --b;
            if (b > 0 && l + S * (float)b > e) --b;
            if (l + S * (float)b > e) b = 0;
        } else {               // upper edge l + S (b + 1) >= e
            if (b < 127 && l + S * (float)(b + 1) < e) ++b;
            if (b < 127 && l + S * (float)(b + 1) < e) ++b;
            if (l + S * (float)(b + 1) < e) b = 127;
        }
        out[(int64_t)m * kClusters * BQ_P] = (uint8_t)b;
    }
}

template <int VSF, int SLCH, int R>
__global__ __launch_bounds__(1024) void adc_bq_kernel(AdcBqParams p)
{
    constexpr int SL = 16 * SLCH;   // subspaces per LDS slice
    extern __shared__ __attribute__((aligned(16))) uint4 lds16[];   // [SL * 256] words of sixteen buckets
    // XCD-aware block order (round 6).  Workgroups are dealt to the 8 XCDs round-robin by linear id; with the plain (group, tile) grid an
    // XCD served the query groups = its number (mod 8) and every tile of codes was pulled into all eight L2s — and, because the blocks
    // of one tile ran at different times, mostly from HBM again: FETCH_SIZE 30 GB per launch at the C4 shard, 24x one pass over the
    // codes (profiles/traffic_r6.json).  Remapped: XCD k takes the tiles = k (mod 8) and runs ALL query groups of a tile back to back,
    // so a tile (786 KB at PQ-96) is read from HBM once and served to the other groups by that XCD's L2.  (The last T % 8 tiles keep
    // the plain order.)
    int g = (int)blockIdx.x, tile = (int)blockIdx.y;
    {
        const int G = (int)gridDim.x, T = (int)gridDim.y, Tfull = T & ~7;
        const int64_t L = (int64_t)blockIdx.x + (int64_t)G * blockIdx.y;
        if (p.xcd_order && L < (int64_t)G * Tfull) {
            const int k = (int)(L & 7);
            const int64_t pos = L >> 3;
            g = (int)(pos % G);
            tile = (int)(pos / G) * 8 + k;
        }
    }
    const int q0 = g * BQ_P;
    const int64_t tile_base = (int64_t)tile * (1024 * R);
    const int tid = (int)threadIdx.x;
    const int nslices = p.M_total / SL;
    const uint4 *gtab = reinterpret_cast<const uint4 *>(p.tab) + (int64_t)g * p.M_total * kClusters;

    // acc[r][2k]: queries 4k (low half) and 4k + 2 (high half); acc[r][2k + 1]: queries 4k + 1 and 4k + 3 — 16-bit sums of buckets
    uint32_t acc[R][8];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[r][k] = 0u;

    for (int s = 0; s < nslices; ++s) {
        __syncthreads();
        const int mb = s * SL;
        for (int idx = tid; idx < SL * kClusters; idx += 1024) lds16[idx] = gtab[mb * kClusters + idx];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t i = tile_base + (int64_t)r * 1024 + tid;
            if (i >= p.count) continue;
            const int64_t row = p.first + i;
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
                    for (int b = 0; b < 4; b += 2) {
                        const int m = c * 16 + e * 4 + b;
                        const uint32_t c0 = (d[e] >> (8 * b)) & 0xFFu, c1 = (d[e] >> (8 * b + 8)) & 0xFFu;
                        const uint4 t0 = lds16[m * kClusters + c0], t1 = lds16[(m + 1) * kClusters + c1];
                        const uint32_t sx = t0.x + t1.x, sy = t0.y + t1.y, sz = t0.z + t1.z, sw = t0.w + t1.w;   // bytes <= 254: no carry
                        acc[r][0] += sx & 0x00FF00FFu;
                        acc[r][1] += (sx >> 8) & 0x00FF00FFu;
                        acc[r][2] += sy & 0x00FF00FFu;
                        acc[r][3] += (sy >> 8) & 0x00FF00FFu;
                        acc[r][4] += sz & 0x00FF00FFu;
                        acc[r][5] += (sz >> 8) & 0x00FF00FFu;
                        acc[r][6] += sw & 0x00FF00FFu;
                        acc[r][7] += (sw >> 8) & 0x00FF00FFu;
                    }
                }
            }
        }
    }

    // ---- epilogue: the bound of every (query, candidate) against the query's threshold; survivors staged per workgroup in the LDS the
    //      table slices no longer need, one reservation per (workgroup, query) in the query's global list ----
    constexpr int STAGE = SL * kClusters * 16 / (BQ_P * 4);   // ids per query
    __shared__ unsigned int s_cnt[BQ_P], s_base[BQ_P];
    __shared__ float s_qbase[BQ_P], s_qS[BQ_P], s_cut[BQ_P], s_bm[BQ_P];
    __shared__ int s_on[BQ_P];
    __shared__ int s_T[BQ_P];   // euclidean / dot product: the threshold as an INTEGER bound on the bucket sum (see below)
    int32_t *st_ids = reinterpret_cast<int32_t *>(lds16);
    __syncthreads();   // every lane is done with the last table slice
    if (tid < BQ_P) {
        const int q = q0 + tid;
        s_cnt[tid] = 0u;
        s_on[tid] = 0;
        if (q < p.Q) {
            const float *mq = p.meta + (int64_t)q * 4;
            const float tq = p.tau[(int64_t)q * p.tau_stride];
            s_qbase[tid] = mq[0];
            s_qS[tid] = mq[1];
            s_bm[tid] = (VSF == VSF_COS) ? p.bmag[q] : 0.0f;
            // the same conservative cuts as adc_mq_kernel's sure-reject test: they only drop what the exact comparison would drop too
            float cut;
            bool on = mq[2] != 0.0f && tq == tq;
            if (VSF == VSF_L2) {
                cut = (1.0f / tq - 1.0f) * 1.000002f + 4e-6f;
                on = on && tq > 1e-6f;
            } else if (VSF == VSF_DOT) {
                cut = (2.0f * tq - 1.0f) - 4e-6f * (fabsf(2.0f * tq - 1.0f) + 1.0f);
            } else {
                cut = 2.0f * tq - 1.0f;
            }
            s_cut[tid] = cut;
            s_on[tid] = on ? 1 : 2;   // 2: no bound for this query — every candidate survives (the caller falls back on overflow)
            // Without a per-candidate factor the test is linear in the bucket sum, so it is made once per query instead of once per pair:
            // euclidean: lower bound base + S isum > cut  <=  isum >= ceil((cut - base) / S) + 1;  dot product: upper bound
            // base + S (isum + M) < cut  <=  isum <= floor((cut - base) / S - M) - 1  (one bucket of margin covers the division's rounding)
            int T = (VSF == VSF_L2) ? 0x7fffffff : -1;   // never rejects
            if (on && VSF != VSF_COS) {
                const float tf = (cut - mq[0]) / mq[1] - ((VSF == VSF_DOT) ? (float)p.M_total : 0.0f);
                if (tf == tf && fabsf(tf) < 1e9f) T = (VSF == VSF_L2) ? (int)ceilf(tf) + 1 : (int)floorf(tf) - 1;
            }
            s_T[tid] = T;
        }
    }
    __syncthreads();
    const float Mf = (float)p.M_total;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t i = tile_base + (int64_t)r * 1024 + tid;
        if (i >= p.count) continue;
        const int64_t row = p.first + i;
        const float nrm = (VSF == VSF_COS) ? p.norms[row] : 0.0f;
#pragma unroll
        for (int j = 0; j < BQ_P; ++j) {
            const int on = s_on[j];
            if (on == 0) continue;
            const uint32_t word = acc[r][2 * (j >> 2) + (j & 1)];
            const int isum_i = (int)((word >> (16 * ((j >> 1) & 1))) & 0xFFFFu);
            const float isum = (float)isum_i;
            bool reject = false;
            if (VSF == VSF_L2) {
                reject = isum_i >= s_T[j];                                                     // lower bound of the distance above the cut
            } else if (VSF == VSF_DOT) {
                reject = isum_i <= s_T[j];                                                     // upper bound of the raw sum below the cut
            } else if (on == 1) {
                {
                    const float ub = s_qbase[j] + s_qS[j] * (isum + Mf);
                    const float prod = nrm * s_bm[j];
                    const float cc = ub * __builtin_amdgcn_rsqf(prod);                          // ~cosine bound, few ulps
                    reject = prod > 1e-30f && prod < 1e30f && cc < s_cut[j] - 8e-6f * (fabsf(cc) + 1.0f);
                }
            }
            if (reject) continue;
            const unsigned int sp = atomicAdd(&s_cnt[j], 1u);
            if (sp < (unsigned int)STAGE) {
                st_ids[j * STAGE + sp] = (int32_t)row;
            } else {   // staging area full: straight to the global list
                const unsigned int pos = atomicAdd(&p.surv_cnt[q0 + j], 1u);
                if (pos < (unsigned int)p.cap2) p.surv_ids[(int64_t)(q0 + j) * p.cap2 + pos] = (int32_t)row;
            }
        }
    }
    __syncthreads();
    if (tid < BQ_P && q0 + tid < p.Q) {
        const unsigned int n = s_cnt[tid] < (unsigned int)STAGE ? s_cnt[tid] : (unsigned int)STAGE;
        s_base[tid] = n ? atomicAdd(&p.surv_cnt[q0 + tid], n) : 0u;
    }
    __syncthreads();
    for (int j = 0; j < BQ_P; ++j) {
        const int q = q0 + j;
        if (q >= p.Q) break;
        const unsigned int n = s_cnt[j] < (unsigned int)STAGE ? s_cnt[j] : (unsigned int)STAGE, base = s_base[j];
        for (unsigned int t = tid; t < n; t += 1024) {
            const unsigned int pos = base + t;
            if (pos < (unsigned int)p.cap2) p.surv_ids[(int64_t)q * p.cap2 + pos] = st_ids[j * STAGE + t];
        }
    }
}

// how many of a query's exactly scored survivors reach its threshold (the caller needs >= rerankK of them)
__global__ __launch_bounds__(256) void adc_bq_count_kernel(const float *__restrict__ scores, const unsigned int *__restrict__ surv_cnt, int cap2,
                                                           const float *__restrict__ tau, int tau_stride, unsigned int *__restrict__ out_cnt)
{
    __shared__ unsigned int s_n;
    const int q = (int)blockIdx.x;
    if (threadIdx.x == 0) s_n = 0u;
    __syncthreads();
    const unsigned int n = surv_cnt[q] < (unsigned int)cap2 ? surv_cnt[q] : (unsigned int)cap2;
    const float t = tau[(int64_t)q * tau_stride];
    unsigned int c = 0;
    for (unsigned int i = threadIdx.x; i < n; i += 256) c += scores[(int64_t)q * cap2 + i] >= t ? 1u : 0u;
    if (c) atomicAdd(&s_n, c);
    __syncthreads();
    if (threadIdx.x == 0) out_cnt[q] = s_n;
}

// (lds_per_block: the bound scan stages 16 x SLCH x 256 16-byte words — 64 KB, 128 KB when M % 32 == 0 — of dynamic LDS; a device
// that does not have them keeps the exact filter)
bool adc_bq_supported(int M, const uint8_t *d_codes, size_t lds_per_block)
{
    const size_t lds = (size_t)16 * ((M % 32 == 0) ? 2 : 1) * kClusters * sizeof(uint4);
    return M % 16 == 0 && M <= 192 && (reinterpret_cast<uintptr_t>(d_codes) & 15) == 0 && lds + 1024 <= lds_per_block;
}
size_t adc_bq_scratch_bytes(int Q, int M) { return (size_t)((Q + BQ_P - 1) / BQ_P) * M * kClusters * BQ_P + sizeof(float) * 4 * (size_t)Q + 256; }

template <int VSF, int SLCH>
static int launch_bq_r(hipStream_t s, const AdcBqParams &p, int R)
{
    const size_t lds = (size_t)16 * SLCH * kClusters * sizeof(uint4);
#define JV_BQ(RR)                                                                                              \
    do {                                                                                                       \
        auto kfn = adc_bq_kernel<VSF, SLCH, RR>;                                                               \
        JV_HIP_CHECK(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        dim3 grid((p.Q + BQ_P - 1) / BQ_P, (unsigned)((p.count + 1024 * RR - 1) / (1024 * RR)));               \
        hipLaunchKernelGGL(kfn, grid, dim3(1024), lds, s, p);                                                  \
    } while (0)
    if (R >= 8) JV_BQ(8);
    else JV_BQ(4);
#undef JV_BQ
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

// Stage 1 of the flat search's filter: bound tables of the batch, then the bound scan.  d_ids: [Q][cap2] survivors (unordered; slots
// behind a query's count keep whatever they held), d_surv_cnt[q] = survivors of the bound (may exceed cap2: the caller falls back).
// d_work: adc_bq_scratch_bytes(Q, M).
int launch_adc_bq_scan(hipStream_t s, const jv_ctx *ctx, const float *d_luts, const float *d_bmag, int Q, int M, int vsf, const uint8_t *d_codes,
                       const float *d_norms, int64_t first, int64_t count, const float *d_tau, int tau_stride, int32_t *d_ids,
                       unsigned int *d_surv_cnt, int cap2, void *d_work)
{
    if (Q == 0 || count == 0) return JV_OK;
    uint8_t *d_tab = (uint8_t *)d_work;
    float *d_meta = (float *)((char *)d_work + (((size_t)((Q + BQ_P - 1) / BQ_P) * M * kClusters * BQ_P + 255) & ~(size_t)255));
    JV_HIP_CHECK(hipMemsetAsync(d_surv_cnt, 0, sizeof(unsigned int) * (size_t)Q, s));
    switch (vsf) {
    case VSF_L2: hipLaunchKernelGGL(adc_bq_table_kernel<VSF_L2>, dim3(Q), dim3(256), 0, s, d_luts, Q, M, d_tab, d_meta); break;
    case VSF_DOT: hipLaunchKernelGGL(adc_bq_table_kernel<VSF_DOT>, dim3(Q), dim3(256), 0, s, d_luts, Q, M, d_tab, d_meta); break;
    default: hipLaunchKernelGGL(adc_bq_table_kernel<VSF_COS>, dim3(Q), dim3(256), 0, s, d_luts, Q, M, d_tab, d_meta); break;
    }
    JV_HIP_CHECK(hipGetLastError());
    AdcBqParams p{};
    p.tab = d_tab; p.meta = d_meta; p.bmag = d_bmag; p.codes = d_codes; p.norms = d_norms; p.tau = d_tau; p.tau_stride = tau_stride;
    p.surv_ids = d_ids; p.surv_cnt = d_surv_cnt; p.cap2 = cap2; p.first = first; p.count = count; p.Q = Q; p.M_total = M;
    const int slch = (M % 32 == 0) ? 2 : 1;
    const int64_t groups = (Q + BQ_P - 1) / BQ_P;
    // tile-wise XCD order when a tile's query groups are few (<= 32: C4 / flat_mode, 16 groups — FETCH_SIZE 24x -> 5x one pass over the
    // codes); with many groups (C2: 64) one XCD would cycle through every group's table per tile and the plain order, where an XCD
    // keeps the tables of ITS eight groups, misses less (9x against 28x; profiles/traffic_r6.json).  Neither changes the scan's time.
    p.xcd_order = (getenv("JVECTOR_HIP_ADC_BQ_PLAIN_ORDER") || groups > 32) ? 0 : 1;
    const int R = groups * ((count + 1024 * 8 - 1) / (1024 * 8)) >= 2 * (int64_t)ctx->num_cus ? 8 : 4;
#define JV_BQV(V)                                                      \
    do {                                                               \
        if (slch == 2) JV_TRY((launch_bq_r<V, 2>(s, p, R)));           \
        else JV_TRY((launch_bq_r<V, 1>(s, p, R)));                     \
    } while (0)
    switch (vsf) {
    case VSF_L2: JV_BQV(VSF_L2); break;
    case VSF_DOT: JV_BQV(VSF_DOT); break;
    default: JV_BQV(VSF_COS); break;
    }
#undef JV_BQV
    return JV_OK;
}

// Stage 2: exact ADC scores of the first `slots` survivors of every query (k_adc.hip's gather kernel: ids outside the code store score
// -inf, so stale slots behind a query's own count are harmless), then d_cnt[q] = how many of the query's survivors reach tau.
int launch_adc_bq_exact(hipStream_t s, const jv_ctx *ctx, const float *d_luts, const float *d_bmag, int Q, int M, int vsf, const uint8_t *d_codes,
                        const float *d_norms, int64_t n_codes, const float *d_tau, int tau_stride, const int32_t *d_ids, float *d_scores,
                        const unsigned int *d_surv_cnt, unsigned int *d_cnt, int cap2, int slots)
{
    if (Q == 0) return JV_OK;
    JV_TRY(launch_adc_pitched(s, ctx, d_luts, d_bmag, Q, M, vsf, d_codes, d_norms, n_codes, slots, cap2, d_ids, d_scores));
    hipLaunchKernelGGL(adc_bq_count_kernel, dim3(Q), dim3(256), 0, s, d_scores, d_surv_cnt, cap2, d_tau, tau_stride, d_cnt);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

}  // namespace jv
