// k_gsearch_ubr.hip — the one-wave traversal with a REGISTER-resident upper-bound table (gs_body.h "UBR") and the dense kernel that
// builds the tables of a whole batch.  A translation unit of its own: the traversal's other instantiations compile in parallel.
//
// Why: round 4's UB8 form (gone from the source since round 6) proved that 60 % of the neighbours a search scores can soundly be dropped behind an 8-bit upper bound of
// their score, and lost 2.3x to three removable overheads — the table built inside the traversal wave, 24 KB of LDS per wave, and
// survivors scored in place.  Here the table is built by ubr_table_kernel for all queries at once (codebook rows reused across 8
// queries, ~25 GB of L2 reads and 3.2 GB of HBM writes per 131 072 queries instead of 355 k clocks per query), lives in 96 of the
// wave's registers (two ds_bpermute_b32 per look-up: the LDS crossbar, no LDS bytes, so 8 waves per CU and the visited set's LDS
// tier stay), and what the bound cannot drop is compacted through LDS and scored eight lanes per neighbour.
#include "jv_device.h"
#include "jv_internal.h"

#define GS_WAVE_SCOPE_BARRIER 1
#define GS_UNIFORM_SHFL 1
#include "gs_wave_hip.h"

#include "gs_body.h"

namespace jv {

static_assert(VSF_L2 == 0 && VSF_DOT == 1 && VSF_COS == 2, "gs_body.h hard-codes the kernel vsf numbering");

template <int VSF, int CH16, bool PROF>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void graph_search_ubr_kernel(GsParams p)
{
    extern __shared__ __attribute__((aligned(16))) char gs_lds[];
    gs_worker<VSF, CH16, true, PROF, false, false, true>(p, (int)blockIdx.x, gs_lds);
}

// the same bound form over the COMPACTED fresh list of rows up to 64 wide, codes by ordinal (gs_body.h PAIRC + UBR): the builder's searches
template <int VSF, int CH16, bool PROF>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void graph_search_ubrc_kernel(GsParams p)
{
    extern __shared__ __attribute__((aligned(16))) char gs_lds[];
    gs_worker<VSF, CH16, false, PROF, false, true, true>(p, (int)blockIdx.x, gs_lds);
}

// ---- the tables of a batch --------------------------------------------------------------------------------------------------------
// One block = UBR_QB queries x all M subspaces, 4 waves.  Wave w takes the steps r in [w M/8, (w + 1) M/8) — i.e. the subspaces r and
// r + M/2, whose bytes share a register pair of the table — and lane s the codes s, s + 64, s + 128, s + 192 of each: a codebook row
// is loaded once and used for every query of the block.  Pass 1: per (query, subspace) the extreme entries (DPP reductions of the
// sortable-int images); then ONE scale per query; pass 2: the entries again, their buckets, and the register pair (2r, 2r + 1) of
// every lane as one 8-byte store.  Arithmetic and layout = gs_ubr_build_ref (gs_host.h), bit for bit (tests/test_zz_ubr_gpu.py).
constexpr int UBR_QB = 8;

__device__ __forceinline__ int32_t ubr_sortable(float f)
{
    const int32_t b = __builtin_bit_cast(int32_t, f);
    return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float ubr_unsortable(int32_t s) { return __builtin_bit_cast(float, s ^ ((s >> 31) & 0x7fffffff)); }

// min of a and max of b over the wave, the two DPP chains interleaved (row_shr 1, 2, 4, 8 inside the rows of 16 lanes, row_bcast15 /
// row_bcast31 across them): lane 63 ends up with both results.  A DPP operand must not be read within two wait states of the VALU
// write that produced it: the partner chain's instruction is one, an s_nop the other.  All 64 lanes must be active.
__device__ __forceinline__ void ubr_wave_minmax(float &a, float &b)
{
#define JV_MM_STEP(CTRL)                                      \
    "v_min_f32_dpp %0, %0, %0 " CTRL "\n"                     \
    "v_max_f32_dpp %1, %1, %1 " CTRL "\n"                     \
    "s_nop 0\n"
    asm volatile("s_nop 1\n" JV_MM_STEP("row_shr:1 row_mask:0xf bank_mask:0xf") JV_MM_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
                     JV_MM_STEP("row_shr:4 row_mask:0xf bank_mask:0xf") JV_MM_STEP("row_shr:8 row_mask:0xf bank_mask:0xf")
                         JV_MM_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf") JV_MM_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(a), "+v"(b));
#undef JV_MM_STEP
}

template <int VSF>
__global__ __launch_bounds__(256) void ubr_table_kernel(const float *__restrict__ codebooks, const float *__restrict__ cq, int Q, int M,
                                                        uint32_t *__restrict__ tab, float *__restrict__ meta)
{
    extern __shared__ __attribute__((aligned(16))) char ubr_lds[];
    const int D = 8 * M;
    float *qs = reinterpret_cast<float *>(ubr_lds);                 // [UBR_QB][D]
    float *lo = qs + UBR_QB * D;                                    // [UBR_QB][M]
    float *hi = lo + UBR_QB * M;                                    // [UBR_QB][M]
    float *qS = hi + UBR_QB * M;                                    // [UBR_QB] scale
    int *qbad = reinterpret_cast<int *>(qS + UBR_QB);               // [UBR_QB] a non-finite entry was seen
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = (int)blockIdx.x * UBR_QB;
    const int nq = min(UBR_QB, Q - q0);
    for (int i = tid; i < UBR_QB * D / 4; i += 256) {
        const int j = i / (D / 4);
        gs_f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (j < nq) v = reinterpret_cast<const gs_f4 *>(cq + (int64_t)(q0 + j) * D)[i - j * (D / 4)];
        reinterpret_cast<gs_f4 *>(qs)[i] = v;
    }
    if (tid < UBR_QB) qbad[tid] = 0;
    __syncthreads();
    const int H = M / 2, per_wave = H / 4;
    // ---- pass 1: lo / hi of every (query, subspace) ----
    // Round 6: 170 -> ~95 instructions per (query, subspace).  The extremes are float minima / maxima (v_min3 / v_max3 per lane, then
    // ONE interleaved chain of v_min_f32_dpp / v_max_f32_dpp: 12 DPP operations + 7 s_nop, where the compiler's rendering of two
    // update_dpp reductions over order-preserving integers took 56); a non-finite entry is found through the largest |entry| bit
    // pattern of each query, accumulated over ALL the wave's subspaces and voted on once (was: four f - f tests and a ballot per
    // pair); the wave's results ride in the lanes of 16 registers and reach LDS once per wave.
    float lo_acc[UBR_QB], hi_acc[UBR_QB];
    uint32_t amax[UBR_QB];
#pragma unroll
    for (int j = 0; j < UBR_QB; ++j) {
        lo_acc[j] = 0.0f;
        hi_acc[j] = 0.0f;
        amax[j] = 0u;
    }
    for (int rr = 0; rr < per_wave; ++rr) {
        const int r = wave * per_wave + rr;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int m = r + half * H;
            gs_f4 c0[4], c1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const gs_f4 *cp = reinterpret_cast<const gs_f4 *>(codebooks + ((int64_t)m * 256 + lane + 64 * k) * 8);
                c0[k] = cp[0];
                c1[k] = cp[1];
            }
            const int slot = 2 * rr + half;   // this wave's slot of the pair: lane `slot` of lo_acc[j] / hi_acc[j]
#pragma unroll
            for (int j = 0; j < UBR_QB; ++j) {
                const float *q = qs + j * D + m * 8;
                float e4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) e4[k] = gs_lut_entry_pk<VSF>(c0[k], c1[k], q);
                // (NaN entries are skipped by v_min / v_max; the bit pattern below marks their query unusable)
                float fmn = __builtin_fminf(__builtin_fminf(e4[0], e4[1]), __builtin_fminf(e4[2], e4[3]));
                float fmx = __builtin_fmaxf(__builtin_fmaxf(e4[0], e4[1]), __builtin_fmaxf(e4[2], e4[3]));
                uint32_t a = amax[j];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t b = __builtin_bit_cast(uint32_t, e4[k]) & 0x7fffffffu;
                    a = b > a ? b : a;
                }
                amax[j] = a;
                ubr_wave_minmax(fmn, fmx);   // lane 63 holds the wave's extremes
                const float wmn = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fmn), 63));
                const float wmx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fmx), 63));
                lo_acc[j] = lane == slot ? wmn : lo_acc[j];
                hi_acc[j] = lane == slot ? wmx : hi_acc[j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < UBR_QB; ++j) {
        if (lane < 2 * per_wave) {
            const int m = wave * per_wave + (lane >> 1) + (lane & 1) * H;
            lo[j * M + m] = lo_acc[j];
            hi[j * M + m] = hi_acc[j];
        }
        if (__ballot(amax[j] >= 0x7f800000u ? 1 : 0) != 0 && lane == 0) qbad[j] = 1;   // (benign race: every writer stores 1)
    }
    __syncthreads();
    // ---- one scale per query; base = sum of the low edges + slack ----
    if (tid < UBR_QB) {
        const int j = tid;
        float range = 0.0f;
        for (int m = 0; m < M; ++m) {
            const float r = hi[j * M + m] - lo[j * M + m];
            if (r > range) range = r;
        }
        float S = range / 255.0f;
        if (!(S > 1e-30f)) S = 1e-30f;
        float sum_lo = 0.0f, sum_abs = 0.0f, max_abs = 0.0f;
        for (int m = 0; m < M; ++m) {
            const float l = lo[j * M + m], h = hi[j * M + m];
            sum_lo += l;
            const float amn = l < 0.0f ? -l : l, amx = h < 0.0f ? -h : h;
            const float a = amn > amx ? amn : amx;
            sum_abs += a + 256.0f * S;
            if (a > max_abs) max_abs = a;
        }
        // (usable only if a bucket is not lost in the rounding of an edge: lo + 256 S then bounds every entry of a subspace in f32 too)
        const bool ok = qbad[j] == 0 && (sum_abs - sum_abs == 0.0f) && S * 1e6f >= max_abs;
        qS[j] = S;
        if (!ok) qbad[j] = 1;
        if (j < nq) {
            float *mq = meta + (int64_t)(q0 + j) * 4;
            mq[0] = (VSF == VSF_L2) ? sum_lo - 4e-5f * sum_abs : sum_lo + 4e-5f * sum_abs;   // (euclidean: a LOWER bound of the distance)
            mq[1] = S;
            mq[2] = ok ? 1.0f : 0.0f;
            mq[3] = 0.0f;
        }
    }
    __syncthreads();
    // ---- pass 2: buckets, packed into the register pairs ----
    for (int rr = 0; rr < per_wave; ++rr) {
        const int r = wave * per_wave + rr;
        gs_f4 c0[2][4], c1[2][4];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const gs_f4 *cp = reinterpret_cast<const gs_f4 *>(codebooks + ((int64_t)(r + half * H) * 256 + lane + 64 * k) * 8);
                c0[half][k] = cp[0];
                c1[half][k] = cp[1];
            }
        }
        for (int j = 0; j < nq; ++j) {
            const float S = qS[j], inv = 1.0f / S;
            uint32_t b[2][4];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int m = r + half * H;
                const float l = lo[j * M + m];
                const float *q = qs + j * D + m * 8;
                float e4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) e4[k] = gs_lut_entry_pk<VSF>(c0[half][k], c1[half][k], q);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float e = e4[k];
                    // (gs_host.h gs_ubr_build_ref: floor of the f32 quotient plus 2^-10 — its upper edge bounds the entry)
                    int bb = (VSF == VSF_L2) ? (int)((e - l) * inv - 0x1p-10f) : (int)((e - l) * inv + 0x1p-10f);   // (L2: lower bucket edges)
                    bb = bb < 0 ? 0 : (bb > 255 ? 255 : bb);
                    b[half][k] = (uint32_t)bb;
                }
            }
            gs_u2 out;
            out.x = b[0][0] | (b[0][1] << 8) | (b[1][0] << 16) | (b[1][1] << 24);   // register 2r:     codes s, s + 64
            out.y = b[0][2] | (b[0][3] << 8) | (b[1][2] << 16) | (b[1][3] << 24);   // register 2r + 1: codes s + 128, s + 192
            if (qbad[j]) out = gs_u2{0u, 0u};
            const int k0 = 2 * r;   // even: the pair (k0, k0 + 1) is the lower or the upper half of one 16-byte group
            uint32_t *dst = tab + (int64_t)(q0 + j) * M * 64 + ((int64_t)(k0 / 4) * 64 + lane) * 4 + (k0 % 4);
            *reinterpret_cast<gs_u2 *>(dst) = out;
        }
    }
}

size_t ubr_table_lds_bytes(int M) { return sizeof(float) * ((size_t)UBR_QB * 8 * M + 2 * (size_t)UBR_QB * M + UBR_QB) + sizeof(int) * UBR_QB; }

bool graph_search_ubr_supported(int M, int /*vsf*/) { return M == 96; }   // (round 6: euclidean too — lower bucket edges, gs_host.h)

// tables + meta of queries [0, Q): tab = Q x gs_ubr_tab_bytes(M), meta = Q x 4 floats
int launch_ubr_tables(hipStream_t s, int vsf, const float *codebooks, const float *cq, int Q, int M, uint32_t *tab, float *meta)
{
    if (Q == 0) return JV_OK;
    if (M % 8 != 0) {
        set_error("ubr tables: M a multiple of 8 (M = %d)", M);
        return JV_ERR_INVALID;
    }
    const size_t lds = ubr_table_lds_bytes(M);
    dim3 grid((unsigned)((Q + UBR_QB - 1) / UBR_QB)), block(256);
    if (vsf == VSF_L2) {
        JV_HIP_CHECK(hipFuncSetAttribute((const void *)ubr_table_kernel<VSF_L2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(ubr_table_kernel<VSF_L2>, grid, block, lds, s, codebooks, cq, Q, M, tab, meta);
    } else if (vsf == VSF_DOT) {
        JV_HIP_CHECK(hipFuncSetAttribute((const void *)ubr_table_kernel<VSF_DOT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(ubr_table_kernel<VSF_DOT>, grid, block, lds, s, codebooks, cq, Q, M, tab, meta);
    } else {
        JV_HIP_CHECK(hipFuncSetAttribute((const void *)ubr_table_kernel<VSF_COS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(ubr_table_kernel<VSF_COS>, grid, block, lds, s, codebooks, cq, Q, M, tab, meta);
    }
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

int launch_graph_search_ubr(hipStream_t s, int vsf, const GsParams &p, int workers, size_t lds)
{
    if (p.Q == 0) return JV_OK;
    if (!p.pair || p.session || p.generic || !graph_search_ubr_supported(p.M, vsf) || !p.ubr_tab || !p.ubr_meta) {
        set_error("graph search kernel: the register-table bound form serves the pair-lane kernels at M = 96");
        return JV_ERR_INVALID;
    }
    dim3 grid(workers), block(64);
#define JV_UBR(VSFV)                                                                                             \
    do {                                                                                                         \
        if (p.pair == 2) {   /* over the compacted fresh list (no phase-clock build) */                           \
            hipLaunchKernelGGL((graph_search_ubrc_kernel<VSFV, 6, false>), grid, block, lds, s, p);               \
        } else if (p.prof) hipLaunchKernelGGL((graph_search_ubr_kernel<VSFV, 6, true>), grid, block, lds, s, p);  \
        else hipLaunchKernelGGL((graph_search_ubr_kernel<VSFV, 6, false>), grid, block, lds, s, p);              \
    } while (0)
    if (vsf == VSF_L2) JV_UBR(VSF_L2);
    else if (vsf == VSF_DOT) JV_UBR(VSF_DOT);
    else JV_UBR(VSF_COS);
#undef JV_UBR
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

}  // namespace jv
