// k_exact.hip — full-resolution float32 dot / L2 / cosine scoring.
//
// One lane = one candidate vector; the lane reproduces the scalar reference's accumulation order
// (DefaultVectorUtilSupport.java:38-105 dot, :158-193 L2, :121-139 cosine) with non-fused mul/add, so
// every score is bit-identical to VectorSimilarityFunction.compare on the Default provider.  The
// per-candidate chain is sequential by definition of that order; parallelism comes from 64 candidates
// per wave x many waves, and (scan form) from scoring QB queries per pass over a candidate row so the
// row is read from HBM once per QB queries.
//
//   gather form (rerank):  grid (Q, ceil(B/64)), block 64.  query in LDS (broadcast reads).
//   scan form (brute force / ground truth): grid (ceil(count/256)), block 256, loops over query tiles.
#include <algorithm>
#include <cstdlib>

#include "gs_params.h"
#include "jv_device.h"
#include "jv_internal.h"

namespace jv {

// --- per-8-block updates in the reference's association order -----------------------------------
__device__ __forceinline__ float dot8(const float *__restrict__ a, const float4 v0, const float4 v1)
{
    // b[i+0]*a[i+0] + b[i+1]*a[i+1] + ... left to right   (a = query, b = candidate; product commutes)
    float t = v0.x * a[0] + v0.y * a[1];
    t = t + v0.z * a[2];
    t = t + v0.w * a[3];
    t = t + v1.x * a[4];
    t = t + v1.y * a[5];
    t = t + v1.z * a[6];
    t = t + v1.w * a[7];
    return t;
}

__device__ __forceinline__ float l28(const float *__restrict__ a, const float4 v0, const float4 v1)
{
    const float d0 = a[0] - v0.x, d1 = a[1] - v0.y, d2 = a[2] - v0.z, d3 = a[3] - v0.w;
    const float d4 = a[4] - v1.x, d5 = a[5] - v1.y, d6 = a[6] - v1.z, d7 = a[7] - v1.w;
    float t = d0 * d0 + d1 * d1;
    t = t + d2 * d2;
    t = t + d3 * d3;
    t = t + d4 * d4;
    t = t + d5 * d5;
    t = t + d6 * d6;
    t = t + d7 * d7;
    return t;
}

// Scores one candidate row against one query held in (LDS) memory `a`.  Generic in D.
template <int VSF>
__device__ __forceinline__ float exact_row(const float *__restrict__ a, const float *__restrict__ b, int D)
{
    if (VSF == VSF_DOT) {
        float res = 0.0f;
        int i = 0;
        const int rem = D % 8;
        for (; i < rem; ++i) res += b[i] * a[i];
        if (D < 8) return res;
        if (rem == 0 && ((reinterpret_cast<uintptr_t>(b) & 15) == 0)) {
            for (; i + 7 < D; i += 8) {
                const float4 v0 = *reinterpret_cast<const float4 *>(b + i);
                const float4 v1 = *reinterpret_cast<const float4 *>(b + i + 4);
                res += dot8(a + i, v0, v1);
            }
        } else {
            for (; i + 7 < D; i += 8) {
                const float4 v0 = make_float4(b[i], b[i + 1], b[i + 2], b[i + 3]);
                const float4 v1 = make_float4(b[i + 4], b[i + 5], b[i + 6], b[i + 7]);
                res += dot8(a + i, v0, v1);
            }
        }
        return res;
    } else if (VSF == VSF_L2) {
        float sq = 0.0f;
        int i = 0;
        if ((reinterpret_cast<uintptr_t>(b) & 15) == 0) {
            for (; i + 8 <= D; i += 8) {
                const float4 v0 = *reinterpret_cast<const float4 *>(b + i);
                const float4 v1 = *reinterpret_cast<const float4 *>(b + i + 4);
                sq += l28(a + i, v0, v1);
            }
        } else {
            for (; i + 8 <= D; i += 8) {
                const float4 v0 = make_float4(b[i], b[i + 1], b[i + 2], b[i + 3]);
                const float4 v1 = make_float4(b[i + 4], b[i + 5], b[i + 6], b[i + 7]);
                sq += l28(a + i, v0, v1);
            }
        }
        for (; i < D; ++i) {
            const float d = a[i] - b[i];
            sq += d * d;
        }
        return sq;
    } else {
        // cosine: three sequential accumulators; norm1 (query side) is candidate independent and is
        // passed in pre-accumulated by the caller, here we return sum and norm2 packed by reference.
        return 0.0f;
    }
}

__device__ __forceinline__ void cosine_row(const float *__restrict__ a, const float *__restrict__ b, int D,
                                           float &sum, float &norm2)
{
    float s = 0.0f, n2 = 0.0f;
    int i = 0;
    if ((reinterpret_cast<uintptr_t>(b) & 15) == 0) {
        for (; i + 4 <= D; i += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(b + i);
            s += a[i] * v.x;
            n2 += v.x * v.x;
            s += a[i + 1] * v.y;
            n2 += v.y * v.y;
            s += a[i + 2] * v.z;
            n2 += v.z * v.z;
            s += a[i + 3] * v.w;
            n2 += v.w * v.w;
        }
    }
    for (; i < D; ++i) {
        const float e2 = b[i];
        s += a[i] * e2;
        n2 += e2 * e2;
    }
    sum = s;
    norm2 = n2;
}

// norm1[q] = sequential sum of e1*e1 (DefaultVectorUtilSupport.cosine :131-137), one thread per query
__global__ void query_sqnorm_kernel(const float *__restrict__ q, int D, int Q, float *__restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Q) return;
    const float *v = q + (int64_t)i * D;
    float n1 = 0.0f;
    for (int j = 0; j < D; ++j) n1 += v[j] * v[j];
    out[i] = n1;
}

template <int VSF>
__global__ __launch_bounds__(64) void exact_gather_kernel(const float *__restrict__ vecs, int64_t n, int D,
                                                          const float *__restrict__ queries,
                                                          const float *__restrict__ qnorm,
                                                          const int32_t *__restrict__ ord, int B,
                                                          float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float qs[];
    const int q = blockIdx.x;
    for (int j = threadIdx.x; j < D; j += 64) qs[j] = queries[(int64_t)q * D + j];
    __syncthreads();
    const int j = blockIdx.y * 64 + threadIdx.x;
    if (j >= B) return;
    const int64_t o = ord[(int64_t)q * B + j];
    float *dst = out + (int64_t)q * B + j;
    if (o < 0 || o >= n) {
        *dst = -INFINITY;
        return;
    }
    const float *b = vecs + o * D;
    float raw;
    if (VSF == VSF_COS) {
        float sum, norm2;
        cosine_row(qs, b, D, sum, norm2);
        raw = cosine_finish(sum, qnorm[q], norm2);
    } else {
        raw = exact_row<VSF>(qs, b, D);
    }
    *dst = score_from_raw(VSF, raw);
}


// ------------------------------------------------------------------------------------------------
// Transposing gather form (D % 8 == 0, 16-byte aligned rows): HBM is read with full-line coalescing and the per-lane
// sequential chains run out of LDS.
//
// One wavefront = 64 candidates of one query.  A row is consumed in chunks of 64 floats: in the LOAD phase the wave
// reads 4 rows x 256 B per instruction (16 lanes x 16 B per row: whole 128-byte lines, 16 instructions = 16 KB in
// flight per wave) and parks the chunk in LDS as tile[row][64 + 4 pad]; in the COMPUTE phase lane j walks row j of the
// tile with ds_read_b128 (row stride 68 dwords: the four 16-lane groups of a b128 read hit 64 distinct banks) and runs
// the reference's chain against the query, which is wave-uniform and comes through scalar loads.  The loads of
// chunk c+1 are issued before chunk c is computed, so HBM latency hides behind the chain.
// The old lane-per-row kernels touched 64 different lines per wave-load and used 16 B of each (0.17 of HBM peak).
// Cosine: norm2 = sum of e2*e2 (DefaultVectorUtilSupport.cosine :131-137) is query independent and a separate
// accumulator, so it is read from a per-row table built once by row_sqnorm_kernel in the same order (same bits).
// ------------------------------------------------------------------------------------------------
constexpr int TR_CH = 64;          // floats of a row per chunk (the norm-table kernel and the default rerank shape)
constexpr int TR_LS = TR_CH + 4;   // LDS row stride (dwords)

// chain over one chunk of `len` floats (multiple of 8) of the lane's row in LDS against the uniform query slice a[]
template <int VSF, int CH>
__device__ __forceinline__ void tr_chunk(const float *__restrict__ row, const float *__restrict__ a, int len, float &acc)
{
    if (len == CH) {
#pragma unroll
        for (int i = 0; i < CH; i += 8) {
            const float4 v0 = *reinterpret_cast<const float4 *>(row + i);
            const float4 v1 = *reinterpret_cast<const float4 *>(row + i + 4);
            if (VSF == VSF_DOT) acc += dot8(a + i, v0, v1);
            else if (VSF == VSF_L2) acc += l28(a + i, v0, v1);
            else {
                float s = acc;
                s += a[i + 0] * v0.x; s += a[i + 1] * v0.y; s += a[i + 2] * v0.z; s += a[i + 3] * v0.w;
                s += a[i + 4] * v1.x; s += a[i + 5] * v1.y; s += a[i + 6] * v1.z; s += a[i + 7] * v1.w;
                acc = s;
            }
        }
    } else {
        for (int i = 0; i < len; i += 8) {
            const float4 v0 = *reinterpret_cast<const float4 *>(row + i);
            const float4 v1 = *reinterpret_cast<const float4 *>(row + i + 4);
            if (VSF == VSF_DOT) acc += dot8(a + i, v0, v1);
            else if (VSF == VSF_L2) acc += l28(a + i, v0, v1);
            else {
                float s = acc;
                s += a[i + 0] * v0.x; s += a[i + 1] * v0.y; s += a[i + 2] * v0.z; s += a[i + 3] * v0.w;
                s += a[i + 4] * v1.x; s += a[i + 5] * v1.y; s += a[i + 6] * v1.z; s += a[i + 7] * v1.w;
                acc = s;
            }
        }
    }
}

// SQ = true: acc = sum of e*e over the row (no query) — builds the cosine norm table.
// R rows per wavefront (lanes >= R carry none), CH floats of a row per chunk, R x CH = 4096: sixteen 1 KB load instructions per
// chunk in every shape — 64 x 64 (four rows x 256 B per instruction), 32 x 128 (two rows x 512 B), 16 x 256 (one row x 1 KB): the
// longer a row's contiguous piece, the fewer DRAM pages a gathered row opens.  LDS row stride CH + 4 dwords: conflict-free b128 reads.
// SQ8: the squares summed in BLOCKS OF EIGHT — t = e0 e0 + e1 e1, t += e2 e2 ... t += e7 e7, acc += t: the order of the dot product of a
// vector with itself (DefaultVectorUtilSupport.dotProduct, k_pq.hip dot_full_order at D % 8 == 0) and of the per-subspace partial sums
// of 8-float sub-vectors (query_mag_kernel kind 1) alike.
template <int VSF, bool SQ, int R = 64, int CH = TR_CH, bool SQ8 = false>
__device__ __forceinline__ float tr_rows(const float *__restrict__ vecs, int D, int64_t my_row /* -1 = none */,
                                         const float *__restrict__ a, float *tile)
{
    static_assert(R * CH == 4096 && (CH == 64 || CH == 128 || CH == 256), "sixteen 1 KB load instructions per chunk");
    constexpr int LS = CH + 4, LPR = CH / 4, RPI = 64 / LPR;   // lanes per row piece, rows per load instruction
    const int lane = threadIdx.x;
    const int seg = (lane % LPR) * 4;   // this lane's 16 bytes inside a row's chunk
    const int sub = lane / LPR;         // load instruction k fetches rows RPI k + sub
    const float *rp[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int64_t ro = __shfl(my_row, RPI * k + sub, 64);
        rp[k] = ro >= 0 ? vecs + ro * D + seg : nullptr;
    }
    const int nc = (D + CH - 1) / CH;
    float4 r[16];
    auto issue = [&](int c) {
        const bool in = c * CH + seg < D;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            r[k] = (rp[k] && in) ? *reinterpret_cast<const float4 *>(rp[k] + c * CH) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    issue(0);
    float acc = 0.0f;
    for (int c = 0; c < nc; ++c) {
#pragma unroll
        for (int k = 0; k < 16; ++k) *reinterpret_cast<float4 *>(tile + (RPI * k + sub) * LS + seg) = r[k];
        __syncthreads();
        if (c + 1 < nc) issue(c + 1);
        const int len = (D - c * CH < CH) ? (D - c * CH) : CH;
        const float *row = tile + (lane % R) * LS;
        if (SQ && SQ8) {
            for (int i = 0; i < len; i += 8) {
                const float4 v0 = *reinterpret_cast<const float4 *>(row + i), v1 = *reinterpret_cast<const float4 *>(row + i + 4);
                float t = v0.x * v0.x + v0.y * v0.y;
                t = t + v0.z * v0.z;
                t = t + v0.w * v0.w;
                t = t + v1.x * v1.x;
                t = t + v1.y * v1.y;
                t = t + v1.z * v1.z;
                t = t + v1.w * v1.w;
                acc += t;
            }
        } else if (SQ) {
            for (int i = 0; i < len; i += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(row + i);
                acc += v.x * v.x; acc += v.y * v.y; acc += v.z * v.z; acc += v.w * v.w;
            }
        } else if (lane < R) {
            tr_chunk<VSF, CH>(row, a + c * CH, len, acc);
        }
        __syncthreads();
    }
    return acc;
}

template <int VSF, int R, int CH>
__global__ __launch_bounds__(64) void exact_gather_tr_kernel(const float *__restrict__ vecs, int64_t n, int D,
                                                             const float *__restrict__ queries,
                                                             const float *__restrict__ qnorm,
                                                             const float *__restrict__ vnorm,
                                                             const int32_t *__restrict__ ord, int B,
                                                             float *__restrict__ out, int B_rows /* rows [0, B_rows) of every list */)
{
    __shared__ __attribute__((aligned(16))) float tile[R * (CH + 4)];
    const int q = blockIdx.x;
    const bool mine = (int)threadIdx.x < R;
    const int j = blockIdx.y * R + threadIdx.x;
    int64_t o = -1;
    if (mine && j < B_rows) {
        o = ord[(int64_t)q * B + j];
        if (o >= n) o = -1;
    }
    const float raw0 = tr_rows<VSF, false, R, CH>(vecs, D, o, queries + (int64_t)q * D, tile);
    if (!mine || j >= B_rows) return;
    float *dst = out + (int64_t)q * B + j;
    if (o < 0) {
        *dst = -INFINITY;
        return;
    }
    const float raw = (VSF == VSF_COS) ? cosine_finish(raw0, qnorm[q], vnorm[o]) : raw0;
    *dst = score_from_raw(VSF, raw);
}

// ---- the REMAINDERS of several queries' lists in one wavefront (round 6) ----
// A list of B = 76 candidates is a full wavefront and one with 12 rows; the second one holds a workgroup slot about as long as the first
// and moves a fifth of the bytes: 1.9 of the rerank's 6.7 ms for 16 % of its rows (profiles/r6_m: B = 64 / 76 / 128 -> 4.77 / 6.69 /
// 8.93 ms).  Here a wavefront carries the last `rem` rows of G = 64 / rem CONSECUTIVE queries (lane = g rem + i).  The rows travel
// exactly as in tr_rows; the queries cannot come through scalar loads any more (a lane group per query), so the G query slices of a
// chunk are staged in LDS next to the tile (requested one chunk ahead, like the rows) and every lane reads ITS query's slice — the
// same chain against the same values, the same bits.
constexpr int TRQ_MAXG = 16;   // queries per wavefront (rem >= 4)
template <int VSF>
__global__ __launch_bounds__(64) void exact_gather_trq_kernel(const float *__restrict__ vecs, int64_t n, int D, const float *__restrict__ queries,
                                                              const float *__restrict__ qnorm, const float *__restrict__ vnorm,
                                                              const int32_t *__restrict__ ord, int B, float *__restrict__ out, int Q, int first,
                                                              int rem, int G)
{
    constexpr int CH = TR_CH, LS = TR_LS;
    __shared__ __attribute__((aligned(16))) float tile[64 * LS];
    __shared__ __attribute__((aligned(16))) float qsl[TRQ_MAXG * CH];
    const int lane = threadIdx.x;
    const int g = lane / rem, i = lane - g * rem;
    const int q0 = (int)blockIdx.x * G;
    const int q = q0 + g;
    const bool mine = g < G && q < Q;
    int64_t o = -1;
    if (mine) {
        o = ord[(int64_t)q * B + first + i];
        if (o >= n) o = -1;
    }
    const int seg = (lane & 15) * 4, sub = lane >> 4;
    const float *rp[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int64_t ro = __shfl(o, 4 * k + sub, 64);
        rp[k] = ro >= 0 ? vecs + ro * D + seg : nullptr;
    }
    // staging of the query slices: float4 slot s = lane + 64 j covers query s / 16, floats (s % 16) * 4 .. + 4 of the chunk
    const float *qp[TRQ_MAXG * CH / 4 / 64];
#pragma unroll
    for (int j = 0; j < TRQ_MAXG * CH / 4 / 64; ++j) {
        const int sl = lane + 64 * j, gq = sl >> 4;
        qp[j] = (gq < G && q0 + gq < Q) ? queries + (int64_t)(q0 + gq) * D + (sl & 15) * 4 : nullptr;
    }
    const int nc = (D + CH - 1) / CH;
    float4 r[16], qr[TRQ_MAXG * CH / 4 / 64];
    auto issue = [&](int c) {
        const bool in = c * CH + seg < D;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            r[k] = (rp[k] && in) ? *reinterpret_cast<const float4 *>(rp[k] + c * CH) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < TRQ_MAXG * CH / 4 / 64; ++j)
            qr[j] = (qp[j] && c * CH + (lane & 15) * 4 < D) ? *reinterpret_cast<const float4 *>(qp[j] + c * CH) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    issue(0);
    float acc = 0.0f;
    for (int c = 0; c < nc; ++c) {
#pragma unroll
        for (int k = 0; k < 16; ++k) *reinterpret_cast<float4 *>(tile + (4 * k + sub) * LS + seg) = r[k];
#pragma unroll
        for (int j = 0; j < TRQ_MAXG * CH / 4 / 64; ++j) *reinterpret_cast<float4 *>(qsl + (lane + 64 * j) * 4) = qr[j];
        __syncthreads();
        if (c + 1 < nc) issue(c + 1);
        const int len = (D - c * CH < CH) ? (D - c * CH) : CH;
        if (mine) tr_chunk<VSF, CH>(tile + lane * LS, qsl + g * CH, len, acc);
        __syncthreads();
    }
    if (!mine) return;
    float *dst = out + (int64_t)q * B + first + i;
    if (o < 0) {
        *dst = -INFINITY;
        return;
    }
    const float raw = (VSF == VSF_COS) ? cosine_finish(acc, qnorm[q], vnorm[o]) : acc;
    *dst = score_from_raw(VSF, raw);
}

// norm table: out[i] = sum_j v[i][j]^2, j ascending (the norm2 accumulator of DefaultVectorUtilSupport.cosine)
__global__ __launch_bounds__(64) void row_sqnorm_tr_kernel(const float *__restrict__ vecs, int64_t n, int D,
                                                           float *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) float tile[64 * TR_LS];
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const float s = tr_rows<VSF_COS, true>(vecs, D, i < n ? i : -1, nullptr, tile);
    if (i < n) out[i] = s;
}

// generic-D fallback of the table: one thread per row
__global__ __launch_bounds__(64) void row_sqnorm8_tr_kernel(const float *__restrict__ vecs, int64_t n, int D, float *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) float tile[64 * TR_LS];
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const float s = tr_rows<VSF_COS, true, 64, TR_CH, true>(vecs, D, i < n ? i : -1, nullptr, tile);
    if (i < n) out[i] = s;
}

__global__ void row_sqnorm_kernel(const float *__restrict__ vecs, int64_t n, int D, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *v = vecs + i * D;
    float s = 0.0f;
    for (int j = 0; j < D; ++j) s += v[j] * v[j];
    out[i] = s;
}

constexpr int kExactTrShapeDefault = 0;

bool exact_tr_supported(const float *d_vecs, int D) { return D % 8 == 0 && D >= 8 && (reinterpret_cast<uintptr_t>(d_vecs) & 15) == 0; }

int launch_row_sqnorms(hipStream_t s, const float *d_vecs, int64_t n, int D, float *d_out)
{
    if (n == 0) return JV_OK;
    if (exact_tr_supported(d_vecs, D))
        hipLaunchKernelGGL(row_sqnorm_tr_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, d_vecs, n, D, d_out);
    else
        hipLaunchKernelGGL(row_sqnorm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_vecs, n, D, d_out);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

// ---- the rerank fused into the traversal wave (gs_body.h gs_rr_round): what is left for this file ----
int exact_fused_rows(const float *d_vecs, int D, const float *d_q, int Q, int vsf, int B, const float *d_vnorm)
{
    if (!exact_tr_supported(d_vecs, D) || (reinterpret_cast<uintptr_t>(d_q) & 15) != 0 || (vsf == VSF_COS && !d_vnorm)) return 0;
    if (getenv("JVECTOR_HIP_EXACT_LANE_ROWS") || (getenv("JVECTOR_HIP_EXACT_TR_SHAPE") && atoi(getenv("JVECTOR_HIP_EXACT_TR_SHAPE")) != 0)) return 0;
    if (B < 1 || B > 64 * GS_RR_MAX_ROUNDS) return 0;
    const int rem = B % 64;
    if (B >= 64 && rem >= 4 && rem <= 32 && Q >= 2 && !getenv("JVECTOR_HIP_EXACT_NO_PACK")) return B - rem;
    return B;
}

// (round 6: 64 queries per wavefront through the transposing norm kernel — the same running sum per row, coalesced: 0.45 -> 0.1 ms per
//  131 072 queries of 768 floats; one thread walking its own 3 KB row touched 64 different lines per load instruction)
int launch_query_sqnorms(hipStream_t s, const float *d_q, int D, int Q, float *d_qnorm)
{
    if (Q == 0) return JV_OK;
    if (exact_tr_supported(d_q, D))
        hipLaunchKernelGGL(row_sqnorm_tr_kernel, dim3((unsigned)((Q + 63) / 64)), dim3(64), 0, s, d_q, (int64_t)Q, D, d_qnorm);
    else
        hipLaunchKernelGGL(query_sqnorm_kernel, dim3((Q + 63) / 64), dim3(64), 0, s, d_q, D, Q, d_qnorm);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

// out[i] = the squares of row i summed in blocks of eight (tr_rows SQ8); rows 16-byte aligned, D % 8 == 0
int launch_block8_sqnorms(hipStream_t s, const float *d_rows, int64_t n, int D, float *d_out)
{
    if (n == 0) return JV_OK;
    if (!exact_tr_supported(d_rows, D)) {
        set_error("block8_sqnorms: rows must be 16-byte aligned with D %% 8 == 0");
        return JV_ERR_INVALID;
    }
    hipLaunchKernelGGL(row_sqnorm8_tr_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, d_rows, n, D, d_out);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

// rows [first, B) of every list, several queries per wavefront (exact_gather_trq_kernel); d_qnorm is READ (launch_query_sqnorms)
int launch_exact_gather_tail(hipStream_t s, const float *d_vecs, int64_t n, int D, const float *d_q, int Q, int vsf, const int32_t *d_ord,
                             int B, int first, float *d_out, const float *d_qnorm, const float *d_vnorm)
{
    const int rem = B - first;
    if (Q == 0 || rem <= 0) return JV_OK;
    if (rem < 4 || rem > 32 || Q < 2 || !exact_tr_supported(d_vecs, D)) {
        set_error("exact_gather_tail: %d rows per list x %d queries is not a packed-remainder shape", rem, Q);
        return JV_ERR_INVALID;
    }
    const int G = std::min(TRQ_MAXG, 64 / rem);
    const dim3 grid((Q + G - 1) / G), block(64);
    switch (vsf) {
    case VSF_L2: hipLaunchKernelGGL((exact_gather_trq_kernel<VSF_L2>), grid, block, 0, s, d_vecs, n, D, d_q, d_qnorm, d_vnorm, d_ord, B, d_out, Q, first, rem, G); break;
    case VSF_DOT: hipLaunchKernelGGL((exact_gather_trq_kernel<VSF_DOT>), grid, block, 0, s, d_vecs, n, D, d_q, d_qnorm, d_vnorm, d_ord, B, d_out, Q, first, rem, G); break;
    default: hipLaunchKernelGGL((exact_gather_trq_kernel<VSF_COS>), grid, block, 0, s, d_vecs, n, D, d_q, d_qnorm, d_vnorm, d_ord, B, d_out, Q, first, rem, G); break;
    }
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

// d_vnorm: per-row sum-of-squares table (launch_row_sqnorms) or nullptr.  With it (cosine) — and always for dot / L2 —
// rows that allow it take the transposing kernel; anything else the lane-per-row kernel.
int launch_exact_gather(hipStream_t s, const float *d_vecs, int64_t n, int D, const float *d_q, int Q, int vsf,
                        const int32_t *d_ord, int B, float *d_out, float *d_qnorm, const float *d_vnorm)
{
    if (Q == 0 || B == 0) return JV_OK;
    // d_qnorm: caller-provided scratch of Q floats (query-side cosine norms)
    if (vsf == VSF_COS) JV_TRY(launch_query_sqnorms(s, d_q, D, Q, d_qnorm));
    dim3 grid(Q, (B + 63) / 64), block(64);
    if (exact_tr_supported(d_vecs, D) && (reinterpret_cast<uintptr_t>(d_q) & 15) == 0 && (vsf != VSF_COS || d_vnorm) &&
        !getenv("JVECTOR_HIP_EXACT_LANE_ROWS")) {
        // rows per wavefront x floats per chunk (see tr_rows); JVECTOR_HIP_EXACT_TR_SHAPE = 0 / 1 / 2 pins 64 x 64 / 32 x 128 / 16 x 256
        const int shape_env = getenv("JVECTOR_HIP_EXACT_TR_SHAPE") ? atoi(getenv("JVECTOR_HIP_EXACT_TR_SHAPE")) : -1;
        const int shape = shape_env >= 0 ? shape_env : kExactTrShapeDefault;
        // remainders packed (exact_gather_trq_kernel): B = 64 f + rem with 4 <= rem <= 32 and more than one query — the main launch then
        // covers the f full wavefronts of every list only (its rows past 64 f are left to the second launch)
        const int rq_rem = B % 64, rq_full = B / 64;
        const bool rq = shape == 0 && Q >= 2 && rq_rem >= 4 && rq_rem <= 32 && !getenv("JVECTOR_HIP_EXACT_NO_PACK");
        const int rq_G = rq ? std::min(TRQ_MAXG, 64 / rq_rem) : 0;
        const int B_main = rq ? 64 * rq_full : B;   // rows the main launch scores
#define JV_TR_LAUNCH(VSFV, R, CH)                                                                                                        \
    do {                                                                                                                                 \
        if (B_main > 0)                                                                                                                  \
            hipLaunchKernelGGL((exact_gather_tr_kernel<VSFV, R, CH>), dim3(Q, (B_main + R - 1) / R), block, 0, s, d_vecs, n, D, d_q, d_qnorm, \
                               d_vnorm, d_ord, B, d_out, B_main);                                                                        \
        if (rq)                                                                                                                          \
            hipLaunchKernelGGL((exact_gather_trq_kernel<VSFV>), dim3((Q + rq_G - 1) / rq_G), block, 0, s, d_vecs, n, D, d_q, d_qnorm, d_vnorm, \
                               d_ord, B, d_out, Q, 64 * rq_full, rq_rem, rq_G);                                                          \
    } while (0)
#define JV_TR_SHAPES(VSFV)                          \
    do {                                            \
        if (shape == 1) JV_TR_LAUNCH(VSFV, 32, 128); \
        else if (shape == 2) JV_TR_LAUNCH(VSFV, 16, 256); \
        else JV_TR_LAUNCH(VSFV, 64, 64);            \
    } while (0)
        switch (vsf) {
        case VSF_L2: JV_TR_SHAPES(VSF_L2); break;
        case VSF_DOT: JV_TR_SHAPES(VSF_DOT); break;
        default: JV_TR_SHAPES(VSF_COS); break;
        }
#undef JV_TR_SHAPES
#undef JV_TR_LAUNCH
        JV_HIP_CHECK(hipGetLastError());
        return JV_OK;
    }
    size_t lds = (size_t)D * sizeof(float);
    switch (vsf) {
    case VSF_L2:
        hipLaunchKernelGGL(exact_gather_kernel<VSF_L2>, grid, block, lds, s, d_vecs, n, D, d_q, d_qnorm, d_ord, B, d_out);
        break;
    case VSF_DOT:
        hipLaunchKernelGGL(exact_gather_kernel<VSF_DOT>, grid, block, lds, s, d_vecs, n, D, d_q, d_qnorm, d_ord, B, d_out);
        break;
    default:
        hipLaunchKernelGGL(exact_gather_kernel<VSF_COS>, grid, block, lds, s, d_vecs, n, D, d_q, d_qnorm, d_ord, B, d_out);
        break;
    }
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

// rows of the vector set as a query batch (the exact build-score provider scores node against node): out[p] = vecs[ord[p]];
// an ordinal outside [0, n) gives a zero row and voids that row's candidate list (ids -> -1, i.e. -inf scores downstream)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ vecs, int64_t n, int D, const int32_t *__restrict__ ord,
                                                          int P, float *__restrict__ out, int32_t *__restrict__ cand, int B)
{
    const int p = blockIdx.x;
    if (p >= P) return;
    const int64_t o = ord[p];
    const bool ok = o >= 0 && o < n;
    for (int j = threadIdx.x; j < D; j += 256) out[(int64_t)p * D + j] = ok ? vecs[o * D + j] : 0.0f;
    if (!ok)
        for (int j = threadIdx.x; j < B; j += 256) cand[(int64_t)p * B + j] = -1;
}

int launch_gather_rows(hipStream_t s, const float *d_vecs, int64_t n, int D, const int32_t *d_ord, int P, float *d_out, int32_t *d_cand,
                       int B)
{
    if (P == 0) return JV_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(P), dim3(256), 0, s, d_vecs, n, D, d_ord, P, d_out, d_cand, B);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

// ------------------------------------------------------------------------------------------------
// scan form: every candidate in [first, first+count) against all Q queries.
// block 256 lanes = 256 candidates; queries are processed in tiles of QB staged in LDS; each lane keeps
// QB accumulator sets in registers and walks its row ONCE per tile (8 floats at a time).
// ------------------------------------------------------------------------------------------------
template <int VSF, int QB>
__global__ __launch_bounds__(256) void exact_scan_kernel(const float *__restrict__ vecs, int D,
                                                         const float *__restrict__ queries,
                                                         const float *__restrict__ qnorm, int Q, int64_t first,
                                                         int64_t count, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float qs[];  // QB x D
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool active = i < count;
    const float *b = vecs + (first + (active ? i : 0)) * D;
    const bool vec_ok = (D % 8 == 0) && ((reinterpret_cast<uintptr_t>(vecs) & 15) == 0);

    for (int q0 = 0; q0 < Q; q0 += QB) {
        const int nq = (Q - q0 < QB) ? (Q - q0) : QB;
        __syncthreads();
        for (int j = threadIdx.x; j < QB * D; j += 256) {
            const int qq = j / D;
            qs[j] = (qq < nq) ? queries[(int64_t)(q0 + qq) * D + (j - qq * D)] : 0.0f;
        }
        __syncthreads();
        if (!active) continue;

        float acc[QB], acc2[QB];
#pragma unroll
        for (int t = 0; t < QB; ++t) { acc[t] = 0.0f; acc2[t] = 0.0f; }

        if (vec_ok) {
            for (int d = 0; d < D; d += 8) {
                const float4 v0 = *reinterpret_cast<const float4 *>(b + d);
                const float4 v1 = *reinterpret_cast<const float4 *>(b + d + 4);
#pragma unroll
                for (int t = 0; t < QB; ++t) {
                    const float *a = qs + t * D + d;
                    if (VSF == VSF_DOT) acc[t] += dot8(a, v0, v1);
                    else if (VSF == VSF_L2) acc[t] += l28(a, v0, v1);
                    else {
                        float s = acc[t], n2 = acc2[t];
                        s += a[0] * v0.x; n2 += v0.x * v0.x;
                        s += a[1] * v0.y; n2 += v0.y * v0.y;
                        s += a[2] * v0.z; n2 += v0.z * v0.z;
                        s += a[3] * v0.w; n2 += v0.w * v0.w;
                        s += a[4] * v1.x; n2 += v1.x * v1.x;
                        s += a[5] * v1.y; n2 += v1.y * v1.y;
                        s += a[6] * v1.z; n2 += v1.z * v1.z;
                        s += a[7] * v1.w; n2 += v1.w * v1.w;
                        acc[t] = s; acc2[t] = n2;
                    }
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < QB; ++t) {
                if (t < nq) {
                    if (VSF == VSF_COS) cosine_row(qs + t * D, b, D, acc[t], acc2[t]);
                    else acc[t] = exact_row<VSF>(qs + t * D, b, D);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < QB; ++t) {
            if (t < nq) {
                float raw = acc[t];
                if (VSF == VSF_COS) raw = cosine_finish(acc[t], qnorm[q0 + t], acc2[t]);
                out[(int64_t)(q0 + t) * count + i] = score_from_raw(VSF, raw);
            }
        }
    }
}

int launch_exact_scan(hipStream_t s, const jv_ctx *ctx, const float *d_vecs, int D, const float *d_q, int Q, int vsf,
                      int64_t first, int64_t count, float *d_out, float *d_qnorm)
{
    if (Q == 0 || count == 0) return JV_OK;
    if (vsf == VSF_COS)
        hipLaunchKernelGGL(query_sqnorm_kernel, dim3((Q + 63) / 64), dim3(64), 0, s, d_q, D, Q, d_qnorm);
    dim3 grid((unsigned)((count + 255) / 256)), block(256);
    // query tile: 16 queries per pass over a row when they fit comfortably in LDS, else 8
    const bool big = (size_t)16 * D * sizeof(float) <= 64 * 1024;
    size_t lds = (size_t)(big ? 16 : 8) * D * sizeof(float);
    if (lds > ctx->lds_per_block) {
        set_error("exact_scan: dimension %d too large for the LDS query tile", D);
        return JV_ERR_UNSUPPORTED;
    }
#define JV_SCAN_QB(V, QB)                                                                                        \
    do {                                                                                                         \
        auto kfn = exact_scan_kernel<V, QB>;                                                                     \
        if (lds > 64 * 1024)                                                                                     \
            JV_HIP_CHECK(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                             (int)lds));                                                         \
        hipLaunchKernelGGL(kfn, grid, block, lds, s, d_vecs, D, d_q, d_qnorm, Q, first, count, d_out);           \
    } while (0)
#define JV_SCAN(V)                  \
    do {                            \
        if (big) JV_SCAN_QB(V, 16); \
        else JV_SCAN_QB(V, 8);      \
    } while (0)
    switch (vsf) {
    case VSF_L2: JV_SCAN(VSF_L2); break;
    case VSF_DOT: JV_SCAN(VSF_DOT); break;
    default: JV_SCAN(VSF_COS); break;
    }
#undef JV_SCAN
#undef JV_SCAN_QB
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

}  // namespace jv
