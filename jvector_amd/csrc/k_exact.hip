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

int launch_exact_gather(hipStream_t s, const float *d_vecs, int64_t n, int D, const float *d_q, int Q, int vsf,
                        const int32_t *d_ord, int B, float *d_out, float *d_qnorm)
{
    if (Q == 0 || B == 0) return JV_OK;
    // d_qnorm: caller-provided scratch of Q floats (query-side cosine norms)
    if (vsf == VSF_COS)
        hipLaunchKernelGGL(query_sqnorm_kernel, dim3((Q + 63) / 64), dim3(64), 0, s, d_q, D, Q, d_qnorm);
    dim3 grid(Q, (B + 63) / 64), block(64);
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
