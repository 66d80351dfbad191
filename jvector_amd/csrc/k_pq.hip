// k_pq.hip — ProductQuantization kernels: self-magnitude table, query centring, ADC LUT build,
// query magnitudes, PQ encode.  gfx950 only.  Compiled with -ffp-contract=off (see jv_device.h).
//
// Arithmetic orders follow the scalar reference exactly:
//   LUT / encode sub-distances: DefaultVectorUtilSupport offset forms (:107-119, :195-208), sequential.
#include "jv_device.h"
#include "jv_internal.h"

namespace jv {

// ------------------------------------------------------------------------------------------------
// calculatePartialSelfMagnitudes (VectorUtilSupport.java:137-142): aMag[m*k+i] = dot(c_i, c_i) sequential
// grid (M), block 256
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void self_mag_kernel(const float *__restrict__ codebooks,
                                                       const int64_t *__restrict__ cb_off,
                                                       const int *__restrict__ sizes, float *__restrict__ out)
{
    const int m = blockIdx.x, i = threadIdx.x;
    const int size = sizes[m];
    const float *c = codebooks + cb_off[m] + (int64_t)i * size;
    float sum = 0.0f;
    for (int j = 0; j < size; ++j) sum += c[j] * c[j];
    out[m * kClusters + i] = sum;
}

// paired[m][i/2][j] = {c[m][i][j], c[m][i+1][j]} (uniform sizes only; see jv_pq::d_cb_paired)
__global__ void pair_codebooks_kernel(const float *__restrict__ codebooks, int size, int64_t total, float *__restrict__ paired)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int64_t per_m = (int64_t)kClusters * size;
    const int64_t m = t / per_m, r = t - m * per_m;
    const int i = (int)(r / size), j = (int)(r - (int64_t)i * size);
    paired[m * per_m + ((int64_t)(i >> 1) * size + j) * 2 + (i & 1)] = codebooks[t];
}

int launch_self_magnitudes(hipStream_t s, const jv_pq *pq)
{
    hipLaunchKernelGGL(self_mag_kernel, dim3(pq->M), dim3(256), 0, s, pq->d_codebooks, pq->d_cb_offsets, pq->d_sizes,
                       pq->d_self_mag);
    if (pq->d_cb_paired) {
        const int64_t total = (int64_t)pq->M * kClusters * pq->max_size;
        hipLaunchKernelGGL(pair_codebooks_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, pq->d_codebooks, pq->max_size,
                           total, pq->d_cb_paired);
    }
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

// ------------------------------------------------------------------------------------------------
// centredQuery = center == null ? query : VectorUtil.sub(query, center)  (PQDecoder.java:46-47)
// ------------------------------------------------------------------------------------------------
__global__ void center_kernel(const float *__restrict__ q, const float *__restrict__ centroid, int D, int64_t total,
                              float *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float v = q[i];
    if (centroid) v = v - centroid[i % D];
    out[i] = v;
}

int launch_center_queries(hipStream_t s, const jv_pq *pq, const float *d_q, int Q, float *d_cq)
{
    int64_t total = (int64_t)Q * pq->D;
    if (total == 0) return JV_OK;
    int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(center_kernel, dim3(blocks), dim3(256), 0, s, d_q, pq->d_centroid, pq->D, total, d_cq);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

// ------------------------------------------------------------------------------------------------
// calculatePartialSums (DefaultVectorUtilSupport.java:351-365) for all M subspaces of Q queries.
// grid (M, Q), block 256: thread i owns centroid i.  The centred sub-query (<= a few dozen floats) is
// broadcast from LDS; the centroid row is streamed from L2 (codebooks are <= 1.5 MB and L2 resident).
// ------------------------------------------------------------------------------------------------
template <int VSF>
__global__ __launch_bounds__(256) void lut_build_kernel(const float *__restrict__ codebooks,
                                                        const int64_t *__restrict__ cb_off,
                                                        const int *__restrict__ sizes,
                                                        const int *__restrict__ offsets,
                                                        const float *__restrict__ cq, int D, int M,
                                                        float *__restrict__ luts)
{
    extern __shared__ __attribute__((aligned(16))) float qsub[];
    const int m = blockIdx.x, q = blockIdx.y, i = threadIdx.x;
    const int size = sizes[m], off = offsets[m];
    for (int j = i; j < size; j += 256) qsub[j] = cq[(int64_t)q * D + off + j];
    __syncthreads();
    const float *c = codebooks + cb_off[m] + (int64_t)i * size;
    float sum = 0.0f;
    if (VSF == VSF_DOT) {
        for (int j = 0; j < size; ++j) sum += c[j] * qsub[j];
    } else {
        for (int j = 0; j < size; ++j) {
            float d = c[j] - qsub[j];
            sum += d * d;
        }
    }
    luts[((int64_t)q * M + m) * kClusters + i] = sum;
}

int launch_lut_build(hipStream_t s, const jv_pq *pq, const float *d_cq, int Q, int lut_vsf, float *d_luts)
{
    if (Q == 0) return JV_OK;
    dim3 grid(pq->M, Q), block(256);
    size_t lds = (size_t)pq->max_size * sizeof(float);
    if (lut_vsf == VSF_DOT)
        hipLaunchKernelGGL(lut_build_kernel<VSF_DOT>, grid, block, lds, s, pq->d_codebooks, pq->d_cb_offsets,
                           pq->d_sizes, pq->d_offsets, d_cq, pq->D, pq->M, d_luts);
    else
        hipLaunchKernelGGL(lut_build_kernel<VSF_L2>, grid, block, lds, s, pq->d_codebooks, pq->d_cb_offsets,
                           pq->d_sizes, pq->d_offsets, d_cq, pq->D, pq->M, d_luts);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

// ------------------------------------------------------------------------------------------------
// cosine query magnitude.
//   kind 0 (PQDecoder.java:121):       bMagnitude = VectorUtil.dotProduct(cq, cq)  — FULL-vector form
//                                       (DefaultVectorUtilSupport.java:38-105: first D%8 elements one by
//                                       one, then 8-element blocks each summed left-to-right then added)
//   kind 1 (FusedPQDecoder.java:188):  sum over subspaces of the OFFSET-form dotProduct(cq,off,cq,off,size)
// One thread per query: D sequential operations (<= a few thousand), negligible.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dot_full_order(const float *__restrict__ a, const float *__restrict__ b, int n)
{
    float res = 0.0f;
    int i = 0;
    const int rem = n % 8;
    for (; i < rem; ++i) res += b[i] * a[i];
    if (n < 8) return res;
    for (; i + 7 < n; i += 8) {
        float t = b[i] * a[i] + b[i + 1] * a[i + 1];
        t = t + b[i + 2] * a[i + 2];
        t = t + b[i + 3] * a[i + 3];
        t = t + b[i + 4] * a[i + 4];
        t = t + b[i + 5] * a[i + 5];
        t = t + b[i + 6] * a[i + 6];
        t = t + b[i + 7] * a[i + 7];
        res += t;
    }
    return res;
}

__global__ void query_mag_kernel(const float *__restrict__ cq, int D, int M, const int *__restrict__ sizes,
                                 const int *__restrict__ offsets, int Q, int kind, float *__restrict__ bmag)
{
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    const float *v = cq + (int64_t)q * D;
    float r;
    if (kind == 0) {
        r = dot_full_order(v, v, D);
    } else {
        r = 0.0f;
        for (int m = 0; m < M; ++m) {
            const float *p = v + offsets[m];
            float s = 0.0f;
            for (int j = 0; j < sizes[m]; ++j) s += p[j] * p[j];
            r += s;
        }
    }
    bmag[q] = r;
}

int launch_query_magnitudes(hipStream_t s, const jv_pq *pq, const float *d_cq, int Q, int kind, float *d_bmag)
{
    if (Q == 0) return JV_OK;
    // (round 6) uniform 8-float sub-vectors: both kinds are the same chain — the squares in blocks of eight, t = e0 e0 + e1 e1, ... ,
    // res += t (kind 1's `s = 0; s += e0 e0` adds an exact zero) — and 64 queries share a wavefront through the transposing kernel of
    // k_exact.hip (0.49 -> 0.1 ms per 131 072 queries of 768 floats)
    if (pq->uniform && pq->max_size == 8 && pq->D == 8 * pq->M && exact_tr_supported(d_cq, pq->D)) return launch_block8_sqnorms(s, d_cq, Q, pq->D, d_bmag);
    hipLaunchKernelGGL(query_mag_kernel, dim3((Q + 63) / 64), dim3(64), 0, s, d_cq, pq->D, pq->M, pq->d_sizes,
                       pq->d_offsets, Q, kind, d_bmag);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

// ------------------------------------------------------------------------------------------------
// ProductQuantization.encodeTo -> encodeUnweighted -> closestCentroidIndex (ProductQuantization.java:
// 422-449, 507-520).  grid (ceil(count/256), M), block 256: thread = one vector, block = one subspace.
// The subspace's 256 centroids live in LDS (256*SIZE*4 B, 8 KB at SIZE 8) and are read as wave-wide
// broadcasts; the thread's (centred) sub-vector lives in registers.  Roofline: FP32 VALU (non-fused
// sub/mul/add, 3*256*SIZE flop per (vector, subspace)), not HBM — see DESIGN.md.
// ------------------------------------------------------------------------------------------------
// Two centroids per step: the codebook sits in LDS as PAIRS — entry (i/2, j) = {c[i][j], c[i+1][j]} — so that the two
// centroids' chains run side by side in the halves of v_pk_add_f32 / v_pk_mul_f32 (each chain still adds its squares in
// ascending j, the reference's order).  Per centroid 8 pk-sub + 8 pk-mul + 7 pk-add over two = 11.5 VALU slots + 3 for the
// compare / selects, against 18 when only the subtractions and squares pack (the kernel is VALU-issue bound: r2 ISA count).
typedef float jv_f2 __attribute__((ext_vector_type(2)));

template <int SIZE>
__global__ __launch_bounds__(256) void pq_encode_kernel(const float *__restrict__ vecs, int64_t count, int D, int M,
                                                        const float *__restrict__ codebooks,
                                                        const int64_t *__restrict__ cb_off,
                                                        const int *__restrict__ offsets,
                                                        const float *__restrict__ centroid,
                                                        uint8_t *__restrict__ codes)
{
    __shared__ __attribute__((aligned(16))) jv_f2 cb2[(kClusters / 2) * SIZE];
    const int m = blockIdx.y;
    const float *src = codebooks + cb_off[m];
    for (int j = threadIdx.x; j < kClusters * SIZE; j += 256) {
        const int i = j / SIZE, d = j - i * SIZE;
        reinterpret_cast<float *>(cb2)[((i >> 1) * SIZE + d) * 2 + (i & 1)] = src[j];
    }
    __syncthreads();

    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= count) return;
    const int off = offsets[m];
    const float *vp = vecs + n * D + off;
    float v[SIZE];
#pragma unroll
    for (int j = 0; j < SIZE; ++j) {
        float x = vp[j];
        if (centroid) x = x - centroid[off + j];  // VectorUtil.sub(vector, globalCentroid) :441-443
        v[j] = x;
    }
    int best = 0;
    float minDist = 3.4028234663852886e+38f;  // Float.MAX_VALUE
#pragma unroll 2
    for (int i2 = 0; i2 < kClusters / 2; ++i2) {
        jv_f2 s = {0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < SIZE; ++j) {
            const jv_f2 vv = {v[j], v[j]};
            const jv_f2 d = vv - cb2[i2 * SIZE + j];
            s += d * d;
        }
        if (s.x < minDist) {  // strict '<': first minimum wins; NaN never wins
            minDist = s.x;
            best = 2 * i2;
        }
        if (s.y < minDist) {
            minDist = s.y;
            best = 2 * i2 + 1;
        }
    }
    codes[n * M + m] = (uint8_t)best;
}

// The same two-centroid chains with the pairs read through the SCALAR cache: centroid values are wave-uniform, so they can sit
// in SGPR pairs and feed v_pk_add_f32 directly (one scalar operand per VALU instruction) — no LDS staging, no LDS reads in
// the loop.  Blocks with consecutive blockIdx.x work on the same subspace, whose 256 * SIZE * 4 bytes (8 KB at SIZE 8) stay in
// the scalar cache.  JVECTOR_HIP_ENCODE_LDS=1 selects the LDS form above.
template <int SIZE>
__global__ __launch_bounds__(256) void pq_encode_sgpr_kernel(const float *__restrict__ vecs, int64_t count, int D, int M,
                                                             const jv_f2 *__restrict__ cb_paired,
                                                             const int *__restrict__ offsets,
                                                             const float *__restrict__ centroid,
                                                             uint8_t *__restrict__ codes)
{
    const int m = blockIdx.y;
    const jv_f2 *__restrict__ cb2 = cb_paired + (int64_t)m * (kClusters / 2) * SIZE;
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= count) return;
    const int off = offsets[m];
    const float *vp = vecs + n * D + off;
    float v[SIZE];
#pragma unroll
    for (int j = 0; j < SIZE; ++j) {
        float x = vp[j];
        if (centroid) x = x - centroid[off + j];  // VectorUtil.sub(vector, globalCentroid) :441-443
        v[j] = x;
    }
    // closestCentroidIndex (ProductQuantization.java:507-520): argmin with strict '<' (the first minimum wins; NaN never wins).
    // Round 2 kept (minDist, best) per centroid: 3.75 of the 15.25 VALU slots per centroid were compare / select / index.
    // Here the bookkeeping is per BLOCK of 8 centroids: their 8 sums are reduced with v_min3 (NaN operands drop out, like a
    // failed '<'), ONE strict compare decides whether the block holds a new minimum (an earlier block keeps a tie), and only
    // the block number is recorded.  The winning block's 8 sums are recomputed once at the end (the same instructions on the
    // same operands: the same bits) and the first one equal to the minimum is the code: 0.9 bookkeeping slots per centroid + 3 %
    // recomputation instead of 3.75.
    constexpr int BLK = 4;  // pairs per block
    int best_block = -1;
    float minDist = 3.4028234663852886e+38f;  // Float.MAX_VALUE
#pragma unroll 1
    for (int b = 0; b < kClusters / (2 * BLK); ++b) {
        jv_f2 s[BLK];
#pragma unroll
        for (int t = 0; t < BLK; ++t) {
            s[t] = {0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < SIZE; ++j) {
                const jv_f2 vv = {v[j], v[j]};
                const jv_f2 d = vv - cb2[(b * BLK + t) * SIZE + j];
                s[t] += d * d;
            }
        }
        const float m01 = __builtin_fminf(__builtin_fminf(s[0].x, s[0].y), s[1].x);
        const float m23 = __builtin_fminf(__builtin_fminf(s[1].y, s[2].x), s[2].y);
        const float mb = __builtin_fminf(__builtin_fminf(m01, m23), __builtin_fminf(s[3].x, s[3].y));
        const bool lt = mb < minDist;
        minDist = lt ? mb : minDist;
        best_block = lt ? b : best_block;
    }
    int best = 0;
    if (best_block >= 0) {  // (else nothing was < Float.MAX_VALUE: the reference's `best` stays 0)
        const jv_f2 *__restrict__ blk = cb2 + (int64_t)best_block * BLK * SIZE;  // lane-dependent: vector loads, L2-resident
        int found = -1;
#pragma unroll
        for (int t = 0; t < BLK; ++t) {
            jv_f2 st = {0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < SIZE; ++j) {
                const jv_f2 vv = {v[j], v[j]};
                const jv_f2 d = vv - blk[t * SIZE + j];
                st += d * d;
            }
            if (found < 0 && st.x == minDist) found = 2 * t;
            if (found < 0 && st.y == minDist) found = 2 * t + 1;
        }
        best = best_block * 2 * BLK + (found < 0 ? 0 : found);
    }
    codes[n * M + m] = (uint8_t)best;
}

// any sub-vector size (non-uniform splits when D % M != 0, or sizes without a specialisation):
// sub-vector re-read from global/L1 per centroid — correct, slow, only a fallback.
__global__ __launch_bounds__(256) void pq_encode_generic_kernel(const float *__restrict__ vecs, int64_t count, int D,
                                                                int M, const float *__restrict__ codebooks,
                                                                const int64_t *__restrict__ cb_off,
                                                                const int *__restrict__ sizes,
                                                                const int *__restrict__ offsets,
                                                                const float *__restrict__ centroid,
                                                                uint8_t *__restrict__ codes)
{
    extern __shared__ __attribute__((aligned(16))) float cbg[];
    const int m = blockIdx.y;
    const int size = sizes[m], off = offsets[m];
    const float *src = codebooks + cb_off[m];
    for (int j = threadIdx.x; j < kClusters * size; j += 256) cbg[j] = src[j];
    __syncthreads();
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= count) return;
    const float *vp = vecs + n * D + off;
    int best = 0;
    float minDist = 3.4028234663852886e+38f;
    for (int i = 0; i < kClusters; ++i) {
        float s = 0.0f;
        for (int j = 0; j < size; ++j) {
            float x = vp[j];
            if (centroid) x = x - centroid[off + j];
            float d = x - cbg[i * size + j];
            s += d * d;
        }
        if (s < minDist) {
            minDist = s;
            best = i;
        }
    }
    codes[n * M + m] = (uint8_t)best;
}

// sub-vectors too long for the codebook of one subspace to fit LDS (256 * size * 4 B > 128 KB, i.e. size > 128 — e.g. M = 1
// over a 1000-dimensional vector): the same loop reading the centroids from global memory (L2-resident across the block)
__global__ __launch_bounds__(256) void pq_encode_global_kernel(const float *__restrict__ vecs, int64_t count, int D, int M,
                                                               const float *__restrict__ codebooks,
                                                               const int64_t *__restrict__ cb_off,
                                                               const int *__restrict__ sizes,
                                                               const int *__restrict__ offsets,
                                                               const float *__restrict__ centroid,
                                                               uint8_t *__restrict__ codes)
{
    const int m = blockIdx.y;
    const int size = sizes[m], off = offsets[m];
    const float *cbg = codebooks + cb_off[m];
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= count) return;
    const float *vp = vecs + n * D + off;
    int best = 0;
    float minDist = 3.4028234663852886e+38f;
    for (int i = 0; i < kClusters; ++i) {
        float s = 0.0f;
        for (int j = 0; j < size; ++j) {
            float x = vp[j];
            if (centroid) x = x - centroid[off + j];
            float d = x - cbg[(int64_t)i * size + j];
            s += d * d;
        }
        if (s < minDist) {
            minDist = s;
            best = i;
        }
    }
    codes[n * M + m] = (uint8_t)best;
}

int launch_pq_encode(hipStream_t s, const jv_pq *pq, const float *d_vecs, int64_t count, uint8_t *d_codes)
{
    if (count == 0) return JV_OK;
    dim3 grid((unsigned)((count + 255) / 256), pq->M), block(256);
#define JV_ENC(SZ)                                                                                             \
    hipLaunchKernelGGL(pq_encode_kernel<SZ>, grid, block, 0, s, d_vecs, count, pq->D, pq->M, pq->d_codebooks,  \
                       pq->d_cb_offsets, pq->d_offsets, pq->d_centroid, d_codes)
    bool done = false;
#define JV_ENC_S(SZ)                                                                                                       \
    hipLaunchKernelGGL(pq_encode_sgpr_kernel<SZ>, grid, block, 0, s, d_vecs, count, pq->D, pq->M, (const jv_f2 *)pq->d_cb_paired, \
                       pq->d_offsets, pq->d_centroid, d_codes)
    if (pq->uniform && pq->d_cb_paired && !getenv("JVECTOR_HIP_ENCODE_LDS")) {
        done = true;
        switch (pq->max_size) {
        case 2: JV_ENC_S(2); break;
        case 4: JV_ENC_S(4); break;
        case 6: JV_ENC_S(6); break;
        case 8: JV_ENC_S(8); break;
        case 12: JV_ENC_S(12); break;
        case 16: JV_ENC_S(16); break;
        default: done = false;
        }
    }
#undef JV_ENC_S
    if (!done && pq->uniform) {
        done = true;
        switch (pq->max_size) {
        case 1: JV_ENC(1); break;
        case 2: JV_ENC(2); break;
        case 3: JV_ENC(3); break;
        case 4: JV_ENC(4); break;
        case 6: JV_ENC(6); break;
        case 8: JV_ENC(8); break;
        case 12: JV_ENC(12); break;
        case 16: JV_ENC(16); break;
        default: done = false;
        }
    }
#undef JV_ENC
    if (!done && (size_t)kClusters * pq->max_size * sizeof(float) > 128 * 1024) {
        hipLaunchKernelGGL(pq_encode_global_kernel, grid, block, 0, s, d_vecs, count, pq->D, pq->M, pq->d_codebooks, pq->d_cb_offsets,
                           pq->d_sizes, pq->d_offsets, pq->d_centroid, d_codes);
        done = true;
    }
    if (!done) {
        size_t lds = (size_t)kClusters * pq->max_size * sizeof(float);
        if (lds > 64 * 1024) {
            JV_HIP_CHECK(hipFuncSetAttribute((const void *)pq_encode_generic_kernel,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        hipLaunchKernelGGL(pq_encode_generic_kernel, grid, block, lds, s, d_vecs, count, pq->D, pq->M, pq->d_codebooks,
                           pq->d_cb_offsets, pq->d_sizes, pq->d_offsets, pq->d_centroid, d_codes);
    }
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

}  // namespace jv
