// k_build_score.hip — build-time scoring kernels.  Bodies in bs_body.h (shared with the CPU tests);
// here: flat launches, one thread per output element.  All four are small gather-bound kernels: the pair table
// (12.6 MB at M = 96) and the codebook stay L2/MALL resident, code rows are 96 B gathers.
#include "jv_device.h"
#include "jv_internal.h"

#define BS_FN __device__ __forceinline__
__device__ __forceinline__ double bs_sqrt(double x) { return sqrt(x); }
#include "bs_body.h"

namespace jv {

static BsPq bs_pq_of(const jv_pq *pq)
{
    return BsPq{pq->d_codebooks, pq->d_cb_offsets, pq->d_sizes, pq->d_offsets, pq->d_centroid, pq->D, pq->M, pq->k};
}

__global__ __launch_bounds__(256) void pair_table_kernel(BsPq pq, int vsf, float *out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < (int64_t)pq.M * pq.k) bs_pair_table_row(pq, vsf, t, out);
}

__global__ __launch_bounds__(256) void pair_scores_kernel(const float *tri, int vsf, int M, int k, const uint8_t *codes, int64_t n,
                                                          const int32_t *node1, const int32_t *node2, int B, int64_t total, float *out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < total) bs_pair_score(tri, vsf, M, k, codes, n, node1, node2, B, t, out);
}

__global__ __launch_bounds__(256) void decode_kernel(BsPq pq, const uint8_t *codes, int64_t n, const int32_t *ordinals, int64_t first,
                                                     int64_t total, float *out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < total) bs_decode(pq, codes, n, ordinals, first, t, out);
}

__global__ __launch_bounds__(256) void query_norm_kernel(const float *cq, int D, int Q, float *out)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < Q) bs_query_norm(cq, D, q, out);
}

__global__ __launch_bounds__(256) void direct_scores_kernel(BsPq pq, int vsf, const uint8_t *codes, int64_t n, const float *cq,
                                                            const float *qnorm, const int32_t *ordinals, int B, int64_t total, float *out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < total) bs_direct_score(pq, vsf, codes, n, cq, qnorm, ordinals, B, t, out);
}

__global__ __launch_bounds__(256) void fused_gather_kernel(const uint8_t *codes, int64_t n_codes, const int32_t *neighbors, int maxDegree,
                                                           int M, int chunk, int64_t total, uint8_t *blocks)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < total) bs_fused_gather(codes, n_codes, neighbors, maxDegree, M, chunk, t, blocks);
}

static dim3 flat_grid(int64_t total) { return dim3((unsigned)((total + 255) / 256)); }

int launch_pair_table(hipStream_t s, const jv_pq *pq, int vsf, float *d_out)
{
    const int64_t total = (int64_t)pq->M * pq->k;
    hipLaunchKernelGGL(pair_table_kernel, flat_grid(total), dim3(256), 0, s, bs_pq_of(pq), vsf, d_out);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

// the triangular table expanded to [M][k][k]: entry (m, i, j) = the triangle's (m, min, max) — a copy, bit for bit
__global__ __launch_bounds__(256) void pair_table_square_kernel(const float *tri, int M, int k, float *sq)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)M * k * k) return;
    const int j = (int)(t % k), i = (int)((t / k) % k), m = (int)(t / ((int64_t)k * k));
    const int r = i < j ? i : j, c = i < j ? j : i;
    sq[t] = tri[(int64_t)m * ((int64_t)k * (k + 1) / 2) + bs_tri_row(r, k) + (c - r)];
}

int launch_pair_table_square(hipStream_t s, const float *d_tri, int M, int k, float *d_sq)
{
    hipLaunchKernelGGL(pair_table_square_kernel, flat_grid((int64_t)M * k * k), dim3(256), 0, s, d_tri, M, k, d_sq);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

int launch_pair_scores(hipStream_t s, const float *d_tri, int vsf, const jv_codes *codes, const int32_t *d_node1, int P,
                       const int32_t *d_node2, int B, float *d_out)
{
    const int64_t total = (int64_t)P * B;
    if (total == 0) return JV_OK;
    hipLaunchKernelGGL(pair_scores_kernel, flat_grid(total), dim3(256), 0, s, d_tri, vsf, codes->M, codes->pq->k, codes->d_codes,
                       codes->count, d_node1, d_node2, B, total, d_out);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

// HBM-bound copy: count * maxDegree * M bytes written (3 KB per node at C3), the reads are 96-byte gathers
int launch_fused_gather(hipStream_t s, const jv_codes *codes, const int32_t *d_neighbors, int maxDegree, int64_t count, uint8_t *d_blocks)
{
    const int M = codes->M;
    const bool wide = M % 16 == 0 && (reinterpret_cast<uintptr_t>(codes->d_codes) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_blocks) & 15) == 0;
    const int chunk = wide ? 16 : 1;
    const int64_t total = count * maxDegree * (M / chunk);
    if (total == 0) return JV_OK;
    hipLaunchKernelGGL(fused_gather_kernel, flat_grid(total), dim3(256), 0, s, codes->d_codes, codes->count, d_neighbors, maxDegree, M, chunk,
                       total, d_blocks);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

int launch_pq_decode(hipStream_t s, const jv_codes *codes, const int32_t *d_ordinals, int64_t first, int64_t count, float *d_out)
{
    const int64_t total = count * codes->pq->D;
    if (total == 0) return JV_OK;
    hipLaunchKernelGGL(decode_kernel, flat_grid(total), dim3(256), 0, s, bs_pq_of(codes->pq), codes->d_codes, codes->count, d_ordinals,
                       first, total, d_out);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

int launch_direct_scores(hipStream_t s, const jv_codes *codes, int vsf, const float *d_cq, int Q, const int32_t *d_ordinals, int B,
                         float *d_qnorm, float *d_out)
{
    const int64_t total = (int64_t)Q * B;
    if (total == 0) return JV_OK;
    if (vsf == VSF_COS) hipLaunchKernelGGL(query_norm_kernel, flat_grid(Q), dim3(256), 0, s, d_cq, codes->pq->D, Q, d_qnorm);
    hipLaunchKernelGGL(direct_scores_kernel, flat_grid(total), dim3(256), 0, s, bs_pq_of(codes->pq), vsf, codes->d_codes, codes->count,
                       d_cq, d_qnorm, d_ordinals, B, total, d_out);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

}  // namespace jv
