// k_sharded.hip — the three elementwise helpers of the sharded search (sharded.cpp): all HBM-trivial
// (Q x rerankK x 4..8 bytes per call), kept as kernels only so the whole exchange stays on the stream.
#include "jv_device.h"
#include "jv_internal.h"

namespace jv {

// gathered partial lists [P][Q][k] (P = ranks x local shards, the all-gather's natural layout) -> one row per query
// [Q][P * k], which is what the NodeQueue-order top-k merge consumes
__global__ void shard_interleave_kernel(const int32_t *__restrict__ ids, const float *__restrict__ sc, int P, int Q, int k,
                                        int32_t *__restrict__ out_ids, float *__restrict__ out_sc)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)P * Q * k;
    if (i >= total) return;
    const int j = (int)(i % k);
    const int q = (int)((i / k) % Q);
    const int p = (int)(i / ((int64_t)k * Q));
    const int64_t dst = ((int64_t)q * P + p) * k + j;
    out_ids[dst] = ids[i];
    out_sc[dst] = sc[i];
}

// global id -> ordinal inside the shard that owns [base, base + count), -1 for everybody else's candidates
__global__ void shard_localize_kernel(const int32_t *__restrict__ gids, int64_t n, int64_t base, int64_t count,
                                      int32_t *__restrict__ local)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t g = gids[i];
    local[i] = (g >= base && g < base + count) ? (int32_t)(g - base) : -1;
}

// every piece p scored the candidates it owns (others: -inf).  out[i] = the OWNER's score of candidate i — a selection, not a
// MAX reduction, so NaN (zero vectors under cosine) and -inf scores arrive exactly as a single index would report them.
// ranges: P x {base, count} int64.
__global__ void shard_select_kernel(const int32_t *__restrict__ gids, const float *__restrict__ exact, const long long *__restrict__ ranges,
                                    int P, int64_t n, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long g = gids[i];
    float v = -INFINITY;
    if (g >= 0)
        for (int p = 0; p < P; ++p)
            if (g >= ranges[2 * p] && g < ranges[2 * p] + ranges[2 * p + 1]) {
                v = exact[(int64_t)p * n + i];
                break;
            }
    out[i] = v;
}

}  // namespace jv

namespace jv {
// caller-produced partial lists (jv_hip_sharded_merge_rerank): an id outside the shard that is said to own its list can have no owner in
// the exchange — it becomes (-1, -inf) before the merge
__global__ void shard_sanitize_kernel(int32_t *__restrict__ ids, float *__restrict__ sc, int64_t n, int64_t lo, int64_t hi)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t g = ids[i];
    if (g < lo || g >= hi) {
        ids[i] = -1;
        sc[i] = -INFINITY;
    }
}
int launch_shard_sanitize(hipStream_t s, int32_t *d_ids, float *d_sc, int64_t n, int64_t lo, int64_t hi)
{
    if (n == 0) return JV_OK;
    hipLaunchKernelGGL(shard_sanitize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_ids, d_sc, n, lo, hi);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

int launch_shard_interleave(hipStream_t s, const int32_t *d_ids, const float *d_sc, int P, int Q, int k, int32_t *d_out_ids,
                            float *d_out_sc)
{
    const int64_t total = (int64_t)P * Q * k;
    if (total == 0) return JV_OK;
    hipLaunchKernelGGL(shard_interleave_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d_ids, d_sc, P, Q, k, d_out_ids,
                       d_out_sc);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

int launch_shard_localize(hipStream_t s, const int32_t *d_gids, int64_t n, int64_t base, int64_t count, int32_t *d_local)
{
    if (n == 0) return JV_OK;
    hipLaunchKernelGGL(shard_localize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_gids, n, base, count, d_local);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

int launch_shard_select(hipStream_t s, const int32_t *d_gids, const float *d_exact, const long long *d_ranges, int P, int64_t n,
                        float *d_out)
{
    if (n == 0) return JV_OK;
    hipLaunchKernelGGL(shard_select_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_gids, d_exact, d_ranges, P, n, d_out);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

}  // namespace jv
