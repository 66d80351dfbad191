// k_builder.hip — GPU side of batched graph construction (builder.cpp): the per-item bodies of bl_body.h, one item per thread,
// and the one device-wide primitive the backlink step needs — a radix sort of the batch's back edges by (target, edge index),
// which makes every target's new edges consecutive and keeps their order deterministic (rocPRIM's radix sort through hipCUB:
// library plumbing, like hipMemcpy; every scoring kernel of the build is the engine's own).
#include <hipcub/hipcub.hpp>

#include "jv_device.h"
#include "jv_internal.h"

#define BL_FN __device__ __forceinline__
#define BL_ATOMIC_INC(p) atomicAdd((p), 1u)
#include "bl_body.h"

namespace jv {

namespace {
template <typename P, void (*BODY)(const P &, long long)>
__global__ __launch_bounds__(256) void bl_items_kernel(P p, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) BODY(p, i);
}
template <typename P, void (*BODY)(const P &, long long)>
int run_items(hipStream_t s, const P &p, long long n)
{
    if (n <= 0) return JV_OK;
    hipLaunchKernelGGL((bl_items_kernel<P, BODY>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}
__global__ __launch_bounds__(256) void bl_count_kernel(const int32_t *cand, int C, int32_t *count, long long B)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < B) bl_count_valid(cand, C, count, i);
}
__global__ __launch_bounds__(256) void bl_copy_rows_kernel(const int32_t *nbrs, int R, const int32_t *tgt, long long cells, int32_t *out)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < cells) out[i] = nbrs[(long long)tgt[i / R] * R + (i % R)];
}
__global__ __launch_bounds__(256) void bl_strided_copy_kernel(const int32_t *src, int R, int Rf, long long cells, int32_t *dst)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < cells) dst[i] = src[(i / Rf) * R + (i % Rf)];
}
}  // namespace

int launch_bl_apply_selection(hipStream_t s, const BlApplyParams &p)
{
    JV_TRY((run_items<BlApplyParams, bl_apply_selection>(s, p, (long long)p.B * p.Rf)));
    return run_items<BlApplyParams, bl_pack_row>(s, p, p.B);
}
int launch_bl_improve_list(hipStream_t s, const BlImproveParams &p) { return run_items<BlImproveParams, bl_improve_list>(s, p, p.B); }
int launch_bl_row_edges(hipStream_t s, const BlRowEdgesParams &p) { return run_items<BlRowEdgesParams, bl_row_edges>(s, p, (long long)p.B * p.Rf); }
int launch_bl_backlink_merge(hipStream_t s, const BlMergeParams &p) { return run_items<BlMergeParams, bl_backlink_merge>(s, p, p.E); }
int launch_bl_rank_sort(hipStream_t s, const BlSortParams &p) { return run_items<BlSortParams, bl_rank_sort>(s, p, (long long)p.P * p.L); }
int launch_bl_rewrite_rows(hipStream_t s, const BlRowsParams &p) { return run_items<BlRowsParams, bl_rewrite_row>(s, p, p.P); }
int launch_bl_list_over_degree(hipStream_t s, const BlOverParams &p) { return run_items<BlOverParams, bl_list_over_degree>(s, p, p.N); }

// reference order (bl_body.h "REFERENCE ORDER")
int launch_bl_ro_apply_selection(hipStream_t s, const BlRoApplyParams &p) { return run_items<BlRoApplyParams, bl_ro_apply_selection>(s, p, p.B); }
int launch_bl_ro_backlink_merge(hipStream_t s, const BlRoMergeParams &p) { return run_items<BlRoMergeParams, bl_ro_backlink_merge>(s, p, p.E); }
int launch_bl_ro_rewrite_rows(hipStream_t s, const BlRoRowsParams &p) { return run_items<BlRoRowsParams, bl_ro_rewrite_row>(s, p, p.P); }
int launch_bl_ro_copy_rows(hipStream_t s, const BlRoCopyParams &p) { return run_items<BlRoCopyParams, bl_ro_copy_row>(s, p, p.P); }
int launch_bl_sel_ids(hipStream_t s, const BlSelIdsParams &p) { return run_items<BlSelIdsParams, bl_sel_ids>(s, p, (long long)p.B * p.Rf); }
int launch_bl_ro_apply_sorted(hipStream_t s, const BlRoApplySortedParams &p) { return run_items<BlRoApplySortedParams, bl_ro_apply_sorted>(s, p, p.B); }
int launch_bl_ro_improve_list(hipStream_t s, const BlRoImproveParams &p) { return run_items<BlRoImproveParams, bl_ro_improve_list>(s, p, p.B); }
int launch_bl_ro_row_edges(hipStream_t s, const BlRoRowEdgesParams &p) { return run_items<BlRoRowEdgesParams, bl_ro_row_edges>(s, p, (long long)p.B * p.Rf); }

int launch_bl_count_valid(hipStream_t s, const int32_t *cand, int C, int32_t *count, long long B)
{
    if (B <= 0) return JV_OK;
    hipLaunchKernelGGL(bl_count_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, cand, C, count, B);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}
int launch_bl_copy_rows(hipStream_t s, const int32_t *nbrs, int R, const int32_t *tgt, long long P, int32_t *out)
{
    const long long cells = P * R;
    if (cells <= 0) return JV_OK;
    hipLaunchKernelGGL(bl_copy_rows_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, s, nbrs, R, tgt, cells, out);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}
int launch_bl_strided_copy(hipStream_t s, const int32_t *src, int R, int Rf, long long N, int32_t *dst)
{
    const long long cells = N * Rf;
    if (cells <= 0) return JV_OK;
    hipLaunchKernelGGL(bl_strided_copy_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, s, src, R, Rf, cells, dst);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

// sort (key, value) pairs by key ascending; temp == nullptr: *temp_bytes = what the sort needs
int launch_bl_sort_edges(hipStream_t s, void *temp, size_t *temp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out,
                         const int32_t *vals_in, int32_t *vals_out, long long n, int end_bit)
{
    if (n <= 0) {
        if (!temp) *temp_bytes = 0;
        return JV_OK;
    }
    JV_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(temp, *temp_bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0, end_bit, s));
    return JV_OK;
}

}  // namespace jv
