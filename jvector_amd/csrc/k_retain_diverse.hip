// k_retain_diverse.hip — batched VamanaDiversityProvider.retainDiverse (robust prune of Vamana construction) with the PQ
// diversity score: one wavefront per node, body in rd_body.h (shared with the CPU lane emulator).  BASELINE config 5's
// "GPU-batched neighbor scoring": the candidate x selected score blocks of many concurrent inserts in one launch.
// Bound: L2/MALL gather latency of the 4-byte pair-table entries (M per (candidate, selected) pair); the table (12.6 MB at
// M = 96, 25 MB at M = 192) is shared by every wave.  Launch: min(P, 16 x CUs) persistent blocks striding over the nodes.
#include "jv_device.h"
#include "jv_internal.h"

#include "gs_wave_hip.h"

#include "rd_body.h"

namespace jv {

__global__ __launch_bounds__(64) void retain_diverse_kernel(RdParams p)
{
    extern __shared__ __attribute__((aligned(16))) char rd_lds[];
    for (int node = (int)blockIdx.x; node < p.P; node += (int)gridDim.x) rd_node<false>(p, node, rd_lds);
}

// table-free (RdParams::codebooks; uniform 8-dimensional sub-vectors): the entries recomputed from the L2-resident codebook
// (measured slower than the look-ups, build_score.cpp retain_diverse_table_free: an option, off by default)
__global__ __launch_bounds__(64) void retain_diverse_tf_kernel(RdParams p)
{
    extern __shared__ __attribute__((aligned(16))) char rd_lds[];
    for (int node = (int)blockIdx.x; node < p.P; node += (int)gridDim.x) rd_node<true>(p, node, rd_lds);
}

size_t retain_diverse_lds_bytes(int C, int M) { return rd_lds_bytes(C, M); }

int launch_retain_diverse(hipStream_t s, const jv_ctx *ctx, const RdParams &p)
{
    if (p.P == 0) return JV_OK;
    const bool tf = p.codebooks != nullptr;
    const size_t lds = rd_lds_bytes(p.C, p.M, tf);
    if (lds > ctx->lds_per_block) {
        set_error("retain_diverse: %d candidates x %d code bytes need %zu bytes of LDS (limit %zu); prune in smaller candidate lists", p.C,
                  p.M, lds, ctx->lds_per_block);
        return JV_ERR_UNSUPPORTED;
    }
    const void *kfn = tf ? (const void *)retain_diverse_tf_kernel : (const void *)retain_diverse_kernel;
    if (lds > 48 * 1024) JV_HIP_CHECK(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(16, (160 * 1024) / (lds + 256)));
    const int blocks = std::min(p.P, ctx->num_cus * per_cu);
    if (tf) hipLaunchKernelGGL(retain_diverse_tf_kernel, dim3(blocks), dim3(64), lds, s, p);
    else hipLaunchKernelGGL(retain_diverse_kernel, dim3(blocks), dim3(64), lds, s, p);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

}  // namespace jv
