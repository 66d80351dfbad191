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
    unsigned long long work2[2] = {0, 0};   // (wave-uniform: tests and (candidate, selected slot) pairs of this block's nodes)
    for (int node = (int)blockIdx.x; node < p.P; node += (int)gridDim.x) rd_node<false>(p, node, rd_lds, work2);
    if (p.counts && threadIdx.x == 0) {
        atomicAdd(p.counts, work2[0]);
        atomicAdd(p.counts + 1, work2[1]);
    }
}

#ifdef JV_EXPERIMENTAL   // the measured-and-switched-off forms (make EXPERIMENTAL=1)
// table-free (RdParams::codebooks; uniform 8-dimensional sub-vectors): the entries recomputed from the L2-resident codebook
// (measured slower than the look-ups, build_score.cpp retain_diverse_table_free: an option, off by default)
__global__ __launch_bounds__(64) void retain_diverse_tf_kernel(RdParams p)
{
    extern __shared__ __attribute__((aligned(16))) char rd_lds[];
    for (int node = (int)blockIdx.x; node < p.P; node += (int)gridDim.x) rd_node<true>(p, node, rd_lds);
}

// the square form of the table (RdParams::sq; rd_square = 1: an experiment)
__global__ __launch_bounds__(64) void retain_diverse_sq_kernel(RdParams p)
{
    extern __shared__ __attribute__((aligned(16))) char rd_lds[];
    for (int node = (int)blockIdx.x; node < p.P; node += (int)gridDim.x) rd_node<false, false, true>(p, node, rd_lds);
}
__global__ __launch_bounds__(64) void retain_diverse_sq_prof_kernel(RdParams p)
{
    extern __shared__ __attribute__((aligned(16))) char rd_lds[];
    for (int node = (int)blockIdx.x; node < p.P; node += (int)gridDim.x) rd_node<false, true, true>(p, node, rd_lds);
}

#endif
// developer aid (rd_prof = 1): the table kernel with per-phase shader-clock counters (rd_body.h RD_PHASE), printed after every launch
__global__ __launch_bounds__(64) void retain_diverse_prof_kernel(RdParams p)
{
    extern __shared__ __attribute__((aligned(16))) char rd_lds[];
    for (int node = (int)blockIdx.x; node < p.P; node += (int)gridDim.x) rd_node<false, true>(p, node, rd_lds);
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
    const bool sqf = !tf && p.sq != nullptr;
#ifdef JV_EXPERIMENTAL
    const void *kfn = tf ? (const void *)retain_diverse_tf_kernel : (sqf ? (const void *)retain_diverse_sq_kernel : (const void *)retain_diverse_kernel);
#else
    if (tf || sqf) {
        set_error("retain_diverse: the table-free / square-table forms are experimental variants (build with make EXPERIMENTAL=1)");
        return JV_ERR_UNSUPPORTED;
    }
    const void *kfn = (const void *)retain_diverse_kernel;
#endif
    if (lds > 48 * 1024) JV_HIP_CHECK(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(16, (160 * 1024) / (lds + 256)));
    const int blocks = std::min(p.P, ctx->num_cus * per_cu);
    if (!tf && ctx_opt(ctx, "rd_prof", 0) != 0) {
        static unsigned long long *d_prof = nullptr;   // (one small buffer for the process: a developer aid)
        if (!d_prof) JV_HIP_CHECK(hipMalloc((void **)&d_prof, sizeof(unsigned long long) * 16));
        JV_HIP_CHECK(hipMemsetAsync(d_prof, 0, sizeof(unsigned long long) * 16, s));
        RdParams pp = p;
        pp.prof = d_prof;
#ifdef JV_EXPERIMENTAL
        const void *pk = p.sq ? (const void *)retain_diverse_sq_prof_kernel : (const void *)retain_diverse_prof_kernel;
        if (lds > 48 * 1024) JV_HIP_CHECK(hipFuncSetAttribute(pk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (p.sq) hipLaunchKernelGGL(retain_diverse_sq_prof_kernel, dim3(blocks), dim3(64), lds, s, pp);
        else hipLaunchKernelGGL(retain_diverse_prof_kernel, dim3(blocks), dim3(64), lds, s, pp);
#else
        if (lds > 48 * 1024) JV_HIP_CHECK(hipFuncSetAttribute((const void *)retain_diverse_prof_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(retain_diverse_prof_kernel, dim3(blocks), dim3(64), lds, s, pp);
#endif
        JV_HIP_CHECK(hipGetLastError());
        unsigned long long h[16];
        JV_HIP_CHECK(hipMemcpyAsync(h, d_prof, sizeof(h), hipMemcpyDeviceToHost, s));
        JV_HIP_CHECK(hipStreamSynchronize(s));
        const double nodes = (double)std::max<unsigned long long>(1, h[10]), tests = (double)std::max<unsigned long long>(1, h[8]);
        fprintf(stderr,
                "[jv rd prof%s] P=%d C=%d M=%d blocks=%d (x%d/CU): clocks per node: stage %.0f  self+init %.0f  prefix %.0f  output %.0f | per test: sums %.0f  decision %.0f  take %.0f  "
                "loop %.0f | tests/node %.1f  slots/test %.1f  candidates/node %.1f  split tests %.2f\n",
                p.sq ? " square" : "", p.P, p.C, p.M, blocks, per_cu, h[0] / nodes, h[1] / nodes, h[2] / nodes, h[7] / nodes, h[3] / tests, h[4] / tests, h[5] / tests, h[6] / tests, tests / nodes,
                h[9] / tests, h[11] / nodes, (double)(h[12] % 1000000) / tests);
        return JV_OK;
    }
#ifdef JV_EXPERIMENTAL
    if (tf) hipLaunchKernelGGL(retain_diverse_tf_kernel, dim3(blocks), dim3(64), lds, s, p);
    else if (sqf) hipLaunchKernelGGL(retain_diverse_sq_kernel, dim3(blocks), dim3(64), lds, s, p);
    else
#endif
    hipLaunchKernelGGL(retain_diverse_kernel, dim3(blocks), dim3(64), lds, s, p);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

}  // namespace jv
