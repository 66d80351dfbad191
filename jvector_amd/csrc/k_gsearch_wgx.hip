// k_gsearch_wgx.hip — the WORKGROUP form of the device-resident traversal (body: gx_body.h on top of gs_body.h's helpers): one query
// per workgroup, the query's ADC table [M][256] f32 in LDS, wave 0 = GraphSearcher's loop, the other waves score adjacency rows it
// requests ahead of time.  A translation unit of its own: inside it gs_barrier() is a WAVE-scope sync point (the control wave is one
// wave of a larger workgroup), everywhere else a workgroup barrier.
#include "jv_device.h"
#include "jv_internal.h"

#define GS_WAVE_SCOPE_BARRIER 1
#define GS_UNIFORM_SHFL 1
#include "gs_wave_hip.h"

#include "gx_body.h"

namespace jv {

static_assert(VSF_L2 == 0 && VSF_DOT == 1 && VSF_COS == 2, "gs_body.h hard-codes the kernel vsf numbering");

// up to 512 threads: one control wave + 3 or 7 expanders; one workgroup per CU (the table leaves no room for a second one), i.e.
// at most 2 waves per SIMD — the compiler may use 256 VGPRs
// FULL: the table covers every subspace (the usual case wherever it fits: the scoring loop then holds no table-free code)
template <int VSF, int CH16, bool PROF, bool FULL>
__global__ __launch_bounds__(512) void graph_search_wgx_kernel(GsParams p)
{
    extern __shared__ __attribute__((aligned(16))) char gs_lds[];
    gx_worker<VSF, CH16, PROF, FULL>(p, (int)blockIdx.x, gs_lds);
}

template <int VSF, bool PROF>
static int launch_wgx_ch(hipStream_t s, const GsParams &p, int ch, int workgroups, int threads, size_t lds)
{
    dim3 grid(workgroups), block(threads);
#define JV_WGX(CH)                                                                                                   \
    do {                                                                                                             \
        if (p.wgx_lut_m == p.M) {                                                                                    \
            auto kfn = graph_search_wgx_kernel<VSF, CH, PROF, true>;                                                 \
            JV_HIP_CHECK(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL(kfn, grid, block, lds, s, p);                                                         \
        } else {                                                                                                     \
            auto kfn = graph_search_wgx_kernel<VSF, CH, PROF, false>;                                                \
            JV_HIP_CHECK(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL(kfn, grid, block, lds, s, p);                                                         \
        }                                                                                                            \
    } while (0)
    switch (ch) {
    case 1: JV_WGX(1); break;
    case 2: JV_WGX(2); break;
    case 3: JV_WGX(3); break;
    case 4: JV_WGX(4); break;
    case 6: JV_WGX(6); break;
    case 8: JV_WGX(8); break;
    case 12: JV_WGX(12); break;
    default:
        set_error("graph search kernel (workgroup form): M = %d has no build (16, 32, 48, 64, 96, 128, 192)", ch * 16);
        return JV_ERR_UNSUPPORTED;
    }
#undef JV_WGX
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

// every specialised shape (the table may cover a prefix of the subspaces only, so M = 192 fits too)
bool graph_search_wgx_supported(int M)
{
    const int ch = M / 16;
    return M % 16 == 0 && (ch == 1 || ch == 2 || ch == 3 || ch == 4 || ch == 6 || ch == 8 || ch == 12);
}

size_t graph_search_wgx_lds_bytes(int D, int rerankK, int cand_cap, int evict_cap, int v1_log2, int slots, int kps, int logcap, int M)
{
    return gx_lds_bytes(D, rerankK, cand_cap, evict_cap > 0 ? evict_cap : GS_EVICT_CAP, v1_log2, slots, kps, logcap, M);
}

int launch_graph_search_wgx(hipStream_t s, int vsf, const GsParams &p, int workgroups, int threads)
{
    if (p.Q == 0) return JV_OK;
    if (p.generic || p.pair || p.session || p.prefetch) {
        set_error("graph search kernel (workgroup form): plain searches over the specialised PQ shapes only");
        return JV_ERR_INVALID;
    }
    if (p.cand_cap != GX_HOT) {
        set_error("graph search kernel (workgroup form): the candidates' LDS tier holds %d keys (cand_cap %d)", GX_HOT, p.cand_cap);
        return JV_ERR_INVALID;
    }
    if (p.wgx_lut_m < 16 || p.wgx_lut_m > p.M || p.wgx_lut_m % 16 != 0) {
        set_error("graph search kernel (workgroup form): the table covers %d of %d subspaces (a multiple of 16 expected)", p.wgx_lut_m, p.M);
        return JV_ERR_INVALID;
    }
    if (threads < 128 || threads > 512 || threads % 64 != 0 || p.wgx_slots < 2 || p.wgx_slots > GX_MAX_SLOTS || (p.wgx_kps != 32 && p.wgx_kps != 64)) {
        set_error("graph search kernel (workgroup form): bad launch shape (threads %d, slots %d, keys per slot %d)", threads, p.wgx_slots, p.wgx_kps);
        return JV_ERR_INVALID;
    }
    const size_t lds = gx_lds_bytes(p.D, p.rerankK, p.cand_cap, p.evict_cap > 0 ? p.evict_cap : GS_EVICT_CAP, p.v1_log2, p.wgx_slots, p.wgx_kps, p.wgx_log, p.wgx_lut_m);
    const int ch = p.M / 16;
    if (p.prof) {
        if (vsf != VSF_COS) {
            set_error("graph search kernel (workgroup form): the phase-clock variant is built for cosine only");
            return JV_ERR_UNSUPPORTED;
        }
        return launch_wgx_ch<VSF_COS, true>(s, p, ch, workgroups, threads, lds);
    }
    switch (vsf) {
    case VSF_L2: return launch_wgx_ch<VSF_L2, false>(s, p, ch, workgroups, threads, lds);
    case VSF_DOT: return launch_wgx_ch<VSF_DOT, false>(s, p, ch, workgroups, threads, lds);
    default: return launch_wgx_ch<VSF_COS, false>(s, p, ch, workgroups, threads, lds);
    }
}

}  // namespace jv
