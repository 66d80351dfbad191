// k_gsearch_ses.hip — the session kernels of the device-resident traversal (GraphSearcher OBJECTS, jv_hip_searcher_*): the body of
// k_gsearch.hip's kernels with SES = true (gs_body.h): layer-0 threshold admission, the TwoPhaseTracker stop, expandedCountBaseLayer.
// A translation unit of its own so that the two halves of the traversal's instantiations compile in parallel.
#include "jv_device.h"
#include "jv_internal.h"

#include "gs_wave_hip.h"

#include "gs_body.h"

namespace jv {

// Built for M = 16 and M = 96 (the test shape and the headline shape); other shapes stay on the host searcher.
template <int VSF, int CH16, bool PAIR>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void graph_search_session_kernel(GsParams p)
{
    extern __shared__ __attribute__((aligned(16))) char gs_lds[];
    gs_worker<VSF, CH16, PAIR, false, false, true>(p, (int)blockIdx.x, gs_lds);
}

template <int VSF>
static int launch_gs_session(hipStream_t s, const GsParams &p, int ch, int workers, size_t lds)
{
    dim3 grid(workers), block(64);
    const bool pair = p.pair != 0;
#define JV_SES(CH)                                                                                               \
    do {                                                                                                         \
        if (pair) hipLaunchKernelGGL((graph_search_session_kernel<VSF, CH, true>), grid, block, lds, s, p);      \
        else hipLaunchKernelGGL((graph_search_session_kernel<VSF, CH, false>), grid, block, lds, s, p);          \
    } while (0)
    switch (ch) {
    case 1: JV_SES(1); break;
    case 6: JV_SES(6); break;
    default:
        set_error("graph search kernel: the GraphSearcher-object form is built for M = 16 and M = 96 (M = %d)", ch * 16);
        return JV_ERR_UNSUPPORTED;
    }
#undef JV_SES
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}


bool graph_search_session_supported(int M) { return M == 16 || M == 96; }

int launch_graph_search_session(hipStream_t s, int vsf, const GsParams &p, int workers, size_t lds)
{
    const int ch = p.M / 16;
    switch (vsf) {
    case VSF_L2: return launch_gs_session<VSF_L2>(s, p, ch, workers, lds);
    case VSF_DOT: return launch_gs_session<VSF_DOT>(s, p, ch, workers, lds);
    default: return launch_gs_session<VSF_COS>(s, p, ch, workers, lds);
    }
}

}  // namespace jv
