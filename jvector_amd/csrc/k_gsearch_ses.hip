// k_gsearch_ses.hip — the session kernels of the device-resident traversal (GraphSearcher OBJECTS, jv_hip_searcher_*): the body of
// k_gsearch.hip's kernels with SES = true (gs_body.h): layer-0 threshold admission, the TwoPhaseTracker stop, expandedCountBaseLayer.
// A translation unit of its own so that the two halves of the traversal's instantiations compile in parallel.
#include "jv_device.h"
#include "jv_internal.h"

#include "gs_wave_hip.h"

#include "gs_body.h"

namespace jv {

// Built for every M the plain kernels are built for (16, 32, 48, 64, 96, 128, 192) + the generic form (CH16 = 0): 2 waves per SIMD, except the lane-per-neighbour
// form at M >= 96 (degrees above 32), whose code words do not fit 256 registers (the plain kernels' OCC = 1 build); the generic form
// holds no code words at all and runs at 4 waves per SIMD.
template <int VSF, int CH16, bool PAIR>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CH16 == 0 ? 4 : ((PAIR || CH16 < 6) ? 2 : 1), CH16 == 0 ? 4 : ((PAIR || CH16 < 6) ? 2 : 1))))
void graph_search_session_kernel(GsParams p)
{
    extern __shared__ __attribute__((aligned(16))) char gs_lds[];
    gs_worker<VSF, CH16, PAIR, false, true>(p, (int)blockIdx.x, gs_lds);
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
    case 0:  // the generic form (any sub-vector geometry): one lane per neighbour
        if (pair) {
            set_error("graph search kernel: the generic form has no pair-lane scoring");
            return JV_ERR_INVALID;
        }
        hipLaunchKernelGGL((graph_search_session_kernel<VSF, 0, false>), grid, block, lds, s, p);
        break;
    case 1: JV_SES(1); break;
    case 2: JV_SES(2); break;
    case 3: JV_SES(3); break;
    case 4: JV_SES(4); break;
    case 6: JV_SES(6); break;
    case 8: JV_SES(8); break;
    case 12: JV_SES(12); break;
    default:
        set_error("graph search kernel: M = %d has no specialised build (16, 32, 48, 64, 96, 128, 192) and the launch did not ask for the generic one", ch * 16);
        return JV_ERR_UNSUPPORTED;
    }
#undef JV_SES
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}


bool graph_search_session_supported(int M) { return M >= 1; }  // (specialised builds for M = 16 ... 192, the generic form otherwise)

int launch_graph_search_session(hipStream_t s, int vsf, const GsParams &p, int workers, size_t lds)
{
    const int ch = p.generic ? 0 : p.M / 16;
    switch (vsf) {
    case VSF_L2: return launch_gs_session<VSF_L2>(s, p, ch, workers, lds);
    case VSF_DOT: return launch_gs_session<VSF_DOT>(s, p, ch, workers, lds);
    default: return launch_gs_session<VSF_COS>(s, p, ch, workers, lds);
    }
}

}  // namespace jv
