// k_gsearch.hip — device-resident graph traversal: the whole GraphSearcher loop of a query runs inside one
// 64-lane wavefront (body: gs_body.h, which the CPU tests also compile for a lane emulator).
//
// Why: the host batched searcher (graph_search.cpp) is bound by the HOST — ~1.2 us of heap / hash work per
// expansion per core, 77-91 k QPS on the 16-core box while its scoring kernels occupy the GPU for 18 % of the step
// (this kernel: 0.8-1.0 M QPS on the same 10M index).
// Moving the three queues and the visited set next to the scoring removes the per-round PCIe exchange, the worker
// pool and the dependence on host cores (which N > 1 ranks have to share).  Per expansion a wave reads one adjacency
// row (128 B) and, with FusedPQ, one packed block (maxDegree x M = 3 KB) from HBM; everything else (candidate and
// result arrays, the centred query) lives in LDS, the visited table and the rarely touched spill tier in L2/MALL, and
// the codebook (768 KB) is shared by every wave through L2.  Launch: persistent waves, one per block, several
// blocks per CU (LDS-limited), pulling queries from an atomic counter — queries differ 2-3x in length, and a
// static split would leave CUs idle at the tail.  XCD placement needs no remapping: a worker touches only its own
// scratch plus read-only data shared by all.
#include "jv_device.h"
#include "jv_internal.h"

// one wavefront per block: gs_barrier() only has to order the wave's own LDS accesses (a WAVE-scope sync point: no s_barrier and,
// above all, no s_waitcnt vmcnt(0) behind every fire-and-forget global store), broadcasts go through v_readlane, and the 64-bit
// max / min reductions of the queue scans through DPP steps instead of ds_bpermute round trips (gs_wave_hip.h)
#define GS_WAVE_SCOPE_BARRIER 1
#define GS_UNIFORM_SHFL 1
#include "gs_wave_hip.h"

#include "gs_body.h"

namespace jv {

static_assert(VSF_L2 == 0 && VSF_DOT == 1 && VSF_COS == 2, "gs_body.h hard-codes the kernel vsf numbering");

// OCC = waves per SIMD the register allocation is held to: 2 (225 VGPRs: the unrolled scoring keeps ~100 codebook
// loads in flight per wave) or 4 (128 VGPRs, no spills: twice the resident queries per CU to hide the
// pop -> load -> probe -> score -> push dependency chain).  Measured on MI355X (1M x 768, 16 384 queries per launch):
// 2 waves/SIMD with pair-lane scoring 19.0 ms, 4 waves/SIMD 25.1 ms, 2 waves/SIMD without pair lanes 33.7 ms, a 3 waves/SIMD
// build 28.3 ms: the kernel is bound by the rate at which a CU's texture path takes 16-byte gather requests
// (tools/gather_bench.hip), and more gathers in flight per wave beat more waves.  2 is the default; 4 stays selectable.
template <int VSF, int CH16, int OCC, bool PAIR>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void graph_search_kernel(GsParams p)
{
    extern __shared__ __attribute__((aligned(16))) char gs_lds[];
    gs_worker<VSF, CH16, PAIR>(p, (int)blockIdx.x, gs_lds);
}

// PAIRC (gs_body.h "PAIRC", GsParams::pair == 2): rows of 33 ... 64 neighbours whose codes are read by ordinal — the builder's
// working rows.  One lane per neighbour probes the visited set, the fresh ones are scored two lanes each (M > 96: four lanes each).
template <int VSF, int CH16>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void graph_search_pairc_kernel(GsParams p)
{
    extern __shared__ __attribute__((aligned(16))) char gs_lds[];
    gs_worker<VSF, CH16, false, false, false, true>(p, (int)blockIdx.x, gs_lds);
}

// developer aid (JVECTOR_HIP_GS_PROF=1): the benched instance with per-phase shader-clock counters (GsParams::prof)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void graph_search_prof_kernel(GsParams p)
{
    extern __shared__ __attribute__((aligned(16))) char gs_lds[];
    gs_worker<VSF_COS, 6, true, true>(p, (int)blockIdx.x, gs_lds);
}

template <int VSF, int OCC>
static int launch_gs_ch(hipStream_t s, const GsParams &p, int ch, int workers, size_t lds)
{
    dim3 grid(workers), block(64);
    const bool pair = p.pair != 0;
    if (p.pair == 2) {   // the compacted pair form
        if constexpr (OCC == 2) {
            switch (ch) {
            case 1: hipLaunchKernelGGL((graph_search_pairc_kernel<VSF, 1>), grid, block, lds, s, p); break;
            case 2: hipLaunchKernelGGL((graph_search_pairc_kernel<VSF, 2>), grid, block, lds, s, p); break;
            case 3: hipLaunchKernelGGL((graph_search_pairc_kernel<VSF, 3>), grid, block, lds, s, p); break;
            case 4: hipLaunchKernelGGL((graph_search_pairc_kernel<VSF, 4>), grid, block, lds, s, p); break;
            case 6: hipLaunchKernelGGL((graph_search_pairc_kernel<VSF, 6>), grid, block, lds, s, p); break;
            case 8: hipLaunchKernelGGL((graph_search_pairc_kernel<VSF, 8>), grid, block, lds, s, p); break;     // (four lanes per neighbour)
            case 12: hipLaunchKernelGGL((graph_search_pairc_kernel<VSF, 12>), grid, block, lds, s, p); break;
            default:
                set_error("graph search kernel: the compacted pair form is built for M = 16 ... 192 (M = %d)", ch * 16);
                return JV_ERR_UNSUPPORTED;
            }
            JV_HIP_CHECK(hipGetLastError());
            return JV_OK;
        } else {
            set_error("graph search kernel: pair-lane scoring is only built for the 2-waves/SIMD variant");
            return JV_ERR_INVALID;
        }
    }
    if (pair && OCC != 2) {
        set_error("graph search kernel: pair-lane scoring is only built for the 2-waves/SIMD variant");
        return JV_ERR_INVALID;
    }
#define JV_GS(CH)                                                                                           \
    do {                                                                                                    \
        if constexpr (OCC == 2) {                                                                           \
            if (pair) hipLaunchKernelGGL((graph_search_kernel<VSF, CH, OCC, true>), grid, block, lds, s, p); \
            else hipLaunchKernelGGL((graph_search_kernel<VSF, CH, OCC, false>), grid, block, lds, s, p);     \
        } else {                                                                                            \
            hipLaunchKernelGGL((graph_search_kernel<VSF, CH, OCC, false>), grid, block, lds, s, p);          \
        }                                                                                                   \
    } while (0)
    switch (ch) {
    case 0:  // the generic form (any sub-vector geometry): one lane per neighbour
        if (pair) {
            set_error("graph search kernel: the generic form has no pair-lane scoring");
            return JV_ERR_INVALID;
        }
        hipLaunchKernelGGL((graph_search_kernel<VSF, 0, OCC, false>), grid, block, lds, s, p);
        break;
    case 1: JV_GS(1); break;
    case 2: JV_GS(2); break;
    case 3: JV_GS(3); break;
    case 4: JV_GS(4); break;
    case 6: JV_GS(6); break;
    case 8: JV_GS(8); break;
    case 12: JV_GS(12); break;
    default:
        set_error("graph search kernel: M = %d has no specialised build (16, 32, 48, 64, 96, 128, 192) and the launch did not ask for the generic one", ch * 16);
        return JV_ERR_UNSUPPORTED;
    }
#undef JV_GS
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

// The specialised kernels: uniform 8-dim sub-vectors, M one of 16 ... 192, 16-byte aligned code rows.
bool graph_search_device_specialised(const jv_pq *pq, const jv_codes *codes, const jv_fused *fused)
{
    const int ch = pq->M / 16;
    return pq->uniform && pq->max_size == 8 && pq->k == kClusters && pq->M % 16 == 0 &&
           (ch == 1 || ch == 2 || ch == 3 || ch == 4 || ch == 6 || ch == 8 || ch == 12) && pq->D == 8 * pq->M &&
           (reinterpret_cast<uintptr_t>(codes->d_codes) & 15) == 0 &&
           (!fused || (reinterpret_cast<uintptr_t>(fused->d_blocks) & 15) == 0);
}

// Every 256-cluster quantizer has a device traversal: the specialised kernels where they apply, the generic form otherwise.
bool graph_search_device_supported(const jv_pq *pq, const jv_codes *codes, const jv_fused *fused, int max_degree, int n_levels)
{
    (void)codes;
    (void)fused;
    return pq->k == kClusters && pq->M >= 1 && max_degree <= kMaxGraphDegree && n_levels <= GS_MAX_LEVELS;
}

size_t graph_search_lds_bytes(int D, int rerankK, int cand_cap, int pair_M, int evict_cap, int v1_log2)
{
    return gs_lds_bytes(D, rerankK, cand_cap, pair_M, evict_cap > 0 ? evict_cap : GS_EVICT_CAP, v1_log2);
}

int launch_graph_search(hipStream_t s, int vsf, const GsParams &p, int workers, int occupancy)
{
    if (p.Q == 0) return JV_OK;
    const size_t lds = gs_lds_bytes(p.D, p.rerankK, p.cand_cap, p.pair ? p.M : 0, p.evict_cap > 0 ? p.evict_cap : GS_EVICT_CAP, p.v1_log2);
    const int ch = p.generic ? 0 : p.M / 16;
    if (p.generic && (p.prof || p.pair)) {
        set_error("graph search kernel: the generic form has no phase-clock / pair-lane variant");
        return JV_ERR_INVALID;
    }
    if (p.session) {
        if (p.prof) {
            set_error("graph search kernel: the GraphSearcher-object form has no phase-clock variant");
            return JV_ERR_INVALID;
        }
        return launch_graph_search_session(s, vsf, p, workers, lds + gs_session_lds_bytes());
    }
    if (p.prof) {
        if (!(vsf == VSF_COS && ch == 6 && p.pair && occupancy < 4)) {
            set_error("graph search kernel: the profiling variant is built for cosine, M = 96, pair-lane scoring only");
            return JV_ERR_UNSUPPORTED;
        }
        hipLaunchKernelGGL(graph_search_prof_kernel, dim3(workers), dim3(64), lds, s, p);
        JV_HIP_CHECK(hipGetLastError());
        return JV_OK;
    }
    if (occupancy >= 4) {
        switch (vsf) {
        case VSF_L2: return launch_gs_ch<VSF_L2, 4>(s, p, ch, workers, lds);
        case VSF_DOT: return launch_gs_ch<VSF_DOT, 4>(s, p, ch, workers, lds);
        default: return launch_gs_ch<VSF_COS, 4>(s, p, ch, workers, lds);
        }
    }
    switch (vsf) {
    case VSF_L2: return launch_gs_ch<VSF_L2, 2>(s, p, ch, workers, lds);
    case VSF_DOT: return launch_gs_ch<VSF_DOT, 2>(s, p, ch, workers, lds);
    default: return launch_gs_ch<VSF_COS, 2>(s, p, ch, workers, lds);
    }
}

}  // namespace jv
