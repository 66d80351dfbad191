// k_exact_dense.hip — MFMA tile form of full-resolution scoring (dense Q x N form).  Body and the
// arithmetic contract in ed_body.h.  Roofline: MFMA f32 (157 TF peak; 2·Q·N·D flop) once Q >= ~64, HBM (4·D bytes per
// vector, read once per 128-query tile row that misses L2) below that.  Launch: 4 wavefronts per block sharing a
// 128 x 128 tile staged K-major in 34 KB of LDS; four independent 32x32 accumulators per wave keep the 64-cycle MFMA issue
// slot full and the next chunk's global loads are in flight under them (two-stage pipeline, ed_body.h).
#include <cstdlib>

#include "gs_wave_hip.h"
#include "jv_internal.h"

#include "ed_body.h"

namespace jv {

// Three builds of the same body: the register allocator's own choice (109 + 64 accumulator registers: 2 waves per SIMD), and
// capped at 3 / 4 waves per SIMD (148 registers without spills / 128 with ~8 spilled dwords).  More resident blocks keep the
// MFMA pipe fed across the two barriers of a K chunk; JVECTOR_HIP_ED_WAVES picks (measured default below).
#define ED_KERNEL(NAME, ATTR)                                                                                                   \
    template <int VSF>                                                                                                          \
    __global__ __launch_bounds__(ED_THREADS) ATTR void NAME(EdParams p, int64_t blocks_padded, int64_t n_tiles, int q_tiles)    \
    {                                                                                                                           \
        __shared__ float lds[ED_LDS_FLOATS];                                                                                    \
        int64_t n_tile;                                                                                                         \
        int q_tile;                                                                                                             \
        if (!ed_block_to_tile((int64_t)blockIdx.x, blocks_padded, n_tiles, q_tiles, &n_tile, &q_tile)) return;                  \
        ed_tile<VSF>(p, n_tile * ED_TN, q_tile * ED_TQ, lds);                                                                   \
    }
ED_KERNEL(exact_dense_kernel, )
ED_KERNEL(exact_dense_kernel_w3, __attribute__((amdgpu_waves_per_eu(3, 3))))
ED_KERNEL(exact_dense_kernel_w4, __attribute__((amdgpu_waves_per_eu(4, 4))))

static int ed_waves()
{
    static const int w = [] {
        const char *e = getenv("JVECTOR_HIP_ED_WAVES");
        return e ? atoi(e) : 3;
    }();
    return w;
}

int launch_exact_scan_dense(hipStream_t s, const float *d_vecs, int D, const float *d_q, int Q, int vsf, int64_t first,
                            int64_t count, float *d_out)
{
    if (Q == 0 || count == 0) return JV_OK;
    const EdParams p{d_vecs, d_q, d_out, first, count, D, Q};
    const int64_t n_tiles = (count + ED_TN - 1) / ED_TN;
    const int q_tiles = (Q + ED_TQ - 1) / ED_TQ;
    const int64_t blocks_padded = (n_tiles * q_tiles + 7) / 8 * 8;
    if (blocks_padded > 0x7fffffffLL) {
        set_error("exact_scan_dense: %lld tiles exceed one launch; scan a smaller range", (long long)blocks_padded);
        return JV_ERR_INVALID;
    }
    const dim3 grid((unsigned)blocks_padded), block(ED_THREADS);
    if (vsf < 0 || vsf > 2) {
        set_error("exact_scan_dense: unknown similarity %d", vsf);
        return JV_ERR_INVALID;
    }
#define ED_LAUNCH(K)                                                                                      \
    switch (vsf) {                                                                                        \
    case 0: hipLaunchKernelGGL(K<0>, grid, block, 0, s, p, blocks_padded, n_tiles, q_tiles); break;      \
    case 1: hipLaunchKernelGGL(K<1>, grid, block, 0, s, p, blocks_padded, n_tiles, q_tiles); break;      \
    default: hipLaunchKernelGGL(K<2>, grid, block, 0, s, p, blocks_padded, n_tiles, q_tiles); break;     \
    }
    const int w = ed_waves();
    if (w == 4) { ED_LAUNCH(exact_dense_kernel_w4) }
    else if (w == 3) { ED_LAUNCH(exact_dense_kernel_w3) }
    else { ED_LAUNCH(exact_dense_kernel) }
#undef ED_LAUNCH
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

}  // namespace jv
