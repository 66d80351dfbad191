// ed_emu.cpp — TEST HARNESS: runs the body of exact_dense_kernel (jvector_amd/csrc/ed_body.h, unchanged) on the 64-lane
// emulator, one emulated wavefront per tile, with the launch geometry of k_exact_dense.hip (XCD-aware block -> tile map
// included).  The emulator's MFMA is the instruction's documented semantics (hip_emu.h), so a passing comparison with the
// k-ordered fmaf-chain specification means the tile indexing, staging, padding and C/D unpacking are right; what the
// hardware run still has to confirm is only that the instruction behaves as documented.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "hip_emu.h"

#define GS_FN inline
static inline int gs_lane() { return emu::lane(); }
static inline void gs_barrier() { emu::barrier(); }
static inline int gs_tid() { return emu::lane(); }
static inline void gs_block_barrier() { emu::barrier(); }
static inline void gs_sched_fence() {}
static inline double gs_sqrt(double x) { return std::sqrt(x); }
static inline float gs_fmaf(float a, float b, float c) { return fmaf(a, b, c); }
typedef emu::f32x16 gs_f32x16;
static inline gs_f32x16 gs_mfma_32x32x2(float a, float b, gs_f32x16 c) { return emu::mfma_32x32x2(a, b, c); }

#include "../../jvector_amd/csrc/ed_body.h"

namespace {
struct Launch {
    const jv::EdParams *p;
    int vsf;
    int64_t n_tile;
    int q_tile;
    float *lds;
};
void tile_main(void *a)
{
    const Launch &L = *(const Launch *)a;
    switch (L.vsf) {
    case 0: jv::ed_tile<0>(*L.p, L.n_tile * jv::ED_TN, L.q_tile * jv::ED_TQ, L.lds); break;
    case 1: jv::ed_tile<1>(*L.p, L.n_tile * jv::ED_TN, L.q_tile * jv::ED_TQ, L.lds); break;
    default: jv::ed_tile<2>(*L.p, L.n_tile * jv::ED_TN, L.q_tile * jv::ED_TQ, L.lds); break;
    }
}
}  // namespace

// returns the number of tiles run, or -1 when the block -> tile map does not cover every tile exactly once
extern "C" long ed_emu_scan(const float *vecs, const float *queries, float *out, int64_t first, int64_t count, int D, int Q, int vsf)
{
    const jv::EdParams p{vecs, queries, out, first, count, D, Q};
    const int64_t n_tiles = (count + jv::ED_TN - 1) / jv::ED_TN;
    const int q_tiles = (Q + jv::ED_TQ - 1) / jv::ED_TQ;
    const int64_t blocks_padded = (n_tiles * q_tiles + 7) / 8 * 8;
    std::vector<int> seen((size_t)(n_tiles * q_tiles), 0);
    float *lds = (float *)aligned_alloc(64, sizeof(float) * (size_t)jv::ED_LDS_FLOATS);
    long ran = 0;
    for (int64_t b = 0; b < blocks_padded; ++b) {
        int64_t nt;
        int qt;
        if (!jv::ed_block_to_tile(b, blocks_padded, n_tiles, q_tiles, &nt, &qt)) continue;
        if (nt < 0 || nt >= n_tiles || qt < 0 || qt >= q_tiles || seen[(size_t)(nt * q_tiles + qt)]++) return -1;
        for (int i = 0; i < jv::ED_LDS_FLOATS; ++i) lds[i] = NAN;  // stale LDS must never reach a result
        Launch L{&p, vsf, nt, qt, lds};
        emu::run_block(tile_main, &L, jv::ED_WAVES);
        ++ran;
    }
    free(lds);
    return ran == n_tiles * q_tiles ? ran : -1;
}
