// an_emu.cpp — the anisotropic PQ encode body (jvector_amd/csrc/an_body.h) on the lane emulator.  TEST HARNESS.
#include "hip_emu.h"

#define GS_FN inline
#define GS_SCHED_FENCE() ((void)0)
static inline int gs_lane() { return emu::lane(); }
static inline void gs_barrier() { emu::barrier(); }
static inline uint64_t gs_ballot(bool p) { return emu::ballot(p); }
static inline long long gs_shfl(long long v, int src) { return emu::shfl(v, src); }
static inline long long gs_shfl_xor(long long v, int m) { return emu::shfl(v, emu::lane() ^ m); }
static inline double gs_sqrt(double x) { return std::sqrt(x); }

#include "../../jvector_amd/csrc/an_body.h"

namespace {
struct Launch {
    const jv::AnParams *p;
    int worker, workers;
    char *lds;
};
void lane_main(void *arg)
{
    const Launch &L = *(const Launch *)arg;
    jv::an_worker(*L.p, L.worker, L.workers, L.lds);
}
}  // namespace

extern "C" void an_emu_encode(const float *codebooks, const int64_t *cb_offsets, const int32_t *sizes, const int32_t *offsets,
                              const float *centroid, const float *cnorm, int D, int M, int k, float pcm, const float *vecs, int64_t n,
                              int workers, uint8_t *codes)
{
    jv::AnParams p{codebooks, cb_offsets, sizes, offsets, centroid, cnorm, vecs, codes, n, D, M, k, pcm};
    for (int w = 0; w < workers; ++w) {
        char *lds = (char *)aligned_alloc(64, jv::an_lds_bytes(D, M) + 64);
        memset(lds, 0xa5, jv::an_lds_bytes(D, M));
        Launch L{&p, w, workers, lds};
        emu::run_wave(lane_main, &L);
        free(lds);
    }
}
