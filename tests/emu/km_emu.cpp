// km_emu.cpp — PQ training kernel bodies (jvector_amd/csrc/km_body.h) on the CPU: per-thread bodies as loops, the
// k-means++ seeding on the 64-lane wave emulator; the host sequence mirrors jvector_amd/csrc/pq_train.cpp.  TEST HARNESS.
#include <vector>

#include "hip_emu.h"

#define GS_FN inline
#define GS_SCHED_FENCE() ((void)0)
#define KM_FN static inline
static inline int gs_lane() { return emu::lane(); }
static inline void gs_barrier() { emu::barrier(); }
static inline long long gs_shfl(long long v, int src) { return emu::shfl(v, src); }
static inline void gs_fence() {}
static inline void gs_gather64(float v, float (&out)[64]) { emu::gather64(v, out); }

#include "../../jvector_amd/csrc/km_body.h"

namespace {
struct PP {
    const jv::KmParams *p;
    int m;
};
void pp_main(void *arg)
{
    const PP &a = *(const PP *)arg;
    jv::km_pp_init(*a.p, a.m);
}
}  // namespace

// mode 0: train from scratch (k-means++ then `rounds` Lloyd rounds); mode 1: refine the codebooks given in C
extern "C" void km_emu_train(const float *X, int64_t n, int D, int M, int k, const int64_t *cb_offsets, const int32_t *sizes,
                             const int32_t *offsets, int globally_center, uint64_t seed, int rounds, int mode, const float *centroid_in,
                             float *C, float *centroid_out, int aniso_rounds, const float *pcm)
{
    std::vector<float> Xc((size_t)n * D), cen((size_t)D, 0.0f);
    const float *cptr = nullptr;
    if (mode == 0 && globally_center) {
        for (int64_t d = 0; d < D; ++d) jv::km_centroid_dim(X, n, D, d, cen.data());
        cptr = cen.data();
        for (int d = 0; d < D; ++d) centroid_out[d] = cen[d];
    } else if (mode == 1 && centroid_in) {
        cptr = centroid_in;
    }
    for (int64_t t = 0; t < n * D; ++t) jv::km_center(X, cptr, D, t, Xc.data());
    size_t total = 0;
    for (int m = 0; m < M; ++m) total += (size_t)k * sizes[m];
    std::vector<uint8_t> A((size_t)n * M), B((size_t)n * M);
    std::vector<float> nums(total), dist((size_t)M * n), cnorm((size_t)M * k);
    std::vector<int32_t> denoms((size_t)M * k), active((size_t)M, 1), changed((size_t)M, 0);
    std::vector<uint64_t> rng((size_t)M);
    for (int m = 0; m < M; ++m) rng[m] = jv::km_stream(seed, m);
    jv::KmParams p{Xc.data(), C, cb_offsets, sizes, offsets, A.data(), B.data(), nums.data(), denoms.data(), active.data(),
                   changed.data(), rng.data(), dist.data(), cnorm.data(), pcm, n, D, M, k};
    if (mode == 0)
        for (int m = 0; m < M; ++m) {
            PP a{&p, m};
            emu::run_wave(pp_main, &a);
        }
    // constructor: initializeAssignedPoints
    for (int64_t t = 0; t < n * M; ++t) jv::km_assign(p, t);
    for (int64_t t = 0; t < (int64_t)M * k; ++t) jv::km_replay(p, 1, t);
    for (int it = 0; it < rounds; ++it) {
        std::swap(p.assign_old, p.assign_new);
        for (int64_t t = 0; t < (int64_t)M * k; ++t) jv::km_centroids(p, t);
        for (int64_t m = 0; m < M; ++m) jv::km_fill_empties(p, m);
        for (int64_t t = 0; t < n * M; ++t) jv::km_assign(p, t);
        for (int64_t t = 0; t < (int64_t)M * k; ++t) jv::km_replay(p, 0, t);
        for (int64_t m = 0; m < M; ++m) jv::km_finish_round(p, m);
    }
    if (aniso_rounds > 0)
        for (int64_t m = 0; m < M; ++m) jv::km_reactivate(p, m);
    for (int it = 0; it < aniso_rounds; ++it) {
        std::swap(p.assign_old, p.assign_new);
        for (int64_t t = 0; t < (int64_t)M * k; ++t) jv::km_centroids_aniso(p, t);
        for (int64_t m = 0; m < M; ++m) jv::km_fill_empties(p, m);
        for (int64_t t = 0; t < (int64_t)M * k; ++t) jv::km_cnorm(p, t);
        for (int64_t t = 0; t < n * M; ++t) jv::km_assign_aniso(p, t);
        for (int64_t m = 0; m < M; ++m) jv::km_count_changed(p, m);
        for (int64_t m = 0; m < M; ++m) jv::km_finish_round(p, m);
    }
}
