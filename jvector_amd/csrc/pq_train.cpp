// pq_train.cpp — C ABI of PQ training: ProductQuantization.compute / refine, unweighted
// (B/quantization/ProductQuantization.java:109-139,194-221; kernels k_pq_train.hip, bodies km_body.h).
#include <algorithm>
#include <cstring>
#include <vector>

#define KM_FN static inline
#include "km_body.h"

#include "jv_internal.h"

using namespace jv;

namespace {

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes)
    {
        JV_HIP_CHECK(hipMalloc(&p, std::max<size_t>(bytes, 16)));
        return JV_OK;
    }
};

// shared by train (init = k-means++) and refine (init = the given codebooks)
// KMeansPlusPlusClusterer.computeParallelCostMultiplier :116-124
float parallel_cost_multiplier(float threshold, int dimensions)
{
    const double t = (double)threshold;
    const double parallelCost = t * t;
    const double perpendicularCost = (1 - parallelCost) / (dimensions - 1);
    return (float)std::max(1.0, parallelCost / perpendicularCost);
}

// getNearestCluster for every (training point, subspace) = closestCentroidIndex of the point under the CURRENT centroids: the same
// squareL2Distance chain, the same strict `<` / first minimum (KMeansPlusPlusClusterer.java:330-343 against ProductQuantization.java:
// 507-520) — so a round's assignment is an encode of the centred sample with the work quantizer, and runs on the encode kernel
// (57 TFLOP/s, k_pq.hip) instead of one thread per (point, subspace) walking 256 centroids through divergent global loads (0.5
// TFLOP/s: 1.5 of the 3.7 s of a PQ-192 training).  A converged subspace's centroids no longer move, so re-encoding it reproduces the
// assignment km_assign would copy.  Uniform sub-vectors of the encode kernel's sizes and 256 clusters; anything else, or
// JVECTOR_HIP_KM_ASSIGN_PLAIN=1, takes km_assign.
static int assign_points(hipStream_t s, const jv_pq *work, const KmParams &p)
{
    const int sz = work->max_size;
    const bool fast = work->uniform && work->d_cb_paired && work->k == kClusters && work->k_user == kClusters &&
                      (sz == 2 || sz == 4 || sz == 6 || sz == 8 || sz == 12 || sz == 16) && !getenv("JVECTOR_HIP_KM_ASSIGN_PLAIN");
    if (!fast) return launch_km_assign(s, p);
    jv_pq view = *work;            // (a shallow copy: the handle owns nothing through its destructor)
    view.d_centroid = nullptr;     // the sample is centred already
    JV_TRY(launch_self_magnitudes(s, &view));   // refreshes the paired copy of the centroids the encode kernel reads
    return launch_pq_encode(s, &view, p.X, p.n, p.assign_new);
}

int run_training(jv_ctx *ctx, jv_pq *work /* layout + d_codebooks in/out */, const float *vectors, int64_t n, const float *d_centroid,
                 bool compute_centroid, float *d_centroid_out, bool seed_with_kmeans_pp, int rounds, uint64_t seed,
                 int aniso_rounds = 0, float aniso_threshold = -1.0f)
{
    const int D = work->D, M = work->M, k = work->k;
    hipStream_t s = ctx->stream;
    const void *d_X = nullptr;
    JV_TRY(stage_in(ctx, vectors, sizeof(float) * (size_t)n * D, ctx->h_in, ctx->d_in, &d_X));
    DevBuf Xc, A, B, nums, denoms, active, changed, rng, dist, cnorm, pcm;
    JV_TRY(Xc.alloc(sizeof(float) * (size_t)n * D));
    JV_TRY(A.alloc((size_t)n * M));
    JV_TRY(B.alloc((size_t)n * M));
    JV_TRY(nums.alloc(sizeof(float) * (size_t)k * D));
    JV_TRY(denoms.alloc(sizeof(int32_t) * (size_t)M * k));
    JV_TRY(active.alloc(sizeof(int32_t) * (size_t)M));
    JV_TRY(changed.alloc(sizeof(int32_t) * (size_t)M));
    JV_TRY(rng.alloc(sizeof(uint64_t) * (size_t)M));
    if (seed_with_kmeans_pp) JV_TRY(dist.alloc(sizeof(float) * (size_t)M * n));
    std::vector<float> h_pcm((size_t)M, 1.0f);
    if (aniso_rounds > 0) {
        JV_TRY(cnorm.alloc(sizeof(float) * (size_t)M * k));
        JV_TRY(pcm.alloc(sizeof(float) * (size_t)M));
        for (int m = 0; m < M; ++m) h_pcm[m] = parallel_cost_multiplier(aniso_threshold, work->sizes[m]);
        JV_HIP_CHECK(hipMemcpyAsync(pcm.p, h_pcm.data(), sizeof(float) * (size_t)M, hipMemcpyHostToDevice, s));
    }
    if (compute_centroid) {
        JV_TRY(launch_km_centroid(s, (const float *)d_X, n, D, d_centroid_out));
        d_centroid = d_centroid_out;
    }
    JV_TRY(launch_km_center(s, (const float *)d_X, d_centroid, n, D, (float *)Xc.p));
    std::vector<int32_t> ones((size_t)M, 1);
    std::vector<uint64_t> streams((size_t)M);
    for (int m = 0; m < M; ++m) streams[m] = km_stream(seed, m);
    JV_HIP_CHECK(hipMemcpyAsync(active.p, ones.data(), sizeof(int32_t) * (size_t)M, hipMemcpyHostToDevice, s));
    JV_HIP_CHECK(hipMemcpyAsync(rng.p, streams.data(), sizeof(uint64_t) * (size_t)M, hipMemcpyHostToDevice, s));
    JV_HIP_CHECK(hipStreamSynchronize(s));  // the two host vectors above go out of scope with this frame: drain now
    KmParams p{(const float *)Xc.p, work->d_codebooks, work->d_cb_offsets, work->d_sizes, work->d_offsets, (uint8_t *)A.p,
               (uint8_t *)B.p, (float *)nums.p, (int32_t *)denoms.p, (int32_t *)active.p, (int32_t *)changed.p, (uint64_t *)rng.p,
               (float *)dist.p, (float *)cnorm.p, (const float *)pcm.p, n, D, M, k};
    if (seed_with_kmeans_pp) JV_TRY(launch_km_pp_init(s, p));
    // KMeansPlusPlusClusterer constructor: initializeAssignedPoints
    JV_TRY(assign_points(s, work, p));
    JV_TRY(launch_km_replay(s, p, 1));
    for (int it = 0; it < rounds; ++it) {  // cluster(rounds, 0): subspaces that converged (<= 1 % moved) go inactive
        std::swap(p.assign_old, p.assign_new);
        JV_TRY(launch_km_update_centroids(s, p));
        JV_TRY(assign_points(s, work, p));
        JV_TRY(launch_km_replay(s, p, 0));
        JV_TRY(launch_km_finish_round(s, p));
    }
    if (aniso_rounds > 0) JV_TRY(launch_km_reactivate(s, p));  // cluster() :141-147: a second loop with its own early stop
    for (int it = 0; it < aniso_rounds; ++it) {
        std::swap(p.assign_old, p.assign_new);
        JV_TRY(launch_km_aniso_round(s, p));
        JV_TRY(launch_km_finish_round(s, p));
    }
    JV_HIP_CHECK(hipStreamSynchronize(s));
    return JV_OK;
}

int finish(jv_ctx *ctx, jv_pq *work, const float *d_centroid, float aniso, jv_pq **out)
{
    const int D = work->D, k = work->k;
    std::vector<float> cb((size_t)k * D), cen;
    JV_HIP_CHECK(hipMemcpy(cb.data(), work->d_codebooks, sizeof(float) * cb.size(), hipMemcpyDeviceToHost));
    if (d_centroid) {
        cen.resize((size_t)D);
        JV_HIP_CHECK(hipMemcpy(cen.data(), d_centroid, sizeof(float) * (size_t)D, hipMemcpyDeviceToHost));
    }
    JV_TRY(jv_hip_pq_create(ctx, D, work->M, k, work->sizes.data(), cb.data(), d_centroid ? cen.data() : nullptr, out));
    (*out)->aniso = aniso;
    return JV_OK;
}

struct PqGuard {
    jv_pq *pq = nullptr;
    ~PqGuard() { if (pq) jv_hip_pq_destroy(pq); }
};

}  // namespace

extern "C" {

int jv_hip_pq_train(jv_ctx *ctx, const float *vectors, int64_t n, int D, int M, int k, int globally_center, uint64_t seed, jv_pq **out)
{
    return jv_hip_pq_train_anisotropic(ctx, vectors, n, D, M, k, globally_center, -1.0f, seed, out);
}

int jv_hip_pq_train_anisotropic(jv_ctx *ctx, const float *vectors, int64_t n, int D, int M, int k, int globally_center,
                                float anisotropic_threshold, uint64_t seed, jv_pq **out)
{
    clear_error();
    const float T = anisotropic_threshold;
    JV_REQUIRE(T == T && T >= -1.0f && T < 1.0f, "Valid range for anisotropic threshold T is -1.0 <= t < 1.0");
    const bool aniso = T > -1.0f;
    JV_REQUIRE(!aniso || (D + M - 1) / M <= KM_ANISO_MAX_LEN, "pq_train: anisotropic k-means supports sub-vectors of at most %d dimensions",
               KM_ANISO_MAX_LEN);
    JV_REQUIRE(!aniso || D / M >= 2, "pq_train: anisotropic k-means needs sub-vectors of at least 2 dimensions");
    JV_REQUIRE(ctx && vectors && out, "pq_train: NULL argument");
    *out = nullptr;
    JV_REQUIRE(D > 0 && M > 0 && M <= D, "Number of subspaces must be less than or equal to the vector dimension");
    // ProductQuantization.compute :119-121
    JV_REQUIRE(n >= k, "Cannot train PQ with %d clusters on %lld points, supply more training vectors or lower cluster count.", k,
               (long long)n);
    JV_REQUIRE((D + M - 1) / M <= 64, "pq_train: sub-vectors longer than 64 dimensions are not supported");
    JV_REQUIRE(k >= 1 && k <= kClusters, "pq_train: clusterCount %d outside 1..256", k);  // ProductQuantization.checkClusterCount
    if (aniso && k != kClusters) {  // (the padded rows of the finished quantizer would take part in anisotropic encoding)
        set_error("pq_train: anisotropic training needs 256 clusters here (clusterCount %d)", k);
        return JV_ERR_UNSUPPORTED;
    }
    JV_TRY(use_device(ctx->device));
    PqGuard work;
    std::vector<float> zeros((size_t)k * D, 0.0f);
    JV_TRY(pq_create_impl(ctx, D, M, k, nullptr, zeros.data(), nullptr, false, &work.pq));  // (k-row work layout: KmParams::k)
    DevBuf cen;
    if (globally_center) JV_TRY(cen.alloc(sizeof(float) * (size_t)D));
    JV_TRY(run_training(ctx, work.pq, vectors, n, nullptr, globally_center != 0, (float *)cen.p, true, 6 /* K_MEANS_ITERATIONS */, seed,
                        aniso ? 6 : 0, T));
    return finish(ctx, work.pq, globally_center ? (const float *)cen.p : nullptr, T, out);
}

int jv_hip_pq_refine(jv_ctx *ctx, const jv_pq *pq, const float *vectors, int64_t n, int lloyds_rounds, uint64_t seed, jv_pq **out)
{
    clear_error();
    JV_REQUIRE(ctx && pq && vectors && out, "pq_refine: NULL argument");
    *out = nullptr;
    JV_REQUIRE(lloyds_rounds >= 0, "lloydsRounds must be non-negative");  // ProductQuantization.refine :205-207

    JV_REQUIRE(n > 0, "pq_refine: no training vectors");
    JV_REQUIRE(pq->max_size <= 64, "pq_refine: sub-vectors longer than 64 dimensions are not supported");
    const bool aniso = pq->aniso > -1.0f;  // refine :212-214: cluster(aniso ? 0 : rounds, aniso ? rounds : 0)
    JV_REQUIRE(!aniso || (pq->max_size <= KM_ANISO_MAX_LEN && pq->D / pq->M >= 2),
               "pq_refine: anisotropic k-means supports sub-vectors of 2..%d dimensions", KM_ANISO_MAX_LEN);
    JV_TRY(use_device(ctx->device));
    PqGuard work;
    std::vector<float> cb((size_t)pq->k_user * pq->D);
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    {
        size_t so = 0, po = 0;
        for (int m = 0; m < pq->M; ++m) {  // the caller's clusters: the first k_user of every codebook's 256 device rows
            const size_t S = (size_t)pq->sizes[(size_t)m];
            JV_HIP_CHECK(hipMemcpy(cb.data() + so, pq->d_codebooks + po, sizeof(float) * S * (size_t)pq->k_user, hipMemcpyDeviceToHost));
            so += S * (size_t)pq->k_user;
            po += S * (size_t)pq->k;
        }
    }
    JV_TRY(pq_create_impl(ctx, pq->D, pq->M, pq->k_user, pq->sizes.data(), cb.data(), nullptr, false, &work.pq));
    JV_TRY(run_training(ctx, work.pq, vectors, n, pq->d_centroid, false, nullptr, false, aniso ? 0 : lloyds_rounds, seed,
                        aniso ? lloyds_rounds : 0, pq->aniso));
    return finish(ctx, work.pq, pq->d_centroid, pq->aniso, out);
}

// ProductQuantization.write(out, version) :560-599, big-endian.  len_out = bytes needed; nothing is written when cap is
// too small (query the size with buf = NULL, cap = 0).
int jv_hip_pq_write(jv_ctx *ctx, const jv_pq *pq, int version, uint8_t *buf, size_t cap, size_t *len_out)
{
    clear_error();
    JV_REQUIRE(ctx && pq && len_out, "pq_write: NULL argument");
    JV_REQUIRE(version >= 0 && version <= 6, "Unsupported serialization version %d", version);
    JV_REQUIRE(version >= 3 || !(pq->aniso > -1.0f), "Anisotropic threshold is only supported in serialization version 3 and above");
    const size_t total = (size_t)pq->k_user * pq->D;   // the caller's clusterCount: padded rows are not part of the format
    const size_t need = (version >= 3 ? 8 : 0) + 4 + (pq->d_centroid ? (size_t)pq->D * 4 : 0) + 4 + (size_t)pq->M * 4 +
                        (version >= 3 ? 4 : 0) + 4 + total * 4;
    *len_out = need;
    if (!buf || cap < need) return JV_OK;
    JV_TRY(use_device(ctx->device));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    std::vector<float> cb(total), cen;
    {
        size_t so = 0, po = 0;
        for (int m = 0; m < pq->M; ++m) {  // codebook m: the first k_user of its 256 device rows
            const size_t S = (size_t)pq->sizes[(size_t)m];
            JV_HIP_CHECK(hipMemcpy(cb.data() + so, pq->d_codebooks + po, sizeof(float) * S * (size_t)pq->k_user, hipMemcpyDeviceToHost));
            so += S * (size_t)pq->k_user;
            po += S * (size_t)pq->k;
        }
    }
    if (pq->d_centroid) {
        cen.resize((size_t)pq->D);
        JV_HIP_CHECK(hipMemcpy(cen.data(), pq->d_centroid, sizeof(float) * (size_t)pq->D, hipMemcpyDeviceToHost));
    }
    size_t p = 0;
    auto wr_i32 = [&](int32_t v) {
        buf[p++] = (uint8_t)((uint32_t)v >> 24);
        buf[p++] = (uint8_t)((uint32_t)v >> 16);
        buf[p++] = (uint8_t)((uint32_t)v >> 8);
        buf[p++] = (uint8_t)(uint32_t)v;
    };
    auto wr_f32 = [&](float f) {
        int32_t b;
        memcpy(&b, &f, 4);
        wr_i32(b);
    };
    if (version >= 3) {
        wr_i32(0x75EC4012);
        wr_i32(version);
    }
    if (cen.empty()) wr_i32(0);
    else {
        wr_i32(pq->D);
        for (float f : cen) wr_f32(f);
    }
    wr_i32(pq->M);
    for (int m = 0; m < pq->M; ++m) wr_i32(pq->sizes[m]);
    if (version >= 3) wr_f32(pq->aniso);
    wr_i32(pq->k_user);
    for (float f : cb) wr_f32(f);
    return JV_OK;
}

}  // extern "C"
