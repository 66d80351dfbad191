// nvq.cpp — host side of the NVQ entry points of include/jvector_hip.h (kernels: k_nvq.hip).
#include "jv_internal.h"

#include <algorithm>
#include <mutex>
#include <vector>

using namespace jv;

namespace {

std::mutex g_nvq_mu;

// The growth rates QuantizedSubVector.quantizeTo's two loops visit (B/quantization/NVQuantization.java:523-541), as the
// table nvq_encode_kernel reads: [0..20) the coarse loop's values, [20] the 1e-2f a search without a winner keeps, then 21
// rows of 21 fine values (row c = the loop around coarse value c), then the 21 fine counts.  The loops run here with the
// same float additions (-ffp-contract=off) the reference performs, so the device sees the same bits.
constexpr int kG = 21;
int build_growth_grid(std::vector<float> &g)
{
    g.assign((size_t)kG + kG * kG + kG, 0.0f);
    int nc = 0;
    for (float gr = 1e-6f; gr < 20.0f; gr += 1.0f) {
        if (nc >= 20) return -1;
        g[(size_t)nc++] = gr;
    }
    if (nc != 20) return -1;
    g[20] = 1e-2f;
    for (int c = 0; c < kG; ++c) {
        const float coarse = g[(size_t)c];
        int nf = 0;
        for (float gr = coarse - 1.0f; gr < coarse + 1.0f; gr += 0.1f) {
            if (nf >= kG) return -1;
            g[(size_t)kG + (size_t)c * kG + nf++] = gr;
        }
        g[(size_t)kG + kG * kG + c] = (float)nf;
    }
    return 0;
}

int nvq_new(jv_ctx *ctx, int D, int S, jv_nvq **out)
{
    JV_REQUIRE(D > 0 && S > 0, "nvq: D and the number of sub-vectors must be positive");
    // getSubvectorSizesAndOffsets :237-239
    JV_REQUIRE(S <= D, "nvq: number of subspaces must be less than or equal to the vector dimension (%d > %d)", S, D);
    JV_TRY(use_device(ctx->device));
    std::vector<float> grid;
    if (build_growth_grid(grid) != 0) {
        set_error("nvq: the growth-rate grid does not fit the kernel's 20 + 21 lanes");
        return JV_ERR_UNSUPPORTED;
    }
    jv_nvq *n = new jv_nvq();
    n->device = ctx->device;
    n->D = D;
    n->S = S;
    hipError_t e = hipMalloc((void **)&n->d_mean, sizeof(float) * (size_t)D);
    if (e == hipSuccess) e = hipMalloc((void **)&n->d_grid, sizeof(float) * grid.size());
    if (e == hipSuccess) e = hipMemcpy(n->d_grid, grid.data(), sizeof(float) * grid.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        set_error("nvq: device allocation failed: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        jv_hip_nvq_destroy(n);
        return JV_ERR_OOM;
    }
    *out = n;
    return JV_OK;
}

// derived quadruples (and, for cosine, the per-row normalisation sums) of rows that changed since the last use
int ensure_nvq_tables(jv_ctx *ctx, jv_nvq_vectors *nv, bool cosine)
{
    std::lock_guard<std::mutex> lk(g_nvq_mu);
    const jv_nvq *q = nv->nvq;
    if (!nv->derived_valid) {
        JV_TRY(launch_nvq_derive(ctx->stream, nv->d_params, nv->count * q->S, nv->d_derived));
        JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        nv->derived_valid = true;
        nv->cosnorm_valid = false;
    }
    if (cosine && !nv->cosnorm_valid) {
        if (!nv->d_cosnorm) JV_HIP_CHECK(hipMalloc((void **)&nv->d_cosnorm, sizeof(float) * (size_t)std::max<int64_t>(nv->count, 1)));
        {
            ProfScope ps(ctx, R_NORMS);
            JV_TRY(launch_nvq_cosnorm(ctx->stream, nv->d_bytes, nv->ld, nv->count, q->D, q->S, nv->d_derived, q->d_mean, nv->d_cosnorm));
        }
        JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        nv->cosnorm_valid = true;
    }
    return JV_OK;
}

int nvq_gather(jv_ctx *ctx, jv_nvq_vectors *nv, const float *d_q, int Q, jv_vsf vsf, const int32_t *d_ord, int B, float *d_out)
{
    const jv_nvq *q = nv->nvq;
    JV_TRY(ensure_nvq_tables(ctx, nv, vsf == JV_COSINE));
    // per-query scalars (dot product with the mean / query norm), and for EUCLIDEAN only the shifted queries
    const size_t work = vsf == JV_EUCLIDEAN ? (size_t)Q * q->D : 0;
    JV_TRY(ctx->d_nvq_q.reserve(sizeof(float) * (work + (size_t)Q)));
    float *qwork = (float *)ctx->d_nvq_q.ptr, *qaux = qwork + work;
    ProfScope ps(ctx, R_EXACT);
    return launch_nvq_gather(ctx->stream, nv->d_bytes, nv->ld, nv->count, q->D, q->S, nv->d_derived, nv->d_cosnorm, q->d_mean, d_q, Q,
                             to_kernel_vsf(vsf), d_ord, B, d_out, qwork, qaux);
}

}  // namespace

namespace jv {

int rerank_gather(jv_ctx *ctx, const jv_vectors *v, const float *d_q, int Q, jv_vsf vsf, const int32_t *d_ord, int B, float *d_out,
                  float *d_qnorm)
{
    if (v->nvq) return nvq_gather(ctx, v->nvq, d_q, Q, vsf, d_ord, B, d_out);
    if (vsf == JV_COSINE) JV_TRY(ensure_vector_norms(ctx, const_cast<jv_vectors *>(v)));
    ProfScope ps(ctx, R_EXACT);
    return launch_exact_gather(ctx->stream, v->d_vecs, v->count, v->D, d_q, Q, to_kernel_vsf(vsf), d_ord, B, d_out, d_qnorm, v->d_sqnorm);
}

}  // namespace jv

extern "C" {

int jv_hip_nvq_create(jv_ctx *ctx, int D, int n_subvectors, const float *global_mean, jv_nvq **out)
{
    clear_error();
    JV_REQUIRE(ctx && global_mean && out, "nvq_create: NULL argument");
    jv_nvq *n = nullptr;
    JV_TRY(nvq_new(ctx, D, n_subvectors, &n));
    hipError_t e = hipMemcpy(n->d_mean, global_mean, sizeof(float) * (size_t)D, hipMemcpyDefault);
    if (e != hipSuccess) {
        set_error("nvq_create: copying the global mean failed: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        jv_hip_nvq_destroy(n);
        return JV_ERR_HIP;
    }
    *out = n;
    return JV_OK;
}

int jv_hip_nvq_compute(jv_ctx *ctx, const jv_vectors *v, int n_subvectors, jv_nvq **out)
{
    clear_error();
    JV_REQUIRE(ctx && v && out, "nvq_compute: NULL argument");
    if (v->nvq) {
        set_error("nvq_compute: the vector set holds NVQ rows, not floats");
        return JV_ERR_UNSUPPORTED;
    }
    JV_REQUIRE(v->count <= 0x7fffffffLL, "nvq_compute: more rows than a RandomAccessVectorValues holds");
    jv_nvq *n = nullptr;
    JV_TRY(nvq_new(ctx, v->D, n_subvectors, &n));
    int rc = launch_nvq_mean(ctx->stream, v->d_vecs, v->count, v->D, n->d_mean);
    if (rc == JV_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) {
        set_error("nvq_compute: the mean kernel failed");
        (void)hipGetLastError();
        rc = JV_ERR_HIP;
    }
    if (rc != JV_OK) {
        jv_hip_nvq_destroy(n);
        return rc;
    }
    *out = n;
    return JV_OK;
}

int jv_hip_nvq_set_learn(jv_nvq *nvq, int learn)
{
    clear_error();
    JV_REQUIRE(nvq, "nvq_set_learn: NULL argument");
    nvq->learn = learn != 0;
    return JV_OK;
}

int jv_hip_nvq_dimension(const jv_nvq *nvq) { return nvq ? nvq->D : 0; }
int jv_hip_nvq_subvectors(const jv_nvq *nvq) { return nvq ? nvq->S : 0; }

int jv_hip_nvq_global_mean(jv_ctx *ctx, const jv_nvq *nvq, float *dst)
{
    clear_error();
    JV_REQUIRE(ctx && nvq && dst, "nvq_global_mean: NULL argument");
    JV_TRY(use_device(ctx->device));
    JV_HIP_CHECK(hipMemcpyAsync(dst, nvq->d_mean, sizeof(float) * (size_t)nvq->D, hipMemcpyDefault, ctx->stream));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return JV_OK;
}

int jv_hip_nvq_destroy(jv_nvq *nvq)
{
    if (!nvq) return JV_OK;
    (void)hipSetDevice(nvq->device);
    (void)hipFree(nvq->d_mean);
    (void)hipFree(nvq->d_grid);
    delete nvq;
    return JV_OK;
}

int jv_hip_nvq_vectors_create(jv_ctx *ctx, const jv_nvq *nvq, int64_t count, jv_nvq_vectors **out)
{
    clear_error();
    JV_REQUIRE(ctx && nvq && out, "nvq_vectors_create: NULL argument");
    JV_REQUIRE(count > 0, "nvq_vectors_create: count must be positive");
    JV_REQUIRE(ctx->device == nvq->device, "nvq_vectors_create: the NVQuantization lives on device %d, the context on %d", nvq->device, ctx->device);
    JV_TRY(use_device(ctx->device));
    jv_nvq_vectors *nv = new jv_nvq_vectors();
    nv->device = ctx->device;
    nv->nvq = nvq;
    nv->count = count;
    nv->ld = (nvq->D + 15) / 16 * 16;
    const size_t units = (size_t)count * nvq->S;
    hipError_t e = hipMalloc((void **)&nv->d_bytes, (size_t)count * nv->ld);
    if (e == hipSuccess) e = hipMalloc((void **)&nv->d_params, sizeof(float) * 4 * units);
    if (e == hipSuccess) e = hipMalloc((void **)&nv->d_derived, sizeof(float) * 4 * units);
    if (e == hipSuccess) e = hipMemsetAsync(nv->d_bytes, 0, (size_t)count * nv->ld, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(nv->d_params, 0, sizeof(float) * 4 * units, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        set_error("nvq_vectors_create: device allocation failed: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        jv_hip_nvq_vectors_destroy(nv);
        return JV_ERR_OOM;
    }
    *out = nv;
    return JV_OK;
}

int jv_hip_nvq_encode(jv_ctx *ctx, const jv_nvq *nvq, const jv_vectors *v, int64_t first, int64_t count, jv_nvq_vectors *dst,
                      int64_t dst_first)
{
    clear_error();
    JV_REQUIRE(ctx && nvq && v && dst, "nvq_encode: NULL argument");
    JV_REQUIRE(dst->nvq == nvq, "nvq_encode: the destination belongs to another NVQuantization");
    JV_REQUIRE(ctx->device == nvq->device && ctx->device == v->device && ctx->device == dst->device,
               "nvq_encode: context, NVQuantization, vectors and destination must share a device (%d / %d / %d / %d)", ctx->device, nvq->device,
               v->device, dst->device);
    if (v->nvq) {
        set_error("nvq_encode: the vector set holds NVQ rows, not floats");
        return JV_ERR_UNSUPPORTED;
    }
    JV_REQUIRE(v->D == nvq->D, "nvq_encode: vector dimension %d != %d", v->D, nvq->D);
    JV_REQUIRE(first >= 0 && count >= 0 && first + count <= v->count, "nvq_encode: source range out of bounds");
    JV_REQUIRE(dst_first >= 0 && dst_first + count <= dst->count, "nvq_encode: destination range out of bounds");
    if (count == 0) return JV_OK;
    JV_TRY(use_device(ctx->device));
    {
        ProfScope ps(ctx, R_ENCODE);
        // at most 2^31 - 1 blocks of three units per launch
        const int64_t step = std::max<int64_t>(1, (int64_t)0x7ffffff0LL * 3 / nvq->S);
        for (int64_t o = 0; o < count; o += step) {
            const int64_t c = std::min(step, count - o);
            JV_TRY(launch_nvq_encode(ctx->stream, ctx, v->d_vecs + (first + o) * v->D, c, nvq->D, nvq->S, nvq->d_mean, nvq->learn ? 1 : 0,
                                     nvq->d_grid, dst->d_bytes + (dst_first + o) * dst->ld, dst->ld,
                                     dst->d_params + (dst_first + o) * 4 * nvq->S));
        }
    }
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    dst->derived_valid = false;
    dst->cosnorm_valid = false;
    return JV_OK;
}

int jv_hip_nvq_vectors_upload(jv_ctx *ctx, jv_nvq_vectors *nv, int64_t first, int64_t count, const uint8_t *bytes, const float *params)
{
    clear_error();
    JV_REQUIRE(ctx && nv && bytes && params, "nvq_vectors_upload: NULL argument");
    JV_REQUIRE(first >= 0 && count >= 0 && first + count <= nv->count, "nvq_vectors_upload: range out of bounds");
    if (count == 0) return JV_OK;
    JV_TRY(use_device(ctx->device));
    const int D = nv->nvq->D, S = nv->nvq->S;
    JV_HIP_CHECK(hipMemcpy2DAsync(nv->d_bytes + first * nv->ld, (size_t)nv->ld, bytes, (size_t)D, (size_t)D, (size_t)count, hipMemcpyDefault,
                                  ctx->stream));
    JV_HIP_CHECK(hipMemcpyAsync(nv->d_params + first * 4 * S, params, sizeof(float) * 4 * (size_t)count * S, hipMemcpyDefault, ctx->stream));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    nv->derived_valid = false;
    nv->cosnorm_valid = false;
    return JV_OK;
}

int jv_hip_nvq_vectors_download(jv_ctx *ctx, const jv_nvq_vectors *nv, int64_t first, int64_t count, uint8_t *bytes, float *params)
{
    clear_error();
    JV_REQUIRE(ctx && nv, "nvq_vectors_download: NULL argument");
    JV_REQUIRE(first >= 0 && count >= 0 && first + count <= nv->count, "nvq_vectors_download: range out of bounds");
    if (count == 0) return JV_OK;
    JV_TRY(use_device(ctx->device));
    const int D = nv->nvq->D, S = nv->nvq->S;
    if (bytes)
        JV_HIP_CHECK(hipMemcpy2DAsync(bytes, (size_t)D, nv->d_bytes + first * nv->ld, (size_t)nv->ld, (size_t)D, (size_t)count, hipMemcpyDefault,
                                      ctx->stream));
    if (params)
        JV_HIP_CHECK(hipMemcpyAsync(params, nv->d_params + first * 4 * S, sizeof(float) * 4 * (size_t)count * S, hipMemcpyDefault, ctx->stream));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return JV_OK;
}

int64_t jv_hip_nvq_vectors_count(const jv_nvq_vectors *nv) { return nv ? nv->count : 0; }

int jv_hip_nvq_vectors_destroy(jv_nvq_vectors *nv)
{
    if (!nv) return JV_OK;
    (void)hipSetDevice(nv->device);
    (void)hipFree(nv->d_bytes);
    (void)hipFree(nv->d_params);
    (void)hipFree(nv->d_derived);
    (void)hipFree(nv->d_cosnorm);
    delete nv;
    return JV_OK;
}

int jv_hip_nvq_scores(jv_ctx *ctx, const jv_nvq_vectors *nv, const float *queries, int Q, jv_vsf vsf, const int32_t *ordinals, int B,
                      float *scores_out)
{
    clear_error();
    JV_REQUIRE(ctx && nv, "nvq_scores: NULL argument");
    JV_REQUIRE(Q >= 0 && B >= 0, "nvq_scores: negative sizes");
    JV_REQUIRE(vsf == JV_EUCLIDEAN || vsf == JV_DOT_PRODUCT || vsf == JV_COSINE, "nvq_scores: unsupported similarity function %d", (int)vsf);
    if (Q == 0 || B == 0) return JV_OK;
    JV_REQUIRE(queries && ordinals && scores_out, "nvq_scores: NULL buffer");
    JV_REQUIRE(ctx->device == nv->device, "nvq_scores: the NVQ rows live on device %d, the context on %d", nv->device, ctx->device);
    JV_TRY(use_device(ctx->device));
    const void *d_q = nullptr, *d_ord = nullptr;
    JV_TRY(stage_in(ctx, queries, sizeof(float) * (size_t)Q * nv->nvq->D, ctx->h_in, ctx->d_in, &d_q));
    JV_TRY(stage_in(ctx, ordinals, sizeof(int32_t) * (size_t)Q * B, ctx->h_in, ctx->d_scratch2, &d_ord));
    OutStage os;
    JV_TRY(stage_out_begin(ctx, scores_out, sizeof(float) * (size_t)Q * B, ctx->d_out, &os));
    JV_TRY(nvq_gather(ctx, const_cast<jv_nvq_vectors *>(nv), (const float *)d_q, Q, vsf, (const int32_t *)d_ord, B, (float *)os.dev));
    return stage_out_end(ctx, os);
}

int jv_hip_vectors_from_nvq(jv_ctx *ctx, jv_nvq_vectors *nv, jv_vectors **out)
{
    clear_error();
    JV_REQUIRE(ctx && nv && out, "vectors_from_nvq: NULL argument");
    JV_REQUIRE(ctx->device == nv->device, "vectors_from_nvq: the NVQ rows live on device %d, the context on %d", nv->device, ctx->device);
    jv_vectors *v = new jv_vectors();
    v->device = nv->device;
    v->count = nv->count;
    v->D = nv->nvq->D;
    v->owns = false;
    v->nvq = nv;
    *out = v;
    return JV_OK;
}

}  // extern "C"
