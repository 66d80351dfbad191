// cabi.cpp — host side of libjvector_hip.so: contexts, device-resident objects, staging of host
// buffers, and the extern "C" entry points declared in include/jvector_hip.h.
//
// No CPU fallback lives here: every compute entry point launches HIP kernels on the context's stream and
// fails with JV_ERR_NO_DEVICE / JV_ERR_HIP when that is impossible.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>

#include "jv_device.h"
#include "jv_internal.h"
#include "gs_params.h"
#include "../../include/jvector_formats.h"

namespace jv {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
void clear_error() { g_err[0] = 0; }

int Buffer::reserve(size_t bytes)
{
    if (bytes <= cap) return JV_OK;
    size_t want = std::max(bytes, cap + cap / 2);
    want = (want + 4095) & ~(size_t)4095;
    void *np = nullptr;
    hipError_t e = pinned_host ? hipHostMalloc(&np, want, hipHostMallocDefault) : hipMalloc(&np, want);
    if (e != hipSuccess) {
        set_error("%s of %zu bytes failed: %s", pinned_host ? "hipHostMalloc" : "hipMalloc", want, hipGetErrorString(e));
        (void)hipGetLastError();
        return JV_ERR_OOM;
    }
    release();
    ptr = np;
    cap = want;
    return JV_OK;
}

void Buffer::release()
{
    if (ptr) {
        if (pinned_host) (void)hipHostFree(ptr);
        else (void)hipFree(ptr);
    }
    ptr = nullptr;
    cap = 0;
}

// true when p is device-accessible memory we can hand to a kernel directly
bool is_device_ptr(const void *p)
{
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // plain host memory: not an error for us
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// Brings `bytes` at `src` onto the device.  Device pointers pass through; host memory is copied through
// the pinned staging buffer `pin` into `dev` on the context's stream.
int stage_in(jv_ctx *ctx, const void *src, size_t bytes, Buffer &pin, Buffer &dev, const void **out)
{
    if (bytes == 0) {
        *out = src;
        return JV_OK;
    }
    if (is_device_ptr(src)) {
        *out = src;
        return JV_OK;
    }
    JV_TRY(dev.reserve(bytes));
    // The pinned buffer is reused across calls: make sure the previous async copy out of it has finished.
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    JV_TRY(pin.reserve(bytes));
    memcpy(pin.ptr, src, bytes);
    JV_HIP_CHECK(hipMemcpyAsync(dev.ptr, pin.ptr, bytes, hipMemcpyHostToDevice, ctx->stream));
    *out = dev.ptr;
    return JV_OK;
}

int stage_out_begin(jv_ctx *ctx, void *dst, size_t bytes, Buffer &dev, OutStage *st)
{
    st->user = dst;
    st->bytes = bytes;
    if (bytes == 0 || is_device_ptr(dst)) {
        st->dev = dst;
        st->host = false;
        return JV_OK;
    }
    JV_TRY(dev.reserve(bytes));
    st->dev = dev.ptr;
    st->host = true;
    return JV_OK;
}

// Copies a staged output back to the user's host buffer (synchronises the stream); no-op for device outputs.
int stage_out_end(jv_ctx *ctx, const OutStage &st)
{
    if (!st.host || st.bytes == 0) return JV_OK;
    JV_TRY(ctx->h_out.reserve(st.bytes));
    JV_HIP_CHECK(hipMemcpyAsync(ctx->h_out.ptr, st.dev, st.bytes, hipMemcpyDeviceToHost, ctx->stream));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    memcpy(st.user, ctx->h_out.ptr, st.bytes);
    return JV_OK;
}

int to_kernel_vsf(jv_vsf v)
{
    switch (v) {
    case JV_EUCLIDEAN: return VSF_L2;
    case JV_DOT_PRODUCT: return VSF_DOT;
    default: return VSF_COS;
    }
}

int use_device(int device)
{
    JV_HIP_CHECK(hipSetDevice(device));
    return JV_OK;
}

// big-endian readers for the reference's wire format (B/disk/IndexWriter.java:36-42)
static int32_t rd_i32(const uint8_t *p)
{
    return (int32_t)(((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]);
}
static float rd_f32(const uint8_t *p)
{
    int32_t b = rd_i32(p);
    float f;
    memcpy(&f, &b, 4);
    return f;
}

// The per-code cosine magnitudes are a lazily built, query-independent cache shared by every context that uses the
// code store: build under a lock and drain the building stream before publishing, so a second context (another
// host thread, another stream) never reads a half-written table.
static std::mutex g_norms_mu;

int ensure_code_norms(jv_ctx *ctx, jv_codes *codes)
{
    std::lock_guard<std::mutex> lk(g_norms_mu);
    if (codes->norms_valid) return JV_OK;
    if (!codes->d_norms) JV_HIP_CHECK(hipMalloc((void **)&codes->d_norms, sizeof(float) * (size_t)std::max<int64_t>(codes->count, 1)));
    {
        ProfScope ps(ctx, R_NORMS);
        JV_TRY(launch_code_norms(ctx->stream, ctx, codes->pq->d_self_mag, codes->M, codes->d_codes, codes->count,
                                 codes->d_norms));
    }
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    codes->norms_valid = true;
    return JV_OK;
}

int ensure_vector_norms(jv_ctx *ctx, jv_vectors *v)
{
    std::lock_guard<std::mutex> lk(g_norms_mu);
    if (v->sqnorm_valid) return JV_OK;
    if (!v->d_sqnorm) JV_HIP_CHECK(hipMalloc((void **)&v->d_sqnorm, sizeof(float) * (size_t)std::max<int64_t>(v->count, 1)));
    {
        ProfScope ps(ctx, R_NORMS);
        JV_TRY(launch_row_sqnorms(ctx->stream, v->d_vecs, v->count, v->D, v->d_sqnorm));
    }
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    v->sqnorm_valid = true;
    return JV_OK;
}

int ensure_fused_norms(jv_ctx *ctx, jv_fused *f)
{
    std::lock_guard<std::mutex> lk(g_norms_mu);
    if (f->norms_valid) return JV_OK;
    const int64_t rows = f->count * f->maxDegree;
    if (!f->d_norms) JV_HIP_CHECK(hipMalloc((void **)&f->d_norms, sizeof(float) * (size_t)std::max<int64_t>(rows, 1)));
    JV_TRY(launch_code_norms(ctx->stream, ctx, f->pq->d_self_mag, f->M, f->d_blocks, rows, f->d_norms));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    f->norms_valid = true;
    return JV_OK;
}

}  // namespace jv

using namespace jv;

extern "C" {

const char *jv_hip_version(void) { return "jvector-hip 0.1 (gfx950)"; }
const char *jv_hip_last_error(void) { return g_err; }

int jv_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

const char *jv_hip_active_arch(int device)
{
    static thread_local char name[256];
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        (void)hipGetLastError();
        return "none";
    }
    snprintf(name, sizeof(name), "%s", prop.gcnArchName);
    return name;
}

int jv_hip_ctx_create(int device, void *stream, jv_ctx **out)
{
    clear_error();
    JV_REQUIRE(out != nullptr, "ctx_create: out is NULL");
    *out = nullptr;
    int n = jv_hip_device_count();
    if (n <= 0) {
        set_error("no HIP device visible (libjvector_hip has no CPU fallback)");
        return JV_ERR_NO_DEVICE;
    }
    JV_REQUIRE(device >= 0 && device < n, "ctx_create: device %d out of range [0,%d)", device, n);
    JV_TRY(use_device(device));
    hipDeviceProp_t prop;
    JV_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library contains gfx950 code objects only", device, prop.gcnArchName);
        return JV_ERR_NO_DEVICE;
    }
    jv_ctx *c = new jv_ctx();
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    c->lds_per_block = prop.maxSharedMemoryPerMultiProcessor ? (size_t)prop.maxSharedMemoryPerMultiProcessor
                                                             : (size_t)prop.sharedMemPerBlock;
    if (c->lds_per_block > 160 * 1024) c->lds_per_block = 160 * 1024;
    if (stream != JV_STREAM_PRIVATE) {
        c->stream = (hipStream_t)stream;  // NULL = the legacy default stream
        c->owns_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
            delete c;
            return JV_ERR_HIP;
        }
        c->owns_stream = true;
    }
    c->h_in.pinned_host = true;
    c->h_out.pinned_host = true;
    *out = c;
    return JV_OK;
}

int jv_hip_ctx_destroy(jv_ctx *ctx)
{
    if (!ctx) return JV_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->host_pool && ctx->host_pool_destroy) ctx->host_pool_destroy(ctx->host_pool);
    ctx->h_in.release();
    ctx->h_out.release();
    ctx->d_in.release();
    ctx->d_out.release();
    ctx->d_scratch.release();
    ctx->d_scratch2.release();
    ctx->d_scratch3.release();
    ctx->d_gs_visited.release();
    ctx->d_gs_spill.release();
    ctx->d_gs_out.release();
    ctx->d_gs_mask.release();
    ctx->d_gs_big.release();
    ctx->d_nvq_q.release();
    ctx->d_gs_extra.release();
    ctx->d_gs_ubr.release();
    ctx->d_bq_work.release();
    ctx->d_rd_counts.release();
    for (auto &e : ctx->prof_pending) {
        (void)hipEventDestroy(e.start);
        (void)hipEventDestroy(e.stop);
    }
    for (auto &e : ctx->prof_free) (void)hipEventDestroy(e);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return JV_OK;
}

int jv_hip_ctx_sync(jv_ctx *ctx)
{
    JV_REQUIRE(ctx, "ctx is NULL");
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return JV_OK;
}

void *jv_hip_ctx_stream(jv_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

static int prof_resolve(jv_ctx *ctx)
{
    if (ctx->prof_pending.empty()) return JV_OK;
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (auto &e : ctx->prof_pending) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, e.start, e.stop) == hipSuccess) {
            ctx->prof_ms[e.region] += ms;
            ctx->prof_count[e.region] += 1;
        } else {
            (void)hipGetLastError();
        }
        ctx->prof_free.push_back(e.start);
        ctx->prof_free.push_back(e.stop);
    }
    ctx->prof_pending.clear();
    return JV_OK;
}

int jv_hip_ctx_profile(jv_ctx *ctx, int enable)
{
    clear_error();
    JV_REQUIRE(ctx, "ctx is NULL");
    JV_TRY(use_device(ctx->device));
    JV_TRY(prof_resolve(ctx));
    for (int r = 0; r < R_COUNT; ++r) {
        ctx->prof_ms[r] = 0.0;
        ctx->prof_count[r] = 0;
    }
    ctx->profiling = enable != 0;
    return JV_OK;
}

int jv_hip_ctx_profile_read(jv_ctx *ctx, const char *region, double *total_ms, int64_t *count)
{
    clear_error();
    JV_REQUIRE(ctx && region, "NULL argument");
    static const char *names[R_COUNT] = {"adc", "topk", "exact", "lut", "encode", "norms", "sample", "gsearch", "prune", "adc_exact"};
    int r = -1;
    for (int i = 0; i < R_COUNT; ++i)
        if (strcmp(names[i], region) == 0) r = i;
    JV_REQUIRE(r >= 0, "unknown profile region '%s'", region);
    JV_TRY(use_device(ctx->device));
    JV_TRY(prof_resolve(ctx));
    if (total_ms) *total_ms = ctx->prof_ms[r];
    if (count) *count = ctx->prof_count[r];
    return JV_OK;
}

// ---- per-context options and counters ----
}  // extern "C"
namespace jv {
namespace {
// every option a context understands; the environment default of option x is JVECTOR_HIP_<X>
const char *const kOptions[] = {"graph_traversal", "gs_occ", "gs_pair", "gs_pairc", "gs_quad", "rd_table_free", "rd_chunk", "rd_split", "rd_wide_stage", "rd_prof", "rd_square", "bl_insert_alpha_x100", "bl_improve_beam", "bl_ref_order", "bl_sorted_lists", "gs_cand_cap", "gs_waves_per_cu", "gs_vcap_log2", "gs_v1_log2", "gs_prefetch", "gs_generic", "gs_wgx", "gs_wgx_waves", "gs_wgx_slots", "gs_wgx_depth", "gs_wgx_per_cu", "gs_wgx_lut_m", "gs_ubr", "gs_ubrc", "gs_ubr_trim", "gs_fused_rerank", "gs_defer", "gs_defer_min_level", "adc_bq",
                                "gs_grow", "gs_retry", "gs_prof", "gs_tie_check", "gs_push_log", "gs_push_log_cap", "graph_timing",
                                "no_filter", "quiet"};
std::string env_name(const char *name)
{
    std::string e = "JVECTOR_HIP_";
    for (const char *c = name; *c; ++c) e.push_back((char)toupper((unsigned char)*c));
    return e;
}
}  // namespace
bool ctx_opt_is_set(const jv_ctx *ctx, const char *name)
{
    if (ctx && ctx->opts.count(name)) return true;
    return getenv(env_name(name).c_str()) != nullptr;
}
long long ctx_opt(const jv_ctx *ctx, const char *name, long long dflt)
{
    if (ctx) {
        auto it = ctx->opts.find(name);
        if (it != ctx->opts.end()) return it->second;
    }
    const char *e = getenv(env_name(name).c_str());
    if (!e) return dflt;
    if (!strcmp(name, "graph_traversal")) return !strcmp(e, "host") ? JV_TRAVERSAL_HOST : (!strcmp(e, "device") ? JV_TRAVERSAL_DEVICE : dflt);
    return atoll(e);
}
}  // namespace jv
extern "C" {

int jv_hip_ctx_set_option(jv_ctx *ctx, const char *name, int64_t value)
{
    clear_error();
    JV_REQUIRE(ctx && name, "ctx_set_option: NULL argument");
    for (const char *k : kOptions)
        if (!strcmp(k, name)) {
            ctx->opts[name] = (long long)value;
            return JV_OK;
        }
    set_error("ctx_set_option: unknown option '%s'", name);
    return JV_ERR_INVALID;
}

int jv_hip_ctx_clear_option(jv_ctx *ctx, const char *name)
{
    clear_error();
    JV_REQUIRE(ctx && name, "ctx_clear_option: NULL argument");
    ctx->opts.erase(name);
    return JV_OK;
}

int jv_hip_ctx_get_stat(jv_ctx *ctx, const char *name, int64_t *out)
{
    clear_error();
    JV_REQUIRE(ctx && name && out, "ctx_get_stat: NULL argument");
    if (strcmp(name, "experimental_build") == 0) {   // 1: the library also holds the measured-and-switched-off kernel variants
#ifdef JV_EXPERIMENTAL
        *out = 1;
#else
        *out = 0;
#endif
        return JV_OK;
    }
    if (strcmp(name, "rd_tests") == 0 || strcmp(name, "rd_pairs") == 0) {   // the robust prune's device-side work counters
        *out = 0;
        if (ctx->d_rd_counts.ptr) {
            unsigned long long h[2] = {0, 0};
            JV_TRY(use_device(ctx->device));
            JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            JV_HIP_CHECK(hipMemcpy(h, ctx->d_rd_counts.ptr, sizeof(h), hipMemcpyDeviceToHost));
            *out = (int64_t)h[name[3] == 't' ? 0 : 1];
        }
        return JV_OK;
    }
    auto it = ctx->stats.find(name);
    *out = it == ctx->stats.end() ? 0 : (int64_t)it->second;
    return JV_OK;
}

int jv_hip_ctx_reset_stats(jv_ctx *ctx)
{
    clear_error();
    JV_REQUIRE(ctx, "ctx_reset_stats: NULL argument");
    ctx->stats.clear();
    return JV_OK;
}

// ------------------------------------------------------------------------------------------------
// ProductQuantization
// ------------------------------------------------------------------------------------------------
}  // extern "C"
namespace jv {
// jv_hip_pq_create with the choice of keeping a quantizer of fewer than 256 clusters UNPADDED: the training kernels (KmParams::k)
// work on such a k-row layout; everything else needs the padded form (pad = true, what the C entry point does)
int pq_create_impl(jv_ctx *ctx, int D, int M, int k, const int *sizes, const float *codebooks, const float *centroid, bool pad,
                   jv_pq **out);
}  // namespace jv
extern "C" {
int jv_hip_pq_create(jv_ctx *ctx, int D, int M, int k, const int *sizes, const float *codebooks, const float *centroid,
                     jv_pq **out)
{
    return jv::pq_create_impl(ctx, D, M, k, sizes, codebooks, centroid, true, out);
}
}  // extern "C"
namespace jv {
int pq_create_impl(jv_ctx *ctx, int D, int M, int k, const int *sizes, const float *codebooks, const float *centroid, bool pad,
                   jv_pq **out)
{
    clear_error();
    JV_REQUIRE(ctx && out && codebooks, "pq_create: NULL argument");
    *out = nullptr;
    JV_REQUIRE(D > 0 && M > 0, "pq_create: D and M must be positive");
    // ProductQuantization.getSubvectorSizesAndOffsets :536-538
    JV_REQUIRE(M <= D, "Number of subspaces must be less than or equal to the vector dimension");
    // ProductQuantization.checkClusterCount: 1..256 (codes are one byte).  Fewer than 256: padded below.
    if (k < 1 || k > kClusters) {
        set_error("pq_create: clusterCount %d outside 1..256 (codes are one byte)", k);
        return JV_ERR_UNSUPPORTED;
    }
    JV_TRY(use_device(ctx->device));
    const int k_user = k;
    std::vector<float> padded;
    if (k_user < kClusters && pad) {
        // rows k_user..255 of every codebook = copies of its row 0 (jv_pq::k_user): the device side only ever sees 256-row codebooks
        std::vector<int> sz((size_t)M);
        size_t total = 0;
        for (int m = 0; m < M; ++m) {
            sz[(size_t)m] = sizes ? sizes[m] : (D / M + (m < D % M ? 1 : 0));
            JV_REQUIRE(sz[(size_t)m] > 0, "pq_create: subvector size %d at m=%d", sz[(size_t)m], m);
            total += (size_t)sz[(size_t)m];
        }
        JV_REQUIRE(total == (size_t)D, "pq_create: subvector sizes sum to %zu, expected D=%d", total, D);
        std::vector<float> src((size_t)k_user * (size_t)D);
        JV_HIP_CHECK(hipMemcpy(src.data(), codebooks, sizeof(float) * src.size(), hipMemcpyDefault));
        padded.resize((size_t)kClusters * (size_t)D);
        size_t so = 0, po = 0;
        for (int m = 0; m < M; ++m) {
            const size_t S = (size_t)sz[(size_t)m];
            memcpy(padded.data() + po, src.data() + so, sizeof(float) * S * (size_t)k_user);
            for (int c = k_user; c < kClusters; ++c) memcpy(padded.data() + po + (size_t)c * S, src.data() + so, sizeof(float) * S);
            so += S * (size_t)k_user;
            po += S * (size_t)kClusters;
        }
        codebooks = padded.data();
        k = kClusters;
    }
    jv_pq *pq = new jv_pq();
    pq->device = ctx->device;
    pq->D = D;
    pq->M = M;
    pq->k = k;
    pq->k_user = k_user;
    pq->sizes.resize(M);
    pq->offsets.resize(M);
    pq->cb_offsets.resize(M);
    int off = 0;
    int64_t cbo = 0;
    for (int m = 0; m < M; ++m) {
        int size = sizes ? sizes[m] : (D / M + (m < D % M ? 1 : 0));
        if (size <= 0) {
            delete pq;
            set_error("pq_create: subvector size %d at m=%d", size, m);
            return JV_ERR_INVALID;
        }
        pq->sizes[m] = size;
        pq->offsets[m] = off;
        pq->cb_offsets[m] = cbo;
        off += size;
        cbo += (int64_t)k * size;
        pq->max_size = std::max(pq->max_size, size);
    }
    if (off != D) {
        delete pq;
        set_error("pq_create: subvector sizes sum to %d, expected D=%d", off, D);
        return JV_ERR_INVALID;
    }
    pq->uniform = std::all_of(pq->sizes.begin(), pq->sizes.end(), [&](int s) { return s == pq->sizes[0]; });

    auto fail = [&](int st) {
        jv_hip_pq_destroy(pq);
        return st;
    };
#define PQ_CHECK(expr)                                                                   \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            set_error("%s failed: %s", #expr, hipGetErrorString(_e));                    \
            (void)hipGetLastError();                                                     \
            return fail(_e == hipErrorOutOfMemory ? JV_ERR_OOM : JV_ERR_HIP);            \
        }                                                                                \
    } while (0)
    PQ_CHECK(hipMalloc((void **)&pq->d_sizes, sizeof(int) * M));
    PQ_CHECK(hipMalloc((void **)&pq->d_offsets, sizeof(int) * M));
    PQ_CHECK(hipMalloc((void **)&pq->d_cb_offsets, sizeof(int64_t) * M));
    PQ_CHECK(hipMalloc((void **)&pq->d_codebooks, sizeof(float) * (size_t)cbo));
    PQ_CHECK(hipMalloc((void **)&pq->d_self_mag, sizeof(float) * (size_t)M * k));
    if (pq->uniform && k % 2 == 0) PQ_CHECK(hipMalloc((void **)&pq->d_cb_paired, sizeof(float) * (size_t)cbo));
    PQ_CHECK(hipMemcpy(pq->d_sizes, pq->sizes.data(), sizeof(int) * M, hipMemcpyHostToDevice));
    PQ_CHECK(hipMemcpy(pq->d_offsets, pq->offsets.data(), sizeof(int) * M, hipMemcpyHostToDevice));
    PQ_CHECK(hipMemcpy(pq->d_cb_offsets, pq->cb_offsets.data(), sizeof(int64_t) * M, hipMemcpyHostToDevice));
    PQ_CHECK(hipMemcpy(pq->d_codebooks, codebooks, sizeof(float) * (size_t)cbo, hipMemcpyDefault));
    if (centroid) {
        PQ_CHECK(hipMalloc((void **)&pq->d_centroid, sizeof(float) * D));
        PQ_CHECK(hipMemcpy(pq->d_centroid, centroid, sizeof(float) * D, hipMemcpyDefault));
    }
#undef PQ_CHECK
    int st = launch_self_magnitudes(ctx->stream, pq);
    if (st != JV_OK) return fail(st);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) {
        set_error("pq_create: self-magnitude kernel failed");
        return fail(JV_ERR_HIP);
    }
    *out = pq;
    return JV_OK;
}
}  // namespace jv
extern "C" {

int jv_hip_pq_load(jv_ctx *ctx, const uint8_t *buf, size_t len, size_t *consumed, jv_pq **out)
{
    clear_error();
    JV_REQUIRE(ctx && buf && out, "pq_load: NULL argument");
    // One validator for the format: jv_fmt_pq_describe (formats.cpp pq_walk) bounds version <= 6, every sub-vector size
    // < 2^20, D < 2^24 (64-bit sum), k, and checks that every section lies inside the input; the walk below only copies.
    {
        size_t blk = 0;
        int ver = 0, dD = 0, dM = 0, dk = 0, hc = 0;
        float an = -1.0f;
        JV_TRY(jv_fmt_pq_describe(buf, len, &blk, &ver, &dD, &dM, &dk, &hc, &an));
    }
    // ProductQuantization.load :649-693
    size_t p = 0;
#define NEED(nb) JV_REQUIRE(p + (size_t)(nb) <= len, "pq_load: truncated input at byte %zu", p)
    NEED(4);
    int32_t maybeMagic = rd_i32(buf + p);
    p += 4;
    int version, gcl;
    if (maybeMagic != 0x75EC4012) {
        version = 0;
        gcl = maybeMagic;
    } else {
        NEED(8);
        version = rd_i32(buf + p);
        p += 4;
        gcl = rd_i32(buf + p);
        p += 4;
    }
    JV_REQUIRE(gcl >= 0 && gcl < (1 << 24), "pq_load: implausible centroid length %d", gcl);
    std::vector<float> centroid;
    if (gcl > 0) {
        NEED((size_t)gcl * 4);
        centroid.resize(gcl);
        for (int i = 0; i < gcl; ++i, p += 4) centroid[i] = rd_f32(buf + p);
    }
    NEED(4);
    int M = rd_i32(buf + p);
    p += 4;
    JV_REQUIRE(M > 0 && M < (1 << 20), "pq_load: implausible M %d", M);
    NEED((size_t)M * 4);
    std::vector<int> sizes(M);
    int D = 0;
    for (int i = 0; i < M; ++i, p += 4) {
        sizes[i] = rd_i32(buf + p);
        JV_REQUIRE(sizes[i] > 0, "pq_load: bad subvector size");
        D += sizes[i];
    }
    float aniso = -1.0f;
    if (version >= 3) {
        NEED(4);
        aniso = rd_f32(buf + p);
        p += 4;
    }
    JV_REQUIRE(aniso == aniso && aniso >= -1.0f && aniso < 1.0f, "Valid range for anisotropic threshold T is -1.0 <= t < 1.0");
    NEED(4);
    int k = rd_i32(buf + p);
    p += 4;
    JV_REQUIRE(k > 0 && k <= 65536, "pq_load: implausible cluster count %d", k);
    size_t total = (size_t)k * (size_t)D;
    NEED(total * 4);
    std::vector<float> cbs(total);
    for (size_t i = 0; i < total; ++i, p += 4) cbs[i] = rd_f32(buf + p);
#undef NEED
    JV_REQUIRE(gcl == 0 || gcl == D, "Global centroid length %d does not match vector dimensionality %d", gcl, D);
    if (consumed) *consumed = p;
    JV_REQUIRE(k <= kClusters, "pq_load: clusterCount %d > 256 (codes are one byte)", k);
    if (aniso > -1.0f && k != kClusters) {
        set_error("pq_load: an anisotropic PQ with clusterCount %d != 256 is not supported", k);
        return JV_ERR_UNSUPPORTED;
    }
    JV_TRY(jv_hip_pq_create(ctx, D, M, k, sizes.data(), cbs.data(), gcl ? centroid.data() : nullptr, out));
    (*out)->aniso = aniso;  // > -1: encode with encodeAnisotropic (ProductQuantization.encodeTo :439-449)
    return JV_OK;
}

int jv_hip_pq_set_anisotropic_threshold(jv_pq *pq, float threshold)
{
    clear_error();
    JV_REQUIRE(pq, "pq is NULL");
    // KMeansPlusPlusClusterer.java:87-92
    JV_REQUIRE(threshold == threshold && threshold >= -1.0f && threshold < 1.0f,
               "Valid range for anisotropic threshold T is -1.0 <= t < 1.0");
    if (threshold > -1.0f && pq->k_user != kClusters) {  // (the padded rows would take part in the coordinate descent)
        set_error("anisotropic encoding needs a 256-cluster PQ here (clusterCount %d)", pq->k_user);
        return JV_ERR_UNSUPPORTED;
    }
    pq->aniso = threshold;
    return JV_OK;
}

float jv_hip_pq_anisotropic_threshold(const jv_pq *pq) { return pq ? pq->aniso : -1.0f; }

int jv_hip_pq_destroy(jv_pq *pq)
{
    if (!pq) return JV_OK;
    (void)hipSetDevice(pq->device);
    (void)hipFree(pq->d_sizes);
    (void)hipFree(pq->d_offsets);
    (void)hipFree(pq->d_cb_offsets);
    (void)hipFree(pq->d_codebooks);
    (void)hipFree(pq->d_centroid);
    (void)hipFree(pq->d_self_mag);
    (void)hipFree(pq->d_cb_paired);
    delete pq;
    return JV_OK;
}

int jv_hip_pq_info(const jv_pq *pq, int *D, int *M, int *k, int *has_centroid)
{
    JV_REQUIRE(pq, "pq is NULL");
    if (D) *D = pq->D;
    if (M) *M = pq->M;
    if (k) *k = pq->k_user;   // the caller's clusterCount (the device side holds 256 rows per sub-space whatever it is)
    if (has_centroid) *has_centroid = pq->d_centroid != nullptr;
    return JV_OK;
}

int jv_hip_pq_self_magnitudes(jv_ctx *ctx, const jv_pq *pq, float *out)
{
    clear_error();
    JV_REQUIRE(ctx && pq && out, "NULL argument");
    JV_TRY(use_device(ctx->device));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    // [M][clusterCount]: rows of the caller's cluster count out of the 256-row device table
    if (pq->k_user == pq->k) {
        JV_HIP_CHECK(hipMemcpy(out, pq->d_self_mag, sizeof(float) * (size_t)pq->M * pq->k, hipMemcpyDefault));
    } else {
        for (int m = 0; m < pq->M; ++m)
            JV_HIP_CHECK(hipMemcpy(out + (size_t)m * pq->k_user, pq->d_self_mag + (size_t)m * pq->k, sizeof(float) * (size_t)pq->k_user,
                                   hipMemcpyDefault));
    }
    return JV_OK;
}

// ------------------------------------------------------------------------------------------------
// PQVectors code store
// ------------------------------------------------------------------------------------------------
int jv_hip_codes_create(jv_ctx *ctx, const jv_pq *pq, int64_t count, jv_codes **out)
{
    clear_error();
    JV_REQUIRE(ctx && pq && out, "codes_create: NULL argument");
    JV_REQUIRE(count > 0, "Invalid vector count %lld", (long long)count);  // PQLayout :516-518
    JV_TRY(use_device(ctx->device));
    jv_codes *c = new jv_codes();
    c->device = ctx->device;
    c->pq = pq;
    c->count = count;
    c->M = pq->M;
    c->owns = true;
    hipError_t e = hipMalloc((void **)&c->d_codes, (size_t)count * pq->M + 64);
    if (e != hipSuccess) {
        set_error("codes_create: hipMalloc(%lld x %d) failed: %s", (long long)count, pq->M, hipGetErrorString(e));
        (void)hipGetLastError();
        delete c;
        return JV_ERR_OOM;
    }
    *out = c;
    return JV_OK;
}

int jv_hip_codes_wrap(jv_ctx *ctx, const jv_pq *pq, int64_t count, void *device_codes, jv_codes **out)
{
    clear_error();
    JV_REQUIRE(ctx && pq && out && device_codes, "codes_wrap: NULL argument");
    JV_REQUIRE(count > 0, "Invalid vector count %lld", (long long)count);
    JV_REQUIRE(is_device_ptr(device_codes), "codes_wrap: pointer is not device memory");
    jv_codes *c = new jv_codes();
    c->device = ctx->device;
    c->pq = pq;
    c->count = count;
    c->M = pq->M;
    c->owns = false;
    c->d_codes = (uint8_t *)device_codes;
    *out = c;
    return JV_OK;
}

int jv_hip_codes_upload(jv_ctx *ctx, jv_codes *codes, int64_t first, int64_t count, const uint8_t *src)
{
    clear_error();
    JV_REQUIRE(ctx && codes && src, "codes_upload: NULL argument");
    JV_REQUIRE(first >= 0 && count >= 0 && first + count <= codes->count,
               "Ordinal range [%lld,%lld) out of bounds for vector count %lld", (long long)first,
               (long long)(first + count), (long long)codes->count);
    JV_TRY(use_device(ctx->device));
    JV_HIP_CHECK(hipMemcpyAsync(codes->d_codes + first * codes->M, src, (size_t)count * codes->M, hipMemcpyDefault,
                                ctx->stream));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    codes->norms_valid = false;
    return JV_OK;
}

int jv_hip_codes_download(jv_ctx *ctx, const jv_codes *codes, int64_t first, int64_t count, uint8_t *dst)
{
    clear_error();
    JV_REQUIRE(ctx && codes && dst, "codes_download: NULL argument");
    JV_REQUIRE(first >= 0 && count >= 0 && first + count <= codes->count, "Ordinal range out of bounds");
    JV_TRY(use_device(ctx->device));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    JV_HIP_CHECK(hipMemcpy(dst, codes->d_codes + first * codes->M, (size_t)count * codes->M, hipMemcpyDefault));
    return JV_OK;
}

int jv_hip_codes_destroy(jv_codes *c)
{
    if (!c) return JV_OK;
    (void)hipSetDevice(c->device);
    if (c->owns) (void)hipFree(c->d_codes);
    (void)hipFree(c->d_norms);
    delete c;
    return JV_OK;
}

int64_t jv_hip_codes_count(const jv_codes *c) { return c ? c->count : 0; }
void *jv_hip_codes_device_ptr(const jv_codes *c) { return c ? (void *)c->d_codes : nullptr; }

// ------------------------------------------------------------------------------------------------
// full-resolution vectors
// ------------------------------------------------------------------------------------------------
int jv_hip_vectors_create(jv_ctx *ctx, int64_t count, int D, jv_vectors **out)
{
    clear_error();
    JV_REQUIRE(ctx && out, "vectors_create: NULL argument");
    JV_REQUIRE(count > 0 && D > 0, "vectors_create: count and D must be positive");
    JV_TRY(use_device(ctx->device));
    jv_vectors *v = new jv_vectors();
    v->device = ctx->device;
    v->count = count;
    v->D = D;
    v->owns = true;
    hipError_t e = hipMalloc((void **)&v->d_vecs, sizeof(float) * (size_t)count * D);
    if (e != hipSuccess) {
        set_error("vectors_create: hipMalloc failed: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        delete v;
        return JV_ERR_OOM;
    }
    *out = v;
    return JV_OK;
}

int jv_hip_vectors_wrap(jv_ctx *ctx, int64_t count, int D, void *device_vectors, jv_vectors **out)
{
    clear_error();
    JV_REQUIRE(ctx && out && device_vectors, "vectors_wrap: NULL argument");
    JV_REQUIRE(count > 0 && D > 0, "vectors_wrap: count and D must be positive");
    JV_REQUIRE(is_device_ptr(device_vectors), "vectors_wrap: pointer is not device memory");
    jv_vectors *v = new jv_vectors();
    v->device = ctx->device;
    v->count = count;
    v->D = D;
    v->owns = false;
    v->d_vecs = (float *)device_vectors;
    *out = v;
    return JV_OK;
}

int jv_hip_vectors_invalidate(jv_vectors *v)
{
    clear_error();
    JV_REQUIRE(v, "vectors_invalidate: NULL argument");
    v->sqnorm_valid = false;  // the cosine rerank's per-row sum-of-squares table is rebuilt on next use
    return JV_OK;
}

int jv_hip_vectors_upload(jv_ctx *ctx, jv_vectors *v, int64_t first, int64_t count, const float *src)
{
    clear_error();
    JV_REQUIRE(ctx && v && src, "vectors_upload: NULL argument");
    JV_FLOAT_ROWS(v, "vectors_upload");
    JV_REQUIRE(first >= 0 && count >= 0 && first + count <= v->count, "vectors_upload: range out of bounds");
    JV_TRY(use_device(ctx->device));
    JV_HIP_CHECK(hipMemcpyAsync(v->d_vecs + first * v->D, src, sizeof(float) * (size_t)count * v->D, hipMemcpyDefault,
                                ctx->stream));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    v->sqnorm_valid = false;
    return JV_OK;
}

int jv_hip_device_alloc(jv_ctx *ctx, size_t bytes, void **out)
{
    clear_error();
    JV_REQUIRE(ctx && out, "device_alloc: NULL argument");
    JV_TRY(use_device(ctx->device));
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) {
        (void)hipGetLastError();
        set_error("device_alloc: hipMalloc of %zu bytes failed", bytes);
        return JV_ERR_OOM;
    }
    *out = p;
    return JV_OK;
}

int jv_hip_device_free(jv_ctx *ctx, void *ptr)
{
    clear_error();
    JV_REQUIRE(ctx, "device_free: NULL context");
    if (!ptr) return JV_OK;
    JV_TRY(use_device(ctx->device));
    JV_HIP_CHECK(hipFree(ptr));
    return JV_OK;
}

int jv_hip_vectors_destroy(jv_vectors *v)
{
    if (!v) return JV_OK;
    (void)hipSetDevice(v->device);
    if (v->owns) (void)hipFree(v->d_vecs);
    (void)hipFree(v->d_sqnorm);
    delete v;
    return JV_OK;
}

// ------------------------------------------------------------------------------------------------
// encode
// ------------------------------------------------------------------------------------------------
int jv_hip_pq_encode(jv_ctx *ctx, const jv_pq *pq, const float *vectors, int64_t count, uint8_t *codes_out)
{
    clear_error();
    JV_REQUIRE(ctx && pq, "pq_encode: NULL argument");
    JV_REQUIRE(count >= 0, "pq_encode: negative count");
    if (count == 0) return JV_OK;
    JV_REQUIRE(vectors && codes_out, "pq_encode: NULL buffer");
    JV_TRY(use_device(ctx->device));
    const void *d_v = nullptr;
    JV_TRY(stage_in(ctx, vectors, sizeof(float) * (size_t)count * pq->D, ctx->h_in, ctx->d_in, &d_v));
    OutStage os;
    JV_TRY(stage_out_begin(ctx, codes_out, (size_t)count * pq->M, ctx->d_out, &os));
    {
        ProfScope ps(ctx, R_ENCODE);
        JV_TRY((pq->aniso > -1.0f ? launch_pq_encode_anisotropic : launch_pq_encode)(ctx->stream, pq, (const float *)d_v, count,
                                                                                   (uint8_t *)os.dev));
    }
    return stage_out_end(ctx, os);
}

int jv_hip_pq_encode_into(jv_ctx *ctx, const jv_pq *pq, const jv_vectors *v, int64_t first, int64_t count,
                          jv_codes *codes)
{
    clear_error();
    JV_REQUIRE(ctx && pq && v && codes, "pq_encode_into: NULL argument");
    JV_FLOAT_ROWS(v, "pq_encode_into");
    JV_REQUIRE(v->D == pq->D, "vector dimension %d does not match PQ dimension %d", v->D, pq->D);
    JV_REQUIRE(codes->M == pq->M, "code store M %d does not match PQ M %d", codes->M, pq->M);
    JV_REQUIRE(first >= 0 && count >= 0 && first + count <= v->count && first + count <= codes->count,
               "pq_encode_into: range out of bounds");
    JV_TRY(use_device(ctx->device));
    {
        ProfScope ps(ctx, R_ENCODE);
        JV_TRY((pq->aniso > -1.0f ? launch_pq_encode_anisotropic : launch_pq_encode)(ctx->stream, pq, v->d_vecs + first * v->D, count,
                                                                                   codes->d_codes + first * codes->M));
    }
    codes->norms_valid = false;
    return JV_OK;
}

// ------------------------------------------------------------------------------------------------
// LUTs
// ------------------------------------------------------------------------------------------------
int jv_hip_luts_create(jv_ctx *ctx, const jv_pq *pq, int max_queries, jv_luts **out)
{
    clear_error();
    JV_REQUIRE(ctx && pq && out, "luts_create: NULL argument");
    JV_REQUIRE(max_queries > 0, "luts_create: max_queries must be positive");
    JV_TRY(use_device(ctx->device));
    jv_luts *l = new jv_luts();
    l->device = ctx->device;
    l->pq = pq;
    l->capacity = max_queries;
    auto fail = [&](hipError_t e) {
        set_error("luts_create: hipMalloc failed: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        jv_hip_luts_destroy(l);
        return JV_ERR_OOM;
    };
    hipError_t e;
    if ((e = hipMalloc((void **)&l->d_luts, sizeof(float) * (size_t)max_queries * pq->M * kClusters)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void **)&l->d_bmag, sizeof(float) * (size_t)max_queries)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void **)&l->d_queries, sizeof(float) * (size_t)max_queries * pq->D)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void **)&l->d_raw_queries, sizeof(float) * (size_t)max_queries * pq->D)) != hipSuccess) return fail(e);
    *out = l;
    return JV_OK;
}

int jv_hip_luts_destroy(jv_luts *l)
{
    if (!l) return JV_OK;
    (void)hipSetDevice(l->device);
    (void)hipFree(l->d_luts);
    (void)hipFree(l->d_bmag);
    (void)hipFree(l->d_queries);
    (void)hipFree(l->d_raw_queries);
    delete l;
    return JV_OK;
}

}  // extern "C"

namespace jv {
int luts_prepare(jv_ctx *ctx, jv_luts *l, const float *queries, int Q, jv_vsf vsf, jv_decoder_kind kind, bool with_tables)
{
    JV_REQUIRE(ctx && l, "luts_build: NULL argument");
    JV_REQUIRE(Q >= 0 && Q <= l->capacity, "luts_build: Q=%d exceeds capacity %d", Q, l->capacity);
    JV_REQUIRE(vsf == JV_EUCLIDEAN || vsf == JV_DOT_PRODUCT || vsf == JV_COSINE, "Unsupported similarity function %d",
               (int)vsf);
    l->Q = Q;
    l->vsf = vsf;
    l->kind = kind;
    l->tables_valid = false;
    if (Q == 0) return JV_OK;
    JV_REQUIRE(queries, "luts_build: queries is NULL");
    JV_TRY(use_device(ctx->device));
    const jv_pq *pq = l->pq;
    const size_t qbytes = sizeof(float) * (size_t)Q * pq->D;
    if (is_device_ptr(queries)) {
        JV_HIP_CHECK(hipMemcpyAsync(l->d_raw_queries, queries, qbytes, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        JV_TRY(ctx->h_in.reserve(qbytes));
        memcpy(ctx->h_in.ptr, queries, qbytes);
        JV_HIP_CHECK(hipMemcpyAsync(l->d_raw_queries, ctx->h_in.ptr, qbytes, hipMemcpyHostToDevice, ctx->stream));
    }
    ProfScope ps(ctx, R_LUT);
    JV_TRY(launch_center_queries(ctx->stream, pq, l->d_raw_queries, Q, l->d_queries));
    if (with_tables) {
        // cosine numerator uses the DOT_PRODUCT partial sums (PQDecoder.java:117, FusedPQDecoder.java:187)
        const int lut_vsf = (vsf == JV_EUCLIDEAN) ? VSF_L2 : VSF_DOT;
        JV_TRY(launch_lut_build(ctx->stream, pq, l->d_queries, Q, lut_vsf, l->d_luts));
        l->tables_valid = true;
    }
    if (vsf == JV_COSINE)
        JV_TRY(launch_query_magnitudes(ctx->stream, pq, l->d_queries, Q, kind == JV_DECODER_FUSED ? 1 : 0, l->d_bmag));
    return JV_OK;
}
}  // namespace jv

extern "C" {

int jv_hip_luts_build(jv_ctx *ctx, jv_luts *l, const float *queries, int Q, jv_vsf vsf, jv_decoder_kind kind)
{
    clear_error();
    return luts_prepare(ctx, l, queries, Q, vsf, kind, true);
}

int jv_hip_luts_download(jv_ctx *ctx, const jv_luts *l, int q, float *lut_out, float *bmag_out)
{
    clear_error();
    JV_REQUIRE(ctx && l, "luts_download: NULL argument");
    JV_REQUIRE(q >= 0 && q < l->Q, "luts_download: query %d out of range", q);
    JV_REQUIRE(l->tables_valid || !lut_out, "luts_download: the tables were overwritten by a table-free graph search; call jv_hip_luts_build again");
    JV_TRY(use_device(ctx->device));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const size_t n = (size_t)l->pq->M * kClusters;
    if (lut_out) JV_HIP_CHECK(hipMemcpy(lut_out, l->d_luts + (size_t)q * n, sizeof(float) * n, hipMemcpyDefault));
    if (bmag_out) {
        if (l->vsf == JV_COSINE) JV_HIP_CHECK(hipMemcpy(bmag_out, l->d_bmag + q, sizeof(float), hipMemcpyDefault));
        else *bmag_out = 0.0f;
    }
    return JV_OK;
}

// The upper-bound tables the register-table traversal (gs_body.h "UBR") would load for the queries of `l`: built by the dense
// kernel of k_gsearch_ubr.hip from the centred queries luts_build staged.  An accessor for tests and studies.
int jv_hip_luts_bound_tables(jv_ctx *ctx, const jv_luts *l, uint32_t *tab_out, float *meta_out)
{
    clear_error();
    JV_REQUIRE(ctx && l && tab_out && meta_out, "luts_bound_tables: NULL argument");
    JV_REQUIRE(l->Q > 0, "luts_bound_tables: no queries staged (call jv_hip_luts_build first)");
    const jv_pq *pq = l->pq;
    JV_REQUIRE(pq->uniform && pq->max_size == 8 && pq->k == kClusters && pq->M % 8 == 0,
               "luts_bound_tables: 256 clusters, uniform 8-dim sub-vectors, M a multiple of 8");
    JV_TRY(use_device(ctx->device));
    const size_t tab_bytes = (gs_ubr_tab_bytes(pq->M) * (size_t)l->Q + 255) & ~(size_t)255;
    JV_TRY(ctx->d_gs_ubr.reserve(tab_bytes + sizeof(float) * 4 * (size_t)l->Q));
    uint32_t *d_tab = (uint32_t *)ctx->d_gs_ubr.ptr;
    float *d_meta = (float *)((char *)ctx->d_gs_ubr.ptr + tab_bytes);
    JV_TRY(launch_ubr_tables(ctx->stream, l->vsf == JV_EUCLIDEAN ? VSF_L2 : (l->vsf == JV_DOT_PRODUCT ? VSF_DOT : VSF_COS), pq->d_codebooks, l->d_queries, l->Q, pq->M, d_tab, d_meta));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    JV_HIP_CHECK(hipMemcpy(tab_out, d_tab, gs_ubr_tab_bytes(pq->M) * (size_t)l->Q, hipMemcpyDefault));
    JV_HIP_CHECK(hipMemcpy(meta_out, d_meta, sizeof(float) * 4 * (size_t)l->Q, hipMemcpyDefault));
    return JV_OK;
}

// ------------------------------------------------------------------------------------------------
// ADC
// ------------------------------------------------------------------------------------------------
int jv_hip_adc_scan(jv_ctx *ctx, const jv_luts *l, const jv_codes *codes, int64_t first, int64_t count,
                    float *scores_out)
{
    clear_error();
    JV_REQUIRE(ctx && l && codes, "adc_scan: NULL argument");
    JV_REQUIRE(codes->pq == l->pq, "adc_scan: LUTs and codes belong to different ProductQuantizations");
    JV_REQUIRE(l->tables_valid || l->Q == 0, "adc_scan: no tables for the current queries; call jv_hip_luts_build first");
    JV_REQUIRE(first >= 0 && count >= 0 && first + count <= codes->count,
               "Ordinal range [%lld,%lld) out of bounds for vector count %lld", (long long)first,
               (long long)(first + count), (long long)codes->count);
    if (l->Q == 0 || count == 0) return JV_OK;
    JV_REQUIRE(scores_out, "adc_scan: scores_out is NULL");
    JV_TRY(use_device(ctx->device));
    if (l->vsf == JV_COSINE) JV_TRY(ensure_code_norms(ctx, const_cast<jv_codes *>(codes)));
    OutStage os;
    JV_TRY(stage_out_begin(ctx, scores_out, sizeof(float) * (size_t)l->Q * count, ctx->d_out, &os));
    {
        ProfScope ps(ctx, R_ADC);
        // >= 2 queries over the same range: the multi-query kernel (4 queries per LDS gather); else single-query
        if (l->Q >= 2 && adc_mq_supported(codes->M, codes->d_codes))
            JV_TRY(launch_adc_mq_store(ctx->stream, ctx, l->d_luts, l->d_bmag, l->Q, codes->M, to_kernel_vsf(l->vsf),
                                       codes->d_codes, codes->d_norms, first, count, 1, (float *)os.dev));
        else
            JV_TRY(launch_adc(ctx->stream, ctx, l->d_luts, l->d_bmag, l->Q, codes->M, to_kernel_vsf(l->vsf),
                              codes->d_codes, codes->d_norms, codes->count, first, count, nullptr, (float *)os.dev));
    }
    return stage_out_end(ctx, os);
}

int jv_hip_adc_scores(jv_ctx *ctx, const jv_luts *l, const jv_codes *codes, const int32_t *ordinals, int B,
                      float *scores_out)
{
    clear_error();
    JV_REQUIRE(ctx && l && codes, "adc_scores: NULL argument");
    JV_REQUIRE(codes->pq == l->pq, "adc_scores: LUTs and codes belong to different ProductQuantizations");
    JV_REQUIRE(l->tables_valid || l->Q == 0, "adc_scores: no tables for the current queries; call jv_hip_luts_build first");
    JV_REQUIRE(B >= 0, "adc_scores: negative batch");
    if (l->Q == 0 || B == 0) return JV_OK;
    JV_REQUIRE(ordinals && scores_out, "adc_scores: NULL buffer");
    JV_TRY(use_device(ctx->device));
    if (l->vsf == JV_COSINE) JV_TRY(ensure_code_norms(ctx, const_cast<jv_codes *>(codes)));
    const void *d_ord = nullptr;
    JV_TRY(stage_in(ctx, ordinals, sizeof(int32_t) * (size_t)l->Q * B, ctx->h_in, ctx->d_in, &d_ord));
    OutStage os;
    JV_TRY(stage_out_begin(ctx, scores_out, sizeof(float) * (size_t)l->Q * B, ctx->d_out, &os));
    {
        ProfScope ps(ctx, R_ADC);
        JV_TRY(launch_adc(ctx->stream, ctx, l->d_luts, l->d_bmag, l->Q, codes->M, to_kernel_vsf(l->vsf), codes->d_codes,
                          codes->d_norms, codes->count, 0, B, (const int32_t *)d_ord, (float *)os.dev));
    }
    return stage_out_end(ctx, os);
}

// ------------------------------------------------------------------------------------------------
// Fused PQ
// ------------------------------------------------------------------------------------------------
int jv_hip_fused_create(jv_ctx *ctx, const jv_pq *pq, int64_t count, int maxDegree, jv_fused **out)
{
    clear_error();
    JV_REQUIRE(ctx && pq && out, "fused_create: NULL argument");
    JV_REQUIRE(count > 0 && maxDegree > 0 && maxDegree < 2048, "fused_create: bad count/maxDegree");
    // FusedPQ requires a 256-cluster PQ (FusedPQ.java:57-59)
    JV_REQUIRE(pq->k_user == kClusters, "FusedPQ requires a 256-cluster PQ");
    JV_TRY(use_device(ctx->device));
    jv_fused *f = new jv_fused();
    f->device = ctx->device;
    f->pq = pq;
    f->count = count;
    f->maxDegree = maxDegree;
    f->M = pq->M;
    hipError_t e1 = hipMalloc((void **)&f->d_blocks, (size_t)count * maxDegree * pq->M + 64);
    hipError_t e2 = hipMalloc((void **)&f->d_neighbors, sizeof(int32_t) * (size_t)count * maxDegree);
    if (e1 != hipSuccess || e2 != hipSuccess) {
        set_error("fused_create: hipMalloc failed");
        (void)hipGetLastError();
        jv_hip_fused_destroy(f);
        return JV_ERR_OOM;
    }
    *out = f;
    return JV_OK;
}

int jv_hip_fused_upload(jv_ctx *ctx, jv_fused *f, int64_t first, int64_t count, const uint8_t *blocks,
                        const int32_t *neighbors)
{
    clear_error();
    JV_REQUIRE(ctx && f && blocks && neighbors, "fused_upload: NULL argument");
    JV_REQUIRE(first >= 0 && count >= 0 && first + count <= f->count, "fused_upload: range out of bounds");
    JV_TRY(use_device(ctx->device));
    const size_t bsz = (size_t)f->maxDegree * f->M;
    JV_HIP_CHECK(hipMemcpyAsync(f->d_blocks + first * bsz, blocks, (size_t)count * bsz, hipMemcpyDefault, ctx->stream));
    JV_HIP_CHECK(hipMemcpyAsync(f->d_neighbors + first * f->maxDegree, neighbors,
                                sizeof(int32_t) * (size_t)count * f->maxDegree, hipMemcpyDefault, ctx->stream));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    f->norms_valid = false;
    f->generation = next_fused_generation();  // process-wide, never reused: a recycled jv_fused address cannot alias a cached check
    return JV_OK;
}

int jv_hip_fused_destroy(jv_fused *f)
{
    if (!f) return JV_OK;
    (void)hipSetDevice(f->device);
    (void)hipFree(f->d_blocks);
    (void)hipFree(f->d_neighbors);
    (void)hipFree(f->d_norms);
    delete f;
    return JV_OK;
}

int jv_hip_fused_scores(jv_ctx *ctx, const jv_luts *l, const jv_fused *f, const int32_t *origins, float *scores_out,
                        int32_t *neighbors_out)
{
    clear_error();
    JV_REQUIRE(ctx && l && f, "fused_scores: NULL argument");
    JV_REQUIRE(f->pq == l->pq, "fused_scores: LUTs and fused blocks belong to different ProductQuantizations");
    JV_REQUIRE(l->tables_valid || l->Q == 0, "fused_scores: no tables for the current queries; call jv_hip_luts_build first");
    if (l->Q == 0) return JV_OK;
    JV_REQUIRE(origins && scores_out, "fused_scores: NULL buffer");
    JV_TRY(use_device(ctx->device));
    if (l->vsf == JV_COSINE) JV_TRY(ensure_fused_norms(ctx, const_cast<jv_fused *>(f)));
    const void *d_org = nullptr;
    JV_TRY(stage_in(ctx, origins, sizeof(int32_t) * (size_t)l->Q, ctx->h_in, ctx->d_in, &d_org));
    OutStage os, ns;
    const size_t cells = (size_t)l->Q * f->maxDegree;
    JV_TRY(stage_out_begin(ctx, scores_out, sizeof(float) * cells, ctx->d_out, &os));
    if (neighbors_out) JV_TRY(stage_out_begin(ctx, neighbors_out, sizeof(int32_t) * cells, ctx->d_scratch, &ns));
    {
        ProfScope ps(ctx, R_ADC);
        JV_TRY(launch_fused(ctx->stream, ctx, l->d_luts, l->d_bmag, l->Q, f->M, to_kernel_vsf(l->vsf), f->d_blocks,
                            f->d_neighbors, f->d_norms, f->maxDegree, f->count, (const int32_t *)d_org, (float *)os.dev,
                            neighbors_out ? (int32_t *)ns.dev : nullptr));
    }
    JV_TRY(stage_out_end(ctx, os));
    if (neighbors_out) JV_TRY(stage_out_end(ctx, ns));
    return JV_OK;
}

// ------------------------------------------------------------------------------------------------
// exact scoring
// ------------------------------------------------------------------------------------------------
int jv_hip_exact_scores(jv_ctx *ctx, const jv_vectors *v, const float *queries, int Q, jv_vsf vsf,
                        const int32_t *ordinals, int B, float *scores_out)
{
    clear_error();
    JV_REQUIRE(ctx && v, "exact_scores: NULL argument");
    JV_REQUIRE(Q >= 0 && B >= 0, "exact_scores: negative sizes");
    if (Q == 0 || B == 0) return JV_OK;
    JV_REQUIRE(queries && ordinals && scores_out, "exact_scores: NULL buffer");
    JV_TRY(use_device(ctx->device));
    // two host inputs may both need staging: use separate device buffers, one pinned staging pass each
    const void *d_q = nullptr, *d_ord = nullptr;
    JV_TRY(stage_in(ctx, queries, sizeof(float) * (size_t)Q * v->D, ctx->h_in, ctx->d_in, &d_q));
    JV_TRY(stage_in(ctx, ordinals, sizeof(int32_t) * (size_t)Q * B, ctx->h_in, ctx->d_scratch2, &d_ord));
    JV_TRY(ctx->d_scratch3.reserve(sizeof(float) * (size_t)Q));
    OutStage os;
    JV_TRY(stage_out_begin(ctx, scores_out, sizeof(float) * (size_t)Q * B, ctx->d_out, &os));
    JV_TRY(rerank_gather(ctx, v, (const float *)d_q, Q, vsf, (const int32_t *)d_ord, B, (float *)os.dev, (float *)ctx->d_scratch3.ptr));
    return stage_out_end(ctx, os);
}

int jv_hip_exact_pair_scores(jv_ctx *ctx, const jv_vectors *v, jv_vsf vsf, const int32_t *node1, int P, const int32_t *node2, int B,
                             float *scores_out)
{
    clear_error();
    JV_REQUIRE(ctx && v, "exact_pair_scores: NULL argument");
    JV_FLOAT_ROWS(v, "exact_pair_scores");
    JV_REQUIRE(P >= 0 && B >= 0, "exact_pair_scores: negative sizes");
    if (P == 0 || B == 0) return JV_OK;
    JV_REQUIRE(node1 && node2 && scores_out, "exact_pair_scores: NULL buffer");
    JV_TRY(use_device(ctx->device));
    // node2 is copied: the gather step voids the lists of rows whose node1 is not a valid ordinal
    const void *d_n1 = nullptr;
    JV_TRY(stage_in(ctx, node1, sizeof(int32_t) * (size_t)P, ctx->h_in, ctx->d_in, &d_n1));
    JV_TRY(ctx->d_scratch2.reserve(sizeof(int32_t) * (size_t)P * B));
    JV_HIP_CHECK(hipMemcpyAsync(ctx->d_scratch2.ptr, node2, sizeof(int32_t) * (size_t)P * B, hipMemcpyDefault, ctx->stream));
    if (!is_device_ptr(node2)) JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // pageable source: the copy must not outlive the call
    JV_TRY(ctx->d_gs_out.reserve(sizeof(float) * (size_t)P * v->D));
    JV_TRY(ctx->d_scratch3.reserve(sizeof(float) * (size_t)P));
    OutStage os;
    JV_TRY(stage_out_begin(ctx, scores_out, sizeof(float) * (size_t)P * B, ctx->d_out, &os));
    if (vsf == JV_COSINE) JV_TRY(ensure_vector_norms(ctx, const_cast<jv_vectors *>(v)));
    {
        ProfScope ps(ctx, R_EXACT);
        JV_TRY(launch_gather_rows(ctx->stream, v->d_vecs, v->count, v->D, (const int32_t *)d_n1, P, (float *)ctx->d_gs_out.ptr,
                                  (int32_t *)ctx->d_scratch2.ptr, B));
        JV_TRY(launch_exact_gather(ctx->stream, v->d_vecs, v->count, v->D, (const float *)ctx->d_gs_out.ptr, P, to_kernel_vsf(vsf),
                                   (const int32_t *)ctx->d_scratch2.ptr, B, (float *)os.dev, (float *)ctx->d_scratch3.ptr, v->d_sqnorm));
    }
    return stage_out_end(ctx, os);
}

int jv_hip_exact_scan(jv_ctx *ctx, const jv_vectors *v, const float *queries, int Q, jv_vsf vsf, int64_t first,
                      int64_t count, float *scores_out)
{
    clear_error();
    JV_REQUIRE(ctx && v, "exact_scan: NULL argument");
    JV_FLOAT_ROWS(v, "exact_scan");
    JV_REQUIRE(first >= 0 && count >= 0 && first + count <= v->count, "exact_scan: range out of bounds");
    if (Q == 0 || count == 0) return JV_OK;
    JV_REQUIRE(queries && scores_out, "exact_scan: NULL buffer");
    JV_TRY(use_device(ctx->device));
    const void *d_q = nullptr;
    JV_TRY(stage_in(ctx, queries, sizeof(float) * (size_t)Q * v->D, ctx->h_in, ctx->d_in, &d_q));
    JV_TRY(ctx->d_scratch3.reserve(sizeof(float) * (size_t)Q));
    OutStage os;
    JV_TRY(stage_out_begin(ctx, scores_out, sizeof(float) * (size_t)Q * count, ctx->d_out, &os));
    {
        ProfScope ps(ctx, R_EXACT);
        JV_TRY(launch_exact_scan(ctx->stream, ctx, v->d_vecs, v->D, (const float *)d_q, Q, to_kernel_vsf(vsf), first,
                                 count, (float *)os.dev, (float *)ctx->d_scratch3.ptr));
    }
    return stage_out_end(ctx, os);
}

int jv_hip_exact_scan_dense(jv_ctx *ctx, const jv_vectors *v, const float *queries, int Q, jv_vsf vsf, int64_t first,
                            int64_t count, float *scores_out)
{
    clear_error();
    JV_REQUIRE(ctx && v, "exact_scan_dense: NULL argument");
    JV_FLOAT_ROWS(v, "exact_scan_dense");
    JV_REQUIRE(Q >= 0, "exact_scan_dense: negative query count");
    JV_REQUIRE(first >= 0 && count >= 0 && first + count <= v->count, "exact_scan_dense: range out of bounds");
    if (Q == 0 || count == 0) return JV_OK;
    JV_REQUIRE(queries && scores_out, "exact_scan_dense: NULL buffer");
    JV_TRY(use_device(ctx->device));
    const void *d_q = nullptr;
    JV_TRY(stage_in(ctx, queries, sizeof(float) * (size_t)Q * v->D, ctx->h_in, ctx->d_in, &d_q));
    OutStage os;
    JV_TRY(stage_out_begin(ctx, scores_out, sizeof(float) * (size_t)Q * count, ctx->d_out, &os));
    {
        ProfScope ps(ctx, R_EXACT);
        JV_TRY(launch_exact_scan_dense(ctx->stream, v->d_vecs, v->D, (const float *)d_q, Q, to_kernel_vsf(vsf), first, count,
                                       (float *)os.dev));
    }
    return stage_out_end(ctx, os);
}

// ------------------------------------------------------------------------------------------------
// top-k
// ------------------------------------------------------------------------------------------------
int jv_hip_topk(jv_ctx *ctx, const float *scores, const int32_t *ids, int Q, int64_t n, int64_t stride,
                int32_t id_base, int k, int32_t *out_ids, float *out_scores)
{
    clear_error();
    JV_REQUIRE(ctx, "topk: ctx is NULL");
    JV_REQUIRE(Q >= 0 && n >= 0 && k >= 0 && stride >= n, "topk: bad sizes");
    if (Q == 0 || k == 0) return JV_OK;
    JV_REQUIRE(scores && out_ids && out_scores, "topk: NULL buffer");
    JV_TRY(use_device(ctx->device));
    const void *d_scores = nullptr, *d_ids = nullptr;
    const size_t cells = (size_t)(Q - 1) * stride + n;
    JV_TRY(stage_in(ctx, scores, sizeof(float) * cells, ctx->h_in, ctx->d_in, &d_scores));
    if (ids) JV_TRY(stage_in(ctx, ids, sizeof(int32_t) * cells, ctx->h_in, ctx->d_scratch2, &d_ids));
    JV_TRY(ctx->d_scratch.reserve(topk_scratch_bytes(Q, k)));
    OutStage oi, osc;
    JV_TRY(stage_out_begin(ctx, out_ids, sizeof(int32_t) * (size_t)Q * k, ctx->d_out, &oi));
    JV_TRY(stage_out_begin(ctx, out_scores, sizeof(float) * (size_t)Q * k, ctx->d_scratch3, &osc));
    {
        ProfScope ps(ctx, R_TOPK);
        JV_TRY(launch_topk(ctx->stream, ctx, (const float *)d_scores, (const int32_t *)d_ids, Q, n, stride, id_base, k,
                           (int32_t *)oi.dev, (float *)osc.dev, ctx->d_scratch.ptr));
    }
    JV_TRY(stage_out_end(ctx, oi));
    return stage_out_end(ctx, osc);
}

// ------------------------------------------------------------------------------------------------
// flat two-pass search
//
// Pass 1 (approximate): two strategies producing the SAME exact top-k1 of the ADC scores.
//   * materialised: scan writes all Q x N scores, radix-select top-k1 reads them back (any shape);
//   * threshold-filtered (multi-query kernel shapes, large N): a strided sample of S candidates is scored
//     first; its k_s-th best score tau_q bounds the k1-th best of the full set from below with overwhelming
//     probability (the expected number of candidates >= tau_q is c_target >= 8*k1); the full scan then appends
//     only candidates with score >= tau_q to a small per-query list and top-k1 runs over that list.  The set
//     {score >= tau} contains the true top-k1 whenever it has >= k1 members, so the result is exact; the host
//     checks the per-query counts (k1 <= count <= capacity) and falls back to the materialised strategy otherwise.
// Pass 2 (exact): gather-score the k1 candidates at full resolution, top-K under the NodeQueue order.
// ------------------------------------------------------------------------------------------------
static int next_pow2_i64(int64_t v)
{
    int64_t p = 1;
    while (p < v) p <<= 1;
    return (int)p;
}

int jv_hip_search_flat(jv_ctx *ctx, jv_luts *l, const jv_codes *codes, const jv_vectors *vectors, const float *queries,
                       int Q, jv_vsf vsf, int topK, int rerankK, int32_t id_base, int32_t *out_ids, float *out_scores)
{
    clear_error();
    JV_REQUIRE(ctx && l && codes, "search_flat: NULL argument");
    JV_REQUIRE(topK > 0, "search_flat: topK must be positive");
    const bool rerank = vectors != nullptr && rerankK > 0;
    // GraphSearcher.search :233 — rerankK must be >= topK
    JV_REQUIRE(!rerank || rerankK >= topK, "rerankK %d must be >= topK %d", rerankK, topK);
    JV_REQUIRE(!rerank || vectors->D == l->pq->D, "search_flat: vector dimension mismatch");
    JV_REQUIRE(!rerank || vectors->count >= codes->count, "search_flat: fewer vectors than codes");
    if (Q == 0) return JV_OK;
    JV_REQUIRE(out_ids && out_scores, "search_flat: NULL output");
    JV_TRY(use_device(ctx->device));
    JV_TRY(jv_hip_luts_build(ctx, l, queries, Q, vsf, JV_DECODER_PQ));
    if (vsf == JV_COSINE) JV_TRY(ensure_code_norms(ctx, const_cast<jv_codes *>(codes)));
    const int64_t N = codes->count;
    const int kvsf = to_kernel_vsf(vsf);
    const int k1 = rerank ? rerankK : topK;

    OutStage oi, osc;
    JV_TRY(stage_out_begin(ctx, out_ids, sizeof(int32_t) * (size_t)Q * topK, ctx->d_out, &oi));
    JV_TRY(stage_out_begin(ctx, out_scores, sizeof(float) * (size_t)Q * topK, ctx->d_in, &osc));

    // scratch3: [cand ids Q*k1][cand approx scores Q*k1][exact scores Q*k1][qnorm Q]
    const size_t c1 = (size_t)Q * k1;
    JV_TRY(ctx->d_scratch3.reserve(sizeof(int32_t) * c1 + sizeof(float) * c1 * 2 + sizeof(float) * (size_t)Q + 1024));
    int32_t *d_cand = (int32_t *)ctx->d_scratch3.ptr;
    float *d_cand_sc = (float *)(d_cand + c1);
    float *d_exact = d_cand_sc + c1;
    float *d_qnorm = d_exact + c1;
    int32_t *d_k1_ids = rerank ? d_cand : (int32_t *)oi.dev;
    float *d_k1_sc = rerank ? d_cand_sc : (float *)osc.dev;

    // ---------------- pass 1 ----------------
    bool done1 = false;
    const bool mq = adc_mq_supported(codes->M, codes->d_codes) && Q >= 2;
    if (mq && N >= (1 << 18) && (int64_t)k1 * 64 <= N && ctx_opt(ctx, "no_filter", 0) == 0) {
        // expected candidates per query behind the threshold.  The threshold is the k_s-th best of a strided sample (k_s = c_target x
        // sample / N: 43 at C4's shard, 157 at C2), so the count it yields is Gamma(k_s)-distributed, ~1 / sqrt(k_s) = 8 - 15 % around
        // c_target: with c_target = 3 x rerankK the `>= k1` check sits 4.4 (C4) to 8 (C2) standard deviations away — a query fails it
        // about once in 2 x 10^5 at C4, i.e. about one CALL in 800 at Q = 256 pays the unfiltered scan (results are the same).
        // (Round 4 used 8 x: 2.7 x the lists, the gather and the top-k input for nothing — at C2's rerankK 3200 that was 0.6 ms of a
        // 3 ms step.)
        const int64_t c_target = std::max<int64_t>(3 * (int64_t)k1, 4096);
        int64_t S_target = std::max<int64_t>(16384, next_pow2_i64(32 * N / c_target));
        const int64_t stride = std::max<int64_t>(1, N / S_target);
        const int64_t S = N / stride;  // sampled rows: 0, stride, 2*stride, ...
        const int k_s = (int)std::max<int64_t>(16, (c_target * S + N - 1) / N);
        const int cap = (int)std::min<int64_t>(N, 4 * c_target);
        if (k_s <= 4096 && S >= k_s) {
            // scratch2: [sample scores Q*S][sample top ids Q*k_s][sample top scores Q*k_s][cand ids Q*cap][cand sc Q*cap][counts Q]
            const size_t b_samp = sizeof(float) * (size_t)Q * S, b_ks = sizeof(float) * (size_t)Q * k_s;
            const size_t b_cap = sizeof(float) * (size_t)Q * cap;
            JV_TRY(ctx->d_scratch2.reserve(b_samp + 2 * b_ks + 2 * b_cap + sizeof(unsigned int) * (size_t)Q + 1024));
            char *base = (char *)ctx->d_scratch2.ptr;
            float *d_samp = (float *)base;
            int32_t *d_ks_ids = (int32_t *)(base + b_samp);
            float *d_ks_sc = (float *)(base + b_samp + b_ks);
            int32_t *d_f_ids = (int32_t *)(base + b_samp + 2 * b_ks);
            float *d_f_sc = (float *)(base + b_samp + 2 * b_ks + b_cap);
            unsigned int *d_cnt = (unsigned int *)(base + b_samp + 2 * b_ks + 2 * b_cap);
            JV_TRY(ctx->d_scratch.reserve(topk_scratch_bytes(Q, std::max(std::max(k1, topK), k_s))));
            {
                ProfScope ps(ctx, R_SAMPLE);
                JV_TRY(launch_adc_mq_store(ctx->stream, ctx, l->d_luts, l->d_bmag, Q, codes->M, kvsf, codes->d_codes,
                                           codes->d_norms, 0, S, stride, d_samp));
            }
            {
                ProfScope ps(ctx, R_TOPK);
                JV_TRY(launch_topk(ctx->stream, ctx, d_samp, nullptr, Q, S, S, 0, k_s, d_ks_ids, d_ks_sc, ctx->d_scratch.ptr));
            }
            // adc_bq (default 1): the filter in two stages — a 7-bit bound scan for sixteen queries per LDS word, then exact scores of the
            // survivors only (k_adc_bq.hip); the lists hold a superset of {score >= tau} with exact scores, so the top-k1 below is the
            // same.  A query whose bound keeps more than cap2 candidates (or that has no usable bound) sends the call to the exact filter.
            bool bq_done = false;
            if (adc_bq_supported(codes->M, codes->d_codes, ctx->lds_per_block) && ctx_opt(ctx, "adc_bq", 1) != 0) {
                // (the exact filter aims at cap / 4 candidates per query; the bound keeps a few more: the same list size leaves 2-3x room)
                const int cap2 = cap;
                const size_t b_cap2 = sizeof(float) * (size_t)Q * cap2;
                JV_TRY(ctx->d_bq_work.reserve(adc_bq_scratch_bytes(Q, codes->M) + 2 * b_cap2 + 2 * sizeof(unsigned int) * (size_t)Q + 1024));
                char *wb = (char *)ctx->d_bq_work.ptr;
                int32_t *d_b_ids = (int32_t *)wb;
                float *d_b_sc = (float *)(wb + b_cap2);
                unsigned int *d_b_cnt2 = (unsigned int *)(wb + 2 * b_cap2);
                unsigned int *d_b_cnt = d_b_cnt2 + Q;
                void *d_b_work = wb + ((2 * b_cap2 + 2 * sizeof(unsigned int) * (size_t)Q + 255) & ~(size_t)255);
                JV_TRY(ctx->h_out.reserve(sizeof(unsigned int) * (size_t)Q));
                {
                    ProfScope ps(ctx, R_ADC);
                    JV_TRY(launch_adc_bq_scan(ctx->stream, ctx, l->d_luts, l->d_bmag, Q, codes->M, kvsf, codes->d_codes, codes->d_norms, 0, N,
                                              d_ks_sc + (k_s - 1), k_s, d_b_ids, d_b_cnt2, cap2, d_b_work));
                }
                // one small D2H + sync: the longest survivor list bounds the exact stage's work; a list that outgrew its slots -> fall back
                JV_HIP_CHECK(hipMemcpyAsync(ctx->h_out.ptr, d_b_cnt2, sizeof(unsigned int) * (size_t)Q, hipMemcpyDeviceToHost, ctx->stream));
                JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
                unsigned int longest = 0;
                bool ok = true;
                for (int q = 0; q < Q; ++q) {
                    const unsigned int c = ((const unsigned int *)ctx->h_out.ptr)[q];
                    longest = std::max(longest, c);
                    ok = ok && c >= (unsigned int)k1 && c <= (unsigned int)cap2;
                }
                if (ok) {
                    const int slots = (int)std::min<int64_t>(cap2, ((int64_t)longest + 63) & ~(int64_t)63);
                    {
                        ProfScope ps(ctx, R_ADC_EXACT);   // (a region of its own: the exact ADC sums of the bound scan's survivors + the count)
                        JV_TRY(launch_adc_bq_exact(ctx->stream, ctx, l->d_luts, l->d_bmag, Q, codes->M, kvsf, codes->d_codes, codes->d_norms, N,
                                                   d_ks_sc + (k_s - 1), k_s, d_b_ids, d_b_sc, d_b_cnt2, d_b_cnt, cap2, slots));
                    }
                    JV_HIP_CHECK(hipMemcpyAsync(ctx->h_out.ptr, d_b_cnt, sizeof(unsigned int) * (size_t)Q, hipMemcpyDeviceToHost, ctx->stream));
                    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
                    for (int q = 0; q < Q; ++q) ok = ok && ((const unsigned int *)ctx->h_out.ptr)[q] >= (unsigned int)k1;
                }
                ctx_stat_add(ctx, ok ? "adc_bq_calls" : "adc_bq_fallbacks", 1);
                if (ok) {
                    ProfScope ps(ctx, R_TOPK);
                    JV_TRY(launch_topk(ctx->stream, ctx, d_b_sc, d_b_ids, Q, cap2, cap2, 0, k1, d_k1_ids, d_k1_sc, ctx->d_scratch.ptr, d_b_cnt2));
                    done1 = bq_done = true;
                }
            }
            if (!bq_done) {
            JV_HIP_CHECK(hipMemsetAsync(d_cnt, 0, sizeof(unsigned int) * (size_t)Q, ctx->stream));
            {
                ProfScope ps(ctx, R_ADC);
                JV_TRY(launch_adc_mq_filter(ctx->stream, ctx, l->d_luts, l->d_bmag, Q, codes->M, kvsf, codes->d_codes,
                                            codes->d_norms, 0, N, d_ks_sc + (k_s - 1), k_s, d_f_ids, d_f_sc, d_cnt, cap));
            }
            // host check of the per-query list sizes (one small D2H + sync per call)
            JV_TRY(ctx->h_out.reserve(sizeof(unsigned int) * (size_t)Q));
            JV_HIP_CHECK(hipMemcpyAsync(ctx->h_out.ptr, d_cnt, sizeof(unsigned int) * (size_t)Q, hipMemcpyDeviceToHost,
                                        ctx->stream));
            JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            const unsigned int *cnt = (const unsigned int *)ctx->h_out.ptr;
            bool ok = true;
            for (int q = 0; q < Q; ++q) ok = ok && cnt[q] >= (unsigned int)k1 && cnt[q] <= (unsigned int)cap;
            if (ok) {
                ProfScope ps(ctx, R_TOPK);
                JV_TRY(launch_topk(ctx->stream, ctx, d_f_sc, d_f_ids, Q, cap, cap, 0, k1, d_k1_ids, d_k1_sc,
                                   ctx->d_scratch.ptr, d_cnt));
                done1 = true;
            }
            }
        }
    }
    if (!done1) {
        JV_TRY(ctx->d_scratch2.reserve(sizeof(float) * (size_t)Q * N));
        float *d_scores = (float *)ctx->d_scratch2.ptr;
        JV_TRY(ctx->d_scratch.reserve(topk_scratch_bytes(Q, std::max(k1, topK))));
        {
            ProfScope ps(ctx, R_ADC);
            if (mq)
                JV_TRY(launch_adc_mq_store(ctx->stream, ctx, l->d_luts, l->d_bmag, Q, codes->M, kvsf, codes->d_codes,
                                           codes->d_norms, 0, N, 1, d_scores));
            else
                JV_TRY(launch_adc(ctx->stream, ctx, l->d_luts, l->d_bmag, Q, codes->M, kvsf, codes->d_codes,
                                  codes->d_norms, N, 0, N, nullptr, d_scores));
        }
        ProfScope ps(ctx, R_TOPK);
        JV_TRY(launch_topk(ctx->stream, ctx, d_scores, nullptr, Q, N, N, 0, k1, d_k1_ids, d_k1_sc, ctx->d_scratch.ptr));
    }

    // ---------------- pass 2 ----------------
    if (rerank) {
        JV_TRY(rerank_gather(ctx, vectors, l->d_raw_queries, Q, vsf, d_cand, k1, d_exact, d_qnorm));
        ProfScope ps(ctx, R_TOPK);
        JV_TRY(launch_topk(ctx->stream, ctx, d_exact, d_cand, Q, k1, k1, 0, topK, (int32_t *)oi.dev, (float *)osc.dev,
                           ctx->d_scratch.ptr));
    }
    JV_TRY(launch_add_id_base(ctx->stream, (int32_t *)oi.dev, (int64_t)Q * topK, id_base));
    JV_TRY(stage_out_end(ctx, oi));
    return stage_out_end(ctx, osc);
}

}  // extern "C"
