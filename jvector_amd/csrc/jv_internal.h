// jv_internal.h — internal structures of libjvector_hip.so (not part of the ABI)
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <map>
#include <string>
#include <vector>

#include "../../include/jvector_hip.h"

namespace jv {

constexpr int kClusters = 256;  // ProductQuantization.DEFAULT_CLUSTERS (ProductQuantization.java:62)
// widest adjacency row the searchers take (the traversal kernels and the frontier kernels walk a row 64 neighbours at a time; the
// builder and the diversity kernel keep their own limit of 64)
constexpr int kMaxGraphDegree = 512;

void set_error(const char *fmt, ...);
void clear_error();

#define JV_HIP_CHECK(expr)                                                                         \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            jv::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            (void)hipGetLastError();                                                               \
            return (_e == hipErrorOutOfMemory) ? JV_ERR_OOM : JV_ERR_HIP;                          \
        }                                                                                          \
    } while (0)

#define JV_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            jv::set_error(__VA_ARGS__);  \
            return JV_ERR_INVALID;       \
        }                                \
    } while (0)

// entry points that read the float rows of a jv_vectors refuse a set made by jv_hip_vectors_from_nvq
#define JV_FLOAT_ROWS(v, what)                                                       \
    do {                                                                             \
        if ((v)->nvq) {                                                              \
            jv::set_error("%s: the vector set holds NVQ rows, not floats", what);   \
            return JV_ERR_UNSUPPORTED;                                               \
        }                                                                            \
    } while (0)

#define JV_TRY(expr)                 \
    do {                             \
        int _s = (expr);             \
        if (_s != JV_OK) return _s;  \
    } while (0)

// growable device / pinned-host scratch owned by a context
struct Buffer {
    void *ptr = nullptr;
    size_t cap = 0;
    bool pinned_host = false;
    int reserve(size_t bytes);
    void release();
};

}  // namespace jv

namespace jv {
enum Region : int { R_ADC = 0, R_TOPK, R_EXACT, R_LUT, R_ENCODE, R_NORMS, R_SAMPLE, R_GSEARCH, R_PRUNE, R_ADC_EXACT, R_COUNT };
struct ProfEvent {
    int region;
    hipEvent_t start, stop;
};
}  // namespace jv

struct jv_ctx {
    bool profiling = false;
    std::vector<jv::ProfEvent> prof_pending;   // recorded, not yet resolved
    std::vector<hipEvent_t> prof_free;         // event pool
    double prof_ms[jv::R_COUNT] = {0};
    int64_t prof_count[jv::R_COUNT] = {0};
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    int num_cus = 256;
    size_t lds_per_block = 65536;  // hipDeviceProp_t.sharedMemPerBlock / maxSharedMemoryPerMultiProcessor
    // staging: host->device inputs, device->host outputs (pinned), device scratch
    jv::Buffer h_in, h_out, d_in, d_out, d_scratch, d_scratch2, d_scratch3;
    // device-resident graph traversal: per-worker visited tables / spill tiers and the per-query result staging
    jv::Buffer d_gs_visited, d_gs_spill, d_gs_out, d_gs_mask, d_gs_big;
    jv::Buffer d_gs_extra;   // session kernels: the evictedResults a resume() pushes back (graph_search.cpp)
    jv::Buffer d_gs_ubr;     // UBR: the batch's upper-bound tables (M x 256 bytes per query) + 4 floats of meta per query
    jv::Buffer d_bq_work;    // flat search: bound tables of the query batch (k_adc_bq.hip)
    jv::Buffer d_rd_counts;  // robust prune: {isDiverse tests, (candidate, selected slot) pairs summed by them}, accumulated by every launch
    jv::Buffer d_nvq_q;   // NVQ rerank: shifted queries + per-query scalars (nvq.cpp)
    // host batched graph searcher: worker pool (graph_search.cpp owns the type) and its destructor
    void *host_pool = nullptr;
    void (*host_pool_destroy)(void *) = nullptr;
    // a context belongs to ONE host thread (jvector_hip.h); the long-running searches take this flag and refuse a second
    // concurrent caller instead of corrupting the shared staging buffers / worker pool
    std::atomic<int> busy{0};
    // jv_hip_ctx_set_option / jv_hip_ctx_get_stat (jvector_hip.h): per-context tuning options (they win over the process-wide
    // JVECTOR_HIP_* environment defaults) and event counters of the searches that ran on this context
    std::map<std::string, long long> opts, stats;
};

namespace jv {
inline uint64_t next_fused_generation()
{
    static std::atomic<uint64_t> g{1};
    return g.fetch_add(1, std::memory_order_relaxed);
}
// option `name` of this context: set by jv_hip_ctx_set_option, else the environment variable JVECTOR_HIP_<NAME>, else dflt
long long ctx_opt(const jv_ctx *ctx, const char *name, long long dflt);
bool ctx_opt_is_set(const jv_ctx *ctx, const char *name);
inline void ctx_stat_add(jv_ctx *ctx, const char *name, long long v) { ctx->stats[name] += v; }
inline void ctx_stat_set(jv_ctx *ctx, const char *name, long long v) { ctx->stats[name] = v; }

struct CtxBusy {
    jv_ctx *c;
    bool ok;
    explicit CtxBusy(jv_ctx *cc) : c(cc), ok(cc->busy.exchange(1, std::memory_order_acquire) == 0) {}
    ~CtxBusy()
    {
        if (ok) c->busy.store(0, std::memory_order_release);
    }
};
}  // namespace jv

struct jv_pq {
    int device = 0;
    int D = 0, M = 0, k = 0;   // k: rows per codebook in device memory — always kClusters (see k_user)
    // clusterCount the caller gave (ProductQuantization allows 1..256).  A smaller count is stored PADDED to 256 rows per
    // sub-space with copies of centroid 0: closestCentroidIndex keeps the FIRST minimum (ProductQuantization.java:507-520: strict
    // `<`), so a copy of row 0 is never chosen, every code stays < k_user, and every kernel keeps its 256-row table stride
    int k_user = 0;
    bool uniform = false;      // all subvector sizes equal
    int max_size = 0;
    std::vector<int> sizes, offsets;
    std::vector<int64_t> cb_offsets;  // float offset of codebook m inside d_codebooks
    int *d_sizes = nullptr, *d_offsets = nullptr;
    int64_t *d_cb_offsets = nullptr;
    float *d_codebooks = nullptr;
    float *d_centroid = nullptr;   // nullable
    float *d_self_mag = nullptr;   // M*k floats, built at create (calculatePartialSelfMagnitudes)
    // uniform sub-vector sizes: the codebooks once more with centroids PAIRED — entry (m, i/2, j) = {c[m][i][j], c[m][i+1][j]} —
    // for pq_encode_kernel's packed two-centroid chains (scalar loads feed both halves of a v_pk operand); built at create
    float *d_cb_paired = nullptr;
    float aniso = -1.0f;           // anisotropicThreshold; -1 = UNWEIGHTED (ProductQuantization.java:72)
};

struct jv_codes {
    int device = 0;
    const jv_pq *pq = nullptr;
    int64_t count = 0;
    int M = 0;
    uint8_t *d_codes = nullptr;
    bool owns = false;
    // cosine: per-ordinal decoded squared magnitude sum_m aMag[m*256+code[m]] (query independent);
    // (re)built lazily after uploads.  See DESIGN.md "cosine ADC".
    float *d_norms = nullptr;
    bool norms_valid = false;
};

struct jv_vectors {
    int device = 0;
    int64_t count = 0;
    int D = 0;
    float *d_vecs = nullptr;
    bool owns = false;
    // cosine rerank: per-row sum of squares (the query-independent norm2 accumulator), built lazily by
    // ensure_vector_norms, invalidated by uploads
    float *d_sqnorm = nullptr;
    bool sqnorm_valid = false;
    // jv_hip_vectors_from_nvq: no float rows at all (d_vecs == nullptr) — every rerank through this set scores the NVQ rows
    // (NVQ.rerankerFor, B/graph/disk/feature/NVQ.java:96-110); calls that need the floats themselves refuse it
    struct jv_nvq_vectors *nvq = nullptr;
};

// NVQuantization (B/quantization/NVQuantization.java): the global mean and the sub-vector split
struct jv_nvq {
    int device = 0;
    int D = 0, S = 0;
    bool learn = true;           // NVQuantization.learn (:129): false = growthRate 1e-2f without the search
    float *d_mean = nullptr;     // D
    float *d_grid = nullptr;     // growth-rate candidates of quantizeTo's two loops (k_nvq.hip nvq_encode_kernel)
};

// NVQVectors on the device (layout: k_nvq.hip header)
struct jv_nvq_vectors {
    int device = 0;
    const jv_nvq *nvq = nullptr;
    int64_t count = 0;
    int ld = 0;                  // row stride of d_bytes: D rounded up to 16
    uint8_t *d_bytes = nullptr;
    float *d_params = nullptr;   // count x S x {min, max, growthRate, midpoint}
    float *d_derived = nullptr;  // count x S x {1/scaledGrowthRate, scaledMidpoint, logisticScale, logisticBias}
    float *d_cosnorm = nullptr;  // count (cosine only, lazily)
    bool derived_valid = false, cosnorm_valid = false;
};

struct jv_pair_table {   // ProductQuantization.createCodebookPartialSums on the device (build_score.cpp)
    int device = 0;
    const jv_pq *pq = nullptr;
    jv_vsf vsf = JV_EUCLIDEAN;
    float *d_tri = nullptr;
    int64_t floats = 0;
    float *d_sq = nullptr;   // the same entries as [M][k][k] (rd_square = 1; built on first use by pair_table_square)
};

struct jv_fused {
    int device = 0;
    const jv_pq *pq = nullptr;
    int64_t count = 0;
    int maxDegree = 0, M = 0;
    uint8_t *d_blocks = nullptr;    // count x maxDegree x M
    int32_t *d_neighbors = nullptr; // count x maxDegree
    float *d_norms = nullptr;       // count x maxDegree (cosine), lazily built
    bool norms_valid = false;
    uint64_t generation = 0;        // set from a process-wide counter by every upload / build (consumers that cached a consistency check compare it)
};

struct jv_luts {
    int device = 0;
    const jv_pq *pq = nullptr;
    int capacity = 0;
    int Q = 0;
    jv_vsf vsf = JV_EUCLIDEAN;
    jv_decoder_kind kind = JV_DECODER_PQ;
    float *d_luts = nullptr;   // capacity x M x 256
    float *d_bmag = nullptr;   // capacity
    float *d_queries = nullptr; // capacity x D : centred queries (cq = q - globalCentroid)
    float *d_raw_queries = nullptr; // capacity x D : un-centred copy (rerank)
    bool tables_valid = false;  // d_luts holds the tables of the current queries (false after a queries-only prepare)
};

// ---- host helpers shared by cabi.cpp and graph_search.cpp ----
namespace jv {

struct OutStage {
    void *user = nullptr;   // user pointer
    void *dev = nullptr;    // device pointer kernels write to
    size_t bytes = 0;
    bool host = false;
};
bool is_device_ptr(const void *p);
int stage_in(jv_ctx *ctx, const void *src, size_t bytes, Buffer &pin, Buffer &dev, const void **out);
int stage_out_begin(jv_ctx *ctx, void *dst, size_t bytes, Buffer &dev, OutStage *st);
int stage_out_end(jv_ctx *ctx, const OutStage &st);
int to_kernel_vsf(jv_vsf v);
int use_device(int device);
int ensure_code_norms(jv_ctx *ctx, jv_codes *codes);
int ensure_vector_norms(jv_ctx *ctx, jv_vectors *v);
int ensure_fused_norms(jv_ctx *ctx, jv_fused *f);

// RAII region timer: two hipEventRecord calls on the context's stream when profiling is on, nothing otherwise.
struct ProfScope {
    jv_ctx *ctx;
    int idx = -1;
    ProfScope(jv_ctx *c, int region) : ctx(c)
    {
        if (!ctx->profiling) return;
        ProfEvent e;
        e.region = region;
        auto get = [&](hipEvent_t *ev) {
            if (!ctx->prof_free.empty()) {
                *ev = ctx->prof_free.back();
                ctx->prof_free.pop_back();
                return true;
            }
            return hipEventCreate(ev) == hipSuccess;
        };
        if (!get(&e.start) || !get(&e.stop)) return;
        (void)hipEventRecord(e.start, ctx->stream);
        ctx->prof_pending.push_back(e);
        idx = (int)ctx->prof_pending.size() - 1;
    }
    ~ProfScope()
    {
        if (idx >= 0) (void)hipEventRecord(ctx->prof_pending[idx].stop, ctx->stream);
    }
};

}  // namespace jv

// ---- kernel launchers (implemented in the .hip translation units) ----
namespace jv {

int launch_self_magnitudes(hipStream_t s, const jv_pq *pq);
int launch_center_queries(hipStream_t s, const jv_pq *pq, const float *d_q, int Q, float *d_cq);
int launch_lut_build(hipStream_t s, const jv_pq *pq, const float *d_cq, int Q, int lut_vsf, float *d_luts);
int launch_query_magnitudes(hipStream_t s, const jv_pq *pq, const float *d_cq, int Q, int kind, float *d_bmag);
int launch_pq_encode(hipStream_t s, const jv_pq *pq, const float *d_vecs, int64_t count, uint8_t *d_codes);
int launch_pq_encode_anisotropic(hipStream_t s, const jv_pq *pq, const float *d_vecs, int64_t count, uint8_t *d_codes);

// raw table sums (used for the cosine norms): out[i] = sum_m table[m*256+code[m]]
int launch_code_norms(hipStream_t s, const jv_ctx *ctx, const float *d_table, int M, const uint8_t *d_codes,
                      int64_t count, float *d_out);
// ADC: ordinals==nullptr -> contiguous scan of [first, first+count); else gather Q x B
int launch_adc(hipStream_t s, const jv_ctx *ctx, const float *d_luts, const float *d_bmag, int Q, int M, int vsf,
               const uint8_t *d_codes, const float *d_norms, int64_t n_codes, int64_t first, int64_t count,
               const int32_t *d_ordinals, float *d_out);
int launch_fused(hipStream_t s, const jv_ctx *ctx, const float *d_luts, const float *d_bmag, int Q, int M, int vsf,
                 const uint8_t *d_blocks, const int32_t *d_neighbors, const float *d_norms, int maxDegree,
                 int64_t n_nodes, const int32_t *d_origins, float *d_out, int32_t *d_neighbors_out);

int launch_gather_rows(hipStream_t s, const float *d_vecs, int64_t n, int D, const int32_t *d_ord, int P, float *d_out, int32_t *d_cand,
                       int B);
int launch_exact_gather(hipStream_t s, const float *d_vecs, int64_t n, int D, const float *d_q, int Q, int vsf,
                        const int32_t *d_ord, int B, float *d_out, float *d_qnorm, const float *d_vnorm);
int launch_row_sqnorms(hipStream_t s, const float *d_vecs, int64_t n, int D, float *d_out);
// the rerank fused into the traversal wave (gs_body.h gs_rr_round, round 6): how many rows [0, r) of every list of B the wave scores
// itself — 0 = this shape keeps the kernel of its own (rows not 16-byte aligned / D % 8 != 0 / no norm table / B > 256 / the
// developer selectors of launch_exact_gather); r < B: the remainder [r, B) is one packed launch of launch_exact_gather_tail
// (4 <= B - r <= 32, Q >= 2).  The query norms of a cosine search must exist before the traversal starts: launch_query_sqnorms.
int exact_fused_rows(const float *d_vecs, int D, const float *d_q, int Q, int vsf, int B, const float *d_vnorm);
int launch_query_sqnorms(hipStream_t s, const float *d_q, int D, int Q, float *d_qnorm);
bool exact_tr_supported(const float *d_vecs, int D);
int launch_block8_sqnorms(hipStream_t s, const float *d_rows, int64_t n, int D, float *d_out);
int launch_exact_gather_tail(hipStream_t s, const float *d_vecs, int64_t n, int D, const float *d_q, int Q, int vsf, const int32_t *d_ord,
                             int B, int first, float *d_out, const float *d_qnorm, const float *d_vnorm);
// NVQ (k_nvq.hip)
int launch_nvq_mean(hipStream_t s, const float *d_vecs, int64_t n, int D, float *d_mean);
size_t nvq_encode_lds_bytes(int D, int S);
int launch_nvq_encode(hipStream_t s, const jv_ctx *ctx, const float *d_vecs, int64_t count, int D, int S, const float *d_mean, int learn,
                      const float *d_grid, uint8_t *d_bytes, int ld, float *d_params);
int launch_nvq_derive(hipStream_t s, const float *d_params, int64_t units, float *d_derived);
int launch_nvq_cosnorm(hipStream_t s, const uint8_t *d_bytes, int ld, int64_t n, int D, int S, const float *d_derived, const float *d_mean,
                       float *d_out);
int launch_nvq_gather(hipStream_t s, const uint8_t *d_bytes, int ld, int64_t n, int D, int S, const float *d_derived, const float *d_cosnorm,
                      const float *d_mean, const float *d_q, int Q, int vsf, const int32_t *d_ord, int B, float *d_out, float *d_qwork,
                      float *d_qaux);
// the reranker of a vector set (nvq.cpp): full-resolution rows through launch_exact_gather (cosine: the norm table is
// made first), NVQ rows for a set made by jv_hip_vectors_from_nvq.  d_qnorm: Q floats of scratch.
int rerank_gather(jv_ctx *ctx, const jv_vectors *v, const float *d_q, int Q, jv_vsf vsf, const int32_t *d_ord, int B, float *d_out,
                  float *d_qnorm);
// MFMA tile form (k_exact_dense.hip / ed_body.h): fused chains, not bit-identical to launch_exact_scan
int launch_exact_scan_dense(hipStream_t s, const float *d_vecs, int D, const float *d_q, int Q, int vsf, int64_t first,
                            int64_t count, float *d_out);
int launch_exact_scan(hipStream_t s, const jv_ctx *ctx, const float *d_vecs, int D, const float *d_q, int Q, int vsf,
                      int64_t first, int64_t count, float *d_out, float *d_qnorm);

int launch_frontier(hipStream_t s, int vsf, const float *d_luts, const float *d_bmag, const int32_t *d_slot_query,
                    const int32_t *d_origins, const int32_t *d_ord_index, const int32_t *d_ords, const jv_fused *fused,
                    const jv_codes *codes, float *d_out, int S, int W, const jv_pq *pq = nullptr,
                    const float *d_cq = nullptr);
// PQ training (k_pq_train.hip; parameters in km_body.h)
struct KmParams;
int launch_km_centroid(hipStream_t s, const float *d_X, int64_t n, int D, float *d_out);
int launch_km_center(hipStream_t s, const float *d_X, const float *d_centroid, int64_t n, int D, float *d_Xc);
int launch_km_pp_init(hipStream_t s, const KmParams &p);
int launch_km_assign(hipStream_t s, const KmParams &p);
int launch_km_replay(hipStream_t s, const KmParams &p, int first_pass);
int launch_km_update_centroids(hipStream_t s, const KmParams &p);
int launch_km_finish_round(hipStream_t s, const KmParams &p);
int launch_km_aniso_round(hipStream_t s, const KmParams &p);
int launch_km_reactivate(hipStream_t s, const KmParams &p);
// build-time scoring (k_build_score.hip)
int launch_pair_table(hipStream_t s, const jv_pq *pq, int vsf, float *d_out);
int launch_pair_table_square(hipStream_t s, const float *d_tri, int M, int k, float *d_sq);
const float *pair_table_square(jv_ctx *ctx, jv_pair_table *t);   // nullptr (and the error set) when it cannot be built
struct RdParams;
int launch_retain_diverse(hipStream_t s, const jv_ctx *ctx, const RdParams &p);
size_t retain_diverse_lds_bytes(int C, int M);
bool retain_diverse_table_free(const jv_ctx *ctx, const jv_pq *pq);
int retain_diverse_chunk(const jv_ctx *ctx);
int launch_pair_scores(hipStream_t s, const float *d_tri, int vsf, const jv_codes *codes, const int32_t *d_node1, int P,
                       const int32_t *d_node2, int B, float *d_out);
int launch_fused_gather(hipStream_t s, const jv_codes *codes, const int32_t *d_neighbors, int maxDegree, int64_t count, uint8_t *d_blocks);
int launch_pq_decode(hipStream_t s, const jv_codes *codes, const int32_t *d_ordinals, int64_t first, int64_t count, float *d_out);
int launch_direct_scores(hipStream_t s, const jv_codes *codes, int vsf, const float *d_cq, int Q, const int32_t *d_ordinals, int B,
                         float *d_qnorm, float *d_out);
// batched graph construction (k_builder.hip; bodies and parameters in bl_body.h)
struct BlApplyParams;
struct BlMergeParams;
struct BlSortParams;
struct BlRowsParams;
struct BlOverParams;
struct BlImproveParams;
struct BlRowEdgesParams;
int graph_search_excluding(jv_ctx *ctx, const jv_graph *g, jv_luts *l, const jv_codes *codes, const float *queries, int Q, jv_vsf vsf, int topK,
                           int rerankK, const int32_t *exclude, int32_t *out_ids, float *out_scores, int64_t *stats);
struct BlRoApplyParams;
struct BlRoMergeParams;
struct BlRoRowsParams;
struct BlRoCopyParams;
int launch_bl_ro_apply_selection(hipStream_t s, const BlRoApplyParams &p);
int launch_bl_ro_backlink_merge(hipStream_t s, const BlRoMergeParams &p);
int launch_bl_ro_rewrite_rows(hipStream_t s, const BlRoRowsParams &p);
int launch_bl_ro_copy_rows(hipStream_t s, const BlRoCopyParams &p);
struct BlRoImproveParams;
struct BlSelIdsParams;
struct BlRoApplySortedParams;
int launch_bl_sel_ids(hipStream_t s, const BlSelIdsParams &p);
int launch_bl_ro_apply_sorted(hipStream_t s, const BlRoApplySortedParams &p);
struct BlRoRowEdgesParams;
int launch_bl_ro_improve_list(hipStream_t s, const BlRoImproveParams &p);
int launch_bl_ro_row_edges(hipStream_t s, const BlRoRowEdgesParams &p);
int launch_bl_apply_selection(hipStream_t s, const BlApplyParams &p);
int launch_bl_backlink_merge(hipStream_t s, const BlMergeParams &p);
int launch_bl_improve_list(hipStream_t s, const BlImproveParams &p);
int launch_bl_row_edges(hipStream_t s, const BlRowEdgesParams &p);
int launch_bl_rank_sort(hipStream_t s, const BlSortParams &p);
int launch_bl_rewrite_rows(hipStream_t s, const BlRowsParams &p);
int launch_bl_list_over_degree(hipStream_t s, const BlOverParams &p);
int launch_bl_count_valid(hipStream_t s, const int32_t *cand, int C, int32_t *count, long long B);
int launch_bl_copy_rows(hipStream_t s, const int32_t *nbrs, int R, const int32_t *tgt, long long P, int32_t *out);
int launch_bl_strided_copy(hipStream_t s, const int32_t *src, int R, int Rf, long long N, int32_t *dst);
int launch_bl_sort_edges(hipStream_t s, void *temp, size_t *temp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out,
                         const int32_t *vals_in, int32_t *vals_out, long long n, int end_bit);
// device-resident graph traversal (k_gsearch.hip; parameters in gs_params.h)
struct GsParams;
// stage the queries of a batch (raw copy, centred copy, cosine query magnitudes); with_tables also builds the ADC look-up
// tables (jv_hip_luts_build = with_tables true).  The table-free traversal kernels pass false: 96 KB per query saved.
int luts_prepare(jv_ctx *ctx, jv_luts *l, const float *queries, int Q, jv_vsf vsf, jv_decoder_kind kind, bool with_tables);
// jv_hip_pq_create; pad = false keeps a quantizer of fewer than 256 clusters in its k-row layout (training work objects only)
int pq_create_impl(jv_ctx *ctx, int D, int M, int k, const int *sizes, const float *codebooks, const float *centroid, bool pad, jv_pq **out);
bool graph_search_device_supported(const jv_pq *pq, const jv_codes *codes, const jv_fused *fused, int max_degree, int n_levels);
bool graph_search_device_specialised(const jv_pq *pq, const jv_codes *codes, const jv_fused *fused);  // else: the generic kernels
size_t graph_search_lds_bytes(int D, int rerankK, int cand_cap, int pair_M, int evict_cap = 0, int v1_log2 = 0);
int launch_graph_search(hipStream_t s, int vsf, const GsParams &p, int workers, int occupancy);
// the register-table bound form (gs_body.h "UBR", k_gsearch_ubr.hip): tables of a batch, then the traversal
bool graph_search_ubr_supported(int M, int kernel_vsf);
int launch_ubr_tables(hipStream_t s, int vsf, const float *codebooks, const float *cq, int Q, int M, uint32_t *tab, float *meta);
int launch_graph_search_ubr(hipStream_t s, int vsf, const GsParams &p, int workers, size_t lds);
// the workgroup form (k_gsearch_wgx.hip): one query per workgroup, the ADC table in LDS
bool graph_search_wgx_supported(int M);
size_t graph_search_wgx_lds_bytes(int D, int rerankK, int cand_cap, int evict_cap, int v1_log2, int slots, int kps, int logcap, int M);
int launch_graph_search_wgx(hipStream_t s, int vsf, const GsParams &p, int workgroups, int threads);
bool graph_search_session_supported(int M);
int launch_graph_search_session(hipStream_t s, int vsf, const GsParams &p, int workers, size_t lds);
size_t topk_scratch_bytes(int Q, int k);
struct RtParams;
int launch_rerank_ties(hipStream_t s, const RtParams &p);
int launch_topk(hipStream_t s, const jv_ctx *ctx, const float *d_scores, const int32_t *d_ids, int Q, int64_t n,
                int64_t stride, int32_t id_base, int k, int32_t *d_out_ids, float *d_out_scores, void *d_scratch,
                const unsigned int *d_row_counts = nullptr);
bool adc_mq_supported(int M, const uint8_t *d_codes);
int launch_adc_mq_store(hipStream_t s, const jv_ctx *ctx, const float *d_luts, const float *d_bmag, int Q, int M,
                        int vsf, const uint8_t *d_codes, const float *d_norms, int64_t first, int64_t count,
                        int64_t row_stride, float *d_out);
int launch_adc_mq_filter(hipStream_t s, const jv_ctx *ctx, const float *d_luts, const float *d_bmag, int Q, int M,
                         int vsf, const uint8_t *d_codes, const float *d_norms, int64_t first, int64_t count,
                         const float *d_tau, int tau_stride, int32_t *d_cand_ids, float *d_cand_scores,
                         unsigned int *d_cand_count, int cap);
// the same filter in two stages (k_adc_bq.hip): 7-bit bound tables for sixteen queries per LDS word drop what cannot reach tau, the
// exact ADC score is computed for the survivors only
bool adc_bq_supported(int M, const uint8_t *d_codes, size_t lds_per_block);
size_t adc_bq_scratch_bytes(int Q, int M);
int launch_adc_bq_scan(hipStream_t s, const jv_ctx *ctx, const float *d_luts, const float *d_bmag, int Q, int M, int vsf, const uint8_t *d_codes,
                       const float *d_norms, int64_t first, int64_t count, const float *d_tau, int tau_stride, int32_t *d_ids,
                       unsigned int *d_surv_cnt, int cap2, void *d_work);
int launch_adc_bq_exact(hipStream_t s, const jv_ctx *ctx, const float *d_luts, const float *d_bmag, int Q, int M, int vsf, const uint8_t *d_codes,
                        const float *d_norms, int64_t n_codes, const float *d_tau, int tau_stride, const int32_t *d_ids, float *d_scores,
                        const unsigned int *d_surv_cnt, unsigned int *d_cnt, int cap2, int slots);
int launch_adc_pitched(hipStream_t s, const jv_ctx *ctx, const float *d_luts, const float *d_bmag, int Q, int M, int vsf, const uint8_t *d_codes,
                       const float *d_norms, int64_t n_codes, int64_t count, int64_t ld, const int32_t *d_ordinals, float *d_out);
int launch_add_id_base(hipStream_t s, int32_t *d_ids, int64_t n, int32_t base);

}  // namespace jv
