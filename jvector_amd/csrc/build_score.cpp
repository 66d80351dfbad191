// build_score.cpp — C ABI of the build-time scoring entry points (kernels in k_build_score.hip).
#include "jv_internal.h"
#include "rd_params.h"

using namespace jv;

extern "C" {

int jv_hip_pair_table_create(jv_ctx *ctx, const jv_pq *pq, jv_vsf vsf, jv_pair_table **out)
{
    clear_error();
    JV_REQUIRE(ctx && pq && out, "pair_table_create: NULL argument");
    JV_REQUIRE(vsf == JV_EUCLIDEAN || vsf == JV_DOT_PRODUCT || vsf == JV_COSINE, "Unsupported similarity function %d", (int)vsf);
    JV_TRY(use_device(ctx->device));
    jv_pair_table *t = new jv_pair_table();
    t->device = ctx->device;
    t->pq = pq;
    t->vsf = vsf;
    t->floats = (int64_t)pq->M * pq->k * (pq->k + 1) / 2;
    hipError_t e = hipMalloc((void **)&t->d_tri, sizeof(float) * (size_t)t->floats);
    if (e != hipSuccess) {
        set_error("pair_table_create: hipMalloc of %lld floats failed: %s", (long long)t->floats, hipGetErrorString(e));
        (void)hipGetLastError();
        delete t;
        return JV_ERR_OOM;
    }
    int st = launch_pair_table(ctx->stream, pq, to_kernel_vsf(vsf), t->d_tri);
    if (st == JV_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) {  // shared by every context that uses the table
        set_error("pair_table_create: kernel failed");
        st = JV_ERR_HIP;
    }
    if (st != JV_OK) {
        (void)hipFree(t->d_tri);
        delete t;
        return st;
    }
    *out = t;
    return JV_OK;
}

int64_t jv_hip_pair_table_size(const jv_pair_table *t) { return t ? t->floats : 0; }

int jv_hip_pair_table_download(jv_ctx *ctx, const jv_pair_table *t, float *dst)
{
    clear_error();
    JV_REQUIRE(ctx && t && dst, "pair_table_download: NULL argument");
    JV_TRY(use_device(ctx->device));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    JV_HIP_CHECK(hipMemcpy(dst, t->d_tri, sizeof(float) * (size_t)t->floats, hipMemcpyDefault));
    return JV_OK;
}

int jv_hip_pair_table_destroy(jv_pair_table *t)
{
    if (!t) return JV_OK;
    (void)hipSetDevice(t->device);
    (void)hipFree(t->d_tri);
    if (t->d_sq) (void)hipFree(t->d_sq);
    delete t;
    return JV_OK;
}

int jv_hip_code_pair_scores(jv_ctx *ctx, const jv_pair_table *t, const jv_codes *codes, const int32_t *node1, int P,
                            const int32_t *node2, int B, float *scores_out)
{
    clear_error();
    JV_REQUIRE(ctx && t && codes, "code_pair_scores: NULL argument");
    JV_REQUIRE(codes->pq == t->pq, "code_pair_scores: the code store and the pair table use different codebooks");
    JV_REQUIRE(P >= 0 && B >= 0, "code_pair_scores: negative sizes");
    if (P == 0 || B == 0) return JV_OK;
    JV_REQUIRE(node1 && node2 && scores_out, "code_pair_scores: NULL buffer");
    JV_TRY(use_device(ctx->device));
    const void *d_n1 = nullptr, *d_n2 = nullptr;
    JV_TRY(stage_in(ctx, node1, sizeof(int32_t) * (size_t)P, ctx->h_in, ctx->d_in, &d_n1));
    JV_TRY(stage_in(ctx, node2, sizeof(int32_t) * (size_t)P * B, ctx->h_in, ctx->d_scratch2, &d_n2));
    OutStage os;
    JV_TRY(stage_out_begin(ctx, scores_out, sizeof(float) * (size_t)P * B, ctx->d_out, &os));
    {
        ProfScope ps(ctx, R_ADC);
        JV_TRY(launch_pair_scores(ctx->stream, t->d_tri, to_kernel_vsf(t->vsf), codes, (const int32_t *)d_n1, P,
                                  (const int32_t *)d_n2, B, (float *)os.dev));
    }
    return stage_out_end(ctx, os);
}

}  // extern "C"
namespace jv {
// rd_table_free = 1: the robust prune recomputes the pair-table entries from the codebook (rd_body.h rd_node<true>; every sub-vector
// 8 dimensions — the shape its 16-byte centroid loads assume) instead of looking them up in the 12.6 / 25 MB table that misses L2
// four times out of five.  Same selections bit for bit — and SLOWER on the MI355X (profiles/r4_t: headline build prune + backlink
// 25.7 -> 35.3 s, C5 37.8 -> 64.3 s): two 16-byte gathers per (candidate, selected, subspace) cost more than one 4-byte look-up
// that goes to the Infinity Cache.  Off by default.
// rd_square = 1 (an experiment, off by default — not yet measured at scale): the robust prune reads the pair table's SQUARE form
// (rd_body.h rd_node<.., SQ>; DESIGN.md §7: the kernel is bound by L2 -> L1 line fills, and a test's lanes then share one 1 KB row per
// subspace).  Built once per table, on first use.
// (rd_table_free / rd_square / rd_chunk are measured-and-switched-off variants: compiled only with JV_EXPERIMENTAL, make EXPERIMENTAL=1)
const float *pair_table_square(jv_ctx *ctx, jv_pair_table *t)
{
#ifndef JV_EXPERIMENTAL
    (void)ctx;
    (void)t;
    return nullptr;
#else
    if (ctx_opt(ctx, "rd_square", 0) == 0) return nullptr;
    if (t->d_sq) return t->d_sq;
    const int M = t->pq->M, k = t->pq->k;
    if ((int64_t)M * k * k >= (1ll << 31)) return nullptr;
    float *sq = nullptr;
    if (hipMalloc((void **)&sq, sizeof(float) * (size_t)M * k * k) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (launch_pair_table_square(ctx->stream, t->d_tri, M, k, sq) != JV_OK || hipStreamSynchronize(ctx->stream) != hipSuccess) {
        (void)hipFree(sq);
        return nullptr;
    }
    t->d_sq = sq;
    return sq;
#endif
}

bool retain_diverse_table_free(const jv_ctx *ctx, const jv_pq *pq)
{
#ifndef JV_EXPERIMENTAL
    (void)ctx;
    (void)pq;
    return false;
#else
    return pq->uniform && pq->max_size == 8 && pq->D == 8 * pq->M && pq->k == kClusters && ctx_opt(ctx, "rd_table_free", 0) != 0;
#endif
}
// rd_chunk > 0: selected slots a robust-prune test examines per step of its INCREMENTAL walk (rd_body.h; RdParams::chunk: a
// candidate remembers the slots it has been tested against and their largest similarity; only new slots are examined, and the walk
// stops at the first violation).  Selections are identical for every value.  Measured on the MI355X (profiles/r4_w, before the
// kernel's instruction diet): 64 (memory of earlier tests only) changes nothing, 8 is 30 % SLOWER — a prune is a chain of ~200
// dependent tests per wave and its cost is the length of that chain, not the lanes that take part.  0 (default): every test examines
// every selected slot in one step.
int retain_diverse_chunk(const jv_ctx *ctx)
{
#ifndef JV_EXPERIMENTAL
    (void)ctx;
    return 0;
#else
    return (int)std::max<long long>(0, std::min<long long>(64, ctx_opt(ctx, "rd_chunk", 0)));
#endif
}
}  // namespace jv
extern "C" {

int jv_hip_retain_diverse(jv_ctx *ctx, const jv_pair_table *t, const jv_codes *codes, int P, int C, const int32_t *cand_nodes,
                          const float *cand_scores, const int32_t *cand_count, const int32_t *diverse_before, int maxDegree, float alpha,
                          int32_t *selected_out, int32_t *n_selected_out, float *short_edges_out)
{
    clear_error();
    JV_REQUIRE(ctx && t && codes, "retain_diverse: NULL argument");
    JV_REQUIRE(codes->pq == t->pq, "retain_diverse: the code store and the pair table use different codebooks");
    JV_REQUIRE(P >= 0 && C >= 0, "retain_diverse: negative sizes");
    JV_REQUIRE(maxDegree >= 1 && maxDegree <= 64, "retain_diverse: maxDegree %d outside 1..64", maxDegree);
    JV_REQUIRE(alpha == alpha && alpha >= 1.0f && alpha <= 64.0f, "retain_diverse: alpha must lie in [1, 64]");
    if (P == 0) return JV_OK;
    JV_REQUIRE(C > 0 && cand_nodes && cand_scores && selected_out && n_selected_out, "retain_diverse: NULL buffer");
    JV_TRY(use_device(ctx->device));
    const size_t cells = (size_t)P * C;
    const void *d_nodes = nullptr, *d_scores = nullptr, *d_count = nullptr, *d_before = nullptr;
    // four inputs, one pinned staging buffer: stage them one after another into distinct device buffers
    JV_TRY(stage_in(ctx, cand_nodes, sizeof(int32_t) * cells, ctx->h_in, ctx->d_in, &d_nodes));
    JV_TRY(stage_in(ctx, cand_scores, sizeof(float) * cells, ctx->h_in, ctx->d_scratch2, &d_scores));
    if (cand_count) JV_TRY(stage_in(ctx, cand_count, sizeof(int32_t) * (size_t)P, ctx->h_in, ctx->d_scratch3, &d_count));
    if (diverse_before) JV_TRY(stage_in(ctx, diverse_before, sizeof(int32_t) * (size_t)P, ctx->h_in, ctx->d_gs_mask, &d_before));
    // outputs: selected [P][maxDegree] + n_selected [P] + short_edges [P] in one device block
    const size_t sel_bytes = sizeof(int32_t) * (size_t)P * maxDegree, cnt_bytes = sizeof(int32_t) * (size_t)P;
    const size_t o_cnt = (sel_bytes + 255) & ~(size_t)255, o_se = (o_cnt + cnt_bytes + 255) & ~(size_t)255;
    JV_TRY(ctx->d_out.reserve(o_se + sizeof(float) * (size_t)P));
    char *base = (char *)ctx->d_out.ptr;
    RdParams p{};
    p.tri = t->d_tri;
    p.sq = pair_table_square(ctx, const_cast<jv_pair_table *>(t));
    p.codebooks = retain_diverse_table_free(ctx, t->pq) ? t->pq->d_codebooks : nullptr;
    p.codes = codes->d_codes;
    p.n = codes->count;
    p.cand_nodes = (const int32_t *)d_nodes;
    p.cand_scores = (const float *)d_scores;
    p.cand_count = (const int32_t *)d_count;
    p.diverse_before = (const int32_t *)d_before;
    p.P = P;
    p.C = C;
    p.M = codes->M;
    p.k = t->pq->k;
    p.vsf = to_kernel_vsf(t->vsf);
    p.maxDegree = maxDegree;
    p.alpha = alpha;
    p.chunk = retain_diverse_chunk(ctx);
    p.split = ctx_opt(ctx, "rd_split", 1) != 0 ? 1 : 0;   // (rd_body.h rd_pair_sum_split: idle lanes share a slot's entries)
    p.wide_stage = ctx_opt(ctx, "rd_wide_stage", 1) != 0 ? 1 : 0;   // (candidate rows staged 16 bytes per lane, every load independent)
    p.selected_out = (int32_t *)base;
    p.n_selected_out = (int32_t *)(base + o_cnt);
    p.short_edges_out = (float *)(base + o_se);
    if (!ctx->d_rd_counts.ptr) {
        JV_TRY(ctx->d_rd_counts.reserve(2 * sizeof(unsigned long long)));
        JV_HIP_CHECK(hipMemsetAsync(ctx->d_rd_counts.ptr, 0, 2 * sizeof(unsigned long long), ctx->stream));
    }
    p.counts = (unsigned long long *)ctx->d_rd_counts.ptr;
    {
        ProfScope ps(ctx, R_PRUNE);
        JV_TRY(launch_retain_diverse(ctx->stream, ctx, p));
    }
    JV_HIP_CHECK(hipMemcpyAsync(selected_out, p.selected_out, sel_bytes, hipMemcpyDefault, ctx->stream));
    JV_HIP_CHECK(hipMemcpyAsync(n_selected_out, p.n_selected_out, cnt_bytes, hipMemcpyDefault, ctx->stream));
    if (short_edges_out) JV_HIP_CHECK(hipMemcpyAsync(short_edges_out, p.short_edges_out, sizeof(float) * (size_t)P, hipMemcpyDefault, ctx->stream));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return JV_OK;
}

int jv_hip_fused_build(jv_ctx *ctx, jv_fused *f, const jv_codes *codes, int64_t first, int64_t count, const int32_t *neighbors)
{
    clear_error();
    JV_REQUIRE(ctx && f && codes && neighbors, "fused_build: NULL argument");
    JV_REQUIRE(codes->pq == f->pq, "fused_build: the code store and the fused blocks use different codebooks");
    JV_REQUIRE(first >= 0 && count >= 0 && first + count <= f->count, "fused_build: range out of bounds");
    if (count == 0) return JV_OK;
    JV_TRY(use_device(ctx->device));
    int32_t *d_nb = f->d_neighbors + first * f->maxDegree;
    JV_HIP_CHECK(hipMemcpyAsync(d_nb, neighbors, sizeof(int32_t) * (size_t)count * f->maxDegree, hipMemcpyDefault, ctx->stream));
    if (!is_device_ptr(neighbors)) JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // the caller may reuse its buffer
    JV_TRY(launch_fused_gather(ctx->stream, codes, d_nb, f->maxDegree, count, f->d_blocks + (size_t)first * f->maxDegree * f->M));
    f->norms_valid = false;
    f->generation = next_fused_generation();
    return JV_OK;
}

int jv_hip_fused_download(jv_ctx *ctx, const jv_fused *f, int64_t first, int64_t count, uint8_t *blocks_out, int32_t *neighbors_out)
{
    clear_error();
    JV_REQUIRE(ctx && f, "fused_download: NULL argument");
    JV_REQUIRE(first >= 0 && count >= 0 && first + count <= f->count, "fused_download: range out of bounds");
    JV_TRY(use_device(ctx->device));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const size_t bsz = (size_t)f->maxDegree * f->M;
    if (blocks_out) JV_HIP_CHECK(hipMemcpy(blocks_out, f->d_blocks + (size_t)first * bsz, (size_t)count * bsz, hipMemcpyDefault));
    if (neighbors_out)
        JV_HIP_CHECK(hipMemcpy(neighbors_out, f->d_neighbors + first * f->maxDegree, sizeof(int32_t) * (size_t)count * f->maxDegree,
                               hipMemcpyDefault));
    return JV_OK;
}

int jv_hip_pq_decode(jv_ctx *ctx, const jv_codes *codes, const int32_t *ordinals, int64_t first, int64_t count, float *vectors_out)
{
    clear_error();
    JV_REQUIRE(ctx && codes, "pq_decode: NULL argument");
    JV_REQUIRE(count >= 0, "pq_decode: negative count");
    JV_REQUIRE(ordinals || (first >= 0 && first + count <= codes->count), "Ordinal range out of bounds");
    if (count == 0) return JV_OK;
    JV_REQUIRE(vectors_out, "pq_decode: NULL buffer");
    JV_TRY(use_device(ctx->device));
    const void *d_ord = nullptr;
    if (ordinals) JV_TRY(stage_in(ctx, ordinals, sizeof(int32_t) * (size_t)count, ctx->h_in, ctx->d_in, &d_ord));
    OutStage os;
    JV_TRY(stage_out_begin(ctx, vectors_out, sizeof(float) * (size_t)count * codes->pq->D, ctx->d_out, &os));
    JV_TRY(launch_pq_decode(ctx->stream, codes, (const int32_t *)d_ord, first, count, (float *)os.dev));
    return stage_out_end(ctx, os);
}

int jv_hip_direct_scores(jv_ctx *ctx, const jv_codes *codes, const float *queries, int Q, jv_vsf vsf, const int32_t *ordinals,
                         int B, float *scores_out)
{
    clear_error();
    JV_REQUIRE(ctx && codes, "direct_scores: NULL argument");
    JV_REQUIRE(vsf == JV_EUCLIDEAN || vsf == JV_DOT_PRODUCT || vsf == JV_COSINE, "Unsupported similarity function %d", (int)vsf);
    JV_REQUIRE(Q >= 0 && B >= 0, "direct_scores: negative sizes");
    if (Q == 0 || B == 0) return JV_OK;
    JV_REQUIRE(queries && ordinals && scores_out, "direct_scores: NULL buffer");
    JV_TRY(use_device(ctx->device));
    const jv_pq *pq = codes->pq;
    const void *d_q = nullptr, *d_ord = nullptr;
    JV_TRY(stage_in(ctx, queries, sizeof(float) * (size_t)Q * pq->D, ctx->h_in, ctx->d_in, &d_q));
    JV_TRY(stage_in(ctx, ordinals, sizeof(int32_t) * (size_t)Q * B, ctx->h_in, ctx->d_scratch2, &d_ord));
    // centred queries + their norms
    JV_TRY(ctx->d_scratch3.reserve(sizeof(float) * ((size_t)Q * pq->D + (size_t)Q) + 256));
    float *d_cq = (float *)ctx->d_scratch3.ptr;
    float *d_qnorm = d_cq + (size_t)Q * pq->D;
    JV_TRY(launch_center_queries(ctx->stream, pq, (const float *)d_q, Q, d_cq));
    OutStage os;
    JV_TRY(stage_out_begin(ctx, scores_out, sizeof(float) * (size_t)Q * B, ctx->d_out, &os));
    {
        ProfScope ps(ctx, R_ADC);
        JV_TRY(launch_direct_scores(ctx->stream, codes, to_kernel_vsf(vsf), d_cq, Q, (const int32_t *)d_ord, B, d_qnorm,
                                    (float *)os.dev));
    }
    return stage_out_end(ctx, os);
}

}  // extern "C"
