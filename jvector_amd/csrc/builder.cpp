// builder.cpp — batched Vamana construction behind the C ABI (jv_hip_builder_*; BASELINE config 5: "GPU-batched neighbor
// scoring + PQ encode", SURVEY §8 f.2 + Appendix C).
//
// The reference inserts nodes one by one from many threads (GraphIndexBuilder.addGraphNode, B/graph/GraphIndexBuilder.java:
// 605-659): search the current graph for the new node with the BuildScoreProvider's PQ score function (beamWidth candidates),
// robust-prune them (VamanaDiversityProvider.retainDiverse, B/graph/diversity/VamanaDiversityProvider.java:43-96), link, and
// backlink with a re-prune when a neighbour's list outgrows maxDegree x neighborOverflow (ConcurrentNeighborMap.insertDiverse /
// backlink, B/graph/ConcurrentNeighborMap.java:104-163, :298-322); cleanup() finally trims every list to maxDegree
// (enforceDegree, GraphIndexBuilder.java:472-508).  Its result depends on thread interleaving, so the contract here is the
// structure of the result (degree <= maxDegree, packed rows, no self loops / duplicates) and the recall of a search over it —
// not a bit-level trace.  What IS bit-exact is every score it consumes: the candidate search is the engine's GraphSearcher
// (device traversal over the builder's own mutable adjacency), the prune is jv_hip_retain_diverse's kernel, the backlink
// re-prune scores with the PQ pair table (ImmutablePQVectors.diversityFunctionFor).
//
// One insert_batch = the batch analogue of B concurrent addGraphNode calls that do not see each other:
//   1. queries = the batch's full-resolution vectors (gathered on the device)            launch_gather_rows
//   2. candidates = GraphSearcher.search(topK = rerankK = beam) on the graph so far       jv_hip_graph_search (device pointers)
//   3. robust prune of every candidate list                                              launch_retain_diverse
//   4. rows of the new nodes <- selection; back edges (target, source) emitted           bl_apply_selection
//   5. back edges sorted by (target, edge index)                                         launch_bl_sort_edges
//   6. per target: append while the row has room, else hand the merged list over          bl_backlink_merge
//   7. overflowed lists: PQ diversity scores, NodeArray order, robust prune, row rewrite  launch_pair_scores, bl_rank_sort,
//                                                                                        launch_retain_diverse, bl_rewrite_row
// Everything stays on the context's stream; the host reads back one counter per batch (how many lists overflowed).
//
// REFERENCE ORDER (context option bl_ref_order = 1): the lists additionally carry the score every entry was inserted under, stay
// in NodeArray order and keep ConcurrentNeighborMap's diverseBefore mark (bl_body.h "REFERENCE ORDER"); steps 4, 6 and 7 become
// insertDiverse / Neighbors.insert / retainDiverse(diverseBefore) as the reference performs them — nothing is re-scored or
// re-sorted — and a build whose batches hold ONE node is addGraphNode + cleanup's enforceDegree operation for operation: the
// adjacency equals the CPU checker's one-thread restatement of GraphIndexBuilder byte for byte (tests/test_builder_reference_order.py
// on the CPU mock, tests/test_zz_builder_reference_order_gpu.py on the MI355X).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <vector>

#include "bl_body.h"
#include "gs_params.h"
#include "jv_internal.h"
#include "rd_params.h"

using namespace jv;

struct jv_builder {
    int device = 0;
    const jv_pq *pq = nullptr;
    const jv_codes *codes = nullptr;
    const jv_vectors *vectors = nullptr;
    jv_vsf vsf = JV_EUCLIDEAN;
    int64_t n = 0;
    int Rf = 32, R = 40, beam = 100;
    float alpha = 1.2f;
    int32_t *d_nbrs = nullptr;       // [n][R]
    bool stored = false;             // the three arrays below exist, rows are NodeArray-ordered under the stored scores (sorted lists / reference order)
    bool use_db = false;             // stored: prunes of back-linked lists and enforceDegree start behind the diverseBefore mark (reference order: yes;
                                     // sorted lists: no, unless bl_sorted_lists = 2; bl_ref_order = 2 switches it off there — the two ablations of DESIGN.md §7)
    bool sym = false;                // stored && sym: SORTED LISTS — the classic path's symmetric PQ diversity scores, stored; every prune re-tests the
                                     // whole list (diverseBefore unused).  stored && !sym: REFERENCE ORDER
    float *d_nsc = nullptr;          // [n][R] the score each entry was inserted under
    int32_t *d_db = nullptr;         // [n] diverseBefore
    int hard_max = 0;                // (int) (neighborOverflow x maxDegree), capped at R: a list longer than this is pruned
    jv_graph *graph = nullptr;       // level 0 = d_nbrs, read in place by the device traversal
    jv_luts *luts = nullptr;
    int luts_cap = 0;
    jv_pair_table *tri = nullptr;
    int64_t inserted = 0;            // nodes that have a row (the search cannot return more than that)
    int32_t entry = -1;
    Buffer d_nodes, d_q, d_cand, d_csc, d_count, d_sel, d_nsel, d_keys, d_keys2, d_src, d_src2, d_sort_tmp, d_over_tgt, d_over_list, d_over_sc,
        d_sorted_ids, d_sorted_sc, d_ctr, d_imp_list, d_esc, d_over_db, d_over_n;
    double search_s = 0, prune_s = 0, backlink_s = 0;
    int64_t reprunes = 0, batches = 0, visited = 0, expanded = 0, improved = 0;  // (visited / expanded: SearchResult counters summed over the inserts)
    std::vector<int64_t> h_stats;
    ~jv_builder()
    {
        for (Buffer *b : {&d_nodes, &d_q, &d_cand, &d_csc, &d_count, &d_sel, &d_nsel, &d_keys, &d_keys2, &d_src, &d_src2, &d_sort_tmp, &d_over_tgt,
                          &d_over_list, &d_over_sc, &d_sorted_ids, &d_sorted_sc, &d_ctr, &d_imp_list, &d_esc, &d_over_db, &d_over_n})
            b->release();
    }
};

namespace {

constexpr long long kBlRefOrderDefault = 0;
constexpr long long kBlSortedListsDefault = 0;

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int run_retain(jv_ctx *ctx, jv_builder *b, const int32_t *d_cand, const float *d_sc, const int32_t *d_count, int P, int C, int32_t *d_sel,
               int32_t *d_nsel, const int32_t *d_before = nullptr)
{
    RdParams p{};
    p.tri = b->tri->d_tri;
    p.sq = pair_table_square(ctx, b->tri);
    p.codebooks = retain_diverse_table_free(ctx, b->pq) ? b->pq->d_codebooks : nullptr;
    p.codes = b->codes->d_codes;
    p.n = b->codes->count;
    p.cand_nodes = d_cand;
    p.cand_scores = d_sc;
    p.cand_count = d_count;
    p.diverse_before = d_before;
    p.P = P;
    p.C = C;
    p.M = b->codes->M;
    p.k = b->pq->k;
    p.vsf = to_kernel_vsf(b->vsf);
    p.maxDegree = b->Rf;
    p.alpha = b->alpha;
    p.chunk = retain_diverse_chunk(ctx);
    p.split = ctx_opt(ctx, "rd_split", 1) != 0 ? 1 : 0;
    p.wide_stage = ctx_opt(ctx, "rd_wide_stage", 1) != 0 ? 1 : 0;
    p.selected_out = d_sel;
    p.n_selected_out = d_nsel;
    p.short_edges_out = nullptr;
    if (!ctx->d_rd_counts.ptr) {   // (zeroed once per context: the counters accumulate; jv_hip_ctx_get_stat "rd_tests" / "rd_pairs" reads them)
        JV_TRY(ctx->d_rd_counts.reserve(2 * sizeof(unsigned long long)));
        JV_HIP_CHECK(hipMemsetAsync(ctx->d_rd_counts.ptr, 0, 2 * sizeof(unsigned long long), ctx->stream));
    }
    p.counts = (unsigned long long *)ctx->d_rd_counts.ptr;
    ProfScope ps(ctx, R_PRUNE);   // (its own region: the pair-score kernel of the backlinks stays under "adc")
    return launch_retain_diverse(ctx->stream, ctx, p);
}

// lists d_list [P][L] of targets d_tgt: score against the target (PQ diversity function), NodeArray order, robust prune, rewrite
int reprune_lists(jv_ctx *ctx, jv_builder *b, const int32_t *d_tgt, const int32_t *d_list, int P, int L)
{
    if (P == 0) return JV_OK;
    const size_t cells = (size_t)P * L;
    JV_TRY(b->d_over_sc.reserve(sizeof(float) * cells));
    JV_TRY(b->d_sorted_ids.reserve(sizeof(int32_t) * cells));
    JV_TRY(b->d_sorted_sc.reserve(sizeof(float) * cells));
    JV_TRY(b->d_count.reserve(sizeof(int32_t) * (size_t)P));
    JV_TRY(b->d_sel.reserve(sizeof(int32_t) * (size_t)P * b->Rf));
    JV_TRY(b->d_nsel.reserve(sizeof(int32_t) * (size_t)P));
    {
        ProfScope ps(ctx, R_ADC);
        JV_TRY(launch_pair_scores(ctx->stream, b->tri->d_tri, to_kernel_vsf(b->vsf), b->codes, d_tgt, P, d_list, L, (float *)b->d_over_sc.ptr));
    }
    BlSortParams sp{};
    sp.ids = d_list;
    sp.scores = (const float *)b->d_over_sc.ptr;
    sp.P = P;
    sp.L = L;
    sp.out_ids = (int32_t *)b->d_sorted_ids.ptr;
    sp.out_scores = (float *)b->d_sorted_sc.ptr;
    sp.out_count = (int32_t *)b->d_count.ptr;
    JV_TRY(launch_bl_rank_sort(ctx->stream, sp));
    JV_TRY(run_retain(ctx, b, sp.out_ids, sp.out_scores, sp.out_count, P, L, (int32_t *)b->d_sel.ptr, (int32_t *)b->d_nsel.ptr));
    BlRowsParams rp{};
    rp.tgt = d_tgt;
    rp.lst = sp.out_ids;
    rp.sel = (const int32_t *)b->d_sel.ptr;
    rp.P = P;
    rp.L = L;
    rp.Rf = b->Rf;
    rp.R = b->R;
    rp.nbrs = b->d_nbrs;
    JV_TRY(launch_bl_rewrite_rows(ctx->stream, rp));
    b->reprunes += P;
    return JV_OK;
}

// reference order: lists d_list / d_lsc [P][L] (already in NodeArray order, with the scores their entries were inserted under) of the
// targets d_tgt: retainDiverse(list, diverseBefore), the selection replaces the row, diverseBefore = size
int reprune_lists_ro(jv_ctx *ctx, jv_builder *b, const int32_t *d_tgt, const int32_t *d_list, const float *d_lsc, const int32_t *d_n,
                     const int32_t *d_before, int P, int L)
{
    if (P == 0) return JV_OK;
    JV_TRY(b->d_sel.reserve(sizeof(int32_t) * (size_t)P * b->Rf));
    JV_TRY(b->d_nsel.reserve(sizeof(int32_t) * (size_t)P));
    JV_TRY(run_retain(ctx, b, d_list, d_lsc, d_n, P, L, (int32_t *)b->d_sel.ptr, (int32_t *)b->d_nsel.ptr, d_before));
    BlRoRowsParams rp{};
    rp.tgt = d_tgt;
    rp.lst = d_list;
    rp.lsc = d_lsc;
    rp.sel = (const int32_t *)b->d_sel.ptr;
    rp.P = P;
    rp.L = L;
    rp.Rf = b->Rf;
    rp.R = b->R;
    rp.nbrs = b->d_nbrs;
    rp.nsc = b->d_nsc;
    rp.db = b->d_db;
    JV_TRY(launch_bl_ro_rewrite_rows(ctx->stream, rp));
    b->reprunes += P;
    return JV_OK;
}

int read_counter(jv_ctx *ctx, jv_builder *b, unsigned int *out)
{
    JV_TRY(ctx->h_out.reserve(64));
    JV_HIP_CHECK(hipMemcpyAsync(ctx->h_out.ptr, b->d_ctr.ptr, sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *out = *(const unsigned int *)ctx->h_out.ptr;
    return JV_OK;
}

}  // namespace

extern "C" {

int jv_hip_builder_create(jv_ctx *ctx, const jv_pq *pq, const jv_codes *codes, const jv_vectors *vectors, jv_vsf vsf, int max_degree,
                          int beam_width, float alpha, float neighbor_overflow, jv_builder **out)
{
    clear_error();
    JV_REQUIRE(ctx && pq && codes && vectors && out, "builder_create: NULL argument");
    *out = nullptr;
    JV_FLOAT_ROWS(vectors, "builder_create");
    JV_REQUIRE(vsf == JV_EUCLIDEAN || vsf == JV_DOT_PRODUCT || vsf == JV_COSINE, "Unsupported similarity function %d", (int)vsf);
    JV_REQUIRE(codes->pq == pq, "builder_create: the code store belongs to another quantizer");
    JV_REQUIRE(vectors->D == pq->D && vectors->count >= codes->count, "builder_create: vectors do not match the codes (%lld x %d vs %lld x %d)",
               (long long)vectors->count, vectors->D, (long long)codes->count, pq->D);
    JV_REQUIRE(codes->count >= 1 && codes->count <= 0x7fffffffLL, "builder_create: %lld nodes", (long long)codes->count);
    JV_REQUIRE(max_degree >= 2 && max_degree <= 64, "builder_create: maxDegree %d outside 2..64", max_degree);
    JV_REQUIRE(beam_width >= 1 && beam_width <= 4096, "builder_create: beamWidth %d outside 1..4096", beam_width);
    JV_REQUIRE(alpha == alpha && alpha >= 1.0f && alpha <= 64.0f, "builder_create: alpha must lie in [1, 64]");
    JV_REQUIRE(neighbor_overflow == neighbor_overflow && neighbor_overflow >= 1.0f && neighbor_overflow <= 8.0f, "builder_create: neighborOverflow must lie in [1, 8]");
    JV_TRY(use_device(ctx->device));
    jv_builder *b = new jv_builder();
    b->device = ctx->device;
    b->pq = pq;
    b->codes = codes;
    b->vectors = vectors;
    b->vsf = vsf;
    b->n = codes->count;
    b->Rf = max_degree;
    b->R = std::max(max_degree, std::min(64, (int)(max_degree * neighbor_overflow)));  // the working row width (ConcurrentNeighborMap.java:298-322)
    b->beam = beam_width;
    b->alpha = alpha;
    auto fail = [&](int rc) {
        jv_hip_builder_destroy(b);
        return rc;
    };
    if (hipMalloc((void **)&b->d_nbrs, sizeof(int32_t) * (size_t)b->n * b->R) != hipSuccess) {
        (void)hipGetLastError();
        set_error("builder_create: cannot allocate the %lld x %d adjacency", (long long)b->n, b->R);
        return fail(JV_ERR_OOM);
    }
    if (hipMemsetAsync(b->d_nbrs, 0xFF, sizeof(int32_t) * (size_t)b->n * b->R, ctx->stream) != hipSuccess) return fail(JV_ERR_HIP);
    const long long ref_opt = ctx_opt(ctx, "bl_ref_order", kBlRefOrderDefault), sorted_opt = ctx_opt(ctx, "bl_sorted_lists", kBlSortedListsDefault);
    const bool ref_order = ref_opt != 0;
    b->sym = !ref_order && sorted_opt != 0;
    b->stored = ref_order || b->sym;
    b->use_db = ref_order ? ref_opt != 2 : sorted_opt == 2;
    b->hard_max = std::min(b->R, (int)(neighbor_overflow * (float)max_degree));   // Neighbors.insert :270
    if (b->stored) {
        if (hipMalloc((void **)&b->d_nsc, sizeof(float) * (size_t)b->n * b->R) != hipSuccess ||
            hipMalloc((void **)&b->d_db, sizeof(int32_t) * (size_t)b->n) != hipSuccess) {
            (void)hipGetLastError();
            set_error("builder_create: cannot allocate the %lld x %d score rows", (long long)b->n, b->R);
            return fail(JV_ERR_OOM);
        }
        if (hipMemsetAsync(b->d_nsc, 0, sizeof(float) * (size_t)b->n * b->R, ctx->stream) != hipSuccess ||
            hipMemsetAsync(b->d_db, 0, sizeof(int32_t) * (size_t)b->n, ctx->stream) != hipSuccess)
            return fail(JV_ERR_HIP);
    }
    int rc = jv_hip_graph_create(ctx, b->n, 1, &b->graph);
    if (rc == JV_OK) rc = jv_hip_graph_set_level0_device(ctx, b->graph, b->d_nbrs, b->R);
    if (rc == JV_OK) rc = jv_hip_graph_set_traversal(b->graph, JV_TRAVERSAL_DEVICE);
    if (rc == JV_OK) rc = jv_hip_pair_table_create(ctx, pq, vsf, &b->tri);
    if (rc == JV_OK) rc = b->d_ctr.reserve(256);
    if (rc != JV_OK) return fail(rc);
    *out = b;
    return JV_OK;
}

int jv_hip_builder_destroy(jv_builder *b)
{
    if (!b) return JV_OK;
    (void)hipSetDevice(b->device);
    if (b->luts) jv_hip_luts_destroy(b->luts);
    if (b->graph) jv_hip_graph_destroy(b->graph);
    if (b->tri) jv_hip_pair_table_destroy(b->tri);
    if (b->d_nbrs) (void)hipFree(b->d_nbrs);
    if (b->d_nsc) (void)hipFree(b->d_nsc);
    if (b->d_db) (void)hipFree(b->d_db);
    delete b;
    return JV_OK;
}

int jv_hip_builder_seed(jv_ctx *ctx, jv_builder *b, int32_t node)
{
    clear_error();
    JV_REQUIRE(ctx && b, "builder_seed: NULL argument");
    JV_REQUIRE(node >= 0 && node < b->n, "builder_seed: node %d outside [0, %lld)", node, (long long)b->n);
    JV_REQUIRE(b->inserted == 0, "builder_seed: the graph already has nodes");
    b->entry = node;
    b->inserted = 1;  // its (empty) row exists; the first batch links to it
    return jv_hip_graph_set_entry(b->graph, node, 0);
}

// ---- shared steps of insert_batch / improve_batch ----
static int check_batch(jv_ctx *ctx, jv_builder *b, const int32_t *nodes, int B, const char *what)
{
    // The caller's ids index the adjacency rows, the code rows and the vector rows unguarded further down (bl_apply_selection,
    // bl_pack_row, the back-edge emission) and two inserts of one id would race on one row: an id outside every one of them, or
    // listed twice, is refused here.  A batch is at most a few hundred KB: a host copy and a sort cost nothing next
    // to the batch's searches.
    JV_REQUIRE((long long)B * b->Rf <= 0x7fffffffll, "%s: %d nodes x %d working slots exceed the edge sorter's 32-bit count; split the batch", what, B, b->Rf);
    std::vector<int32_t> h((size_t)B);
    JV_HIP_CHECK(hipMemcpyAsync(h.data(), nodes, sizeof(int32_t) * (size_t)B, hipMemcpyDefault, ctx->stream));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const long long limit = std::min<long long>({(long long)b->n, (long long)b->codes->count, (long long)b->vectors->count});
    for (int i = 0; i < B; ++i)
        JV_REQUIRE(h[(size_t)i] >= 0 && h[(size_t)i] < limit, "%s: node id %d (position %d) outside [0, %lld)", what, h[(size_t)i], i, limit);
    std::sort(h.begin(), h.end());
    for (int i = 1; i < B; ++i) JV_REQUIRE(h[(size_t)i] != h[(size_t)i - 1], "%s: node id %d appears twice in the batch", what, h[(size_t)i]);
    return JV_OK;
}

// 1 + 2: the batch's vectors as queries, GraphSearcher.search(topK = rerankK = k) over the graph built so far -> d_cand / d_csc [B][k]
static int search_candidates(jv_ctx *ctx, jv_builder *b, const int32_t *d_nodes, int B, int k, bool exclude_self = false)
{
    const int D = b->pq->D;
    const int search_chunk = 65536;
    if (!b->luts || b->luts_cap < std::min(B, search_chunk)) {
        if (b->luts) jv_hip_luts_destroy(b->luts);
        b->luts = nullptr;
        b->luts_cap = std::min(search_chunk, std::max(1024, 2 * B));
        JV_TRY(jv_hip_luts_create(ctx, b->pq, b->luts_cap, &b->luts));
    }
    JV_TRY(b->d_cand.reserve(sizeof(int32_t) * (size_t)B * k));
    JV_TRY(b->d_csc.reserve(sizeof(float) * (size_t)B * k));
    JV_TRY(b->d_q.reserve(sizeof(float) * (size_t)std::min(B, search_chunk) * D));
    int32_t *d_cand = (int32_t *)b->d_cand.ptr;
    float *d_csc = (float *)b->d_csc.ptr;
    const double t0 = now_s();
    for (int s = 0; s < B; s += search_chunk) {
        const int bc = std::min(search_chunk, B - s);
        JV_TRY(launch_gather_rows(ctx->stream, b->vectors->d_vecs, b->vectors->count, D, d_nodes + s, bc, (float *)b->d_q.ptr, nullptr, 0));
        b->h_stats.resize(2 * (size_t)bc);
        if (exclude_self)   // acceptOrds = ExcludingBits(node) (improveConnections :518): the node is traversed, never returned
            JV_TRY(graph_search_excluding(ctx, b->graph, b->luts, b->codes, (const float *)b->d_q.ptr, bc, b->vsf, k, k, d_nodes + s,
                                          d_cand + (size_t)s * k, d_csc + (size_t)s * k, b->h_stats.data()));
        else
            JV_TRY(jv_hip_graph_search(ctx, b->graph, b->luts, b->codes, nullptr, nullptr, (const float *)b->d_q.ptr, bc, b->vsf, k, k,
                                       d_cand + (size_t)s * k, d_csc + (size_t)s * k, b->h_stats.data()));
        for (int q = 0; q < bc; ++q) {
            b->visited += b->h_stats[2 * (size_t)q];
            b->expanded += b->h_stats[2 * (size_t)q + 1];
        }
    }
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    b->search_s += now_s() - t0;
    return JV_OK;
}

// 5 - 7: the E back edges in d_keys / d_src: sorted by (target, edge index), appended while rows have room, overflowed lists re-pruned
static int link_back_edges(jv_ctx *ctx, jv_builder *b, long long E)
{
    const int R = b->R;
    const double t0 = now_s();
    size_t tmp_bytes = 0;
    JV_TRY(launch_bl_sort_edges(ctx->stream, nullptr, &tmp_bytes, nullptr, nullptr, nullptr, nullptr, E, 64));
    JV_TRY(b->d_sort_tmp.reserve(tmp_bytes + 256));
    JV_TRY(launch_bl_sort_edges(ctx->stream, b->d_sort_tmp.ptr, &tmp_bytes, (const unsigned long long *)b->d_keys.ptr,
                                (unsigned long long *)b->d_keys2.ptr, (const int32_t *)b->d_src.ptr, (int32_t *)b->d_src2.ptr, E, 64));
    const int Knew = 2 * R, L = R + Knew;
    const unsigned int over_cap = (unsigned int)std::min<long long>(E, b->n);  // at most one overflow per distinct target
    JV_TRY(b->d_over_tgt.reserve(sizeof(int32_t) * (size_t)over_cap));
    JV_TRY(b->d_over_list.reserve(sizeof(int32_t) * (size_t)over_cap * L));
    JV_HIP_CHECK(hipMemsetAsync(b->d_ctr.ptr, 0, sizeof(unsigned int), ctx->stream));
    BlMergeParams mp{};
    mp.keys = (const unsigned long long *)b->d_keys2.ptr;
    mp.src = (const int32_t *)b->d_src2.ptr;
    mp.E = E;
    mp.R = R;
    mp.Knew = Knew;
    mp.nbrs = b->d_nbrs;
    mp.over_tgt = (int32_t *)b->d_over_tgt.ptr;
    mp.over_list = (int32_t *)b->d_over_list.ptr;
    mp.over_count = (unsigned int *)b->d_ctr.ptr;
    mp.over_cap = over_cap;
    JV_TRY(launch_bl_backlink_merge(ctx->stream, mp));
    unsigned int n_over = 0;
    JV_TRY(read_counter(ctx, b, &n_over));
    n_over = std::min(n_over, over_cap);
    JV_TRY(reprune_lists(ctx, b, mp.over_tgt, mp.over_list, (int)n_over, L));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    b->backlink_s += now_s() - t0;
    return JV_OK;
}

// reference order: the E back edges (keys / src / scores in d_keys / d_src / d_esc) -> Neighbors.insert per target, in batch order
static int link_back_edges_ro(jv_ctx *ctx, jv_builder *b, long long E, int dedupe_ids)
{
    const int R = b->R;
    const double t0 = now_s();
    size_t tmp_bytes = 0;
    JV_TRY(launch_bl_sort_edges(ctx->stream, nullptr, &tmp_bytes, nullptr, nullptr, nullptr, nullptr, E, 64));
    JV_TRY(b->d_sort_tmp.reserve(tmp_bytes + 256));
    JV_TRY(launch_bl_sort_edges(ctx->stream, b->d_sort_tmp.ptr, &tmp_bytes, (const unsigned long long *)b->d_keys.ptr,
                                (unsigned long long *)b->d_keys2.ptr, (const int32_t *)b->d_src.ptr, (int32_t *)b->d_src2.ptr, E, 64));
    const int Knew = 2 * R, L = R + Knew;
    static_assert(BL_RO_MAX_LIST >= 3 * 64, "the merge step's working list holds R + 2 R entries, R <= 64");
    const unsigned int over_cap = (unsigned int)std::min<long long>(E, b->n);
    JV_TRY(b->d_over_tgt.reserve(sizeof(int32_t) * (size_t)over_cap));
    JV_TRY(b->d_over_db.reserve(sizeof(int32_t) * (size_t)over_cap));
    JV_TRY(b->d_over_n.reserve(sizeof(int32_t) * (size_t)over_cap));
    JV_TRY(b->d_over_list.reserve(sizeof(int32_t) * (size_t)over_cap * L));
    JV_TRY(b->d_over_sc.reserve(sizeof(float) * (size_t)over_cap * L));
    JV_HIP_CHECK(hipMemsetAsync(b->d_ctr.ptr, 0, sizeof(unsigned int), ctx->stream));
    BlRoMergeParams mp{};
    mp.keys = (const unsigned long long *)b->d_keys2.ptr;
    mp.src = (const int32_t *)b->d_src.ptr;
    mp.esc = (const float *)b->d_esc.ptr;
    mp.E = E;
    mp.R = R;
    mp.hard_max = b->hard_max;
    mp.Knew = Knew;
    mp.dedupe_ids = dedupe_ids;
    mp.nbrs = b->d_nbrs;
    mp.nsc = b->d_nsc;
    mp.db = b->d_db;
    mp.over_tgt = (int32_t *)b->d_over_tgt.ptr;
    mp.over_list = (int32_t *)b->d_over_list.ptr;
    mp.over_sc = (float *)b->d_over_sc.ptr;
    mp.over_db = (int32_t *)b->d_over_db.ptr;
    mp.over_n = (int32_t *)b->d_over_n.ptr;
    mp.over_count = (unsigned int *)b->d_ctr.ptr;
    mp.over_cap = over_cap;
    JV_TRY(launch_bl_ro_backlink_merge(ctx->stream, mp));
    unsigned int n_over = 0;
    JV_TRY(read_counter(ctx, b, &n_over));
    n_over = std::min(n_over, over_cap);
    JV_TRY(reprune_lists_ro(ctx, b, mp.over_tgt, mp.over_list, mp.over_sc, mp.over_n, b->use_db ? mp.over_db : nullptr, (int)n_over, L));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    b->backlink_s += now_s() - t0;
    return JV_OK;
}

static int reserve_edges(jv_builder *b, long long E)
{
    JV_TRY(b->d_keys.reserve(sizeof(unsigned long long) * (size_t)E));
    JV_TRY(b->d_keys2.reserve(sizeof(unsigned long long) * (size_t)E));
    JV_TRY(b->d_src.reserve(sizeof(int32_t) * (size_t)E));
    JV_TRY(b->d_src2.reserve(sizeof(int32_t) * (size_t)E));
    return JV_OK;
}

int jv_hip_builder_insert_batch(jv_ctx *ctx, jv_builder *b, const int32_t *nodes, int B)
{
    clear_error();
    JV_REQUIRE(ctx && b, "builder_insert_batch: NULL argument");
    JV_REQUIRE(B >= 0, "builder_insert_batch: negative batch");
    if (B == 0) return JV_OK;
    JV_REQUIRE(nodes, "builder_insert_batch: NULL nodes");
    JV_REQUIRE(b->entry >= 0, "builder_insert_batch: seed the graph first (jv_hip_builder_seed)");
    JV_REQUIRE(ctx->device == b->device, "builder_insert_batch: the builder lives on device %d", b->device);
    JV_TRY(use_device(ctx->device));
    const int Rf = b->Rf, R = b->R;
    const int k = (int)std::min<int64_t>(b->beam, b->inserted);  // cannot ask for more candidates than the graph holds
    JV_TRY(check_batch(ctx, b, nodes, B, "builder_insert_batch"));
    JV_TRY(b->d_nodes.reserve(sizeof(int32_t) * (size_t)B));
    JV_HIP_CHECK(hipMemcpyAsync(b->d_nodes.ptr, nodes, sizeof(int32_t) * (size_t)B, hipMemcpyDefault, ctx->stream));
    if (!is_device_ptr(nodes)) JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // the caller may reuse its buffer
    const int32_t *d_nodes = (const int32_t *)b->d_nodes.ptr;

    // ---- 1 + 2. candidate search on the graph built so far ----
    JV_TRY(search_candidates(ctx, b, d_nodes, B, k));
    const int32_t *d_cand = (const int32_t *)b->d_cand.ptr;
    const float *d_csc = (const float *)b->d_csc.ptr;

    // ---- 3 + 4. robust prune of every new node's candidates (best first), rows + back edges ----
    double t0 = now_s();
    JV_TRY(b->d_count.reserve(sizeof(int32_t) * (size_t)B));
    JV_TRY(b->d_sel.reserve(sizeof(int32_t) * (size_t)B * Rf));
    JV_TRY(b->d_nsel.reserve(sizeof(int32_t) * (size_t)B));
    JV_TRY(launch_bl_count_valid(ctx->stream, d_cand, k, (int32_t *)b->d_count.ptr, B));
    JV_TRY(run_retain(ctx, b, d_cand, d_csc, (const int32_t *)b->d_count.ptr, B, k, (int32_t *)b->d_sel.ptr, (int32_t *)b->d_nsel.ptr));
    const long long E = (long long)B * Rf;
    JV_TRY(reserve_edges(b, E));
    if (b->sym) {   // sorted lists: the row under the symmetric scores (what the classic path's first re-prune of it would compute), sorted
        JV_TRY(b->d_esc.reserve(sizeof(float) * (size_t)E));
        JV_TRY(b->d_imp_list.reserve(sizeof(int32_t) * (size_t)E));
        JV_TRY(b->d_over_sc.reserve(sizeof(float) * (size_t)E));
        JV_TRY(b->d_sorted_ids.reserve(sizeof(int32_t) * (size_t)E));
        JV_TRY(b->d_sorted_sc.reserve(sizeof(float) * (size_t)E));
        JV_TRY(b->d_over_n.reserve(sizeof(int32_t) * (size_t)B));
        BlSelIdsParams ip{};
        ip.nodes = d_nodes;
        ip.cand = d_cand;
        ip.sel = (const int32_t *)b->d_sel.ptr;
        ip.B = B;
        ip.C = k;
        ip.Rf = Rf;
        ip.out_ids = (int32_t *)b->d_imp_list.ptr;
        JV_TRY(launch_bl_sel_ids(ctx->stream, ip));
        {
            ProfScope ps(ctx, R_ADC);
            JV_TRY(launch_pair_scores(ctx->stream, b->tri->d_tri, to_kernel_vsf(b->vsf), b->codes, d_nodes, B, ip.out_ids, Rf, (float *)b->d_over_sc.ptr));
        }
        BlSortParams sp{};
        sp.ids = ip.out_ids;
        sp.scores = (const float *)b->d_over_sc.ptr;
        sp.P = B;
        sp.L = Rf;
        sp.out_ids = (int32_t *)b->d_sorted_ids.ptr;
        sp.out_scores = (float *)b->d_sorted_sc.ptr;
        sp.out_count = (int32_t *)b->d_over_n.ptr;
        JV_TRY(launch_bl_rank_sort(ctx->stream, sp));
        BlRoApplySortedParams rp{};
        rp.nodes = d_nodes;
        rp.ids = sp.out_ids;
        rp.sc = sp.out_scores;
        rp.B = B;
        rp.Rf = Rf;
        rp.R = R;
        rp.nbrs = b->d_nbrs;
        rp.nsc = b->d_nsc;
        rp.db = b->d_db;
        rp.edge_keys = (unsigned long long *)b->d_keys.ptr;
        rp.edge_src = (int32_t *)b->d_src.ptr;
        rp.edge_sc = (float *)b->d_esc.ptr;
        JV_TRY(launch_bl_ro_apply_sorted(ctx->stream, rp));
        JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        b->prune_s += now_s() - t0;
        JV_TRY(link_back_edges_ro(ctx, b, E, 1));
        b->inserted += B;
        b->batches += 1;
        return JV_OK;
    }
    if (b->stored) {   // insertDiverse on the new nodes' empty lists, then backlink -> Neighbors.insert
        JV_TRY(b->d_esc.reserve(sizeof(float) * (size_t)E));
        BlRoApplyParams rp{};
        rp.nodes = d_nodes;
        rp.cand = d_cand;
        rp.cand_sc = d_csc;
        rp.sel = (const int32_t *)b->d_sel.ptr;
        rp.B = B;
        rp.C = k;
        rp.Rf = Rf;
        rp.R = R;
        rp.nbrs = b->d_nbrs;
        rp.nsc = b->d_nsc;
        rp.db = b->d_db;
        rp.edge_keys = (unsigned long long *)b->d_keys.ptr;
        rp.edge_src = (int32_t *)b->d_src.ptr;
        rp.edge_sc = (float *)b->d_esc.ptr;
        JV_TRY(launch_bl_ro_apply_selection(ctx->stream, rp));
        JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        b->prune_s += now_s() - t0;
        // (dedupe: a fresh node is in nobody's list, so this changes nothing for an insert — it keeps a RE-insertion of a node the
        //  graph already holds, build_vamana's `passes` experiment, from listing it twice under two scores)
        JV_TRY(link_back_edges_ro(ctx, b, E, 1));
        b->inserted += B;
        b->batches += 1;
        return JV_OK;
    }
    BlApplyParams ap{};
    ap.nodes = d_nodes;
    ap.cand = d_cand;
    ap.sel = (const int32_t *)b->d_sel.ptr;
    ap.B = B;
    ap.C = k;
    ap.Rf = Rf;
    ap.R = R;
    ap.nbrs = b->d_nbrs;
    ap.edge_keys = (unsigned long long *)b->d_keys.ptr;
    ap.edge_src = (int32_t *)b->d_src.ptr;
    JV_TRY(launch_bl_apply_selection(ctx->stream, ap));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    b->prune_s += now_s() - t0;

    // ---- 5 - 7. backlinks ----
    JV_TRY(link_back_edges(ctx, b, E));
    b->inserted += B;
    b->batches += 1;
    return JV_OK;
}

// improveConnections for a batch of nodes that are IN the graph (GraphIndexBuilder.java:510-560, called by cleanup() :472-508 — the
// reference runs it for the nodes of the upper layers and calls a pass over every node "empirically unnecessary"; here any node may be
// given): search the finished graph for the node (beamWidth candidates, PQ scores), MERGE them with the neighbours it already has
// (ConcurrentNeighborMap.insertDiverse :104-163 — a re-insertion that REPLACED the row measured worse), score the merged
// list against the node with the PQ diversity function, robust-prune it to maxDegree, rewrite the row, then backlink the row's members
// like an insert does (an edge its target already holds is dropped by the merge step).
int jv_hip_builder_improve_batch(jv_ctx *ctx, jv_builder *b, const int32_t *nodes, int B)
{
    clear_error();
    JV_REQUIRE(ctx && b, "builder_improve_batch: NULL argument");
    JV_REQUIRE(B >= 0, "builder_improve_batch: negative batch");
    if (B == 0) return JV_OK;
    JV_REQUIRE(nodes, "builder_improve_batch: NULL nodes");
    JV_REQUIRE(b->entry >= 0 && b->inserted >= 2, "builder_improve_batch: nothing to improve in an empty graph");

    JV_REQUIRE(ctx->device == b->device, "builder_improve_batch: the builder lives on device %d", b->device);
    JV_TRY(use_device(ctx->device));
    const int Rf = b->Rf, R = b->R;
    const int k = (int)std::min<int64_t>(b->beam, b->inserted);
    {   // the merged lists (row + beam candidates) must fit the robust prune's LDS block: refused up front, not after the searches
        const size_t need = retain_diverse_lds_bytes(R + k, b->codes->M);
        JV_REQUIRE(need <= ctx->lds_per_block, "builder_improve_batch: lists of %d + %d candidates x %d code bytes need %zu bytes of LDS (limit %zu); lower the beam width",
                   R, k, b->codes->M, need, ctx->lds_per_block);
    }
    JV_TRY(check_batch(ctx, b, nodes, B, "builder_improve_batch"));
    JV_TRY(b->d_nodes.reserve(sizeof(int32_t) * (size_t)B));
    JV_HIP_CHECK(hipMemcpyAsync(b->d_nodes.ptr, nodes, sizeof(int32_t) * (size_t)B, hipMemcpyDefault, ctx->stream));
    if (!is_device_ptr(nodes)) JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const int32_t *d_nodes = (const int32_t *)b->d_nodes.ptr;
    JV_TRY(search_candidates(ctx, b, d_nodes, B, k, b->stored && !b->sym));

    double t0 = now_s();
    const int L = R + k;
    if (b->stored) {   // insertDiverse(merge(list, candidates)) with the stored / search scores, then backlink every member of the new list
        JV_TRY(b->d_imp_list.reserve(sizeof(int32_t) * (size_t)B * L));
        JV_TRY(b->d_over_sc.reserve(sizeof(float) * (size_t)B * L));
        JV_TRY(b->d_over_n.reserve(sizeof(int32_t) * (size_t)B));
        const int32_t *cand_ids = (const int32_t *)b->d_cand.ptr;
        const float *cand_sc = (const float *)b->d_csc.ptr;
        if (b->sym) {   // the candidates under the symmetric score against their node, in NodeArray order (the classic path scores and sorts row + candidates)
            JV_TRY(b->d_sorted_ids.reserve(sizeof(int32_t) * (size_t)B * k));
            JV_TRY(b->d_sorted_sc.reserve(sizeof(float) * (size_t)B * k));
            {
                ProfScope ps(ctx, R_ADC);
                JV_TRY(launch_pair_scores(ctx->stream, b->tri->d_tri, to_kernel_vsf(b->vsf), b->codes, d_nodes, B, cand_ids, k, (float *)b->d_over_sc.ptr));
            }
            BlSortParams sp{};
            sp.ids = cand_ids;
            sp.scores = (const float *)b->d_over_sc.ptr;
            sp.P = B;
            sp.L = k;
            sp.out_ids = (int32_t *)b->d_sorted_ids.ptr;
            sp.out_scores = (float *)b->d_sorted_sc.ptr;
            sp.out_count = (int32_t *)b->d_over_n.ptr;
            JV_TRY(launch_bl_rank_sort(ctx->stream, sp));
            cand_ids = sp.out_ids;
            cand_sc = sp.out_scores;
        }
        BlRoImproveParams ip{};
        ip.nodes = d_nodes;
        ip.cand = cand_ids;
        ip.cand_sc = cand_sc;
        ip.skip_empty = b->sym ? 0 : 1;
        ip.B = B;
        ip.C = k;
        ip.R = R;
        ip.nbrs = b->d_nbrs;
        ip.nsc = b->d_nsc;
        ip.list = (int32_t *)b->d_imp_list.ptr;
        ip.lsc = (float *)b->d_over_sc.ptr;
        ip.ln = (int32_t *)b->d_over_n.ptr;
        JV_TRY(launch_bl_ro_improve_list(ctx->stream, ip));
        JV_TRY(reprune_lists_ro(ctx, b, d_nodes, ip.list, ip.lsc, ip.ln, nullptr, B, L));
        const long long E = (long long)B * Rf;
        JV_TRY(reserve_edges(b, E));
        JV_TRY(b->d_esc.reserve(sizeof(float) * (size_t)E));
        BlRoRowEdgesParams ep{};
        ep.nodes = d_nodes;
        ep.B = B;
        ep.Rf = Rf;
        ep.R = R;
        ep.nbrs = b->d_nbrs;
        ep.nsc = b->d_nsc;
        ep.edge_keys = (unsigned long long *)b->d_keys.ptr;
        ep.edge_src = (int32_t *)b->d_src.ptr;
        ep.edge_sc = (float *)b->d_esc.ptr;
        JV_TRY(launch_bl_ro_row_edges(ctx->stream, ep));
        JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        b->prune_s += now_s() - t0;
        JV_TRY(link_back_edges_ro(ctx, b, E, 1));
        b->batches += 1;
        b->improved += B;
        return JV_OK;
    }
    JV_TRY(b->d_imp_list.reserve(sizeof(int32_t) * (size_t)B * L));
    BlImproveParams ip{};
    ip.nodes = d_nodes;
    ip.cand = (const int32_t *)b->d_cand.ptr;
    ip.B = B;
    ip.C = k;
    ip.R = R;
    ip.nbrs = b->d_nbrs;
    ip.list = (int32_t *)b->d_imp_list.ptr;
    JV_TRY(launch_bl_improve_list(ctx->stream, ip));
    JV_TRY(reprune_lists(ctx, b, d_nodes, ip.list, B, L));
    const long long E = (long long)B * Rf;
    JV_TRY(reserve_edges(b, E));
    BlRowEdgesParams ep{};
    ep.nodes = d_nodes;
    ep.B = B;
    ep.Rf = Rf;
    ep.R = R;
    ep.nbrs = b->d_nbrs;
    ep.edge_keys = (unsigned long long *)b->d_keys.ptr;
    ep.edge_src = (int32_t *)b->d_src.ptr;
    JV_TRY(launch_bl_row_edges(ctx->stream, ep));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    b->prune_s += now_s() - t0;
    JV_TRY(link_back_edges(ctx, b, E));
    b->batches += 1;
    b->improved += B;
    return JV_OK;
}

int jv_hip_builder_finish(jv_ctx *ctx, jv_builder *b, int32_t *neighbors_out)
{
    clear_error();
    JV_REQUIRE(ctx && b, "builder_finish: NULL argument");
    JV_REQUIRE(ctx->device == b->device, "builder_finish: the builder lives on device %d", b->device);
    JV_TRY(use_device(ctx->device));
    const double t0 = now_s();
    if (b->R > b->Rf) {  // GraphIndexBuilder.cleanup -> enforceDegree: lists still above maxDegree are pruned back
        JV_TRY(b->d_over_tgt.reserve(sizeof(int32_t) * (size_t)b->n));
        JV_HIP_CHECK(hipMemsetAsync(b->d_ctr.ptr, 0, sizeof(unsigned int), ctx->stream));
        BlOverParams op{};
        op.nbrs = b->d_nbrs;
        op.N = b->n;
        op.R = b->R;
        op.Rf = b->Rf;
        op.over_tgt = (int32_t *)b->d_over_tgt.ptr;
        op.over_count = (unsigned int *)b->d_ctr.ptr;
        op.over_cap = (unsigned int)b->n;
        JV_TRY(launch_bl_list_over_degree(ctx->stream, op));
        unsigned int n_over = 0;
        JV_TRY(read_counter(ctx, b, &n_over));
        const int piece = 1 << 20;
        JV_TRY(b->d_over_list.reserve(sizeof(int32_t) * (size_t)std::min<unsigned int>(n_over, piece) * b->R));
        if (b->stored) {   // enforceDegree: retainDiverse(copy, diverseBefore) over the stored scores (ConcurrentNeighborMap.java:190-200)
            const size_t pc = (size_t)std::min<unsigned int>(n_over, piece);
            JV_TRY(b->d_over_sc.reserve(sizeof(float) * pc * b->R));
            JV_TRY(b->d_over_db.reserve(sizeof(int32_t) * pc));
            JV_TRY(b->d_over_n.reserve(sizeof(int32_t) * pc));
        }
        for (unsigned int s = 0; s < n_over; s += piece) {
            const int P = (int)std::min<unsigned int>(piece, n_over - s);
            const int32_t *tgt = (const int32_t *)b->d_over_tgt.ptr + s;
            if (b->stored) {
                BlRoCopyParams cp{};
                cp.tgt = tgt;
                cp.P = P;
                cp.R = b->R;
                cp.nbrs = b->d_nbrs;
                cp.nsc = b->d_nsc;
                cp.db = b->d_db;
                cp.lst = (int32_t *)b->d_over_list.ptr;
                cp.lsc = (float *)b->d_over_sc.ptr;
                cp.ldb = (int32_t *)b->d_over_db.ptr;
                cp.ln = (int32_t *)b->d_over_n.ptr;
                JV_TRY(launch_bl_ro_copy_rows(ctx->stream, cp));
                JV_TRY(reprune_lists_ro(ctx, b, tgt, cp.lst, cp.lsc, cp.ln, b->use_db ? cp.ldb : nullptr, P, b->R));
                continue;
            }
            JV_TRY(launch_bl_copy_rows(ctx->stream, b->d_nbrs, b->R, tgt, P, (int32_t *)b->d_over_list.ptr));
            JV_TRY(reprune_lists(ctx, b, tgt, (const int32_t *)b->d_over_list.ptr, P, b->R));
        }
    }
    if (neighbors_out) {
        OutStage os;
        JV_TRY(stage_out_begin(ctx, neighbors_out, sizeof(int32_t) * (size_t)b->n * b->Rf, ctx->d_out, &os));
        JV_TRY(launch_bl_strided_copy(ctx->stream, b->d_nbrs, b->R, b->Rf, b->n, (int32_t *)os.dev));
        JV_TRY(stage_out_end(ctx, os));
    }
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    b->backlink_s += now_s() - t0;
    return JV_OK;
}

int jv_hip_builder_stats(const jv_builder *b, double *seconds3, int64_t *counts5)
{
    clear_error();
    JV_REQUIRE(b, "builder_stats: NULL argument");
    if (seconds3) {
        seconds3[0] = b->search_s;
        seconds3[1] = b->prune_s;
        seconds3[2] = b->backlink_s;
    }
    if (counts5) {
        counts5[0] = b->batches;
        counts5[1] = b->reprunes;
        counts5[2] = b->inserted;
        counts5[3] = b->visited;
        counts5[4] = b->expanded;
    }
    return JV_OK;
}

// ---- the whole layered build behind one call (verdict r3 #7: the hierarchy used to be assembled by the Python mirror) ----
// GraphIndexBuilder with addHierarchy: a node's top level is floor(-ln(U) / ln(maxDegree)) (getRandomGraphLevel :562-575, HNSW's
// sampling), a node is present on every level up to its own, each level is a Vamana graph over its nodes (here: one jv_builder per
// level — inserts in a seeded random order, prefix-doubling batches — then `improve_passes` passes of improveConnections over every
// node of the level, then enforceDegree), the entry point is a node of the top level (here: the one most similar to the mean of the
// top level's vectors under the index's own similarity function — the reference re-centres its entry on the medoid).  The draws
// come from a seeded splitmix64 instead of the reference's SplittableRandom: the build is a function of (data, parameters, seed).
struct jv_layered {
    int device = 0, max_degree = 0;
    int64_t n = 0;
    int32_t entry = -1;
    int entry_level = 0;
    std::vector<std::vector<int32_t>> nodes;   // level >= 1: ascending node ids (level 0: every ordinal, not stored)
    std::vector<std::vector<int32_t>> nbrs;    // level >= 1: [count][max_degree] global ids, rows packed, -1 padded (host)
    int32_t *d_level0 = nullptr;               // [n][max_degree] (device)
    double seconds[3] = {0, 0, 0}, total_s = 0;
    int64_t counts[5] = {0, 0, 0, 0, 0};
    std::vector<int64_t> level_counts;
};

}  // extern "C"

namespace {
inline uint64_t splitmix64(uint64_t &s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// a seeded permutation of 0..n-1 (Fisher-Yates)
std::vector<int32_t> seeded_permutation(int64_t n, uint64_t seed)
{
    std::vector<int32_t> p((size_t)n);
    for (int64_t i = 0; i < n; ++i) p[(size_t)i] = (int32_t)i;
    uint64_t st = seed;
    for (int64_t i = n - 1; i > 0; --i) {
        const int64_t j = (int64_t)(splitmix64(st) % (uint64_t)(i + 1));
        std::swap(p[(size_t)i], p[(size_t)j]);
    }
    return p;
}

// one level: a builder over (codes, vectors) — n_l nodes with LOCAL ids 0..n_l-1 — inserts in `order`, improve passes, enforceDegree;
// rows to out_rows (host or device memory, n_l x max_degree)
int build_one_level(jv_ctx *ctx, const jv_pq *pq, const jv_codes *codes, const jv_vectors *vectors, jv_vsf vsf, int max_degree, int beam,
                    float alpha, float overflow, int max_batch, int improve_passes, uint64_t seed, int32_t *out_rows, jv_layered *acc)
{
    const int64_t n = codes->count;
    jv_builder *b = nullptr;
    JV_TRY(jv_hip_builder_create(ctx, pq, codes, vectors, vsf, max_degree, beam, alpha, overflow, &b));
    auto run = [&]() -> int {
        const std::vector<int32_t> perm = seeded_permutation(n, seed);
        JV_TRY(jv_hip_builder_seed(ctx, b, perm[0]));
        // experiment knobs: another alpha for the insert phase (DiskANN builds its first pass with alpha = 1), another
        // beam for the improve passes; unset = the call's alpha / beam throughout
        const long long ins_alpha = ctx_opt(ctx, "bl_insert_alpha_x100", 0), imp_beam = ctx_opt(ctx, "bl_improve_beam", 0);
        if (ins_alpha >= 100 && ins_alpha <= 6400) b->alpha = (float)ins_alpha / 100.0f;
        int64_t lo = 1;
        while (lo < n) {   // prefix doubling: a batch never exceeds what the graph already holds
            const int64_t hi = std::min<int64_t>(n, lo + std::min<int64_t>(max_batch, lo));
            JV_TRY(jv_hip_builder_insert_batch(ctx, b, perm.data() + lo, (int)(hi - lo)));
            lo = hi;
        }
        b->alpha = alpha;
        if (imp_beam >= 1 && imp_beam <= 4096) b->beam = (int)imp_beam;
        for (int pass = 0; pass < improve_passes && n >= 2; ++pass)
            for (int64_t s = 0; s < n; s += max_batch)
                JV_TRY(jv_hip_builder_improve_batch(ctx, b, perm.data() + s, (int)std::min<int64_t>(max_batch, n - s)));
        JV_TRY(jv_hip_builder_finish(ctx, b, out_rows));
        double sec[3];
        int64_t cnt[5];
        JV_TRY(jv_hip_builder_stats(b, sec, cnt));
        for (int i = 0; i < 3; ++i) acc->seconds[i] += sec[i];
        for (int i = 0; i < 5; ++i) acc->counts[i] += cnt[i];
        return JV_OK;
    };
    const int rc = run();
    jv_hip_builder_destroy(b);
    return rc;
}
}  // namespace

extern "C" {

int jv_hip_build_layered(jv_ctx *ctx, const jv_pq *pq, const jv_codes *codes, const jv_vectors *vectors, jv_vsf vsf, int max_degree,
                         int beam_width, float alpha, float neighbor_overflow, int max_batch, int improve_passes, uint64_t seed, int min_top,
                         jv_layered **out)
{
    clear_error();
    JV_REQUIRE(ctx && pq && codes && vectors && out, "build_layered: NULL argument");
    *out = nullptr;
    JV_FLOAT_ROWS(vectors, "build_layered");
    JV_REQUIRE(max_batch >= 1 && improve_passes >= 0 && improve_passes <= 8 && min_top >= 1, "build_layered: bad schedule (max_batch %d, improve passes %d, min_top %d)",
               max_batch, improve_passes, min_top);
    JV_REQUIRE(max_degree >= 2 && max_degree <= 64, "build_layered: maxDegree %d outside 2..64", max_degree);
    if (improve_passes > 0) {   // (the improve pass prunes row + beam candidates: checked before the insert phase runs)
        const int Rw = std::max(max_degree, std::min(64, (int)(max_degree * neighbor_overflow)));
        const size_t need = retain_diverse_lds_bytes(Rw + beam_width, codes->M);
        JV_REQUIRE(need <= ctx->lds_per_block, "build_layered: an improveConnections pass over lists of %d + %d candidates x %d code bytes needs %zu bytes of LDS (limit %zu); "
                   "lower the beam width or pass improve_passes = 0", Rw, beam_width, codes->M, need, ctx->lds_per_block);
    }
    JV_REQUIRE(codes->count >= 1 && codes->count <= 0x7fffffffLL && vectors->count >= codes->count, "build_layered: %lld nodes", (long long)codes->count);
    JV_TRY(use_device(ctx->device));
    const double t_start = now_s();
    const int64_t n = codes->count;
    const int D = pq->D;
    jv_layered *L = new jv_layered();
    L->device = ctx->device;
    L->max_degree = max_degree;
    L->n = n;
    auto fail = [&](int rc) {
        jv_hip_layered_destroy(L);
        return rc;
    };
    // ---- levels: getRandomGraphLevel per node (seeded), levels too small to be worth a graph (< min_top nodes) fold into the one below
    std::vector<int8_t> lvl((size_t)n);
    {
        const double ml = max_degree == 1 ? 1.0 : 1.0 / std::log((double)max_degree);
        uint64_t st = seed ^ 0xA5A5A5A55A5A5A5Aull;
        for (int64_t i = 0; i < n; ++i) {
            double u;
            do {
                u = (double)(splitmix64(st) >> 11) * (1.0 / 9007199254740992.0);
            } while (u == 0.0);   // log(0) is undefined
            const int l = (int)(-std::log(u) * ml);
            lvl[(size_t)i] = (int8_t)std::min(l, 31);
        }
    }
    int top = 0;
    {
        int64_t at_least[33] = {0};
        for (int64_t i = 0; i < n; ++i)
            for (int l = 0; l <= lvl[(size_t)i]; ++l) at_least[l]++;
        while (top + 1 < GS_MAX_LEVELS && top + 1 <= 31 && at_least[top + 1] >= min_top) ++top;
    }
    L->nodes.resize((size_t)top + 1);
    L->nbrs.resize((size_t)top + 1);
    L->level_counts.assign((size_t)top + 1, 0);
    L->level_counts[0] = n;
    for (int l = 1; l <= top; ++l) {
        for (int64_t i = 0; i < n; ++i)
            if (lvl[(size_t)i] >= l) L->nodes[(size_t)l].push_back((int32_t)i);
        L->level_counts[(size_t)l] = (int64_t)L->nodes[(size_t)l].size();
    }
    // ---- level 0 over every node, rows straight into device memory (jv_hip_fused_build reads them there) ----
    if (hipMalloc((void **)&L->d_level0, sizeof(int32_t) * (size_t)n * max_degree) != hipSuccess) {
        (void)hipGetLastError();
        set_error("build_layered: cannot allocate the %lld x %d level-0 rows", (long long)n, max_degree);
        return fail(JV_ERR_OOM);
    }
    int rc = build_one_level(ctx, pq, codes, vectors, vsf, max_degree, beam_width, alpha, neighbor_overflow, max_batch, improve_passes, seed,
                             L->d_level0, L);
    if (rc != JV_OK) return fail(rc);
    // ---- upper levels: the level's rows of the vectors gathered into a set of their own, encoded with the same quantizer (the same
    //      codes as their level-0 rows), a builder over LOCAL ids, rows mapped back to global ids ----
    for (int l = 1; l <= top; ++l) {
        const std::vector<int32_t> &ids = L->nodes[(size_t)l];
        const int64_t nl = (int64_t)ids.size();
        jv_vectors *sv = nullptr;
        jv_codes *sc = nullptr;
        Buffer d_ids;
        auto level = [&]() -> int {
            JV_TRY(jv_hip_vectors_create(ctx, nl, D, &sv));
            JV_TRY(d_ids.reserve(sizeof(int32_t) * (size_t)nl));
            JV_HIP_CHECK(hipMemcpyAsync(d_ids.ptr, ids.data(), sizeof(int32_t) * (size_t)nl, hipMemcpyHostToDevice, ctx->stream));
            for (int64_t s0 = 0; s0 < nl; s0 += (1 << 20)) {
                const int pc = (int)std::min<int64_t>(1 << 20, nl - s0);
                JV_TRY(launch_gather_rows(ctx->stream, vectors->d_vecs, vectors->count, D, (const int32_t *)d_ids.ptr + s0, pc,
                                          sv->d_vecs + (size_t)s0 * D, nullptr, 0));
            }
            JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            JV_TRY(jv_hip_vectors_invalidate(sv));
            JV_TRY(jv_hip_codes_create(ctx, pq, nl, &sc));
            JV_TRY(jv_hip_pq_encode_into(ctx, pq, sv, 0, nl, sc));
            std::vector<int32_t> &rows = L->nbrs[(size_t)l];
            rows.assign((size_t)nl * max_degree, -1);
            JV_TRY(build_one_level(ctx, pq, sc, sv, vsf, max_degree, beam_width, alpha, neighbor_overflow, max_batch, improve_passes,
                                   seed + (uint64_t)l, rows.data(), L));
            for (int32_t &x : rows)
                if (x >= 0) x = ids[(size_t)x];
            return JV_OK;
        };
        rc = level();
        if (sc) jv_hip_codes_destroy(sc);
        if (sv) jv_hip_vectors_destroy(sv);
        d_ids.release();
        if (rc != JV_OK) return fail(rc);
    }
    // ---- entry point: the top level's node most similar to the mean of (up to 4096 of) the top level's vectors ----
    L->entry_level = top;
    if (top == 0) {
        L->entry = seeded_permutation(n, seed)[0];   // a flat graph is entered where its construction started
    } else {
        const std::vector<int32_t> &ids = L->nodes[(size_t)top];
        const int cnt = (int)std::min<size_t>(ids.size(), 4096);
        std::vector<float> rowsf((size_t)cnt * D), mean((size_t)D);
        Buffer d_ids, d_rows, d_sc;
        auto pick = [&]() -> int {
            JV_TRY(d_ids.reserve(sizeof(int32_t) * ids.size()));
            JV_TRY(d_rows.reserve(sizeof(float) * (size_t)cnt * D));
            JV_TRY(d_sc.reserve(sizeof(float) * ids.size()));
            JV_HIP_CHECK(hipMemcpyAsync(d_ids.ptr, ids.data(), sizeof(int32_t) * ids.size(), hipMemcpyHostToDevice, ctx->stream));
            JV_TRY(launch_gather_rows(ctx->stream, vectors->d_vecs, vectors->count, D, (const int32_t *)d_ids.ptr, cnt, (float *)d_rows.ptr, nullptr, 0));
            JV_HIP_CHECK(hipMemcpyAsync(rowsf.data(), d_rows.ptr, sizeof(float) * (size_t)cnt * D, hipMemcpyDeviceToHost, ctx->stream));
            JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            for (int d = 0; d < D; ++d) {
                double acc = 0.0;
                for (int r = 0; r < cnt; ++r) acc += (double)rowsf[(size_t)r * D + d];
                mean[(size_t)d] = (float)(acc / cnt);
            }
            std::vector<float> sc(ids.size());
            JV_TRY(jv_hip_exact_scores(ctx, vectors, mean.data(), 1, vsf, (const int32_t *)d_ids.ptr, (int)ids.size(), (float *)d_sc.ptr));
            JV_HIP_CHECK(hipMemcpyAsync(sc.data(), d_sc.ptr, sizeof(float) * ids.size(), hipMemcpyDeviceToHost, ctx->stream));
            JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            size_t best = 0;
            for (size_t i = 1; i < ids.size(); ++i)
                if (sc[i] > sc[best]) best = i;   // (a NaN never wins; ties keep the smaller id)
            L->entry = ids[best];
            return JV_OK;
        };
        rc = pick();
        d_ids.release();
        d_rows.release();
        d_sc.release();
        if (rc != JV_OK) return fail(rc);
    }
    L->total_s = now_s() - t_start;
    *out = L;
    return JV_OK;
}

int jv_hip_layered_info(const jv_layered *l, int *n_levels, int32_t *entry_node, int *entry_level, int64_t *level_counts)
{
    clear_error();
    JV_REQUIRE(l, "layered_info: NULL argument");
    if (n_levels) *n_levels = (int)l->level_counts.size();
    if (entry_node) *entry_node = l->entry;
    if (entry_level) *entry_level = l->entry_level;
    if (level_counts)
        for (size_t i = 0; i < l->level_counts.size(); ++i) level_counts[i] = l->level_counts[i];
    return JV_OK;
}

int jv_hip_layered_level(jv_ctx *ctx, const jv_layered *l, int level, int32_t *nodes_out, int32_t *neighbors_out)
{
    clear_error();
    JV_REQUIRE(ctx && l, "layered_level: NULL argument");
    JV_REQUIRE(level >= 0 && level < (int)l->level_counts.size(), "layered_level: level %d of %d", level, (int)l->level_counts.size());
    if (level == 0) {
        JV_REQUIRE(!nodes_out, "layered_level: level 0 holds every ordinal (nodes_out must be NULL)");
        if (neighbors_out) {
            JV_REQUIRE(ctx->device == l->device, "layered_level: the graph lives on device %d", l->device);
            JV_TRY(use_device(ctx->device));
            JV_HIP_CHECK(hipMemcpyAsync(neighbors_out, l->d_level0, sizeof(int32_t) * (size_t)l->n * l->max_degree, hipMemcpyDefault, ctx->stream));
            JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        }
        return JV_OK;
    }
    const std::vector<int32_t> &ids = l->nodes[(size_t)level], &rows = l->nbrs[(size_t)level];
    if (nodes_out) memcpy(nodes_out, ids.data(), sizeof(int32_t) * ids.size());
    if (neighbors_out) memcpy(neighbors_out, rows.data(), sizeof(int32_t) * rows.size());
    return JV_OK;
}

const int32_t *jv_hip_layered_level0_device(const jv_layered *l) { return l ? l->d_level0 : nullptr; }

int jv_hip_layered_stats(const jv_layered *l, double *seconds4, int64_t *counts5)
{
    clear_error();
    JV_REQUIRE(l, "layered_stats: NULL argument");
    if (seconds4) {
        for (int i = 0; i < 3; ++i) seconds4[i] = l->seconds[i];
        seconds4[3] = l->total_s;
    }
    if (counts5)
        for (int i = 0; i < 5; ++i) counts5[i] = l->counts[i];
    return JV_OK;
}

int jv_hip_layered_destroy(jv_layered *l)
{
    if (!l) return JV_OK;
    if (l->d_level0) (void)hipFree(l->d_level0);
    delete l;
    return JV_OK;
}

int jv_hip_builder_working_lists(jv_ctx *ctx, const jv_builder *b, int32_t *ids_out, float *scores_out, int32_t *diverse_before_out)
{
    clear_error();
    JV_REQUIRE(ctx && b, "builder_working_lists: NULL argument");
    JV_REQUIRE(ctx->device == b->device, "builder_working_lists: the builder lives on device %d", b->device);
    JV_REQUIRE(b->stored || (!scores_out && !diverse_before_out), "builder_working_lists: scores and marks exist with sorted lists / in reference order only (bl_sorted_lists / bl_ref_order)");
    JV_TRY(use_device(ctx->device));
    const size_t cells = (size_t)b->n * b->R;
    if (ids_out) JV_HIP_CHECK(hipMemcpyAsync(ids_out, b->d_nbrs, sizeof(int32_t) * cells, hipMemcpyDefault, ctx->stream));
    if (scores_out) JV_HIP_CHECK(hipMemcpyAsync(scores_out, b->d_nsc, sizeof(float) * cells, hipMemcpyDefault, ctx->stream));
    if (diverse_before_out) JV_HIP_CHECK(hipMemcpyAsync(diverse_before_out, b->d_db, sizeof(int32_t) * (size_t)b->n, hipMemcpyDefault, ctx->stream));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return JV_OK;
}

const int32_t *jv_hip_builder_neighbors_device(const jv_builder *b, int *row_width)
{
    if (!b) return nullptr;
    if (row_width) *row_width = b->R;
    return b->d_nbrs;
}

}  // extern "C"
