// graph_search.cpp — host batched searcher: a multi-query, lock-step restatement of GraphSearcher
// (B/graph/GraphSearcher.java:222-507) that keeps the irregular Vamana/HNSW traversal on
// the HOST and ships each round's frontier — one expanded node per live query — to the GPU for scoring.
//
// Per query the control flow is exactly the reference's (initializeInternal -> upper layers with rerankK = 1 ->
// setEntryPointsFromPreviousLayer -> searchLayer0 -> reranking); queries never interact, so advancing Q of them
// in lock-step changes nothing observable per query.  One ROUND =
//     host (parallel over queries): stopSearch test, pop the best candidate, addTopCandidate, pick the origin
//     GPU  (one launch for the batch): layer 0 + FusedPQ: jv_fused scores of the origin's packed block
//                                      (FusedPQDecoder.similarityToNeighbor); otherwise an ADC gather of the
//                                      unvisited neighbours' codes (PQDecoder.similarityTo / cached codes)
//     host (parallel): visited.mark + candidates.push in neighbour order (View.processNeighbors)
// Final exact rerank and top-K run on the GPU for the whole batch.
//
// Deviation (documented in DESIGN.md): after the rerank the top-K is taken under the NodeQueue order on the exact
// scores; for EXACT-score ties at the K-th place the reference's choice depends on heap array order
// (NodeQueue.java:197-214).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>

#include "jv_internal.h"

namespace jv {

// ---------------------------------------------------------------------------------------------
// worker pool: persistent threads, static partition of [0, n)
// ---------------------------------------------------------------------------------------------
class HostPool {
public:
    explicit HostPool(int nthreads) : n_(nthreads)
    {
        for (int t = 1; t < n_; ++t) threads_.emplace_back([this, t] { worker(t); });
    }
    ~HostPool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (auto &th : threads_) th.join();
    }
    int size() const { return n_; }
    void parallel_for(int n, const std::function<void(int, int)> &fn)
    {
        if (n_ == 1 || n < 2 * n_) {
            fn(0, n);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn;
            total_ = n;
            pending_ = n_ - 1;
            ++gen_;
        }
        cv_.notify_all();
        run_chunk(0);
        std::unique_lock<std::mutex> lk(m_);
        done_cv_.wait(lk, [this] { return pending_ == 0; });
    }

private:
    void run_chunk(int t)
    {
        const int per = (total_ + n_ - 1) / n_;
        const int lo = t * per, hi = std::min(total_, lo + per);
        if (lo < hi) (*fn_)(lo, hi);
    }
    void worker(int t)
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
            }
            run_chunk(t);
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--pending_ == 0) done_cv_.notify_one();
            }
        }
    }
    int n_;
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(int, int)> *fn_ = nullptr;
    int total_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

static void destroy_pool(void *p) { delete static_cast<HostPool *>(p); }

static HostPool *get_pool(jv_ctx *ctx)
{
    if (!ctx->host_pool) {
        int n = (int)std::thread::hardware_concurrency();
        if (const char *e = getenv("JVECTOR_HIP_HOST_THREADS")) n = atoi(e);
        n = std::max(1, std::min(n, 128));
        ctx->host_pool = new HostPool(n);
        ctx->host_pool_destroy = destroy_pool;
    }
    return static_cast<HostPool *>(ctx->host_pool);
}

// ---------------------------------------------------------------------------------------------
// NodeQueue keys on the host (NodeQueue.java:125-129, NumericUtils.java:49-65)
// ---------------------------------------------------------------------------------------------
static inline int64_t nq_encode(int32_t node, float score)
{
    int32_t bits;
    if (score != score) bits = 0x7fc00000;
    else memcpy(&bits, &score, 4);
    const int32_t s = bits ^ ((bits >> 31) & 0x7fffffff);
    return (int64_t)(((uint64_t)(uint32_t)s) << 32) | (int64_t)(0xFFFFFFFFull & (uint64_t)(uint32_t)(~node));
}
static inline int32_t nq_node(int64_t k) { return (int32_t)~(uint32_t)(k & 0xFFFFFFFFll); }
static inline float nq_score(int64_t k)
{
    const int32_t e = (int32_t)(k >> 32);
    const int32_t bits = e ^ ((e >> 31) & 0x7fffffff);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

// open-addressing int set (the reference uses agrona's IntHashSet, GraphSearcher.java:83-87)
struct IntSet {
    std::vector<int32_t> slots;
    int count = 0;
    void reset(int cap_pow2)
    {
        slots.assign((size_t)cap_pow2, -1);
        count = 0;
    }
    static inline uint32_t hash(int32_t v) { return (uint32_t)v * 0x9E3779B1u; }
    bool add(int32_t v)  // true when newly added
    {
        if ((size_t)(count + 1) * 2 > slots.size()) grow();
        const uint32_t mask = (uint32_t)slots.size() - 1;
        uint32_t i = hash(v) & mask;
        while (slots[i] != -1) {
            if (slots[i] == v) return false;
            i = (i + 1) & mask;
        }
        slots[i] = v;
        ++count;
        return true;
    }
    void grow()
    {
        std::vector<int32_t> old;
        old.swap(slots);
        slots.assign(old.size() * 2, -1);
        count = 0;
        for (int32_t v : old)
            if (v != -1) add(v);
    }
};

struct QState {
    std::vector<int64_t> cand;     // max-heap
    std::vector<int64_t> res;      // min-heap (top = worst kept)
    std::vector<int64_t> evicted;
    IntSet visited;
    bool active = true;
    int32_t origin = -1;
    int n_pending = 0;
    int64_t n_visited = 0, n_expanded = 0;
};

}  // namespace jv

using namespace jv;

struct jv_graph {
    int64_t n_nodes = 0;
    int n_levels = 0;
    int32_t entry_node = -1;
    int entry_level = 0;
    struct Level {
        int count = 0, degree = 0;
        std::vector<int32_t> nodes;   // sorted ascending; empty = every node (level 0)
        std::vector<int32_t> nbrs;    // count x degree, packed, -1 padded
    };
    std::vector<Level> levels;
    const int32_t *row(int level, int32_t node) const
    {
        const Level &L = levels[level];
        if (L.nodes.empty()) return (node >= 0 && node < L.count) ? L.nbrs.data() + (size_t)node * L.degree : nullptr;
        auto it = std::lower_bound(L.nodes.begin(), L.nodes.end(), node);
        if (it == L.nodes.end() || *it != node) return nullptr;
        return L.nbrs.data() + (size_t)(it - L.nodes.begin()) * L.degree;
    }
};

extern "C" {

int jv_hip_graph_create(jv_ctx *ctx, int64_t n_nodes, int n_levels, jv_graph **out)
{
    clear_error();
    JV_REQUIRE(ctx && out, "graph_create: NULL argument");
    JV_REQUIRE(n_nodes > 0 && n_nodes <= 0x7fffffffLL && n_levels >= 1 && n_levels <= 64, "graph_create: bad sizes");
    jv_graph *g = new jv_graph();
    g->n_nodes = n_nodes;
    g->n_levels = n_levels;
    g->levels.resize(n_levels);
    *out = g;
    return JV_OK;
}

int jv_hip_graph_set_level(jv_ctx *ctx, jv_graph *g, int level, int count, const int32_t *node_ids,
                           const int32_t *neighbors, int degree)
{
    clear_error();
    JV_REQUIRE(ctx && g && neighbors, "graph_set_level: NULL argument");
    JV_REQUIRE(level >= 0 && level < g->n_levels, "graph_set_level: level %d out of range", level);
    JV_REQUIRE(count > 0 && degree > 0 && degree < 2048, "graph_set_level: bad count/degree");
    JV_REQUIRE(level > 0 || (node_ids == nullptr && count == g->n_nodes), "graph_set_level: level 0 holds every node");
    JV_REQUIRE(!is_device_ptr(neighbors) && !is_device_ptr(node_ids), "graph_set_level: adjacency must be host memory");
    jv_graph::Level &L = g->levels[level];
    L.count = count;
    L.degree = degree;
    L.nbrs.assign(neighbors, neighbors + (size_t)count * degree);
    L.nodes.clear();
    if (node_ids) {
        L.nodes.assign(node_ids, node_ids + count);
        JV_REQUIRE(std::is_sorted(L.nodes.begin(), L.nodes.end()), "graph_set_level: node ids must be ascending");
    }
    return JV_OK;
}

int jv_hip_graph_set_entry(jv_graph *g, int32_t node, int level)
{
    clear_error();
    JV_REQUIRE(g, "graph_set_entry: NULL graph");
    JV_REQUIRE(node >= 0 && node < g->n_nodes && level >= 0 && level < g->n_levels, "graph_set_entry: out of range");
    g->entry_node = node;
    g->entry_level = level;
    return JV_OK;
}

int jv_hip_graph_destroy(jv_graph *g)
{
    delete g;
    return JV_OK;
}

int jv_hip_graph_search(jv_ctx *ctx, const jv_graph *g, jv_luts *l, const jv_codes *codes, const jv_fused *fused,
                        const jv_vectors *vectors, const float *queries, int Q, jv_vsf vsf, int topK, int rerankK,
                        int32_t *out_ids, float *out_scores, int64_t *stats)
{
    clear_error();
    JV_REQUIRE(ctx && g && l && codes, "graph_search: NULL argument");
    JV_REQUIRE(topK > 0, "graph_search: topK must be positive");
    JV_REQUIRE(rerankK >= topK, "rerankK %d must be >= topK %d", rerankK, topK);  // GraphSearcher.java:233
    JV_REQUIRE(g->entry_node >= 0, "graph_search: the graph has no entry node");
    JV_REQUIRE(codes->pq == l->pq && codes->count >= g->n_nodes, "graph_search: code store does not match the graph");
    JV_REQUIRE(!fused || (fused->pq == l->pq && fused->count == g->n_nodes && fused->maxDegree == g->levels[0].degree),
               "graph_search: fused blocks do not match the graph");
    JV_REQUIRE(!vectors || (vectors->D == l->pq->D && vectors->count >= g->n_nodes), "graph_search: vectors mismatch");
    JV_REQUIRE(Q <= l->capacity, "graph_search: Q=%d exceeds the LUT capacity %d", Q, l->capacity);
    for (int lv = 0; lv <= g->entry_level; ++lv)
        JV_REQUIRE(g->levels[lv].count > 0, "graph_search: level %d was never set", lv);
    if (Q == 0) return JV_OK;
    JV_REQUIRE(queries && out_ids && out_scores, "graph_search: NULL buffer");
    JV_TRY(use_device(ctx->device));

    const jv_decoder_kind kind = fused ? JV_DECODER_FUSED : JV_DECODER_PQ;
    JV_TRY(jv_hip_luts_build(ctx, l, queries, Q, vsf, kind));
    const int kvsf = to_kernel_vsf(vsf);
    if (vsf == JV_COSINE) {
        JV_TRY(ensure_code_norms(ctx, const_cast<jv_codes *>(codes)));
        if (fused) JV_TRY(ensure_fused_norms(ctx, const_cast<jv_fused *>(fused)));
    }
    HostPool *pool = get_pool(ctx);
    int max_deg = 0;
    for (int lv = 0; lv <= g->entry_level; ++lv) max_deg = std::max(max_deg, g->levels[lv].degree);

    // pinned host + device buffers for the per-round exchange
    const size_t cells = (size_t)Q * max_deg;
    // luts_build may still be DMA-ing the queries out of the pinned staging buffer: drain before reusing it
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    JV_TRY(ctx->h_in.reserve(sizeof(int32_t) * cells));
    JV_TRY(ctx->h_out.reserve(sizeof(float) * cells));
    JV_TRY(ctx->d_in.reserve(sizeof(int32_t) * cells));
    JV_TRY(ctx->d_out.reserve(sizeof(float) * cells));
    int32_t *h_ord = (int32_t *)ctx->h_in.ptr;   // fused: Q origins; gather: Q x deg ordinals
    float *h_sc = (float *)ctx->h_out.ptr;
    int32_t *d_ord = (int32_t *)ctx->d_in.ptr;
    float *d_sc = (float *)ctx->d_out.ptr;

    std::vector<QState> st((size_t)Q);
    // initializeInternal: score the entry node for every query (one gather of Q x 1)
    for (int q = 0; q < Q; ++q) h_ord[q] = g->entry_node;
    JV_HIP_CHECK(hipMemcpyAsync(d_ord, h_ord, sizeof(int32_t) * (size_t)Q, hipMemcpyHostToDevice, ctx->stream));
    {
        ProfScope ps(ctx, R_ADC);
        JV_TRY(launch_adc(ctx->stream, ctx, l->d_luts, l->d_bmag, Q, codes->M, kvsf, codes->d_codes, codes->d_norms,
                          codes->count, 0, 1, d_ord, d_sc));
    }
    JV_HIP_CHECK(hipMemcpyAsync(h_sc, d_sc, sizeof(float) * (size_t)Q, hipMemcpyDeviceToHost, ctx->stream));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    pool->parallel_for(Q, [&](int lo, int hi) {
        for (int q = lo; q < hi; ++q) {
            QState &s = st[q];
            s.visited.reset(1024);
            s.visited.add(g->entry_node);
            s.cand.push_back(nq_encode(g->entry_node, h_sc[q]));
        }
    });

    using clk = std::chrono::steady_clock;
    const bool timing = getenv("JVECTOR_HIP_GRAPH_TIMING") != nullptr;
    double t_pre = 0, t_gpu = 0, t_post = 0;
    long n_rounds = 0;
    auto since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };

    for (int lvl = g->entry_level; lvl >= 0; --lvl) {
        const int rk = lvl > 0 ? 1 : rerankK;
        const int deg = g->levels[lvl].degree;
        const bool use_fused = (lvl == 0 && fused != nullptr);
        for (auto &s : st) s.active = true;
        for (;;) {
            std::atomic<int> n_active{0};
            auto tp0 = clk::now();
            // ---- host: stop test, pop, addTopCandidate, choose origin (searchOneLayer :421-433) ----
            pool->parallel_for(Q, [&](int lo, int hi) {
                int local_active = 0;
                for (int q = lo; q < hi; ++q) {
                    QState &s = st[q];
                    s.origin = -1;
                    s.n_pending = 0;
                    int32_t *ords = h_ord + (use_fused ? (size_t)q : (size_t)q * deg);
                    if (use_fused) ords[0] = -1;
                    else std::fill(ords, ords + deg, -1);
                    if (!s.active) continue;
                    if (s.cand.empty()) { s.active = false; continue; }
                    const int64_t top = s.cand.front();
                    const float top_score = nq_score(top);
                    if ((int)s.res.size() >= rk && top_score < nq_score(s.res.front())) { s.active = false; continue; }  // stopSearch
                    std::pop_heap(s.cand.begin(), s.cand.end());
                    s.cand.pop_back();
                    const int32_t node = nq_node(top);
                    // addTopCandidate :515-530
                    if ((int)s.res.size() < rk) {
                        s.res.push_back(top);
                        std::push_heap(s.res.begin(), s.res.end(), std::greater<int64_t>());
                    } else if (top_score > nq_score(s.res.front())) {
                        s.evicted.push_back(s.res.front());
                        std::pop_heap(s.res.begin(), s.res.end(), std::greater<int64_t>());
                        s.res.back() = top;
                        std::push_heap(s.res.begin(), s.res.end(), std::greater<int64_t>());
                    }
                    s.n_expanded++;
                    s.origin = node;
                    ++local_active;
                    if (use_fused) {
                        ords[0] = node;
                    } else {
                        const int32_t *row = g->row(lvl, node);
                        if (row) {
                            for (int i = 0; i < deg; ++i) {
                                const int32_t nb = row[i];
                                if (nb < 0) break;
                                if (s.visited.add(nb)) ords[s.n_pending++] = nb;  // visited.mark, in neighbour order
                            }
                        }
                    }
                }
                n_active += local_active;
            });
            if (n_active.load() == 0) break;
            t_pre += since(tp0);
            ++n_rounds;
            auto tg0 = clk::now();
            // ---- GPU: score this round's frontier ----
            const size_t n_ord = use_fused ? (size_t)Q : (size_t)Q * deg;
            JV_HIP_CHECK(hipMemcpyAsync(d_ord, h_ord, sizeof(int32_t) * n_ord, hipMemcpyHostToDevice, ctx->stream));
            {
                ProfScope ps(ctx, R_ADC);
                if (use_fused)
                    JV_TRY(launch_fused(ctx->stream, ctx, l->d_luts, l->d_bmag, Q, fused->M, kvsf, fused->d_blocks,
                                        fused->d_neighbors, fused->d_norms, fused->maxDegree, fused->count, d_ord, d_sc,
                                        nullptr));
                else
                    JV_TRY(launch_adc(ctx->stream, ctx, l->d_luts, l->d_bmag, Q, codes->M, kvsf, codes->d_codes,
                                      codes->d_norms, codes->count, 0, deg, d_ord, d_sc));
            }
            JV_HIP_CHECK(hipMemcpyAsync(h_sc, d_sc, sizeof(float) * (size_t)Q * deg, hipMemcpyDeviceToHost, ctx->stream));
            JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            t_gpu += since(tg0);
            auto tq0 = clk::now();
            // ---- host: push the scored neighbours (View.processNeighbors) ----
            pool->parallel_for(Q, [&](int lo, int hi) {
                for (int q = lo; q < hi; ++q) {
                    QState &s = st[q];
                    if (s.origin < 0) continue;
                    const float *sc = h_sc + (size_t)q * deg;
                    if (use_fused) {
                        const int32_t *row = g->row(0, s.origin);
                        for (int i = 0; i < deg; ++i) {
                            const int32_t nb = row[i];
                            if (nb < 0) break;
                            if (s.visited.add(nb)) {
                                s.cand.push_back(nq_encode(nb, sc[i]));
                                std::push_heap(s.cand.begin(), s.cand.end());
                                s.n_visited++;
                            }
                        }
                    } else {
                        const int32_t *ords = h_ord + (size_t)q * deg;
                        for (int j = 0; j < s.n_pending; ++j) {
                            s.cand.push_back(nq_encode(ords[j], sc[j]));
                            std::push_heap(s.cand.begin(), s.cand.end());
                            s.n_visited++;
                        }
                    }
                }
            });
            t_post += since(tq0);
        }
        if (lvl > 0) {  // setEntryPointsFromPreviousLayer :324-331
            pool->parallel_for(Q, [&](int lo, int hi) {
                for (int q = lo; q < hi; ++q) {
                    QState &s = st[q];
                    for (int64_t k : s.res) { s.cand.push_back(k); std::push_heap(s.cand.begin(), s.cand.end()); }
                    for (int64_t k : s.evicted) { s.cand.push_back(k); std::push_heap(s.cand.begin(), s.cand.end()); }
                    s.res.clear();
                    s.evicted.clear();
                }
            });
        }
    }

    if (timing)
        fprintf(stderr, "[jv graph_search] Q=%d rounds=%ld threads=%d host-pre %.2f ms, gpu+copies %.2f ms, host-post %.2f ms\n", Q,
                n_rounds, pool->size(), t_pre, t_gpu, t_post);
    // ---- reranking :471-507 ----
    OutStage oi, osc;
    JV_TRY(stage_out_begin(ctx, out_ids, sizeof(int32_t) * (size_t)Q * topK, ctx->d_scratch2, &oi));
    JV_TRY(stage_out_begin(ctx, out_scores, sizeof(float) * (size_t)Q * topK, ctx->d_scratch3, &osc));
    const size_t c1 = (size_t)Q * rerankK;
    JV_TRY(ctx->h_in.reserve(sizeof(int32_t) * c1 + sizeof(float) * c1));
    int32_t *h_cand = (int32_t *)ctx->h_in.ptr;
    float *h_cand_sc = (float *)(h_cand + c1);
    pool->parallel_for(Q, [&](int lo, int hi) {
        for (int q = lo; q < hi; ++q) {
            const QState &s = st[q];
            for (int i = 0; i < rerankK; ++i) {
                const bool have = i < (int)s.res.size();
                h_cand[(size_t)q * rerankK + i] = have ? nq_node(s.res[i]) : -1;
                h_cand_sc[(size_t)q * rerankK + i] = have ? nq_score(s.res[i]) : -INFINITY;
            }
        }
    });
    // device layout: [cand ids][approx or exact scores][qnorm]
    JV_TRY(ctx->d_in.reserve(sizeof(int32_t) * c1 + sizeof(float) * c1 + sizeof(float) * (size_t)Q + 256));
    int32_t *d_cand = (int32_t *)ctx->d_in.ptr;
    float *d_cand_sc = (float *)(d_cand + c1);
    float *d_qnorm = d_cand_sc + c1;
    JV_HIP_CHECK(hipMemcpyAsync(d_cand, h_cand, sizeof(int32_t) * c1 + sizeof(float) * c1, hipMemcpyHostToDevice,
                                ctx->stream));
    JV_TRY(ctx->d_scratch.reserve(topk_scratch_bytes(Q, topK)));
    if (vectors) {
        ProfScope ps(ctx, R_EXACT);
        JV_TRY(launch_exact_gather(ctx->stream, vectors->d_vecs, vectors->count, vectors->D, l->d_raw_queries, Q, kvsf,
                                   d_cand, rerankK, d_cand_sc, d_qnorm));
    }
    {
        ProfScope ps(ctx, R_TOPK);
        JV_TRY(launch_topk(ctx->stream, ctx, d_cand_sc, d_cand, Q, rerankK, rerankK, 0, topK, (int32_t *)oi.dev,
                           (float *)osc.dev, ctx->d_scratch.ptr));
    }
    JV_TRY(stage_out_end(ctx, oi));
    JV_TRY(stage_out_end(ctx, osc));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (stats) {
        for (int q = 0; q < Q; ++q) {
            stats[2 * q] = st[q].n_visited;
            stats[2 * q + 1] = st[q].n_expanded;
        }
    }
    return JV_OK;
}

}  // extern "C"
