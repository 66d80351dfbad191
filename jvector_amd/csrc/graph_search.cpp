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
#include <memory>
#include <unordered_map>
#include <mutex>
#include <thread>

#include "gs_host.h"
#include "gs_params.h"
#include "jv_internal.h"

namespace jv {
// gs_ubr unset: the register-table bound form (gs_body.h "UBR") serves every launch it applies to
constexpr long long kGsUbrDefault = 1;
constexpr long long kGsUbrcDefault = 1;   // the same form over the builder's compacted 33 ... 64-wide rows (gs_ubrc): the 10M x 768 build's searches 21.0 -> 13.9 s,
                                          // the identical graph (profiles/r5_u)


// ---------------------------------------------------------------------------------------------
// worker pool: persistent threads, static partition of [0, n)
// ---------------------------------------------------------------------------------------------
class HostPool {
public:
    // Persistent workers.  Inside a search (between begin() and end()) they SPIN on an atomic generation counter:
    // a traversal round is two parallel phases of a few tens of microseconds each, far below the cost of a
    // mutex/condvar wake-up of ~100 threads (measured: 1.5-2 ms per phase with 128 condvar workers).  Outside a
    // search they sleep on a condition variable.
    explicit HostPool(int nthreads) : n_(nthreads)
    {
        for (int t = 1; t < n_; ++t) threads_.emplace_back([this, t] { worker(t); });
    }
    ~HostPool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_.store(true);
            awake_ = true;
        }
        cv_.notify_all();
        gen_.fetch_add(1, std::memory_order_release);
        for (auto &th : threads_) th.join();
    }
    int size() const { return n_; }
    void begin()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            awake_ = true;
        }
        cv_.notify_all();
    }
    void end()
    {
        std::lock_guard<std::mutex> lk(m_);
        awake_ = false;
    }
    void parallel_for(int n, const std::function<void(int, int)> &fn)
    {
        if (n_ == 1 || n < 2 * n_) {
            fn(0, n);
            return;
        }
        fn_ = &fn;
        total_ = n;
        done_.store(0, std::memory_order_relaxed);
        gen_.fetch_add(1, std::memory_order_release);
        run_chunk(0);
        while (done_.load(std::memory_order_acquire) != n_ - 1) cpu_relax();
    }

private:
    static inline void cpu_relax() { __builtin_ia32_pause(); }
    void run_chunk(int t)
    {
        const int per = (total_ + n_ - 1) / n_;
        const int lo = t * per, hi = std::min(total_, lo + per);
        if (lo < hi) (*fn_)(lo, hi);
    }
    void worker(int t)
    {
        // The generation this worker has served.  It must start at the value gen_ had when the pool was BUILT (0), not at
        // whatever the counter reads when the thread first gets to run: a thread that is scheduled late would otherwise
        // adopt the generation of a parallel_for that is already waiting for it and never execute its chunk, leaving the
        // caller spinning on done_ forever (seen with a mock device, where nothing delays the first parallel_for).
        uint64_t seen = 0;
        for (;;) {
            {   // sleep while no search is running
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return awake_ || stop_.load(); });
            }
            if (stop_.load()) return;
            // spin for work while awake
            long idle = 0;
            for (;;) {
                const uint64_t g = gen_.load(std::memory_order_acquire);
                if (g != seen) {
                    seen = g;
                    if (stop_.load()) return;
                    run_chunk(t);
                    done_.fetch_add(1, std::memory_order_release);
                    idle = 0;
                    continue;
                }
                cpu_relax();
                if (++idle > 4096) {
                    idle = 0;
                    bool aw;
                    {
                        std::lock_guard<std::mutex> lk(m_);
                        aw = awake_;
                    }
                    if (!aw || stop_.load()) break;  // back to the condvar
                }
            }
        }
    }
    int n_;
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_;
    bool awake_ = false;
    std::atomic<bool> stop_{false};
    std::atomic<uint64_t> gen_{0};
    std::atomic<int> done_{0};
    const std::function<void(int, int)> *fn_ = nullptr;
    int total_ = 0;
};

static void destroy_pool(void *p) { delete static_cast<HostPool *>(p); }

static HostPool *get_pool(jv_ctx *ctx)
{
    if (!ctx->host_pool) {
        // default: half the hardware threads (one per physical core, like the reference's PhysicalCoreExecutor,
        // B/util/PhysicalCoreExecutor.java:121), capped at 64
        int n = (int)std::thread::hardware_concurrency() / 2;
        // containers: honour the cgroup CPU quota (spinning workers beyond it only steal each other's time slices;
        // measured on the 16-CPU-quota GPU box: 16 threads 228 ms, 32 threads 402 ms, 64 threads 1036 ms per batch)
        {
            double quota = 0;
            if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
                char a[64];
                double period = 0;
                if (fscanf(f, "%63s %lf", a, &period) == 2 && strcmp(a, "max") != 0 && period > 0) quota = atof(a) / period;
                fclose(f);
            } else if (FILE *f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
                double q = 0, period = 100000;
                if (fscanf(f1, "%lf", &q) != 1) q = 0;
                fclose(f1);
                if (FILE *f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                    if (fscanf(f2, "%lf", &period) != 1) period = 100000;
                    fclose(f2);
                }
                if (q > 0 && period > 0) quota = q / period;
            }
            if (quota >= 1.0) n = std::min(n, (int)quota);
        }
        // one process per GPU: share the host cores between the ranks of this node (torchrun sets LOCAL_WORLD_SIZE)
        if (const char *lw = getenv("LOCAL_WORLD_SIZE")) {
            const int ranks = atoi(lw);
            if (ranks > 1) n = std::max(1, n / ranks);
        }
        if (const char *e = getenv("JVECTOR_HIP_HOST_THREADS")) n = atoi(e);
        n = std::max(1, std::min(n, 64));
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
    void prefetch(int32_t v) const { __builtin_prefetch(&slots[hash(v) & ((uint32_t)slots.size() - 1)]); }
    bool contains(int32_t v) const
    {
        const uint32_t mask = (uint32_t)slots.size() - 1;
        uint32_t i = hash(v) & mask;
        while (slots[i] != -1) {
            if (slots[i] == v) return true;
            i = (i + 1) & mask;
        }
        return false;
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

// 8-ary max-heap of NodeQueue keys for the candidate queue (the reference's GrowableLongHeap is binary; only the
// pop ORDER is observable and keys are unique, so any correct heap yields the same sequence).  Between two rounds a
// slot's heap falls out of the core's caches: a binary sift walks ~log2(n) dependent cache misses, an 8-ary heap
// whose 8 children share one 64-byte line walks ~log8(n).
struct MaxHeap8 {
    int64_t *a = nullptr;   // 64-byte aligned; element i lives at a[i + 7] so that children 8i+1..8i+8 share a line
    int n = 0, cap = 0;
    ~MaxHeap8() { free(a); }
    MaxHeap8() = default;
    MaxHeap8(const MaxHeap8 &) = delete;
    MaxHeap8 &operator=(const MaxHeap8 &) = delete;
    MaxHeap8(MaxHeap8 &&o) noexcept : a(o.a), n(o.n), cap(o.cap) { o.a = nullptr; o.n = o.cap = 0; }
    void clear() { n = 0; }
    bool empty() const { return n == 0; }
    int64_t top() const { return a[7]; }
    const void *root_line() const { return a ? a + 7 : nullptr; }
    void reserve(int c)
    {
        if (c <= cap) return;
        int nc = cap ? cap : 1024;
        while (nc < c) nc *= 2;
        int64_t *na = (int64_t *)aligned_alloc(64, sizeof(int64_t) * ((size_t)nc + 8));
        if (a) memcpy(na, a, sizeof(int64_t) * ((size_t)n + 7));
        free(a);
        a = na;
        cap = nc;
    }
    void push(int64_t v)
    {
        reserve(n + 1);
        int i = n++;
        while (i > 0) {
            const int p = (i - 1) >> 3;
            if (a[p + 7] >= v) break;
            a[i + 7] = a[p + 7];
            i = p;
        }
        a[i + 7] = v;
    }
    void pop()
    {
        const int64_t v = a[--n + 7];
        if (n == 0) return;
        int i = 0;
        for (;;) {
            const int c0 = 8 * i + 1;
            if (c0 >= n) break;
            const int c1 = c0 + 8 < n ? c0 + 8 : n;
            int best = c0;
            int64_t bv = a[c0 + 7];
            for (int c = c0 + 1; c < c1; ++c)
                if (a[c + 7] > bv) { bv = a[c + 7]; best = c; }
            if (bv <= v) break;
            a[i + 7] = bv;
            i = best;
        }
        a[i + 7] = v;
    }
};

// approximateResults / rerankedResults: the reference's binary min-heap (AbstractLongHeap.java:77-85,136-146,158-187; 0-based
// here).  NodeQueue keys are unique, so after the same push / updateTop / pop sequence the ARRAY holds the same order as
// the reference's — and NodeQueue.rerank (:160-230) walks that array, which decides who survives an exact-score tie.
struct JMinHeap {
    std::vector<int64_t> a;
    int size() const { return (int)a.size(); }
    bool empty() const { return a.empty(); }
    int64_t top() const { return a[0]; }
    void clear() { a.clear(); }
    void up(int i)
    {
        const int64_t v = a[i];
        while (i > 0) {
            const int p = (i - 1) >> 1;
            if (!(v < a[p])) break;
            a[i] = a[p];
            i = p;
        }
        a[i] = v;
    }
    void down(int i)
    {
        const int n = (int)a.size();
        const int64_t v = a[i];
        for (;;) {
            int j = 2 * i + 1;
            if (j >= n) break;
            if (j + 1 < n && a[j + 1] < a[j]) ++j;
            if (!(a[j] < v)) break;
            a[i] = a[j];
            i = j;
        }
        a[i] = v;
    }
    void push(int64_t v) { a.push_back(v); up((int)a.size() - 1); }
    void update_top(int64_t v) { a[0] = v; down(0); }
    int64_t pop()
    {
        const int64_t r = a[0];
        a[0] = a.back();
        a.pop_back();
        if (!a.empty()) down(0);
        return r;
    }
};

// ScoreTracker.TwoPhaseTracker (B/graph/ScoreTracker.java:80-140): threshold searches stop once the 99th percentile of the
// last 500 scores falls below both the 100th best score seen and the threshold.  StatUtils.percentile = commons-math3's
// Percentile, LEGACY estimation: pos = 0.99 * 501, linear interpolation between the order statistics around it, in double.
struct TwoPhaseTracker {
    static constexpr int kRecent = 500, kBest = 100;
    double recent[kRecent];
    int recent_idx = 0, obs = 0, n_best = 0;
    int32_t best[kBest];  // min-heap of the best sortable-int scores (BoundedLongHeap :58-69)
    double threshold = 0.0;
    bool on = false;
    void reset(float thr)
    {
        on = thr > 0;
        n_best = 0;
        obs = 0;
        threshold = (double)thr;
    }
    static inline int32_t sortable(float f)
    {
        int32_t bits;
        if (f != f) bits = 0x7fc00000;
        else memcpy(&bits, &f, 4);
        return bits ^ ((bits >> 31) & 0x7fffffff);
    }
    void track(float score)
    {
        if (!on) return;
        const int32_t v = sortable(score);
        if (n_best < kBest) {
            int i = n_best++;
            while (i > 0 && best[(i - 1) >> 1] > v) {
                best[i] = best[(i - 1) >> 1];
                i = (i - 1) >> 1;
            }
            best[i] = v;
        } else if (!(v < best[0])) {
            int i = 0;
            for (;;) {
                int j = 2 * i + 1;
                if (j >= n_best) break;
                if (j + 1 < n_best && best[j + 1] < best[j]) ++j;
                if (!(best[j] < v)) break;
                best[i] = best[j];
                i = j;
            }
            best[i] = v;
        }
        recent[recent_idx] = (double)score;
        recent_idx = (recent_idx + 1) % kRecent;
        ++obs;
    }
    bool should_stop() const
    {
        if (!on || obs < kRecent || obs % 100 != 0) return false;
        double w[kRecent];
        memcpy(w, recent, sizeof(w));
        const double pos = (99.0 / 100.0) * (double)(kRecent + 1);
        const int ipos = (int)pos;  // floor: pos > 0
        std::nth_element(w, w + ipos - 1, w + kRecent);
        const double lower = w[ipos - 1];
        const double upper = *std::min_element(w + ipos, w + kRecent);
        const double window = lower + (pos - (double)ipos) * (upper - lower);
        const int32_t e = best[0];
        const int32_t bits = e ^ ((e >> 31) & 0x7fffffff);
        float worst_best;
        memcpy(&worst_best, &bits, 4);
        return window < (double)worst_best && window < threshold;
    }
};

struct QState {
    MaxHeap8 cand;                 // max-heap (best candidate on top)
    JMinHeap res;                  // approximateResults (top = worst kept)
    std::vector<int64_t> evicted;  // evictedResults (NodesUnsorted)
    IntSet visited;
    std::unique_ptr<TwoPhaseTracker> tracker;  // layer 0 of a threshold > 0 search
    bool active = true;
    int32_t origin = -1;
    int n_pending = 0;
    uint64_t fresh_mask[kMaxGraphDegree / 64] = {};  // fused layer 0: neighbours newly marked visited in this round (bit i = neighbour i)
    int64_t n_visited = 0, n_expanded = 0, n_expanded_base = 0;
    // CachingReranker (GraphSearcher.java:554-581): exact scores survive from search() to resume()
    std::unordered_map<int32_t, float> exact_cache;
    bool searched = false;
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
    // gs_defer: a batch on this graph had more than a tenth of its queries start over — its searches run without deferral from then on
    mutable std::atomic<int> defer_off{0};
    // device mirror for the device-resident traversal (k_gsearch.hip), built on first use; immutable afterwards
    struct DevLevel {
        int32_t *nbrs = nullptr, *hkeys = nullptr, *hvals = nullptr;
        uint32_t hmask = 0;
        int32_t hshift = 32;
    };
    std::vector<DevLevel> dev;
    std::mutex dev_mu;
    bool dev_ready = false;
    int dev_device = -1;
    int traversal = JV_TRAVERSAL_AUTO;
    // drop the device mirror (graph mutated, upload failed midway, or destruction); caller holds dev_mu or owns the graph
    void free_mirror()
    {
        if (!dev.empty() && dev_device >= 0) {
            (void)hipSetDevice(dev_device);
            for (DevLevel &d : dev) {
                if (d.nbrs != dev_level0) (void)hipFree(d.nbrs);
                (void)hipFree(d.hkeys);
                (void)hipFree(d.hvals);
            }
        }
        dev.clear();
        dev_ready = false;
        fused_checked = nullptr;
    }
    // level 0 living in CALLER-owned device memory (jv_hip_graph_set_level0_device): read in place by the device traversal,
    // may be rewritten by its owner between searches (incremental construction); no host copy exists
    const int32_t *dev_level0 = nullptr;
    const void *fused_checked = nullptr;  // the jv_fused whose neighbour table was last compared with level 0's adjacency
    uint64_t fused_checked_gen = 0;
    ~jv_graph() { free_mirror(); }
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
    // every kernel and the host searcher index code / vector / visited tables with these ids unchecked: validate once here
    for (size_t i = 0, n = (size_t)count * degree; i < n; ++i)
        JV_REQUIRE(neighbors[i] >= -1 && neighbors[i] < g->n_nodes, "graph_set_level: level %d row %zu holds neighbour id %d (n_nodes %lld)",
                   level, i / degree, neighbors[i], (long long)g->n_nodes);
    if (node_ids) {
        JV_REQUIRE(std::is_sorted(node_ids, node_ids + count), "graph_set_level: node ids must be ascending");
        JV_REQUIRE(node_ids[0] >= 0 && node_ids[count - 1] < g->n_nodes, "graph_set_level: node id out of range");
    }
    std::lock_guard<std::mutex> lk(g->dev_mu);
    g->free_mirror();  // a search may already have built the device mirror: it is stale now
    if (level == 0) g->dev_level0 = nullptr;
    jv_graph::Level &L = g->levels[level];
    L.count = count;
    L.degree = degree;
    L.nbrs.assign(neighbors, neighbors + (size_t)count * degree);
    L.nodes.clear();
    if (node_ids) L.nodes.assign(node_ids, node_ids + count);
    return JV_OK;
}

int jv_hip_graph_set_entry(jv_graph *g, int32_t node, int level)
{
    clear_error();
    JV_REQUIRE(g, "graph_set_entry: NULL graph");
    JV_REQUIRE(node >= 0 && node < g->n_nodes && level >= 0 && level < g->n_levels, "graph_set_entry: out of range");
    std::lock_guard<std::mutex> lk(g->dev_mu);
    g->free_mirror();  // the mirror covers levels 0..entry_level of the graph it was built from
    g->entry_node = node;
    g->entry_level = level;
    return JV_OK;
}

int jv_hip_graph_set_level0_device(jv_ctx *ctx, jv_graph *g, const int32_t *d_neighbors, int degree)
{
    clear_error();
    JV_REQUIRE(ctx && g && d_neighbors, "graph_set_level0_device: NULL argument");
    JV_REQUIRE(degree > 0 && degree <= 64, "graph_set_level0_device: degree %d outside 1..64", degree);
    JV_REQUIRE(is_device_ptr(d_neighbors), "graph_set_level0_device: the adjacency must be device memory (use jv_hip_graph_set_level for host rows)");
    std::lock_guard<std::mutex> lk(g->dev_mu);
    g->free_mirror();
    jv_graph::Level &L = g->levels[0];
    L.count = (int)g->n_nodes;
    L.degree = degree;
    L.nbrs.clear();
    L.nodes.clear();
    g->dev_level0 = d_neighbors;
    return JV_OK;
}

int jv_hip_graph_destroy(jv_graph *g)
{
    delete g;
    return JV_OK;
}

int jv_hip_graph_set_traversal(jv_graph *g, int mode)
{
    clear_error();
    JV_REQUIRE(g, "graph_set_traversal: NULL graph");
    JV_REQUIRE(mode == JV_TRAVERSAL_AUTO || mode == JV_TRAVERSAL_HOST || mode == JV_TRAVERSAL_DEVICE,
               "graph_set_traversal: unknown mode %d", mode);
    g->traversal = mode;
    return JV_OK;
}

// The fused layer-0 path takes neighbour IDS from the graph's level-0 adjacency and SCORES from slot i of the FusedPQ
// block: a FusedPQ written from a different (or reordered) adjacency would silently pair the wrong score with each id.
// Compared once per (graph, fused object, upload generation), in 16 MB pieces.
static int check_fused_matches_graph(jv_ctx *ctx, jv_graph *g, const jv_fused *fused)
{
    std::lock_guard<std::mutex> lk(g->dev_mu);
    if (g->fused_checked == (const void *)fused && g->fused_checked_gen == fused->generation) return JV_OK;
    const std::vector<int32_t> &nb = g->levels[0].nbrs;
    const size_t total = nb.size(), piece = (size_t)4 << 20;
    std::vector<int32_t> tmp(std::min(total, piece));
    for (size_t off = 0; off < total; off += piece) {
        const size_t n = std::min(piece, total - off);
        JV_HIP_CHECK(hipMemcpy(tmp.data(), fused->d_neighbors + off, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
        if (memcmp(tmp.data(), nb.data() + off, sizeof(int32_t) * n) != 0) {
            size_t i = 0;
            while (tmp[i] == nb[off + i]) ++i;
            const int deg = g->levels[0].degree;
            set_error("graph_search: the FusedPQ blocks were written from a different adjacency than the graph's level 0 (node %zu slot %zu: "
                      "fused %d, graph %d)", (off + i) / deg, (off + i) % deg, tmp[i], nb[off + i]);
            return JV_ERR_INVALID;
        }
    }
    g->fused_checked = fused;
    g->fused_checked_gen = fused->generation;
    return JV_OK;
}

// acceptOrds for a batch: a little-endian bit array over node ids per query (stride_words apart; 0 = one mask shared by all)
struct AcceptMask {
    const uint64_t *bits = nullptr;  // HOST memory for the host traversal, DEVICE memory for the device traversal
    int64_t stride_words = 0;
    const int32_t *exclude = nullptr;   // [Q] ExcludingBits(node) of the builder's searches: query q never returns exclude[q] (same memory rule)
    bool accepts(int q, int32_t node) const
    {
        if (exclude && exclude[q] == node) return false;
        return !bits || ((bits[(int64_t)q * stride_words + (node >> 6)] >> (node & 63)) & 1ull);
    }
};

// One GraphSearcher per query, kept between search() and resume() (jv_hip_searcher_*)
struct jv_searcher {
    const jv_graph *g = nullptr;
    jv_luts *luts = nullptr;
    const jv_codes *codes = nullptr;
    const jv_fused *fused = nullptr;
    const jv_vectors *vectors = nullptr;
    std::vector<std::unique_ptr<QState>> states;
    std::vector<float> queries;     // the last search()'s queries: resume rebuilds their tables
    std::vector<uint64_t> accept;   // this.acceptOrds (GraphSearcher.java:338), kept for resume
    int64_t accept_stride = 0;
    bool has_accept = false;
    int Q = 0;
    jv_vsf vsf = JV_DOT_PRODUCT;
    bool searched = false;
    // the calls since the last search() ran on the device traversal: the per-query candidate queues / visited sets were never
    // brought to the host.  resume() replays them (the traversal is deterministic): inside the session kernel (GsParams::n_phases)
    // while the history is short, else on the host searcher.  One record per call — its parameters and the evictedResults every
    // query held when it returned (what the next resume() pushes back into the candidates)
    bool device_searched = false;
    struct Call {
        int topK = 0, rerankK = 0;
        float threshold = 0.0f, floor = 0.0f;
        std::vector<int32_t> ev_off;     // Q + 1
        std::vector<long long> ev_keys;  // evictedResults of every query, query-major
    };
    std::vector<Call> history;
};

// What the plain jv_hip_graph_search entry points leave at their defaults
struct HostSearchOpts {
    AcceptMask accept;
    float threshold = 0.0f;      // search(..., threshold, ...) :222-243: TwoPhaseTracker + `score >= threshold` at layer 0
    float rerank_floor = 0.0f;   // NodeQueue.rerank's floor
    jv_searcher *session = nullptr;
    bool resume = false;         // resume(additionalK, rerankK) :538-547: continue layer 0 from the kept state
    int32_t *counts = nullptr;   // Q: results per query (host memory)
    int64_t *stats4 = nullptr;   // Q x 4: visitedCount, expandedCount, expandedCountBaseLayer, rerankedCount (host memory)
    float *worst = nullptr;      // Q: worstApproximateScoreInTopK (host memory)
};

static int copy_out(void *dst, const void *src_host, size_t bytes)
{
    if (bytes == 0) return JV_OK;
    if (is_device_ptr(dst)) JV_HIP_CHECK(hipMemcpy(dst, src_host, bytes, hipMemcpyHostToDevice));
    else memcpy(dst, src_host, bytes);
    return JV_OK;
}

// ---- reranking (GraphSearcher.reranking :471-507 + NodeQueue.rerank :160-230) over every query's approximateResults heap ARRAY
//      (`fin`): shared by the host traversal and by the device traversal of GraphSearcher objects (which rebuilds the array from
//      the kernel's addTopCandidate log).  Fills the outputs and, with a session, its evictedResults / CachingReranker state.
static int rerank_stage(jv_ctx *ctx, HostPool *pool, jv_luts *l, const jv_vectors *vectors, jv_vsf vsf, int kvsf, int Q, int topK,
                        const HostSearchOpts &opt, jv_searcher *ses, std::vector<std::vector<int64_t>> &fin,
                        const std::vector<int64_t> &q_visited, const std::vector<int64_t> &q_expanded,
                        const std::vector<int64_t> &q_expanded_base, int32_t *out_ids, float *out_scores, int64_t *stats)
{
    // ---- reranking :471-507.  The exact scores come from the GPU (one gather over every position that needs one); the
    //      selection itself is NodeQueue.rerank's loop (:160-230) over each query's heap ARRAY, on the host: an entry is kept
    //      while the bounded queue has room or its exact score is STRICTLY better than the worst kept, so exact-score ties at
    //      the K-th place resolve as in the reference. ----
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    int Rmax = 0;
    for (int q = 0; q < Q; ++q) Rmax = std::max(Rmax, (int)fin[q].size());
    std::vector<float> h_exact;
    std::vector<int32_t> need;  // Q x Rmax: node whose exact score the GPU computes for this position, -1 = none
    if (vectors && Rmax > 0) {
        const size_t c1 = (size_t)Q * Rmax;
        need.assign(c1, -1);
        pool->parallel_for(Q, [&](int lo, int hi) {
            for (int q = lo; q < hi; ++q) {
                const std::vector<int64_t> &r = fin[q];
                const QState *st = ses ? ses->states[(size_t)q].get() : nullptr;
                int above = 0, best = -1;
                float best_score = -INFINITY;
                for (int i = 0; i < (int)r.size(); ++i) {
                    const float sc = nq_score(r[i]);
                    if (sc > best_score) { best_score = sc; best = i; }
                    if (sc >= opt.rerank_floor) {
                        ++above;
                        const int32_t id = nq_node(r[i]);
                        if (!st || !st->exact_cache.count(id)) need[(size_t)q * Rmax + i] = id;
                    }
                }
                if (above == 0 && best >= 0) {  // nothing above the floor: the best one is reranked (:186-191)
                    const int32_t id = nq_node(r[best]);
                    if (!st || !st->exact_cache.count(id)) need[(size_t)q * Rmax + best] = id;
                }
            }
        });
        JV_TRY(ctx->h_in.reserve(sizeof(int32_t) * c1));
        JV_TRY(ctx->d_in.reserve(sizeof(int32_t) * c1 + sizeof(float) * c1 + sizeof(float) * (size_t)Q + 256));
        JV_TRY(ctx->h_out.reserve(sizeof(float) * c1));
        memcpy(ctx->h_in.ptr, need.data(), sizeof(int32_t) * c1);
        int32_t *d_cand = (int32_t *)ctx->d_in.ptr;
        float *d_cand_sc = (float *)(d_cand + c1);
        float *d_qnorm = d_cand_sc + c1;
        JV_HIP_CHECK(hipMemcpyAsync(d_cand, ctx->h_in.ptr, sizeof(int32_t) * c1, hipMemcpyHostToDevice, ctx->stream));
        JV_TRY(rerank_gather(ctx, vectors, l->d_raw_queries, Q, vsf, d_cand, Rmax, d_cand_sc, d_qnorm));
        JV_HIP_CHECK(hipMemcpyAsync(ctx->h_out.ptr, d_cand_sc, sizeof(float) * c1, hipMemcpyDeviceToHost, ctx->stream));
        JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        h_exact.assign((const float *)ctx->h_out.ptr, (const float *)ctx->h_out.ptr + c1);
    }
    std::vector<int32_t> r_ids((size_t)Q * topK, -1);
    std::vector<float> r_sc((size_t)Q * topK, -INFINITY);
    std::vector<int32_t> r_count((size_t)Q, 0);
    std::vector<int64_t> r_reranked((size_t)Q, 0);
    std::vector<float> r_worst((size_t)Q, INFINITY);
    pool->parallel_for(Q, [&](int lo, int hi) {
        JMinHeap rer;
        std::vector<int32_t> ids;
        std::vector<float> ex;
        for (int q = lo; q < hi; ++q) {
            std::vector<int64_t> &r = fin[q];
            QState *st = ses ? ses->states[(size_t)q].get() : nullptr;
            const int n = (int)r.size();
            int32_t *oid = r_ids.data() + (size_t)q * topK;
            float *osc = r_sc.data() + (size_t)q * topK;
            JMinHeap *from = &rer;
            JMinHeap approx;
            rer.clear();
            if (!vectors) {  // cachingReranker == null :478-487: the worst approximate results go to evictedResults
                approx.a.swap(r);
                while (approx.size() > topK) {
                    const int64_t k = approx.pop();
                    if (st) st->evicted.push_back(k);
                }
                from = &approx;
            } else {
                ids.assign((size_t)n, -1);
                ex.assign((size_t)n, 0.0f);
                int above = 0, best = -1;
                float best_score = -INFINITY;
                auto exact_of = [&](int i, int32_t id) {
                    if (need[(size_t)q * Rmax + i] >= 0) {
                        const float v = h_exact[(size_t)q * Rmax + i];
                        r_reranked[q]++;
                        if (st) st->exact_cache.emplace(id, v);
                        return v;
                    }
                    return st->exact_cache.find(id)->second;  // only a session skips the GPU for a position
                };
                for (int i = 0; i < n; ++i) {
                    const float sc = nq_score(r[i]);
                    if (sc > best_score) { best_score = sc; best = i; }
                    if (sc >= opt.rerank_floor) {
                        ids[i] = nq_node(r[i]);
                        ex[i] = exact_of(i, ids[i]);
                        ++above;
                    }
                }
                if (above == 0 && best >= 0) {
                    ids[best] = nq_node(r[best]);
                    ex[best] = exact_of(best, ids[best]);
                }
                auto approx_of = [&](int32_t node) {
                    for (int j = 0; j < n; ++j)
                        if (ids[j] == node) return nq_score(r[j]);
                    return -INFINITY;
                };
                for (int i = 0; i < n; ++i) {
                    if (ids[i] == -1) {
                        if (st) st->evicted.push_back(r[i]);
                        continue;
                    }
                    if (rer.size() < topK) {
                        rer.push(nq_encode(ids[i], ex[i]));
                    } else if (ex[i] > nq_score(rer.top())) {
                        if (st) {
                            const int32_t ev = nq_node(rer.top());
                            st->evicted.push_back(nq_encode(ev, approx_of(ev)));
                        }
                        rer.update_top(nq_encode(ids[i], ex[i]));
                    } else if (st) {
                        st->evicted.push_back(r[i]);
                    }
                }
                if (rer.size() >= topK)
                    for (int64_t k : rer.a) r_worst[q] = std::min(r_worst[q], approx_of(nq_node(k)));
            }
            const int nres = from->size();
            r_count[q] = nres;
            for (int i = nres - 1; i >= 0; --i) {  // :497-502: the worst is popped first
                const int64_t k = from->pop();
                oid[i] = nq_node(k);
                osc[i] = nq_score(k);
            }
            r.clear();
        }
    });
    JV_TRY(copy_out(out_ids, r_ids.data(), sizeof(int32_t) * (size_t)Q * topK));
    JV_TRY(copy_out(out_scores, r_sc.data(), sizeof(float) * (size_t)Q * topK));
    if (stats) {
        for (int q = 0; q < Q; ++q) {
            stats[2 * q] = q_visited[q];
            stats[2 * q + 1] = q_expanded[q];
        }
    }
    if (opt.stats4) {
        for (int q = 0; q < Q; ++q) {
            opt.stats4[4 * q] = q_visited[q];
            opt.stats4[4 * q + 1] = q_expanded[q];
            opt.stats4[4 * q + 2] = q_expanded_base[q];
            opt.stats4[4 * q + 3] = r_reranked[q];
        }
    }
    if (opt.counts) memcpy(opt.counts, r_count.data(), sizeof(int32_t) * (size_t)Q);
    if (opt.worst) memcpy(opt.worst, r_worst.data(), sizeof(float) * (size_t)Q);
    return JV_OK;
}

static int graph_search_host(jv_ctx *ctx, const jv_graph *g, jv_luts *l, const jv_codes *codes, const jv_fused *fused,
                        const jv_vectors *vectors, const float *queries, int Q, jv_vsf vsf, int topK, int rerankK,
                        int32_t *out_ids, float *out_scores, int64_t *stats, const HostSearchOpts &opt = HostSearchOpts())
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
    JV_REQUIRE(!(opt.threshold != opt.threshold) && !(opt.rerank_floor != opt.rerank_floor), "graph_search: NaN threshold / rerankFloor");
    for (int lv = 0; lv <= g->entry_level; ++lv)
        JV_REQUIRE(g->levels[lv].count > 0, "graph_search: level %d was never set", lv);
    if (Q == 0) return JV_OK;
    JV_REQUIRE(queries && out_ids && out_scores, "graph_search: NULL buffer");
    JV_TRY(use_device(ctx->device));
    const AcceptMask &accept = opt.accept;
    const float threshold = opt.threshold;
    jv_searcher *const ses = opt.session;

    const jv_decoder_kind kind = fused ? JV_DECODER_FUSED : JV_DECODER_PQ;
    JV_TRY(jv_hip_luts_build(ctx, l, queries, Q, vsf, kind));
    const int kvsf = to_kernel_vsf(vsf);
    if (vsf == JV_COSINE) {
        JV_TRY(ensure_code_norms(ctx, const_cast<jv_codes *>(codes)));
        if (fused) JV_TRY(ensure_fused_norms(ctx, const_cast<jv_fused *>(fused)));
    }
    HostPool *pool = get_pool(ctx);
    struct PoolSession {
        HostPool *p;
        explicit PoolSession(HostPool *pp) : p(pp) { p->begin(); }
        ~PoolSession() { p->end(); }
    } pool_session(pool);
    int W = 0;
    for (int lv = 0; lv <= g->entry_level; ++lv) W = std::max(W, g->levels[lv].degree);
    JV_REQUIRE(W <= kMaxGraphDegree, "graph_search: degree %d > %d is not supported", W, kMaxGraphDegree);

    // ---- slots: queries stream through a fixed number of traversal slots (continuous batching), in NG groups that
    //      alternate between the host phase and the GPU phase so that one group's scoring overlaps the other's
    //      heap work.  Per query nothing changes: its own pop -> score -> push sequence is the reference's.
    int S_total = 4096;
    if (const char *e = getenv("JVECTOR_HIP_GRAPH_SLOTS")) S_total = std::max(1, atoi(e));
    S_total = std::min(S_total, Q);
    int NG = (S_total >= 512) ? 2 : 1;
    if (const char *e = getenv("JVECTOR_HIP_GRAPH_GROUPS")) NG = std::max(1, std::min(4, atoi(e)));
    const int SG = (S_total + NG - 1) / NG;  // slots per group

    // luts_build may still be DMA-ing the queries out of the pinned staging buffer: drain before reusing it
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    // per group, control block (int32): [slot_query SG][origins SG][ord_index SG][ords SG*W]; scores SG*W floats
    const size_t ctrl_ints = (size_t)SG * (3 + W);
    const size_t sc_floats = (size_t)SG * W;
    JV_TRY(ctx->h_in.reserve(std::max(sizeof(int32_t) * ctrl_ints * NG, sizeof(int32_t) * (size_t)Q)));
    JV_TRY(ctx->h_out.reserve(std::max(sizeof(float) * sc_floats * NG, sizeof(float) * (size_t)Q)));
    JV_TRY(ctx->d_in.reserve(std::max(sizeof(int32_t) * ctrl_ints * NG, sizeof(int32_t) * (size_t)Q)));
    JV_TRY(ctx->d_out.reserve(std::max(sizeof(float) * sc_floats * NG, sizeof(float) * (size_t)Q)));

    // initializeInternal for every query: score the entry node (one gather of Q x 1)
    std::vector<float> entry_score((size_t)Q);
    if (!opt.resume) {
        int32_t *h = (int32_t *)ctx->h_in.ptr;
        for (int q = 0; q < Q; ++q) h[q] = g->entry_node;
        JV_HIP_CHECK(hipMemcpyAsync(ctx->d_in.ptr, h, sizeof(int32_t) * (size_t)Q, hipMemcpyHostToDevice, ctx->stream));
        {
            ProfScope ps(ctx, R_ADC);
            JV_TRY(launch_adc(ctx->stream, ctx, l->d_luts, l->d_bmag, Q, codes->M, kvsf, codes->d_codes, codes->d_norms,
                              codes->count, 0, 1, (const int32_t *)ctx->d_in.ptr, (float *)ctx->d_out.ptr));
        }
        JV_HIP_CHECK(hipMemcpyAsync(ctx->h_out.ptr, ctx->d_out.ptr, sizeof(float) * (size_t)Q, hipMemcpyDeviceToHost,
                                    ctx->stream));
        JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        memcpy(entry_score.data(), ctx->h_out.ptr, sizeof(float) * (size_t)Q);
    }

    struct Slot {
        QState own;           // a slot's own state, recycled from query to query ...
        QState *st = nullptr; // ... or the session's state of the query the slot is running
        int query = -1;
        int lvl = 0;
    };
    struct Group {
        std::vector<Slot> slots;
        int32_t *h_ctrl = nullptr, *d_ctrl = nullptr;
        float *h_sc = nullptr, *d_sc = nullptr;
        hipEvent_t ev = nullptr;
        bool launched = false;
        int n_active = 0, n_gather = 0;
    };
    std::vector<Group> groups((size_t)NG);
    for (int gi = 0; gi < NG; ++gi) {
        Group &G = groups[gi];
        const int lo = gi * SG, hi = std::min(S_total, lo + SG);
        G.slots = std::vector<Slot>((size_t)std::max(0, hi - lo));
        for (Slot &s : G.slots) s.st = &s.own;
        G.h_ctrl = (int32_t *)ctx->h_in.ptr + ctrl_ints * gi;
        G.d_ctrl = (int32_t *)ctx->d_in.ptr + ctrl_ints * gi;
        G.h_sc = (float *)ctx->h_out.ptr + sc_floats * gi;
        G.d_sc = (float *)ctx->d_out.ptr + sc_floats * gi;
        JV_HIP_CHECK(hipEventCreateWithFlags(&G.ev, hipEventDisableTiming));
    }
    struct EventGuard {
        std::vector<Group> &gs;
        ~EventGuard() { for (auto &G : gs) if (G.ev) (void)hipEventDestroy(G.ev); }
    } event_guard{groups};

    std::vector<std::vector<int64_t>> fin((size_t)Q);       // per query: approximateResults' heap ARRAY at the end of the search
    std::vector<int64_t> q_visited((size_t)Q, 0), q_expanded((size_t)Q, 0), q_expanded_base((size_t)Q, 0);
    std::atomic<int> next_q{0};
    const int deg0 = g->levels[0].degree;

    // visited-set capacity hint: a search marks roughly maxDegree/2 x rerankK nodes; starting at 1024 cost three
    // rehashes per query (~10 % of the host time at rerankK 150).  Growth stays automatic beyond the hint.
    int visited_cap = 1024;
    while (visited_cap < 32 * rerankK && visited_cap < (1 << 16)) visited_cap <<= 1;
    // getScoreTracker (ScoreTracker.java:38-58): layer 0 of a threshold search gets the TwoPhaseTracker, everything else none
    auto enter_layer = [&](Slot &s) {
        QState &st = *s.st;
        if (s.lvl == 0 && threshold > 0) {
            if (!st.tracker) st.tracker.reset(new TwoPhaseTracker());
            st.tracker->reset(threshold);
        } else if (st.tracker) {
            st.tracker->on = false;
        }
    };
    auto start_query = [&](Slot &s) -> bool {
        const int qi = next_q.fetch_add(1);
        if (qi >= Q) {
            s.query = -1;
            return false;
        }
        s.query = qi;
        s.st = ses ? ses->states[(size_t)qi].get() : &s.own;
        QState &st = *s.st;
        st.n_visited = st.n_expanded = st.n_expanded_base = 0;
        if (opt.resume) {  // searchLayer0 :459-469: the evicted results go back onto the candidate queue
            s.lvl = 0;
            for (int64_t k : st.evicted) st.cand.push(k);
            st.evicted.clear();
            st.res.clear();
        } else {           // initializeInternal :334-353
            s.lvl = g->entry_level;
            st.cand.clear();
            st.res.clear();
            st.evicted.clear();
            st.exact_cache.clear();
            st.visited.reset(visited_cap);
            st.visited.add(g->entry_node);
            st.cand.push(nq_encode(g->entry_node, entry_score[qi]));
            st.searched = true;
        }
        enter_layer(s);
        return true;
    };

    using clk = std::chrono::steady_clock;
    const bool timing = getenv("JVECTOR_HIP_GRAPH_TIMING") != nullptr;
    double t_host = 0, t_wait = 0;
    long n_rounds = 0;
    auto since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };

    // ---- host phase A: advance every slot of the group to its next expansion (searchOneLayer :421-433) ----
    auto pre = [&](Group &G) {
        const int SGn = (int)G.slots.size();
        int32_t *slot_query = G.h_ctrl, *origins = G.h_ctrl + SG, *ord_index = G.h_ctrl + 2 * SG;
        int32_t *ords = G.h_ctrl + 3 * SG;
        std::atomic<int> n_active{0}, n_gather{0};
        pool->parallel_for(SGn, [&](int lo, int hi) {
            int local_active = 0;
            for (int si = lo; si < hi; ++si) {
                Slot &s = G.slots[si];
                if (si + 1 < hi) {  // the next slot's heap roots have gone cold since the last round
                    const QState &nx = *G.slots[si + 1].st;
                    __builtin_prefetch(nx.cand.root_line());
                    if (!nx.res.empty()) __builtin_prefetch(nx.res.a.data());
                }
                s.st->origin = -1;
                s.st->n_pending = 0;
                slot_query[si] = -1;
                origins[si] = -1;
                ord_index[si] = -1;
                if (s.query < 0 && !start_query(s)) continue;
                for (;;) {
                    QState &st = *s.st;
                    const int rk = s.lvl > 0 ? 1 : rerankK;
                    const float thr = s.lvl > 0 ? 0.0f : threshold;  // upper layers run with threshold 0 (:276)
                    const TwoPhaseTracker *trk = (st.tracker && st.tracker->on) ? st.tracker.get() : nullptr;
                    bool layer_done = st.cand.empty();
                    int64_t top = 0;
                    float top_score = 0.0f;
                    if (!layer_done) {
                        top = st.cand.top();
                        top_score = nq_score(top);
                        layer_done = st.res.size() >= rk && top_score < nq_score(st.res.top());  // stopSearch :358-361
                        if (!layer_done && trk && trk->should_stop()) layer_done = true;          // :364-366
                    }
                    if (layer_done) {
                        if (s.lvl > 0) {  // setEntryPointsFromPreviousLayer :324-331
                            for (int64_t k : st.res.a) st.cand.push(k);
                            for (int64_t k : st.evicted) st.cand.push(k);
                            st.res.clear();
                            st.evicted.clear();
                            s.lvl--;
                            enter_layer(s);
                            continue;
                        }
                        fin[s.query].swap(st.res.a);
                        st.res.clear();
                        q_visited[s.query] = st.n_visited;
                        q_expanded[s.query] = st.n_expanded;
                        q_expanded_base[s.query] = st.n_expanded_base;
                        if (!start_query(s)) break;
                        continue;
                    }
                    st.cand.pop();
                    const int32_t node = nq_node(top);
                    // `topCandidateScore >= threshold` (:437) keeps weaker (with threshold 0: negative / NaN) scores out of
                    // the results; the node is expanded all the same.  acceptOrds gates the same decision at layer 0
                    // (upper layers run with Bits.ALL, :276)
                    if (!(top_score >= thr) || (s.lvl == 0 && !accept.accepts(s.query, node))) {
                    } else if (st.res.size() < rk) {  // addTopCandidate :515-530
                        st.res.push(top);
                    } else if (top_score > nq_score(st.res.top())) {
                        st.evicted.push_back(st.res.top());
                        st.res.update_top(top);
                    }
                    // "skip edge loading if we've found a local maximum and we have enough results" :441-444
                    if (trk && trk->should_stop() && st.cand.n >= rk - st.res.size()) continue;
                    if (s.lvl == 0) st.n_expanded_base++;
                    st.n_expanded++;
                    const int32_t *row = g->row(s.lvl, node);
                    if (!row) continue;  // node without a row at this level: nothing to score
                    if (s.lvl == 0 && fused) {
                        // scored from the packed block.  visited.mark happens here (the reference marks before it
                        // scores, OnDiskGraphIndex.java:646-650); the bitmask of newly marked neighbours tells the
                        // push phase which block scores to use.  No unvisited neighbour => no GPU round needed.
                        uint64_t any = 0;
                        for (int wd = 0; wd < (deg0 + 63) / 64; ++wd) st.fresh_mask[wd] = 0;
                        // the <= maxDegree probes are independent: issue their cache misses together
                        for (int i = 0; i < deg0 && row[i] >= 0; ++i) st.visited.prefetch(row[i]);
                        for (int i = 0; i < deg0; ++i) {
                            const int32_t nb = row[i];
                            if (nb < 0) break;
                            if (st.visited.add(nb)) {
                                st.fresh_mask[i >> 6] |= (1ull << (i & 63));
                                any = 1;
                            }
                        }
                        if (any == 0) continue;
                        origins[si] = node;
                    } else {
                        const int deg = g->levels[s.lvl].degree;
                        int32_t tmp[kMaxGraphDegree];
                        int np = 0;
                        for (int i = 0; i < deg; ++i) {
                            const int32_t nb = row[i];
                            if (nb < 0) break;
                            if (st.visited.add(nb)) tmp[np++] = nb;  // visited.mark, in neighbour order
                        }
                        if (np == 0) continue;
                        const int oi = n_gather.fetch_add(1);
                        ord_index[si] = oi;
                        int32_t *dst = ords + (size_t)oi * W;
                        for (int j = 0; j < W; ++j) dst[j] = j < np ? tmp[j] : -1;
                        st.n_pending = np;
                    }
                    st.origin = node;
                    slot_query[si] = s.query;
                    ++local_active;
                    break;
                }
            }
            n_active += local_active;
        });
        G.n_active = n_active.load();
        G.n_gather = n_gather.load();
    };

    auto launch = [&](Group &G) -> int {
        const int SGn = (int)G.slots.size();
        const size_t n_ctrl = (size_t)3 * SG + (size_t)G.n_gather * W;
        JV_HIP_CHECK(hipMemcpyAsync(G.d_ctrl, G.h_ctrl, sizeof(int32_t) * n_ctrl, hipMemcpyHostToDevice, ctx->stream));
        {
            ProfScope ps(ctx, R_ADC);
            JV_TRY(launch_frontier(ctx->stream, kvsf, l->d_luts, l->d_bmag, G.d_ctrl, G.d_ctrl + SG, G.d_ctrl + 2 * SG,
                                   G.d_ctrl + 3 * SG, fused, codes, G.d_sc, SGn, W, l->pq, l->d_queries));
        }
        JV_HIP_CHECK(hipMemcpyAsync(G.h_sc, G.d_sc, sizeof(float) * (size_t)SGn * W, hipMemcpyDeviceToHost, ctx->stream));
        JV_HIP_CHECK(hipEventRecord(G.ev, ctx->stream));
        G.launched = true;
        return JV_OK;
    };

    // ---- host phase B: push the scored neighbours (View.processNeighbors) ----
    auto post = [&](Group &G) {
        const int SGn = (int)G.slots.size();
        const int32_t *ord_index = G.h_ctrl + 2 * SG, *ords = G.h_ctrl + 3 * SG;
        pool->parallel_for(SGn, [&](int lo, int hi) {
            for (int si = lo; si < hi; ++si) {
                Slot &s = G.slots[si];
                QState &st = *s.st;
                if (si + 1 < hi) __builtin_prefetch(G.h_sc + (size_t)(si + 1) * W);
                if (st.origin < 0) continue;
                TwoPhaseTracker *trk = (st.tracker && st.tracker->on) ? st.tracker.get() : nullptr;
                const float *sc = G.h_sc + (size_t)si * W;
                if (ord_index[si] < 0) {  // fused layer 0
                    const int32_t *row = g->row(0, st.origin);
                    for (int wd = 0; wd < (deg0 + 63) / 64; ++wd) {
                        uint64_t mask = st.fresh_mask[wd];
                        while (mask) {  // ascending bit order == neighbour order
                            const int i = wd * 64 + __builtin_ctzll(mask);
                            mask &= mask - 1;
                            if (trk) trk->track(sc[i]);
                            st.cand.push(nq_encode(row[i], sc[i]));
                            st.n_visited++;
                        }
                    }
                } else {
                    const int32_t *o = ords + (size_t)ord_index[si] * W;
                    for (int j = 0; j < st.n_pending; ++j) {
                        if (trk) trk->track(sc[j]);
                        st.cand.push(nq_encode(o[j], sc[j]));
                        st.n_visited++;
                    }
                }
                // the next pop is (almost always) the current heap top: start fetching its adjacency row now
                if (s.lvl == 0 && !st.cand.empty()) {
                    const int32_t *nr = g->row(0, nq_node(st.cand.top()));
                    if (nr) {
                        __builtin_prefetch(nr);
                        __builtin_prefetch(nr + 16);
                    }
                }
            }
        });
    };

    // ---- the pipeline: while one group's frontier is on the GPU the other group is in its host phases ----
    for (auto &G : groups) {
        auto t0 = clk::now();
        pre(G);
        t_host += since(t0);
        if (G.n_active > 0) JV_TRY(launch(G));
    }
    for (;;) {
        bool any = false;
        for (auto &G : groups) {
            if (!G.launched) continue;
            any = true;
            auto tw = clk::now();
            JV_HIP_CHECK(hipEventSynchronize(G.ev));
            t_wait += since(tw);
            G.launched = false;
            ++n_rounds;
            auto t0 = clk::now();
            post(G);
            pre(G);
            t_host += since(t0);
            if (G.n_active > 0) JV_TRY(launch(G));
        }
        if (!any) break;
    }
    if (timing)
        fprintf(stderr, "[jv graph_search] Q=%d slots=%d groups=%d threads=%d rounds=%ld host %.2f ms, gpu-wait %.2f ms\n", Q,
                S_total, NG, pool->size(), n_rounds, t_host, t_wait);

    return rerank_stage(ctx, pool, l, vectors, vsf, kvsf, Q, topK, opt, ses, fin, q_visited, q_expanded, q_expanded_base, out_ids, out_scores, stats);
}

// ---------------------------------------------------------------------------------------------
// Device-resident traversal (k_gsearch.hip / gs_body.h): same search, the queues live on the GPU.
// ---------------------------------------------------------------------------------------------
static int ensure_device_graph(jv_ctx *ctx, jv_graph *g)
{
    std::lock_guard<std::mutex> lk(g->dev_mu);
    if (g->dev_ready) {
        JV_REQUIRE(g->dev_device == ctx->device, "graph: device mirror lives on device %d, search runs on device %d",
                   g->dev_device, ctx->device);
        return JV_OK;
    }
    g->free_mirror();  // leftovers of an upload that failed midway
    g->dev_device = ctx->device;
    g->dev.resize((size_t)g->entry_level + 1);
    auto upload = [&](int32_t **dst, const int32_t *src, size_t n) -> int {
        JV_HIP_CHECK(hipMalloc((void **)dst, sizeof(int32_t) * std::max<size_t>(n, 1)));
        JV_HIP_CHECK(hipMemcpy(*dst, src, sizeof(int32_t) * n, hipMemcpyHostToDevice));
        return JV_OK;
    };
    auto build = [&]() -> int {
        for (int lv = 0; lv <= g->entry_level; ++lv) {
            const jv_graph::Level &L = g->levels[lv];
            jv_graph::DevLevel &d = g->dev[lv];
            if (lv == 0 && g->dev_level0) {
                d.nbrs = const_cast<int32_t *>(g->dev_level0);  // caller-owned, read in place
                continue;
            }
            JV_TRY(upload(&d.nbrs, L.nbrs.data(), L.nbrs.size()));
            if (!L.nodes.empty()) {
                const GsLevelMap m = gs_build_level_map(L.nodes.data(), L.count);
                JV_TRY(upload(&d.hkeys, m.keys.data(), m.keys.size()));
                JV_TRY(upload(&d.hvals, m.vals.data(), m.vals.size()));
                d.hmask = m.mask;
                d.hshift = m.shift;
            }
        }
        return JV_OK;
    };
    const int rc = build();
    if (rc != JV_OK) {
        g->free_mirror();
        return rc;
    }
    g->dev_ready = true;
    return JV_OK;
}

// What a GraphSearcher-object search (jv_hip_searcher_search) takes from the device traversal instead of final results: the
// per-query addTopCandidate log (from which the host rebuilds approximateResults' heap array and the layer-0 evictedResults),
// the counters, and which queries could not be finished on the device.
struct DeviceSessionOut {
    float threshold = 0.0f;
    // resume(): the calls to replay before the one being served (jv_searcher::history) and the new call's parameters
    const std::vector<jv_searcher::Call> *history = nullptr;
    int new_rerankK = 0;
    int log_cap = 0;
    std::vector<int32_t> status, base, log_n;
    std::vector<int64_t> stats;      // Q x 2
    std::vector<long long> log;      // Q x log_cap
};

static int graph_search_device(jv_ctx *ctx, const jv_graph *g, jv_luts *l, const jv_codes *codes, const jv_fused *fused,
                               const jv_vectors *vectors, const float *queries, int Q, jv_vsf vsf, int topK, int rerankK,
                               int32_t *out_ids, float *out_scores, int64_t *stats, AcceptMask host_accept, AcceptMask dev_accept,
                               DeviceSessionOut *so = nullptr)
{
    const jv_decoder_kind kind = fused ? JV_DECODER_FUSED : JV_DECODER_PQ;
    // centred queries, query magnitudes, raw copy for the rerank — NO look-up tables: the kernel scores table-free
    JV_TRY(luts_prepare(ctx, l, queries, Q, vsf, kind, false));
    const int kvsf = to_kernel_vsf(vsf);
    if (vsf == JV_COSINE) {
        JV_TRY(ensure_code_norms(ctx, const_cast<jv_codes *>(codes)));
        if (fused) JV_TRY(ensure_fused_norms(ctx, const_cast<jv_fused *>(fused)));
    }
    JV_TRY(ensure_device_graph(ctx, const_cast<jv_graph *>(g)));
    const jv_pq *pq = l->pq;

    // ---- sizing: LDS tier, workers, per-worker scratch ----
    // register-allocation variant: waves per SIMD the kernel is compiled for — 2 (default: 225 VGPRs, the most codebook gathers
    // in flight per wave) or 4 (JVECTOR_HIP_GS_OCC=4: 128 VGPRs, no pair-lane scoring).  Measured on MI355X at 1M x 768:
    // 2 waves/SIMD + pair lanes 19.0 ms per 16384-query batch; 4 waves/SIMD 25.1 ms; a 3 waves/SIMD build (168 VGPRs, 12 waves
    // per CU) 28.3 ms — fewer gathers in flight per wave cost more than the extra waves hide (profiles/r2_sweeps.md).
    // PQ shapes outside the specialised builds (ragged / non-8 sub-vectors, other M) run the generic kernels: one lane per
    // neighbour, the per-subspace geometry read from the quantizer's device tables
    const bool generic = !graph_search_device_specialised(pq, codes, fused) || ctx_opt(ctx, "gs_generic", 0) != 0;  // (option: tests / benches)
    // (the generic kernels need 113 VGPRs: 4 waves per SIMD by default — their byte-wise code reads and short gathers are
    // latency-bound, more resident queries hide more of it)
    const int occ = ctx_opt(ctx, "gs_occ", generic ? 4 : 2) >= 4 ? 4 : 2;
    // pair-lane scoring (two lanes per neighbour) when no level has more than 32 neighbours; it needs an M/2 x 32 float
    // exchange area in LDS.  gs_pair = 0 turns it off.
    bool pair = occ == 2 && !generic && ctx_opt(ctx, "gs_pair", 1) != 0;
    for (int lv = 0; lv <= g->entry_level; ++lv) pair = pair && g->levels[lv].degree <= 32;
    // M >= 128: the pair form's two half rows + exchange indices no longer fit 256 VGPRs (156 / 588 bytes of scratch per
    // lane at M = 128 / 192, -Rpass-analysis=kernel-resource-usage) while the one-lane-per-neighbour form still does
    pair = pair && pq->M <= 96;
    // The WORKGROUP form (gx_body.h, k_gsearch_wgx.hip) — one query per workgroup / CU, the query's ADC table in LDS, a control wave
    // + expander waves that score adjacency rows ahead of time; plain searches, degrees <= 64.  A query takes ~0.3 ms in it against
    // ~1.5 ms in the one-wave form (10M x 768, rerankK 95), but a CU serves one query at a time instead of eight: it is the form
    // for SMALL batches — gs_wgx unset: batches of up to 4 queries per CU; 1 / 0: always / never.
    const long long wgx_opt = ctx_opt(ctx, "gs_wgx", -1);
    // (AUTO leaves a launch alone that carries tuning options of the one-wave form: tests and benches that pin them mean that form)
    bool one_wave_tuned = false;
    for (const char *o : {"gs_vcap_log2", "gs_v1_log2", "gs_grow", "gs_retry", "gs_occ", "gs_pair", "gs_cand_cap", "gs_waves_per_cu", "gs_prefetch", "gs_prof"})
        one_wave_tuned = one_wave_tuned || ctx_opt_is_set(ctx, o);
    bool wgx = !so && !generic && graph_search_wgx_supported(pq->M) &&
               (wgx_opt > 0 || (wgx_opt < 0 && !one_wave_tuned && Q <= 4 * ctx->num_cus));
    for (int lv = 0; lv <= g->entry_level; ++lv) wgx = wgx && g->levels[lv].degree <= 64;
    const int wgx_log = std::min(512, std::max(256, 4 * rerankK));   // push-log entries buffered in LDS (8 bytes each)
    int wgx_kps = 32;
    for (int lv = 0; lv <= g->entry_level; ++lv)
        if (g->levels[lv].degree > 32) wgx_kps = 64;
    const int wgx_waves = std::max(2, std::min(8, (int)ctx_opt(ctx, "gs_wgx_waves", 8)));
    const int wgx_slots = std::max(2, std::min((int)GX_MAX_SLOTS, (int)ctx_opt(ctx, "gs_wgx_slots", 16)));
    int evict_cap = GS_EVICT_CAP;
    const int wgx_cand_cap = 256;   // (gx_body.h GX_HOT: the control wave scans its candidate tier from registers, four keys per lane)
    // LDS of one workgroup: gs_wgx_per_cu workgroups (= control waves = queries in flight) share a CU; the table covers the first
    // gs_wgx_lut_m subspaces (default: as many as fit), the others are scored table-free by the expanders
    const int wgx_per_cu = std::max(1, std::min(4, (int)ctx_opt(ctx, "gs_wgx_per_cu", 1)));
    const size_t wgx_budget = ctx->lds_per_block / (size_t)wgx_per_cu - (wgx_per_cu > 1 ? 256 : 512);
    auto wgx_bytes = [&](int lg, int lm) { return graph_search_wgx_lds_bytes(pq->D, rerankK, wgx_cand_cap, evict_cap, lg, wgx_slots, wgx_kps, wgx_log, lm); };
    int wgx_lut_m = (int)ctx_opt(ctx, "gs_wgx_lut_m", 0);
    if (wgx) {
        const int idb = gs_idbits(g->n_nodes);
        int min_v1 = 0;   // room for an LDS tier of the visited set is set aside before the table takes the rest
        for (int lg = 12; lg >= 8 && min_v1 == 0; --lg)
            if (gs_v1_fits(lg, idb)) min_v1 = lg;
        if (wgx_lut_m <= 0) {
            wgx_lut_m = pq->M;
            while (wgx_lut_m > 16 && wgx_bytes(min_v1, wgx_lut_m) > wgx_budget) wgx_lut_m -= 16;
        }
        wgx_lut_m = std::max(16, std::min(pq->M, wgx_lut_m / 16 * 16));
        wgx = wgx_bytes(0, wgx_lut_m) <= wgx_budget;
    }
    if (wgx) pair = false;
    // Rows of 33 ... 64 neighbours with the codes read by ordinal (the builder's working rows: maxDegree x neighborOverflow): the
    // compacted pair form — a lane per neighbour for the visited probe, two lanes per FRESH neighbour for the score (four above
    // M = 96; gs_body.h "PAIRC").  gs_pairc = 0 turns it off.
    bool pairc = occ == 2 && !so && !wgx && !pair && !generic && !fused && pq->M <= 192 && ctx_opt(ctx, "gs_pair", 1) != 0 &&
                 ctx_opt(ctx, "gs_pairc", 1) != 0;
    for (int lv = 0; lv <= g->entry_level; ++lv) pairc = pairc && g->levels[lv].degree <= 64;
    // gs_ubr (default: on where it applies): the pair-lane kernel with the batch's upper-bound tables PREBUILT by a dense kernel and
    // held in the wave's registers, survivors compacted and scored eight lanes each, the candidate tier trimmed to what can still be
    // popped (gs_body.h "UBR").  No LDS beyond the pair form's.  Tables: M x 256 bytes per query of the batch.
    // ... and over the compacted fresh list of the builder's 33 ... 64-wide rows (gs_ubrc, default on): PAIRC + UBR
    const bool ubr = !so && !wgx && (pair || (pairc && ctx_opt(ctx, "gs_ubrc", kGsUbrcDefault) != 0)) && occ == 2 && !dev_accept.bits && !dev_accept.exclude &&
                     ctx_opt(ctx, "gs_ubr", kGsUbrDefault) != 0 && graph_search_ubr_supported(pq->M, kvsf)
#ifdef JV_EXPERIMENTAL
                     && ctx_opt(ctx, "gs_quad", 0) == 0   // (a launch that asks for the four-lane path of the plain pair kernel means that kernel)
#endif
        ;
    const int pair_M = (pair || pairc) ? pq->M : 0;
    int cand_cap = wgx ? wgx_cand_cap : std::max(128, (int)ctx_opt(ctx, "gs_cand_cap", (occ == 4 ? 512 : ((pair || pairc) ? 256 : 1024)))) & ~63;
    while (!wgx && cand_cap > 256 && graph_search_lds_bytes(pq->D, rerankK, cand_cap, pair_M, evict_cap) > 40 * 1024) cand_cap = (cand_cap / 2) & ~63;
    // Visited set, tier 1 (gs_body.h gs_visit1): a two-choice bucketed LDS table of 16-bit entries in whatever the other
    // per-worker structures leave of 160 KB / (4 x occ workers per CU).  Preference: the largest table first (4096 slots = 8 KB
    // hold ~3900 nodes: all but the longest searches of the headline workload, median ~2200 visited nodes), giving up
    // candidate-tier capacity before table size (LDS tier sizes 256 / 512 / 768 measured within 3 % of each other in round 2);
    // gs_v1_log2 = 0 turns the tier off, a positive value pins it.  Graphs too large for the entry format (more than 14
    // remainder bits) get none.
    const int idbits = gs_idbits(g->n_nodes);
    const int want_per_cu = 4 * occ;
    const size_t lut_lds = so ? gs_session_lds_bytes() : 0;  // (session kernels: the tracker's arrays)
    const size_t lds_budget = (160 * 1024) / (size_t)want_per_cu - 256 - lut_lds;
    int v1_log2 = 0;
    if (wgx) {
        // the workgroup owns the CU's LDS: the largest tier that fits next to the table (16384 slots hold every search of the
        // headline workload: p99.9 of the visited count is 4.8 k)
        const long long pin = ctx_opt(ctx, "gs_v1_log2", ctx_opt_is_set(ctx, "gs_vcap_log2") ? 0 : -1);
        for (int lg = pin > 0 ? (int)pin : 14; lg >= (pin > 0 ? (int)pin : 8) && pin != 0; --lg) {
            if (!gs_v1_fits(lg, idbits)) continue;
            if (wgx_bytes(lg, wgx_lut_m) <= wgx_budget) {
                v1_log2 = lg;
                break;
            }
        }
    } else {
        // (a pinned gs_vcap_log2 is how tests drive the overflow paths of tier 2: no LDS tier in front of it then, unless asked for)
        const long long pin = ctx_opt(ctx, "gs_v1_log2", ctx_opt_is_set(ctx, "gs_vcap_log2") ? 0 : -1);
        struct Caps { int cand, evict; };
        // (third choice: the evicted list takes what is left, but no less than 96 entries — round 3's first hardware run used 64
        // and sent 0.3 % of the 10M queries through a retry launch that cost 7 % of the step)
        const int cand_small = std::min(cand_cap, 128);
        const long long left = (long long)lds_budget - (long long)graph_search_lds_bytes(pq->D, rerankK, cand_small, pair_M, 1, 12) + 8;
        const Caps caps[3] = {{cand_cap, evict_cap}, {cand_small, evict_cap}, {cand_small, (int)std::max<long long>(96, std::min<long long>(evict_cap, left / 8))}};
        bool done = false;
        for (int lg = pin > 0 ? (int)pin : 12; lg >= (pin > 0 ? (int)pin : 10) && !done && pin != 0; --lg) {
            if (!gs_v1_fits(lg, idbits)) continue;
            for (const Caps &c : caps) {
                if (ctx_opt_is_set(ctx, "gs_cand_cap") && c.cand != cand_cap) continue;
                if (graph_search_lds_bytes(pq->D, rerankK, c.cand, pair_M, c.evict, lg) <= lds_budget || (pin > 0 && &c == &caps[2])) {
                    v1_log2 = lg;
                    cand_cap = c.cand;
                    evict_cap = c.evict;
                    done = true;
                    break;
                }
            }
        }
    }
    const size_t lds = wgx ? wgx_bytes(v1_log2, wgx_lut_m)
                           : graph_search_lds_bytes(pq->D, rerankK, cand_cap, pair_M, evict_cap, v1_log2) + lut_lds;
    if (lds > ctx->lds_per_block) {
        set_error("graph_search(device): rerankK %d needs %zu bytes of LDS per wave (limit %zu); use the host traversal", rerankK,
                  lds, ctx->lds_per_block);
        return JV_ERR_UNSUPPORTED;
    }
    int per_cu = (int)std::min<size_t>((size_t)want_per_cu, std::max<size_t>(1, (160 * 1024) / (lds + 256)));
    per_cu = std::max(1, (int)ctx_opt(ctx, "gs_waves_per_cu", per_cu));
    if (wgx) per_cu = wgx_per_cu;
    const int workers = std::max(1, std::min(Q, ctx->num_cus * per_cu));
    // JVECTOR_HIP_GS_VCAP_LOG2 overrides the visited-table size (tests use a tiny table to drive the host fallback)
    const int vcap_log2 = std::max(8, std::min(24, (int)ctx_opt(ctx, "gs_vcap_log2", gs_vcap_log2(rerankK))));
    const size_t vcap = (size_t)1 << vcap_log2;
    // pushes <= visited <= (tier-2 capacity vcap / 2) + (tier-1 capacity 1 << v1_log2): the spill tier cannot overflow first.
    // (Round 3's first builds sized it for tier 2 alone: the 0.24 % longest searches of every 10M batch overflowed it and went
    // through a 3.5 ms retry launch — found in the closing profile's kernel trace.)
    const size_t v1_slots = v1_log2 > 0 ? ((size_t)1 << v1_log2) : 0;
    const int spill_cap = (int)(vcap / 2 + v1_slots) + 64;
    JV_TRY(ctx->d_gs_visited.reserve(sizeof(int32_t) * vcap * (size_t)workers));
    JV_TRY(ctx->d_gs_spill.reserve(sizeof(long long) * (size_t)spill_cap * (size_t)workers));
    // growth pool: the few queries of a batch that outgrow the base table (a long search visits 3-6x the average) move to
    // one of these 8x tables inside the kernel instead of costing a second launch; JVECTOR_HIP_GS_GROW=0 turns it off
    const int big_log2 = std::min(24, vcap_log2 + 3);
    // (a pinned JVECTOR_HIP_GS_VCAP_LOG2 is how tests reach the retry / host-fallback paths: no pool then unless asked for)
    const bool grow_on = ctx_opt_is_set(ctx, "gs_grow") ? ctx_opt(ctx, "gs_grow", 1) != 0 : !ctx_opt_is_set(ctx, "gs_vcap_log2");
    // pool size: ~3 % of the batch (p99.9 of the visited count is 2.2x the median on the benched graphs, so well under 1 % of
    // the queries outgrow a base table sized at 64 x rerankK), at least 64, at most 2048 tables (1 GB at rerankK 110)
    const int big_count = (grow_on && big_log2 > vcap_log2) ? std::max(1, std::min(Q, std::max(64, std::min(2048, Q / 32)))) : 0;
    const int big_spill_cap = (int)(((size_t)1 << big_log2) / 2 + v1_slots) + 64;
    const size_t big_vis_bytes = sizeof(int32_t) * ((size_t)1 << big_log2) * (size_t)big_count;
    const size_t big_spill_off = (big_vis_bytes + 255) & ~(size_t)255;
    const size_t big_ctr_off = (big_spill_off + sizeof(long long) * (size_t)big_spill_cap * (size_t)big_count + 255) & ~(size_t)255;
    if (big_count > 0) {
        JV_TRY(ctx->d_gs_big.reserve(big_ctr_off + 256));
        JV_HIP_CHECK(hipMemsetAsync((char *)ctx->d_gs_big.ptr + big_ctr_off, 0, sizeof(uint32_t), ctx->stream));
    }
    // result staging: [ids Q*rk][scores Q*rk][qnorm Q][pad][stats Q*2 i64][status Q][counter]
    const size_t c1 = (size_t)Q * rerankK;
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        const size_t o = off;
        off = (off + bytes + 255) & ~(size_t)255;
        return o;
    };
    const size_t o_ids = carve(sizeof(int32_t) * c1), o_sc = carve(sizeof(float) * c1), o_qn = carve(sizeof(float) * (size_t)Q);
    const size_t o_stats = carve(sizeof(long long) * 2 * (size_t)Q), o_status = carve(sizeof(int32_t) * (size_t)Q);
    const size_t o_counter = carve(sizeof(uint32_t) * 2);
    const size_t o_qmap = carve(sizeof(int32_t) * (size_t)Q);
    bool gs_prof = !so && ctx_opt(ctx, "gs_prof", 0) != 0;
    if (gs_prof && pairc) {   // (the compacted pair form has no phase-clock build: say so instead of printing sixteen zeros)
        if (ctx_opt(ctx, "quiet", 0) == 0)
            fprintf(stderr, "[jv gs prof] the compacted pair form (rows of 33..64 neighbours: the builder's searches) has no phase-clock variant; gs_prof ignored for this launch\n");
        gs_prof = false;
    }
    const size_t o_prof = carve(sizeof(unsigned long long) * 36);
    // DEFER (gs_body.h, GsParams::defer): the register-table bound form over the row puts off the exact scores of what it meets above
    // level 1 behind the layer's best result; gs_defer = 0 scores everything at once (results are identical either way)
    // (an index on which more than a tenth of a batch's queries had to start over — data without neighbourhood structure above level 1 —
    //  searches without deferral from then on, unless gs_defer is set explicitly: the restarts cost more than the deferred scores save)
    const bool defer_on = !so && !wgx && ubr && pair && ctx_opt(ctx, "gs_defer", 1) != 0 &&
                          (ctx_opt_is_set(ctx, "gs_defer") || g->defer_off.load(std::memory_order_relaxed) == 0);
    // exact-score ties across the K-th place of the rerank are decided by the order of the reference's result-heap array
    // (NodeQueue.java:197-214): the traversal logs its addTopCandidate sequence (avg ~1 entry per expansion) and
    // rerank_tie_kernel rebuilds the reference's answer for the (rare) tied queries; an overflowed log sends the query to
    // the host searcher instead.  JVECTOR_HIP_GS_TIE_CHECK=0 turns the whole check off, JVECTOR_HIP_GS_PUSH_LOG=0 the log only.
    const bool tie_check = !so && vectors != nullptr && rerankK > topK && ctx_opt(ctx, "gs_tie_check", 1) != 0;
    int log_cap = 0;
    if (so || (tie_check && ctx_opt(ctx, "gs_push_log", 1) != 0)) {
        // (a session needs the whole addTopCandidate sequence: threshold searches accept far more candidates than they keep)
        log_cap = so ? std::max(4096, 8 * rerankK) : std::max(256, 4 * rerankK);
        const size_t budget = (size_t)1 << 30;
        if ((size_t)Q * log_cap * sizeof(long long) > budget) log_cap = (int)std::max<size_t>(64, budget / ((size_t)Q * sizeof(long long)));
        if (ctx_opt_is_set(ctx, "gs_push_log_cap")) log_cap = std::max(1, (int)ctx_opt(ctx, "gs_push_log_cap", log_cap));
    }
    const size_t o_log = carve(sizeof(long long) * (size_t)Q * (size_t)log_cap), o_log_n = carve(sizeof(int32_t) * (size_t)Q);
    const size_t o_base = carve(sizeof(int32_t) * (size_t)Q);
    JV_TRY(ctx->d_gs_out.reserve(off));
    char *base = (char *)ctx->d_gs_out.ptr;
    int32_t *d_cand = (int32_t *)(base + o_ids);
    float *d_cand_sc = (float *)(base + o_sc);
    float *d_qnorm = (float *)(base + o_qn);
    long long *d_stats = (long long *)(base + o_stats);
    int32_t *d_status = (int32_t *)(base + o_status);
    uint32_t *d_counter = (uint32_t *)(base + o_counter);
    JV_HIP_CHECK(hipMemsetAsync(d_counter, 0, sizeof(uint32_t), ctx->stream));
    if (gs_prof || ubr) JV_HIP_CHECK(hipMemsetAsync(base + o_prof, 0, sizeof(unsigned long long) * 36, ctx->stream));

    GsParams p{};
    for (int lv = 0; lv <= g->entry_level; ++lv) {
        const jv_graph::DevLevel &d = g->dev[lv];
        p.lv[lv].nbrs = d.nbrs;
        p.lv[lv].hkeys = d.hkeys;
        p.lv[lv].hvals = d.hvals;
        p.lv[lv].hmask = d.hmask;
        p.lv[lv].hshift = d.hshift;
        p.lv[lv].count = g->levels[lv].count;
        p.lv[lv].degree = g->levels[lv].degree;
    }
    p.entry_node = g->entry_node;
    p.entry_level = g->entry_level;
    p.codebooks = pq->d_codebooks;
    p.cq = l->d_queries;
    p.bmag = l->d_bmag;
    p.codes = codes->d_codes;
    p.code_norms = codes->d_norms;
    p.blocks = fused ? fused->d_blocks : nullptr;
    p.fused_norms = fused ? fused->d_norms : nullptr;
    p.D = pq->D;
    p.M = pq->M;
    p.deg0 = g->levels[0].degree;
    p.generic = generic ? 1 : 0;
    p.sub_uniform4 = (generic && pq->uniform && pq->max_size % 4 == 0) ? pq->max_size : 0;
    p.sub_sizes = pq->d_sizes;
    p.sub_offsets = pq->d_offsets;
    p.cb_offsets = reinterpret_cast<const long long *>(pq->d_cb_offsets);
    p.Q = Q;
    p.rerankK = rerankK;
    p.accept = (const unsigned long long *)dev_accept.bits;
    p.accept_stride = dev_accept.stride_words;
    p.exclude = dev_accept.exclude;
    p.visited = (int32_t *)ctx->d_gs_visited.ptr;
    p.vcap_log2 = vcap_log2;
    p.spill = (long long *)ctx->d_gs_spill.ptr;
    p.spill_cap = spill_cap;
    p.cand_cap = cand_cap;
    p.evict_cap = evict_cap;
    p.v1_log2 = v1_log2;
    p.v1_idbits = idbits;
    p.prefetch = (ctx_opt(ctx, "gs_prefetch", 0) != 0 && evict_cap >= 48 && !generic) ? 1 : 0;  // (dword touches: aligned rows only)
    if (ubr) {
        const size_t tab_bytes = (gs_ubr_tab_bytes(pq->M) * (size_t)Q + 255) & ~(size_t)255;
        JV_TRY(ctx->d_gs_ubr.reserve(tab_bytes + sizeof(float) * 4 * (size_t)Q));
        p.ubr = 1;
        p.ubr_tab = (const uint32_t *)ctx->d_gs_ubr.ptr;
        p.ubr_meta = (const float *)((const char *)ctx->d_gs_ubr.ptr + tab_bytes);
        p.ubr_trim = std::max(1, (int)ctx_opt(ctx, "gs_ubr_trim", 16));   // (round 6: trims run from registers and cost a fifth of what they did: 16 is 0.6 % faster than 48, profiles/r6_l)
        p.ubr_count = (unsigned long long *)(base + o_prof) + 15;
        ProfScope ps(ctx, R_LUT);
        JV_TRY(launch_ubr_tables(ctx->stream, kvsf, pq->d_codebooks, l->d_queries, Q, pq->M, (uint32_t *)ctx->d_gs_ubr.ptr,
                                 (float *)((char *)ctx->d_gs_ubr.ptr + tab_bytes)));
    }
    if (wgx) {
        p.prefetch = 0;
        p.wgx = 1;
        p.wgx_slots = wgx_slots;
        p.wgx_kps = wgx_kps;
        p.wgx_log = wgx_log;
        p.wgx_lut_m = wgx_lut_m;
        p.wgx_depth = (int)ctx_opt(ctx, "gs_wgx_depth", 1);
    }
    auto launch = [&](const GsParams &pp, int w) -> int {
        if (pp.ubr) return launch_graph_search_ubr(ctx->stream, kvsf, pp, w, lds);
        return wgx ? launch_graph_search_wgx(ctx->stream, kvsf, pp, w, 64 * wgx_waves) : launch_graph_search(ctx->stream, kvsf, pp, w, occ);
    };
    if (so) {
        p.session = 1;
        p.threshold = so->threshold;
        p.out_base = (int32_t *)(base + o_base);
        if (so->history && !so->history->empty()) {   // resume(): search() + the earlier resume() calls + this one, in one launch
            const std::vector<jv_searcher::Call> &h = *so->history;
            const int np = (int)h.size() + 1;
            std::vector<int32_t> off((size_t)(np - 1) * Q + 1);
            size_t total = 0;
            for (int t = 0; t < np - 1; ++t) {
                p.ph_rerankK[t] = h[(size_t)t].rerankK;
                p.ph_threshold[t] = h[(size_t)t].threshold;
                for (int q = 0; q < Q; ++q) {
                    off[(size_t)t * Q + q] = (int32_t)total;
                    total += (size_t)(h[(size_t)t].ev_off[(size_t)q + 1] - h[(size_t)t].ev_off[(size_t)q]);
                }
            }
            off.back() = (int32_t)total;
            p.ph_rerankK[np - 1] = so->new_rerankK;
            p.ph_threshold[np - 1] = so->threshold;
            p.n_phases = np;
            p.ph_Q = Q;
            const size_t off_bytes = (sizeof(int32_t) * off.size() + 15) & ~(size_t)15;
            JV_TRY(ctx->d_gs_extra.reserve(off_bytes + sizeof(long long) * std::max<size_t>(total, 1)));
            JV_HIP_CHECK(hipMemcpyAsync(ctx->d_gs_extra.ptr, off.data(), sizeof(int32_t) * off.size(), hipMemcpyHostToDevice, ctx->stream));
            size_t at = 0;
            for (int t = 0; t < np - 1; ++t) {
                const size_t n = h[(size_t)t].ev_keys.size();
                if (n) JV_HIP_CHECK(hipMemcpyAsync((char *)ctx->d_gs_extra.ptr + off_bytes + sizeof(long long) * at, h[(size_t)t].ev_keys.data(),
                                                   sizeof(long long) * n, hipMemcpyHostToDevice, ctx->stream));
                at += n;
            }
            JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));   // `off` goes out of scope with this block
            p.ph_extra_off = (const int32_t *)ctx->d_gs_extra.ptr;
            p.ph_extra = (const long long *)((const char *)ctx->d_gs_extra.ptr + off_bytes);
        }
    }
    if (big_count > 0) {
        p.big_visited = (int32_t *)ctx->d_gs_big.ptr;
        p.big_spill = (long long *)((char *)ctx->d_gs_big.ptr + big_spill_off);
        p.big_next = (uint32_t *)((char *)ctx->d_gs_big.ptr + big_ctr_off);
        p.big_count = big_count;
        p.big_log2 = big_log2;
        p.big_spill_cap = big_spill_cap;
    }
    p.pair = pair ? 1 : (pairc ? 2 : 0);
    // gs_quad = 1: the pair-lane kernels score the fresh neighbours of an expansion with at most 16 of them FOUR lanes each (half the
    // gather instructions, the same number of lane addresses).  Measured on the headline (10M x 768, rerankK 74, profiles/r4_s): 86.0 ms
    // per 131 072 queries against 80.9 ms — the vector-memory path charges lane addresses, not instructions, and the redistribution
    // (14 ds_bpermute, two barriers) comes on top.  Off by default; results are identical either way.
#ifdef JV_EXPERIMENTAL
    p.quad = ctx_opt(ctx, "gs_quad", 0) != 0 ? 1 : 0;
#else
    p.quad = 0;   // (an experimental variant: make EXPERIMENTAL=1)
#endif
    p.out_ids = d_cand;
    p.out_scores = d_cand_sc;
    p.out_stats = d_stats;
    p.out_status = d_status;
    p.next_query = d_counter;
    if (log_cap > 0) {
        p.push_log = (long long *)(base + o_log);
        p.push_log_n = (int32_t *)(base + o_log_n);
        p.push_log_cap = log_cap;
    }
    p.prof = gs_prof ? (unsigned long long *)(base + o_prof) : nullptr;
    if (defer_on) {
        p.defer = 1;
        p.defer_count = (unsigned long long *)(base + o_prof) + 33;
        p.defer_min_level = (int)std::max<long long>(1, ctx_opt(ctx, "gs_defer_min_level", 2));
    }
    // ---- round 6: the rerank's exact scores inside the traversal wave (gs_body.h gs_rr_round).  Full-resolution rows that
    //      exact_gather_tr_kernel would take (16-byte aligned, D % 8 == 0, the norm table for cosine), lists of <= 256 results, the
    //      the register-table bound form over the row (the only kernel that carries the code: gs_body.h RR):
    //      rows [0, rr_rows) of every list leave the kernel with their exact score, the packed-remainder launch does the rest.
    //      gs_fused_rerank = 0: the rerank stays a kernel of its own.  Same arithmetic, same bits either way.
    int rr_rows = 0;
    if (!so && !wgx && ubr && pair && vectors && !vectors->nvq && vectors->D == pq->D && gs_rr_lds_bytes() <= lds && ctx_opt(ctx, "gs_fused_rerank", 1) != 0) {
        if (vsf == JV_COSINE) JV_TRY(ensure_vector_norms(ctx, const_cast<jv_vectors *>(vectors)));
        rr_rows = exact_fused_rows(vectors->d_vecs, vectors->D, l->d_raw_queries, Q, kvsf, rerankK, vectors->d_sqnorm);
    }
    if (rr_rows > 0) {
        if (vsf == JV_COSINE) {
            ProfScope ps(ctx, R_EXACT);
            JV_TRY(launch_query_sqnorms(ctx->stream, l->d_raw_queries, vectors->D, Q, d_qnorm));
        }
        p.rr_vecs = vectors->d_vecs;
        p.rr_queries = l->d_raw_queries;
        p.rr_qnorm = d_qnorm;
        p.rr_vnorm = vectors->d_sqnorm;
        p.rr_n = vectors->count;
        p.rr_rows = rr_rows;
    }
    {
        ProfScope ps(ctx, R_GSEARCH);
        JV_TRY(launch(p, workers));
    }
    // ---- queries that outgrew the fixed-size structures (visited table half full, spill / evicted list full): run them
    //      again on the device with a visited table 8x, then 64x the size (few queries -> few, roomy workers); whatever
    //      still overflows goes to the host searcher below.  One 4-byte-per-query status read per batch.
    std::vector<int32_t> status((size_t)Q);
    std::vector<int> redo;
    size_t n_overflow_first = 0;
    {
        JV_TRY(ctx->h_out.reserve(sizeof(int32_t) * (size_t)Q));
        JV_HIP_CHECK(hipMemcpyAsync(ctx->h_out.ptr, d_status, sizeof(int32_t) * (size_t)Q, hipMemcpyDeviceToHost, ctx->stream));
        unsigned long long ubr_dropped = 0;
        if (ubr) JV_HIP_CHECK(hipMemcpyAsync(&ubr_dropped, base + o_prof + 15 * sizeof(unsigned long long), sizeof(ubr_dropped), hipMemcpyDeviceToHost, ctx->stream));
        unsigned long long defer_counts[3] = {0, 0, 0};
        if (defer_on) JV_HIP_CHECK(hipMemcpyAsync(defer_counts, base + o_prof + 33 * sizeof(unsigned long long), sizeof(defer_counts), hipMemcpyDeviceToHost, ctx->stream));
        JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ubr) ctx_stat_add(ctx, "gs_ubr_dropped", (long long)ubr_dropped);
        if (defer_on) {
            ctx_stat_add(ctx, "gs_deferred", (long long)defer_counts[0]);
            ctx_stat_add(ctx, "gs_defer_restarts", (long long)defer_counts[1]);
            if (Q >= 64 && defer_counts[1] * 10 > (unsigned long long)Q && g->defer_off.exchange(1, std::memory_order_relaxed) == 0)
                ctx_stat_add(ctx, "gs_defer_switched_off", 1);
        }
        memcpy(status.data(), ctx->h_out.ptr, sizeof(int32_t) * (size_t)Q);
        for (int q = 0; q < Q; ++q)
            if (status[q] != GS_OK) redo.push_back(q);
        n_overflow_first = redo.size();
    }
    // JVECTOR_HIP_GS_VCAP_LOG2 pins a (tiny) table so that tests reach the host fallback: no retry then, unless
    // JVECTOR_HIP_GS_RETRY=1 asks for it (the test of this very path)
    const bool retry = ctx_opt_is_set(ctx, "gs_retry") ? ctx_opt(ctx, "gs_retry", 1) != 0 : !ctx_opt_is_set(ctx, "gs_vcap_log2");
    for (int attempt = 1; attempt <= 2 && !redo.empty() && retry; ++attempt) {
        const int vlog2 = std::min(24, vcap_log2 + 3 * attempt);
        if (vlog2 <= vcap_log2 + 3 * (attempt - 1)) break;
        const size_t vcap2 = (size_t)1 << vlog2;
        const int spill2 = (int)(vcap2 / 2 + v1_slots) + 64;
        const int R = (int)redo.size();
        const int workers2 = std::max(1, std::min<int>({R, workers, (int)(((size_t)512 << 20) / (vcap2 * 12))}));
        JV_TRY(ctx->d_gs_visited.reserve(sizeof(int32_t) * vcap2 * (size_t)workers2));
        JV_TRY(ctx->d_gs_spill.reserve(sizeof(long long) * (size_t)spill2 * (size_t)workers2));
        memcpy(ctx->h_out.ptr, redo.data(), sizeof(int32_t) * (size_t)R);
        JV_HIP_CHECK(hipMemcpyAsync(base + o_qmap, ctx->h_out.ptr, sizeof(int32_t) * (size_t)R, hipMemcpyHostToDevice, ctx->stream));
        JV_HIP_CHECK(hipMemsetAsync(d_counter, 0, sizeof(uint32_t), ctx->stream));
        GsParams p2 = p;
        p2.qmap = (const int32_t *)(base + o_qmap);
        p2.Q = R;
        p2.visited = (int32_t *)ctx->d_gs_visited.ptr;
        p2.vcap_log2 = vlog2;
        p2.spill = (long long *)ctx->d_gs_spill.ptr;
        p2.spill_cap = spill2;
        p2.big_visited = nullptr;  // the retry pass has roomy tables of its own
        p2.big_count = 0;
        p2.big_log2 = 0;
        p2.prof = nullptr;
        p2.defer = 0;   // (the retry pass: few queries, nothing to win)
        {
            ProfScope ps(ctx, R_GSEARCH);
            JV_TRY(launch(p2, workers2));
        }
        JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // h_out is reused below
        JV_HIP_CHECK(hipMemcpyAsync(ctx->h_out.ptr, d_status, sizeof(int32_t) * (size_t)Q, hipMemcpyDeviceToHost, ctx->stream));
        JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        memcpy(status.data(), ctx->h_out.ptr, sizeof(int32_t) * (size_t)Q);
        std::vector<int> still;
        for (int q : redo)
            if (status[q] != GS_OK) still.push_back(q);
        redo.swap(still);
    }
    if (gs_prof) {
        unsigned long long h[33];
        JV_HIP_CHECK(hipMemcpyAsync(h, base + o_prof, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        const double e = (double)std::max<unsigned long long>(h[5], 1);
        fprintf(stderr, "[jv gs prof] clocks/expansion: pop %.0f  result %.0f  row+block+visited %.0f  score %.0f  push %.0f | expansions %llu "
                        "queries %llu  setup+epilogue clocks/query %.0f\n", h[0] / e, h[1] / e, h[2] / e, h[3] / e, h[4] / e, h[5], h[6],
                (double)h[7] / (double)std::max<unsigned long long>(h[6], 1));
        const double fs = (double)std::max<unsigned long long>(h[8] + h[9] + h[10] + h[11], 1);
        {
            const double nq = (double)std::max<unsigned long long>(h[6], 1);
            fprintf(stderr, "[jv gs prof] clocks/query outside the expansion loop: setup %.0f  level transitions %.0f  epilogue %.0f  table build %.0f\n",
                    h[12] / nq, h[13] / nq, h[14] / nq, h[15] / nq);
            fprintf(stderr, "[jv gs prof] levels above 0: %.2f expansions per query, %.0f clocks per query (searchOneLayer + transitions) = %.0f per expansion\n",
                    h[24] / nq, h[25] / nq, (double)h[25] / (double)std::max<unsigned long long>(h[24], 1));
            {
                const double eu = (double)std::max<unsigned long long>(h[24], 1);
                fprintf(stderr, "[jv gs prof] levels above 0, clocks/expansion: pop %.0f  result %.0f  row+visited %.0f  score %.0f  push %.0f | scoring passes %.2f per expansion, %.0f clocks per pass\n",
                        h[26] / eu, h[27] / eu, h[28] / eu, h[29] / eu, h[30] / eu, h[32] / eu, (double)h[31] / (double)std::max<unsigned long long>(h[32], 1));
            }
        }
        if (wgx)
            fprintf(stderr, "[jv gs prof] workgroup form, per expansion: row found in a slot %.3f (of those still being scored at use: %.3f of all)  "
                            "requested at the pop %.3f  rows requested ahead %.3f\n", h[8] / e, h[9] / e, h[10] / e, h[11] / e);
        else if (ubr) {
            fprintf(stderr, "[jv gs prof] register-table bound form, per expansion: bound + staging %.0f clocks  bound + staging + exact scores %.0f clocks  "
                            "trims %.0f clocks | dropped %.2f  exactly scored %.2f neighbours\n", h[8] / e, h[9] / e, h[10] / e, h[15] / e, h[11] / e);
            fprintf(stderr, "[jv gs prof] register-table bound form, per expansion: wait for the row %.0f clocks  scoring rounds %.0f  owner sum + finish %.0f  "
                            "passes %.2f | expansions with <= 4 / <= 8 / <= 16 survivors: %.3f / %.3f / %.3f\n", h[16] / e, h[17] / e, h[18] / e, h[19] / e,
                    h[20] / e, h[21] / e, h[22] / e);
        }
        else
        fprintf(stderr, "[jv gs prof] scored neighbours by fresh count of their expansion: <=8 %.3f  <=16 %.3f  <=24 %.3f  <=32 %.3f\n", h[8] / fs,
                h[9] / fs, h[10] / fs, h[11] / fs);
    }

    if (so) {  // GraphSearcher objects: the host finishes (searcher_search_device) from the log and the counters
        so->status.assign(status.begin(), status.end());
        so->base.resize((size_t)Q);
        so->log_n.resize((size_t)Q);
        so->stats.resize(2 * (size_t)Q);
        JV_HIP_CHECK(hipMemcpyAsync(so->base.data(), base + o_base, sizeof(int32_t) * (size_t)Q, hipMemcpyDeviceToHost, ctx->stream));
        JV_HIP_CHECK(hipMemcpyAsync(so->log_n.data(), base + o_log_n, sizeof(int32_t) * (size_t)Q, hipMemcpyDeviceToHost, ctx->stream));
        JV_HIP_CHECK(hipMemcpyAsync(so->stats.data(), d_stats, sizeof(long long) * 2 * (size_t)Q, hipMemcpyDeviceToHost, ctx->stream));
        JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        // only the used prefix of every query's log crosses PCIe: [Q][longest log] instead of [Q][log_cap]
        int longest = 0;
        bool overflowed = false;
        for (int q = 0; q < Q; ++q) {
            overflowed = overflowed || so->log_n[(size_t)q] > log_cap;
            longest = std::max(longest, std::min(so->log_n[(size_t)q], log_cap));
        }
        so->log_cap = overflowed ? -1 : std::max(longest, 1);   // -1: some log overflowed -> the caller falls back
        if (!overflowed) {
            so->log.resize((size_t)Q * (size_t)so->log_cap);
            JV_HIP_CHECK(hipMemcpy2DAsync(so->log.data(), sizeof(long long) * (size_t)so->log_cap, base + o_log, sizeof(long long) * (size_t)log_cap,
                                          sizeof(long long) * (size_t)so->log_cap, (size_t)Q, hipMemcpyDeviceToHost, ctx->stream));
            JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        }
        ctx_stat_add(ctx, "gs_calls_device", 1);
        ctx_stat_add(ctx, "gs_queries_device", Q);
        ctx_stat_add(ctx, "gs_session_calls_device", 1);
        return JV_OK;
    }
    // ---- reranking :471-507 on the device-resident candidates ----
    OutStage oi, osc;
    JV_TRY(stage_out_begin(ctx, out_ids, sizeof(int32_t) * (size_t)Q * topK, ctx->d_scratch2, &oi));
    JV_TRY(stage_out_begin(ctx, out_scores, sizeof(float) * (size_t)Q * topK, ctx->d_scratch3, &osc));
    JV_TRY(ctx->d_scratch.reserve(topk_scratch_bytes(Q, topK)));
    if (vectors && rr_rows > 0) {   // (the traversal scored rows [0, rr_rows) itself)
        if (rr_rows < rerankK) {
            ProfScope ps(ctx, R_EXACT);
            JV_TRY(launch_exact_gather_tail(ctx->stream, vectors->d_vecs, vectors->count, vectors->D, l->d_raw_queries, Q, kvsf, d_cand, rerankK, rr_rows,
                                            d_cand_sc, d_qnorm, vectors->d_sqnorm));
        }
    } else if (vectors) {
        JV_TRY(rerank_gather(ctx, vectors, l->d_raw_queries, Q, vsf, d_cand, rerankK, d_cand_sc, d_qnorm));
    }
    {
        ProfScope ps(ctx, R_TOPK);
        JV_TRY(launch_topk(ctx->stream, ctx, d_cand_sc, d_cand, Q, rerankK, rerankK, 0, topK, (int32_t *)oi.dev,
                           (float *)osc.dev, ctx->d_scratch.ptr));
    }
    // exact-score ties across the K-th place are decided by the order of the reference's result heap array
    // (NodeQueue.java:197-214), which only the host searcher holds: find those queries, re-run them there
    if (tie_check) {
        JV_HIP_CHECK(hipMemsetAsync(d_counter, 0, sizeof(uint32_t) * 2, ctx->stream));
        ProfScope ps(ctx, R_TOPK);
        RtParams rt{};
        rt.cand_sc = d_cand_sc;
        rt.cand_ids = d_cand;
        rt.R = rerankK;
        rt.out_sc = (float *)osc.dev;
        rt.out_ids = (int32_t *)oi.dev;
        rt.K = topK;
        rt.Q = Q;
        rt.push_log = p.push_log;
        rt.push_log_n = p.push_log_n;
        rt.log_cap = p.push_log_cap;
        rt.rerankK = rerankK;
        rt.status = d_status;
        rt.count = d_counter;
        JV_TRY(launch_rerank_ties(ctx->stream, rt));
    }
    JV_TRY(stage_out_end(ctx, oi));
    JV_TRY(stage_out_end(ctx, osc));
    // stage_out_end stages through ctx->h_out: fetch the per-query counters only after it is done with it
    const size_t stats_bytes = stats ? sizeof(long long) * 2 * (size_t)Q : 0;
    JV_TRY(ctx->h_out.reserve(stats_bytes + 64));
    unsigned int *h_ties = (unsigned int *)((char *)ctx->h_out.ptr + stats_bytes);
    h_ties[0] = h_ties[1] = 0;
    if (tie_check) JV_HIP_CHECK(hipMemcpyAsync(h_ties, d_counter, sizeof(uint32_t) * 2, hipMemcpyDeviceToHost, ctx->stream));
    if (stats) {
        long long *h_stats = (long long *)ctx->h_out.ptr;
        JV_HIP_CHECK(hipMemcpyAsync(h_stats, d_stats, sizeof(long long) * 2 * (size_t)Q, hipMemcpyDeviceToHost, ctx->stream));
        JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        for (int q = 0; q < Q; ++q) {
            stats[2 * q] = h_stats[2 * q];
            stats[2 * q + 1] = h_stats[2 * q + 1];
        }
    } else {
        JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    const size_t n_ties = h_ties[0], n_ties_resolved = h_ties[1];
    if (n_ties > 0) {
        JV_TRY(ctx->h_out.reserve(sizeof(int32_t) * (size_t)Q));
        JV_HIP_CHECK(hipMemcpyAsync(ctx->h_out.ptr, d_status, sizeof(int32_t) * (size_t)Q, hipMemcpyDeviceToHost, ctx->stream));
        JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        const int32_t *hs = (const int32_t *)ctx->h_out.ptr;
        for (int q = 0; q < Q; ++q)
            if (hs[q] == GS_RERANK_TIE) redo.push_back(q);
        std::sort(redo.begin(), redo.end());
    }
    ctx_stat_add(ctx, "gs_calls_device", 1);
    ctx_stat_add(ctx, "gs_queries_device", Q);
    ctx_stat_add(ctx, "gs_queries_retried", (long long)n_overflow_first);
    ctx_stat_add(ctx, "gs_ties_resolved_device", (long long)n_ties_resolved);
    ctx_stat_add(ctx, "gs_ties_to_host", (long long)n_ties);
    ctx_stat_add(ctx, "gs_queries_host_fallback", (long long)redo.size());
    ctx_stat_set(ctx, "gs_last_v1_log2", v1_log2);
    ctx_stat_set(ctx, "gs_last_workers_per_cu", per_cu);
    ctx_stat_set(ctx, "gs_last_wgx", wgx ? 1 : 0);
    ctx_stat_set(ctx, "gs_last_ubr", ubr ? 1 : 0);
    ctx_stat_set(ctx, "gs_last_rr_rows", rr_rows);   // rows of every list the traversal wave reranked itself (0: the rerank was a kernel of its own)
    ctx_stat_set(ctx, "gs_last_pair", pair ? 1 : (pairc ? 2 : 0));   // 1: pair lanes over the row, 2: over the compacted fresh list
    if (wgx) ctx_stat_add(ctx, "gs_calls_wgx", 1);
    if (ctx_opt(ctx, "graph_timing", 0) != 0)
        fprintf(stderr, "[jv graph_search device] Q=%d workers=%d (x%d/CU, occ %d, pair %d, wgx %d) lds=%zu cand_cap=%d evict_cap=%d v1_log2=%d vcap=%zu overflow=%zu rerank ties=%zu (+%zu to the host) -> host %zu\n", Q,
                workers, per_cu, occ, pair ? 1 : (pairc ? 2 : 0), wgx ? wgx_waves : 0, lds, cand_cap, evict_cap, v1_log2, vcap, n_overflow_first, n_ties_resolved, n_ties, redo.size());
    if (redo.empty()) return JV_OK;

    // ---- queries that outgrew the fixed-size device structures: same search on the host ----
    // A graph whose level 0 lives in caller-owned device memory (jv_hip_graph_set_level0_device) has no host adjacency: the
    // host searcher walks a temporary copy of it (a build-time graph is small next to its vectors; the case is rare: a query
    // that overflowed the 64x visited table, or an exact-score tie whose push log overflowed).
    jv_graph *gm = const_cast<jv_graph *>(g);
    std::unique_lock<std::mutex> loan_lock(gm->dev_mu, std::defer_lock);  // two contexts falling back at once: one copy at a time
    if (g->dev_level0) loan_lock.lock();
    struct Level0Loan {  // (declared after the lock: the copy is dropped before the lock is released)
        jv_graph *g;
        bool active = false;
        ~Level0Loan()
        {
            if (active) {
                g->levels[0].nbrs.clear();
                g->levels[0].nbrs.shrink_to_fit();
            }
        }
    } loan{gm};
    if (g->dev_level0 && g->levels[0].nbrs.empty()) {
        const size_t cells = (size_t)g->n_nodes * (size_t)g->levels[0].degree;
        try {
            gm->levels[0].nbrs.resize(cells);
        } catch (const std::bad_alloc &) {
            set_error("graph_search: %zu queries need the host searcher but the device-resident level 0 (%zu ids) cannot be copied to the host", redo.size(), cells);
            return JV_ERR_OOM;
        }
        loan.active = true;
        JV_HIP_CHECK(hipMemcpy(gm->levels[0].nbrs.data(), g->dev_level0, sizeof(int32_t) * cells, hipMemcpyDeviceToHost));
    }
    const int D = pq->D, R = (int)redo.size();
    std::vector<float> sub((size_t)R * D);
    const bool q_dev = is_device_ptr(queries);
    for (int i = 0; i < R; ++i) {
        const float *src = queries + (size_t)redo[i] * D;
        if (q_dev) JV_HIP_CHECK(hipMemcpy(sub.data() + (size_t)i * D, src, sizeof(float) * D, hipMemcpyDeviceToHost));
        else memcpy(sub.data() + (size_t)i * D, src, sizeof(float) * D);
    }
    std::vector<int32_t> sub_ids((size_t)R * topK);
    std::vector<float> sub_sc((size_t)R * topK);
    std::vector<int64_t> sub_stats((size_t)R * 2);
    std::vector<uint64_t> sub_mask;  // per-query masks of the redone queries, compacted like the queries themselves
    AcceptMask sub_accept = host_accept;
    if (host_accept.bits && host_accept.stride_words > 0) {
        sub_mask.resize((size_t)R * host_accept.stride_words);
        for (int i = 0; i < R; ++i)
            memcpy(sub_mask.data() + (size_t)i * host_accept.stride_words, host_accept.bits + (size_t)redo[i] * host_accept.stride_words,
                   sizeof(uint64_t) * (size_t)host_accept.stride_words);
        sub_accept.bits = sub_mask.data();
    }
    std::vector<int32_t> sub_excl;
    if (host_accept.exclude) {
        sub_excl.resize((size_t)R);
        for (int i = 0; i < R; ++i) sub_excl[(size_t)i] = host_accept.exclude[redo[i]];
        sub_accept.exclude = sub_excl.data();
    }
    HostSearchOpts sub_opt;
    sub_opt.accept = sub_accept;
    JV_TRY(graph_search_host(ctx, g, l, codes, fused, vectors, sub.data(), R, vsf, topK, rerankK, sub_ids.data(), sub_sc.data(),
                             sub_stats.data(), sub_opt));
    const bool ids_dev = is_device_ptr(out_ids), sc_dev = is_device_ptr(out_scores);
    for (int i = 0; i < R; ++i) {
        const size_t dst = (size_t)redo[i] * topK, src = (size_t)i * topK;
        if (ids_dev) JV_HIP_CHECK(hipMemcpy(out_ids + dst, sub_ids.data() + src, sizeof(int32_t) * topK, hipMemcpyHostToDevice));
        else memcpy(out_ids + dst, sub_ids.data() + src, sizeof(int32_t) * topK);
        if (sc_dev) JV_HIP_CHECK(hipMemcpy(out_scores + dst, sub_sc.data() + src, sizeof(float) * topK, hipMemcpyHostToDevice));
        else memcpy(out_scores + dst, sub_sc.data() + src, sizeof(float) * topK);
        if (stats) {
            stats[2 * redo[i]] = sub_stats[2 * (size_t)i];
            stats[2 * redo[i] + 1] = sub_stats[2 * (size_t)i + 1];
        }
    }
    return JV_OK;
}

int jv_hip_graph_search(jv_ctx *ctx, const jv_graph *g, jv_luts *l, const jv_codes *codes, const jv_fused *fused,
                        const jv_vectors *vectors, const float *queries, int Q, jv_vsf vsf, int topK, int rerankK,
                        int32_t *out_ids, float *out_scores, int64_t *stats)
{
    return jv_hip_graph_search_filtered(ctx, g, l, codes, fused, vectors, queries, Q, vsf, topK, rerankK, nullptr, 0, out_ids, out_scores,
                                        stats);
}

static int graph_search_filtered_impl(jv_ctx *ctx, const jv_graph *g, jv_luts *l, const jv_codes *codes, const jv_fused *fused,
                                      const jv_vectors *vectors, const float *queries, int Q, jv_vsf vsf, int topK, int rerankK,
                                      const uint64_t *accept_bits, int64_t accept_stride_words, const int32_t *exclude_dev, int32_t *out_ids,
                                      float *out_scores, int64_t *stats);

int jv_hip_graph_search_filtered(jv_ctx *ctx, const jv_graph *g, jv_luts *l, const jv_codes *codes, const jv_fused *fused,
                                 const jv_vectors *vectors, const float *queries, int Q, jv_vsf vsf, int topK, int rerankK,
                                 const uint64_t *accept_bits, int64_t accept_stride_words, int32_t *out_ids, float *out_scores,
                                 int64_t *stats)
{
    return graph_search_filtered_impl(ctx, g, l, codes, fused, vectors, queries, Q, vsf, topK, rerankK, accept_bits, accept_stride_words, nullptr,
                                      out_ids, out_scores, stats);
}

// exclude_dev: [Q] node ids in DEVICE memory or nullptr — ExcludingBits(node) of the builder's own searches (jv::graph_search_excluding)
static int graph_search_filtered_impl(jv_ctx *ctx, const jv_graph *g, jv_luts *l, const jv_codes *codes, const jv_fused *fused,
                                      const jv_vectors *vectors, const float *queries, int Q, jv_vsf vsf, int topK, int rerankK,
                                      const uint64_t *accept_bits, int64_t accept_stride_words, const int32_t *exclude_dev, int32_t *out_ids,
                                      float *out_scores, int64_t *stats)
{
    clear_error();
    JV_REQUIRE(ctx && g && l && codes, "graph_search: NULL argument");
    CtxBusy busy(ctx);
    JV_REQUIRE(busy.ok, "graph_search: this jv_ctx is already inside a call on another thread (one context per host thread)");
    // Traversal: the graph's setting, overridden by JVECTOR_HIP_GRAPH_TRAVERSAL=host|device.  AUTO = the device-resident
    // traversal wherever it applies (256-cluster codebooks, degree <= 512, queues fit LDS: specialised kernels for uniform 8-dim
    // sub-vectors at M = 16 ... 192, the generic form for every other quantizer), the host searcher otherwise.
    int mode = (int)ctx_opt(ctx, "graph_traversal", g->traversal);
    if (mode != JV_TRAVERSAL_HOST && mode != JV_TRAVERSAL_DEVICE) mode = JV_TRAVERSAL_AUTO;
    const bool was_auto = mode == JV_TRAVERSAL_AUTO;
    if (mode == JV_TRAVERSAL_AUTO) {
        int Wd = 0;
        for (int lv = 0; lv <= g->entry_level && lv < (int)g->levels.size(); ++lv) Wd = std::max(Wd, g->levels[lv].degree);
        const bool fits = g->entry_node >= 0 && rerankK > 0 && l->pq && codes->pq == l->pq &&
                          graph_search_device_supported(l->pq, codes, fused, Wd, g->entry_level + 1) &&
                          graph_search_lds_bytes(l->pq->D, rerankK, 256, 0) <= std::min<size_t>(ctx->lds_per_block, 40 * 1024);
        mode = fits ? JV_TRAVERSAL_DEVICE : JV_TRAVERSAL_HOST;
    }
    // acceptOrds: the host traversal reads the masks from host memory, the device traversal from device memory
    const int64_t words = g->n_nodes > 0 ? (g->n_nodes + 63) / 64 : 0;
    JV_REQUIRE(!accept_bits || accept_stride_words == 0 || accept_stride_words >= words,
               "graph_search: accept_stride_words %lld is smaller than the %lld words one mask needs", (long long)accept_stride_words,
               (long long)words);
    const size_t mask_words = !accept_bits ? 0 : (accept_stride_words == 0 ? (size_t)words : (size_t)accept_stride_words * (size_t)std::max(Q, 0));
    AcceptMask host_accept, dev_accept;
    std::vector<uint64_t> host_copy;
    std::vector<int32_t> host_excl;
    if (exclude_dev && Q > 0) {   // (the host copy serves the few queries the device traversal hands back, and the host traversal)
        JV_TRY(use_device(ctx->device));
        host_excl.resize((size_t)Q);
        JV_HIP_CHECK(hipMemcpy(host_excl.data(), exclude_dev, sizeof(int32_t) * (size_t)Q, hipMemcpyDefault));
        host_accept.exclude = host_excl.data();
        dev_accept.exclude = exclude_dev;
    }
    if (accept_bits && Q > 0) {
        JV_TRY(use_device(ctx->device));
        host_accept.stride_words = dev_accept.stride_words = accept_stride_words;
        if (is_device_ptr(accept_bits)) {
            dev_accept.bits = accept_bits;
            host_copy.resize(mask_words);
            JV_HIP_CHECK(hipMemcpy(host_copy.data(), accept_bits, sizeof(uint64_t) * mask_words, hipMemcpyDeviceToHost));
            host_accept.bits = host_copy.data();
        } else {
            host_accept.bits = accept_bits;
        }
    }
    if (fused && Q > 0 && fused->count == g->n_nodes && fused->maxDegree == g->levels[0].degree &&
        g->levels[0].nbrs.size() == (size_t)g->n_nodes * fused->maxDegree) {
        JV_TRY(use_device(ctx->device));
        JV_TRY(check_fused_matches_graph(ctx, const_cast<jv_graph *>(g), fused));
    }
    if (g->dev_level0 && Q > 0) {
        JV_REQUIRE(mode == JV_TRAVERSAL_DEVICE, "graph_search: a graph whose level 0 lives in device memory needs the device traversal "
                                                "(shape unsupported by it, or JV_TRAVERSAL_HOST was requested)");
        JV_REQUIRE(!fused, "graph_search: FusedPQ blocks cannot be checked against a device-resident, mutable adjacency; search it with the code store");
    }
    if (mode != JV_TRAVERSAL_DEVICE || Q == 0) {
        if (Q > 0) {
            ctx_stat_add(ctx, "gs_calls_host", 1);
            if (was_auto) {  // say so: the host searcher is ~13x slower than the device traversal
                ctx_stat_add(ctx, "gs_calls_host_auto", 1);
                int64_t n_auto = 0;
                (void)jv_hip_ctx_get_stat(ctx, "gs_calls_host_auto", &n_auto);
                if (n_auto == 1 && ctx_opt(ctx, "quiet", 0) == 0)
                    fprintf(stderr, "[jvector_hip] graph_search: JV_TRAVERSAL_AUTO takes the HOST searcher for this search (the device traversal needs "
                                    "256-cluster codebooks, degree <= 512, <= %d levels and queues that fit LDS); counter "
                                    "gs_calls_host_auto counts further calls\n", GS_MAX_LEVELS);
            }
        }
        HostSearchOpts opt;
        opt.accept = host_accept;
        return graph_search_host(ctx, g, l, codes, fused, vectors, queries, Q, vsf, topK, rerankK, out_ids, out_scores, stats, opt);
    }

    JV_REQUIRE(topK > 0, "graph_search: topK must be positive");
    JV_REQUIRE(rerankK >= topK, "rerankK %d must be >= topK %d", rerankK, topK);  // GraphSearcher.java:233
    JV_REQUIRE(g->entry_node >= 0, "graph_search: the graph has no entry node");
    JV_REQUIRE(codes->pq == l->pq && codes->count >= g->n_nodes, "graph_search: code store does not match the graph");
    JV_REQUIRE(!fused || (fused->pq == l->pq && fused->count == g->n_nodes && fused->maxDegree == g->levels[0].degree),
               "graph_search: fused blocks do not match the graph");
    JV_REQUIRE(!vectors || (vectors->D == l->pq->D && vectors->count >= g->n_nodes), "graph_search: vectors mismatch");
    JV_REQUIRE(Q <= l->capacity, "graph_search: Q=%d exceeds the LUT capacity %d", Q, l->capacity);
    int W = 0;
    for (int lv = 0; lv <= g->entry_level; ++lv) {
        JV_REQUIRE(g->levels[lv].count > 0, "graph_search: level %d was never set", lv);
        W = std::max(W, g->levels[lv].degree);
    }
    JV_REQUIRE(queries && out_ids && out_scores, "graph_search: NULL buffer");
    JV_TRY(use_device(ctx->device));
    if (!graph_search_device_supported(l->pq, codes, fused, W, g->entry_level + 1)) {
        set_error("graph_search: the device traversal needs 256-cluster codebooks, degree <= 512 and <= %d levels; use the host traversal",
                  GS_MAX_LEVELS);
        return JV_ERR_UNSUPPORTED;
    }
    if (accept_bits && !dev_accept.bits) {  // host masks: the kernel needs a device copy
        JV_TRY(ctx->d_gs_mask.reserve(sizeof(uint64_t) * mask_words));
        JV_HIP_CHECK(hipMemcpy(ctx->d_gs_mask.ptr, accept_bits, sizeof(uint64_t) * mask_words, hipMemcpyHostToDevice));
        dev_accept.bits = (const uint64_t *)ctx->d_gs_mask.ptr;
    }
    return graph_search_device(ctx, g, l, codes, fused, vectors, queries, Q, vsf, topK, rerankK, out_ids, out_scores, stats, host_accept,
                               dev_accept);
}

// ---------------------------------------------------------------------------------------------
// GraphSearcher objects: search with every option, then resume (jvector_hip.h)
// ---------------------------------------------------------------------------------------------
int jv_hip_searcher_create(jv_ctx *ctx, const jv_graph *g, jv_luts *l, const jv_codes *codes, const jv_fused *fused,
                           const jv_vectors *vectors, jv_searcher **out)
{
    clear_error();
    JV_REQUIRE(ctx && g && l && codes && out, "searcher_create: NULL argument");
    JV_REQUIRE(!g->dev_level0, "searcher_create: a graph whose level 0 lives in device memory has no host adjacency to walk");
    jv_searcher *s = new jv_searcher();
    s->g = g;
    s->luts = l;
    s->codes = codes;
    s->fused = fused;
    s->vectors = vectors;
    *out = s;
    return JV_OK;
}

int jv_hip_searcher_destroy(jv_searcher *s)
{
    delete s;
    return JV_OK;
}

// search() of Q GraphSearcher objects on the DEVICE traversal (session kernels, gs_body.h SES): threshold admission, the
// TwoPhaseTracker stop and acceptOrds run inside the kernel; the host then rebuilds every query's approximateResults heap ARRAY
// by replaying the kernel's addTopCandidate log through the reference's own push / updateTop sequence (:515-530) — which also
// yields the layer-0 evictedResults — and runs the shared rerank stage (rerankFloor, CachingReranker, worstApproximateScoreInTopK).
// *done = false: the shape is outside the session kernels' coverage, the traversal is pinned to the host, or some query
// outgrew the device structures / its log — the caller runs the whole batch on the host searcher instead (same answers).
// search() — or, with `resume`, resume() after calls that all ran here (s->history): the session kernel replays them and goes on
static int searcher_search_device(jv_ctx *ctx, jv_searcher *s, int Q, int topK, int rerankK, float threshold, float rerankFloor, bool resume,
                                  int32_t *out_ids, float *out_scores, int32_t *out_counts, int64_t *stats, float *worst, bool *done)
{
    *done = false;
    if (resume && (s->history.empty() || (int)s->history.size() >= GS_MAX_PHASES)) return JV_OK;
    int rk_max = rerankK;   // LDS result array / output stride of the launch: the largest rerankK of the replayed calls
    if (resume)
        for (const jv_searcher::Call &c : s->history) rk_max = std::max(rk_max, c.rerankK);
    const jv_graph *g = s->g;
    jv_luts *l = s->luts;
    const jv_pq *pq = l->pq;
    if (Q == 0 || !pq) return JV_OK;
    int mode = (int)ctx_opt(ctx, "graph_traversal", g->traversal);
    if (mode == JV_TRAVERSAL_HOST) return JV_OK;
    int Wd = 0;
    for (int lv = 0; lv <= g->entry_level && lv < (int)g->levels.size(); ++lv) {
        if (g->levels[lv].count <= 0) return JV_OK;
        Wd = std::max(Wd, g->levels[lv].degree);
    }
    const bool fits = g->entry_node >= 0 && topK > 0 && rerankK >= topK && s->codes->pq == pq && s->codes->count >= g->n_nodes &&
                      Q <= l->capacity && graph_search_device_supported(pq, s->codes, s->fused, Wd, g->entry_level + 1) &&
                      graph_search_session_supported(pq->M) &&
                      (!s->fused || (s->fused->pq == pq && s->fused->count == g->n_nodes && s->fused->maxDegree == g->levels[0].degree)) &&
                      (!s->vectors || (s->vectors->D == pq->D && s->vectors->count >= g->n_nodes)) &&
                      graph_search_lds_bytes(pq->D, rk_max, 256, 0) + gs_session_lds_bytes() <= std::min<size_t>(ctx->lds_per_block, 40 * 1024);
    if (!fits) {
        if (mode == JV_TRAVERSAL_DEVICE) ctx_stat_add(ctx, "gs_session_calls_host_unsupported", 1);
        return JV_OK;
    }
    JV_TRY(use_device(ctx->device));
    AcceptMask host_accept, dev_accept;
    if (s->has_accept) {
        host_accept.bits = s->accept.data();
        host_accept.stride_words = s->accept_stride;
        JV_TRY(ctx->d_gs_mask.reserve(sizeof(uint64_t) * s->accept.size()));
        JV_HIP_CHECK(hipMemcpy(ctx->d_gs_mask.ptr, s->accept.data(), sizeof(uint64_t) * s->accept.size(), hipMemcpyHostToDevice));
        dev_accept.bits = (const uint64_t *)ctx->d_gs_mask.ptr;
        dev_accept.stride_words = s->accept_stride;
    }
    DeviceSessionOut so;
    so.threshold = threshold;
    if (resume) {
        so.history = &s->history;
        so.new_rerankK = rerankK;
    }
    const bool timing = ctx_opt(ctx, "graph_timing", 0) != 0;
    const auto t_start = std::chrono::steady_clock::now();
    JV_TRY(graph_search_device(ctx, g, l, s->codes, s->fused, s->vectors, s->queries.data(), Q, s->vsf, std::min(topK, rk_max), rk_max, nullptr,
                               nullptr, nullptr, host_accept, dev_accept, &so));
    for (int q = 0; q < Q; ++q)
        if (so.log_cap < 0 || so.status[(size_t)q] != GS_OK || so.log_n[(size_t)q] < 0 || so.log_n[(size_t)q] > so.log_cap) {
            ctx_stat_add(ctx, "gs_session_calls_host_overflow", 1);
            return JV_OK;  // (rare: one query outgrew the device structures or its log) -> the whole batch on the host
        }
    const auto t_kernel = std::chrono::steady_clock::now();
    // approximateResults + layer-0 evictedResults of every query from its addTopCandidate sequence
    std::vector<std::vector<int64_t>> fin((size_t)Q);
    std::vector<int64_t> q_visited((size_t)Q), q_expanded((size_t)Q), q_expanded_base((size_t)Q);
    HostPool *pool = get_pool(ctx);
    pool->begin();
    pool->parallel_for(Q, [&](int lo, int hi) {
        JMinHeap res;
        for (int q = lo; q < hi; ++q) {
            QState &st = *s->states[(size_t)q];
            st.cand.clear();
            st.res.clear();
            st.evicted.clear();                       // search(): initializeInternal; resume(): searchLayer0 moved it to the candidates
            if (!resume) st.exact_cache.clear();      // the CachingReranker lives as long as the search does (:568-576)
            st.searched = true;
            res.clear();
            const long long *log = so.log.data() + (size_t)q * (size_t)so.log_cap;
            for (int i = 0; i < so.log_n[(size_t)q]; ++i) {
                const int64_t top = (int64_t)log[i];
                if (res.size() < rerankK) {
                    res.push(top);
                } else if (nq_score(top) > nq_score(res.top())) {
                    st.evicted.push_back(res.top());
                    res.update_top(top);
                }
            }
            fin[(size_t)q] = res.a;
            st.n_visited = q_visited[(size_t)q] = so.stats[2 * (size_t)q];
            st.n_expanded = q_expanded[(size_t)q] = so.stats[2 * (size_t)q + 1];
            st.n_expanded_base = q_expanded_base[(size_t)q] = so.base[(size_t)q];
        }
    });
    pool->end();
    const auto t_replay = std::chrono::steady_clock::now();
    HostSearchOpts opt;
    opt.threshold = threshold;
    opt.rerank_floor = rerankFloor;
    opt.session = s;
    std::vector<int32_t> counts((size_t)Q);
    std::vector<int64_t> st4((size_t)Q * 4);
    std::vector<float> w((size_t)Q);
    opt.counts = counts.data();
    opt.stats4 = st4.data();
    opt.worst = w.data();
    pool->begin();
    const int rc = rerank_stage(ctx, pool, l, s->vectors, s->vsf, to_kernel_vsf(s->vsf), Q, topK, opt, s, fin, q_visited, q_expanded, q_expanded_base,
                                out_ids, out_scores, nullptr);
    pool->end();
    JV_TRY(rc);
    if (timing) {
        const auto t_end = std::chrono::steady_clock::now();
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return std::chrono::duration<double, std::milli>(b - a).count();
        };
        long long log_total = 0;
        for (int q = 0; q < Q; ++q) log_total += so.log_n[(size_t)q];
        fprintf(stderr, "[jv searcher objects] Q=%d %s: tables + traversal + log download %.2f ms, log replay %.2f ms (%.0f entries per query), rerank stage %.2f ms\n",
                Q, resume ? "resume" : "search", ms(t_start, t_kernel), ms(t_kernel, t_replay), (double)log_total / Q, ms(t_replay, t_end));
    }
    if (out_counts) JV_TRY(copy_out(out_counts, counts.data(), sizeof(int32_t) * (size_t)Q));
    if (stats) JV_TRY(copy_out(stats, st4.data(), sizeof(int64_t) * 4 * (size_t)Q));
    if (worst) JV_TRY(copy_out(worst, w.data(), sizeof(float) * (size_t)Q));
    // this call joins the history: its parameters and what every query's evictedResults holds now
    if (!resume) s->history.clear();
    s->history.emplace_back();
    jv_searcher::Call &c = s->history.back();
    c.topK = topK;
    c.rerankK = rerankK;
    c.threshold = threshold;
    c.floor = rerankFloor;
    c.ev_off.resize((size_t)Q + 1);
    size_t total = 0;
    for (int q = 0; q < Q; ++q) {
        c.ev_off[(size_t)q] = (int32_t)total;
        total += s->states[(size_t)q]->evicted.size();
    }
    c.ev_off[(size_t)Q] = (int32_t)total;
    c.ev_keys.resize(total);
    static_assert(sizeof(long long) == sizeof(int64_t), "NodeQueue keys are 64-bit");
    for (int q = 0; q < Q; ++q) {
        const std::vector<int64_t> &ev = s->states[(size_t)q]->evicted;
        if (!ev.empty()) memcpy(c.ev_keys.data() + c.ev_off[(size_t)q], ev.data(), sizeof(int64_t) * ev.size());
    }
    if (resume) ctx_stat_add(ctx, "gs_session_resume_device", 1);
    *done = true;
    return JV_OK;
}

static int searcher_run(jv_ctx *ctx, jv_searcher *s, int Q, int topK, int rerankK, float threshold, float rerankFloor, bool resume,
                        int32_t *out_ids, float *out_scores, int32_t *out_counts, int64_t *stats, float *worst)
{
    const jv_graph *g = s->g;
    // (whatever traversal serves the call: the host searcher has the same check, the session kernels would not stop on a NaN)
    JV_REQUIRE(!(threshold != threshold) && !(rerankFloor != rerankFloor), "graph_search: NaN threshold / rerankFloor");
    if (s->fused && Q > 0 && s->fused->count == g->n_nodes && s->fused->maxDegree == g->levels[0].degree &&
        g->levels[0].nbrs.size() == (size_t)g->n_nodes * s->fused->maxDegree) {
        JV_TRY(use_device(ctx->device));
        JV_TRY(check_fused_matches_graph(ctx, const_cast<jv_graph *>(g), s->fused));
    }
    if (!resume) {
        s->device_searched = false;
        s->history.clear();
        bool done = false;
        JV_TRY(searcher_search_device(ctx, s, Q, topK, rerankK, threshold, rerankFloor, false, out_ids, out_scores, out_counts, stats, worst, &done));
        if (done) {
            s->device_searched = true;
            return JV_OK;
        }
    } else if (s->device_searched) {
        // resume() after calls that ran on the device: the candidate queues / visited sets never left it.  The session kernel
        // replays those calls and continues (one launch); if it cannot (history too long, a structure or the log overflowed),
        // the same history is replayed on the host searcher instead and the searcher stays there.
        bool done = false;
        JV_TRY(searcher_search_device(ctx, s, Q, topK, rerankK, threshold, rerankFloor, true, out_ids, out_scores, out_counts, stats, worst, &done));
        if (done) return JV_OK;
        const std::vector<jv_searcher::Call> calls = std::move(s->history);
        s->history.clear();
        for (size_t i = 0; i < calls.size(); ++i) {
            HostSearchOpts ro;
            ro.threshold = calls[i].threshold;
            ro.rerank_floor = calls[i].floor;
            ro.session = s;
            ro.resume = i > 0;
            if (s->has_accept) {
                ro.accept.bits = s->accept.data();
                ro.accept.stride_words = s->accept_stride;
            }
            std::vector<int32_t> ids((size_t)Q * calls[i].topK);
            std::vector<float> sc((size_t)Q * calls[i].topK);
            JV_TRY(graph_search_host(ctx, g, s->luts, s->codes, s->fused, s->vectors, s->queries.data(), Q, s->vsf, calls[i].topK, calls[i].rerankK,
                                     ids.data(), sc.data(), nullptr, ro));
        }
        s->device_searched = false;
        ctx_stat_add(ctx, "gs_session_resume_replays", 1);
    }
    HostSearchOpts opt;
    opt.threshold = threshold;
    opt.rerank_floor = rerankFloor;
    opt.session = s;
    opt.resume = resume;
    if (s->has_accept) {
        opt.accept.bits = s->accept.data();
        opt.accept.stride_words = s->accept_stride;
    }
    std::vector<int32_t> counts((size_t)Q);
    std::vector<int64_t> st4((size_t)Q * 4);
    std::vector<float> w((size_t)Q);
    opt.counts = counts.data();
    opt.stats4 = st4.data();
    opt.worst = w.data();
    JV_TRY(graph_search_host(ctx, g, s->luts, s->codes, s->fused, s->vectors, s->queries.data(), Q, s->vsf, topK, rerankK, out_ids,
                             out_scores, nullptr, opt));
    if (out_counts) JV_TRY(copy_out(out_counts, counts.data(), sizeof(int32_t) * (size_t)Q));
    if (stats) JV_TRY(copy_out(stats, st4.data(), sizeof(int64_t) * 4 * (size_t)Q));
    if (worst) JV_TRY(copy_out(worst, w.data(), sizeof(float) * (size_t)Q));
    return JV_OK;
}

int jv_hip_searcher_search(jv_ctx *ctx, jv_searcher *s, const float *queries, int Q, jv_vsf vsf, int topK, int rerankK, float threshold,
                           float rerankFloor, const uint64_t *accept_bits, int64_t accept_stride_words, int32_t *out_ids,
                           float *out_scores, int32_t *out_counts, int64_t *stats, float *worst)
{
    clear_error();
    JV_REQUIRE(ctx && s, "searcher_search: NULL argument");
    CtxBusy busy(ctx);
    JV_REQUIRE(busy.ok, "searcher_search: this jv_ctx is already inside a call on another thread (one context per host thread)");
    JV_REQUIRE(Q >= 0 && (Q == 0 || queries), "searcher_search: NULL queries");
    JV_REQUIRE(s->luts->pq, "searcher_search: the look-up tables have no quantizer");
    const jv_graph *g = s->g;
    const int D = s->luts->pq->D;
    const int64_t words = (g->n_nodes + 63) / 64;
    JV_REQUIRE(!accept_bits || accept_stride_words == 0 || accept_stride_words >= words,
               "searcher_search: accept_stride_words %lld is smaller than the %lld words one mask needs", (long long)accept_stride_words,
               (long long)words);
    JV_TRY(use_device(ctx->device));
    s->searched = false;
    s->Q = Q;
    s->vsf = vsf;
    s->queries.resize((size_t)Q * D);
    if (Q > 0) {
        if (is_device_ptr(queries)) JV_HIP_CHECK(hipMemcpy(s->queries.data(), queries, sizeof(float) * (size_t)Q * D, hipMemcpyDeviceToHost));
        else memcpy(s->queries.data(), queries, sizeof(float) * (size_t)Q * D);
    }
    s->has_accept = accept_bits != nullptr;
    s->accept_stride = accept_stride_words;
    s->accept.clear();
    if (accept_bits && Q > 0) {
        const size_t mask_words = accept_stride_words == 0 ? (size_t)words : (size_t)accept_stride_words * (size_t)Q;
        s->accept.resize(mask_words);
        if (is_device_ptr(accept_bits)) JV_HIP_CHECK(hipMemcpy(s->accept.data(), accept_bits, sizeof(uint64_t) * mask_words, hipMemcpyDeviceToHost));
        else memcpy(s->accept.data(), accept_bits, sizeof(uint64_t) * mask_words);
    }
    while ((int)s->states.size() < Q) s->states.emplace_back(new QState());
    JV_TRY(searcher_run(ctx, s, Q, topK, rerankK, threshold, rerankFloor, false, out_ids, out_scores, out_counts, stats, worst));
    s->searched = true;
    return JV_OK;
}

int jv_hip_searcher_resume(jv_ctx *ctx, jv_searcher *s, int additionalK, int rerankK, int32_t *out_ids, float *out_scores,
                           int32_t *out_counts, int64_t *stats, float *worst)
{
    clear_error();
    JV_REQUIRE(ctx && s, "searcher_resume: NULL argument");
    CtxBusy busy(ctx);
    JV_REQUIRE(busy.ok, "searcher_resume: this jv_ctx is already inside a call on another thread (one context per host thread)");
    JV_REQUIRE(s->searched, "searcher_resume: resume() is only valid after search() (GraphSearcher.java:533-536)");
    return searcher_run(ctx, s, s->Q, additionalK, rerankK, 0.0f, 0.0f, true, out_ids, out_scores, out_counts, stats, worst);
}

}  // extern "C"

namespace jv {
// GraphSearcher.search with acceptOrds = ExcludingBits(exclude[q]) per query (GraphIndexBuilder.improveConnections :518): for the
// builder, whose batches make a per-query bit mask over every node impractical.  exclude: [Q] ids in device memory.
int graph_search_excluding(jv_ctx *ctx, const jv_graph *g, jv_luts *l, const jv_codes *codes, const float *queries, int Q, jv_vsf vsf, int topK,
                           int rerankK, const int32_t *exclude, int32_t *out_ids, float *out_scores, int64_t *stats)
{
    return graph_search_filtered_impl(ctx, g, l, codes, nullptr, nullptr, queries, Q, vsf, topK, rerankK, nullptr, 0, exclude, out_ids, out_scores,
                                      stats);
}
}  // namespace jv
