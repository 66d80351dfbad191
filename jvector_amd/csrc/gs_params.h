// gs_params.h — launch parameters of the device-resident graph search (gs_body.h / k_gsearch.hip), shared with
// the host driver in graph_search.cpp.  Plain data only.
#pragma once

#include <cstddef>
#include <cstdint>

namespace jv {

constexpr int GS_MAX_LEVELS = 32;
constexpr int GS_EVICT_CAP = 128;
constexpr int GS_MAX_PHASES = 8;   // session kernels: a search() and up to 7 resume() calls replayed in one launch
enum : int32_t { GS_OK = 0, GS_OVERFLOW = 1, GS_RERANK_TIE = 2 /* set by rerank_tie_kernel, not by the traversal */ };
enum : int32_t { GS_RESTART = 3 };   // internal to gs_search_one (DEFER): the query starts over without deferral; never written out

struct GsLevel {
    const int32_t *nbrs;    // count x degree, packed rows padded with -1
    const int32_t *hkeys;   // upper levels: open-addressing map node id -> row (keys, -1 = empty); level 0: nullptr
    const int32_t *hvals;
    uint32_t hmask;         // table size - 1
    int32_t hshift;         // 32 - log2(table size)
    int32_t count, degree;
};

struct GsParams {
    GsLevel lv[GS_MAX_LEVELS];
    int32_t entry_node, entry_level;
    // scoring
    const float *codebooks;   // [M][256][8]
    const float *cq;          // [Q][D] centred queries
    const float *bmag;        // [Q] query magnitude (cosine)
    const uint8_t *codes;     // [n][M]
    const float *code_norms;  // [n] decoded magnitudes (cosine)
    const uint8_t *blocks;    // layer-0 FusedPQ blocks [n][deg0][M], or nullptr
    const float *fused_norms; // [n][deg0] (cosine)
    int32_t D, M, deg0;
    // every other PQ shape (the GENERIC kernels, CH16 = 0 in gs_body.h: ragged sub-vectors, sizes other than 8, any M): the
    // per-subspace geometry replaces the [M][256][8] assumption; one lane per neighbour, no register table
    int32_t generic;              // 1: sub_sizes / sub_offsets / cb_offsets describe `codebooks`
    int32_t sub_uniform4;         // > 0: every sub-vector has this size, a multiple of 4 (codebook rows are read as 16-byte words)
    const int32_t *sub_sizes;     // [M]
    const int32_t *sub_offsets;   // [M] offset of sub-vector m inside a vector
    const long long *cb_offsets;  // [M] float offset of codebook m inside `codebooks`
    // search
    int32_t Q, rerankK;
    const int32_t *qmap;      // nullptr: work items 0..Q-1 ARE the query indices; else item i runs query qmap[i] (retry pass)
    const unsigned long long *accept;  // acceptOrds bit array (bit n of word n / 64) or nullptr = Bits.ALL; layer 0 only
    long long accept_stride;           // words between the masks of consecutive queries; 0 = one mask for the batch
    const int32_t *exclude;            // [Q] or nullptr: ExcludingBits(node) — query q's own node may be traversed but never becomes a result
                                       // (GraphIndexBuilder.java:518,623: the builder's searches); layer 0 only, like accept
    // per-worker scratch
    int32_t *visited;         // [workers][1 << vcap_log2]
    int32_t vcap_log2;
    long long *spill;         // [workers][spill_cap]
    int32_t spill_cap;
    // growth pool: a query whose visited table reaches half full claims one of big_count roomier tables (+ spill tier),
    // re-inserts its visited set there and carries on; pool empty -> GS_OVERFLOW as before.  nullptr = no pool.
    int32_t *big_visited;     // [big_count][1 << big_log2], claimed tables are cleared by their claimant
    long long *big_spill;     // [big_count][big_spill_cap]
    uint32_t *big_next;       // claim counter (zeroed by the host before the launch)
    int32_t big_count, big_log2, big_spill_cap;
    int32_t cand_cap;         // LDS tier capacity (>= 256)
    int32_t evict_cap;        // capacity of the upper-layer evicted list in LDS (0 = GS_EVICT_CAP)
    int32_t pair;             // 1: pair-lane scoring (every degree <= 32; LDS has the M/2 x 32 exchange area); 2: over the compacted fresh list
    int32_t quad;             // pair == 1: expansions with at most 16 fresh neighbours score them FOUR lanes each (half the gather instructions)
    // visited set, tier 1: an open-addressing table of 16-bit entries in LDS (gs_body.h "two-tier visited set"); the global
    // table above is tier 2 and is only touched (and only then cleared) by a query whose tier 1 fills up.
    // GraphSearcher objects (session kernels, gs_body.h SES)
    int32_t session;          // 1: launch the session kernel (SES = true)
    float threshold;          // layer-0 admission `score >= threshold` (:437); > 0 also arms the TwoPhaseTracker
    int32_t *out_base;        // [Q] expandedCountBaseLayer, or nullptr
    // resume() on the device = the whole history of the searcher in ONE launch: phase 0 is the original search(), phase i > 0 the
    // i-th resume().  Between two phases the kernel does what reranking() + searchLayer0() do to the traversal state
    // (GraphSearcher.java:459-507): approximateResults is empty, evictedResults — ph_extra, the keys the host kept from that
    // call — goes back to the candidates, the counters restart.  Candidates and the visited set simply live on.  n_phases <= 1:
    // a plain search (ph_* unused, rerankK / threshold above apply).
    int32_t n_phases;
    int32_t ph_rerankK[GS_MAX_PHASES];
    float ph_threshold[GS_MAX_PHASES];
    const long long *ph_extra;      // keys of all transitions, query-major inside a phase
    const int32_t *ph_extra_off;    // (n_phases - 1) * ph_Q + 1 offsets into ph_extra: transition t of query q = [t * ph_Q + q, t * ph_Q + q + 1)
    int32_t ph_Q;                   // queries of the searcher (a launch may cover a part of them: Q / qmap above)
    int32_t prefetch;         // 1: touch the runner-up candidate's adjacency row + fused block while the popped one is scored (layer 0)
    int32_t v1_log2;          // log2(slots) of the LDS tier (slots / 4 buckets of four 16-bit entries), 0 = no LDS tier
    int32_t v1_idbits;        // node ids are < 1 << v1_idbits; v1_idbits - (v1_log2 - 2) <= 14 remainder bits + the choice bit
    // the workgroup form (gx_body.h, k_gsearch_wgx.hip): ONE query per workgroup — the query's ADC table (M x 256 f32) lives in LDS,
    // wave 0 runs the GraphSearcher loop and the other waves ("expanders") score whole adjacency rows it asks for ahead of time
    // UBR (gs_body.h "UBR", k_gsearch_ubr.hip; dot product / cosine, layer 0, M = 96): the 8-bit upper-bound table of every query is
    // PREBUILT by a dense kernel (ubr_table_kernel: [Q][M / 4][64] x 16 bytes, one scale per query) and held in the wave's
    // REGISTERS (M dwords per lane, looked up with ds_bpermute) — no LDS, the visited set's LDS tier and 8 waves per CU stay; the
    // neighbours the bound cannot drop are compacted through LDS and scored EIGHT lanes each; the candidate queue is trimmed to
    // what can still be popped.  Results, scores, visitedCount and expandedCount are the reference's.
    int32_t ubr;
    const uint32_t *ubr_tab;  // [Q][M][64] dwords: register k of lane s for query q at ((q * (M / 4) + k / 4) * 64 + s) * 4 + k % 4
    const float *ubr_meta;    // [Q][4]: {sum of the low edges + slack, scale, 1 = usable (every entry finite), unused}
    int32_t ubr_trim;         // candidates pushed between two trims of the queue (>= 1)
    unsigned long long *ubr_count;  // += neighbours dropped behind their bound (one atomic per query), or nullptr
    int32_t wgx;              // 1: launch the workgroup form
    int32_t wgx_slots;        // scored-row slots in LDS (<= 64)
    int32_t wgx_kps;          // keys per slot: 32 or 64 (>= every level's degree)
    int32_t wgx_depth;        // candidates below the popped one whose rows are requested ahead of time (0..3)
    int32_t wgx_lut_m;        // subspaces [0, wgx_lut_m) are scored from the LDS table, the rest table-free from the codebook (a multiple
                              // of 16, <= M): a shorter table lets two or three workgroups — control waves — share a CU
    int32_t wgx_log;          // addTopCandidate keys buffered in LDS and written to push_log when the query ends (0 = straight to push_log)
    // outputs
    int32_t *out_ids;         // [Q][rerankK] kept approximate results (unordered), -1 padded
    float *out_scores;        // [Q][rerankK] their approximate scores, -inf padded
    long long *out_stats;     // [Q][2] visitedCount, expandedCount
    int32_t *out_status;      // [Q] GS_OK / GS_OVERFLOW
    // the sequence of addTopCandidate calls at layer 0 (NodeQueue keys), kept so that rt_body.h can rebuild the reference's
    // result-heap ARRAY for queries whose rerank ties on the exact score at the K-th place; nullptr = not recorded
    long long *push_log;      // [Q][push_log_cap]
    int32_t *push_log_n;      // [Q] entries offered (> push_log_cap: the log overflowed)
    int push_log_cap;
    // round 6: NodeQueue.rerank's exact scores INSIDE the traversal wave (gs_body.h gs_rr_round): once a query's search has ended its
    // wave gathers the full-resolution rows of the kept results [0, rr_rows) — 64 per round, the rows transposed through the (now
    // idle) LDS block exactly as exact_gather_tr_kernel does, the same sequential chain per row — and out_scores receives the EXACT
    // similarity instead of the approximate one; rows [rr_rows, rerankK) keep the approximate score for the packed-remainder kernel
    // (launch_exact_gather_tail).  nullptr = the rerank is a kernel of its own.
    const float *rr_vecs;     // [rr_n][D] full-resolution vectors (16-byte aligned, D % 8 == 0)
    const float *rr_queries;  // [Q][D] the raw queries
    const float *rr_qnorm;    // [Q] sum of squares of a query (cosine), query_sqnorm_kernel
    const float *rr_vnorm;    // [rr_n] sum of squares of a row (cosine), launch_row_sqnorms
    long long rr_n;
    int32_t rr_rows;          // <= 64 * GS_RR_MAX_ROUNDS
    // round 6, last: DEFERRED exact scores above level 0 (gs_body.h DEFER; the register-table bound form over the row only).  A fresh
    // neighbour met at a level >= defer_min_level whose bound lies below that layer's best result (topK = 1) is not scored: (node, an
    // upper bound U of its score, exact < U) is remembered only through the largest such U.  It can never be popped above level 0 (the result minimum only
    // grows and the next layer's first pop is at least the last layer's best); at level 0 a pop is valid while its score >= the
    // largest U, and they are all forgotten once the pop threshold exceeds that U.  A pop that a deferred node
    // might outrank (0.6 % of the headline's queries) makes the worker START THE QUERY OVER without deferral — no rarely-run scoring
    // code in the expansion loop, whose registers are all taken.  Pops, results, visitedCount and expandedCount are the reference's.
    // nullptr = every fresh neighbour above level 0 is scored at once.
    int32_t defer;            // 1 = on
    int32_t defer_min_level;  // >= 1
    unsigned long long *defer_count;  // += {deferred, queries started over, unused} (one atomic each per query), or nullptr
    uint32_t *next_query;     // work counter (zeroed by the host before the launch)
    unsigned long long *prof; // developer aid (JVECTOR_HIP_GS_PROF=1): 8 phase counters, see gs_search_one; else nullptr
};

// the fused rerank's LDS tile: 64 rows x (64 + 4) floats (the row stride of exact_gather_tr_kernel: conflict-free 16-byte reads), laid
// over the head of the worker's block once the search has ended
constexpr int GS_RR_CH = 64, GS_RR_LS = GS_RR_CH + 4, GS_RR_MAX_ROUNDS = 4;
constexpr size_t gs_rr_lds_bytes() { return sizeof(float) * 64 * (size_t)GS_RR_LS; }

// the centred query at the head of a worker's LDS block, padded so that the 8-byte arrays behind it stay aligned for any D
constexpr size_t gs_q_bytes(int D) { return (sizeof(float) * (size_t)D + 15) & ~(size_t)15; }

// floats of the pair-lane exchange area: [M/2][32] entries (two lanes per neighbour, 32 neighbours per pass); above M = 96 only the
// compacted form exists and runs four lanes per neighbour, 16 per pass: [3 M/4][16]
constexpr size_t gs_xchg_floats(int pair_M) { return pair_M <= 96 ? (size_t)32 * (size_t)(pair_M / 2) : (size_t)16 * (size_t)(3 * pair_M / 4); }

// LDS bytes one worker needs
constexpr size_t gs_lds_bytes(int D, int rerankK, int cand_cap, int pair_M /* M when pair-lane scoring is on, else 0 */,
                           int evict_cap = GS_EVICT_CAP, int v1_log2 = 0)
{
    // (the 64-key sample buffer of the partition step shares the pair-lane exchange area when there is one)
    // (+ 8 with an exchange area: it starts at the next 16-byte boundary behind the 8-byte queues — its hand-over columns are read as
    // 16-byte words — whatever the parity of rerankK + cand_cap + evict_cap)
    const size_t base = gs_q_bytes(D) + sizeof(long long) * ((size_t)rerankK + (size_t)cand_cap + (size_t)evict_cap + (pair_M ? 0 : 64)) +
                        sizeof(float) * gs_xchg_floats(pair_M) + (pair_M ? 8 : 0);
    return v1_log2 > 0 ? ((base + 15) & ~(size_t)15) + ((size_t)2 << v1_log2) : base;
}

// UBR lives in the pair-lane exchange area: [7 M / 8 entries x 8 owners of f32][32 node ids][32 magnitudes][32 x M code bytes]
constexpr size_t gs_ubr_xchg_bytes(int M) { return sizeof(float) * 7 * (size_t)M + 256 + 32 * (size_t)M; }
static_assert(gs_ubr_xchg_bytes(96) <= sizeof(float) * gs_xchg_floats(96), "UBR's staging must fit the pair-lane exchange area");
// device bytes of one query's prebuilt bound table / all of them
constexpr size_t gs_ubr_tab_bytes(int M) { return (size_t)M * 64 * sizeof(uint32_t); }

// LDS bytes of the session kernels' TwoPhaseTracker state (500 recent scores + the 100 best)
constexpr size_t gs_session_lds_bytes() { return sizeof(float) * 500 + sizeof(int32_t) * 100; }

// The LDS tier's 16-bit entry = choice bit + remainder: idbits - log2(buckets) <= 14 (0xFFFF stays free for "empty").
inline bool gs_v1_fits(int v1_log2, int idbits) { return v1_log2 >= 4 && v1_log2 <= 15 && idbits - (v1_log2 - 2) <= 14 && idbits <= 31; }
inline int gs_idbits(long long n_nodes)
{
    int b = 1;
    while ((1ll << b) < n_nodes && b < 31) ++b;
    return b;
}

// ---- the workgroup form's LDS block: [the control wave's block = gs_lds_bytes(..., pair_M = 0, ...)] [header + ring + slot
//      tables] [slot keys] [the ADC table].  Offsets in bytes from the start of the header.
constexpr int GX_RING = 64;                 // request ring entries (>= slots: every request owns a slot)
constexpr int GX_MAX_SLOTS = 64;
enum : int32_t { GX_ITEM = 0, GX_REQ_HEAD = 1, GX_REQ_TAIL = 2, GX_QUIT = 3, GX_HDR_INTS = 8 };
enum : int32_t { GX_FREE = 0, GX_REQUESTED = 1, GX_READY = 2 };
constexpr size_t gx_off_ring() { return sizeof(int32_t) * GX_HDR_INTS; }
constexpr size_t gx_off_slot_node() { return gx_off_ring() + sizeof(int32_t) * GX_RING; }
constexpr size_t gx_off_slot_lvl() { return gx_off_slot_node() + sizeof(int32_t) * GX_MAX_SLOTS; }
constexpr size_t gx_off_slot_state() { return gx_off_slot_lvl() + sizeof(int32_t) * GX_MAX_SLOTS; }
constexpr size_t gx_off_keys() { return (gx_off_slot_state() + sizeof(int32_t) * GX_MAX_SLOTS + 15) & ~(size_t)15; }
constexpr size_t gx_off_log(int slots, int kps) { return (gx_off_keys() + sizeof(long long) * (size_t)slots * (size_t)kps + 15) & ~(size_t)15; }
constexpr size_t gx_off_lut(int slots, int kps, int logcap) { return (gx_off_log(slots, kps) + sizeof(long long) * (size_t)logcap + 15) & ~(size_t)15; }
constexpr size_t gx_shared_bytes(int slots, int kps, int logcap, int M) { return gx_off_lut(slots, kps, logcap) + sizeof(float) * 256 * (size_t)M; }
// where the header starts inside the workgroup's LDS block
constexpr size_t gx_ctl_bytes(int D, int rerankK, int cand_cap, int evict_cap, int v1_log2)
{
    return (gs_lds_bytes(D, rerankK, cand_cap, 0, evict_cap, v1_log2) + 15) & ~(size_t)15;
}
constexpr size_t gx_lds_bytes(int D, int rerankK, int cand_cap, int evict_cap, int v1_log2, int slots, int kps, int logcap, int M)
{
    return gx_ctl_bytes(D, rerankK, cand_cap, evict_cap, v1_log2) + gx_shared_bytes(slots, kps, logcap, M);
}

// rerank tie resolution (rt_body.h / rerank_tie_kernel)
struct RtParams {
    const float *cand_sc;       // [Q][R] exact scores of the kept approximate results (device order)
    const int32_t *cand_ids;    // [Q][R], -1 = empty slot
    int R;
    float *out_sc;              // [Q][K] the selection's result, rewritten for tied queries
    int32_t *out_ids;
    int K, Q;
    const long long *push_log;  // [Q][log_cap] NodeQueue keys in addTopCandidate order, nullptr = no log (mark only)
    const int32_t *push_log_n;  // [Q] entries offered (may exceed log_cap: overflow)
    int log_cap, rerankK;
    int32_t *status;            // [Q] GS_OK -> stays GS_OK (resolved here) or becomes GS_RERANK_TIE (host searcher)
    unsigned int *count;        // [2]: tied queries left for the host, tied queries resolved here
};

inline size_t rt_lds_bytes(int rerankK, int K) { return sizeof(long long) * ((size_t)rerankK + (size_t)K + 2); }

}  // namespace jv
