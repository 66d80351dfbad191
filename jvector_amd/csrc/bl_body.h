// bl_body.h — the array plumbing of batched graph construction (builder.cpp): what sits between the engine's scoring calls when
// a batch of nodes is inserted into a Vamana level.  Each function is the body of ONE work item (an inserted node, a back edge, a
// list element), written as plain C++ so that k_builder.hip runs it one item per GPU thread and the CPU mock
// (tests/mock/mock_kernels.cpp) runs the very same code in a loop.  No floating-point arithmetic lives here except the order
// comparison of scores (NodeQueue's total order, NodeQueue.java:125-129 / NumericUtils.java:49-65).
//
// Reference behaviour restated (host code there, batched here):
//   GraphIndexBuilder.updateNeighbors / ConcurrentNeighborMap.insertDiverse (B/graph/GraphIndexBuilder.java:644-659,
//   B/graph/ConcurrentNeighborMap.java:104-163): a new node's pruned candidate list becomes its neighbour list; every chosen
//   neighbour s gets the BACKLINK s -> node appended to its list (insertEdgeNotDiverse / backlink); a list that outgrows
//   maxDegree x neighborOverflow is re-pruned with retainDiverse over the merged list (existing + appended, by score).
#pragma once

#include <cstdint>

#ifndef BL_FN
#define BL_FN inline
#endif

namespace jv {

struct BlApplyParams {
    const int32_t *nodes;     // [B] inserted nodes (rows to write)
    const int32_t *cand;      // [B][C] candidate ids, best first, -1 padded
    const int32_t *sel;       // [B][Rf] selected candidate positions, ascending, -1 padded (retain_diverse output)
    int B, C, Rf, R;
    int32_t *nbrs;            // [N][R] adjacency, rows packed, -1 padded
    unsigned long long *edge_keys;  // [B*Rf] (target << 32 | edge index), or ~0 for "no edge"; nullptr = rows only
    int32_t *edge_src;        // [B*Rf]
};

// item = b * Rf + j  (one selected slot of one inserted node)
BL_FN void bl_apply_selection(const BlApplyParams &p, long long item)
{
    const int b = (int)(item / p.Rf), j = (int)(item % p.Rf);
    const int32_t v = p.nodes[b];
    const int32_t s = p.sel[(long long)b * p.Rf + j];
    int32_t chosen = s >= 0 && s < p.C ? p.cand[(long long)b * p.C + s] : -1;
    if (chosen == v) chosen = -1;  // never link a node to itself (re-insertion passes search a graph that already holds it)
    p.nbrs[(long long)v * p.R + j] = chosen;
    if (j == 0)
        for (int t = p.Rf; t < p.R; ++t) p.nbrs[(long long)v * p.R + t] = -1;
    if (p.edge_keys) {
        p.edge_keys[item] = chosen >= 0 ? (((unsigned long long)(uint32_t)chosen) << 32) | (unsigned long long)(uint32_t)item : ~0ull;
        p.edge_src[item] = v;
    }
}

// A row written by bl_apply_selection may have holes (a dropped self link): close them.  item = b.
BL_FN void bl_pack_row(const BlApplyParams &p, long long b)
{
    int32_t *row = p.nbrs + (long long)p.nodes[b] * p.R;
    int w = 0;
    for (int t = 0; t < p.Rf; ++t) {
        const int32_t x = row[t];
        if (x >= 0) row[w++] = x;
    }
    for (; w < p.Rf; ++w) row[w] = -1;
}

// ---- improveConnections (GraphIndexBuilder.java:510-560: search the FINISHED graph for a node and merge what the search found
//      with the neighbours the node already has — ConcurrentNeighborMap.insertDiverse, :104-163 — then prune): the merged list ----
struct BlImproveParams {
    const int32_t *nodes;   // [B] nodes being improved (each has a row)
    const int32_t *cand;    // [B][C] search results, best first, -1 padded
    int B, C, R;
    const int32_t *nbrs;    // [N][R]
    int32_t *list;          // [B][R + C]: the node's current row, then the candidates it does not hold yet (never the node itself), -1 padded
};

// item = b
BL_FN void bl_improve_list(const BlImproveParams &p, long long b)
{
    const int32_t v = p.nodes[b];
    const int32_t *row = p.nbrs + (long long)v * p.R;
    int32_t *out = p.list + b * (long long)(p.R + p.C);
    int w = 0;
    for (int t = 0; t < p.R && row[t] >= 0; ++t) out[w++] = row[t];
    const int have = w;
    for (int c = 0; c < p.C; ++c) {
        const int32_t x = p.cand[b * (long long)p.C + c];
        if (x < 0 || x == v) continue;
        bool dup = false;
        for (int t = 0; t < have && !dup; ++t) dup = out[t] == x;
        if (!dup) out[w++] = x;
    }
    for (; w < p.R + p.C; ++w) out[w] = -1;
}

// back edges of rows that were just rewritten: item = b * Rf + j -> edge (row[j] <- nodes[b]); the merge step drops an edge its target
// already holds
struct BlRowEdgesParams {
    const int32_t *nodes;   // [B]
    int B, Rf, R;
    const int32_t *nbrs;    // [N][R]
    unsigned long long *edge_keys;   // [B * Rf]
    int32_t *edge_src;               // [B * Rf]
};

BL_FN void bl_row_edges(const BlRowEdgesParams &p, long long item)
{
    const int b = (int)(item / p.Rf), j = (int)(item % p.Rf);
    const int32_t v = p.nodes[b];
    const int32_t u = p.nbrs[(long long)v * p.R + j];
    p.edge_keys[item] = u >= 0 ? (((unsigned long long)(uint32_t)u) << 32) | (unsigned long long)(uint32_t)item : ~0ull;
    p.edge_src[item] = v;
}

struct BlMergeParams {
    const unsigned long long *keys;  // [E] sorted ascending: edges of one target are consecutive, in edge-index order
    const int32_t *src;              // [E] sorted along
    long long E;
    int R, Knew;                     // row width; at most Knew appended ids are kept for a re-prune
    int32_t *nbrs;                   // [N][R]
    int32_t *over_tgt;               // [cap] targets whose list overflowed
    int32_t *over_list;              // [cap][R + Knew] merged list: existing row, then the appended ids, -1 padded
    unsigned int *over_count;        // atomic slot counter
    unsigned int over_cap;
};

#ifndef BL_ATOMIC_INC
#define BL_ATOMIC_INC(p) ((*(p))++)
#endif

// item = index into the sorted edge array; only the first edge of each target's run does the work
BL_FN void bl_backlink_merge(const BlMergeParams &p, long long i)
{
    const unsigned long long k = p.keys[i];
    if (k == ~0ull) return;
    const int32_t tgt = (int32_t)(k >> 32);
    if (i > 0 && (int32_t)(p.keys[i - 1] >> 32) == tgt) return;
    int32_t *row = p.nbrs + (long long)tgt * p.R;
    int deg = 0;
    while (deg < p.R && row[deg] >= 0) ++deg;
    long long e = i;
    // append while the row has room (duplicates of an existing neighbour are dropped: s may already point at the new node when a
    // re-insertion pass runs)
    for (; e < p.E && (int32_t)(p.keys[e] >> 32) == tgt; ++e) {
        const int32_t s = p.src[e];
        bool dup = false;
        for (int t = 0; t < deg; ++t) dup = dup || row[t] == s;
        if (dup) continue;
        if (deg == p.R) break;
        row[deg++] = s;
    }
    if (e >= p.E || (int32_t)(p.keys[e] >> 32) != tgt) return;  // everything fitted
    // overflow: hand the merged list to the re-prune
    const unsigned int slot = BL_ATOMIC_INC(p.over_count);
    if (slot >= p.over_cap) return;  // cannot happen (cap = number of targets); the row simply keeps what fitted
    const int L = p.R + p.Knew;
    int32_t *lst = p.over_list + (long long)slot * L;
    p.over_tgt[slot] = tgt;
    for (int t = 0; t < p.R; ++t) lst[t] = row[t];
    int n = p.R;
    for (; e < p.E && (int32_t)(p.keys[e] >> 32) == tgt && n < L; ++e) {
        const int32_t s = p.src[e];
        bool dup = false;
        for (int t = 0; t < n; ++t) dup = dup || lst[t] == s;
        if (!dup) lst[n++] = s;
    }
    for (; n < L; ++n) lst[n] = -1;
}

// NodeQueue's order on scores (NumericUtils.floatToSortableInt), as a sortable 32-bit key
BL_FN int32_t bl_sortable(float f)
{
    int32_t b;
    __builtin_memcpy(&b, &f, 4);
    if (f != f) b = 0x7fc00000;
    return b ^ ((b >> 31) & 0x7fffffff);
}

struct BlSortParams {
    const int32_t *ids;     // [P][L]
    const float *scores;    // [P][L]  (-inf for the -1 padding)
    int P, L;
    int32_t *out_ids;       // [P][L] best first; equal scores keep their input order; -1 ids last
    float *out_scores;
    int32_t *out_count;     // [P] ids >= 0
};

// item = p * L + i: the element finds its rank among the L of its row (NodeArray order: score descending, stable)
BL_FN void bl_rank_sort(const BlSortParams &p, long long item)
{
    const long long row = item / p.L;
    const int i = (int)(item % p.L);
    const int32_t *ids = p.ids + row * p.L;
    const float *sc = p.scores + row * p.L;
    const int32_t my_id = ids[i];
    const int32_t mine = bl_sortable(sc[i]);
    int rank = 0, valid = 0;
    for (int j = 0; j < p.L; ++j) {
        const int32_t oid = ids[j];
        valid += oid >= 0 ? 1 : 0;
        if (j == i) continue;
        const int32_t other = bl_sortable(sc[j]);
        bool before;
        if ((oid >= 0) != (my_id >= 0)) before = oid >= 0;            // padding goes last
        else before = other > mine || (other == mine && j < i);
        rank += before ? 1 : 0;
    }
    p.out_ids[row * p.L + rank] = my_id;
    p.out_scores[row * p.L + rank] = sc[i];
    if (i == 0) p.out_count[row] = valid;
}

struct BlRowsParams {
    const int32_t *tgt;     // [P] rows to rewrite
    const int32_t *lst;     // [P][L] sorted merged lists
    const int32_t *sel;     // [P][Rf] selected positions
    int P, L, Rf, R;
    int32_t *nbrs;          // [N][R]
};

// item = p: the re-pruned list replaces the row
BL_FN void bl_rewrite_row(const BlRowsParams &p, long long r)
{
    int32_t *row = p.nbrs + (long long)p.tgt[r] * p.R;
    int w = 0;
    for (int j = 0; j < p.Rf; ++j) {
        const int32_t s = p.sel[r * p.Rf + j];
        if (s >= 0 && s < p.L) {
            const int32_t x = p.lst[r * p.L + s];
            if (x >= 0) row[w++] = x;
        }
    }
    for (; w < p.R; ++w) row[w] = -1;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// REFERENCE ORDER (builder option bl_ref_order): the lists are kept the way ConcurrentNeighborMap.Neighbors keeps them — every entry
// with the score it was inserted under, in NodeArray order (score descending, a new entry BEHIND its equals), and with the
// diverseBefore mark — so that a batch of ONE node performs addGraphNode's list operations exactly (CPU checker: jvo_builder_*,
// "GraphIndexBuilder, one thread"; tests/test_builder_reference_order*.py compare the adjacency byte for byte):
//   insertDiverse on the new node's empty list        ConcurrentNeighborMap.java:222-243   bl_ro_apply_selection
//   backlink -> Neighbors.insert                      :139-146, :262-296                    bl_ro_backlink_merge
//       insertionPoint / duplicateExistsNear          NodeArray.java:166-171, :212-228, :308-318
//       diverseBefore = min(insertionPoint, diverseBefore); size > (int) (overflow x maxDegree) -> retainDiverse(diverseBefore)
//   the re-pruned list replaces the row, diverseBefore = size                                  bl_ro_rewrite_row
//   enforceDegree                                     :190-200                              bl_ro_copy_row + the same prune
// A batch of B > 1 nodes is the concurrent case: the batch's nodes do not see each other's edges during their searches, a target's
// back edges are applied in batch order, and a list that outgrows the limit inside one batch is pruned ONCE, after the batch's last
// edge to it went in (the reference would prune at every crossing).
struct BlRoApplyParams {
    const int32_t *nodes;     // [B]
    const int32_t *cand;      // [B][C] best first
    const float *cand_sc;     // [B][C]
    const int32_t *sel;       // [B][Rf] selected positions, ascending, -1 padded
    int B, C, Rf, R;
    int32_t *nbrs;            // [N][R]
    float *nsc;               // [N][R]
    int32_t *db;              // [N] diverseBefore
    unsigned long long *edge_keys;   // [B * Rf] (target << 32 | item), ~0 = none
    int32_t *edge_src;               // [B * Rf]
    float *edge_sc;                  // [B * Rf] the score the target is listed under = the score the back edge is inserted with (:143-144)
};

// item = b
BL_FN void bl_ro_apply_selection(const BlRoApplyParams &p, long long b)
{
    const int32_t v = p.nodes[b];
    int32_t *row = p.nbrs + (long long)v * p.R;
    float *rsc = p.nsc + (long long)v * p.R;
    int w = 0;
    for (int j = 0; j < p.Rf; ++j) {
        const long long item = b * p.Rf + j;
        const int32_t s = p.sel[item];
        int32_t chosen = s >= 0 && s < p.C ? p.cand[b * (long long)p.C + s] : -1;
        if (chosen == v) chosen = -1;
        if (chosen >= 0) {
            const float x = p.cand_sc[b * (long long)p.C + s];
            row[w] = chosen;
            rsc[w] = x;
            ++w;
            p.edge_keys[item] = (((unsigned long long)(uint32_t)chosen) << 32) | (unsigned long long)(uint32_t)item;
            p.edge_sc[item] = x;
        } else {
            p.edge_keys[item] = ~0ull;
            p.edge_sc[item] = 0.0f;
        }
        p.edge_src[item] = v;
    }
    p.db[v] = w;   // new Neighbors(...): diverseBefore = size()
    for (; w < p.R; ++w) {
        row[w] = -1;
        rsc[w] = 0.0f;
    }
}

// SORTED LISTS (builder option bl_sorted_lists): the same list mechanics with the scores the classic path would compute — the PQ
// diversity function of (node, member), symmetric, so a back edge carries the score its forward edge has — stored instead of
// recomputed: what a re-prune of the classic path does first (score the whole merged list against its node, sort it) is already
// there.  The selection of a new node's row, in candidate order: item = b * Rf + j -> out_ids (-1: no entry / the node itself).
struct BlSelIdsParams {
    const int32_t *nodes;     // [B]
    const int32_t *cand;      // [B][C]
    const int32_t *sel;       // [B][Rf]
    int B, C, Rf;
    int32_t *out_ids;         // [B][Rf]
};

BL_FN void bl_sel_ids(const BlSelIdsParams &p, long long item)
{
    const long long b = item / p.Rf;
    const int32_t s = p.sel[item];
    int32_t chosen = s >= 0 && s < p.C ? p.cand[b * (long long)p.C + s] : -1;
    if (chosen == p.nodes[b]) chosen = -1;
    p.out_ids[item] = chosen;
}

// rows from lists that are already in NodeArray order under the scores to store ([B][Rf], -1 last); edges as bl_ro_apply_selection
struct BlRoApplySortedParams {
    const int32_t *nodes;     // [B]
    const int32_t *ids;       // [B][Rf]
    const float *sc;          // [B][Rf]
    int B, Rf, R;
    int32_t *nbrs;
    float *nsc;
    int32_t *db;
    unsigned long long *edge_keys;
    int32_t *edge_src;
    float *edge_sc;
};

// item = b
BL_FN void bl_ro_apply_sorted(const BlRoApplySortedParams &p, long long b)
{
    const int32_t v = p.nodes[b];
    int32_t *row = p.nbrs + (long long)v * p.R;
    float *rsc = p.nsc + (long long)v * p.R;
    int w = 0;
    for (int j = 0; j < p.Rf; ++j) {
        const long long item = b * p.Rf + j;
        const int32_t chosen = p.ids[item];
        if (chosen >= 0) {
            row[w] = chosen;
            rsc[w] = p.sc[item];
            ++w;
            p.edge_keys[item] = (((unsigned long long)(uint32_t)chosen) << 32) | (unsigned long long)(uint32_t)item;
            p.edge_sc[item] = p.sc[item];
        } else {
            p.edge_keys[item] = ~0ull;
            p.edge_sc[item] = 0.0f;
        }
        p.edge_src[item] = v;
    }
    p.db[v] = w;
    for (; w < p.R; ++w) {
        row[w] = -1;
        rsc[w] = 0.0f;
    }
}

constexpr int BL_RO_MAX_LIST = 192;   // R <= 64 and at most 2 R appended entries are kept

struct BlRoMergeParams {
    const unsigned long long *keys;  // [E] sorted ascending (target, item)
    const int32_t *src;              // [E] UNSORTED: indexed by the key's item
    const float *esc;                // [E] UNSORTED
    long long E;
    int R, hard_max, Knew, dedupe_ids;   // hard_max = (int) (overflow x maxDegree), <= R
    int32_t *nbrs;
    float *nsc;
    int32_t *db;
    int32_t *over_tgt;               // [cap]
    int32_t *over_list;              // [cap][R + Knew]
    float *over_sc;                  // [cap][R + Knew]
    int32_t *over_db;                // [cap]
    int32_t *over_n;                 // [cap]
    unsigned int *over_count;
    unsigned int over_cap;
};

// NodeArray.insertionPoint: -1 = (node, score) already listed
BL_FN int bl_ro_insertion_point(const int32_t *ids, const float *sc, int n, int32_t node, float x)
{
    int at = 0;
    while (at < n && !(sc[at] < x)) ++at;   // descSortFindRightMostInsertionPoint on a descending array
    for (int i = at - 1; i >= 0 && sc[i] == x; --i)
        if (ids[i] == node) return -1;
    for (int i = at; i < n && sc[i] == x; ++i)
        if (ids[i] == node) return -1;
    return at;
}

// item = index into the sorted keys; the first edge of a target's run applies the whole run
BL_FN void bl_ro_backlink_merge(const BlRoMergeParams &p, long long i)
{
    const unsigned long long k = p.keys[i];
    if (k == ~0ull) return;
    const int32_t tgt = (int32_t)(k >> 32);
    if (i > 0 && (int32_t)(p.keys[i - 1] >> 32) == tgt) return;
    int32_t ids[BL_RO_MAX_LIST];
    float sc[BL_RO_MAX_LIST];
    int32_t *row = p.nbrs + (long long)tgt * p.R;
    float *rsc = p.nsc + (long long)tgt * p.R;
    int n = 0;
    while (n < p.R && row[n] >= 0) {
        ids[n] = row[n];
        sc[n] = rsc[n];
        ++n;
    }
    const int L = p.R + p.Knew;
    int dbv = p.db[tgt];
    bool changed = false;
    for (long long e = i; e < p.E && (int32_t)(p.keys[e] >> 32) == tgt; ++e) {
        const long long item = (long long)(uint32_t)p.keys[e];
        const int32_t s = p.src[item];
        const float x = p.esc[item];
        const int at = bl_ro_insertion_point(ids, sc, n, s, x);
        if (at < 0) continue;                                  // "new" node already existed (:275-278)
        if (p.dedupe_ids) {
            bool have = false;
            for (int t = 0; t < n; ++t) have = have || ids[t] == s;
            if (have) continue;
        }
        if (n == L) continue;                                  // (a batch larger than the working list: the edge is dropped)
        for (int t = n; t > at; --t) {
            ids[t] = ids[t - 1];
            sc[t] = sc[t - 1];
        }
        ids[at] = s;
        sc[at] = x;
        ++n;
        dbv = at < dbv ? at : dbv;
        changed = true;
    }
    if (!changed) return;
    if (n <= p.hard_max) {
        for (int t = 0; t < n; ++t) {
            row[t] = ids[t];
            rsc[t] = sc[t];
        }
        p.db[tgt] = dbv;
        return;
    }
    const unsigned int slot = BL_ATOMIC_INC(p.over_count);
    if (slot >= p.over_cap) return;   // cannot happen (cap = number of targets)
    int32_t *lst = p.over_list + (long long)slot * L;
    float *lsc = p.over_sc + (long long)slot * L;
    for (int t = 0; t < L; ++t) {
        lst[t] = t < n ? ids[t] : -1;
        lsc[t] = t < n ? sc[t] : 0.0f;
    }
    p.over_tgt[slot] = tgt;
    p.over_db[slot] = dbv;
    p.over_n[slot] = n;
}

struct BlRoRowsParams {
    const int32_t *tgt;     // [P]
    const int32_t *lst;     // [P][L]
    const float *lsc;       // [P][L]
    const int32_t *sel;     // [P][Rf]
    int P, L, Rf, R;
    int32_t *nbrs;
    float *nsc;
    int32_t *db;
};

// item = r: retain(selected) + diverseBefore = size
BL_FN void bl_ro_rewrite_row(const BlRoRowsParams &p, long long r)
{
    const int32_t v = p.tgt[r];
    int32_t *row = p.nbrs + (long long)v * p.R;
    float *rsc = p.nsc + (long long)v * p.R;
    int w = 0;
    for (int j = 0; j < p.Rf; ++j) {
        const int32_t s = p.sel[r * p.Rf + j];
        if (s >= 0 && s < p.L) {
            const int32_t x = p.lst[r * (long long)p.L + s];
            if (x >= 0) {
                row[w] = x;
                rsc[w] = p.lsc[r * (long long)p.L + s];
                ++w;
            }
        }
    }
    p.db[v] = w;
    for (; w < p.R; ++w) {
        row[w] = -1;
        rsc[w] = 0.0f;
    }
}

// improveConnections in reference order (GraphIndexBuilder.java:510-545 -> addEdges -> Neighbors.insertDiverse :222-243): the node's
// list and the search's candidates merged the way NodeArray.merge walks them (:63-143: the better score first; at EQUAL scores one
// entry of the list, then one of the candidates), except that a node the merged list already holds is dropped whatever its score —
// the reference drops it only at an equal score and so can list a node twice (the CPU checker restates both: jvo_builder_set_deviations).
struct BlRoImproveParams {
    const int32_t *nodes;     // [B]
    const int32_t *cand;      // [B][C] best first, -1 padded
    const float *cand_sc;     // [B][C]
    int B, C, R;
    int skip_empty;           // reference order: "if (graph.getNeighborsIterator(lvl, node).size() > 0)" (:527) — a node without neighbours is left alone
    const int32_t *nbrs;      // [N][R]
    const float *nsc;         // [N][R]
    int32_t *list;            // [B][R + C]
    float *lsc;               // [B][R + C]
    int32_t *ln;              // [B] merged entries
};

BL_FN void bl_ro_take(int32_t *out, float *osc, int &w, int32_t x, float s, int32_t self)
{
    if (x < 0 || x == self) return;
    for (int t = 0; t < w; ++t)
        if (out[t] == x) return;
    out[w] = x;
    osc[w] = s;
    ++w;
}

// item = b
BL_FN void bl_ro_improve_list(const BlRoImproveParams &p, long long b)
{
    const int32_t v = p.nodes[b];
    const int32_t *a1 = p.nbrs + (long long)v * p.R;
    const float *s1 = p.nsc + (long long)v * p.R;
    const int32_t *a2 = p.cand + b * (long long)p.C;
    const float *s2 = p.cand_sc + b * (long long)p.C;
    int n1 = 0, n2 = 0;
    while (n1 < p.R && a1[n1] >= 0) ++n1;
    while (n2 < p.C && a2[n2] >= 0) ++n2;
    if (n1 == 0 && p.skip_empty) n2 = 0;
    const int L = p.R + p.C;
    int32_t *out = p.list + b * (long long)L;
    float *osc = p.lsc + b * (long long)L;
    int w = 0, i = 0, j = 0;
    while (i < n1 && j < n2) {
        if (s1[i] < s2[j]) {
            bl_ro_take(out, osc, w, a2[j], s2[j], v);
            ++j;
        } else if (s1[i] > s2[j]) {
            bl_ro_take(out, osc, w, a1[i], s1[i], v);
            ++i;
        } else {
            bl_ro_take(out, osc, w, a1[i], s1[i], v);
            bl_ro_take(out, osc, w, a2[j], s2[j], v);
            ++i;
            ++j;
        }
    }
    for (; i < n1; ++i) bl_ro_take(out, osc, w, a1[i], s1[i], v);
    for (; j < n2; ++j) bl_ro_take(out, osc, w, a2[j], s2[j], v);
    p.ln[b] = w;
    for (; w < L; ++w) {
        out[w] = -1;
        osc[w] = 0.0f;
    }
}

// back edges of rows that were just rewritten, with the score each member is listed under: item = b * Rf + j
struct BlRoRowEdgesParams {
    const int32_t *nodes;     // [B]
    int B, Rf, R;
    const int32_t *nbrs;
    const float *nsc;
    unsigned long long *edge_keys;
    int32_t *edge_src;
    float *edge_sc;
};

BL_FN void bl_ro_row_edges(const BlRoRowEdgesParams &p, long long item)
{
    const int b = (int)(item / p.Rf), j = (int)(item % p.Rf);
    const int32_t v = p.nodes[b];
    const int32_t u = p.nbrs[(long long)v * p.R + j];
    p.edge_keys[item] = u >= 0 ? (((unsigned long long)(uint32_t)u) << 32) | (unsigned long long)(uint32_t)item : ~0ull;
    p.edge_src[item] = v;
    p.edge_sc[item] = u >= 0 ? p.nsc[(long long)v * p.R + j] : 0.0f;
}

struct BlRoCopyParams {
    const int32_t *tgt;     // [P]
    int P, R;
    const int32_t *nbrs;
    const float *nsc;
    const int32_t *db;
    int32_t *lst;           // [P][R]
    float *lsc;             // [P][R]
    int32_t *ldb;           // [P]
    int32_t *ln;            // [P]
};

// item = r: a row with its scores and mark, for enforceDegree's prune
BL_FN void bl_ro_copy_row(const BlRoCopyParams &p, long long r)
{
    const int32_t v = p.tgt[r];
    int n = 0;
    for (int t = 0; t < p.R; ++t) {
        const int32_t x = p.nbrs[(long long)v * p.R + t];
        p.lst[r * (long long)p.R + t] = x;
        p.lsc[r * (long long)p.R + t] = p.nsc[(long long)v * p.R + t];
        n += x >= 0 ? 1 : 0;
    }
    p.ldb[r] = p.db[v];
    p.ln[r] = n;
}

struct BlOverParams {
    const int32_t *nbrs;    // [N][R]
    long long N;
    int R, Rf;
    int32_t *over_tgt;      // [cap]
    unsigned int *over_count;
    unsigned int over_cap;
};

// final pass (GraphIndexBuilder.cleanup -> enforceDegree): item = node; rows longer than Rf are listed for a re-prune
BL_FN void bl_list_over_degree(const BlOverParams &p, long long v)
{
    const int32_t *row = p.nbrs + v * p.R;
    if (p.R > p.Rf && row[p.Rf] >= 0) {
        const unsigned int slot = BL_ATOMIC_INC(p.over_count);
        if (slot < p.over_cap) p.over_tgt[slot] = (int32_t)v;
    }
}

// candidates per row = ids >= 0 (the search pads with -1); item = b
BL_FN void bl_count_valid(const int32_t *cand, int C, int32_t *count, long long b)
{
    int n = 0;
    for (int j = 0; j < C; ++j) n += cand[b * C + j] >= 0 ? 1 : 0;
    count[b] = n;
}

}  // namespace jv
