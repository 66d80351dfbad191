// (the test harness keeps the measured-and-switched-off kernel variants compiled: they stay under test)
#define JV_EXPERIMENTAL 1
// gs_emu.cpp — compiles the device-resident graph search body (jvector_amd/csrc/gs_body.h) for the lane emulator and
// exposes one C entry point to the CPU tests.  TEST HARNESS: g++ -O2 -ffp-contract=off, never linked into the product.
#include <vector>

#include "hip_emu.h"

#define GS_FN inline
#define GS_SCHED_FENCE() ((void)0)
#define GS_NOINLINE static
#define GS_LDS_AS
#define GS_GLOBAL_AS
static inline int gs_lane() { return emu::lane(); }
// gs_body.h's sync point is wave-scope here: in a one-wave block (every form but WGX) that IS the block barrier, and in the
// workgroup form (gx_body.h) the control wave must not wait for the expander waves
static inline void gs_barrier() { emu::wave_barrier(); }
static inline int gs_tid() { return emu::lane(); }
static inline int gs_block_threads() { return emu::current()->nl; }
static inline void gs_block_barrier() { emu::barrier(); }
// LDS flags between waves: plain accesses (one host thread runs all lanes); a spin-wait must let the other lanes run
static inline int32_t gs_lds_load(const int32_t *p) { return *(const volatile int32_t *)p; }
static inline void gs_lds_store(int32_t *p, int32_t v) { *(volatile int32_t *)p = v; }
static inline int32_t gs_lds_add(int32_t *p, int32_t v)
{
    const int32_t old = *p;
    *p = old + v;
    return old;
}
static inline void gs_spin_pause() { emu::switch_to_next_live(); }
static inline uint64_t gs_ballot(bool p) { return emu::ballot(p); }
static inline long long gs_shfl(long long v, int src) { return emu::shfl(v, src); }
static inline uint32_t gs_bcast32(uint32_t v, int src) { return (uint32_t)emu::shfl((long long)v, src); }
static inline long long gs_shfl_xor(long long v, int m) { return emu::shfl(v, emu::lane() ^ m); }
static inline int32_t gs_shfl32(int32_t v, int src) { return (int32_t)emu::shfl((long long)v, src); }
static inline uint32_t gs_perm(uint32_t hi, uint32_t lo, uint32_t sel)   // v_perm_b32 (selectors 0..7 and 0x0c only)
{
    const uint64_t pool = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t sb = (sel >> (8 * i)) & 0xFFu;
        const uint32_t byte = sb < 8 ? (uint32_t)((pool >> (8 * sb)) & 0xFFu) : (sb == 0x0c ? 0u : 0xFFu);
        r |= byte << (8 * i);
    }
    return r;
}
#define GS_OPAQUE_I32(x) ((void)0)
static inline int32_t gs_cas(int32_t *p, int32_t expect, int32_t desired)
{
    const int32_t old = *p;
    if (old == expect) *p = desired;
    return old;
}
static inline uint32_t gs_lds_cas(uint32_t *p, uint32_t expect, uint32_t desired)
{
    const uint32_t old = *p;
    if (old == expect) *p = desired;
    return old;
}
static inline void gs_prefetch_lds(const void *g, void *lds)
{
    // the emulated lane really performs the touch: an address outside the arrays it names would fault here, and the landing
    // bytes are scribbled so that any read of them shows up as a wrong result
    ((volatile uint32_t *)lds)[emu::lane() & 63] = *(const volatile uint32_t *)g ^ 0xA5A5A5A5u;
}
static inline uint32_t gs_fetch_add(uint32_t *p, uint32_t v)
{
    const uint32_t old = *p;
    *p = old + v;
    return old;
}
static inline void gs_fetch_add64(unsigned long long *p, unsigned long long v) { *p += v; }
static inline void gs_fence() {}
static inline double gs_sqrt(double x) { return std::sqrt(x); }
static inline float gs_rsq_approx(float x) { return 1.0f / std::sqrt(x); }
static inline void gs_gather64(float v, float (&out)[64]) { emu::gather64(v, out); }

#include "../../jvector_amd/csrc/gs_body.h"
#include "../../jvector_amd/csrc/gx_body.h"
#include "../../jvector_amd/csrc/gs_host.h"

namespace {
struct Launch {
    const jv::GsParams *p;
    int vsf, ch, worker;
    char *lds;
};

template <int VSF, bool PAIR>
void run_ch(const Launch &L)
{
    switch (L.ch) {
    case 1: jv::gs_worker<VSF, 1, PAIR>(*L.p, L.worker, L.lds); break;
    case 2: jv::gs_worker<VSF, 2, PAIR>(*L.p, L.worker, L.lds); break;
    case 3: jv::gs_worker<VSF, 3, PAIR>(*L.p, L.worker, L.lds); break;
    case 4: jv::gs_worker<VSF, 4, PAIR>(*L.p, L.worker, L.lds); break;
    case 6: jv::gs_worker<VSF, 6, PAIR>(*L.p, L.worker, L.lds); break;
    case 8: jv::gs_worker<VSF, 8, PAIR>(*L.p, L.worker, L.lds); break;
    case 12: jv::gs_worker<VSF, 12, PAIR>(*L.p, L.worker, L.lds); break;
    default: abort();
    }
}
template <bool PAIR>
void run_vsf(const Launch &L)
{
    if (L.vsf == 0) run_ch<0, PAIR>(L);
    else if (L.vsf == 1) run_ch<1, PAIR>(L);
    else run_ch<2, PAIR>(L);
}
template <int VSF>
void run_ubr(const Launch &L)
{
    switch (L.ch) {
    case 4: jv::gs_worker<VSF, 4, true, false, false, false, true>(*L.p, L.worker, L.lds); break;
    case 6: jv::gs_worker<VSF, 6, true, false, false, false, true>(*L.p, L.worker, L.lds); break;
    default: abort();
    }
}
template <int VSF>
void run_ubrc(const Launch &L)   // the bound form over the compacted fresh list (rows up to 64 wide, codes by ordinal)
{
    switch (L.ch) {
    case 4: jv::gs_worker<VSF, 4, false, false, false, true, true>(*L.p, L.worker, L.lds); break;
    case 6: jv::gs_worker<VSF, 6, false, false, false, true, true>(*L.p, L.worker, L.lds); break;
    default: abort();
    }
}
template <int VSF>
void run_pairc(const Launch &L)
{
    switch (L.ch) {
    case 1: jv::gs_worker<VSF, 1, false, false, false, true>(*L.p, L.worker, L.lds); break;
    case 2: jv::gs_worker<VSF, 2, false, false, false, true>(*L.p, L.worker, L.lds); break;
    case 3: jv::gs_worker<VSF, 3, false, false, false, true>(*L.p, L.worker, L.lds); break;
    case 4: jv::gs_worker<VSF, 4, false, false, false, true>(*L.p, L.worker, L.lds); break;
    case 6: jv::gs_worker<VSF, 6, false, false, false, true>(*L.p, L.worker, L.lds); break;
    case 8: jv::gs_worker<VSF, 8, false, false, false, true>(*L.p, L.worker, L.lds); break;
    case 12: jv::gs_worker<VSF, 12, false, false, false, true>(*L.p, L.worker, L.lds); break;
    default: abort();
    }
}
template <int VSF>
void run_wgx(const Launch &L)
{
    switch (L.ch) {
    case 1: jv::gx_worker<VSF, 1>(*L.p, L.worker, L.lds); break;
    case 2: jv::gx_worker<VSF, 2>(*L.p, L.worker, L.lds); break;
    case 3: jv::gx_worker<VSF, 3>(*L.p, L.worker, L.lds); break;
    case 4: jv::gx_worker<VSF, 4>(*L.p, L.worker, L.lds); break;
    case 6: jv::gx_worker<VSF, 6>(*L.p, L.worker, L.lds); break;
    case 8: jv::gx_worker<VSF, 8>(*L.p, L.worker, L.lds); break;
    case 12: jv::gx_worker<VSF, 12>(*L.p, L.worker, L.lds); break;
    default: abort();
    }
}
void lane_main(void *arg)
{
    const Launch &L = *(const Launch *)arg;
    if (L.p->wgx) {
        if (L.vsf == 0) run_wgx<0>(L);
        else if (L.vsf == 1) run_wgx<1>(L);
        else run_wgx<2>(L);
    } else if (L.p->ubr && L.p->pair == 2) {
        if (L.vsf == 0) run_ubrc<0>(L);
        else if (L.vsf == 1) run_ubrc<1>(L);
        else run_ubrc<2>(L);
    } else if (L.p->ubr) {
        if (L.vsf == 0) run_ubr<0>(L);
        else if (L.vsf == 1) run_ubr<1>(L);
        else run_ubr<2>(L);
    } else if (L.p->pair == 2) {
        if (L.vsf == 0) run_pairc<0>(L);
        else if (L.vsf == 1) run_pairc<1>(L);
        else run_pairc<2>(L);
    } else if (L.p->pair) run_vsf<true>(L);
    else run_vsf<false>(L);
}
}  // namespace

static unsigned long long g_last_defer[3] = {0, 0, 0};   // DEFER counters of the last gs_emu_search: deferred, queries started over, unused
extern "C" void gs_emu_last_defer(unsigned long long *out) { for (int i = 0; i < 3; ++i) out[i] = g_last_defer[i]; }

extern "C" long gs_emu_search(int n_levels, const int32_t *const *lv_nodes, const int32_t *const *lv_nbrs, const int32_t *lv_count,
                              const int32_t *lv_degree, int entry_node, int entry_level, const float *codebooks, const float *cq,
                              const float *bmag, const uint8_t *codes, const float *code_norms, const uint8_t *blocks,
                              const float *fused_norms, int D, int M, int deg0, int Q, int rerankK, int vsf, int vcap_log2,
                              int spill_cap, int cand_cap, int workers, int pair_mode /* 0 off, 1 when degrees allow */, int32_t *out_ids, float *out_scores, long long *out_stats,
                              int32_t *out_status, int v1_log2 /* LDS tier of the visited set: log2(slots), 0 = none */, int v1_idbits,
                              int evict_cap /* 0 = GS_EVICT_CAP */, int lutr /* must be 0 (the register-resident ADC table form left the source in round 6) */,
                              int wgx_waves /* > 0: the workgroup form (gx_body.h) with this many waves (2..4 here), M <= 128 */,
                              int wgx_slots, int wgx_depth, int wgx_lut_m /* 0 = M */,
                              int ub8 /* 1: the pair-lane kernel with the 8-bit upper-bound table (dot / cosine, M <= 96, degrees <= 32);
                                         2: UBR — the table prebuilt (gs_ubr_build_ref) and held in registers, survivors compacted, eight lanes each (M = 64 / 96) */,
                              long long *ub8_dropped_out /* nullable */)
{
    if (lutr) return -4;
    if (wgx_waves && (M == 80 || M == 112 || wgx_waves < 2 || wgx_waves > emu::MAX_WAVES)) return -5;
    if (n_levels < 1 || n_levels > jv::GS_MAX_LEVELS || M % 16 != 0 || D != 8 * M || cand_cap < 128) return -1;
    if (v1_log2 > 0 && !jv::gs_v1_fits(v1_log2, v1_idbits)) return -2;
    jv::GsParams p{};
    std::vector<jv::GsLevelMap> maps((size_t)n_levels);
    for (int l = 0; l < n_levels; ++l) {
        p.lv[l].nbrs = lv_nbrs[l];
        p.lv[l].count = lv_count[l];
        p.lv[l].degree = lv_degree[l];
        if (l > 0) {
            maps[l] = jv::gs_build_level_map(lv_nodes[l], lv_count[l]);
            p.lv[l].hkeys = maps[l].keys.data();
            p.lv[l].hvals = maps[l].vals.data();
            p.lv[l].hmask = maps[l].mask;
            p.lv[l].hshift = maps[l].shift;
        }
    }
    p.entry_node = entry_node;
    p.entry_level = entry_level;
    p.codebooks = codebooks; p.cq = cq; p.bmag = bmag; p.codes = codes; p.code_norms = code_norms;
    p.blocks = blocks; p.fused_norms = fused_norms;
    p.D = D; p.M = M; p.deg0 = deg0; p.Q = Q; p.rerankK = rerankK;
    const size_t vcap = (size_t)1 << vcap_log2;
    int32_t *visited = (int32_t *)aligned_alloc(64, sizeof(int32_t) * vcap * workers);
    long long *spill = (long long *)aligned_alloc(64, sizeof(long long) * (size_t)(spill_cap > 0 ? spill_cap : 1) * workers + 64);
    memset(visited, 0x5a, sizeof(int32_t) * vcap * workers);  // garbage: the kernel must clear it itself
    bool pair = pair_mode != 0 && !wgx_waves;  // same rule as graph_search.cpp
    for (int l = 0; l < n_levels; ++l) pair = pair && lv_degree[l] <= 32;
    pair = pair && M <= 96;   // (graph_search.cpp: above, the two half rows no longer fit the registers; the exchange area is sized for the compacted form)
    // pair_mode 2: the compacted pair form (rows of up to 64 neighbours, codes by ordinal) where the plain pair form does not apply
    bool pairc = pair_mode == 2 && !pair && !wgx_waves && !blocks && M <= 192;
    for (int l = 0; l < n_levels; ++l) pairc = pairc && lv_degree[l] <= 64;
    if (pair_mode == 2 && !pair && !pairc) return -8;
    p.pair = pair ? 1 : (pairc ? 2 : 0);
    p.quad = getenv("GS_EMU_QUAD") ? atoi(getenv("GS_EMU_QUAD")) : 1;   // (on in the emulator unless a test turns it off: more code under test)
    if (ub8 && (ub8 != 2 || !(pair || pairc) || M > 96 || wgx_waves)) return -7;   // (2 = the register-table bound form; 1 = round 4's UB8, gone)
    if (ub8 == 2 && M != 64 && M != 96) return -9;
    std::vector<uint32_t> ubr_tab;
    std::vector<float> ubr_meta;
    if (ub8 == 2) {
        p.ubr = 1;
        p.ubr_trim = getenv("GS_EMU_UBR_TRIM") ? atoi(getenv("GS_EMU_UBR_TRIM")) : 8;   // (small: many trims per search under test)
        ubr_tab.resize((size_t)Q * M * 64);
        ubr_meta.resize((size_t)Q * 4);
        for (int q = 0; q < Q; ++q) jv::gs_ubr_build_ref(codebooks, cq + (size_t)q * D, M, ubr_tab.data() + (size_t)q * M * 64, ubr_meta.data() + (size_t)q * 4, vsf == 0);
        p.ubr_tab = ubr_tab.data();
        p.ubr_meta = ubr_meta.data();
    }
    p.v1_log2 = v1_log2; p.v1_idbits = v1_idbits; p.evict_cap = evict_cap;
    p.prefetch = getenv("GS_EMU_PREFETCH") ? atoi(getenv("GS_EMU_PREFETCH")) : 1;  // on by default in the emulator: more code under test
    int kps = 32;
    for (int l = 0; l < n_levels; ++l)
        if (lv_degree[l] > 32) kps = 64;
    if (wgx_waves) {
        for (int l = 0; l < n_levels; ++l)
            if (lv_degree[l] > 64) return -6;
        p.wgx = 1;
        p.wgx_slots = wgx_slots;
        p.wgx_kps = kps;
        p.wgx_depth = wgx_depth;
        p.wgx_log = 16;
        p.wgx_lut_m = wgx_lut_m > 0 ? wgx_lut_m : M;
        p.prefetch = 0;
    }
    const int ecap = evict_cap > 0 ? evict_cap : jv::GS_EVICT_CAP;
    p.visited = visited; p.vcap_log2 = vcap_log2; p.spill = spill; p.spill_cap = spill_cap; p.cand_cap = cand_cap;
    p.out_ids = out_ids; p.out_scores = out_scores; p.out_stats = out_stats; p.out_status = out_status;
    uint32_t next = 0;
    p.next_query = &next;
    unsigned long long prof[24] = {0};
    if (ub8 == 2) p.ubr_count = prof + 15;
    // DEFER (gs_body.h): on in the emulator from level 1 up (the product defers from level 2: more code under test here);
    // GS_EMU_DEFER=0 turns it off, GS_EMU_DEFER_MIN_LEVEL moves the first deferring level
    if (ub8 == 2 && !(getenv("GS_EMU_DEFER") && atoi(getenv("GS_EMU_DEFER")) == 0)) {
        p.defer = 1;
        p.defer_min_level = getenv("GS_EMU_DEFER_MIN_LEVEL") ? atoi(getenv("GS_EMU_DEFER_MIN_LEVEL")) : 1;
        p.defer_count = prof + 20;
    }
    long collectives = 0;
    // "workers" waves run one after another; each drains part of the queue so that scratch reuse across queries and
    // distinct worker slices are both exercised
    for (int w = 0; w < workers; ++w) {
        jv::GsParams pw = p;
        pw.Q = (int)((long long)Q * (w + 1) / workers);
        const size_t lds_bytes = wgx_waves ? jv::gx_lds_bytes(D, rerankK, cand_cap, ecap, v1_log2, wgx_slots, kps, p.wgx_log, p.wgx_lut_m)
                                           : jv::gs_lds_bytes(D, rerankK, cand_cap, (pair || pairc) ? M : 0, ecap, v1_log2);
        char *lds = (char *)aligned_alloc(64, lds_bytes + 64);
        memset(lds, 0xa5, lds_bytes);
        memset(lds + lds_bytes, 0x3c, 64);  // canary behind the block
        Launch L{&pw, vsf, M / 16, w, lds};
        collectives += wgx_waves ? emu::run_block(lane_main, &L, wgx_waves) : emu::run_wave(lane_main, &L);
        next = (uint32_t)pw.Q;  // the drained worker overshot the counter by one
        for (int i = 0; i < 64; ++i)
            if (lds[lds_bytes + i] != 0x3c) return -3;  // the worker wrote past its LDS block
        free(lds);
    }
    free(visited);
    free(spill);
    if (ub8 && getenv("GS_EMU_PRINT_UB8")) fprintf(stderr, "[gs_emu] ub8 dropped %llu neighbours\n", prof[15]);
    if (ub8 && getenv("GS_EMU_PRINT_UB8")) fprintf(stderr, "[gs_emu] deferred %llu, queries started over %llu (%llu)\n", prof[20], prof[21], prof[22]);
    if (ub8 && ub8_dropped_out) *ub8_dropped_out = (long long)prof[15];
    for (int i = 0; i < 3; ++i) g_last_defer[i] = prof[20 + i];
    return collectives;
}

// gs_host.h's map builder, probed the way the device does (for a direct unit test)
extern "C" int gs_emu_level_lookup(const int32_t *nodes, int count, int32_t node)
{
    jv::GsLevelMap m = jv::gs_build_level_map(nodes, count);
    uint32_t h = ((uint32_t)node * 0x9E3779B1u) >> m.shift;
    for (;;) {
        if (m.keys[h] == node) return m.vals[h];
        if (m.keys[h] == -1) return -1;
        h = (h + 1) & m.mask;
    }
}

// gs_host.h's restatement of ubr_table_kernel (the GPU test compares the kernel's bytes with it)
extern "C" void gs_emu_ubr_table(const float *codebooks, const float *cq, int M, uint32_t *tab, float *meta4)
{
    jv::gs_ubr_build_ref(codebooks, cq, M, tab, meta4);
}
extern "C" void gs_emu_ubr_table_vsf(const float *codebooks, const float *cq, int M, uint32_t *tab, float *meta4, int vsf /* 0 = euclidean */)
{
    jv::gs_ubr_build_ref(codebooks, cq, M, tab, meta4, vsf == 0);
}
