// (the test harness keeps the measured-and-switched-off kernel variants compiled: they stay under test)
#define JV_EXPERIMENTAL 1
// rd_emu.cpp — TEST HARNESS: runs the body of retain_diverse_kernel (jvector_amd/csrc/rd_body.h, unchanged) on the lane
// emulator, one emulated wavefront per node.
#include <cmath>
#include <cstdint>
#include <cstdlib>

#include "hip_emu.h"

#define GS_FN inline
static inline int gs_lane() { return emu::lane(); }
static inline void gs_barrier() { emu::barrier(); }
static inline uint64_t gs_ballot(bool p) { return emu::ballot(p); }
static inline long long gs_shfl(long long v, int src) { return emu::shfl(v, src); }
static inline long long gs_shfl_xor(long long v, int m) { return emu::shfl(v, emu::lane() ^ m); }
static inline int32_t gs_shfl32(int32_t v, int src) { return (int32_t)emu::shfl((long long)v, src); }
static inline void gs_fetch_add64(unsigned long long *p, unsigned long long v) { *p += v; }
static inline double gs_sqrt(double x) { return std::sqrt(x); }

#include "../../jvector_amd/csrc/rd_body.h"

namespace {
struct Launch {
    const jv::RdParams *p;
    int node;
    char *lds;
};
void node_main(void *a)
{
    const Launch &L = *(const Launch *)a;
    if (L.p->codebooks) jv::rd_node<true>(*L.p, L.node, L.lds);
    else if (L.p->sq) jv::rd_node<false, false, true>(*L.p, L.node, L.lds);
    else jv::rd_node<false>(*L.p, L.node, L.lds);
}
}  // namespace

static int rd_emu_run_any(const float *tri, const float *codebooks, const uint8_t *codes, int64_t n, const int32_t *cand_nodes, const float *cand_scores,
                          const int32_t *cand_count, const int32_t *diverse_before, int P, int C, int M, int k, int vsf, int maxDegree,
                          float alpha, int32_t *selected_out, int32_t *n_selected_out, float *short_edges_out);

// the table-free form: `codebooks` = [M][k][8] centroids (what RdParams::codebooks points at on the device)
extern "C" int rd_emu_run_tf(const float *tri, const float *codebooks, const uint8_t *codes, int64_t n, const int32_t *cand_nodes, const float *cand_scores,
                             const int32_t *cand_count, const int32_t *diverse_before, int P, int C, int M, int k, int vsf, int maxDegree,
                             float alpha, int32_t *selected_out, int32_t *n_selected_out, float *short_edges_out)
{
    return rd_emu_run_any(tri, codebooks, codes, n, cand_nodes, cand_scores, cand_count, diverse_before, P, C, M, k, vsf, maxDegree, alpha,
                          selected_out, n_selected_out, short_edges_out);
}

extern "C" int rd_emu_run(const float *tri, const uint8_t *codes, int64_t n, const int32_t *cand_nodes, const float *cand_scores,
                          const int32_t *cand_count, const int32_t *diverse_before, int P, int C, int M, int k, int vsf, int maxDegree,
                          float alpha, int32_t *selected_out, int32_t *n_selected_out, float *short_edges_out)
{
    return rd_emu_run_any(tri, nullptr, codes, n, cand_nodes, cand_scores, cand_count, diverse_before, P, C, M, k, vsf, maxDegree, alpha,
                          selected_out, n_selected_out, short_edges_out);
}

static int rd_emu_run_any(const float *tri, const float *codebooks, const uint8_t *codes, int64_t n, const int32_t *cand_nodes, const float *cand_scores,
                          const int32_t *cand_count, const int32_t *diverse_before, int P, int C, int M, int k, int vsf, int maxDegree,
                          float alpha, int32_t *selected_out, int32_t *n_selected_out, float *short_edges_out)
{
    jv::RdParams p{};
    p.codebooks = codebooks;
    p.wide_stage = getenv("RD_EMU_WIDE") ? atoi(getenv("RD_EMU_WIDE")) : 1;
    p.split = getenv("RD_EMU_SPLIT") ? atoi(getenv("RD_EMU_SPLIT")) : 1;   // (idle lanes share a slot's entries; 0 = one lane per slot)
    p.chunk = getenv("RD_EMU_CHUNK") ? atoi(getenv("RD_EMU_CHUNK")) : 8;   // (incremental tests; 0 = every test examines every slot)
    p.tri = tri; p.codes = codes; p.n = n; p.cand_nodes = cand_nodes; p.cand_scores = cand_scores; p.cand_count = cand_count;
    p.diverse_before = diverse_before; p.P = P; p.C = C; p.M = M; p.k = k; p.vsf = vsf; p.maxDegree = maxDegree; p.alpha = alpha;
    p.selected_out = selected_out; p.n_selected_out = n_selected_out; p.short_edges_out = short_edges_out;
    float *sq = nullptr;
    if (getenv("RD_EMU_SQUARE") && atoi(getenv("RD_EMU_SQUARE")) && !codebooks) {   // the square form of the same table
        sq = (float *)malloc(sizeof(float) * (size_t)M * k * k);
        const int64_t block = (int64_t)k * (k + 1) / 2;
        for (int m = 0; m < M; ++m)
            for (int i = 0; i < k; ++i)
                for (int j = 0; j < k; ++j) {
                    const int r = i < j ? i : j, c = i < j ? j : i;
                    sq[((size_t)m * k + i) * k + j] = tri[m * block + jv::rd_tri_row(r, k) + (c - r)];
                }
        p.sq = sq;
    }
    const size_t lds_bytes = jv::rd_lds_bytes(C, M, codebooks != nullptr);
    char *lds = (char *)aligned_alloc(64, (lds_bytes + 63) & ~(size_t)63);
    for (int node = 0; node < P; ++node) {
        for (size_t i = 0; i < lds_bytes; ++i) lds[i] = (char)0xA5;  // stale LDS must never reach a result
        Launch L{&p, node, lds};
        emu::run_wave(node_main, &L);
    }
    free(lds);
    free(sq);
    return 0;
}
