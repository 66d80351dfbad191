// mock_kernels.cpp — CPU implementations of the library's kernel LAUNCHERS (namespace jv, declared in jv_internal.h) for
// the mock device.  Arithmetic comes from the oracle's primitives (this is test infrastructure; the point of the mock is the
// HOST code around the kernels, not the kernels), except where a kernel BODY is shared source: the device-resident graph
// traversal runs gs_body.h on the 64-lane emulator, build-time scoring and training run bs_body.h / km_body.h as loops.
#include <algorithm>
#include <cmath>
#include <mutex>
#include <vector>

#include "../emu/hip_emu.h"

#define GS_FN inline
#define GS_SCHED_FENCE() ((void)0)
#define BS_FN static inline
#define KM_FN static inline
#define GS_NOINLINE static
#define GS_LDS_AS
#define GS_GLOBAL_AS
static inline int gs_lane() { return emu::lane(); }
// gs_body.h's sync point: wave scope (= the block barrier in a one-wave block; the control wave of the workgroup form must not
// wait for the expanders).  Bodies written for several waves use gs_block_barrier().
static inline void gs_barrier() { emu::wave_barrier(); }
static inline int gs_tid() { return emu::lane(); }
static inline int gs_block_threads() { return emu::current()->nl; }
static inline void gs_block_barrier() { emu::barrier(); }
static inline int32_t gs_lds_load(const int32_t *p) { return *(const volatile int32_t *)p; }
static inline void gs_lds_store(int32_t *p, int32_t v) { *(volatile int32_t *)p = v; }
static inline int32_t gs_lds_add(int32_t *p, int32_t v)
{
    const int32_t old = *p;
    *p = old + v;
    return old;
}
static inline void gs_spin_pause() { emu::switch_to_next_live(); }
static inline void gs_sched_fence() {}
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
static inline void gs_gather64(float v, float (&out)[64]) { emu::gather64(v, out); }
static inline double gs_sqrt(double x) { return std::sqrt(x); }
static inline float gs_rsq_approx(float x) { return 1.0f / std::sqrt(x); }
static inline double bs_sqrt(double x) { return std::sqrt(x); }
static inline float gs_fmaf(float a, float b, float c) { return fmaf(a, b, c); }
typedef emu::f32x16 gs_f32x16;
static inline gs_f32x16 gs_mfma_32x32x2(float a, float b, gs_f32x16 c) { return emu::mfma_32x32x2(a, b, c); }

#include "../../jvector_amd/csrc/jv_device.h"
#include "../../jvector_amd/csrc/jv_internal.h"

#include "../../jvector_amd/csrc/bl_body.h"
#include "../../jvector_amd/csrc/bs_body.h"
#include "../../jvector_amd/csrc/ed_body.h"
#include "../../jvector_amd/csrc/gs_body.h"
#include "../../jvector_amd/csrc/gx_body.h"
#include "../../jvector_amd/csrc/gs_host.h"
#include "../../jvector_amd/csrc/km_body.h"
#include "../../jvector_amd/csrc/rd_body.h"
#include "../../jvector_amd/csrc/rt_body.h"
#include "../../oracle/jv_oracle.h"

// The lane emulator keeps the running wave's context in statics: emulated launches from several host threads take turns.
static std::mutex g_emu_mu;
template <typename Arg>
static void run_wave_locked(void (*fn)(void *), Arg *arg)
{
    std::lock_guard<std::mutex> lock(g_emu_mu);
    emu::run_wave(fn, (void *)arg);
}
template <typename Arg>
static void run_block_locked(void (*fn)(void *), Arg *arg, int waves)
{
    std::lock_guard<std::mutex> lock(g_emu_mu);
    emu::run_block(fn, (void *)arg, waves);
}

namespace jv {

namespace {
const float NEG_INF = -INFINITY;

float finish(int vsf, float sum, float norm, float bmag)
{
    if (vsf == VSF_RAW) return sum;
    if (vsf == VSF_L2) return 1.0f / (1.0f + sum);
    if (vsf == VSF_COS) {
        const float prod = norm * bmag;
        sum = (float)((double)sum / std::sqrt((double)prod));
    }
    return (1.0f + sum) / 2.0f;
}
float row_sum(const float *lut, const uint8_t *row, int M)
{
    float s = 0.0f;
    for (int m = 0; m < M; ++m) s += lut[m * kClusters + row[m]];
    return s;
}
jvo_pq as_oracle(const jv_pq *pq)
{
    return jvo_pq{pq->D, pq->M, pq->k, pq->sizes.data(), pq->offsets.data(), pq->d_codebooks, pq->d_centroid};
}
}  // namespace

int launch_self_magnitudes(hipStream_t, const jv_pq *pq)
{
    for (int m = 0; m < pq->M; ++m)
        jvo_calculate_partial_self_magnitudes(pq->d_codebooks + pq->cb_offsets[m], m, pq->sizes[m], pq->k, pq->d_self_mag);
    return JV_OK;
}
int launch_center_queries(hipStream_t, const jv_pq *pq, const float *d_q, int Q, float *d_cq)
{
    for (int64_t i = 0; i < (int64_t)Q * pq->D; ++i) d_cq[i] = pq->d_centroid ? d_q[i] - pq->d_centroid[i % pq->D] : d_q[i];
    return JV_OK;
}
int launch_lut_build(hipStream_t, const jv_pq *pq, const float *d_cq, int Q, int lut_vsf, float *d_luts)
{
    for (int q = 0; q < Q; ++q)
        for (int m = 0; m < pq->M; ++m)
            jvo_calculate_partial_sums(pq->d_codebooks + pq->cb_offsets[m], m, pq->sizes[m], pq->k, d_cq + (size_t)q * pq->D,
                                       pq->offsets[m], lut_vsf == VSF_L2 ? JVO_EUCLIDEAN : JVO_DOT_PRODUCT,
                                       d_luts + (size_t)q * pq->M * pq->k);
    return JV_OK;
}
int launch_query_magnitudes(hipStream_t, const jv_pq *pq, const float *d_cq, int Q, int kind, float *d_bmag)
{
    for (int q = 0; q < Q; ++q) {
        const float *cq = d_cq + (size_t)q * pq->D;
        if (kind == 0) d_bmag[q] = jvo_dot(cq, cq, pq->D);  // PQDecoder.CosineDecoder :121
        else {                                              // FusedPQDecoder.CosineDecoder :188
            float qm = 0.0f;
            for (int m = 0; m < pq->M; ++m) qm += jvo_dot_off(cq, pq->offsets[m], cq, pq->offsets[m], pq->sizes[m]);
            d_bmag[q] = qm;
        }
    }
    return JV_OK;
}
int launch_pq_encode(hipStream_t, const jv_pq *pq, const float *d_vecs, int64_t count, uint8_t *d_codes)
{
    const jvo_pq o = as_oracle(pq);
    for (int64_t i = 0; i < count; ++i) jvo_pq_encode(&o, d_vecs + i * pq->D, d_codes + i * pq->M);
    return JV_OK;
}
int launch_pq_encode_anisotropic(hipStream_t, const jv_pq *pq, const float *d_vecs, int64_t count, uint8_t *d_codes)
{
    const jvo_pq o = as_oracle(pq);
    for (int64_t i = 0; i < count; ++i) jvo_pq_encode_anisotropic(&o, pq->aniso, d_vecs + i * pq->D, d_codes + i * pq->M);
    return JV_OK;
}
int launch_code_norms(hipStream_t, const jv_ctx *, const float *d_table, int M, const uint8_t *d_codes, int64_t count, float *d_out)
{
    for (int64_t i = 0; i < count; ++i) d_out[i] = row_sum(d_table, d_codes + i * M, M);
    return JV_OK;
}
int launch_adc(hipStream_t, const jv_ctx *, const float *d_luts, const float *d_bmag, int Q, int M, int vsf, const uint8_t *d_codes,
               const float *d_norms, int64_t n_codes, int64_t first, int64_t count, const int32_t *d_ordinals, float *d_out)
{
    for (int q = 0; q < Q; ++q)
        for (int64_t i = 0; i < count; ++i) {
            const int64_t row = d_ordinals ? (int64_t)d_ordinals[(int64_t)q * count + i] : first + i;
            float *o = d_out + (int64_t)q * count + i;
            if (d_ordinals && (row < 0 || row >= n_codes)) {
                *o = NEG_INF;
                continue;
            }
            const float s = row_sum(d_luts + (size_t)q * M * kClusters, d_codes + row * M, M);
            *o = finish(vsf, s, vsf == VSF_COS ? d_norms[row] : 0.0f, vsf == VSF_COS ? d_bmag[q] : 0.0f);
        }
    return JV_OK;
}
int launch_fused(hipStream_t, const jv_ctx *, const float *d_luts, const float *d_bmag, int Q, int M, int vsf, const uint8_t *d_blocks,
                 const int32_t *d_neighbors, const float *d_norms, int maxDegree, int64_t n_nodes, const int32_t *d_origins,
                 float *d_out, int32_t *d_neighbors_out)
{
    for (int q = 0; q < Q; ++q)
        for (int i = 0; i < maxDegree; ++i) {
            const int64_t origin = d_origins[q], row = origin * maxDegree + i;
            float *o = d_out + (int64_t)q * maxDegree + i;
            const bool in = origin >= 0 && origin < n_nodes;
            const int32_t nb = in ? d_neighbors[row] : -1;
            if (d_neighbors_out) d_neighbors_out[(int64_t)q * maxDegree + i] = nb;
            if (nb < 0) {
                *o = NEG_INF;
                continue;
            }
            const float s = row_sum(d_luts + (size_t)q * M * kClusters, d_blocks + row * M, M);
            *o = finish(vsf, s, vsf == VSF_COS ? d_norms[row] : 0.0f, vsf == VSF_COS ? d_bmag[q] : 0.0f);
        }
    return JV_OK;
}
namespace {
struct RdLaunch {
    const RdParams *p;
    int node;
    char *lds;
};
void rd_main(void *a)
{
    const RdLaunch &L = *(const RdLaunch *)a;
    if (L.p->codebooks) rd_node<true>(*L.p, L.node, L.lds);
    else if (L.p->sq) rd_node<false, false, true>(*L.p, L.node, L.lds);
    else rd_node<false>(*L.p, L.node, L.lds);
}
}  // namespace
size_t retain_diverse_lds_bytes(int C, int M) { return rd_lds_bytes(C, M); }
// the shared kernel body (rd_body.h) on the lane emulator, one wavefront per node
int launch_retain_diverse(hipStream_t, const jv_ctx *ctx, const RdParams &p)
{
    const size_t lds_bytes = rd_lds_bytes(p.C, p.M, p.codebooks != nullptr);
    if (lds_bytes > ctx->lds_per_block) {
        set_error("retain_diverse: %d candidates x %d code bytes need %zu bytes of LDS (limit %zu)", p.C, p.M, lds_bytes, ctx->lds_per_block);
        return JV_ERR_UNSUPPORTED;
    }
    std::vector<char> lds(lds_bytes + 64);
    char *base = (char *)(((uintptr_t)lds.data() + 15) & ~(uintptr_t)15);
    for (int node = 0; node < p.P; ++node) {
        RdLaunch L{&p, node, base};
        run_wave_locked(rd_main, &L);
    }
    return JV_OK;
}
int launch_shard_interleave(hipStream_t, const int32_t *d_ids, const float *d_sc, int P, int Q, int k, int32_t *d_out_ids, float *d_out_sc)
{
    for (int p = 0; p < P; ++p)
        for (int q = 0; q < Q; ++q)
            for (int j = 0; j < k; ++j) {
                const int64_t src = ((int64_t)p * Q + q) * k + j, dst = ((int64_t)q * P + p) * k + j;
                d_out_ids[dst] = d_ids[src];
                d_out_sc[dst] = d_sc[src];
            }
    return JV_OK;
}
int launch_shard_sanitize(hipStream_t, int32_t *d_ids, float *d_sc, int64_t n, int64_t lo, int64_t hi)
{
    for (int64_t i = 0; i < n; ++i)
        if (d_ids[i] < lo || d_ids[i] >= hi) {
            d_ids[i] = -1;
            d_sc[i] = -INFINITY;
        }
    return JV_OK;
}
int launch_shard_localize(hipStream_t, const int32_t *d_gids, int64_t n, int64_t base, int64_t count, int32_t *d_local)
{
    for (int64_t i = 0; i < n; ++i) d_local[i] = (d_gids[i] >= base && d_gids[i] < base + count) ? (int32_t)(d_gids[i] - base) : -1;
    return JV_OK;
}
int launch_shard_select(hipStream_t, const int32_t *d_gids, const float *d_exact, const long long *d_ranges, int P, int64_t n, float *d_out)
{
    for (int64_t i = 0; i < n; ++i) {
        float v = NEG_INF;
        for (int p = 0; p < P && d_gids[i] >= 0; ++p)
            if (d_gids[i] >= d_ranges[2 * p] && d_gids[i] < d_ranges[2 * p] + d_ranges[2 * p + 1]) {
                v = d_exact[(int64_t)p * n + i];
                break;
            }
        d_out[i] = v;
    }
    return JV_OK;
}
int launch_row_sqnorms(hipStream_t, const float *d_vecs, int64_t n, int D, float *d_out)
{
    for (int64_t i = 0; i < n; ++i) {
        float s = 0.0f;
        for (int j = 0; j < D; ++j) s += d_vecs[i * D + j] * d_vecs[i * D + j];
        d_out[i] = s;
    }
    return JV_OK;
}
int launch_gather_rows(hipStream_t, const float *d_vecs, int64_t n, int D, const int32_t *d_ord, int P, float *d_out, int32_t *d_cand, int B)
{
    for (int p = 0; p < P; ++p) {
        const int64_t o = d_ord[p];
        const bool ok = o >= 0 && o < n;
        for (int j = 0; j < D; ++j) d_out[(int64_t)p * D + j] = ok ? d_vecs[o * D + j] : 0.0f;
        if (!ok)
            for (int j = 0; j < B; ++j) d_cand[(int64_t)p * B + j] = -1;
    }
    return JV_OK;
}
int launch_exact_gather(hipStream_t, const float *d_vecs, int64_t n, int D, const float *d_q, int Q, int vsf, const int32_t *d_ord,
                        int B, float *d_out, float *, const float *)
{
    for (int q = 0; q < Q; ++q)
        for (int b = 0; b < B; ++b) {
            const int64_t o = d_ord[(int64_t)q * B + b];
            d_out[(int64_t)q * B + b] = (o < 0 || o >= n) ? NEG_INF : jvo_compare(vsf, d_q + (size_t)q * D, d_vecs + o * D, D);
        }
    return JV_OK;
}
// the rerank fused into the traversal wave (the emulated gs_body.h does the arithmetic): the same shape rule as k_exact.hip
int exact_fused_rows(const float *d_vecs, int D, const float *d_q, int Q, int vsf, int B, const float *d_vnorm)
{
    if (D % 8 != 0 || D < 8 || (reinterpret_cast<uintptr_t>(d_vecs) & 15) != 0 || (reinterpret_cast<uintptr_t>(d_q) & 15) != 0 || (vsf == VSF_COS && !d_vnorm)) return 0;
    if (B < 1 || B > 64 * GS_RR_MAX_ROUNDS) return 0;
    const int rem = B % 64;
    if (B >= 64 && rem >= 4 && rem <= 32 && Q >= 2) return B - rem;
    return B;
}
int launch_query_sqnorms(hipStream_t, const float *d_q, int D, int Q, float *d_qnorm)
{
    for (int q = 0; q < Q; ++q) {
        float n1 = 0.0f;
        for (int j = 0; j < D; ++j) n1 += d_q[(size_t)q * D + j] * d_q[(size_t)q * D + j];
        d_qnorm[q] = n1;
    }
    return JV_OK;
}
int launch_exact_gather_tail(hipStream_t, const float *d_vecs, int64_t n, int D, const float *d_q, int Q, int vsf, const int32_t *d_ord, int B,
                             int first, float *d_out, const float *, const float *)
{
    for (int q = 0; q < Q; ++q)
        for (int b = first; b < B; ++b) {
            const int64_t o = d_ord[(int64_t)q * B + b];
            d_out[(int64_t)q * B + b] = (o < 0 || o >= n) ? NEG_INF : jvo_compare(vsf, d_q + (size_t)q * D, d_vecs + o * D, D);
        }
    return JV_OK;
}
// ---- NVQ (k_nvq.hip): the oracle's restatement; the mock's "derived" array simply carries the raw parameters ----
int launch_nvq_mean(hipStream_t, const float *d_vecs, int64_t n, int D, float *d_mean)
{
    jvo_nvq_global_mean(d_vecs, n, D, d_mean);
    return JV_OK;
}
size_t nvq_encode_lds_bytes(int D, int S) { return sizeof(float) * (3 * (size_t)(D / S + 1) + 96); }
int launch_nvq_encode(hipStream_t, const jv_ctx *ctx, const float *d_vecs, int64_t count, int D, int S, const float *d_mean, int learn,
                      const float *, uint8_t *d_bytes, int ld, float *d_params)
{
    if (nvq_encode_lds_bytes(D, S) > 65536) {
        set_error("nvq_encode: sub-vector too long for LDS");
        return JV_ERR_UNSUPPORTED;
    }
    for (int64_t i = 0; i < count; ++i) jvo_nvq_encode(d_mean, D, S, d_vecs + i * D, learn, d_bytes + i * ld, d_params + i * 4 * S);
    return JV_OK;
}
int launch_nvq_derive(hipStream_t, const float *d_params, int64_t units, float *d_derived)
{
    memcpy(d_derived, d_params, sizeof(float) * 4 * (size_t)units);
    return JV_OK;
}
int launch_nvq_cosnorm(hipStream_t, const uint8_t *, int, int64_t n, int, int, const float *, const float *, float *d_out)
{
    for (int64_t i = 0; i < n; ++i) d_out[i] = 0.0f;   // the mock's gather recomputes the sum
    return JV_OK;
}
int launch_nvq_gather(hipStream_t, const uint8_t *d_bytes, int ld, int64_t n, int D, int S, const float *d_derived, const float *,
                      const float *d_mean, const float *d_q, int Q, int vsf, const int32_t *d_ord, int B, float *d_out, float *, float *)
{
    for (int q = 0; q < Q; ++q)
        for (int b = 0; b < B; ++b) {
            const int64_t o = d_ord[(int64_t)q * B + b];
            d_out[(int64_t)q * B + b] = (o < 0 || o >= n) ? NEG_INF
                : jvo_nvq_score(vsf, d_mean, D, S, d_q + (size_t)q * D, d_bytes + o * ld, d_derived + o * 4 * S);
        }
    return JV_OK;
}
int launch_exact_scan(hipStream_t, const jv_ctx *, const float *d_vecs, int D, const float *d_q, int Q, int vsf, int64_t first,
                      int64_t count, float *d_out, float *)
{
    for (int q = 0; q < Q; ++q)
        for (int64_t i = 0; i < count; ++i)
            d_out[(int64_t)q * count + i] = jvo_compare(vsf, d_q + (size_t)q * D, d_vecs + (first + i) * D, D);
    return JV_OK;
}
namespace {
struct EdLaunch {
    const EdParams *p;
    int vsf;
    int64_t n_tile;
    int q_tile;
    float *lds;
};
void ed_main(void *a)
{
    const EdLaunch &L = *(const EdLaunch *)a;
    if (L.vsf == VSF_L2) ed_tile<VSF_L2>(*L.p, L.n_tile * ED_TN, L.q_tile * ED_TQ, L.lds);
    else if (L.vsf == VSF_DOT) ed_tile<VSF_DOT>(*L.p, L.n_tile * ED_TN, L.q_tile * ED_TQ, L.lds);
    else ed_tile<VSF_COS>(*L.p, L.n_tile * ED_TN, L.q_tile * ED_TQ, L.lds);
}
}  // namespace
// the shared kernel body (ed_body.h) on the lane emulator, launch geometry of k_exact_dense.hip
int launch_exact_scan_dense(hipStream_t, const float *d_vecs, int D, const float *d_q, int Q, int vsf, int64_t first, int64_t count,
                            float *d_out)
{
    if (Q == 0 || count == 0) return JV_OK;
    const EdParams p{d_vecs, d_q, d_out, first, count, D, Q};
    const int64_t n_tiles = (count + ED_TN - 1) / ED_TN;
    const int q_tiles = (Q + ED_TQ - 1) / ED_TQ;
    const int64_t blocks_padded = (n_tiles * q_tiles + 7) / 8 * 8;
    std::vector<float> lds((size_t)ED_LDS_FLOATS);
    for (int64_t b = 0; b < blocks_padded; ++b) {
        int64_t nt;
        int qt;
        if (!ed_block_to_tile(b, blocks_padded, n_tiles, q_tiles, &nt, &qt)) continue;
        std::fill(lds.begin(), lds.end(), NAN);
        EdLaunch L{&p, vsf, nt, qt, lds.data()};
        emu::run_block(ed_main, &L, jv::ED_WAVES);
    }
    return JV_OK;
}
int launch_frontier(hipStream_t, int vsf, const float *d_luts, const float *d_bmag, const int32_t *d_slot_query, const int32_t *d_origins,
                    const int32_t *d_ord_index, const int32_t *d_ords, const jv_fused *fused, const jv_codes *codes, float *d_out, int S,
                    int W, const jv_pq *, const float *)
{
    const int M = codes->M;
    for (int slot = 0; slot < S; ++slot)
        for (int i = 0; i < W; ++i) {
            float *o = d_out + (int64_t)slot * W + i;
            *o = NEG_INF;
            const int lq = d_slot_query[slot];
            if (lq < 0) continue;
            const int64_t origin = (fused && d_origins) ? d_origins[slot] : -1;
            const uint8_t *rp;
            float nrm = 0.0f;
            if (origin >= 0) {
                if (i >= fused->maxDegree || origin >= fused->count) continue;
                const int64_t row = origin * fused->maxDegree + i;
                if (fused->d_neighbors[row] < 0) continue;
                rp = fused->d_blocks + row * M;
                if (vsf == VSF_COS) nrm = fused->d_norms[row];
            } else {
                const int oi = d_ord_index[slot];
                if (oi < 0) continue;
                const int64_t ord = d_ords[(int64_t)oi * W + i];
                if (ord < 0 || ord >= codes->count) continue;
                rp = codes->d_codes + ord * M;
                if (vsf == VSF_COS) nrm = codes->d_norms[ord];
            }
            *o = finish(vsf, row_sum(d_luts + (size_t)lq * M * kClusters, rp, M), nrm, vsf == VSF_COS ? d_bmag[lq] : 0.0f);
        }
    return JV_OK;
}

size_t topk_scratch_bytes(int, int) { return 256; }
int launch_topk(hipStream_t, const jv_ctx *, const float *d_scores, const int32_t *d_ids, int Q, int64_t n, int64_t stride,
                int32_t id_base, int k, int32_t *d_out_ids, float *d_out_scores, void *, const unsigned int *d_row_counts)
{
    if (Q == 0 || k == 0) return JV_OK;
    if (k > 8192) {
        set_error("topk: k=%d exceeds the supported maximum %d", k, 8192);
        return JV_ERR_UNSUPPORTED;
    }
    std::vector<int64_t> keys;
    for (int q = 0; q < Q; ++q) {
        keys.clear();
        const int64_t row_n = d_row_counts ? std::min<int64_t>(n, d_row_counts[q]) : n;
        for (int64_t c = 0; c < row_n; ++c) {
            const int32_t id = d_ids ? d_ids[(int64_t)q * stride + c] : (int32_t)(id_base + c);
            if (id < 0) continue;
            keys.push_back(jvo_nodequeue_encode(id, d_scores[(int64_t)q * stride + c]));
        }
        std::sort(keys.begin(), keys.end(), std::greater<int64_t>());
        for (int j = 0; j < k; ++j) {
            const bool have = j < (int)keys.size();
            d_out_ids[(int64_t)q * k + j] = have ? (int32_t)~(uint32_t)(keys[j] & 0xFFFFFFFFll) : -1;
            d_out_scores[(int64_t)q * k + j] = have ? jvo_sortable_int_to_float((int32_t)(keys[j] >> 32)) : NEG_INF;
        }
    }
    return JV_OK;
}
// the shared kernel body (rt_body.h) on the lane emulator, one wavefront per query
namespace {
struct RtLaunch {
    const jv::RtParams *p;
    int q;
    char *lds;
};
void rt_main(void *a)
{
    const RtLaunch &L = *(const RtLaunch *)a;
    jv::rt_query(*L.p, L.q, L.lds);
}
}  // namespace
int launch_rerank_ties(hipStream_t, const RtParams &p)
{
    const size_t lds_bytes = jv::rt_lds_bytes(p.rerankK, p.K);
    char *lds = (char *)aligned_alloc(64, (lds_bytes + 63) & ~(size_t)63);
    for (int q = 0; q < p.Q; ++q) {
        memset(lds, 0xA5, lds_bytes);
        RtLaunch L{&p, q, lds};
        run_wave_locked(rt_main, &L);
    }
    free(lds);
    return JV_OK;
}
bool adc_mq_supported(int, const uint8_t *) { return false; }  // the multi-query scan kernels are not mocked
int launch_adc_mq_store(hipStream_t, const jv_ctx *, const float *, const float *, int, int, int, const uint8_t *, const float *, int64_t,
                        int64_t, int64_t, float *)
{
    set_error("mock device: adc_mq_store is not available");
    return JV_ERR_UNSUPPORTED;
}
int launch_adc_mq_filter(hipStream_t, const jv_ctx *, const float *, const float *, int, int, int, const uint8_t *, const float *, int64_t,
                         int64_t, const float *, int, int32_t *, float *, unsigned int *, int)
{
    set_error("mock device: adc_mq_filter is not available");
    return JV_ERR_UNSUPPORTED;
}
bool adc_bq_supported(int, const uint8_t *, size_t) { return false; }   // (the flat scan's filter kernels are not mocked)
size_t adc_bq_scratch_bytes(int, int) { return 0; }
int launch_adc_bq_scan(hipStream_t, const jv_ctx *, const float *, const float *, int, int, int, const uint8_t *, const float *, int64_t, int64_t, const float *, int,
                       int32_t *, unsigned int *, int, void *)
{
    set_error("mock device: adc_bq_scan is not available");
    return JV_ERR_UNSUPPORTED;
}
int launch_adc_bq_exact(hipStream_t, const jv_ctx *, const float *, const float *, int, int, int, const uint8_t *, const float *, int64_t, const float *, int,
                        const int32_t *, float *, const unsigned int *, unsigned int *, int, int)
{
    set_error("mock device: adc_bq_exact is not available");
    return JV_ERR_UNSUPPORTED;
}
int launch_add_id_base(hipStream_t, int32_t *d_ids, int64_t n, int32_t base)
{
    for (int64_t i = 0; i < n; ++i)
        if (d_ids[i] >= 0) d_ids[i] += base;
    return JV_OK;
}

// ---- build-time scoring: bs_body.h as loops ----
static BsPq bs_pq_of(const jv_pq *pq)
{
    return BsPq{pq->d_codebooks, pq->d_cb_offsets, pq->d_sizes, pq->d_offsets, pq->d_centroid, pq->D, pq->M, pq->k};
}
int launch_pair_table(hipStream_t, const jv_pq *pq, int vsf, float *d_out)
{
    const BsPq b = bs_pq_of(pq);
    for (int64_t t = 0; t < (int64_t)pq->M * pq->k; ++t) bs_pair_table_row(b, vsf, t, d_out);
    return JV_OK;
}
int launch_pair_table_square(hipStream_t, const float *d_tri, int M, int k, float *d_sq)
{
    const int64_t block = (int64_t)k * (k + 1) / 2;
    for (int m = 0; m < M; ++m)
        for (int i = 0; i < k; ++i)
            for (int j = 0; j < k; ++j) {
                const int r = i < j ? i : j, c = i < j ? j : i;
                d_sq[((size_t)m * k + i) * k + j] = d_tri[m * block + bs_tri_row(r, k) + (c - r)];
            }
    return JV_OK;
}
int launch_pair_scores(hipStream_t, const float *d_tri, int vsf, const jv_codes *codes, const int32_t *d_node1, int P, const int32_t *d_node2,
                       int B, float *d_out)
{
    for (int64_t t = 0; t < (int64_t)P * B; ++t)
        bs_pair_score(d_tri, vsf, codes->M, codes->pq->k, codes->d_codes, codes->count, d_node1, d_node2, B, t, d_out);
    return JV_OK;
}
int launch_fused_gather(hipStream_t, const jv_codes *codes, const int32_t *d_neighbors, int maxDegree, int64_t count, uint8_t *d_blocks)
{
    const int M = codes->M, chunk = M % 16 == 0 ? 16 : 1;
    for (int64_t t = 0; t < count * maxDegree * (M / chunk); ++t)
        bs_fused_gather(codes->d_codes, codes->count, d_neighbors, maxDegree, M, chunk, t, d_blocks);
    return JV_OK;
}
int launch_pq_decode(hipStream_t, const jv_codes *codes, const int32_t *d_ordinals, int64_t first, int64_t count, float *d_out)
{
    const BsPq b = bs_pq_of(codes->pq);
    for (int64_t t = 0; t < count * codes->pq->D; ++t) bs_decode(b, codes->d_codes, codes->count, d_ordinals, first, t, d_out);
    return JV_OK;
}
int launch_direct_scores(hipStream_t, const jv_codes *codes, int vsf, const float *d_cq, int Q, const int32_t *d_ordinals, int B,
                         float *d_qnorm, float *d_out)
{
    const BsPq b = bs_pq_of(codes->pq);
    if (vsf == VSF_COS)
        for (int64_t q = 0; q < Q; ++q) bs_query_norm(d_cq, codes->pq->D, q, d_qnorm);
    for (int64_t t = 0; t < (int64_t)Q * B; ++t) bs_direct_score(b, vsf, codes->d_codes, codes->count, d_cq, d_qnorm, d_ordinals, B, t, d_out);
    return JV_OK;
}

// ---- PQ training: km_body.h as loops, seeding on the lane emulator ----
int launch_km_centroid(hipStream_t, const float *d_X, int64_t n, int D, float *d_out)
{
    for (int64_t d = 0; d < D; ++d) km_centroid_dim(d_X, n, D, d, d_out);
    return JV_OK;
}
int launch_km_center(hipStream_t, const float *d_X, const float *d_centroid, int64_t n, int D, float *d_Xc)
{
    for (int64_t t = 0; t < n * D; ++t) km_center(d_X, d_centroid, D, t, d_Xc);
    return JV_OK;
}
namespace {
struct PP {
    const KmParams *p;
    int m;
};
void pp_main(void *a)
{
    const PP &x = *(const PP *)a;
    km_pp_init(*x.p, x.m);
}
}  // namespace
int launch_km_pp_init(hipStream_t, const KmParams &p)
{
    for (int m = 0; m < p.M; ++m) {
        PP a{&p, m};
        run_wave_locked(pp_main, &a);
    }
    return JV_OK;
}
int launch_km_assign(hipStream_t, const KmParams &p)
{
    for (int64_t t = 0; t < p.n * p.M; ++t) km_assign(p, t);
    return JV_OK;
}
int launch_km_replay(hipStream_t, const KmParams &p, int first_pass)
{
    for (int64_t t = 0; t < (int64_t)p.M * p.k; ++t) km_replay(p, first_pass, t);
    return JV_OK;
}
int launch_km_update_centroids(hipStream_t, const KmParams &p)
{
    for (int64_t t = 0; t < (int64_t)p.M * p.k; ++t) km_centroids(p, t);
    for (int64_t m = 0; m < p.M; ++m) km_fill_empties(p, m);
    return JV_OK;
}
int launch_km_aniso_round(hipStream_t, const KmParams &p)
{
    for (int64_t t = 0; t < (int64_t)p.M * p.k; ++t) km_centroids_aniso(p, t);
    for (int64_t m = 0; m < p.M; ++m) km_fill_empties(p, m);
    for (int64_t t = 0; t < (int64_t)p.M * p.k; ++t) km_cnorm(p, t);
    for (int64_t t = 0; t < p.n * p.M; ++t) km_assign_aniso(p, t);
    for (int64_t m = 0; m < p.M; ++m) km_count_changed(p, m);
    return JV_OK;
}
int launch_km_reactivate(hipStream_t, const KmParams &p)
{
    for (int64_t m = 0; m < p.M; ++m) km_reactivate(p, m);
    return JV_OK;
}
int launch_km_finish_round(hipStream_t, const KmParams &p)
{
    for (int64_t m = 0; m < p.M; ++m) km_finish_round(p, m);
    return JV_OK;
}

// ---- batched graph construction: bl_body.h's per-item bodies as loops (in REVERSE item order where the GPU's order is
//      unspecified, so that nothing silently depends on ascending execution) ----
int launch_bl_apply_selection(hipStream_t, const BlApplyParams &p)
{
    for (long long i = (long long)p.B * p.Rf - 1; i >= 0; --i) bl_apply_selection(p, i);
    for (long long b = p.B - 1; b >= 0; --b) bl_pack_row(p, b);
    return JV_OK;
}
int launch_bl_improve_list(hipStream_t, const BlImproveParams &p)
{
    for (long long b = p.B - 1; b >= 0; --b) bl_improve_list(p, b);
    return JV_OK;
}
int launch_bl_row_edges(hipStream_t, const BlRowEdgesParams &p)
{
    for (long long i = (long long)p.B * p.Rf - 1; i >= 0; --i) bl_row_edges(p, i);
    return JV_OK;
}
int launch_bl_backlink_merge(hipStream_t, const BlMergeParams &p)
{
    for (long long i = p.E - 1; i >= 0; --i) bl_backlink_merge(p, i);
    return JV_OK;
}
int launch_bl_rank_sort(hipStream_t, const BlSortParams &p)
{
    for (long long i = (long long)p.P * p.L - 1; i >= 0; --i) bl_rank_sort(p, i);
    return JV_OK;
}
int launch_bl_rewrite_rows(hipStream_t, const BlRowsParams &p)
{
    for (long long i = p.P - 1; i >= 0; --i) bl_rewrite_row(p, i);
    return JV_OK;
}
int launch_bl_list_over_degree(hipStream_t, const BlOverParams &p)
{
    for (long long i = p.N - 1; i >= 0; --i) bl_list_over_degree(p, i);
    return JV_OK;
}
int launch_bl_ro_apply_selection(hipStream_t, const BlRoApplyParams &p)
{
    for (long long b = p.B - 1; b >= 0; --b) bl_ro_apply_selection(p, b);
    return JV_OK;
}
int launch_bl_ro_backlink_merge(hipStream_t, const BlRoMergeParams &p)
{
    for (long long i = p.E - 1; i >= 0; --i) bl_ro_backlink_merge(p, i);
    return JV_OK;
}
int launch_bl_ro_rewrite_rows(hipStream_t, const BlRoRowsParams &p)
{
    for (long long i = p.P - 1; i >= 0; --i) bl_ro_rewrite_row(p, i);
    return JV_OK;
}
int launch_bl_ro_copy_rows(hipStream_t, const BlRoCopyParams &p)
{
    for (long long i = p.P - 1; i >= 0; --i) bl_ro_copy_row(p, i);
    return JV_OK;
}
int launch_bl_sel_ids(hipStream_t, const BlSelIdsParams &p)
{
    for (long long i = (long long)p.B * p.Rf - 1; i >= 0; --i) bl_sel_ids(p, i);
    return JV_OK;
}
int launch_bl_ro_apply_sorted(hipStream_t, const BlRoApplySortedParams &p)
{
    for (long long b = p.B - 1; b >= 0; --b) bl_ro_apply_sorted(p, b);
    return JV_OK;
}
int launch_bl_ro_improve_list(hipStream_t, const BlRoImproveParams &p)
{
    for (long long b = p.B - 1; b >= 0; --b) bl_ro_improve_list(p, b);
    return JV_OK;
}
int launch_bl_ro_row_edges(hipStream_t, const BlRoRowEdgesParams &p)
{
    for (long long i = (long long)p.B * p.Rf - 1; i >= 0; --i) bl_ro_row_edges(p, i);
    return JV_OK;
}
int launch_bl_count_valid(hipStream_t, const int32_t *cand, int C, int32_t *count, long long B)
{
    for (long long b = 0; b < B; ++b) bl_count_valid(cand, C, count, b);
    return JV_OK;
}
int launch_bl_copy_rows(hipStream_t, const int32_t *nbrs, int R, const int32_t *tgt, long long P, int32_t *out)
{
    for (long long i = 0; i < P * R; ++i) out[i] = nbrs[(long long)tgt[i / R] * R + (i % R)];
    return JV_OK;
}
int launch_bl_strided_copy(hipStream_t, const int32_t *src, int R, int Rf, long long N, int32_t *dst)
{
    for (long long i = 0; i < N * Rf; ++i) dst[i] = src[(i / Rf) * R + (i % Rf)];
    return JV_OK;
}
int launch_bl_sort_edges(hipStream_t, void *temp, size_t *temp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out,
                         const int32_t *vals_in, int32_t *vals_out, long long n, int)
{
    if (!temp) {
        *temp_bytes = 64;
        return JV_OK;
    }
    std::vector<long long> idx((size_t)n);
    for (long long i = 0; i < n; ++i) idx[(size_t)i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](long long a, long long b) { return keys_in[a] < keys_in[b]; });
    for (long long i = 0; i < n; ++i) {
        keys_out[i] = keys_in[idx[(size_t)i]];
        vals_out[i] = vals_in[idx[(size_t)i]];
    }
    return JV_OK;
}

// ---- device-resident traversal: gs_body.h on the lane emulator ----
bool graph_search_session_supported(int M) { return M >= 1; }
bool graph_search_device_specialised(const jv_pq *pq, const jv_codes *codes, const jv_fused *fused)
{
    const int ch = pq->M / 16;  // the same predicates as k_gsearch.hip
    return pq->uniform && pq->max_size == 8 && pq->k == kClusters && pq->M % 16 == 0 &&
           (ch == 1 || ch == 2 || ch == 3 || ch == 4 || ch == 6 || ch == 8 || ch == 12) &&
           pq->D == 8 * pq->M && (reinterpret_cast<uintptr_t>(codes->d_codes) & 15) == 0 &&
           (!fused || (reinterpret_cast<uintptr_t>(fused->d_blocks) & 15) == 0);
}
bool graph_search_device_supported(const jv_pq *pq, const jv_codes *, const jv_fused *, int max_degree, int n_levels)
{
    return pq->k == kClusters && pq->M >= 1 && max_degree <= kMaxGraphDegree && n_levels <= GS_MAX_LEVELS;
}
size_t graph_search_lds_bytes(int D, int rerankK, int cand_cap, int pair_M, int evict_cap, int v1_log2)
{
    return gs_lds_bytes(D, rerankK, cand_cap, pair_M, evict_cap > 0 ? evict_cap : GS_EVICT_CAP, v1_log2);
}
namespace {
struct GsLaunch {
    const GsParams *p;
    int vsf, worker;
    char *lds;
};
template <int VSF>
void gs_run_pairc(const GsLaunch &L)
{
    switch (L.p->M / 16) {
    case 1: gs_worker<VSF, 1, false, false, false, true>(*L.p, L.worker, L.lds); break;
    case 2: gs_worker<VSF, 2, false, false, false, true>(*L.p, L.worker, L.lds); break;
    case 3: gs_worker<VSF, 3, false, false, false, true>(*L.p, L.worker, L.lds); break;
    case 4: gs_worker<VSF, 4, false, false, false, true>(*L.p, L.worker, L.lds); break;
    case 6: gs_worker<VSF, 6, false, false, false, true>(*L.p, L.worker, L.lds); break;
    case 8: gs_worker<VSF, 8, false, false, false, true>(*L.p, L.worker, L.lds); break;
    case 12: gs_worker<VSF, 12, false, false, false, true>(*L.p, L.worker, L.lds); break;
    default: abort();
    }
}
template <int VSF, bool PAIR>
void gs_run_session(const GsLaunch &L)
{
    if (L.p->generic) {
        if constexpr (!PAIR) gs_worker<VSF, 0, false, false, true>(*L.p, L.worker, L.lds);
        else abort();
        return;
    }
    switch (L.p->M / 16) {
    case 1: gs_worker<VSF, 1, PAIR, false, true>(*L.p, L.worker, L.lds); break;
    case 2: gs_worker<VSF, 2, PAIR, false, true>(*L.p, L.worker, L.lds); break;
    case 3: gs_worker<VSF, 3, PAIR, false, true>(*L.p, L.worker, L.lds); break;
    case 4: gs_worker<VSF, 4, PAIR, false, true>(*L.p, L.worker, L.lds); break;
    case 6: gs_worker<VSF, 6, PAIR, false, true>(*L.p, L.worker, L.lds); break;
    case 8: gs_worker<VSF, 8, PAIR, false, true>(*L.p, L.worker, L.lds); break;
    case 12: gs_worker<VSF, 12, PAIR, false, true>(*L.p, L.worker, L.lds); break;
    default: abort();
    }
}
template <int VSF, bool PAIR>
void gs_run_ch(const GsLaunch &L)
{
    if (L.p->generic) {
        if constexpr (!PAIR) gs_worker<VSF, 0, false>(*L.p, L.worker, L.lds);
        else abort();
        return;
    }
    switch (L.p->M / 16) {
    case 1: gs_worker<VSF, 1, PAIR>(*L.p, L.worker, L.lds); break;
    case 2: gs_worker<VSF, 2, PAIR>(*L.p, L.worker, L.lds); break;
    case 3: gs_worker<VSF, 3, PAIR>(*L.p, L.worker, L.lds); break;
    case 4: gs_worker<VSF, 4, PAIR>(*L.p, L.worker, L.lds); break;
    case 6: gs_worker<VSF, 6, PAIR>(*L.p, L.worker, L.lds); break;
    case 8: gs_worker<VSF, 8, PAIR>(*L.p, L.worker, L.lds); break;
    case 12: gs_worker<VSF, 12, PAIR>(*L.p, L.worker, L.lds); break;
    default: abort();
    }
}
template <bool PAIR>
void gs_run_vsf(const GsLaunch &L)
{
    if (L.vsf == VSF_L2) gs_run_ch<VSF_L2, PAIR>(L);
    else if (L.vsf == VSF_DOT) gs_run_ch<VSF_DOT, PAIR>(L);
    else gs_run_ch<VSF_COS, PAIR>(L);
}
void gs_main(void *a)
{
    const GsLaunch &L = *(const GsLaunch *)a;
    if (L.p->session) {
        if (L.p->pair) {
            if (L.vsf == VSF_L2) gs_run_session<VSF_L2, true>(L);
            else if (L.vsf == VSF_DOT) gs_run_session<VSF_DOT, true>(L);
            else gs_run_session<VSF_COS, true>(L);
        } else {
            if (L.vsf == VSF_L2) gs_run_session<VSF_L2, false>(L);
            else if (L.vsf == VSF_DOT) gs_run_session<VSF_DOT, false>(L);
            else gs_run_session<VSF_COS, false>(L);
        }
    } else if (L.p->ubr && L.p->pair == 2) {
        if (L.vsf == VSF_L2) gs_worker<VSF_L2, 6, false, false, false, true, true>(*L.p, L.worker, L.lds);
        else if (L.vsf == VSF_DOT) gs_worker<VSF_DOT, 6, false, false, false, true, true>(*L.p, L.worker, L.lds);
        else gs_worker<VSF_COS, 6, false, false, false, true, true>(*L.p, L.worker, L.lds);
    } else if (L.p->ubr) {
        if (L.vsf == VSF_L2) gs_worker<VSF_L2, 6, true, false, false, false, true>(*L.p, L.worker, L.lds);
        else if (L.vsf == VSF_DOT) gs_worker<VSF_DOT, 6, true, false, false, false, true>(*L.p, L.worker, L.lds);
        else gs_worker<VSF_COS, 6, true, false, false, false, true>(*L.p, L.worker, L.lds);
    } else if (L.p->pair == 2) {
        if (L.vsf == VSF_L2) gs_run_pairc<VSF_L2>(L);
        else if (L.vsf == VSF_DOT) gs_run_pairc<VSF_DOT>(L);
        else gs_run_pairc<VSF_COS>(L);
    } else if (L.p->pair) gs_run_vsf<true>(L);
    else gs_run_vsf<false>(L);
}
}  // namespace
// ---- the workgroup form (gx_body.h): up to emu::MAX_WAVES waves per workgroup on the emulator ----
bool graph_search_wgx_supported(int M)
{
    const int ch = M / 16;
    return M % 16 == 0 && (ch == 1 || ch == 2 || ch == 3 || ch == 4 || ch == 6 || ch == 8 || ch == 12);
}
size_t graph_search_wgx_lds_bytes(int D, int rerankK, int cand_cap, int evict_cap, int v1_log2, int slots, int kps, int logcap, int M)
{
    return gx_lds_bytes(D, rerankK, cand_cap, evict_cap > 0 ? evict_cap : GS_EVICT_CAP, v1_log2, slots, kps, logcap, M);
}
namespace {
template <int VSF>
void gx_run_ch(const GsLaunch &L)
{
    switch (L.p->M / 16) {
    case 1: gx_worker<VSF, 1>(*L.p, L.worker, L.lds); break;
    case 2: gx_worker<VSF, 2>(*L.p, L.worker, L.lds); break;
    case 3: gx_worker<VSF, 3>(*L.p, L.worker, L.lds); break;
    case 4: gx_worker<VSF, 4>(*L.p, L.worker, L.lds); break;
    case 6: gx_worker<VSF, 6>(*L.p, L.worker, L.lds); break;
    case 8: gx_worker<VSF, 8>(*L.p, L.worker, L.lds); break;
    case 12: gx_worker<VSF, 12>(*L.p, L.worker, L.lds); break;
    default: abort();
    }
}
void gx_main(void *a)
{
    const GsLaunch &L = *(const GsLaunch *)a;
    if (L.vsf == VSF_L2) gx_run_ch<VSF_L2>(L);
    else if (L.vsf == VSF_DOT) gx_run_ch<VSF_DOT>(L);
    else gx_run_ch<VSF_COS>(L);
}
}  // namespace
int launch_graph_search_wgx(hipStream_t, int vsf, const GsParams &p, int workgroups, int threads)
{
    if (p.Q == 0) return JV_OK;
    const int waves = std::min(emu::MAX_WAVES, std::max(2, threads / 64));   // (the emulator runs up to 4 waves per block)
    const size_t lds_bytes = gx_lds_bytes(p.D, p.rerankK, p.cand_cap, p.evict_cap > 0 ? p.evict_cap : GS_EVICT_CAP, p.v1_log2, p.wgx_slots, p.wgx_kps, p.wgx_log, p.wgx_lut_m);
    for (int w = 0; w < workgroups; ++w) {
        GsParams pw = p;
        pw.Q = (int)((long long)p.Q * (w + 1) / workgroups);
        char *lds = (char *)aligned_alloc(64, lds_bytes + 64);
        memset(lds, 0xa5, lds_bytes);
        GsLaunch L{&pw, vsf, w, lds};
        run_block_locked(gx_main, &L, waves);
        *p.next_query = (uint32_t)pw.Q;
        free(lds);
    }
    return JV_OK;
}

// the register-table bound form: the tables by gs_host.h's restatement of ubr_table_kernel, the traversal on the lane emulator
bool graph_search_ubr_supported(int M, int /*vsf*/) { return M == 96; }
int launch_ubr_tables(hipStream_t, int vsf, const float *codebooks, const float *cq, int Q, int M, uint32_t *tab, float *meta)
{
    if (M % 8 != 0) return JV_ERR_INVALID;
    for (int q = 0; q < Q; ++q) gs_ubr_build_ref(codebooks, cq + (size_t)q * 8 * M, M, tab + (size_t)q * M * 64, meta + (size_t)q * 4, vsf == VSF_L2);
    return JV_OK;
}
int launch_graph_search(hipStream_t, int vsf, const GsParams &p, int workers, int /*occupancy*/);
int launch_graph_search_ubr(hipStream_t s, int vsf, const GsParams &p, int workers, size_t /*lds*/) { return launch_graph_search(s, vsf, p, workers, 2); }

int launch_graph_search(hipStream_t, int vsf, const GsParams &p, int workers, int /*occupancy*/)
{
    if (p.Q == 0) return JV_OK;
    const size_t lds_bytes = gs_lds_bytes(p.D, p.rerankK, p.cand_cap, p.pair ? p.M : 0, p.evict_cap > 0 ? p.evict_cap : GS_EVICT_CAP, p.v1_log2) +
                             (p.session ? gs_session_lds_bytes() : 0);
    // the waves of a persistent launch, one after another; wave w stops after its share so that several workers'
    // scratch slices are exercised (a real launch interleaves them).  The lane emulator keeps one wave's context in statics:
    // launches from several host threads (one context each) take turns (run_wave_locked).
    for (int w = 0; w < workers; ++w) {
        GsParams pw = p;
        pw.Q = (int)((long long)p.Q * (w + 1) / workers);
        char *lds = (char *)aligned_alloc(64, lds_bytes + 64);
        memset(lds, 0xa5, lds_bytes);
        GsLaunch L{&pw, vsf, w, lds};
        run_wave_locked(gs_main, &L);
        *p.next_query = (uint32_t)pw.Q;
        free(lds);
    }
    return JV_OK;
}

}  // namespace jv
