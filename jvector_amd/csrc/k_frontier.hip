// k_frontier.hip — frontier scoring for the host batched graph searcher (graph_search.cpp).
//
// One launch scores the current round of S traversal SLOTS.  A slot is either
//   * expanding a layer-0 node with FusedPQ: its candidates are the origin's packed neighbour block
//     (FusedPQDecoder.similarityToNeighbor, B/quantization/FusedPQDecoder.java:104-111,206-213), or
//   * expanding a node of an upper layer / a non-fused index: its candidates are the listed ordinals, scored from
//     the PQVectors code store (PQDecoder.similarityTo, B/quantization/PQDecoder.java:65-80,124-135), or
//   * idle (slot_query < 0).
// slot_query maps the slot to the query whose look-up table it uses, so queries can stream through a fixed set of
// slots (continuous batching) while all tables are built once up front.
//
// One lane = one candidate; the table stays in L2/MALL and is gathered through the vector-memory path (a slot has
// <= maxDegree candidates: staging 96 KB in LDS for 32 rows would cost 30x the useful traffic).  Sums are formed in
// ascending m into one f32 — bit-identical to DefaultVectorUtilSupport.assembleAndSum / pqDecodedCosineSimilarity.
#include <cstdlib>

#include "jv_device.h"
#include "jv_internal.h"

namespace jv {

struct FrontierParams {
    const float *luts;            // [nQ][M*256]
    const float *bmag;            // [nQ] (cosine)
    const int32_t *slot_query;    // [S]
    const int32_t *origins;       // [S] layer-0 fused origin node, or -1
    const int32_t *ord_index;     // [S] row of `ords` used by the slot (gather mode), or -1
    const int32_t *ords;          // [n_gather][W] ordinals, -1 padded
    const uint8_t *blocks;        // fused: [n_nodes][maxDegree*M]   (nullable)
    const int32_t *fused_nbrs;    // fused: [n_nodes][maxDegree]
    const float *fused_norms;     // fused: [n_nodes][maxDegree] (cosine)
    const uint8_t *codes;         // [n_codes][M]
    const float *code_norms;      // [n_codes] (cosine)
    float *out;                   // [S][W]
    const float *codebooks;       // table-free mode: [M][256][8] f32 (uniform 8-dim sub-vectors)
    const float *cq;              // table-free mode: centred queries [nQ][D]
    int D;
    int64_t n_nodes, n_codes;
    int S, W, M, maxDegree;
};

template <int CH16>
__device__ __forceinline__ float frontier_row_sum(const float *lut, const uint8_t *row, int M)
{
    float sum = 0.0f;
    if (CH16 > 0) {
        const uint4 *r4 = reinterpret_cast<const uint4 *>(row);
        uint4 w[CH16 > 0 ? CH16 : 1];
#pragma unroll
        for (int c = 0; c < CH16; ++c) w[c] = r4[c];
#pragma unroll
        for (int c = 0; c < CH16; ++c) {
            const uint32_t d[4] = {w[c].x, w[c].y, w[c].z, w[c].w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    sum += lut[(c * 16 + e * 4 + b) * kClusters + ((d[e] >> (8 * b)) & 0xFFu)];
        }
    } else {
        for (int m = 0; m < M; ++m) sum += lut[m * kClusters + row[m]];
    }
    return sum;
}

// block = 64 lanes.  W <= 32: two slots per wave (lane>>5 picks the slot); W <= 64: one slot per wave; wider rows: blockIdx.y
// picks the 64-neighbour chunk.
template <int VSF, int CH16, bool TWO>
__global__ __launch_bounds__(64) void frontier_kernel(FrontierParams p)
{
    const int lane = threadIdx.x;
    const int slot = TWO ? (blockIdx.x * 2 + (lane >> 5)) : blockIdx.x;
    const int i = TWO ? (lane & 31) : (int)blockIdx.y * 64 + lane;
    if (slot >= p.S || i >= p.W) return;
    float *o = p.out + (int64_t)slot * p.W + i;
    const int lq = p.slot_query[slot];
    if (lq < 0) {
        *o = -INFINITY;
        return;
    }
    const int64_t origin = p.origins ? p.origins[slot] : -1;
    const uint8_t *rp;
    float nrm = 0.0f;
    if (origin >= 0) {
        if (i >= p.maxDegree || origin >= p.n_nodes) { *o = -INFINITY; return; }
        const int64_t row = origin * p.maxDegree + i;
        if (p.fused_nbrs[row] < 0) { *o = -INFINITY; return; }
        rp = p.blocks + row * p.M;
        if (VSF == VSF_COS) nrm = p.fused_norms[row];
    } else {
        const int oi = p.ord_index[slot];
        if (oi < 0) { *o = -INFINITY; return; }
        const int64_t ord = p.ords[(int64_t)oi * p.W + i];
        if (ord < 0 || ord >= p.n_codes) { *o = -INFINITY; return; }
        rp = p.codes + ord * p.M;
        if (VSF == VSF_COS) nrm = p.code_norms[ord];
    }
    const float *lut = p.luts + (int64_t)lq * p.M * kClusters;
    float sum = frontier_row_sum<CH16>(lut, rp, p.M);
    if (VSF == VSF_COS) sum = score_from_raw(VSF_COS, cosine_finish(sum, nrm, p.bmag[lq]));
    else sum = score_from_raw(VSF, sum);
    *o = sum;
}

template <int VSF, bool TWO>
static int launch_frontier_ch(hipStream_t s, const FrontierParams &p, int ch)
{
    dim3 grid(TWO ? (p.S + 1) / 2 : p.S, TWO ? 1 : (p.W + 63) / 64), block(64);
#define JV_FR(CH) hipLaunchKernelGGL((frontier_kernel<VSF, CH, TWO>), grid, block, 0, s, p)
    switch (ch) {
    case 1: JV_FR(1); break;
    case 2: JV_FR(2); break;
    case 3: JV_FR(3); break;
    case 4: JV_FR(4); break;
    case 6: JV_FR(6); break;
    case 8: JV_FR(8); break;
    case 12: JV_FR(12); break;
    default: JV_FR(0); break;
    }
#undef JV_FR
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

// ------------------------------------------------------------------------------------------------
// Table-free variant (uniform 8-dim sub-vectors): instead of gathering the query's 96 KB look-up table from
// L2/HBM (32 rows touch ~85 % of its sectors: PMC showed 146 MB of HBM traffic per 2048-slot launch against
// 6.5 MB algorithmic), each lane RECOMPUTES the table entry it needs,
//     entry(m, c) = sum_j codebook[m][c][j] (*|-) cq[m*8 + j]      j ascending, non-fused,
// exactly the arithmetic of lut_build_kernel / DefaultVectorUtilSupport.calculatePartialSums (:351-365), so the
// entry — and therefore the m-ascending running sum — is bit-identical to the table path.  The codebook
// (M x 8 KB = 768 KB) is shared by every query and stays L2 resident; the slot's centred query (3 KB) is staged
// in LDS and read as broadcasts.
// ------------------------------------------------------------------------------------------------
template <int VSF, int CH16, bool TWO>
__global__ __launch_bounds__(64) void frontier_direct_kernel(FrontierParams p)
{
    extern __shared__ __attribute__((aligned(16))) float qlds[];  // (TWO ? 2 : 1) x D
    const int lane = threadIdx.x;
    const int half = TWO ? (lane >> 5) : 0;
    const int slot = TWO ? (blockIdx.x * 2 + half) : blockIdx.x;
    const int li = TWO ? (lane & 31) : lane;
    const int lanes_per_slot = TWO ? 32 : 64;
    const int lq = slot < p.S ? p.slot_query[slot] : -1;
    float *qs = qlds + half * p.D;
    if (lq >= 0) {
        const float4 *src = reinterpret_cast<const float4 *>(p.cq + (int64_t)lq * p.D);
        float4 *dst = reinterpret_cast<float4 *>(qs);
        for (int j = li; j < p.D / 4; j += lanes_per_slot) dst[j] = src[j];
    }
    __syncthreads();
    const int i = TWO ? li : (int)blockIdx.y * 64 + li;  // rows wider than 64: blockIdx.y picks the chunk
    if (slot >= p.S || i >= p.W) return;
    float *o = p.out + (int64_t)slot * p.W + i;
    if (lq < 0) { *o = -INFINITY; return; }
    const int64_t origin = p.origins ? p.origins[slot] : -1;
    const uint8_t *rp;
    float nrm = 0.0f;
    if (origin >= 0) {
        if (i >= p.maxDegree || origin >= p.n_nodes) { *o = -INFINITY; return; }
        const int64_t row = origin * p.maxDegree + i;
        if (p.fused_nbrs[row] < 0) { *o = -INFINITY; return; }
        rp = p.blocks + row * p.M;
        if (VSF == VSF_COS) nrm = p.fused_norms[row];
    } else {
        const int oi = p.ord_index[slot];
        if (oi < 0) { *o = -INFINITY; return; }
        const int64_t ord = p.ords[(int64_t)oi * p.W + i];
        if (ord < 0 || ord >= p.n_codes) { *o = -INFINITY; return; }
        rp = p.codes + ord * p.M;
        if (VSF == VSF_COS) nrm = p.code_norms[ord];
    }
    const uint4 *r4 = reinterpret_cast<const uint4 *>(rp);
    uint4 w[CH16];
#pragma unroll
    for (int c = 0; c < CH16; ++c) w[c] = r4[c];
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < CH16; ++c) {
        const uint32_t d[4] = {w[c].x, w[c].y, w[c].z, w[c].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int m = c * 16 + e * 4 + b;
                const uint32_t code = (d[e] >> (8 * b)) & 0xFFu;
                const float4 *cp = reinterpret_cast<const float4 *>(p.codebooks + ((int64_t)(m * kClusters) + code) * 8);
                const float4 c0 = cp[0], c1 = cp[1];
                const float *q = qs + m * 8;
                float ent = 0.0f;
                if (VSF == VSF_L2) {
                    float t;
                    t = c0.x - q[0]; ent += t * t;
                    t = c0.y - q[1]; ent += t * t;
                    t = c0.z - q[2]; ent += t * t;
                    t = c0.w - q[3]; ent += t * t;
                    t = c1.x - q[4]; ent += t * t;
                    t = c1.y - q[5]; ent += t * t;
                    t = c1.z - q[6]; ent += t * t;
                    t = c1.w - q[7]; ent += t * t;
                } else {
                    ent += c0.x * q[0];
                    ent += c0.y * q[1];
                    ent += c0.z * q[2];
                    ent += c0.w * q[3];
                    ent += c1.x * q[4];
                    ent += c1.y * q[5];
                    ent += c1.z * q[6];
                    ent += c1.w * q[7];
                }
                sum += ent;
            }
        }
    }
    if (VSF == VSF_COS) sum = score_from_raw(VSF_COS, cosine_finish(sum, nrm, p.bmag[lq]));
    else sum = score_from_raw(VSF, sum);
    *o = sum;
}

template <int VSF, bool TWO>
static int launch_frontier_direct_ch(hipStream_t s, const FrontierParams &p, int ch)
{
    dim3 grid(TWO ? (p.S + 1) / 2 : p.S, TWO ? 1 : (p.W + 63) / 64), block(64);
    const size_t lds = sizeof(float) * (size_t)p.D * (TWO ? 2 : 1);
#define JV_FD(CH) hipLaunchKernelGGL((frontier_direct_kernel<VSF, CH, TWO>), grid, block, lds, s, p)
    switch (ch) {
    case 1: JV_FD(1); break;
    case 2: JV_FD(2); break;
    case 3: JV_FD(3); break;
    case 4: JV_FD(4); break;
    case 6: JV_FD(6); break;
    case 8: JV_FD(8); break;
    case 12: JV_FD(12); break;
    default: return JV_ERR_UNSUPPORTED;
    }
#undef JV_FD
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

int launch_frontier(hipStream_t s, int vsf, const float *d_luts, const float *d_bmag, const int32_t *d_slot_query,
                    const int32_t *d_origins, const int32_t *d_ord_index, const int32_t *d_ords, const jv_fused *fused,
                    const jv_codes *codes, float *d_out, int S, int W, const jv_pq *pq, const float *d_cq)
{
    if (S == 0 || W == 0) return JV_OK;
    if (W > kMaxGraphDegree) {
        set_error("frontier: degree %d > %d is not supported", W, kMaxGraphDegree);
        return JV_ERR_UNSUPPORTED;
    }
    FrontierParams p{};
    p.luts = d_luts; p.bmag = d_bmag; p.slot_query = d_slot_query; p.origins = fused ? d_origins : nullptr; p.ord_index = d_ord_index; p.ords = d_ords;
    if (fused) {
        p.blocks = fused->d_blocks; p.fused_nbrs = fused->d_neighbors; p.fused_norms = fused->d_norms;
        p.maxDegree = fused->maxDegree; p.n_nodes = fused->count;
    }
    p.codes = codes->d_codes; p.code_norms = codes->d_norms; p.n_codes = codes->count;
    p.out = d_out; p.S = S; p.W = W; p.M = codes->M;
    const bool aligned = (p.M % 16 == 0) && ((reinterpret_cast<uintptr_t>(p.codes) & 15) == 0) &&
                         (!fused || (reinterpret_cast<uintptr_t>(p.blocks) & 15) == 0);
    const int ch = aligned ? p.M / 16 : 0;
    const bool two = W <= 32;
    // table-free path: uniform 8-dim sub-vectors, 16-byte aligned rows, D*8 B of LDS
    static const bool no_direct = getenv("JVECTOR_HIP_FRONTIER_TABLES") != nullptr;
    const bool direct_ok = !no_direct && pq && d_cq && pq->uniform && pq->max_size == 8 && aligned &&
                           (ch == 1 || ch == 2 || ch == 3 || ch == 4 || ch == 6 || ch == 8 || ch == 12) && pq->D % 4 == 0 &&
                           pq->D <= 8192;
    if (direct_ok) {
        p.codebooks = pq->d_codebooks;
        p.cq = d_cq;
        p.D = pq->D;
        switch (vsf) {
        case VSF_L2: return two ? launch_frontier_direct_ch<VSF_L2, true>(s, p, ch) : launch_frontier_direct_ch<VSF_L2, false>(s, p, ch);
        case VSF_DOT: return two ? launch_frontier_direct_ch<VSF_DOT, true>(s, p, ch) : launch_frontier_direct_ch<VSF_DOT, false>(s, p, ch);
        default: return two ? launch_frontier_direct_ch<VSF_COS, true>(s, p, ch) : launch_frontier_direct_ch<VSF_COS, false>(s, p, ch);
        }
    }
#define JV_FV(V)                                                    \
    do {                                                            \
        if (two) return launch_frontier_ch<V, true>(s, p, ch);      \
        return launch_frontier_ch<V, false>(s, p, ch);              \
    } while (0)
    switch (vsf) {
    case VSF_L2: JV_FV(VSF_L2);
    case VSF_DOT: JV_FV(VSF_DOT);
    default: JV_FV(VSF_COS);
    }
#undef JV_FV
}

}  // namespace jv
