// k_topk.hip — exact, deterministic top-k under the NodeQueue total order.
//
// Order: key64 = (ordered_u32(score) << 32) | (~id as u32), larger key = better — bit-for-bit the order
// of NodeQueue.encode (NodeQueue.java:125-129) over NumericUtils.floatToSortableInt (NumericUtils.java:
// 49-65): higher score first, equal scores -> smaller node id first.  Ids within a row are unique, hence
// keys are unique and "the k largest keys" is a well-defined set: no tie handling, no overflow path.
//
// Algorithm: MSB radix select on the 64-bit key in digits of 11/11/10 | 11/11/10 bits.  Each pass is a
// grid-wide histogram of the current digit over the elements matching the already-decided prefix, then a
// one-workgroup-per-query selection of the bin holding the k-th largest.  Passes stop (device-side flag,
// remaining launches exit immediately) as soon as the bin is needed in full — in practice after the three
// score digits, the id digits only run when the k-th score is tied.  A collect pass gathers the k keys and
// a one-workgroup bitonic sort orders them best-first.
#include "jv_device.h"
#include "jv_internal.h"
#include "gs_wave_hip.h"
#include "rt_body.h"

namespace jv {

constexpr int kBins = 2048;
constexpr int kPasses = 6;
constexpr int kMaxK = 8192;

struct SelState {
    unsigned long long prefix;  // decided high bits of the threshold key
    unsigned int k_rem;         // how many still to take inside the current prefix
    unsigned int k_eff;         // min(k, valid elements)
    unsigned int done;          // threshold final: select key >= prefix
    unsigned int collected;     // collect-pass cursor
    unsigned int pad[2];
};

__device__ __constant__ int c_shift[kPasses] = {53, 42, 32, 21, 10, 0};
__device__ __constant__ int c_width[kPasses] = {11, 11, 10, 11, 11, 10};

__device__ __forceinline__ bool load_key(const float *__restrict__ scores, const int32_t *__restrict__ ids,
                                         int64_t row_off, int64_t col, int32_t id_base, unsigned long long &key, int64_t row_n)
{
    if (col >= row_n) return false;
    int32_t id = ids ? ids[row_off + col] : (int32_t)(id_base + col);
    if (id < 0) return false;
    const uint32_t u = float_to_ordered_u32(scores[row_off + col]);
    key = ((unsigned long long)u << 32) | (unsigned long long)(uint32_t)(~id);
    return true;
}

// grid (nblk, Q), block 256
__global__ __launch_bounds__(256) void topk_hist_kernel(const float *__restrict__ scores,
                                                        const int32_t *__restrict__ ids, int64_t n, int64_t stride,
                                                        int32_t id_base, int pass, const SelState *__restrict__ st, const unsigned int *__restrict__ row_counts,
                                                        unsigned int *__restrict__ hist)
{
    __shared__ unsigned int lh[kBins];
    const int q = blockIdx.y;
    if (pass > 0 && st[q].done) return;
    for (int i = threadIdx.x; i < kBins; i += 256) lh[i] = 0;
    __syncthreads();
    const int shift = c_shift[pass], width = c_width[pass];
    const unsigned long long prefix = pass > 0 ? st[q].prefix : 0ull;
    const int hi = shift + width;  // bits >= hi are decided
    const int64_t row_off = (int64_t)q * stride;
    const int64_t row_n = row_counts ? (row_counts[q] < (unsigned long long)n ? (int64_t)row_counts[q] : n) : n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < row_n; i += (int64_t)gridDim.x * 256) {
        unsigned long long key;
        if (!load_key(scores, ids, row_off, i, id_base, key, row_n)) continue;
        if (hi < 64 && (key >> hi) != (prefix >> hi)) continue;
        const unsigned int bin = (unsigned int)((key >> shift) & ((1u << width) - 1u));
        atomicAdd(&lh[bin], 1u);
    }
    __syncthreads();
    unsigned int *gh = hist + ((int64_t)q * kPasses + pass) * kBins;
    for (int i = threadIdx.x; i < kBins; i += 256) {
        const unsigned int c = lh[i];
        if (c) atomicAdd(&gh[i], c);
    }
}

// grid (Q), block 256: walk the bins from the top until the cumulative count reaches k_rem.
__global__ __launch_bounds__(256) void topk_select_kernel(int pass, int k, SelState *__restrict__ st,
                                                          const unsigned int *__restrict__ hist)
{
    __shared__ unsigned int lh[kBins];
    __shared__ unsigned int chunk_sum[256];
    const int q = blockIdx.x;
    SelState s = st[q];
    if (pass > 0 && s.done) return;
    const unsigned int *gh = hist + ((int64_t)q * kPasses + pass) * kBins;
    const int nb = 1 << c_width[pass];
    // thread t owns bins [t*8, t*8+8) counted from the TOP (descending key order)
    unsigned int local = 0;
    for (int j = 0; j < kBins / 256; ++j) {
        const int pos = threadIdx.x * (kBins / 256) + j;  // 0 = top bin
        const int bin = nb - 1 - pos;
        const unsigned int c = (bin >= 0) ? gh[bin] : 0u;
        lh[pos] = c;
        local += c;
    }
    chunk_sum[threadIdx.x] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int total = 0;
        for (int t = 0; t < 256; ++t) total += chunk_sum[t];
        if (pass == 0) {
            s.prefix = 0ull;
            s.k_eff = total < (unsigned int)k ? total : (unsigned int)k;
            s.k_rem = s.k_eff;
            s.done = 0;
            s.collected = 0;
            if (total <= (unsigned int)k) {  // everything valid is selected
                s.done = 1;
                st[q] = s;
                return;
            }
        }
        // find chunk, then bin
        // (every loop is bounded: if the input is modified concurrently by another stream the histogram may
        //  be inconsistent with k_rem; we then settle on the last bin instead of spinning)
        unsigned int cum = 0;
        int t = 0;
        for (; t < 255; ++t) {
            if (cum + chunk_sum[t] >= s.k_rem) break;
            cum += chunk_sum[t];
        }
        int pos = t * (kBins / 256);
        for (; pos < kBins - 1; ++pos) {
            if (cum + lh[pos] >= s.k_rem) break;
            cum += lh[pos];
        }
        if (pos > nb - 1) pos = nb - 1;
        const int bin = nb - 1 - pos;
        s.prefix |= ((unsigned long long)bin) << c_shift[pass];
        s.k_rem = s.k_rem > cum ? s.k_rem - cum : 0u;  // still needed from inside this bin
        if (lh[pos] == s.k_rem || pass == kPasses - 1) s.done = 1;  // whole bin needed: lower bits irrelevant
        st[q] = s;
    }
}

// grid (nblk, Q): gather every key >= threshold (exactly k_eff of them)
__global__ __launch_bounds__(256) void topk_collect_kernel(const float *__restrict__ scores,
                                                           const int32_t *__restrict__ ids, int64_t n, int64_t stride,
                                                           int32_t id_base, SelState *__restrict__ st, const unsigned int *__restrict__ row_counts,
                                                           unsigned long long *__restrict__ keys, int kpad)
{
    const int q = blockIdx.y;
    const unsigned long long thr = st[q].prefix;
    const unsigned int k_eff = st[q].k_eff;
    const int64_t row_off = (int64_t)q * stride;
    const int64_t row_n = row_counts ? (row_counts[q] < (unsigned long long)n ? (int64_t)row_counts[q] : n) : n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < row_n; i += (int64_t)gridDim.x * 256) {
        unsigned long long key;
        if (!load_key(scores, ids, row_off, i, id_base, key, row_n)) continue;
        if (key >= thr) {
            const unsigned int pos = atomicAdd(&st[q].collected, 1u);
            if (pos < k_eff) keys[(int64_t)q * kpad + pos] = key;
        }
    }
}

// grid (Q), block 1024: bitonic sort (descending) of kpad keys in LDS, emit ids / scores best first
__global__ __launch_bounds__(1024) void topk_sort_kernel(const SelState *__restrict__ st,
                                                         const unsigned long long *__restrict__ keys, int kpad, int k,
                                                         int32_t *__restrict__ out_ids, float *__restrict__ out_scores)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];
    const int q = blockIdx.x;
    const unsigned int k_eff = st[q].k_eff;
    for (int i = threadIdx.x; i < kpad; i += 1024) sk[i] = (i < (int)k_eff) ? keys[(int64_t)q * kpad + i] : 0ull;
    __syncthreads();
    for (int size = 2; size <= kpad; size <<= 1) {
        for (int strd = size >> 1; strd > 0; strd >>= 1) {
            for (int i = threadIdx.x; i < kpad / 2; i += 1024) {
                const int lo = (i / strd) * (strd << 1) + (i % strd);
                const int hi = lo + strd;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = sk[lo], b = sk[hi];
                if ((a < b) == desc) {
                    sk[lo] = b;
                    sk[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < k; i += 1024) {
        if (i < (int)k_eff) {
            const unsigned long long key = sk[i];
            out_ids[(int64_t)q * k + i] = (int32_t)(~(uint32_t)(key & 0xFFFFFFFFull));
            out_scores[(int64_t)q * k + i] = ordered_u32_to_float((uint32_t)(key >> 32));
        } else {
            out_ids[(int64_t)q * k + i] = -1;
            out_scores[(int64_t)q * k + i] = -INFINITY;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Short rows (n <= 4096, k <= 64 — the rerank's "top 10 of rerankK"): one wavefront per row keeps the keys in registers
// (NPL per lane, coalesced loads) and extracts the k best one at a time: lane-local max, 6-step wave max, the owning lane
// retires its key.  Keys are unique, 0 = empty.  65 536 rows x 110 -> top 10 took 1.9 ms through the six-pass radix
// select above (built for rows of millions); this form is launch-latency sized.
// ------------------------------------------------------------------------------------------------
template <int NPL>
__global__ __launch_bounds__(256) void topk_small_kernel(const float *__restrict__ scores, const int32_t *__restrict__ ids, int n,
                                                         int64_t stride, int32_t id_base, const unsigned int *__restrict__ row_counts,
                                                         int k, int Q, int32_t *__restrict__ out_ids, float *__restrict__ out_scores)
{
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= Q) return;
    const int64_t row_off = (int64_t)q * stride;
    const int64_t row_n = row_counts ? (row_counts[q] < (unsigned int)n ? (int64_t)row_counts[q] : (int64_t)n) : (int64_t)n;
    unsigned long long key[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        unsigned long long kk = 0ull;
        if (!load_key(scores, ids, row_off, (int64_t)j * 64 + lane, id_base, kk, row_n)) kk = 0ull;
        key[j] = kk;
    }
    unsigned long long mine = 0ull;
    for (int r = 0; r < k; ++r) {
        unsigned long long m = key[0];
#pragma unroll
        for (int j = 1; j < NPL; ++j) m = key[j] > m ? key[j] : m;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long t = (unsigned long long)__shfl_xor((long long)m, o, 64);
            m = t > m ? t : m;
        }
        if (m != 0ull) {
#pragma unroll
            for (int j = 0; j < NPL; ++j)
                if (key[j] == m) key[j] = 0ull;
        }
        if (lane == r) mine = m;  // k <= 64: round r's winner is written by lane r
    }
    if (lane < k) {
        const bool have = mine != 0ull;
        out_ids[(int64_t)q * k + lane] = have ? (int32_t)(~(uint32_t)(mine & 0xFFFFFFFFull)) : -1;
        out_scores[(int64_t)q * k + lane] = have ? ordered_u32_to_float((uint32_t)(mine >> 32)) : -INFINITY;
    }
}

static bool launch_topk_small(hipStream_t s, const float *d_scores, const int32_t *d_ids, int Q, int64_t n, int64_t stride,
                              int32_t id_base, int k, int32_t *d_out_ids, float *d_out_scores, const unsigned int *d_row_counts)
{
    if (n > 4096 || k > 64 || getenv("JVECTOR_HIP_TOPK_RADIX")) return false;
    const dim3 grid((unsigned)((Q + 3) / 4)), block(256);
#define JV_TS(NPL)                                                                                                        \
    hipLaunchKernelGGL(topk_small_kernel<NPL>, grid, block, 0, s, d_scores, d_ids, (int)n, stride, id_base, d_row_counts, k, Q, \
                       d_out_ids, d_out_scores)
    if (n <= 128) JV_TS(2);
    else if (n <= 256) JV_TS(4);
    else if (n <= 512) JV_TS(8);
    else if (n <= 1024) JV_TS(16);
    else if (n <= 2048) JV_TS(32);
    else JV_TS(64);
#undef JV_TS
    return true;
}

// ------------------------------------------------------------------------------------------------
// Medium rows (n <= 8192, any k): one workgroup per row loads every key into LDS, sorts them all (bitonic, descending) and
// emits the first k — ONE launch where the radix select above needs ten (six digit passes of histogram + selection, collect,
// sort: sized for rows of millions).  The flat search's "top rerankK of the filtered survivors" is this shape (1024 rows of
// ~6 k survivors, k = 3200: 0.93 ms through the radix select).  Keys are unique, 0 = empty (sorts last).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void topk_sortall_kernel(const float *__restrict__ scores, const int32_t *__restrict__ ids, int n,
                                                            int npad, int64_t stride, int32_t id_base,
                                                            const unsigned int *__restrict__ row_counts, int k,
                                                            int32_t *__restrict__ out_ids, float *__restrict__ out_scores)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];
    const int q = blockIdx.x;
    const int64_t row_off = (int64_t)q * stride;
    const int64_t row_n = row_counts ? (row_counts[q] < (unsigned int)n ? (int64_t)row_counts[q] : (int64_t)n) : (int64_t)n;
    for (int i = threadIdx.x; i < npad; i += 1024) {
        unsigned long long key = 0ull;
        if (!load_key(scores, ids, row_off, i, id_base, key, row_n)) key = 0ull;
        sk[i] = key;
    }
    __syncthreads();
    for (int size = 2; size <= npad; size <<= 1) {
        for (int strd = size >> 1; strd > 0; strd >>= 1) {
            for (int i = threadIdx.x; i < npad / 2; i += 1024) {
                const int lo = (i / strd) * (strd << 1) + (i % strd);
                const int hi = lo + strd;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = sk[lo], b = sk[hi];
                if ((a < b) == desc) {
                    sk[lo] = b;
                    sk[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < k; i += 1024) {
        const unsigned long long key = i < npad ? sk[i] : 0ull;
        const bool have = key != 0ull;
        out_ids[(int64_t)q * k + i] = have ? (int32_t)(~(uint32_t)(key & 0xFFFFFFFFull)) : -1;
        out_scores[(int64_t)q * k + i] = have ? ordered_u32_to_float((uint32_t)(key >> 32)) : -INFINITY;
    }
}

static int next_pow2(int v)
{
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// scratch layout: [SelState Q] [hist Q*6*2048 u32] [keys Q*kpad u64]
size_t topk_scratch_bytes(int Q, int k)
{
    const size_t kpad = (size_t)next_pow2(k < 2 ? 2 : k);
    size_t b = 0;
    b += ((sizeof(SelState) * (size_t)Q + 255) / 256) * 256;
    b += sizeof(unsigned int) * (size_t)Q * kPasses * kBins;
    b += sizeof(unsigned long long) * (size_t)Q * kpad;
    return b + 256;
}

int launch_topk(hipStream_t s, const jv_ctx *ctx, const float *d_scores, const int32_t *d_ids, int Q, int64_t n,
                int64_t stride, int32_t id_base, int k, int32_t *d_out_ids, float *d_out_scores, void *d_scratch,
                const unsigned int *d_row_counts)
{
    if (Q == 0 || k == 0) return JV_OK;
    if (k > kMaxK) {
        set_error("topk: k=%d exceeds the supported maximum %d", k, kMaxK);
        return JV_ERR_UNSUPPORTED;
    }
    if (launch_topk_small(s, d_scores, d_ids, Q, n, stride, id_base, k, d_out_ids, d_out_scores, d_row_counts)) {
        JV_HIP_CHECK(hipGetLastError());
        return JV_OK;
    }
    if (n <= 8192 && !getenv("JVECTOR_HIP_TOPK_RADIX")) {
        const int npad = next_pow2(n < 2 ? 2 : (int)n);
        hipLaunchKernelGGL(topk_sortall_kernel, dim3(Q), dim3(1024), sizeof(unsigned long long) * (size_t)npad, s, d_scores, d_ids, (int)n,
                           npad, stride, id_base, d_row_counts, k, d_out_ids, d_out_scores);
        JV_HIP_CHECK(hipGetLastError());
        return JV_OK;
    }
    const int kpad = next_pow2(k < 2 ? 2 : k);
    char *base = (char *)d_scratch;
    SelState *st = (SelState *)base;
    size_t off = ((sizeof(SelState) * (size_t)Q + 255) / 256) * 256;
    unsigned int *hist = (unsigned int *)(base + off);
    const size_t hist_bytes = sizeof(unsigned int) * (size_t)Q * kPasses * kBins;
    unsigned long long *keys = (unsigned long long *)(base + off + hist_bytes);
    JV_HIP_CHECK(hipMemsetAsync(base, 0, off + hist_bytes, s));

    int nblk = (int)((n + 256 * 16 - 1) / (256 * 16));
    const int max_blk = ctx->num_cus * 8 / (Q < 8 ? Q : 8);
    if (nblk > max_blk) nblk = max_blk;
    if (nblk < 1) nblk = 1;
    dim3 grid(nblk, Q);
    for (int pass = 0; pass < kPasses; ++pass) {
        hipLaunchKernelGGL(topk_hist_kernel, grid, dim3(256), 0, s, d_scores, d_ids, n, stride, id_base, pass, st,
                           d_row_counts, hist);
        hipLaunchKernelGGL(topk_select_kernel, dim3(Q), dim3(256), 0, s, pass, k, st, hist);
    }
    hipLaunchKernelGGL(topk_collect_kernel, grid, dim3(256), 0, s, d_scores, d_ids, n, stride, id_base, st, d_row_counts,
                       keys, kpad);
    size_t lds = sizeof(unsigned long long) * (size_t)kpad;
    hipLaunchKernelGGL(topk_sort_kernel, dim3(Q), dim3(1024), lds, s, st, keys, kpad, k, d_out_ids, d_out_scores);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

__global__ void add_base_kernel(int32_t *ids, int64_t n, int32_t base)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && ids[i] >= 0) ids[i] += base;
}

int launch_add_id_base(hipStream_t s, int32_t *d_ids, int64_t n, int32_t base)
{
    if (n == 0 || base == 0) return JV_OK;
    hipLaunchKernelGGL(add_base_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_ids, n, base);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

// ------------------------------------------------------------------------------------------------
// Exact-score ties at the K-th place of the rerank (rt_body.h): one wavefront per query detects them and rebuilds the
// reference's answer from the traversal's push log; what it cannot resolve is marked for the host searcher.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void rerank_tie_kernel(RtParams p)
{
    extern __shared__ __attribute__((aligned(16))) char rt_lds[];
    const int q = blockIdx.x;
    if (q < p.Q) rt_query(p, q, rt_lds);
}

int launch_rerank_ties(hipStream_t s, const RtParams &p)
{
    if (p.Q == 0) return JV_OK;
    const size_t lds = rt_lds_bytes(p.rerankK, p.K);
    if (lds > 64 * 1024)
        JV_HIP_CHECK(hipFuncSetAttribute((const void *)rerank_tie_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(rerank_tie_kernel, dim3(p.Q), dim3(64), lds, s, p);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

}  // namespace jv
