// k_nvq.hip — NVQ ("NuVeQ", non-uniform vector quantization), the reference's compressed RERANK codec, on gfx950.
//
// Replaces (scalar reference path, bit for bit):
//   NVQuantization.compute / encodeAll / QuantizedSubVector.quantizeTo   B/quantization/NVQuantization.java:153-163,182-216,508-557
//   NVQScorer.scoreFunctionFor(query, vsf).similarityTo(vector)          B/quantization/NVQScorer.java:33-137
//   nvqQuantize8bit / nvqLoss / nvqUniformLoss / nvqDotProduct8bit / nvqSquareL2Distance8bit / nvqCosine8bit
//                                                                       B/vector/DefaultVectorUtilSupport.java:385-548
// (native counterparts NC/src/jvector_simd_kernels.cpp:1029-1643).  The reference writes these chains with Math.fma, so
// the fused operations below are explicit __builtin_fmaf calls; the translation unit is still compiled with
// -ffp-contract=off so that nothing ELSE fuses.
//
// HBM layout of a jv_nvq_vectors (jv_internal.h): bytes[count][ld] (one byte per dimension, the sub-vectors' bytes
// concatenated, ld = D rounded up to 16, padding zero), params[count][S][4] = {min, max, growthRate, midpoint} as the file
// format stores them, derived[count][S][4] = {1/scaledGrowthRate, scaledMidpoint, logisticScale, logisticBias} — the
// numbers every nvq* function of the reference recomputes per call from the same four inputs (same bits, computed once by
// nvq_derive_kernel) — and, for cosine, cosnorm[count] = the query-independent squaredNormalization sum.
//
// Roofline: a reranked candidate costs D + 16 S bytes of HBM instead of 4 D, and ~26 VALU instructions per dimension
// (u8 -> float, fma, IEEE divide, exponent / mantissa split, two fma): the gather kernel is VALU-bound, not HBM-bound —
// see DESIGN.md "NVQ".
#include "jv_device.h"
#include "jv_internal.h"

#include <climits>

namespace jv {

// ---- Java arithmetic the chains are made of ------------------------------------------------------
__device__ __forceinline__ int nq_round(float x)   // Math.round(float): floor(x + 1/2) exactly, NaN -> 0, saturating
{
    if (x != x) return 0;
    const float f = floorf(x);
    const float r = (x - f >= 0.5f) ? f + 1.0f : f;   // x - f is exact; |x| >= 2^23 is integral already
    if (r <= -2147483648.0f) return INT_MIN;
    if (r >= 2147483648.0f) return INT_MAX;
    return (int)r;
}
__device__ __forceinline__ int nq_bits(float x) { return (x != x) ? 0x7fc00000 : __float_as_int(x); }   // Float.floatToIntBits
__device__ __forceinline__ float nq_min(float a, float b)   // Math.min(float, float): NaN wins, -0 < +0
{
    if (a != a) return a;
    if (b != b) return b;
    if (a == 0.0f && b == 0.0f) return (__float_as_int(a) < 0) ? a : b;
    return a <= b ? a : b;
}
__device__ __forceinline__ float nq_max(float a, float b)
{
    if (a != a) return a;
    if (b != b) return b;
    if (a == 0.0f && b == 0.0f) return (__float_as_int(a) < 0) ? b : a;
    return a >= b ? a : b;
}
// logisticFunctionNQT :441-448
__device__ __forceinline__ float nq_logistic(float value, float alpha, float x0)
{
    float temp = __builtin_fmaf(value, alpha, -alpha * x0);
    const int p = nq_round(temp + 0.5f);
    const int m = nq_bits(__builtin_fmaf(temp - (float)p, 0.5f, 1.0f));
    temp = __int_as_float((int)((uint32_t)m + ((uint32_t)p << 23)));
    return temp / (temp + 1.0f);
}
// logitNQT :450-459
__device__ __forceinline__ float nq_logit(float value, float inverseAlpha, float x0)
{
    const float z = value / (1.0f - value);
    const int temp = nq_bits(z);
    const int e = temp & 0x7f800000;
    const float p = (float)((e >> 23) - 128);
    const float m = __int_as_float((temp & 0x007fffff) + 0x3f800000);
    return __builtin_fmaf(m + p, inverseAlpha, x0);
}
struct NqDerived { float sgr, smid, inv, bias, scale; };
// the preamble of every nvq* function (:386-391): levels = 255 for 8 bits
__device__ __forceinline__ NqDerived nq_derive(float growthRate, float midpoint, float minV, float maxV, float levels)
{
    NqDerived p;
    const float delta = maxV - minV;
    p.sgr = growthRate / delta;
    p.smid = midpoint * delta;
    p.inv = 1.0f / p.sgr;
    p.bias = nq_logistic(minV, p.sgr, p.smid);
    p.scale = (nq_logistic(maxV, p.sgr, p.smid) - p.bias) / levels;
    return p;
}
// scaledLogisticFunction :461-464 with 1 / logisticScale hoisted (same operands every call)
__device__ __forceinline__ float nq_scaled_logistic(float v, const NqDerived &p, float inv_scale)
{
    return (nq_logistic(v, p.sgr, p.smid) - p.bias) * inv_scale;
}
// scaledLogitFunctionNQT :466-469
__device__ __forceinline__ float nq_scaled_logit(float v, float inv, float smid, float scale, float bias)
{
    return nq_logit(__builtin_fmaf(v, scale, bias), inv, smid);
}

// VectorUtil.dotProduct (DefaultVectorUtilSupport.dotProduct :38-75): len % 8 leading elements one by one, then blocks of
// eight summed left to right and added to the running result — one thread
__device__ float nq_dot(const float *__restrict__ a, const float *__restrict__ b, int D)
{
    float res = 0.0f;
    int i = 0;
    const int rem = D % 8;
    for (; i < rem; ++i) res += b[i] * a[i];
    for (; i + 7 < D; i += 8) {
        float t = b[i] * a[i] + b[i + 1] * a[i + 1];
        t = t + b[i + 2] * a[i + 2];
        t = t + b[i + 3] * a[i + 3];
        t = t + b[i + 4] * a[i + 4];
        t = t + b[i + 5] * a[i + 5];
        t = t + b[i + 6] * a[i + 6];
        t = t + b[i + 7] * a[i + 7];
        res += t;
    }
    return res;
}

// ------------------------------------------------------------------------------------------------
// NVQuantization.compute: globalMean[j] = (sum over the vectors IN ORDER of v[j]) * (1.0f / n).
// One lane per column keeps the order; NQ_MU rows are in flight per lane.
// ------------------------------------------------------------------------------------------------
constexpr int NQ_MU = 32;
__global__ __launch_bounds__(64) void nvq_mean_kernel(const float *__restrict__ vecs, int64_t n, int D, float *__restrict__ mean)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j >= D) return;
    const float *col = vecs + j;
    float s = 0.0f;
    int64_t i = 0;
    for (; i + NQ_MU <= n; i += NQ_MU) {
        float r[NQ_MU];
#pragma unroll
        for (int u = 0; u < NQ_MU; ++u) r[u] = __builtin_nontemporal_load(col + (i + u) * D);
#pragma unroll
        for (int u = 0; u < NQ_MU; ++u) s = s + r[u];
    }
    for (; i < n; ++i) s = s + col[i * D];
    mean[j] = s * (1.0f / (float)(int)n);
}

// ------------------------------------------------------------------------------------------------
// encode: QuantizedSubVector.quantizeTo for every (vector, sub-vector) unit.
// A unit's growth-rate search evaluates nvqLoss for 20 coarse and then <= 21 fine candidates; every evaluation is a
// sequential chain over the sub-vector.  21 lanes take one unit (one candidate chain per lane), three units share a
// wavefront (63 of 64 lanes busy); the sub-vector (already minus the global mean) sits in LDS and is read as a broadcast.
// grid: [0..21) coarse values (entry 20 = the 1e-2f a search without a winner falls back to), then 21 rows of 21 fine
// values, then 21 fine counts (as floats) — the float loops of :523-541 run once on the host (nvq.cpp).
// ------------------------------------------------------------------------------------------------
constexpr int NQ_G = 21;
__device__ __forceinline__ float nq_loss_chain(const float *__restrict__ v, int n, int nloop, float gr, float minV, float maxV)
{
    const NqDerived p = nq_derive(gr, 0.0f, minV, maxV, 255.0f);
    const float inv_scale = 1.0f / p.scale;
    float sq = 0.0f;
    for (int j = 0; j < nloop; ++j) {
        if (j < n) {
            const float x = v[j];
            float r = nq_scaled_logistic(x, p, inv_scale);
            r = (float)nq_round(r);
            r = nq_scaled_logit(r, p.inv, p.smid, p.scale, p.bias);
            const float diff = x - r;
            sq = __builtin_fmaf(diff, diff, sq);
        }
    }
    return sq;
}

__global__ __launch_bounds__(64) void nvq_encode_kernel(const float *__restrict__ vecs, int64_t count, int D, int S,
                                                        const float *__restrict__ mean, int learn,
                                                        const float *__restrict__ grid, uint8_t *__restrict__ bytes, int ld,
                                                        float *__restrict__ params, int nstride)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 3 x nstride values, then 3 x 32 scratch
    const int lane = threadIdx.x;
    const int g = lane / NQ_G, l = lane - g * NQ_G;
    const int64_t unit = (int64_t)blockIdx.x * 3 + g;
    const bool valid = g < 3 && unit < count * S;
    const int64_t i = valid ? unit / S : 0;
    const int s = valid ? (int)(unit - i * S) : 0;
    const int base = D / S, rem = D % S;
    const int n = valid ? base + (s < rem ? 1 : 0) : 0;
    const int off = s * base + (s < rem ? s : rem);
    float *v = lds + (g < 3 ? g : 0) * nstride;
    float *scr = lds + 3 * nstride + (g < 3 ? g : 0) * 32;
    const int nloop = base + (rem ? 1 : 0);   // wave-uniform trip count

    for (int j = l; j < n; j += NQ_G) v[j] = vecs[i * D + off + j] - mean[off + j];   // VectorUtil.sub(v, globalMean) :214
    __syncthreads();
    // VectorUtil.min / max :365-383 (order-free: Math.min / Math.max are associative and commutative)
    float mn = 3.4028234663852886e38f, mx = -3.4028234663852886e38f;
    for (int j = l; j < n; j += NQ_G) {
        mn = nq_min(mn, v[j]);
        mx = nq_max(mx, v[j]);
    }
    if (valid) scr[l] = mn;
    __syncthreads();
    if (valid)
        for (int t = 0; t < NQ_G; ++t) mn = nq_min(mn, scr[t]);
    __syncthreads();
    if (valid) scr[l] = mx;
    __syncthreads();
    if (valid)
        for (int t = 0; t < NQ_G; ++t) mx = nq_max(mx, scr[t]);
    __syncthreads();

    float growthRate = 1e-2f;
    if (learn) {   // wave-uniform
        // nvqUniformLoss :518-536 (NonuniformQuantizationLossFunction.setVector :672-677) — every lane of the unit, same value
        float baseline = 0.0f;
        for (int j = 0; j < nloop; ++j) {
            if (j < n) {
                const float x = v[j];
                float r = (x - mn) / (mx - mn);
                r = (float)nq_round(255.0f * r) / 255.0f;
                r = r * (mx - mn) + mn;
                const float diff = x - r;
                baseline = __builtin_fmaf(diff, diff, baseline);
            }
        }
        // coarse pass :523-531
        float loss = 0.0f;
        if (valid && l < 20) loss = baseline / nq_loss_chain(v, n, nloop, grid[l], mn, mx);
        if (valid) scr[l] = loss;
        __syncthreads();
        float best = 1.401298464324817e-45f;   // Float.MIN_VALUE
        int cidx = 20;
        if (valid)
            for (int t = 0; t < 20; ++t) {
                const float lv = scr[t];
                if (lv > best) {
                    best = lv;
                    cidx = t;
                }
            }
        __syncthreads();
        // fine pass :532-540
        const float *fine = grid + NQ_G + cidx * NQ_G;
        const int nf = (int)grid[NQ_G + NQ_G * NQ_G + cidx];
        loss = 0.0f;
        if (valid && l < nf) loss = baseline / nq_loss_chain(v, n, nloop, fine[l], mn, mx);
        if (valid) scr[l] = loss;
        __syncthreads();
        growthRate = grid[cidx];
        if (valid)
            for (int t = 0; t < nf; ++t) {
                const float lv = scr[t];
                if (lv > best) {
                    best = lv;
                    growthRate = fine[t];
                }
            }
    }
    // nvqQuantize8bit :471-488
    if (valid) {
        const NqDerived p = nq_derive(growthRate, 0.0f, mn, mx, 255.0f);
        const float inv_scale = 1.0f / p.scale;
        uint8_t *dst = bytes + i * ld + off;
        for (int j = l; j < n; j += NQ_G) dst[j] = (uint8_t)((uint32_t)nq_round(nq_scaled_logistic(v[j], p, inv_scale)) & 0xffu);
        if (l == 0) {
            float *pp = params + unit * 4;
            pp[0] = mn;
            pp[1] = mx;
            pp[2] = growthRate;
            pp[3] = 0.0f;
        }
    }
}

// derived[u] = {1 / scaledGrowthRate, scaledMidpoint, logisticScale, logisticBias} of unit u = (row, sub-vector)
__global__ __launch_bounds__(256) void nvq_derive_kernel(const float4 *__restrict__ params, int64_t units, float4 *__restrict__ derived)
{
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= units) return;
    const float4 r = params[u];   // min, max, growthRate, midpoint
    const NqDerived p = nq_derive(r.z, r.w, r.x, r.y, 255.0f);
    derived[u] = make_float4(p.inv, p.smid, p.scale, p.bias);
}

// ------------------------------------------------------------------------------------------------
// gather form: score[q][b] = NVQScorer...similarityTo(row ord[q][b]).
// Same structure as exact_gather_tr_kernel (k_exact.hip): one wavefront = 64 candidates of one query; a row is consumed
// in chunks of 128 bytes — the wave loads 8 rows x 128 B per instruction (whole lines) and parks the chunk in LDS as
// tile[row][128 + 16 pad]; lane j then walks row j with ds_read_b128 (16 dimensions per read; row stride 36 dwords: the
// 16 lanes of a b128 phase start at 16 distinct multiples of 4 banks) and runs the reference's chain: de-quantise, fma
// into the accumulator.  The query slice is wave-uniform (scalar loads).  The kernel is VALU-bound (the de-quantisation is
// ~26 instructions per dimension, half of them the IEEE divide), so the tile is kept small: 9 KB of LDS per wave lets
// 16 waves share a CU (the 256-byte chunks of the float kernel would cap it at 9).
// Sub-vector boundaries are wave-uniform too: at each one the chain's value is added to the running total
// (`nvqDot += ...` :66) and the lane fetches its row's next derived quadruple.
// NORM = true builds the cosine table instead: cosnorm[row] = sum over sub-vectors of normDQ (:448, NVQScorer :127).
// ------------------------------------------------------------------------------------------------
constexpr int NQ_CH = 128;           // bytes of a row per chunk
constexpr int NQ_LS = NQ_CH + 16;    // LDS row stride (bytes)
constexpr int NQ_LPR = NQ_CH / 16;   // lanes that cover one row's chunk with 16 bytes each (8)
constexpr int NQ_RPI = 64 / NQ_LPR;  // rows fetched per load instruction (8)
constexpr int NQ_NI = 64 / NQ_RPI;   // load instructions per chunk (8)

// a / b for operands well inside the normal range: the fma sequence the compiler's IEEE division is made of (reciprocal
// estimate, one Newton step on it, quotient with two residual corrections — correctly rounded) without the v_div_scale /
// v_div_fmas / v_div_fixup wrapping that only matters at the ends of the exponent range.  Plain fma chains also let the
// compiler pair two dimensions per v_pk_fma_f32.
__device__ __forceinline__ float nq_div_fast(float a, float b)
{
    float r = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = a * r;
    float t = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(t, r, q);
    t = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(t, r, q);
}
// is every byte of a sub-vector with these derived numbers safe for nq_div_fast?  sv = fma(byte, scale, bias) is monotone in
// the byte, so both ends inside [2^-40, 1 - 2^-20] put every sv there, 1 - sv inside [2^-20, 1) and the quotient inside
// [2^-41, 2^20]: finite, normal, never NaN.  (Comparisons are false for NaN parameters -> the IEEE path.)
__device__ __forceinline__ bool nq_fast_ok(const float4 prm)
{
    const float s0 = prm.w, s1 = __builtin_fmaf(255.0f, prm.z, prm.w);
    const float lo = s0 < s1 ? s0 : s1, hi = s0 < s1 ? s1 : s0;
    return lo >= 9.094947017729282e-13f && hi <= 0.99999905f;
}

// two dimensions at a time through the short division, on packed f32 operations (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32
// are full rate: half the issue slots of the fma chain); only the two reciprocal estimates, the exponent / mantissa splits
// and the two chain steps (sequential by definition) stay scalar.  Same operations per lane as nq_elem<.., FAST = true>.
typedef float nq_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ nq_f2 nq_fma2(nq_f2 a, nq_f2 b, nq_f2 c) { return __builtin_elementwise_fma(a, b, c); }

template <int VSF, bool NORM>
__device__ __forceinline__ void nq_pair(uint32_t b0, uint32_t b1, const float4 prm, float qa0, float qa1, float ca0, float ca1, float &acc)
{
    const nq_f2 x = {(float)b0, (float)b1};
    const nq_f2 sv = nq_fma2(x, nq_f2{prm.z, prm.z}, nq_f2{prm.w, prm.w});
    const nq_f2 den = nq_f2{1.0f, 1.0f} - sv;
    nq_f2 r = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    const nq_f2 e = nq_fma2(-den, r, nq_f2{1.0f, 1.0f});
    r = nq_fma2(e, r, r);
    nq_f2 q = sv * r;
    nq_f2 t = nq_fma2(-den, q, sv);
    q = nq_fma2(t, r, q);
    t = nq_fma2(-den, q, sv);
    const nq_f2 z = nq_fma2(t, r, q);
    const int z0 = __float_as_int(z.x), z1 = __float_as_int(z.y);
    const nq_f2 pe = {(float)(((z0 & 0x7f800000) >> 23) - 128), (float)(((z1 & 0x7f800000) >> 23) - 128)};
    const nq_f2 mn = {__int_as_float((z0 & 0x007fffff) + 0x3f800000), __int_as_float((z1 & 0x007fffff) + 0x3f800000)};
    nq_f2 val = nq_fma2(mn + pe, nq_f2{prm.x, prm.x}, nq_f2{prm.y, prm.y});
    if (NORM) {
        val = val + nq_f2{ca0, ca1};
        acc = __builtin_fmaf(val.x, val.x, acc);
        acc = __builtin_fmaf(val.y, val.y, acc);
    } else if (VSF == VSF_DOT) {
        acc = __builtin_fmaf(qa0, val.x, acc);
        acc = __builtin_fmaf(qa1, val.y, acc);
    } else if (VSF == VSF_L2) {
        const nq_f2 d = val - nq_f2{qa0, qa1};
        acc = __builtin_fmaf(d.x, d.x, acc);
        acc = __builtin_fmaf(d.y, d.y, acc);
    } else {
        val = val + nq_f2{ca0, ca1};
        acc = __builtin_fmaf(qa0, val.x, acc);
        acc = __builtin_fmaf(qa1, val.y, acc);
    }
}

template <int VSF, bool NORM, bool FAST>
__device__ __forceinline__ void nq_elem(uint32_t byte, const float4 prm, float qa, float ca, float &acc)
{
    float val;
    if (FAST) {   // nq_scaled_logit with the division above; the quotient is a positive normal number: its bits are Java's bits
        const float sv = __builtin_fmaf((float)byte, prm.z, prm.w);
        const int zb = __float_as_int(nq_div_fast(sv, 1.0f - sv));
        const float p = (float)(((zb & 0x7f800000) >> 23) - 128);
        const float m = __int_as_float((zb & 0x007fffff) + 0x3f800000);
        val = __builtin_fmaf(m + p, prm.x, prm.y);
    } else {
        val = nq_scaled_logit((float)byte, prm.x, prm.y, prm.z, prm.w);
    }
    if (NORM) {
        val += ca;
        acc = __builtin_fmaf(val, val, acc);
    } else if (VSF == VSF_DOT) {
        acc = __builtin_fmaf(qa, val, acc);
    } else if (VSF == VSF_L2) {
        const float t = val - qa;
        acc = __builtin_fmaf(t, t, acc);
    } else {
        val += ca;
        acc = __builtin_fmaf(qa, val, acc);
    }
}

// dimensions [i, iend) of the lane's row in the LDS tile (offsets inside the chunk; a / cen point at the chunk's first dimension)
template <int VSF, bool NORM, bool FAST>
__device__ __forceinline__ void nq_segment(const uint8_t *__restrict__ row, int i, int iend, const float4 prm, const float *__restrict__ a,
                                           const float *__restrict__ cen, float &acc)
{
    constexpr bool CEN = NORM || VSF == VSF_COS;
    for (; i < iend && (i & 15); ++i) nq_elem<VSF, NORM, FAST>(row[i], prm, NORM ? 0.0f : a[i], CEN ? cen[i] : 0.0f, acc);
    for (; i + 16 <= iend; i += 16) {
        const uint4 w = *reinterpret_cast<const uint4 *>(row + i);
        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
        if (FAST) {
#pragma unroll
            for (int t = 0; t < 16; t += 2)
                nq_pair<VSF, NORM>((ws[t >> 2] >> (8 * (t & 3))) & 0xffu, (ws[t >> 2] >> (8 * ((t + 1) & 3))) & 0xffu, prm,
                                   NORM ? 0.0f : a[i + t], NORM ? 0.0f : a[i + t + 1], CEN ? cen[i + t] : 0.0f, CEN ? cen[i + t + 1] : 0.0f, acc);
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t)
                nq_elem<VSF, NORM, FAST>((ws[t >> 2] >> (8 * (t & 3))) & 0xffu, prm, NORM ? 0.0f : a[i + t], CEN ? cen[i + t] : 0.0f, acc);
        }
    }
    for (; i < iend; ++i) nq_elem<VSF, NORM, FAST>(row[i], prm, NORM ? 0.0f : a[i], CEN ? cen[i] : 0.0f, acc);
}

template <int VSF, bool NORM>
__device__ __forceinline__ float nq_rows(const uint8_t *__restrict__ bytes, int ld, int D, int S, int64_t my_row /* -1 = none */,
                                         const float4 *__restrict__ derived, const float *__restrict__ a,
                                         const float *__restrict__ cen, uint8_t *tile)
{
    const int lane = threadIdx.x;
    const int seg = (lane % NQ_LPR) * 16;   // this lane's 16 bytes inside a row's chunk
    const int sub = lane / NQ_LPR;          // load instruction k fetches rows NQ_RPI * k + sub
    const uint8_t *rp[NQ_NI];
#pragma unroll
    for (int k = 0; k < NQ_NI; ++k) {
        const int64_t ro = __shfl(my_row, NQ_RPI * k + sub, 64);
        rp[k] = ro >= 0 ? bytes + ro * ld + seg : nullptr;
    }
    const int nc = (D + NQ_CH - 1) / NQ_CH;
    uint4 r[NQ_NI];
    auto issue = [&](int c) {
        const bool in = c * NQ_CH + seg < ld;
#pragma unroll
        for (int k = 0; k < NQ_NI; ++k)
            r[k] = (rp[k] && in) ? *reinterpret_cast<const uint4 *>(rp[k] + c * NQ_CH) : make_uint4(0u, 0u, 0u, 0u);
    };
    issue(0);
    const int base = D / S, rem = D % S;
    int s = 0, sub_end = base + (rem ? 1 : 0);
    const float4 *dp = derived + (my_row >= 0 ? my_row : 0) * S;
    const float4 none = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float4 prm = my_row >= 0 ? dp[0] : none;
    float total = 0.0f, acc = 0.0f;
    for (int c = 0; c < nc; ++c) {
#pragma unroll
        for (int k = 0; k < NQ_NI; ++k) *reinterpret_cast<uint4 *>(tile + (NQ_RPI * k + sub) * NQ_LS + seg) = r[k];
        __syncthreads();
        if (c + 1 < nc) issue(c + 1);
        const uint8_t *row = tile + lane * NQ_LS;
        const int c0 = c * NQ_CH;
        int d = c0;
        const int dend = (D - c0 < NQ_CH) ? D : c0 + NQ_CH;
        while (d < dend) {
            const int seg_end = sub_end < dend ? sub_end : dend;
            const int ibeg = d - c0, iend = seg_end - c0;
            // wave-uniform choice: every lane's row allows the short division for this sub-vector (almost always)
            if (__all(my_row < 0 || nq_fast_ok(prm))) nq_segment<VSF, NORM, true>(row, ibeg, iend, prm, a + c0, cen + c0, acc);
            else nq_segment<VSF, NORM, false>(row, ibeg, iend, prm, a + c0, cen + c0, acc);
            d = seg_end;
            if (d == sub_end) {
                total += acc;
                acc = 0.0f;
                ++s;
                if (s < S) {
                    sub_end += base + (s < rem ? 1 : 0);
                    prm = my_row >= 0 ? dp[s] : none;
                }
            }
        }
        __syncthreads();
    }
    return total;
}

// per query: DOT -> qaux[q] = VectorUtil.dotProduct(query, globalMean) (NVQScorer :55); EUCLIDEAN -> qwork[q] = query - globalMean
// (:82); COSINE -> qaux[q] = (float) Math.sqrt(dotProduct(query, query)) (:109)
__global__ __launch_bounds__(64) void nvq_query_prep_kernel(const float *__restrict__ queries, int Q, int D, int vsf,
                                                            const float *__restrict__ mean, float *__restrict__ qwork,
                                                            float *__restrict__ qaux)
{
    const int q = blockIdx.x;
    const float *qq = queries + (int64_t)q * D;
    if (vsf == VSF_L2) {
        for (int j = threadIdx.x; j < D; j += 64) qwork[(int64_t)q * D + j] = qq[j] - mean[j];
    } else if (threadIdx.x == 0) {
        qaux[q] = (vsf == VSF_DOT) ? nq_dot(qq, mean, D) : (float)sqrt((double)nq_dot(qq, qq, D));
    }
}

template <int VSF>
__global__ __launch_bounds__(64) void nvq_gather_kernel(const uint8_t *__restrict__ bytes, int ld, int64_t n, int D, int S,
                                                        const float4 *__restrict__ derived, const float *__restrict__ cosnorm,
                                                        const float *__restrict__ mean, const float *__restrict__ queries,
                                                        const float *__restrict__ qaux, const int32_t *__restrict__ ord, int B,
                                                        float *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[64 * NQ_LS];
    const int q = blockIdx.x;
    const int j = blockIdx.y * 64 + threadIdx.x;
    int64_t o = -1;
    if (j < B) {
        o = ord[(int64_t)q * B + j];
        if (o >= n) o = -1;
    }
    const float chain = nq_rows<VSF, false>(bytes, ld, D, S, o, derived, queries + (int64_t)q * D, mean, tile);
    if (j >= B) return;
    float *dst = out + (int64_t)q * B + j;
    if (o < 0) {
        *dst = -INFINITY;
        return;
    }
    if (VSF == VSF_DOT) {
        *dst = (1.0f + chain + qaux[q]) / 2.0f;                       // :71
    } else if (VSF == VSF_L2) {
        *dst = 1.0f / (1.0f + chain);                                 // :101
    } else {
        const float cosine = (chain / qaux[q]) / (float)sqrt((double)cosnorm[o]);   // :129
        *dst = (1.0f + cosine) / 2.0f;
    }
}

__global__ __launch_bounds__(64) void nvq_cosnorm_kernel(const uint8_t *__restrict__ bytes, int ld, int64_t n, int D, int S,
                                                         const float4 *__restrict__ derived, const float *__restrict__ mean,
                                                         float *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[64 * NQ_LS];
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const float sq = nq_rows<VSF_COS, true>(bytes, ld, D, S, i < n ? i : -1, derived, nullptr, mean, tile);
    if (i < n) out[i] = sq;
}

// ---- launchers -----------------------------------------------------------------------------------
int launch_nvq_mean(hipStream_t s, const float *d_vecs, int64_t n, int D, float *d_mean)
{
    hipLaunchKernelGGL(nvq_mean_kernel, dim3((D + 63) / 64), dim3(64), 0, s, d_vecs, n, D, d_mean);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

size_t nvq_encode_lds_bytes(int D, int S) { return sizeof(float) * (3 * (size_t)(D / S + 1) + 96); }

int launch_nvq_encode(hipStream_t s, const jv_ctx *ctx, const float *d_vecs, int64_t count, int D, int S, const float *d_mean, int learn,
                      const float *d_grid, uint8_t *d_bytes, int ld, float *d_params)
{
    if (count == 0) return JV_OK;
    const size_t lds = nvq_encode_lds_bytes(D, S);
    const size_t lds_max = ctx->lds_per_block < 65536 ? ctx->lds_per_block : 65536;   // dynamic LDS without a per-function opt-in
    if (lds > lds_max) {
        set_error("nvq_encode: a sub-vector of %d dimensions needs %zu bytes of LDS (limit %zu); use more sub-vectors", D / S + 1, lds,
                  lds_max);
        return JV_ERR_UNSUPPORTED;
    }
    const int64_t units = count * S, blocks = (units + 2) / 3;
    if (blocks > 0x7fffffffLL) {
        set_error("nvq_encode: %lld units in one launch", (long long)units);
        return JV_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(nvq_encode_kernel, dim3((unsigned)blocks), dim3(64), lds, s, d_vecs, count, D, S, d_mean, learn, d_grid, d_bytes,
                       ld, d_params, D / S + 1);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

int launch_nvq_derive(hipStream_t s, const float *d_params, int64_t units, float *d_derived)
{
    if (units == 0) return JV_OK;
    hipLaunchKernelGGL(nvq_derive_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, s, (const float4 *)d_params, units,
                       (float4 *)d_derived);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

int launch_nvq_cosnorm(hipStream_t s, const uint8_t *d_bytes, int ld, int64_t n, int D, int S, const float *d_derived, const float *d_mean,
                       float *d_out)
{
    if (n == 0) return JV_OK;
    hipLaunchKernelGGL(nvq_cosnorm_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, d_bytes, ld, n, D, S, (const float4 *)d_derived,
                       d_mean, d_out);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

// d_qwork: Q x D floats (EUCLIDEAN: the shifted queries), d_qaux: Q floats — caller-provided scratch
int launch_nvq_gather(hipStream_t s, const uint8_t *d_bytes, int ld, int64_t n, int D, int S, const float *d_derived, const float *d_cosnorm,
                      const float *d_mean, const float *d_q, int Q, int vsf, const int32_t *d_ord, int B, float *d_out, float *d_qwork,
                      float *d_qaux)
{
    if (Q == 0 || B == 0) return JV_OK;
    hipLaunchKernelGGL(nvq_query_prep_kernel, dim3(Q), dim3(64), 0, s, d_q, Q, D, vsf, d_mean, d_qwork, d_qaux);
    const dim3 grid(Q, (B + 63) / 64), block(64);
    const float4 *dv = (const float4 *)d_derived;
    switch (vsf) {
    case VSF_L2:
        hipLaunchKernelGGL(nvq_gather_kernel<VSF_L2>, grid, block, 0, s, d_bytes, ld, n, D, S, dv, d_cosnorm, d_mean, (const float *)d_qwork,
                           d_qaux, d_ord, B, d_out);
        break;
    case VSF_DOT:
        hipLaunchKernelGGL(nvq_gather_kernel<VSF_DOT>, grid, block, 0, s, d_bytes, ld, n, D, S, dv, d_cosnorm, d_mean, d_q, d_qaux, d_ord, B,
                           d_out);
        break;
    default:
        hipLaunchKernelGGL(nvq_gather_kernel<VSF_COS>, grid, block, 0, s, d_bytes, ld, n, D, S, dv, d_cosnorm, d_mean, d_q, d_qaux, d_ord, B,
                           d_out);
        break;
    }
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

}  // namespace jv
