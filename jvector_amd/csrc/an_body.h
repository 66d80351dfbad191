// an_body.h — anisotropic PQ encode: ProductQuantization.encodeTo with anisotropicThreshold > -1
// (B/quantization/ProductQuantization.java:269-306 encodeAnisotropic, :384-420 computeResiduals / computeResidual,
// :364-379 initializeToMinResidualNorms, :308-349 optimizeSingleSubspace; pcm from
// KMeansPlusPlusClusterer.computeParallelCostMultiplier :116-124).  Integer output: codes must be bit-exact.
//
// One 64-lane wavefront encodes one vector.  The reference materialises an M x 256 table of
// {residualNormSquared, parallelResidualComponent}; at M = 96 that is 196 KB per vector — more than LDS — so the wave
// RECOMPUTES an entry from the centroid row (L2-resident codebook, 8 products) whenever it needs one: same float
// operations in the same order, hence the same values.  Each coordinate-descent step (one subspace) is a 256-way
// search for the first minimum of costDelta; lane l scans candidates l, l+64, ... in ascending order with the
// reference's strict `<`, and the wave takes the minimum of (costDelta, index) keys — the sequential scan's winner.
// Steps are inherently sequential (the running parallel-residual sum feeds the next subspace): <= 10 sweeps x M steps.
//
// Written against the wave API of gs_body.h (GS_FN, gs_lane, gs_barrier, gs_ballot, gs_shfl, gs_shfl_xor, gs_sqrt) so the
// CPU tests run the same source on the lane emulator.
#pragma once

#include <cstdint>

namespace jv {

struct AnParams {
    const float *codebooks;     // concatenated [m][k][size_m]
    const int64_t *cb_offsets;
    const int32_t *sizes, *offsets;
    const float *centroid;      // global centroid or nullptr
    const float *cnorm;         // centroidNormsSquared [M][k] (ProductQuantization.java:241-248) == jv_pq::d_self_mag
    const float *vecs;          // [n][D]
    uint8_t *codes;             // [n][M] out
    int64_t n;
    int32_t D, M, k;
    float pcm;                  // computeParallelCostMultiplier(threshold, D), computed on the host in double
};

inline size_t an_lds_bytes(int D, int M) { return sizeof(float) * (size_t)D + (((size_t)M + 15) & ~(size_t)15); }

GS_FN long long an_key(float v, int idx)  // order by float value, then by index (first minimum wins)
{
    int32_t b;
    __builtin_memcpy(&b, &v, 4);
    const int32_t s = b ^ ((b >> 31) & 0x7fffffff);
    return (long long)((((unsigned long long)(uint32_t)s) << 32) | (unsigned long long)(uint32_t)idx);
}
constexpr long long AN_KEY_NONE = (long long)0x7fffffffffffffffull;

GS_FN long long an_wave_min(long long v)
{
    for (int o = 32; o > 0; o >>= 1) {
        const long long t = gs_shfl_xor(v, o);
        v = t < v ? t : v;
    }
    return v;
}

// VectorUtil.dotProduct(a, b) full-vector form (first len%8 elements, then blocks of eight summed left to right)
GS_FN float an_full_dot(const float *a, const float *b, int n)
{
    float res = 0.0f;
    int i = 0;
    for (; i < n % 8; ++i) res += b[i] * a[i];
    for (; i + 7 < n; i += 8) {
        const float *x = a + i, *y = b + i;
        float t = y[0] * x[0] + y[1] * x[1];
        t = t + y[2] * x[2];
        t = t + y[3] * x[3];
        t = t + y[4] * x[4];
        t = t + y[5] * x[5];
        t = t + y[6] * x[6];
        t = t + y[7] * x[7];
        res += t;
    }
    return res;
}

// computeResidual :414-420 for centroid row c of length len against sub-vector x
GS_FN void an_residual(const float *c, const float *x, int len, float cNormSquared, float xNormSquared, float inverseNorm,
                       float *rns, float *par)
{
    float cDotX = 0.0f;
    for (int d = 0; d < len; ++d) cDotX += c[d] * x[d];  // dotProduct offsets form: sequential
    const float two = 2 * cDotX;
    *rns = cNormSquared - two + xNormSquared;
    const float pes = cDotX - xNormSquared;
    *par = (pes * pes) * inverseNorm;
}

GS_FN void an_encode_one(const AnParams &p, int64_t v, char *lds)
{
    const int lane = gs_lane();
    float *xs = reinterpret_cast<float *>(lds);
    uint8_t *cs = reinterpret_cast<uint8_t *>(lds + sizeof(float) * (size_t)p.D);
    const float *src = p.vecs + v * p.D;
    for (int d = lane; d < p.D; d += 64) xs[d] = p.centroid ? src[d] - p.centroid[d] : src[d];  // VectorUtil.sub
    gs_barrier();
    const float inverseNorm = (float)(1.0 / gs_sqrt((double)an_full_dot(xs, xs, p.D)));

    // ---- initializeToMinResidualNorms over freshly computed residual norms ----
    for (int i = 0; i < p.M; ++i) {
        const int len = p.sizes[i];
        const float *x = xs + p.offsets[i];
        const float xNorm = an_full_dot(x, x, len);
        const float *cb = p.codebooks + p.cb_offsets[i];
        long long best = AN_KEY_NONE;
        float bestv = 0.0f;
        bool have = false;
        for (int j = lane; j < p.k; j += 64) {
            float rns, par;
            an_residual(cb + (int64_t)j * len, x, len, p.cnorm[i * p.k + j], xNorm, inverseNorm, &rns, &par);
            if (!have ? (rns < __builtin_inff()) : (rns < bestv)) {  // strict < from Double.MAX_VALUE: first minimum, never NaN / +inf
                have = true;
                bestv = rns;
                best = an_key(rns, j);
            }
        }
        const long long m = an_wave_min(best);
        if (lane == 0) cs[i] = (uint8_t)(m == AN_KEY_NONE ? 255 : (int)((unsigned long long)m & 0xFFFFFFFFull));  // (byte) -1
    }
    gs_barrier();

    // ---- initial parallel residual sum (sequential over subspaces) ----
    float parSum = 0.0f;
    for (int i = 0; i < p.M; ++i) {
        const int len = p.sizes[i];
        const float *x = xs + p.offsets[i];
        const int c = cs[i];
        float rns, par;
        an_residual(p.codebooks + p.cb_offsets[i] + (int64_t)c * len, x, len, p.cnorm[i * p.k + c], an_full_dot(x, x, len), inverseNorm,
                    &rns, &par);
        parSum += par;
    }

    // ---- coordinate descent: <= 10 sweeps ----
    for (int iter = 0; iter < 10; ++iter) {
        bool changed = false;
        for (int i = 0; i < p.M; ++i) {
            const int len = p.sizes[i];
            const float *x = xs + p.offsets[i];
            const float xNorm = an_full_dot(x, x, len);
            const float *cb = p.codebooks + p.cb_offsets[i];
            const int oldIdx = cs[i];
            float oldRns, oldPar;
            an_residual(cb + (int64_t)oldIdx * len, x, len, p.cnorm[i * p.k + oldIdx], xNorm, inverseNorm, &oldRns, &oldPar);
            float bestCost = 0.0f, bestParSum = parSum;
            int bestIdx = -1;
            for (int t = lane; t < p.k; t += 64) {
                if (t == oldIdx) continue;
                float rns, par;
                an_residual(cb + (int64_t)t * len, x, len, p.cnorm[i * p.k + t], xNorm, inverseNorm, &rns, &par);
                const float thisParSum = parSum - oldPar + par;
                const float parallelNormDelta = thisParSum * thisParSum - parSum * parSum;
                if (parallelNormDelta > 0) continue;
                const float residualNormDelta = rns - oldRns;
                const float perpendicularNormDelta = residualNormDelta - parallelNormDelta;
                const float costDelta = p.pcm * parallelNormDelta + perpendicularNormDelta;
                if (costDelta < bestCost) {
                    bestCost = costDelta;
                    bestIdx = t;
                    bestParSum = thisParSum;
                }
            }
            const long long key = bestIdx >= 0 ? an_key(bestCost, bestIdx) : AN_KEY_NONE;
            const long long m = an_wave_min(key);
            if (m != AN_KEY_NONE) {  // uniform
                const int winner = __builtin_ctzll(gs_ballot(key == m));
                int32_t bits;
                __builtin_memcpy(&bits, &bestParSum, 4);
                bits = (int32_t)gs_shfl((long long)bits, winner);
                __builtin_memcpy(&parSum, &bits, 4);
                gs_barrier();  // every lane has read cs[i] (oldIdx) before it changes
                if (lane == 0) cs[i] = (uint8_t)((unsigned long long)m & 0xFFFFFFFFull);
                changed = true;
                gs_barrier();
            }
        }
        if (!changed) break;
    }
    gs_barrier();
    for (int m = lane; m < p.M; m += 64) p.codes[v * p.M + m] = cs[m];
    gs_barrier();
}

GS_FN void an_worker(const AnParams &p, int worker, int workers, char *lds)
{
    for (int64_t v = worker; v < p.n; v += workers) an_encode_one(p, v, lds);
}

}  // namespace jv
