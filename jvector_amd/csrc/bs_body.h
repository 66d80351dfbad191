// bs_body.h — build-time scoring: the PQ-only score functions graph CONSTRUCTION uses
// (BuildScoreProvider.pqBuildScoreProvider, B/graph/similarity/BuildScoreProvider.java:167-212), batched:
//   pair table     ProductQuantization.createCodebookPartialSums (B/quantization/ProductQuantization.java:609-628):
//                  per subspace the upper triangle of centroid x centroid dot products / squared distances
//                  (M * k(k+1)/2 floats, 12.6 MB at M = 96)
//   pair scores    ImmutablePQVectors.diversityFunctionFor(node1).similarityTo(node2)
//                  (B/quantization/ImmutablePQVectors.java:61-104) = VectorUtil.assembleAndSumPQ
//                  (DefaultVectorUtilSupport.java:312-335; native jvector_simd_kernels.cpp:729-815) + the transform
//   decode         ProductQuantization.decode (:454-471) — searchProviderFor(node1) scores from the DECODED vector
//   direct scores  PQVectors.scoreFunctionFor(q, vsf) (B/quantization/PQVectors.java:223-281): query vs code without a
//                  look-up table
// Every function is the body of ONE thread of a flat launch (its global index is an argument), so the same source is
// compiled for the GPU by k_build_score.hip and as plain loops by the CPU tests (tests/emu/bs_emu.cpp).  The includer
// defines BS_FN and bs_sqrt(double).  Arithmetic: sequential f32 sums in the reference's order, no contraction.
#pragma once

#include <cstdint>

namespace jv {

struct BsPq {
    const float *codebooks;     // concatenated [m][k][size_m]
    const int64_t *cb_offsets;  // float offset of codebook m
    const int32_t *sizes;       // sub-vector length per m
    const int32_t *offsets;     // first dimension of sub-vector m
    const float *centroid;      // global centroid or nullptr
    int32_t D, M, k;
};

// row offset inside one subspace's triangle (assembleAndSumPQ: offsetRow = r*k - r*(r-1)/2)
BS_FN int64_t bs_tri_row(int r, int k) { return (int64_t)r * k - ((int64_t)r * (r - 1)) / 2; }

// ---- pair table: thread t = (m, i) fills row i of subspace m (entries j = i..k-1) ----
// vsf: 0 = EUCLIDEAN (squareL2Distance offsets form :195-208), else dotProduct offsets form (:107-119), both sequential
BS_FN void bs_pair_table_row(const BsPq &pq, int vsf, int64_t t, float *out)
{
    const int m = (int)(t / pq.k), i = (int)(t % pq.k);
    if (m >= pq.M) return;
    const int size = pq.sizes[m];
    const float *cb = pq.codebooks + pq.cb_offsets[m];
    const float *a = cb + (int64_t)i * size;
    float *row = out + (int64_t)m * ((int64_t)pq.k * (pq.k + 1) / 2) + bs_tri_row(i, pq.k);
    for (int j = i; j < pq.k; ++j) {
        const float *b = cb + (int64_t)j * size;
        float sum = 0.0f;
        if (vsf == 0) {
            for (int d = 0; d < size; ++d) {
                const float diff = a[d] - b[d];
                sum += diff * diff;
            }
        } else {
            for (int d = 0; d < size; ++d) sum += a[d] * b[d];
        }
        row[j - i] = sum;
    }
}

// VectorUtil.assembleAndSumPQ.  The table entries are independent loads feeding one sequential f32 sum: fetched 16 at a time,
// then added in ascending m (same association, the load latency paid once per 16 entries).
BS_FN float bs_assemble_pq(const float *tri, int M, int k, const uint8_t *c1v, const uint8_t *c2v)
{
    const int64_t block = (int64_t)k * (k + 1) / 2;
    float res = 0.0f;
    int m = 0;
    for (; m + 16 <= M; m += 16) {
        float e[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int c1 = c1v[m + j], c2 = c2v[m + j];
            const int r = c1 < c2 ? c1 : c2, c = c1 < c2 ? c2 : c1;
            e[j] = tri[(int64_t)(m + j) * block + bs_tri_row(r, k) + (c - r)];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) res += e[j];
    }
    for (; m < M; ++m) {
        const int c1 = c1v[m], c2 = c2v[m];
        const int r = c1 < c2 ? c1 : c2, c = c1 < c2 ? c2 : c1;
        res += tri[(int64_t)m * block + bs_tri_row(r, k) + (c - r)];
    }
    return res;
}

// ---- pair scores: thread t = (p, b): node1[p] vs node2[p*B + b]; ordinals outside [0, n) give -inf ----
BS_FN void bs_pair_score(const float *tri, int vsf, int M, int k, const uint8_t *codes, int64_t n, const int32_t *node1,
                         const int32_t *node2, int B, int64_t t, float *out)
{
    const int64_t p = t / B;
    const int32_t n1 = node1[p], n2 = node2[t];
    if (n1 < 0 || n1 >= n || n2 < 0 || n2 >= n) {
        out[t] = -__builtin_inff();
        return;
    }
    const uint8_t *c1 = codes + (int64_t)n1 * M, *c2 = codes + (int64_t)n2 * M;
    const float sum = bs_assemble_pq(tri, M, k, c1, c2);
    float r;
    if (vsf == 0) {
        r = 1.0f / (1.0f + sum);
    } else if (vsf == 1) {
        r = (1.0f + sum) / 2.0f;
    } else {  // ImmutablePQVectors.java:80-91: sum / (float) Math.sqrt(norm1 * norm2), all in float but the sqrt
        const float norm1 = bs_assemble_pq(tri, M, k, c1, c1), norm2 = bs_assemble_pq(tri, M, k, c2, c2);
        const float prod = norm1 * norm2;
        const float cosine = sum / (float)bs_sqrt((double)prod);
        r = (1.0f + cosine) / 2.0f;
    }
    out[t] = r;
}

// ---- decode: thread t = (row, dimension) ----
BS_FN void bs_decode(const BsPq &pq, const uint8_t *codes, int64_t n, const int32_t *ordinals, int64_t first, int64_t t, float *out)
{
    const int64_t row = t / pq.D;
    const int d = (int)(t % pq.D);
    const int64_t ord = ordinals ? (int64_t)ordinals[row] : first + row;
    if (ord < 0 || ord >= n) {
        out[t] = 0.0f;
        return;
    }
    int m = 0;  // sub-vector holding dimension d (sizes differ by at most one: ProductQuantization.getSubvectorSizesAndOffsets)
    while (m + 1 < pq.M && pq.offsets[m + 1] <= d) ++m;
    const int code = codes[ord * pq.M + m];
    float v = pq.codebooks[pq.cb_offsets[m] + (int64_t)code * pq.sizes[m] + (d - pq.offsets[m])];
    if (pq.centroid) v = v + pq.centroid[d];
    out[t] = v;
}

struct alignas(16) bs_b16 { uint32_t w[4]; };

// ---- FusedPQ.writeInline (B/graph/disk/feature/FusedPQ.java:146-161): block[node][j] = code of neighbour j, zero padded.
//      thread t = (node, j, 16-byte chunk c) when M % 16 == 0 (chunk = 16), else (node, j, byte) (chunk = 1).
BS_FN void bs_fused_gather(const uint8_t *codes, int64_t n_codes, const int32_t *neighbors, int maxDegree, int M, int chunk, int64_t t,
                           uint8_t *blocks)
{
    const int per_row = M / chunk;
    const int64_t row = t / per_row;  // node * maxDegree + j
    const int c = (int)(t % per_row);
    const int32_t nb = neighbors[row];
    uint8_t *dst = blocks + row * M + (int64_t)c * chunk;
    const bool pad = nb < 0 || nb >= n_codes;
    const uint8_t *src = pad ? nullptr : codes + (int64_t)nb * M + (int64_t)c * chunk;
    if (chunk == 16) {  // one 16-byte move per thread (the launcher checked M % 16 == 0 and the base alignment)
        bs_b16 v = {{0u, 0u, 0u, 0u}};
        if (!pad) v = *reinterpret_cast<const bs_b16 *>(src);
        *reinterpret_cast<bs_b16 *>(dst) = v;
        return;
    }
    for (int b = 0; b < chunk; ++b) dst[b] = pad ? (uint8_t)0 : src[b];
}

// VectorUtil.dotProduct(a, b) full-vector form (DefaultVectorUtilSupport.java:38-105): the FIRST len%8 elements one by
// one, then blocks of eight whose products are summed left to right before joining the running sum (the 32-wide
// unrolling of the reference is four such statements in sequence: same association).
BS_FN float bs_full_dot(const float *a, const float *b, int n)
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

// thread q: norm1 of PQVectors.scoreFunctionFor's COSINE branch (:244)
BS_FN void bs_query_norm(const float *cq, int D, int64_t q, float *out) { out[q] = bs_full_dot(cq + q * D, cq + q * D, D); }

// ---- direct scores: thread t = (q, b): centred query q vs the code of ordinals[q*B + b] ----
// cq: centred queries [Q][D]; qnorm[q] = dotProduct(cq, cq) (VectorUtil.dotProduct full-vector form, computed by the caller)
BS_FN void bs_direct_score(const BsPq &pq, int vsf, const uint8_t *codes, int64_t n, const float *cq, const float *qnorm,
                           const int32_t *ordinals, int B, int64_t t, float *out)
{
    const int64_t q = t / B;
    const int32_t ord = ordinals[t];
    if (ord < 0 || ord >= n) {
        out[t] = -__builtin_inff();
        return;
    }
    const uint8_t *code = codes + (int64_t)ord * pq.M;
    const float *query = cq + q * pq.D;
    float sum = 0.0f, norm2 = 0.0f;
    for (int m = 0; m < pq.M; ++m) {
        const int len = pq.sizes[m];
        const float *c = pq.codebooks + pq.cb_offsets[m] + (int64_t)code[m] * len;
        const float *x = query + pq.offsets[m];
        float part = 0.0f;
        if (vsf == 0) {
            for (int d = 0; d < len; ++d) {
                const float diff = c[d] - x[d];
                part += diff * diff;
            }
        } else {
            for (int d = 0; d < len; ++d) part += c[d] * x[d];
        }
        sum += part;
        if (vsf == 2) {
            float self = 0.0f;
            for (int d = 0; d < len; ++d) self += c[d] * c[d];
            norm2 += self;
        }
    }
    float r;
    if (vsf == 0) r = 1.0f / (1.0f + sum);
    else if (vsf == 1) r = (1.0f + sum) / 2.0f;
    else {
        const float prod = qnorm[q] * norm2;
        r = (1.0f + sum / (float)bs_sqrt((double)prod)) / 2.0f;
    }
    out[t] = r;
}

}  // namespace jv
