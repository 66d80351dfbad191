/*
 * jv_oracle.c — CPU oracle (scalar, Java-semantics) for the JVector distance/quantization
 * hot path.  TEST INFRASTRUCTURE ONLY — see jv_oracle.h for who may load this and for the
 * parity-pin status.  Build: see oracle/Makefile (-O2 -ffp-contract=off, no fast-math).
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference,
 * B/ = jvector-base/src/main/java/io/github/jbellis/jvector/).
 *
 * Java float semantics reproduced here:
 *   - binary32 arithmetic with one rounding per operation, never contracted into FMA;
 *   - `x += a + b + c` evaluates the right-hand side left-to-right first, then adds to x;
 *   - (float)(double expr) narrows with round-to-nearest-even.
 */
#include "jv_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Row 1 — dot / L2 / cosine   (B/vector/DefaultVectorUtilSupport.java)
 * ---------------------------------------------------------------------------------------- */

/* DefaultVectorUtilSupport.dotProduct(av, bv) :38-105.
 * NOTE the remainder elements are the FIRST len%8 (:50-52), then 32-blocks (:56-89) as four
 * statements each adding a left-to-right sum of eight products, then 8-blocks (:90-100). */
float jvo_dot(const float *a, const float *b, int n)
{
    float res = 0.0f;
    int i;
    for (i = 0; i < n % 8; i++) res += b[i] * a[i];
    if (n < 8) return res;
    for (; i + 31 < n; i += 32) {
        for (int s = 0; s < 32; s += 8) {
            const float *x = a + i + s, *y = b + i + s;
            float t = y[0] * x[0] + y[1] * x[1];
            t = t + y[2] * x[2];
            t = t + y[3] * x[3];
            t = t + y[4] * x[4];
            t = t + y[5] * x[5];
            t = t + y[6] * x[6];
            t = t + y[7] * x[7];
            res += t;
        }
    }
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

/* DefaultVectorUtilSupport.dotProduct(av, aoffset, bv, boffset, length) :107-119 — sequential. */
float jvo_dot_off(const float *a, int aoff, const float *b, int boff, int n)
{
    float sum = 0.0f;
    for (int i = 0; i < n; i++) sum += a[aoff + i] * b[boff + i];
    return sum;
}

/* DefaultVectorUtilSupport.squareDistance(av, bv) :158-193 (blocks of 8 via squareDistanceUnrolled
 * :175-193: eight diffs, left-to-right sum of eight squares, then added to squareSum; sequential tail). */
float jvo_l2(const float *a, const float *b, int n)
{
    float sq = 0.0f;
    int i;
    for (i = 0; i + 8 <= n; i += 8) {
        float d0 = a[i + 0] - b[i + 0], d1 = a[i + 1] - b[i + 1];
        float d2 = a[i + 2] - b[i + 2], d3 = a[i + 3] - b[i + 3];
        float d4 = a[i + 4] - b[i + 4], d5 = a[i + 5] - b[i + 5];
        float d6 = a[i + 6] - b[i + 6], d7 = a[i + 7] - b[i + 7];
        float t = d0 * d0 + d1 * d1;
        t = t + d2 * d2;
        t = t + d3 * d3;
        t = t + d4 * d4;
        t = t + d5 * d5;
        t = t + d6 * d6;
        t = t + d7 * d7;
        sq += t;
    }
    for (; i < n; i++) {
        float d = a[i] - b[i];
        sq += d * d;
    }
    return sq;
}

/* DefaultVectorUtilSupport.squareDistance(av, aoffset, bv, boffset, length) :195-208 — sequential. */
float jvo_l2_off(const float *a, int aoff, const float *b, int boff, int n)
{
    float sq = 0.0f;
    for (int i = 0; i < n; i++) {
        float d = a[aoff + i] - b[boff + i];
        sq += d * d;
    }
    return sq;
}

/* DefaultVectorUtilSupport.cosine :121-156 — three sequential float accumulators; the product
 * norm1*norm2 is formed in float, sqrt and divide are done in double, then narrowed (:138,:155). */
float jvo_cosine_off(const float *a, int aoff, const float *b, int boff, int n)
{
    float sum = 0.0f, norm1 = 0.0f, norm2 = 0.0f;
    for (int i = 0; i < n; i++) {
        float e1 = a[aoff + i], e2 = b[boff + i];
        sum += e1 * e2;
        norm1 += e1 * e1;
        norm2 += e2 * e2;
    }
    float prod = norm1 * norm2;
    return (float)((double)sum / sqrt((double)prod));
}
float jvo_cosine(const float *a, const float *b, int n) { return jvo_cosine_off(a, 0, b, 0, n); }

/* Score transforms: VectorSimilarityFunction.compare B/vector/VectorSimilarityFunction.java:40,54,67;
 * same formulas in PQDecoder.java:68,79,126 and FusedPQDecoder.java:125,139,217. */
float jvo_score_from_raw(int vsf, float raw)
{
    switch (vsf) {
    case JVO_EUCLIDEAN:   return 1.0f / (1.0f + raw);
    case JVO_DOT_PRODUCT: return (1.0f + raw) / 2.0f;
    default:              return (1.0f + raw) / 2.0f;
    }
}

float jvo_compare(int vsf, const float *a, const float *b, int n)
{
    switch (vsf) {
    case JVO_EUCLIDEAN:   return jvo_score_from_raw(vsf, jvo_l2(a, b, n));
    case JVO_DOT_PRODUCT: return jvo_score_from_raw(vsf, jvo_dot(a, b, n));
    default:              return jvo_score_from_raw(vsf, jvo_cosine(a, b, n));
    }
}

/* DefaultVectorUtilSupport.sub(a, aOffset, b, bOffset, length) :282-288 */
/* SPECIFICATION (not a restatement of reference code) of the engine's MFMA tile form of full-resolution scoring,
 * jvector_amd/csrc/ed_body.h: fused multiply-adds in ascending k from +0 — what v_mfma_f32_32x32x2_f32 computes — for the
 * dot product and both squared norms, then the finishes written out there.  The reference's counterpart is its native
 * library's fused flavour (jvector_simd_kernels.cpp:208-286), which the reference itself only holds to 1e-4 of the scalar
 * order; tests hold this form to 1e-5 of jvo_compare and to bit equality with this function. */
float jvo_dense_compare(int vsf, const float *q, const float *v, int n)
{
    float dot = 0.0f, qn = 0.0f, vn = 0.0f;
    for (int k = 0; k < n; k++) {
        dot = fmaf(q[k], v[k], dot);
        qn = fmaf(q[k], q[k], qn);
        vn = fmaf(v[k], v[k], vn);
    }
    if (vsf == JVO_EUCLIDEAN) {
        float d2 = fmaf(-2.0f, dot, qn + vn);
        if (d2 < 0.0f) d2 = 0.0f;
        return 1.0f / (1.0f + d2);
    }
    if (vsf == JVO_COSINE) {
        float prod = qn * vn;
        dot = (float)((double)dot / sqrt((double)prod));
    }
    return (1.0f + dot) / 2.0f;
}

void jvo_dense_scan(int vsf, const float *queries, int Q, const float *vecs, int64_t n, int D, float *out)
{
    for (int q = 0; q < Q; q++)
        for (int64_t i = 0; i < n; i++)
            out[(size_t)q * n + i] = jvo_dense_compare(vsf, queries + (size_t)q * D, vecs + (size_t)i * D, D);
}

void jvo_sub(const float *a, const float *b, float *out, int n)
{
    for (int i = 0; i < n; i++) out[i] = a[i] - b[i];
}

/* ------------------------------------------------------------------------------------------
 * Rows 2, 5, 6 — ADC tables and lookups
 * ---------------------------------------------------------------------------------------- */

/* DefaultVectorUtilSupport.assembleAndSum :302-309 */
float jvo_assemble_and_sum(const float *data, int dataBase, const uint8_t *offs, int offsOff, int len)
{
    float sum = 0.0f;
    for (int i = 0; i < len; i++) sum += data[dataBase * i + (int)offs[i + offsOff]];
    return sum;
}

/* DefaultVectorUtilSupport.assembleAndSumPQ :311-339 (upper-triangular table) */
float jvo_assemble_and_sum_pq(const float *tri, int M, const uint8_t *c1v, int o1,
                              const uint8_t *c2v, int o2, int k)
{
    const int blockSize = k * (k + 1) / 2;
    float res = 0.0f;
    for (int i = 0; i < M; i++) {
        int c1 = c1v[i + o1], c2 = c2v[i + o2];
        int r = c1 < c2 ? c1 : c2;
        int c = c1 < c2 ? c2 : c1;
        int offsetRow = r * k - (r * (r - 1) / 2);
        int idxInBlock = offsetRow + (c - r);
        res += tri[i * blockSize + idxInBlock];
    }
    return res;
}

/* DefaultVectorUtilSupport.calculatePartialSums :351-365 — the *offset* (sequential) forms */
void jvo_calculate_partial_sums(const float *codebook, int cbIndex, int size, int k,
                                const float *query, int qoff, int vsf, float *out)
{
    int base = cbIndex * k;
    for (int i = 0; i < k; i++) {
        if (vsf == JVO_DOT_PRODUCT)
            out[base + i] = jvo_dot_off(codebook, i * size, query, qoff, size);
        else
            out[base + i] = jvo_l2_off(codebook, i * size, query, qoff, size);
    }
}

/* VectorUtilSupport.calculatePartialSelfMagnitudes (default) B/vector/VectorUtilSupport.java:137-142 */
void jvo_calculate_partial_self_magnitudes(const float *codebook, int cbIndex, int size, int k, float *out)
{
    int base = cbIndex * k;
    for (int i = 0; i < k; i++)
        out[base + i] = jvo_dot_off(codebook, i * size, codebook, i * size, size);
}

/* VectorUtilSupport.pqDecodedCosineSimilarity (default) :152-165 */
float jvo_pq_decoded_cosine(const uint8_t *enc, int encOff, int encLen, int k,
                            const float *partialSums, const float *aMagT, float bMag)
{
    float sum = 0.0f, aMag = 0.0f;
    for (int m = 0; m < encLen; ++m) {
        int idx = m * k + (int)enc[m + encOff];
        sum += partialSums[idx];
        aMag += aMagT[idx];
    }
    float prod = aMag * bMag;
    return (float)((double)sum / sqrt((double)prod));
}

/* ------------------------------------------------------------------------------------------
 * Row 3 — ProductQuantization  (B/quantization/ProductQuantization.java)
 * ---------------------------------------------------------------------------------------- */

/* getSubvectorSizesAndOffsets :535-550 */
void jvo_subvector_sizes_offsets(int D, int M, int *sizes, int *offsets)
{
    int baseSize = D / M, rem = D % M, off = 0;
    for (int i = 0; i < M; i++) {
        int size = baseSize + (i < rem ? 1 : 0);
        sizes[i] = size;
        offsets[i] = off;
        off += size;
    }
}

static const float *pq_codebook(const jvo_pq *pq, int m)
{
    size_t off = 0;
    for (int i = 0; i < m; i++) off += (size_t)pq->k * pq->sizes[i];
    return pq->codebooks + off;
}

/* closestCentroidIndex :507-520 — minDist = Float.MAX_VALUE, strict '<', first minimum wins */
static int closest_centroid(const float *vec, int voff, const float *cb, int size, int k)
{
    int index = 0;
    float minDist = 3.4028234663852886e+38f; /* Float.MAX_VALUE */
    for (int i = 0; i < k; i++) {
        float dist = jvo_l2_off(vec, voff, cb, i * size, size);
        if (dist < minDist) { minDist = dist; index = i; }
    }
    return index;
}

int jvo_closest_centroid(const jvo_pq *pq, const float *vec, int m)
{
    return closest_centroid(vec, pq->offsets[m], pq_codebook(pq, m), pq->sizes[m], pq->k);
}

/* encodeTo :439-449 (centre, then encodeUnweighted :422-426).  The anisotropic branch (anisotropicThreshold > -1,
 * SURVEY §8a row 4) is jvo_pq_encode_anisotropic below. */
void jvo_pq_encode(const jvo_pq *pq, const float *vec, uint8_t *dst)
{
    float *tmp = NULL;
    const float *v = vec;
    if (pq->centroid) {
        tmp = (float *)malloc(sizeof(float) * (size_t)pq->D);
        jvo_sub(vec, pq->centroid, tmp, pq->D);
        v = tmp;
    }
    size_t cboff = 0;
    for (int m = 0; m < pq->M; m++) {
        dst[m] = (uint8_t)closest_centroid(v, pq->offsets[m], pq->codebooks + cboff, pq->sizes[m], pq->k);
        cboff += (size_t)pq->k * pq->sizes[m];
    }
    free(tmp);
}

/* KMeansPlusPlusClusterer.computeParallelCostMultiplier :116-124 (double arithmetic, narrowed) */
float jvo_parallel_cost_multiplier(float threshold, int dimensions)
{
    double t = (double)threshold;
    double parallelCost = t * t;
    double perpendicularCost = (1 - parallelCost) / (dimensions - 1);
    double r = parallelCost / perpendicularCost;
    return (float)(r > 1.0 ? r : 1.0);  /* Math.max(1.0, r) */
}

/* encodeTo :439-449 with anisotropicThreshold > UNWEIGHTED -> encodeAnisotropic :269-306
 * (computeResiduals :384-399 + computeResidual :414-420, initializeToMinResidualNorms :364-379,
 * optimizeSingleSubspace :308-349, <= 10 sweeps).  centroidNormsSquared as the constructor builds it (:241-248). */
void jvo_pq_encode_anisotropic(const jvo_pq *pq, float threshold, const float *vec, uint8_t *dst)
{
    const int M = pq->M, k = pq->k;
    float *v = (float *)malloc(sizeof(float) * (size_t)pq->D);
    if (pq->centroid) jvo_sub(vec, pq->centroid, v, pq->D);
    else memcpy(v, vec, sizeof(float) * (size_t)pq->D);
    float *rns = (float *)malloc(sizeof(float) * (size_t)M * k);  /* residualNormSquared */
    float *par = (float *)malloc(sizeof(float) * (size_t)M * k);  /* parallelResidualComponent */
    const float inverseNorm = (float)(1.0 / sqrt((double)jvo_dot(v, v, pq->D)));
    size_t cboff = 0;
    for (int i = 0; i < M; i++) {
        const int len = pq->sizes[i];
        const float *x = v + pq->offsets[i];               /* getSubVector copies; same values */
        const float xNormSquared = jvo_dot(x, x, len);     /* full-vector form on the copy */
        const float *cb = pq->codebooks + cboff;
        for (int j = 0; j < k; j++) {
            const float cNormSquared = jvo_dot_off(cb, j * len, cb, j * len, len);
            const float cDotX = jvo_dot_off(cb, j * len, x, 0, len);
            const float two = 2 * cDotX;
            const float residualNormSquared = cNormSquared - two + xNormSquared;
            const float pes = cDotX - xNormSquared;
            rns[(size_t)i * k + j] = residualNormSquared;
            par[(size_t)i * k + j] = (pes * pes) * inverseNorm;
        }
        cboff += (size_t)k * len;
    }
    for (int i = 0; i < M; i++) {  /* initializeToMinResidualNorms: strict <, compared as double */
        int minIndex = -1;
        double minNormSquared = 1.7976931348623157e308;
        for (int j = 0; j < k; j++)
            if ((double)rns[(size_t)i * k + j] < minNormSquared) { minNormSquared = rns[(size_t)i * k + j]; minIndex = j; }
        dst[i] = (uint8_t)minIndex;
    }
    float parSum = 0.0f;
    for (int i = 0; i < M; i++) parSum += par[(size_t)i * k + dst[i]];
    const float pcm = jvo_parallel_cost_multiplier(threshold, pq->D);
    for (int iter = 0; iter < 10; iter++) {
        int changed = 0;
        for (int i = 0; i < M; i++) {
            const int oldIdx = dst[i];
            const float *R = rns + (size_t)i * k, *P = par + (size_t)i * k;
            const float oldRns = R[oldIdx], oldPar = P[oldIdx];
            float bestCostDelta = 0.0f, bestParSum = parSum;
            int bestIndex = oldIdx;
            for (int t = 0; t < k; t++) {
                if (t == oldIdx) continue;
                const float thisParSum = parSum - oldPar + P[t];
                const float parallelNormDelta = thisParSum * thisParSum - parSum * parSum;
                if (parallelNormDelta > 0) continue;
                const float residualNormDelta = R[t] - oldRns;
                const float perpendicularNormDelta = residualNormDelta - parallelNormDelta;
                const float costDelta = pcm * parallelNormDelta + perpendicularNormDelta;
                if (costDelta < bestCostDelta) { bestCostDelta = costDelta; bestIndex = t; bestParSum = thisParSum; }
            }
            if (bestIndex != oldIdx) { parSum = bestParSum; dst[i] = (uint8_t)bestIndex; changed = 1; }
        }
        if (!changed) break;
    }
    free(v); free(rns); free(par);
}

typedef struct { const jvo_pq *pq; const float *vecs; uint8_t *dst; int64_t lo, hi; } enc_job;
static void *enc_worker(void *p)
{
    enc_job *j = (enc_job *)p;
    for (int64_t i = j->lo; i < j->hi; i++)
        jvo_pq_encode(j->pq, j->vecs + i * j->pq->D, j->dst + i * j->pq->M);
    return NULL;
}

/* PQVectors.encodeAndBuild :137-149 — parallel forEach over ordinals; output ordinal-major */
void jvo_pq_encode_all(const jvo_pq *pq, const float *vecs, int64_t n, uint8_t *dst, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    pthread_t th[64];
    enc_job jobs[64];
    int64_t per = (n + nthreads - 1) / nthreads;
    int started = 0;
    for (int t = 0; t < nthreads; t++) {
        int64_t lo = t * per, hi = lo + per > n ? n : lo + per;
        if (lo >= hi) break;
        jobs[t] = (enc_job){pq, vecs, dst, lo, hi};
        pthread_create(&th[t], NULL, enc_worker, &jobs[t]);
        started++;
    }
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
}

/* decode :454-471 */
void jvo_pq_decode(const jvo_pq *pq, const uint8_t *code, float *dst)
{
    size_t cboff = 0;
    for (int m = 0; m < pq->M; m++) {
        int size = pq->sizes[m];
        memcpy(dst + pq->offsets[m], pq->codebooks + cboff + (size_t)code[m] * size, sizeof(float) * (size_t)size);
        cboff += (size_t)pq->k * size;
    }
    if (pq->centroid)
        for (int i = 0; i < pq->D; i++) dst[i] = dst[i] + pq->centroid[i];
}

/* createCodebookPartialSums :609-628 */
void jvo_pq_codebook_partial_sums(const jvo_pq *pq, int vsf, float *out)
{
    size_t idx = 0, cboff = 0;
    for (int m = 0; m < pq->M; m++) {
        int size = pq->sizes[m];
        const float *cb = pq->codebooks + cboff;
        for (int i = 0; i < pq->k; i++)
            for (int j = i; j < pq->k; j++)
                out[idx++] = (vsf == JVO_EUCLIDEAN) ? jvo_l2_off(cb, i * size, cb, j * size, size)
                                                    : jvo_dot_off(cb, i * size, cb, j * size, size);
        cboff += (size_t)pq->k * size;
    }
}

/* ImmutablePQVectors.diversityFunctionFor(node1, vsf).similarityTo(node2) :61-104 — via the triangular table
 * (VectorUtil.assembleAndSumPQ); cosine: sum / (float) Math.sqrt(norm1 * norm2), a FLOAT division (:88). */
float jvo_pq_diversity_score(const float *tri, int M, int k, int vsf, const uint8_t *code1, const uint8_t *code2)
{
    float sum = jvo_assemble_and_sum_pq(tri, M, code1, 0, code2, 0, k);
    if (vsf == JVO_DOT_PRODUCT) return (1.0f + sum) / 2.0f;
    if (vsf == JVO_EUCLIDEAN) return 1.0f / (1.0f + sum);
    float norm1 = jvo_assemble_and_sum_pq(tri, M, code1, 0, code1, 0, k);
    float norm2 = jvo_assemble_and_sum_pq(tri, M, code2, 0, code2, 0, k);
    float prod = norm1 * norm2;
    float cosine = sum / (float)sqrt((double)prod);
    return (1.0f + cosine) / 2.0f;
}

/* PQVectors.diversityFunctionFor (the MutablePQVectors path) :284-345 — straight from the codebooks; must equal the
 * table path bit for bit (same per-subspace dot / distance, same ascending-m sum). */
float jvo_pq_diversity_score_direct(const jvo_pq *pq, int vsf, const uint8_t *code1, const uint8_t *code2)
{
    float sum = 0.0f, norm1 = 0.0f, norm2 = 0.0f;
    size_t cboff = 0;
    for (int m = 0; m < pq->M; m++) {
        int len = pq->sizes[m];
        const float *cb = pq->codebooks + cboff;
        if (vsf == JVO_EUCLIDEAN) sum += jvo_l2_off(cb, code1[m] * len, cb, code2[m] * len, len);
        else if (vsf == JVO_DOT_PRODUCT) sum += jvo_dot_off(cb, code1[m] * len, cb, code2[m] * len, len);
        else {
            sum += jvo_dot_off(cb, code2[m] * len, cb, code1[m] * len, len);
            norm2 += jvo_dot_off(cb, code2[m] * len, cb, code2[m] * len, len);
            norm1 += jvo_dot_off(cb, code1[m] * len, cb, code1[m] * len, len);
        }
        cboff += (size_t)pq->k * len;
    }
    if (vsf == JVO_DOT_PRODUCT) return (1.0f + sum) / 2.0f;
    if (vsf == JVO_EUCLIDEAN) return 1.0f / (1.0f + sum);
    float prod = norm1 * norm2;
    float cosine = sum / (float)sqrt((double)prod);
    return (1.0f + cosine) / 2.0f;
}

/* VamanaDiversityProvider.retainDiverse (B/graph/diversity/VamanaDiversityProvider.java:43-78) + isDiverse (:82-96) — the
 * robust prune of Vamana construction — with the PQ diversity score above as scoreProvider.diversityScoreFunctionFor
 * (BuildScoreProvider.pqBuildScoreProvider, BuildScoreProvider.java:181-186).
 *   nodes / scores: the NodeArray (sorted by score descending by its owner), n entries;  selected: n bytes (the BitSet), cleared
 *   here;  returns nSelected, *short_edges = the return value of retainDiverse (NaN when the loop never ran).
 * Float details kept: `currentAlpha <= alpha + 1E-6` compares in double, `currentAlpha += 0.2f` and `score * alpha` are float,
 * shortEdges = nSelected / (float) maxDegree. */
int jvo_retain_diverse(const float *tri, int M, int k, int vsf, const uint8_t *codes, const int32_t *nodes, const float *scores,
                       int n, int maxDegree, int diverseBefore, float alpha, uint8_t *selected, double *short_edges)
{
    memset(selected, 0, (size_t)(n > 0 ? n : 0));
    for (int i = 0; i < (diverseBefore < maxDegree ? diverseBefore : maxDegree) && i < n; i++) selected[i] = 1;
    int nSelected = diverseBefore;
    double shortEdges = NAN;
    float currentAlpha = 1.0f;
    while ((double)currentAlpha <= (double)alpha + 1E-6 && nSelected < maxDegree) {
        for (int i = diverseBefore; i < n && nSelected < maxDegree; i++) {
            if (selected[i]) continue;
            const int cNode = nodes[i];
            const float cScore = scores[i];
            int diverse = 1;
            for (int j = 0; j < n; j++) {             /* selected.nextSetBit ascending */
                if (!selected[j]) continue;
                const int other = nodes[j];
                if (cNode == other) break;
                if (jvo_pq_diversity_score(tri, M, k, vsf, codes + (size_t)cNode * M, codes + (size_t)other * M) > cScore * currentAlpha) {
                    diverse = 0;
                    break;
                }
            }
            if (diverse) {
                selected[i] = 1;
                nSelected++;
            }
        }
        if (currentAlpha == 1.0f) shortEdges = nSelected / (float)maxDegree;
        currentAlpha += 0.2f;
    }
    if (short_edges) *short_edges = shortEdges;
    return nSelected;
}

/* ------------------------------------------------------------------------------------------
 * PQ training: KMeansPlusPlusClusterer (unweighted path) + ProductQuantization.compute / refine.
 * The reference draws from ThreadLocalRandom (unseedable), so its output is not reproducible; this restatement
 * substitutes a seeded splitmix64 stream per subspace (jvo_rng) and is otherwise operation for operation:
 *   chooseInitialCentroids :171-226, initializeAssignedPoints :232-240, updateAssignedPointsUnweighted :251-272,
 *   getNearestCluster :330-343, updateCentroidsUnweighted :360-372, cluster :131-150 (stop when <= 1 % changed),
 *   centroidOf :437-446, ProductQuantization.compute :109-139 / createCodebooks :487-495 (6 rounds) / refine :194-221.
 * The anisotropic k-means variants (:274-320, :380-432) are not restated.
 * ---------------------------------------------------------------------------------------- */
static uint64_t rng_next(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static int rng_int(uint64_t *s, int64_t bound) { return (int)((rng_next(s) >> 33) % (uint64_t)bound); }
static float rng_float(uint64_t *s) { return (float)(rng_next(s) >> 40) * (1.0f / 16777216.0f); }  /* 24 bits, as Random.nextFloat */
uint64_t jvo_kmeans_stream(uint64_t seed, int m) { return seed * 0x9E3779B97F4A7C15ULL + (uint64_t)m * 0xD1B54A32D192ED03ULL + 1; }

#define PT(i) (X + (size_t)(i) * stride + off)

void jvo_kmeans_pp_init(const float *X, int64_t n, int stride, int off, int len, int k, uint64_t *rng, float *C)
{
    float *dist = (float *)malloc(sizeof(float) * (size_t)n);
    for (int64_t i = 0; i < n; i++) dist[i] = 3.4028234663852886e38f;
    int64_t sel = rng_int(rng, n);
    for (int c = 0; c < k; c++) {
        if (c > 0) {
            float total = 0.0f;
            for (int64_t j = 0; j < n; j++) total += dist[j];
            float r = rng_float(rng) * total;
            sel = -1;
            for (int64_t j = 0; j < n; j++) {
                r -= dist[j];
                if ((double)r < 1e-6) { sel = j; break; }
            }
            if (sel == -1) sel = rng_int(rng, n);
        }
        memcpy(C + (size_t)c * len, PT(sel), sizeof(float) * (size_t)len);
        for (int64_t j = 0; j < n; j++) {
            float d = jvo_l2(PT(j), C + (size_t)c * len, len);  /* squareL2Distance(points[j], centroid): full-vector form */
            if (d < dist[j]) dist[j] = d;                          /* minInPlace (Math.min; no NaNs here) */
        }
    }
    free(dist);
}

static int km_nearest(const float *p, const float *C, int len, int k)
{
    float minDistance = 3.4028234663852886e38f;
    int nearest = 0;
    for (int i = 0; i < k; i++) {
        float d = jvo_l2_off(p, 0, C, i * len, len);
        if (d < minDistance) { minDistance = d; nearest = i; }
    }
    return nearest;
}

/* constructor (initializeAssignedPoints) + cluster(rounds, 0); returns the rounds actually run */
/* Matrix.invert (B/vector/Matrix.java:70-118): Gauss-Jordan with partial pivoting on an N x 2N augmented matrix, float */
static void km_invert(const float *A, int N, float *inv /* N*N */, float *aug /* N*2N scratch */)
{
    const int W = 2 * N;
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) { aug[i * W + j] = A[i * N + j]; aug[i * W + j + N] = (i == j) ? 1.0f : 0.0f; }
    for (int i = 0; i < N; i++) {
        int maxRow = i;
        for (int r = i + 1; r < N; r++)
            if (fabsf(aug[r * W + i]) > fabsf(aug[maxRow * W + i])) maxRow = r;
        if (maxRow != i)
            for (int j = 0; j < W; j++) { float t = aug[i * W + j]; aug[i * W + j] = aug[maxRow * W + j]; aug[maxRow * W + j] = t; }
        const float s = 1 / aug[i * W + i];
        for (int j = 0; j < W; j++) aug[i * W + j] = aug[i * W + j] * s;
        for (int r = 0; r < N; r++) {
            if (r == i) continue;
            const float factor = aug[r * W + i];
            for (int j = 0; j < W; j++) aug[r * W + j] = aug[r * W + j] + (-factor * aug[i * W + j]);
        }
    }
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) inv[i * N + j] = aug[i * W + j + N];
}

/* clusterOnceAnisotropic (KMeansPlusPlusClusterer.java:163-166): updateCentroidsAnisotropic :380-432 then
 * updateAssignedPointsAnisotropic :274-306 with weightedDistance :311-320.  Returns the number of points that moved. */
static int64_t km_cluster_once_anisotropic(const float *X, int64_t n, int stride, int off, int len, int k, float *C, int *assign,
                                           float threshold, uint64_t *rng)
{
    const float pcm = jvo_parallel_cost_multiplier(threshold, len);
    const float ocm = 1.0f / pcm;
    float *mean = (float *)malloc(sizeof(float) * (size_t)len);
    float *outer = (float *)malloc(sizeof(float) * (size_t)len * len);
    float *inv = (float *)malloc(sizeof(float) * (size_t)len * len);
    float *aug = (float *)malloc(sizeof(float) * (size_t)len * 2 * len);
    for (int c = 0; c < k; c++) {
        int64_t cnt = 0;
        for (int d = 0; d < len; d++) mean[d] = 0.0f;
        for (int d = 0; d < len * len; d++) outer[d] = 0.0f;
        for (int64_t j = 0; j < n; j++) {  /* pointsByCluster lists are in point order */
            if (assign[j] != c) continue;
            const float *p = PT(j);
            cnt++;
            for (int d = 0; d < len; d++) mean[d] = mean[d] + p[d];
            const float denom = jvo_dot(p, p, len);
            if (denom > 0) {
                const float invd = 1.0f / denom;
                for (int r = 0; r < len; r++)
                    for (int d = 0; d < len; d++) outer[r * len + d] = outer[r * len + d] + (p[d] * p[r]) * invd;
            }
        }
        if (cnt == 0) {
            memcpy(C + (size_t)c * len, PT(rng_int(rng, n)), sizeof(float) * (size_t)len);
            continue;
        }
        const float sc = (1 - ocm) / (float)cnt, invc = 1.0f / (float)cnt;
        for (int d = 0; d < len * len; d++) outer[d] = outer[d] * sc;
        for (int d = 0; d < len; d++) mean[d] = mean[d] * invc;
        for (int d = 0; d < len; d++) outer[d * len + d] = outer[d * len + d] + ocm;
        km_invert(outer, len, inv, aug);
        for (int r = 0; r < len; r++) C[(size_t)c * len + r] = jvo_dot(inv + r * len, mean, len);  /* Matrix.multiply */
    }
    float *cnorm = (float *)malloc(sizeof(float) * (size_t)k);
    for (int c = 0; c < k; c++) cnorm[c] = jvo_dot_off(C, c * len, C, c * len, len);
    int64_t changed = 0;
    for (int64_t i = 0; i < n; i++) {
        const float *x = PT(i);
        const float xNorm = jvo_dot(x, x, len);
        int index = assign[i];
        float minDist = 3.4028234663852886e38f;
        for (int j = 0; j < k; j++) {
            const float cDotX = jvo_dot_off(C, j * len, x, 0, len);
            const float pes = cDotX - xNorm;
            const float two = 2 * cDotX;
            const float rsn = cnorm[j] - two + xNorm;
            const float parErr = pes * pes;
            const float perp = rsn - parErr;
            const float dist = pcm * parErr + perp;
            if (dist < minDist) { minDist = dist; index = j; }
        }
        if (index != assign[i]) { changed++; assign[i] = index; }
    }
    free(mean); free(outer); free(inv); free(aug); free(cnorm);
    return changed;
}

/* aniso_rounds > 0 with threshold > -1: cluster(rounds, aniso_rounds) :131-150 */
int jvo_kmeans_lloyd_aniso(const float *X, int64_t n, int stride, int off, int len, int k, float *C, int rounds, int aniso_rounds,
                           float threshold, uint64_t *rng);

int jvo_kmeans_lloyd(const float *X, int64_t n, int stride, int off, int len, int k, float *C, int rounds, uint64_t *rng)
{
    return jvo_kmeans_lloyd_aniso(X, n, stride, off, len, k, C, rounds, 0, -1.0f, rng);
}

int jvo_kmeans_lloyd_aniso(const float *X, int64_t n, int stride, int off, int len, int k, float *C, int rounds, int aniso_rounds,
                           float threshold, uint64_t *rng)
{
    float *nums = (float *)calloc((size_t)k * len, sizeof(float));
    int *denoms = (int *)calloc((size_t)k, sizeof(int));
    int *assign = (int *)malloc(sizeof(int) * (size_t)n);
    for (int64_t i = 0; i < n; i++) {
        int a = km_nearest(PT(i), C, len, k);
        denoms[a]++;
        for (int d = 0; d < len; d++) nums[(size_t)a * len + d] = nums[(size_t)a * len + d] + PT(i)[d];
        assign[i] = a;
    }
    int it = 0;
    for (; it < rounds; it++) {
        for (int c = 0; c < k; c++) {  /* updateCentroidsUnweighted */
            if (denoms[c] == 0) memcpy(C + (size_t)c * len, PT(rng_int(rng, n)), sizeof(float) * (size_t)len);
            else {
                float inv = 1.0f / denoms[c];
                for (int d = 0; d < len; d++) C[(size_t)c * len + d] = nums[(size_t)c * len + d] * inv;
            }
        }
        int64_t changed = 0;
        for (int64_t i = 0; i < n; i++) {  /* updateAssignedPointsUnweighted */
            int o = assign[i], a = km_nearest(PT(i), C, len, k);
            if (a != o) {
                denoms[o]--;
                for (int d = 0; d < len; d++) nums[(size_t)o * len + d] = nums[(size_t)o * len + d] - PT(i)[d];
                denoms[a]++;
                for (int d = 0; d < len; d++) nums[(size_t)a * len + d] = nums[(size_t)a * len + d] + PT(i)[d];
                assign[i] = a;
                changed++;
            }
        }
        if ((double)changed <= 0.01 * (double)n) { it++; break; }
    }
    for (int a = 0; a < aniso_rounds; a++) {  /* optionally refine with anisotropic clustering */
        int64_t changed = km_cluster_once_anisotropic(X, n, stride, off, len, k, C, assign, threshold, rng);
        if ((double)changed <= 0.01 * (double)n) break;
    }
    free(nums); free(denoms); free(assign);
    return it;
}
#undef PT

/* centroidOf: VectorUtil.sum(List) (point order) then scale(1.0f / n) */
void jvo_centroid_of(const float *X, int64_t n, int D, float *out)
{
    for (int d = 0; d < D; d++) out[d] = 0.0f;
    for (int64_t i = 0; i < n; i++)
        for (int d = 0; d < D; d++) out[d] = out[d] + X[(size_t)i * D + d];
    float inv = 1.0f / (float)n;
    for (int d = 0; d < D; d++) out[d] = out[d] * inv;
}

/* ProductQuantization.compute (unweighted): codebooks [sum_m k*size_m], centroid [D] (written iff globallyCenter) */
void jvo_pq_train(const float *X, int64_t n, int D, int M, int k, int globallyCenter, uint64_t seed, int rounds,
                  float *codebooks, float *centroid, int *rounds_run /* [M] or NULL */)
{
    jvo_pq_train_aniso(X, n, D, M, k, globallyCenter, -1.0f, seed, rounds, codebooks, centroid, rounds_run);
}

/* ProductQuantization.compute with an anisotropic threshold: createCodebooks runs cluster(6, threshold > -1 ? 6 : 0) :487-495 */
void jvo_pq_train_aniso(const float *X, int64_t n, int D, int M, int k, int globallyCenter, float threshold, uint64_t seed, int rounds,
                        float *codebooks, float *centroid, int *rounds_run /* [M] or NULL */)
{
    int *sizes = (int *)malloc(sizeof(int) * (size_t)M), *offs = (int *)malloc(sizeof(int) * (size_t)M);
    jvo_subvector_sizes_offsets(D, M, sizes, offs);
    float *Xc = (float *)malloc(sizeof(float) * (size_t)n * D);
    if (globallyCenter) {
        jvo_centroid_of(X, n, D, centroid);
        for (int64_t i = 0; i < n; i++) jvo_sub(X + (size_t)i * D, centroid, Xc + (size_t)i * D, D);
    } else memcpy(Xc, X, sizeof(float) * (size_t)n * D);
    size_t cboff = 0;
    for (int m = 0; m < M; m++) {
        uint64_t rng = jvo_kmeans_stream(seed, m);
        jvo_kmeans_pp_init(Xc, n, D, offs[m], sizes[m], k, &rng, codebooks + cboff);
        int r = jvo_kmeans_lloyd_aniso(Xc, n, D, offs[m], sizes[m], k, codebooks + cboff, rounds, threshold > -1.0f ? rounds : 0,
                                       threshold, &rng);
        if (rounds_run) rounds_run[m] = r;
        cboff += (size_t)k * sizes[m];
    }
    free(Xc); free(sizes); free(offs);
}

/* ProductQuantization.refine (unweighted): starts from pq's codebooks, `rounds` Lloyd rounds on new data */
void jvo_pq_refine(const jvo_pq *pq, const float *X, int64_t n, int rounds, uint64_t seed, float *codebooks)
{
    jvo_pq_refine_aniso(pq, -1.0f, X, n, rounds, seed, codebooks);
}

/* ProductQuantization.refine :194-221: cluster(threshold == UNWEIGHTED ? rounds : 0, threshold == UNWEIGHTED ? 0 : rounds) */
void jvo_pq_refine_aniso(const jvo_pq *pq, float threshold, const float *X, int64_t n, int rounds, uint64_t seed, float *codebooks)
{
    const int D = pq->D;
    float *Xc = (float *)malloc(sizeof(float) * (size_t)n * D);
    if (pq->centroid) for (int64_t i = 0; i < n; i++) jvo_sub(X + (size_t)i * D, pq->centroid, Xc + (size_t)i * D, D);
    else memcpy(Xc, X, sizeof(float) * (size_t)n * D);
    size_t cboff = 0;
    for (int m = 0; m < pq->M; m++) {
        size_t cnt = (size_t)pq->k * pq->sizes[m];
        memcpy(codebooks + cboff, pq->codebooks + cboff, sizeof(float) * cnt);
        uint64_t rng = jvo_kmeans_stream(seed, m);
        jvo_kmeans_lloyd_aniso(Xc, n, D, pq->offsets[m], pq->sizes[m], pq->k, codebooks + cboff, threshold > -1.0f ? 0 : rounds,
                               threshold > -1.0f ? rounds : 0, threshold, &rng);
        cboff += cnt;
    }
    free(Xc);
}

/* ------------------------------------------------------------------------------------------
 * PQDecoder / FusedPQDecoder set-up and per-node scores
 * ---------------------------------------------------------------------------------------- */

/* cpu_baseline switch (bench.py only): 0 = the scalar checker arithmetic of this file (default, what every parity test
 * compares against); non-zero = the SIMD restatement of the reference's native kernels (jv_oracle_simd.c) inside the
 * SEARCH entry points below (decoder tables, per-node ADC scores, exact rerank).  Set before any search thread starts. */
static int g_simd = 0;
int jvo_set_simd(int on)
{
    g_simd = on ? jvs_tier() : 0;
    return g_simd;
}
static inline float adc_score_x(int vsf, int M, int k, const float *lut, const float *amag, float bmag, const uint8_t *code)
{
    return g_simd ? jvs_adc_score(vsf, M, k, lut, amag, bmag, code) : jvo_adc_score(vsf, M, k, lut, amag, bmag, code);
}
static inline float compare_x(int vsf, const float *a, const float *b, int n)
{
    return g_simd ? jvs_compare(vsf, a, b, n) : jvo_compare(vsf, a, b, n);
}
/* the reranker of the search entry points: full-resolution rows (view.rerankerFor with INLINE_VECTORS) or, after
 * jvo_set_nvq_reranker (jv_nvq.c), the NVQ feature's score function (B/graph/disk/feature/NVQ.java:96-110); `vecs` is then
 * only the "rerank at all" flag */
static inline float rerank_x(int vsf, const float *query, const float *vecs, int32_t id, int D)
{
    if (jvo_nvq_reranker_active()) return jvo_nvq_rerank_score(vsf, query, id);
    return compare_x(vsf, query, vecs + (size_t)id * D, D);
}

static void build_tables(const jvo_pq *pq, const float *cq, int lutVsf, float *lut, float *amag)
{
    size_t cboff = 0;
    for (int m = 0; m < pq->M; m++) {
        int size = pq->sizes[m];
        if (g_simd) jvs_calculate_partial_sums(pq->codebooks + cboff, m, size, pq->k, cq, pq->offsets[m], lutVsf, lut);
        else jvo_calculate_partial_sums(pq->codebooks + cboff, m, size, pq->k, cq, pq->offsets[m], lutVsf, lut);
        if (amag) jvo_calculate_partial_self_magnitudes(pq->codebooks + cboff, m, size, pq->k, amag);
        cboff += (size_t)pq->k * size;
    }
}

/* PQDecoder.CachingDecoder ctor B/quantization/PQDecoder.java:41-54 (dot / L2) and
 * PQDecoder.CosineDecoder ctor :88-122 (dot LUT + aMagnitude table + bMagnitude =
 * dotProduct(centeredQuery, centeredQuery) using the FULL-vector form, :121). */
void jvo_pqdecoder_init(const jvo_pq *pq, const float *query, int vsf, float *lut, float *amag, float *bmag)
{
    float *tmp = NULL;
    const float *cq = query;
    if (pq->centroid) {
        tmp = (float *)malloc(sizeof(float) * (size_t)pq->D);
        jvo_sub(query, pq->centroid, tmp, pq->D);
        cq = tmp;
    }
    if (vsf == JVO_COSINE) {
        build_tables(pq, cq, JVO_DOT_PRODUCT, lut, amag);
        if (bmag) *bmag = jvo_dot(cq, cq, pq->D);
    } else {
        build_tables(pq, cq, vsf, lut, NULL);
    }
    free(tmp);
}

/* FusedPQDecoder ctor B/quantization/FusedPQDecoder.java:49-77 (dot / L2) and CosineDecoder ctor
 * :146-192: same tables, but queryMagnitudeSquared = sum over subspaces of the OFFSET-form
 * dotProduct(centeredQuery, off, centeredQuery, off, size) accumulated in a float (:188). */
void jvo_fuseddecoder_init(const jvo_pq *pq, const float *query, int vsf, float *lut, float *amag, float *bmag)
{
    float *tmp = NULL;
    const float *cq = query;
    if (pq->centroid) {
        tmp = (float *)malloc(sizeof(float) * (size_t)pq->D);
        jvo_sub(query, pq->centroid, tmp, pq->D);
        cq = tmp;
    }
    if (vsf == JVO_COSINE) {
        build_tables(pq, cq, JVO_DOT_PRODUCT, lut, amag);
        float qm = 0.0f;
        for (int m = 0; m < pq->M; m++)
            qm += jvo_dot_off(cq, pq->offsets[m], cq, pq->offsets[m], pq->sizes[m]);
        if (bmag) *bmag = qm;
    } else {
        build_tables(pq, cq, vsf, lut, NULL);
    }
    free(tmp);
}

/* PQDecoder.{DotProduct,Euclidean,Cosine}Decoder.similarityTo :65-80,124-135 and
 * FusedPQDecoder.similarityToNeighbor :104-111,206-213 (identical arithmetic). */
float jvo_adc_score(int vsf, int M, int k, const float *lut, const float *amag, float bmag, const uint8_t *code)
{
    if (vsf == JVO_COSINE)
        return jvo_score_from_raw(vsf, jvo_pq_decoded_cosine(code, 0, M, k, lut, amag, bmag));
    return jvo_score_from_raw(vsf, jvo_assemble_and_sum(lut, k, code, 0, M));
}

void jvo_adc_scores(int vsf, int M, int k, const float *lut, const float *amag, float bmag,
                    const uint8_t *codes, const int32_t *ord, int64_t n, float *scores)
{
    for (int64_t i = 0; i < n; i++) {
        int64_t o = ord ? (int64_t)ord[i] : i;
        scores[i] = jvo_adc_score(vsf, M, k, lut, amag, bmag, codes + o * M);
    }
}

/* PQVectors.scoreFunctionFor B/quantization/PQVectors.java:222-280 (direct, table-free path) */
float jvo_pq_direct_score(const jvo_pq *pq, const float *query, int vsf, const uint8_t *code)
{
    float *tmp = NULL;
    const float *cq = query;
    if (pq->centroid) {
        tmp = (float *)malloc(sizeof(float) * (size_t)pq->D);
        jvo_sub(query, pq->centroid, tmp, pq->D);
        cq = tmp;
    }
    float result;
    size_t cboff = 0;
    if (vsf == JVO_DOT_PRODUCT) {
        float dp = 0.0f;
        for (int m = 0; m < pq->M; m++) {
            int len = pq->sizes[m];
            dp += jvo_dot_off(pq->codebooks + cboff, code[m] * len, cq, pq->offsets[m], len);
            cboff += (size_t)pq->k * len;
        }
        result = (1.0f + dp) / 2.0f;
    } else if (vsf == JVO_COSINE) {
        float norm1 = jvo_dot(cq, cq, pq->D), sum = 0.0f, norm2 = 0.0f;
        for (int m = 0; m < pq->M; m++) {
            int len = pq->sizes[m];
            const float *cb = pq->codebooks + cboff;
            sum += jvo_dot_off(cb, code[m] * len, cq, pq->offsets[m], len);
            norm2 += jvo_dot_off(cb, code[m] * len, cb, code[m] * len, len);
            cboff += (size_t)pq->k * len;
        }
        float prod = norm1 * norm2;
        float cosine = sum / (float)sqrt((double)prod);
        result = (1.0f + cosine) / 2.0f;
    } else {
        float sum = 0.0f;
        for (int m = 0; m < pq->M; m++) {
            int len = pq->sizes[m];
            sum += jvo_l2_off(pq->codebooks + cboff, code[m] * len, cq, pq->offsets[m], len);
            cboff += (size_t)pq->k * len;
        }
        result = 1.0f / (1.0f + sum);
    }
    free(tmp);
    return result;
}

/* ------------------------------------------------------------------------------------------
 * Row 9 — NodeQueue order  (B/graph/NodeQueue.java:125-137, B/util/NumericUtils.java:49-65)
 * ---------------------------------------------------------------------------------------- */

int32_t jvo_float_to_sortable_int(float v)
{
    int32_t bits;
    if (v != v) bits = 0x7fc00000; /* Float.floatToIntBits canonicalises NaN */
    else memcpy(&bits, &v, 4);
    return bits ^ ((bits >> 31) & 0x7fffffff);
}

float jvo_sortable_int_to_float(int32_t e)
{
    int32_t bits = e ^ ((e >> 31) & 0x7fffffff);
    float v;
    memcpy(&v, &bits, 4);
    return v;
}

/* NodeQueue.encode with MAX_HEAP order (identity): (sortableInt(score) << 32) | (0xFFFFFFFF & ~node) */
int64_t jvo_nodequeue_encode(int32_t node, float score)
{
    return (int64_t)(((uint64_t)(uint32_t)jvo_float_to_sortable_int(score)) << 32) |
           (int64_t)(0xFFFFFFFFULL & (uint64_t)(uint32_t)(~node));
}

static int cmp_desc_i64(const void *x, const void *y)
{
    int64_t a = *(const int64_t *)x, b = *(const int64_t *)y;
    return a < b ? 1 : (a > b ? -1 : 0);
}

/* Best-first top-k under the NodeQueue total order: higher score first, ties -> smaller node id.
 * (GraphSearcher.reranking emits results best-first, B/graph/GraphSearcher.java:495-506.) */
int jvo_topk(const int32_t *ids, const float *scores, int64_t n, int k, int32_t *out_ids, float *out_scores)
{
    int64_t *keys = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; i++)
        keys[i] = jvo_nodequeue_encode(ids ? ids[i] : (int32_t)i, scores[i]);
    qsort(keys, (size_t)n, sizeof(int64_t), cmp_desc_i64);
    int cnt = (int)(n < k ? n : k);
    for (int i = 0; i < cnt; i++) {
        out_ids[i] = (int32_t)~(uint32_t)(keys[i] & 0xFFFFFFFFLL);
        out_scores[i] = jvo_sortable_int_to_float((int32_t)(keys[i] >> 32));
    }
    free(keys);
    return cnt;
}

/* ------------------------------------------------------------------------------------------
 * PQVectors.PQLayout  (B/quantization/PQVectors.java:515-540) — int arithmetic as in Java
 * ---------------------------------------------------------------------------------------- */
static int highest_one_bit(int v)
{
    if (v <= 0) return v == 0 ? 0 : (int)0x80000000;
    int r = 1;
    while ((v >>= 1) != 0) r <<= 1;
    return r;
}

int jvo_pq_layout_compute(int vectorCount, int compressedDimension, jvo_pq_layout *o)
{
    if (vectorCount <= 0) return -1;          /* IllegalArgumentException :518 */
    if (compressedDimension <= 0) return -2;  /* IllegalArgumentException :523 */
    /* Java int arithmetic wraps; do the shifts/multiplies in uint32 to get the same bits */
    int32_t layoutBytesPerVector =
        compressedDimension == 1 ? 1 : (int32_t)((uint32_t)highest_one_bit(compressedDimension - 1) << 1);
    int addressable = 2147483647 / layoutBytesPerVector;
    o->fullChunkVectors = vectorCount < addressable ? vectorCount : addressable;
    o->lastChunkVectors = vectorCount % o->fullChunkVectors;
    o->fullChunkBytes = (int32_t)((uint32_t)o->fullChunkVectors * (uint32_t)compressedDimension);
    o->lastChunkBytes = (int32_t)((uint32_t)o->lastChunkVectors * (uint32_t)compressedDimension);
    o->fullSizeChunks = vectorCount / o->fullChunkVectors;
    o->totalChunks = o->fullSizeChunks + (o->lastChunkVectors == 0 ? 0 : 1);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * ProductQuantization.load :649-693 / write :560-599 — big-endian (B/disk/IndexWriter.java:36-42)
 * ---------------------------------------------------------------------------------------- */
#define PQ_MAGIC 0x75EC4012

static int32_t rd_i32(const uint8_t *p)
{
    return (int32_t)(((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]);
}
static float rd_f32(const uint8_t *p)
{
    int32_t b = rd_i32(p);
    float f;
    memcpy(&f, &b, 4);
    return f;
}
static void wr_i32(uint8_t *p, int32_t v)
{
    p[0] = (uint8_t)((uint32_t)v >> 24); p[1] = (uint8_t)((uint32_t)v >> 16);
    p[2] = (uint8_t)((uint32_t)v >> 8);  p[3] = (uint8_t)v;
}
static void wr_f32(uint8_t *p, float f)
{
    int32_t b;
    memcpy(&b, &f, 4);
    wr_i32(p, b);
}

int jvo_pq_parse(const uint8_t *buf, size_t len, int *version, int *D, int *M, int *k,
                 int *centroidLen, float *aniso, int *sizes, int maxM,
                 float *centroid, float *codebooks, size_t codebookCap, size_t *consumed)
{
    size_t p = 0;
#define NEED(nb) do { if (p + (nb) > len) return -1; } while (0)
    NEED(4);
    int32_t maybeMagic = rd_i32(buf + p); p += 4;
    int ver, gcl;
    if (maybeMagic != PQ_MAGIC) { ver = 0; gcl = maybeMagic; }
    else { NEED(8); ver = rd_i32(buf + p); p += 4; gcl = rd_i32(buf + p); p += 4; }
    *version = ver;
    *centroidLen = gcl;
    if (gcl > 0) {
        NEED((size_t)gcl * 4);
        for (int i = 0; i < gcl; i++) { if (centroid) centroid[i] = rd_f32(buf + p); p += 4; }
    }
    NEED(4);
    int m = rd_i32(buf + p); p += 4;
    if (m <= 0 || m > maxM) return -2;
    *M = m;
    int dim = 0;
    NEED((size_t)m * 4);
    for (int i = 0; i < m; i++) { sizes[i] = rd_i32(buf + p); p += 4; dim += sizes[i]; }
    *D = dim;
    if (ver < 3) *aniso = -1.0f; /* UNWEIGHTED, KMeansPlusPlusClusterer.java:41 */
    else { NEED(4); *aniso = rd_f32(buf + p); p += 4; }
    NEED(4);
    int clusters = rd_i32(buf + p); p += 4;
    *k = clusters;
    size_t total = 0;
    for (int i = 0; i < m; i++) total += (size_t)clusters * (size_t)sizes[i];
    if (total > codebookCap) return -3;
    NEED(total * 4);
    for (size_t i = 0; i < total; i++) { codebooks[i] = rd_f32(buf + p); p += 4; }
    if (consumed) *consumed = p;
    return 0;
#undef NEED
}

size_t jvo_pq_serialize(int version, int D, int M, int k, const int *sizes, const float *centroid,
                        float aniso, const float *codebooks, uint8_t *out, size_t cap)
{
    size_t total = 0;
    for (int i = 0; i < M; i++) total += (size_t)k * (size_t)sizes[i];
    size_t need = (version >= 3 ? 8 : 0) + 4 + (centroid ? (size_t)D * 4 : 0) + 4 + (size_t)M * 4 +
                  (version >= 3 ? 4 : 0) + 4 + total * 4;
    if (need > cap) return 0;
    size_t p = 0;
    if (version >= 3) { wr_i32(out + p, PQ_MAGIC); p += 4; wr_i32(out + p, version); p += 4; }
    if (!centroid) { wr_i32(out + p, 0); p += 4; }
    else {
        wr_i32(out + p, D); p += 4;
        for (int i = 0; i < D; i++) { wr_f32(out + p, centroid[i]); p += 4; }
    }
    wr_i32(out + p, M); p += 4;
    for (int i = 0; i < M; i++) { wr_i32(out + p, sizes[i]); p += 4; }
    if (version >= 3) { wr_f32(out + p, aniso); p += 4; }
    wr_i32(out + p, k); p += 4;
    for (size_t i = 0; i < total; i++) { wr_f32(out + p, codebooks[i]); p += 4; }
    return p;
}

/* NC/tests/test_helpers.cpp:78-87 — the reference's deterministic known-answer generator */
void jvo_make_vec(float *v, size_t n, float seed)
{
    for (size_t i = 0; i < n; ++i) {
        v[i] = seed * (1.0f + (float)(i % 7) * 0.13f);
        if (i % 3 == 0) v[i] = -v[i];
        v[i] += 0.5f;
    }
}

/* ------------------------------------------------------------------------------------------
 * CPU baseline driver ("port"): the two-pass flat search the GPU bench times, restated with the
 * oracle's per-candidate arithmetic, one query per worker thread (the reference parallelises over
 * queries: ThroughputBenchmark's parallel stream, EX/benchmarks/ThroughputBenchmark.java:146-231).
 * Bounded-heap top-k replaces the full sort of jvo_topk (same order, NodeQueue keys).
 * ---------------------------------------------------------------------------------------- */
static void heap_sift_down(int64_t *h, int n, int i)
{
    for (;;) {
        int l = 2 * i + 1, r = l + 1, s = i;
        if (l < n && h[l] < h[s]) s = l;
        if (r < n && h[r] < h[s]) s = r;
        if (s == i) return;
        int64_t t = h[i]; h[i] = h[s]; h[s] = t;
        i = s;
    }
}

/* min-heap of the k largest keys (BoundedLongHeap.push semantics: reject value < top) */
static int topk_heap_push(int64_t *h, int size, int k, int64_t key)
{
    if (size < k) {
        int i = size++;
        h[i] = key;
        while (i > 0 && h[(i - 1) / 2] > h[i]) {
            int p = (i - 1) / 2;
            int64_t t = h[i]; h[i] = h[p]; h[p] = t;
            i = p;
        }
        return size;
    }
    if (key < h[0]) return size;
    h[0] = key;
    heap_sift_down(h, size, 0);
    return size;
}

typedef struct {
    const jvo_pq *pq;
    const uint8_t *codes;
    const float *vecs;   /* N x D, may be NULL (no rerank) */
    int64_t n;
    const float *queries;
    int q_lo, q_hi, vsf, topK, rerankK;
    int32_t *out_ids;
    float *out_scores;
} flat_job;

static void *flat_worker(void *arg)
{
    flat_job *j = (flat_job *)arg;
    const jvo_pq *pq = j->pq;
    const int M = pq->M, k = pq->k, D = pq->D;
    float *lut = (float *)malloc(sizeof(float) * (size_t)M * k);
    float *amag = (float *)malloc(sizeof(float) * (size_t)M * k);
    const int k1 = (j->vecs && j->rerankK > 0) ? j->rerankK : j->topK;
    int64_t *heap = (int64_t *)malloc(sizeof(int64_t) * (size_t)k1);
    int64_t *heap2 = (int64_t *)malloc(sizeof(int64_t) * (size_t)j->topK);
    for (int q = j->q_lo; q < j->q_hi; q++) {
        const float *query = j->queries + (size_t)q * D;
        float bmag = 0.0f;
        /* partialSquaredMagnitudes is cached per ProductQuantization (ProductQuantization.java:75,238) */
        const float *am = (j->vsf == JVO_COSINE && pq->self_magnitudes) ? pq->self_magnitudes : amag;
        jvo_pqdecoder_init(pq, query, j->vsf, lut, am == amag ? amag : NULL, &bmag);
        int size = 0;
        for (int64_t i = 0; i < j->n; i++) {
            float s = adc_score_x(j->vsf, M, k, lut, am, bmag, j->codes + i * M);
            size = topk_heap_push(heap, size, k1, jvo_nodequeue_encode((int32_t)i, s));
        }
        int64_t *res = heap;
        int rsize = size;
        if (j->vecs && j->rerankK > 0) {
            int s2 = 0;
            for (int c = 0; c < size; c++) {
                int32_t id = (int32_t)~(uint32_t)(heap[c] & 0xFFFFFFFFLL);
                float ex = rerank_x(j->vsf, query, j->vecs, id, D);
                s2 = topk_heap_push(heap2, s2, j->topK, jvo_nodequeue_encode(id, ex));
            }
            res = heap2;
            rsize = s2;
        }
        qsort(res, (size_t)rsize, sizeof(int64_t), cmp_desc_i64);
        for (int c = 0; c < j->topK; c++) {
            if (c < rsize) {
                j->out_ids[(size_t)q * j->topK + c] = (int32_t)~(uint32_t)(res[c] & 0xFFFFFFFFLL);
                j->out_scores[(size_t)q * j->topK + c] = jvo_sortable_int_to_float((int32_t)(res[c] >> 32));
            } else {
                j->out_ids[(size_t)q * j->topK + c] = -1;
                j->out_scores[(size_t)q * j->topK + c] = -INFINITY;
            }
        }
    }
    free(lut); free(amag); free(heap); free(heap2);
    return NULL;
}

void jvo_search_flat(const jvo_pq *pq, const uint8_t *codes, const float *vecs, int64_t n, const float *queries,
                     int Q, int vsf, int topK, int rerankK, int32_t *out_ids, float *out_scores, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    if (nthreads > Q) nthreads = Q > 0 ? Q : 1;
    pthread_t th[64];
    flat_job jobs[64];
    int per = (Q + nthreads - 1) / nthreads, started = 0;
    for (int t = 0; t < nthreads; t++) {
        int lo = t * per, hi = lo + per > Q ? Q : lo + per;
        if (lo >= hi) break;
        jobs[t] = (flat_job){pq, codes, vecs, n, queries, lo, hi, vsf, topK, rerankK, out_ids, out_scores};
        pthread_create(&th[t], NULL, flat_worker, &jobs[t]);
        started++;
    }
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
}

/* Exact rerank of pre-gathered candidates (NodeQueue.rerank's scoring, B/graph/NodeQueue.java:160-195, with the
 * deterministic (score desc, id asc) order): cand_vecs holds Q x R rows of D floats, cand_ids Q x R ids (-1 = skip). */
typedef struct {
    const float *queries, *cand_vecs; const int32_t *cand_ids;
    int q_lo, q_hi, R, D, vsf, topK; int32_t *out_ids; float *out_scores;
} rr_job;

static void *rr_worker(void *arg)
{
    rr_job *j = (rr_job *)arg;
    int64_t *heap = (int64_t *)malloc(sizeof(int64_t) * (size_t)j->topK);
    for (int q = j->q_lo; q < j->q_hi; q++) {
        int size = 0;
        for (int c = 0; c < j->R; c++) {
            int32_t id = j->cand_ids[(size_t)q * j->R + c];
            if (id < 0) continue;
            float ex = compare_x(j->vsf, j->queries + (size_t)q * j->D,
                                   j->cand_vecs + ((size_t)q * j->R + c) * j->D, j->D);
            size = topk_heap_push(heap, size, j->topK, jvo_nodequeue_encode(id, ex));
        }
        qsort(heap, (size_t)size, sizeof(int64_t), cmp_desc_i64);
        for (int c = 0; c < j->topK; c++) {
            j->out_ids[(size_t)q * j->topK + c] = c < size ? (int32_t)~(uint32_t)(heap[c] & 0xFFFFFFFFLL) : -1;
            j->out_scores[(size_t)q * j->topK + c] =
                c < size ? jvo_sortable_int_to_float((int32_t)(heap[c] >> 32)) : -INFINITY;
        }
    }
    free(heap);
    return NULL;
}

void jvo_rerank(const float *queries, const float *cand_vecs, const int32_t *cand_ids, int Q, int R, int D, int vsf,
                int topK, int32_t *out_ids, float *out_scores, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    if (nthreads > Q) nthreads = Q > 0 ? Q : 1;
    pthread_t th[64];
    rr_job jobs[64];
    int per = (Q + nthreads - 1) / nthreads, started = 0;
    for (int t = 0; t < nthreads; t++) {
        int lo = t * per, hi = lo + per > Q ? Q : lo + per;
        if (lo >= hi) break;
        jobs[t] = (rr_job){queries, cand_vecs, cand_ids, lo, hi, R, D, vsf, topK, out_ids, out_scores};
        pthread_create(&th[t], NULL, rr_worker, &jobs[t]);
        started++;
    }
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
}

/* ------------------------------------------------------------------------------------------
 * GraphSearcher restatement — checker for the host batched searcher.
 * One query at a time, the reference's exact control flow (B/graph/GraphSearcher.java):
 *   search :222-243, internalSearch :263-282, initializeInternal :334-353, stopSearch :355-369,
 *   searchOneLayer :406-457 (threshold = 0 => NoOpTracker), setEntryPointsFromPreviousLayer :324-331,
 *   searchLayer0 :459-469, addTopCandidate :515-530, reranking :471-507 + NodeQueue.rerank :160-230.
 * Score functions: PQDecoder / FusedPQDecoder arithmetic (jvo_adc_score over the code of the node);
 * in fused mode layer-0 neighbour scores come from the origin's packed block
 * (FusedPQDecoder.similarityToNeighbor :104-111) — identical values by construction.
 * The rerank walks the result heap in array order (NodeQueue.java:197-214), so membership for exact-score ties at the K-th
 * place is the reference's too.  jvo_searcher_* below restates the remaining options: threshold > 0 (ScoreTracker.java:80-140),
 * rerankFloor, resume() and the rerankedCount / worstApproximateInTopK outputs.
 * ---------------------------------------------------------------------------------------- */
/* NodeQueue over a BoundedLongHeap / GrowableLongHeap (B/graph/NodeQueue.java:36-58,83-85,125-137; B/util/
 * BoundedLongHeap.java:58-69): exported so that the reference's TestNodeQueue literals can pin it.
 * order: 0 = MIN_HEAP (apply(v) = v), 1 = MAX_HEAP (apply(v) = -1 - v).  cap <= 0 = growable (caller provides room).
 * heap[] is a binary min-heap of the order-applied keys.  Returns 1 if the value was added (BoundedLongHeap.push). */
int jvo_nodequeue_push(int64_t *heap, int *size, int cap, int order, int32_t node, float score)
{
    int64_t v = jvo_nodequeue_encode(node, score);
    if (order) v = -1 - v;
    if (cap > 0 && *size >= cap) {
        if (v < heap[0]) return 0;
        heap[0] = v;                       /* updateTop */
        heap_sift_down(heap, *size, 0);
        return 1;
    }
    int i = (*size)++;
    heap[i] = v;
    while (i > 0 && heap[(i - 1) / 2] > heap[i]) {
        int p = (i - 1) / 2;
        int64_t t = heap[i]; heap[i] = heap[p]; heap[p] = t;
        i = p;
    }
    return 1;
}
/* topNode / topScore / pop */
void jvo_nodequeue_top(const int64_t *heap, int order, int32_t *node, float *score)
{
    int64_t v = order ? -1 - heap[0] : heap[0];
    *node = (int32_t)~(uint32_t)(v & 0xFFFFFFFFLL);
    *score = jvo_sortable_int_to_float((int32_t)(v >> 32));
}
void jvo_nodequeue_pop(int64_t *heap, int *size)
{
    heap[0] = heap[--(*size)];
    heap_sift_down(heap, *size, 0);
}

typedef struct { int64_t *a; int n, cap; } lheap;  /* min-heap of keys */
static void lh_push(lheap *h, int64_t v)
{
    if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 64; h->a = (int64_t *)realloc(h->a, sizeof(int64_t) * (size_t)h->cap); }
    int i = h->n++;
    h->a[i] = v;
    while (i > 0 && h->a[(i - 1) / 2] > h->a[i]) { int p = (i - 1) / 2; int64_t t = h->a[i]; h->a[i] = h->a[p]; h->a[p] = t; i = p; }
}
static int64_t lh_pop(lheap *h)
{
    int64_t top = h->a[0];
    h->a[0] = h->a[--h->n];
    heap_sift_down(h->a, h->n, 0);
    return top;
}
static void jvo_nodequeue_push_lh(lheap *h, int order, int32_t node, float score)
{
    int64_t v = jvo_nodequeue_encode(node, score);
    lh_push(h, order ? -1 - v : v);
}
static inline int32_t key_node(int64_t k) { return (int32_t)~(uint32_t)(k & 0xFFFFFFFFLL); }
static inline float key_score(int64_t k) { return jvo_sortable_int_to_float((int32_t)(k >> 32)); }

/* level l: nodes[l] (sorted ascending, NULL for level 0 = every node), neighbors[l] rows of degree[l] ids (-1 pad) */
static const int32_t *level_row(const jvo_graph *g, int level, int32_t node)
{
    if (level == 0 || g->level_nodes[level] == NULL) return g->level_neighbors[level] + (size_t)node * g->level_degree[level];
    int lo = 0, hi = g->level_count[level] - 1;
    while (lo <= hi) {
        int mid = (lo + hi) / 2;
        int32_t v = g->level_nodes[level][mid];
        if (v == node) return g->level_neighbors[level] + (size_t)mid * g->level_degree[level];
        if (v < node) lo = mid + 1; else hi = mid - 1;
    }
    return NULL;
}

/* analysis aid (scripts/lut_cache_study.py): the calling thread's next jvo_graph_search* records the scored node ids in order */
static __thread int32_t *tl_visit_log = NULL;
static __thread int64_t tl_visit_cap = 0, tl_visit_n = 0;
void jvo_set_visit_log(int32_t *buf, int64_t cap) { tl_visit_log = buf; tl_visit_cap = cap; tl_visit_n = 0; }
int64_t jvo_visit_log_count(void) { return tl_visit_n; }

void jvo_graph_search(const jvo_graph *g, const jvo_pq *pq, const uint8_t *codes, const float *vecs,
                      const float *query, int vsf, int fused, int topK, int rerankK,
                      int32_t *out_ids, float *out_scores, int64_t *stats /* visited, expanded */)
{
    jvo_graph_search_filtered(g, pq, codes, vecs, query, vsf, fused, topK, rerankK, NULL, out_ids, out_scores, stats);
}

/* search(scoreProvider, topK, threshold = 0, acceptOrds): `accept` is the Bits filter as a little-endian bit array over node
 * ids (bit n of word n / 64), NULL = Bits.ALL.  Only layer 0 consults it (upper layers run with Bits.ALL, :276), and only to
 * decide whether the popped candidate becomes a RESULT (:437); traversal is unaffected. */
void jvo_graph_search_filtered(const jvo_graph *g, const jvo_pq *pq, const uint8_t *codes, const float *vecs,
                               const float *query, int vsf, int fused, int topK, int rerankK, const uint64_t *accept,
                               int32_t *out_ids, float *out_scores, int64_t *stats /* visited, expanded */)
{
    const int M = pq->M, k = pq->k;
    float *lut = (float *)malloc(sizeof(float) * (size_t)M * k);
    float *amag = (float *)malloc(sizeof(float) * (size_t)M * k);
    float bmag = 0.0f;
    /* partialSquaredMagnitudes is query independent and cached per ProductQuantization (ProductQuantization.java:75,238):
     * take the caller's copy when there is one instead of rebuilding it for every query */
    const float *am = (vsf == JVO_COSINE && pq->self_magnitudes) ? pq->self_magnitudes : amag;
    if (fused) jvo_fuseddecoder_init(pq, query, vsf, lut, am == amag ? amag : NULL, &bmag);
    else jvo_pqdecoder_init(pq, query, vsf, lut, am == amag ? amag : NULL, &bmag);
#define SCORE(node) adc_score_x(vsf, M, k, lut, am, bmag, codes + (size_t)(node) * M)
    /* visited set: a per-thread array of epoch stamps reused from query to query (the reference clears a growable bit
     * set per search; a fresh calloc of n_nodes bytes per query would make this baseline pay ~visited page faults each
     * time, which the reference does not) */
    static __thread uint32_t *tl_stamp = NULL;
    static __thread int64_t tl_cap = 0;
    static __thread uint32_t tl_epoch = 0;
    if (tl_cap < g->n_nodes) {
        free(tl_stamp);
        tl_stamp = (uint32_t *)calloc((size_t)g->n_nodes, sizeof(uint32_t));
        tl_cap = g->n_nodes;
        tl_epoch = 0;
    }
    if (++tl_epoch == 0) {  /* wrapped: start over */
        memset(tl_stamp, 0, sizeof(uint32_t) * (size_t)tl_cap);
        tl_epoch = 1;
    }
    uint32_t *const stamp = tl_stamp;
    const uint32_t epoch = tl_epoch;
    lheap cand = {0}, res = {0}, evicted = {0};
    int64_t n_visited = 0, n_expanded = 0;

    /* initializeInternal */
    stamp[g->entry_node] = epoch;
    lh_push(&cand, -1 - jvo_nodequeue_encode(g->entry_node, SCORE(g->entry_node)));  /* MAX_HEAP: -1 - v */

    for (int lvl = g->entry_level; lvl >= 0; lvl--) {
        const int rk = lvl > 0 ? 1 : rerankK;
        /* searchOneLayer */
        while (cand.n > 0) {
            int64_t topKey = -1 - cand.a[0];
            float topScore = key_score(topKey);
            if (res.n >= rk && topScore < key_score(res.a[0])) break;  /* stopSearch */
            lh_pop(&cand);
            int32_t node = key_node(topKey);
            /* acceptOrds = ALL, threshold = 0.0f: `topCandidateScore >= threshold` (:437) still keeps negative and NaN
             * scores out of the results (they are expanded all the same); then addTopCandidate :515-530 */
            if (!(topScore >= 0.0f) || (lvl == 0 && accept && !((accept[node >> 6] >> (node & 63)) & 1))) { /* not a result */ }
            else if (res.n < rk) lh_push(&res, topKey);
            else if (topScore > key_score(res.a[0])) {
                lh_push(&evicted, res.a[0]);
                res.a[0] = topKey;              /* BoundedLongHeap.updateTop */
                heap_sift_down(res.a, res.n, 0);
            }
            n_expanded++;
            const int32_t *row = level_row(g, lvl, node);
            if (!row) continue;
            for (int i = 0; i < g->level_degree[lvl]; i++) {
                int32_t nb = row[i];
                if (nb < 0) break;  /* neighbour lists are packed: first -1 ends the row */
                if (stamp[nb] == epoch) continue;
                stamp[nb] = epoch;
                lh_push(&cand, -1 - jvo_nodequeue_encode(nb, SCORE(nb)));
                if (tl_visit_log && tl_visit_n < tl_visit_cap) tl_visit_log[tl_visit_n] = nb;
                tl_visit_n++;
                n_visited++;
            }
        }
        if (lvl > 0) {  /* setEntryPointsFromPreviousLayer */
            for (int i = 0; i < res.n; i++) lh_push(&cand, -1 - res.a[i]);
            for (int i = 0; i < evicted.n; i++) lh_push(&cand, -1 - evicted.a[i]);
            res.n = 0;
            evicted.n = 0;
        }
    }
    /* reranking :471-507.  With a reranker: NodeQueue.rerank :160-230 walks approximateResults in HEAP ARRAY order (res.a is
     * that array: the same push / updateTop sequence on the same binary heap, AbstractLongHeap.java:77-85,158-187) and keeps
     * an entry only while the bounded queue has room or its exact score is STRICTLY better than the worst kept (:204-211), so
     * which of several candidates tied on the exact score at the K-th place survives follows that order.  Without one
     * (:478-487): the worst approximate results are popped until topK remain. */
    int64_t *fin = (int64_t *)malloc(sizeof(int64_t) * (size_t)(res.n > 0 ? res.n : 1));
    int nf = 0;
    if (vecs) {
        for (int i = 0; i < res.n; i++) {
            int32_t id = key_node(res.a[i]);
            float ex = rerank_x(vsf, query, vecs, id, pq->D);
            if (nf < topK) jvo_nodequeue_push(fin, &nf, 0, 0, id, ex);
            else if (ex > key_score(fin[0])) jvo_nodequeue_push(fin, &nf, topK, 0, id, ex);
        }
    } else {
        while (res.n > topK) lh_pop(&res);
        for (int i = 0; i < res.n; i++) fin[nf++] = res.a[i];
    }
    qsort(fin, (size_t)nf, sizeof(int64_t), cmp_desc_i64);
    for (int i = 0; i < topK; i++) {
        out_ids[i] = i < nf ? key_node(fin[i]) : -1;
        out_scores[i] = i < nf ? key_score(fin[i]) : -INFINITY;
    }
    if (stats) { stats[0] = n_visited; stats[1] = n_expanded; }
#undef SCORE
    free(fin); free(cand.a); free(res.a); free(evicted.a); free(lut); free(amag);
}

/* ------------------------------------------------------------------------------------------
 * GraphSearcher as an OBJECT — the options the one-shot function above leaves out:
 *   threshold > 0 : ScoreTrackerFactory.getScoreTracker (B/graph/ScoreTracker.java:38-58) hands layer 0 a TwoPhaseTracker
 *                   (:80-140); stopSearch (GraphSearcher.java:355-369) and the edge-loading skip (:441-444) consult it, and
 *                   `topCandidateScore >= threshold` (:437) keeps weaker nodes out of the results.
 *   rerankFloor   : NodeQueue.rerank (B/graph/NodeQueue.java:160-230) scores exactly only the entries whose approximate
 *                   score reaches the floor (or the best one when none does); the rest go to evictedResults.
 *   resume()      : GraphSearcher.java:459-469,509-513,538-547 — candidates, visited and evictedResults survive a search;
 *                   resume pushes the evicted nodes back and continues layer 0.  CachingReranker (:554-581) remembers exact
 *                   scores, rerankedCount counts the new ones.
 * TwoPhaseTracker.shouldStop calls org.apache.commons.math3.stat.StatUtils.percentile(recentScores, 99) — commons-math3
 * 3.6.1 (pom.xml:189-190), a dependency that is not in /root/reference.  Restated from its published algorithm (class
 * Percentile, EstimationType.LEGACY, the default): pos = (p / 100) * (n + 1); pos < 1 -> min; pos >= n -> max; else with the
 * sorted values, lower = x[floor(pos) - 1], upper = x[floor(pos)], result = lower + (pos - floor(pos)) * (upper - lower),
 * all in double.  jvo_percentile_legacy exports it so the class' documented examples can pin it.
 * ---------------------------------------------------------------------------------------- */
static int cmp_asc_f64(const void *x, const void *y)
{
    double a = *(const double *)x, b = *(const double *)y;
    return a < b ? -1 : (a > b ? 1 : 0);
}
double jvo_percentile_legacy(const double *values, int n, double p)
{
    if (n <= 0) return NAN;
    if (n == 1) return values[0];
    double *w = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(w, values, sizeof(double) * (size_t)n);
    qsort(w, (size_t)n, sizeof(double), cmp_asc_f64);
    const double q = p / 100.0;
    const double pos = q == 1.0 ? (double)n : q * (double)(n + 1);
    const double fpos = floor(pos);
    const int ipos = (int)fpos;
    const double dif = pos - fpos;
    double r;
    if (pos < 1.0) r = w[0];
    else if (pos >= (double)n) r = w[n - 1];
    else { const double lower = w[ipos - 1], upper = w[ipos]; r = lower + dif * (upper - lower); }
    free(w);
    return r;
}

#define JVO_RECENT_SCORES_TRACKED 500  /* ScoreTracker.java:81 */
#define JVO_BEST_SCORES_TRACKED 100    /* :82 */
struct jvo_searcher {
    const jvo_graph *g; const jvo_pq *pq; const uint8_t *codes; const float *vecs; int vsf, fused;
    float *lut, *amag, *query; const float *am; float bmag;
    uint8_t *visited;                 /* IntHashSet visited (:64) */
    lheap cand, res, evicted, rer;    /* candidates (keys stored as -1 - v: MAX_HEAP), approximateResults, evictedResults
                                         (NodesUnsorted: plain list, `a` used as an array), rerankedResults */
    uint64_t *accept; int has_accept; /* this.acceptOrds (:338), kept for resume */
    uint8_t *cached; float *cache_val; int64_t rerank_calls; /* CachingReranker :554-581 */
    int64_t visitedCount, expandedCount, expandedBase;
    /* TwoPhaseTracker */
    double recent[JVO_RECENT_SCORES_TRACKED]; int recent_idx; int obs; double thr; int tracking;
    int32_t best[JVO_BEST_SCORES_TRACKED]; int n_best;
    int searched;
};

jvo_searcher *jvo_searcher_new(const jvo_graph *g, const jvo_pq *pq, const uint8_t *codes, const float *vecs, int vsf,
                               int fused)
{
    jvo_searcher *s = (jvo_searcher *)calloc(1, sizeof(jvo_searcher));
    s->g = g; s->pq = pq; s->codes = codes; s->vecs = vecs; s->vsf = vsf; s->fused = fused;
    s->lut = (float *)malloc(sizeof(float) * (size_t)pq->M * pq->k);
    s->amag = (float *)malloc(sizeof(float) * (size_t)pq->M * pq->k);
    s->query = (float *)malloc(sizeof(float) * (size_t)pq->D);
    s->visited = (uint8_t *)calloc((size_t)g->n_nodes, 1);
    s->accept = (uint64_t *)calloc((size_t)(g->n_nodes + 63) / 64, sizeof(uint64_t));
    s->cached = (uint8_t *)calloc((size_t)g->n_nodes, 1);
    s->cache_val = (float *)malloc(sizeof(float) * (size_t)g->n_nodes);
    return s;
}
void jvo_searcher_free(jvo_searcher *s)
{
    if (!s) return;
    free(s->lut); free(s->amag); free(s->query); free(s->visited); free(s->accept); free(s->cached); free(s->cache_val);
    free(s->cand.a); free(s->res.a); free(s->evicted.a); free(s->rer.a);
    free(s);
}

static void ev_add(lheap *h, int64_t key)  /* NodesUnsorted.add */
{
    if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 64; h->a = (int64_t *)realloc(h->a, sizeof(int64_t) * (size_t)h->cap); }
    h->a[h->n++] = key;
}
static void tr_reset(jvo_searcher *s, float threshold)  /* getScoreTracker :38-58, TwoPhaseTracker.reset :104-108 */
{
    s->tracking = threshold > 0;
    s->n_best = 0; s->obs = 0; s->thr = (double)threshold;
}
static void tr_track(jvo_searcher *s, float score)  /* :110-116 */
{
    if (!s->tracking) return;
    int32_t v = jvo_float_to_sortable_int(score);
    if (s->n_best < JVO_BEST_SCORES_TRACKED) {  /* BoundedLongHeap.push: min-heap of the best scores */
        int i = s->n_best++;
        s->best[i] = v;
        while (i > 0 && s->best[(i - 1) / 2] > s->best[i]) { int p = (i - 1) / 2; int32_t t = s->best[i]; s->best[i] = s->best[p]; s->best[p] = t; i = p; }
    } else if (!(v < s->best[0])) {
        s->best[0] = v;
        for (int i = 0;;) {
            int l = 2 * i + 1, r = l + 1, m = i;
            if (l < s->n_best && s->best[l] < s->best[m]) m = l;
            if (r < s->n_best && s->best[r] < s->best[m]) m = r;
            if (m == i) break;
            int32_t t = s->best[i]; s->best[i] = s->best[m]; s->best[m] = t; i = m;
        }
    }
    s->recent[s->recent_idx] = (double)score;
    s->recent_idx = (s->recent_idx + 1) % JVO_RECENT_SCORES_TRACKED;
    s->obs++;
}
static int tr_should_stop(const jvo_searcher *s)  /* :118-137 */
{
    if (!s->tracking) return 0;
    if (s->obs < JVO_RECENT_SCORES_TRACKED) return 0;
    if (s->obs % 100 != 0) return 0;
    double windowMedian = jvo_percentile_legacy(s->recent, JVO_RECENT_SCORES_TRACKED, 99);
    double worstBestScore = (double)jvo_sortable_int_to_float(s->best[0]);
    return windowMedian < worstBestScore && windowMedian < s->thr;
}

static float searcher_score(const jvo_searcher *s, int32_t node)
{
    return adc_score_x(s->vsf, s->pq->M, s->pq->k, s->lut, s->am, s->bmag, s->codes + (size_t)node * s->pq->M);
}

/* searchOneLayer :406-457 */
static void searcher_one_layer(jvo_searcher *s, int rk, float threshold, int level, int use_accept)
{
    const jvo_graph *g = s->g;
    tr_reset(s, threshold);
    while (s->cand.n > 0) {
        int64_t topKey = -1 - s->cand.a[0];
        float topScore = key_score(topKey);
        if (s->res.n >= rk && topScore < key_score(s->res.a[0])) break;  /* stopSearch :358-361 */
        if (threshold > 0 && tr_should_stop(s)) break;                     /* :364-366 */
        lh_pop(&s->cand);
        int32_t node = key_node(topKey);
        int ok = !use_accept || !s->has_accept || ((s->accept[node >> 6] >> (node & 63)) & 1);
        if (ok && topScore >= threshold) {  /* :437; addTopCandidate :515-530 */
            if (s->res.n < rk) lh_push(&s->res, topKey);
            else if (topScore > key_score(s->res.a[0])) {
                ev_add(&s->evicted, s->res.a[0]);
                s->res.a[0] = topKey;
                heap_sift_down(s->res.a, s->res.n, 0);
            }
        }
        if (tr_should_stop(s) && s->cand.n >= rk - s->res.n) continue;  /* :441-444 */
        if (level == 0) s->expandedBase++;
        s->expandedCount++;
        const int32_t *row = level_row(g, level, node);
        if (!row) continue;
        for (int i = 0; i < g->level_degree[level]; i++) {
            int32_t nb = row[i];
            if (nb < 0) break;
            if (s->visited[nb]) continue;
            s->visited[nb] = 1;
            float sc = searcher_score(s, nb);
            tr_track(s, sc);
            lh_push(&s->cand, -1 - jvo_nodequeue_encode(nb, sc));
            s->visitedCount++;
        }
    }
}

static void searcher_layer0(jvo_searcher *s, int topK, int rerankK, float threshold)  /* searchLayer0 :459-469 */
{
    s->rer.n = 0;
    for (int i = 0; i < s->evicted.n; i++) lh_push(&s->cand, -1 - s->evicted.a[i]);
    s->evicted.n = 0;
    searcher_one_layer(s, rerankK, threshold, 0, 1);
    (void)topK;
}

static float searcher_exact(jvo_searcher *s, int32_t node)  /* CachingReranker.similarityTo :568-576 */
{
    if (s->cached[node]) return s->cache_val[node];
    s->rerank_calls++;
    float ex = rerank_x(s->vsf, s->query, s->vecs, node, s->pq->D);
    s->cached[node] = 1;
    s->cache_val[node] = ex;
    return ex;
}

/* reranking :471-507; returns the number of results */
static int searcher_reranking(jvo_searcher *s, int topK, float rerankFloor, int32_t *out_ids, float *out_scores,
                              int64_t *stats, float *worst_out)
{
    int64_t reranked = 0;
    float worst = INFINITY;
    lheap *from;
    if (!s->vecs) {  /* cachingReranker == null :478-487 */
        while (s->res.n > topK) ev_add(&s->evicted, lh_pop(&s->res));
        from = &s->res;
    } else {         /* NodeQueue.rerank :160-230 */
        const int64_t before = s->rerank_calls;
        const int n = s->res.n;
        int32_t *ids = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
        float *ex = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
        float *approxById = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));  /* approximateScoresById, by position */
        float bestScore = -INFINITY; int bestIndex = -1, above = 0;
        for (int i = 0; i < n; i++) {
            float sc = key_score(s->res.a[i]);
            int32_t id = key_node(s->res.a[i]);
            approxById[i] = sc;
            if (sc > bestScore) { bestScore = sc; bestIndex = i; }
            if (sc >= rerankFloor) { ids[i] = id; ex[i] = searcher_exact(s, id); above++; }
            else ids[i] = -1;
        }
        if (above == 0 && bestIndex >= 0) { ids[bestIndex] = key_node(s->res.a[bestIndex]); ex[bestIndex] = searcher_exact(s, ids[bestIndex]); }
        /* positions of the kept entries, parallel to s->rer (to find approximate scores again) */
        for (int i = 0; i < n; i++) {
            if (ids[i] == -1) { ev_add(&s->evicted, s->res.a[i]); continue; }
            if (s->rer.n < topK) jvo_nodequeue_push_lh(&s->rer, 0, ids[i], ex[i]);
            else if (ex[i] > key_score(s->rer.a[0])) {
                int32_t evNode = key_node(s->rer.a[0]);
                for (int j = 0; j < n; j++) if (ids[j] == evNode) { ev_add(&s->evicted, jvo_nodequeue_encode(evNode, approxById[j])); break; }
                s->rer.a[0] = jvo_nodequeue_encode(ids[i], ex[i]);
                heap_sift_down(s->rer.a, s->rer.n, 0);
            } else ev_add(&s->evicted, s->res.a[i]);
        }
        if (s->rer.n >= topK) {
            for (int i = 0; i < s->rer.n; i++) {
                int32_t node = key_node(s->rer.a[i]);
                for (int j = 0; j < n; j++) if (ids[j] == node) { if (approxById[j] < worst) worst = approxById[j]; break; }
            }
        }
        reranked = s->rerank_calls - before;
        s->res.n = 0;
        from = &s->rer;
        free(ids); free(ex); free(approxById);
    }
    const int nres = from->n;
    for (int i = nres - 1; i >= 0; i--) {  /* :497-502: pop worst first */
        int64_t k = lh_pop(from);
        out_ids[i] = key_node(k);
        out_scores[i] = key_score(k);
    }
    for (int i = nres; i < topK; i++) { out_ids[i] = -1; out_scores[i] = -INFINITY; }
    if (stats) { stats[0] = s->visitedCount; stats[1] = s->expandedCount; stats[2] = s->expandedBase; stats[3] = reranked; }
    if (worst_out) *worst_out = worst;
    return nres;
}

/* search(scoreProvider, topK, rerankK, threshold, rerankFloor, acceptOrds) :222-243.  accept: bit array over node ids or NULL =
 * Bits.ALL (copied).  out_*: topK entries, best first, (-1, -inf) padded.  stats (nullable): {visitedCount, expandedCount,
 * expandedCountBaseLayer, rerankedCount}.  Returns the number of results, -1 on rerankK < topK (:233-235). */
int jvo_searcher_search(jvo_searcher *s, const float *query, int topK, int rerankK, float threshold, float rerankFloor,
                        const uint64_t *accept, int32_t *out_ids, float *out_scores, int64_t *stats, float *worst_out)
{
    const jvo_graph *g = s->g;
    if (rerankK < topK) return -1;
    /* initializeInternal :334-353 */
    memcpy(s->query, query, sizeof(float) * (size_t)s->pq->D);
    s->am = (s->vsf == JVO_COSINE && s->pq->self_magnitudes) ? s->pq->self_magnitudes : s->amag;
    if (s->fused) jvo_fuseddecoder_init(s->pq, query, s->vsf, s->lut, s->am == s->amag ? s->amag : NULL, &s->bmag);
    else jvo_pqdecoder_init(s->pq, query, s->vsf, s->lut, s->am == s->amag ? s->amag : NULL, &s->bmag);
    s->has_accept = accept != NULL;
    if (accept) memcpy(s->accept, accept, sizeof(uint64_t) * (size_t)((g->n_nodes + 63) / 64));
    memset(s->cached, 0, (size_t)g->n_nodes);  /* a new CachingReranker per search :104-112 */
    s->rerank_calls = 0;
    s->res.n = s->evicted.n = s->cand.n = s->rer.n = 0;
    memset(s->visited, 0, (size_t)g->n_nodes);
    s->visited[g->entry_node] = 1;
    lh_push(&s->cand, -1 - jvo_nodequeue_encode(g->entry_node, searcher_score(s, g->entry_node)));
    s->visitedCount = s->expandedCount = s->expandedBase = 0;
    s->searched = 1;
    /* internalSearch :263-282 */
    for (int lvl = g->entry_level; lvl > 0; lvl--) {
        searcher_one_layer(s, 1, 0.0f, lvl, 0);
        for (int i = 0; i < s->res.n; i++) lh_push(&s->cand, -1 - s->res.a[i]);  /* setEntryPointsFromPreviousLayer */
        for (int i = 0; i < s->evicted.n; i++) lh_push(&s->cand, -1 - s->evicted.a[i]);
        s->res.n = s->evicted.n = 0;
    }
    searcher_layer0(s, topK, rerankK, threshold);
    return searcher_reranking(s, topK, rerankFloor, out_ids, out_scores, stats, worst_out);
}

/* resume(additionalK, rerankK) :538-547 (threshold = rerankFloor = 0).  -1 when no search came first. */
int jvo_searcher_resume(jvo_searcher *s, int additionalK, int rerankK, int32_t *out_ids, float *out_scores, int64_t *stats,
                        float *worst_out)
{
    if (!s->searched || rerankK < additionalK) return -1;
    s->visitedCount = s->expandedCount = s->expandedBase = 0;
    searcher_layer0(s, additionalK, rerankK, 0.0f);
    return searcher_reranking(s, additionalK, 0.0f, out_ids, out_scores, stats, worst_out);
}

/* ------------------------------------------------------------------------------------------
 * GraphIndexBuilder, one thread (checker for jv_hip_builder_* / jv_hip_build_layered in reference order).
 *   java.util.Random(0) + getRandomGraphLevel                     B/graph/GraphIndexBuilder.java:337,562-575
 *   addGraphNode / updateNeighborsOneLayer / updateNeighbors       :605-659, :800-825 (no concurrent candidates with one thread)
 *   cleanup / improveConnections                                   :472-545
 *   OnHeapGraphIndex.addNode / markComplete / addEdges             B/graph/OnHeapGraphIndex.java:161-168,214-225,279-282
 *   ConcurrentNeighborMap.Neighbors.insertDiverse / insert / enforceDegree / retainDiverseInternal / backlink
 *                                                                 B/graph/ConcurrentNeighborMap.java:139-146,190-200,222-243,262-296
 *   NodeArray.merge / insertionPoint / duplicateExistsNear / insertSorted / retain
 *                                                                 B/graph/NodeArray.java:63-143,166-238,240-256,308-318
 *   scores: BuildScoreProvider.pqBuildScoreProvider (:171-205) — searches with the precomputed ADC tables of the node's
 *   full-resolution vector, no rerank; diversity with the PQ pair function (jvo_retain_diverse above).
 * One degree for every level (the List.of(M) the reference's callers pass).  Rows live in dense per-level arrays indexed by node id
 * (the reference's SparseIntMap for the upper levels holds the same rows), packed and -1 padded so that the searcher above reads
 * them in place — in NodeArray order, which is the order the reference's neighbour iterator walks.
 * ---------------------------------------------------------------------------------------- */
#define JVO_BUILD_MAX_LEVELS 32

void jvo_java_random_seed(int64_t *state, int64_t seed) { *state = (seed ^ 0x5DEECE66DLL) & ((1LL << 48) - 1); }
static int32_t java_random_next(int64_t *state, int bits)
{
    *state = (int64_t)(((uint64_t)*state * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1));
    return (int32_t)((uint64_t)*state >> (48 - bits));
}
double jvo_java_random_next_double(int64_t *state)
{
    const int64_t hi = (int64_t)java_random_next(state, 26), lo = (int64_t)java_random_next(state, 27);
    return (double)((hi << 27) + lo) * 0x1.0p-53;
}
/* getRandomGraphLevel :562-575 */
int jvo_random_graph_level(int64_t *state, int degree0, int addHierarchy)
{
    double ml, r;
    if (addHierarchy) {
        ml = degree0 == 1 ? 1.0 : 1.0 / log(1.0 * degree0);
        do { r = jvo_java_random_next_double(state); } while (r == 0.0);
    } else { ml = 0; r = 0; }
    if (!addHierarchy) return 0;   /* (int) (-log(0) * 0) = (int) NaN = 0 in Java */
    return (int)(-log(r) * ml);
}

typedef struct { int32_t *node; float *score; int size; } nodearr;   /* views into caller-sized storage */

/* NodeArray.merge :63-143 — out must hold a.size + b.size entries */
static void na_merge(const nodearr *a1, const nodearr *a2, nodearr *m)
{
    int i = 0, j = 0, nl = 0;
    int32_t *last = (int32_t *)malloc(sizeof(int32_t) * (size_t)(a1->size + a2->size + 1));   /* nodesWithLastScore */
    float lastScore = NAN;
    m->size = 0;
#define NA_LAST_ADD(x, ok) do { ok = 1; for (int t_ = 0; t_ < nl; t_++) if (last[t_] == (x)) { ok = 0; break; } if (ok) last[nl++] = (x); } while (0)
#define NA_TAKE(arr, idx) do { if ((arr)->score[idx] != lastScore) { nl = 0; lastScore = (arr)->score[idx]; } int ok_; NA_LAST_ADD((arr)->node[idx], ok_); \
        if (ok_) { m->node[m->size] = (arr)->node[idx]; m->score[m->size] = (arr)->score[idx]; m->size++; } } while (0)
    while (i < a1->size && j < a2->size) {
        if (a1->score[i] < a2->score[j]) { NA_TAKE(a2, j); j++; }
        else if (a1->score[i] > a2->score[j]) { NA_TAKE(a1, i); i++; }
        else { NA_TAKE(a1, i); { int ok_; NA_LAST_ADD(a2->node[j], ok_); if (ok_) { m->node[m->size] = a2->node[j]; m->score[m->size] = a2->score[j]; m->size++; } } i++; j++; }
    }
    const nodearr *rest = i < a1->size ? a1 : a2;
    int r = i < a1->size ? i : j;
    if (r < rest->size) {
        while (r < rest->size && rest->score[r] == lastScore) {
            int seen = 0;
            for (int t = 0; t < nl; t++) if (last[t] == rest->node[r]) { seen = 1; break; }
            if (!seen) { m->node[m->size] = rest->node[r]; m->score[m->size] = rest->score[r]; m->size++; }   /* (contains(), not add(): :123,137) */
            r++;
        }
        for (; r < rest->size; r++) { m->node[m->size] = rest->node[r]; m->score[m->size] = rest->score[r]; m->size++; }
    }
#undef NA_TAKE
#undef NA_LAST_ADD
    free(last);
}
/* descSortFindRightMostInsertionPoint :308-318 + duplicateExistsNear :212-228: -1 = (node, score) already there */
static int na_insertion_point(const nodearr *a, int32_t node, float score)
{
    int start = 0, end = a->size - 1;
    while (start <= end) {
        int mid = (start + end) / 2;
        if (a->score[mid] < score) end = mid - 1; else start = mid + 1;
    }
    for (int i = start - 1; i >= 0 && a->score[i] == score; i--) if (a->node[i] == node) return -1;
    for (int i = start; i < a->size && a->score[i] == score; i++) if (a->node[i] == node) return -1;
    return start;
}
static void na_insert_at(nodearr *a, int at, int32_t node, float score)
{
    memmove(a->node + at + 1, a->node + at, sizeof(int32_t) * (size_t)(a->size - at));
    memmove(a->score + at + 1, a->score + at, sizeof(float) * (size_t)(a->size - at));
    a->node[at] = node; a->score[at] = score; a->size++;
}

struct jvo_builder {
    const jvo_pq *pq; const uint8_t *codes; const float *vecs; int64_t n; int vsf;
    int maxDegree, beam, W, hardMaxDegree, addHierarchy, refine, dedupe_ids, improve_full_vectors, improve_sorted_candidates;
    float alpha, overflow;
    float *tri;
    int n_levels;                                   /* layers.size() */
    int32_t *ids[JVO_BUILD_MAX_LEVELS]; float *sc[JVO_BUILD_MAX_LEVELS];   /* [n][W] */
    int32_t *size[JVO_BUILD_MAX_LEVELS], *db[JVO_BUILD_MAX_LEVELS];        /* [n]; size -1 = the node is not on the level */
    int32_t entry_node; int entry_level;
    int64_t rng;
    const int8_t *forced_levels;
    jvo_graph g; int lcount[JVO_BUILD_MAX_LEVELS], ldeg[JVO_BUILD_MAX_LEVELS];
    const int32_t *lnodes[JVO_BUILD_MAX_LEVELS]; const int32_t *lnbrs[JVO_BUILD_MAX_LEVELS];
    jvo_searcher *s;
    int64_t reprunes;
};

static void bld_ensure_level(jvo_builder *b, int level)   /* ensureLayersExist :180-193 */
{
    for (int l = b->n_levels; l <= level; l++) {
        b->ids[l] = (int32_t *)malloc(sizeof(int32_t) * (size_t)b->n * b->W);
        memset(b->ids[l], 0xFF, sizeof(int32_t) * (size_t)b->n * b->W);
        b->sc[l] = (float *)calloc((size_t)b->n * b->W, sizeof(float));
        b->size[l] = (int32_t *)malloc(sizeof(int32_t) * (size_t)b->n);
        b->db[l] = (int32_t *)calloc((size_t)b->n, sizeof(int32_t));
        for (int64_t i = 0; i < b->n; i++) b->size[l][i] = -1;
        b->lcount[l] = (int)b->n; b->ldeg[l] = b->W; b->lnodes[l] = NULL; b->lnbrs[l] = b->ids[l];
        b->n_levels = l + 1;
    }
    b->g.n_levels = b->n_levels;
}

jvo_builder *jvo_builder_new(const jvo_pq *pq, const uint8_t *codes, const float *vecs, int64_t n, int vsf, int maxDegree, int beamWidth,
                             float alpha, float neighborOverflow, int addHierarchy, int refineFinalGraph)
{
    jvo_builder *b = (jvo_builder *)calloc(1, sizeof(jvo_builder));
    b->pq = pq; b->codes = codes; b->vecs = vecs; b->n = n; b->vsf = vsf;
    b->maxDegree = maxDegree; b->beam = beamWidth; b->alpha = alpha; b->overflow = neighborOverflow;
    b->hardMaxDegree = (int)(neighborOverflow * maxDegree);       /* Neighbors.insert :270 ((int) (overflow * map.maxDegree)) */
    b->W = (int)(maxDegree * neighborOverflow) + 1;               /* nodeArrayLength :128-131 over maxOverflowDegree (OnHeapGraphIndex :187) */
    if (b->W < b->hardMaxDegree + 1) b->W = b->hardMaxDegree + 1;
    b->addHierarchy = addHierarchy; b->refine = refineFinalGraph;
    b->tri = (float *)malloc(sizeof(float) * (size_t)pq->M * ((size_t)pq->k * (pq->k + 1) / 2));
    jvo_pq_codebook_partial_sums(pq, vsf, b->tri);
    b->entry_node = -1; b->entry_level = -1;
    jvo_java_random_seed(&b->rng, 0);
    b->g.n_nodes = n; b->g.level_count = b->lcount; b->g.level_degree = b->ldeg; b->g.level_nodes = b->lnodes; b->g.level_neighbors = b->lnbrs;
    bld_ensure_level(b, 0);
    b->s = jvo_searcher_new(&b->g, pq, codes, NULL, vsf, 0);      /* "deliberately skips reranking" :197-200 */
    return b;
}
void jvo_builder_free(jvo_builder *b)
{
    if (!b) return;
    for (int l = 0; l < b->n_levels; l++) { free(b->ids[l]); free(b->sc[l]); free(b->size[l]); free(b->db[l]); }
    jvo_searcher_free(b->s);
    free(b->tri);
    free(b);
}
/* per-node levels instead of the Random(0) draws (NULL = draw); and the three places where the engine's improve pass leaves the
 * reference ON PURPOSE: dedupe_ids != 0 — an entry whose node a list already holds is dropped whatever its score, in
 * insertDiverse's merge and in the backlink's insert (the reference drops it only at an equal score, so a node can sit in a list twice:
 * an improve search scores a pair differently from the insert that linked it); full_vectors != 0 — the improve search is given the
 * node's full-resolution vector, not its decoded code (searchProviderFor(int) :189-194); sorted_candidates != 0 — the improve search's
 * results are taken in SearchResult order (score descending, EQUAL scores by ascending node id) instead of the result heap's array
 * order, which decides the order among equal scores only. */
void jvo_builder_set_levels(jvo_builder *b, const int8_t *levels) { b->forced_levels = levels; }
void jvo_builder_set_deviations(jvo_builder *b, int dedupe_ids, int full_vectors, int sorted_candidates)
{
    b->dedupe_ids = dedupe_ids; b->improve_full_vectors = full_vectors; b->improve_sorted_candidates = sorted_candidates;
}

static nodearr bld_row(jvo_builder *b, int level, int32_t node)
{
    nodearr a = { b->ids[level] + (size_t)node * b->W, b->sc[level] + (size_t)node * b->W, b->size[level][node] };
    return a;
}
static void bld_store(jvo_builder *b, int level, int32_t node, const nodearr *a, int diverseBefore)
{
    int32_t *ids = b->ids[level] + (size_t)node * b->W; float *sc = b->sc[level] + (size_t)node * b->W;
    if (a->node != ids) { memcpy(ids, a->node, sizeof(int32_t) * (size_t)a->size); memcpy(sc, a->score, sizeof(float) * (size_t)a->size); }
    for (int i = a->size; i < b->W; i++) { ids[i] = -1; sc[i] = 0.0f; }
    b->size[level][node] = a->size; b->db[level][node] = diverseBefore;
}
/* retainDiverseInternal :262-267: neighbors updated in place */
static void bld_retain(jvo_builder *b, nodearr *a, int diverseBefore)
{
    uint8_t *sel = (uint8_t *)malloc((size_t)a->size + 1);
    jvo_retain_diverse(b->tri, b->pq->M, b->pq->k, b->vsf, b->codes, a->node, a->score, a->size, b->maxDegree, diverseBefore, b->alpha, sel, NULL);
    int w = 0;
    for (int r = 0; r < a->size; r++) if (sel[r]) { a->node[w] = a->node[r]; a->score[w] = a->score[r]; w++; }
    a->size = w;
    free(sel);
    b->reprunes++;
}
/* addEdges :279-282 = insertDiverse(node, candidates) :222-243, then backlink(newNeighbors, node, overflow) :139-146 -> insert :262-296 */
static void bld_add_edges(jvo_builder *b, int level, int32_t node, const nodearr *cand)
{
    nodearr cur = bld_row(b, level, node);
    if (cand->size > 0) {
        nodearr merged; merged.node = (int32_t *)malloc(sizeof(int32_t) * (size_t)(cur.size + cand->size + 1));
        merged.score = (float *)malloc(sizeof(float) * (size_t)(cur.size + cand->size + 1));
        if (cur.size > 0) {
            na_merge(&cur, cand, &merged);
            if (b->dedupe_ids) {   /* (deviation, see jvo_builder_set_deviations: the first = better-scored entry of a node stays) */
                int w = 0;
                for (int r = 0; r < merged.size; r++) {
                    int dup = 0;
                    for (int t = 0; t < w; t++) if (merged.node[t] == merged.node[r]) { dup = 1; break; }
                    if (!dup) { merged.node[w] = merged.node[r]; merged.score[w] = merged.score[r]; w++; }
                }
                merged.size = w;
            }
        } else {
            memcpy(merged.node, cand->node, sizeof(int32_t) * (size_t)cand->size);
            memcpy(merged.score, cand->score, sizeof(float) * (size_t)cand->size);
            merged.size = cand->size;
        }
        bld_retain(b, &merged, 0);
        bld_store(b, level, node, &merged, merged.size);           /* new Neighbors(...): diverseBefore = size() :161-165 */
        free(merged.node); free(merged.score);
    }
    /* backlink over the (new or unchanged) neighbours — a private copy: a list never holds its own node, so the loop below cannot
     * change it, but the reference iterates the snapshot insertDiverse returned */
    cur = bld_row(b, level, node);
    int32_t *nn = (int32_t *)malloc(sizeof(int32_t) * (size_t)(cur.size + 1)); float *ns = (float *)malloc(sizeof(float) * (size_t)(cur.size + 1));
    memcpy(nn, cur.node, sizeof(int32_t) * (size_t)cur.size); memcpy(ns, cur.score, sizeof(float) * (size_t)cur.size);
    const int cnt = cur.size;
    for (int i = 0; i < cnt; i++) {
        const int32_t nbr = nn[i];
        nodearr t = bld_row(b, level, nbr);
        if (t.size < 0) continue;                                 /* (cannot happen: a result of the level's search is on the level) */
        const int at = na_insertion_point(&t, node, ns[i]);
        if (at == -1) continue;                                   /* "new" node already existed :275-278 */
        if (b->dedupe_ids) {                                      /* (deviation, see jvo_builder_set_deviations) */
            int have = 0;
            for (int r = 0; r < t.size; r++) if (t.node[r] == node) { have = 1; break; }
            if (have) continue;
        }
        na_insert_at(&t, at, node, ns[i]);
        int dbf = b->db[level][nbr] < at ? b->db[level][nbr] : at;   /* min(insertionPoint, diverseBefore) :285 */
        if (t.size > b->hardMaxDegree) { bld_retain(b, &t, dbf); dbf = t.size; }
        bld_store(b, level, nbr, &t, dbf);
    }
    free(nn); free(ns);
}

static void bld_search_init(jvo_builder *b, const float *query, int32_t exclude)   /* initializeInternal :334-353, acceptOrds = ExcludingBits(node) */
{
    jvo_searcher *s = b->s;
    memcpy(s->query, query, sizeof(float) * (size_t)s->pq->D);
    s->am = (s->vsf == JVO_COSINE && s->pq->self_magnitudes) ? s->pq->self_magnitudes : s->amag;
    jvo_pqdecoder_init(s->pq, query, s->vsf, s->lut, s->am == s->amag ? s->amag : NULL, &s->bmag);
    s->has_accept = 1;
    memset(s->accept, 0xFF, sizeof(uint64_t) * (size_t)((b->n + 63) / 64));
    s->accept[exclude >> 6] &= ~(1ULL << (exclude & 63));
    s->res.n = s->evicted.n = s->cand.n = s->rer.n = 0;
    memset(s->visited, 0, (size_t)b->n);
    s->visited[b->entry_node] = 1;
    lh_push(&s->cand, -1 - jvo_nodequeue_encode(b->entry_node, searcher_score(s, b->entry_node)));
    s->visitedCount = s->expandedCount = s->expandedBase = 0;
    s->searched = 1;
}
static void bld_entry_points_from_previous_layer(jvo_searcher *s)   /* :324-331 */
{
    for (int i = 0; i < s->res.n; i++) lh_push(&s->cand, -1 - s->res.a[i]);
    for (int i = 0; i < s->evicted.n; i++) lh_push(&s->cand, -1 - s->evicted.a[i]);
    s->res.n = s->evicted.n = 0;
}
static int cmp_nodescore(const void *x, const void *y)   /* SearchResult.NodeScore.compareTo :101-106 (keys: NodeQueue.encode) */
{
    const int64_t a = *(const int64_t *)x, c = *(const int64_t *)y;
    const float sa = key_score(a), sc = key_score(c);
    if (sa != sc) return sa > sc ? -1 : 1;
    const int32_t na = key_node(a), nc = key_node(c);
    return na < nc ? -1 : (na > nc ? 1 : 0);
}

/* addGraphNode(node, vector) :605-659.  Returns the node's level. */
int jvo_builder_add(jvo_builder *b, int32_t node)
{
    const int level = b->forced_levels ? (int)b->forced_levels[node] : jvo_random_graph_level(&b->rng, b->maxDegree, b->addHierarchy);
    bld_ensure_level(b, level);
    for (int l = 0; l <= level; l++) { b->size[l][node] = 0; b->db[l][node] = 0; }   /* graph.addNode :161-168 */
    b->g.entry_node = b->entry_node; b->g.entry_level = b->entry_level;
    nodearr cand; cand.node = (int32_t *)malloc(sizeof(int32_t) * (size_t)(b->beam + 1)); cand.score = (float *)malloc(sizeof(float) * (size_t)(b->beam + 1));
    cand.size = 0;
    if (b->entry_node >= 0) {
        jvo_searcher *s = b->s;
        bld_search_init(b, b->vecs + (size_t)node * b->pq->D, node);
        for (int lvl = b->entry_level; lvl > 0; lvl--) {
            if (lvl > level) searcher_one_layer(s, 1, 0.0f, lvl, 0);
            else {
                searcher_one_layer(s, b->beam, 0.0f, lvl, 0);
                int64_t *keys = (int64_t *)malloc(sizeof(int64_t) * (size_t)(s->res.n + 1));
                memcpy(keys, s->res.a, sizeof(int64_t) * (size_t)s->res.n);     /* approximateResults.foreach: heap array order */
                qsort(keys, (size_t)s->res.n, sizeof(int64_t), cmp_nodescore);   /* Arrays.sort(neighbors): total order, no ties left */
                nodearr up; up.node = (int32_t *)malloc(sizeof(int32_t) * (size_t)(s->res.n + 1)); up.score = (float *)malloc(sizeof(float) * (size_t)(s->res.n + 1));
                up.size = s->res.n;
                for (int i = 0; i < s->res.n; i++) { up.node[i] = key_node(keys[i]); up.score[i] = key_score(keys[i]); }
                bld_add_edges(b, lvl, node, &up);
                free(keys); free(up.node); free(up.score);
            }
            bld_entry_points_from_previous_layer(s);
        }
        /* gs.resume(beamWidth, beamWidth, 0, 0) :509-512 */
        searcher_layer0(s, b->beam, b->beam, 0.0f);
        int32_t *oi = (int32_t *)malloc(sizeof(int32_t) * (size_t)b->beam); float *os = (float *)malloc(sizeof(float) * (size_t)b->beam);
        const int nres = searcher_reranking(s, b->beam, 0.0f, oi, os, NULL, NULL);
        for (int i = 0; i < nres; i++) { cand.node[i] = oi[i]; cand.score[i] = os[i]; }
        cand.size = nres;
        free(oi); free(os);
    }
    bld_add_edges(b, 0, node, &cand);
    free(cand.node); free(cand.score);
    if (b->entry_node < 0 || level > b->entry_level) { b->entry_node = node; b->entry_level = level; }   /* markComplete :214-225 */
    return level;
}

/* improveConnections(node) :510-545 */
void jvo_builder_improve(jvo_builder *b, int32_t node)
{
    jvo_searcher *s = b->s;
    float *q = (float *)malloc(sizeof(float) * (size_t)b->pq->D);
    if (b->improve_full_vectors) memcpy(q, b->vecs + (size_t)node * b->pq->D, sizeof(float) * (size_t)b->pq->D);
    else jvo_pq_decode(b->pq, b->codes + (size_t)node * b->pq->M, q);   /* searchProviderFor(int node1) :189-194 */
    b->g.entry_node = b->entry_node; b->g.entry_level = b->entry_level;
    bld_search_init(b, q, node);
    for (int lvl = b->entry_level; lvl >= 0; lvl--) {
        if (b->size[lvl][node] > 0) {
            searcher_one_layer(s, b->beam, 0.0f, lvl, 1);
            nodearr c; c.node = (int32_t *)malloc(sizeof(int32_t) * (size_t)(s->res.n + 1)); c.score = (float *)malloc(sizeof(float) * (size_t)(s->res.n + 1));
            c.size = 0;
            int64_t *keys = (int64_t *)malloc(sizeof(int64_t) * (size_t)(s->res.n + 1));
            memcpy(keys, s->res.a, sizeof(int64_t) * (size_t)s->res.n);
            if (b->improve_sorted_candidates) qsort(keys, (size_t)s->res.n, sizeof(int64_t), cmp_nodescore);   /* (deviation) */
            for (int i = 0; i < s->res.n; i++) {   /* approximateResults.foreach(candidates::insertSorted): heap array order */
                const int32_t nd = key_node(keys[i]); const float sc = key_score(keys[i]);
                const int at = na_insertion_point(&c, nd, sc);
                if (at >= 0) na_insert_at(&c, at, nd, sc);
            }
            free(keys);
            bld_add_edges(b, lvl, node, &c);
            free(c.node); free(c.score);
        } else searcher_one_layer(s, 1, 0.0f, lvl, 1);
        bld_entry_points_from_previous_layer(s);
    }
    free(q);
}
/* enforceDegree(node) on every level :79-90, :190-200 */
void jvo_builder_enforce_degree(jvo_builder *b, int32_t node)
{
    for (int l = 0; l < b->n_levels; l++) {
        if (b->size[l][node] <= b->maxDegree) continue;
        nodearr t = bld_row(b, l, node);
        bld_retain(b, &t, b->db[l][node]);
        bld_store(b, l, node, &t, t.size);
    }
}
/* cleanup() :472-508 without deletions: improveConnections over the nodes of level 1 when the graph has levels and refineFinalGraph is
 * set, then enforceDegree over every id (both in ascending id order here; the reference runs them from a parallel stream) */
void jvo_builder_cleanup(jvo_builder *b)
{
    if (b->entry_node < 0) return;
    if (b->refine && b->n_levels - 1 > 0)
        for (int64_t i = 0; i < b->n; i++) if (b->size[1][i] >= 0) jvo_builder_improve(b, (int32_t)i);
    for (int64_t i = 0; i < b->n; i++) if (b->size[0][i] >= 0) jvo_builder_enforce_degree(b, (int32_t)i);
}
/* a node's list on a level: returns its size (-1: not on the level); ids / scores (nullable) get `size` entries, *diverseBefore too */
int jvo_builder_row(const jvo_builder *b, int level, int32_t node, int32_t *ids, float *scores, int *diverseBefore)
{
    if (level < 0 || level >= b->n_levels) return -1;
    const int sz = b->size[level][node];
    if (sz < 0) return -1;
    if (ids) memcpy(ids, b->ids[level] + (size_t)node * b->W, sizeof(int32_t) * (size_t)sz);
    if (scores) memcpy(scores, b->sc[level] + (size_t)node * b->W, sizeof(float) * (size_t)sz);
    if (diverseBefore) *diverseBefore = b->db[level][node];
    return sz;
}
void jvo_builder_info(const jvo_builder *b, int32_t *entry_node, int *entry_level, int *n_levels, int64_t *reprunes)
{
    if (entry_node) *entry_node = b->entry_node;
    if (entry_level) *entry_level = b->entry_level;
    if (n_levels) *n_levels = b->n_levels;
    if (reprunes) *reprunes = b->reprunes;
}

/* NodeArray.insertSorted (:181-193) on caller-owned arrays with room for one more entry — for the reference's TestNodeArray literals.
 * Returns the insertion point, -1 when the (node, score) pair is already listed. */
int jvo_nodearray_insert_sorted(int32_t *nodes, float *scores, int *size, int32_t node, float score)
{
    nodearr a = { nodes, scores, *size };
    const int at = na_insertion_point(&a, node, score);
    if (at < 0) return -1;
    na_insert_at(&a, at, node, score);
    *size = a.size;
    return at;
}

/* NodeArray.merge (:63-143); out arrays hold size1 + size2 entries; returns the merged size */
int jvo_nodearray_merge(const int32_t *n1, const float *s1, int size1, const int32_t *n2, const float *s2, int size2, int32_t *out_n, float *out_s)
{
    nodearr a1 = { (int32_t *)n1, (float *)s1, size1 }, a2 = { (int32_t *)n2, (float *)s2, size2 }, m = { out_n, out_s, 0 };
    na_merge(&a1, &a2, &m);
    return m.size;
}
