/*
 * jv_nvq.c — CPU oracle for NVQ ("NuVeQ", non-uniform vector quantization), the reference's compressed RERANK codec.
 *
 * TEST INFRASTRUCTURE ONLY (see jv_oracle.h): a plain-C restatement of the SCALAR reference path
 *   B/quantization/NVQuantization.java        (compute :153-163, encodeTo :213-216, QuantizedSubVector.quantizeTo :508-557,
 *                                              the loss function :660-702, getSubvectorSizesAndOffsets :236-252)
 *   B/quantization/NVQScorer.java             (dot :51-76, euclidean :78-106, cosine :108-137)
 *   B/vector/DefaultVectorUtilSupport.java    (nvq* :385-548, min / max :365-383)
 * with Java's float semantics: strict binary32, left-to-right, Math.fma where the reference writes Math.fma (fmaf here; the
 * file is compiled with -ffp-contract=off so nothing else fuses), Math.round(float) = floor(x + 1/2) saturating, NaN -> 0,
 * Float.floatToIntBits (NaN canonical), int arithmetic wrapping.
 *
 * Parity pin: the reference holds no literal NVQ vectors; its own test of this path is statistical
 * (TS/quantization/TestCompressedVectors.java:171-228 — mean score error per similarity function under a tolerance, for
 * d = 256..2048, 1/2/4/8 sub-vectors, learn on / off).  tests/test_nvq_cpu.py restates exactly that test on this file;
 * "parity unpinned at the literal-value level" applies as for PQ code bytes (jv_oracle.h).
 */
#include "jv_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

static inline int32_t f2i(float f)   /* Float.floatToIntBits */
{
    if (f != f) return 0x7fc00000;
    int32_t b;
    memcpy(&b, &f, 4);
    return b;
}
static inline float i2f(int32_t b)
{
    float f;
    memcpy(&f, &b, 4);
    return f;
}
static inline int32_t java_round(float x)   /* Math.round(float) */
{
    if (x != x) return 0;
    double r = floor((double)x + 0.5);
    if (r <= -2147483648.0) return INT32_MIN;
    if (r >= 2147483647.0) return INT32_MAX;
    return (int32_t)r;
}
static inline float java_min(float a, float b)   /* Math.min(float, float) */
{
    if (a != a) return a;
    if (b != b) return b;
    if (a == 0.0f && b == 0.0f) return signbit(a) ? a : b;
    return a <= b ? a : b;
}
static inline float java_max(float a, float b)
{
    if (a != a) return a;
    if (b != b) return b;
    if (a == 0.0f && b == 0.0f) return signbit(a) ? b : a;
    return a >= b ? a : b;
}

/* DefaultVectorUtilSupport.java:441-448 */
static inline float logistic_nqt(float value, float alpha, float x0)
{
    float temp = fmaf(value, alpha, -alpha * x0);
    int32_t p = java_round(temp + 0.5f);
    int32_t m = f2i(fmaf(temp - (float)p, 0.5f, 1.0f));
    temp = i2f((int32_t)((uint32_t)m + ((uint32_t)p << 23)));
    return temp / (temp + 1.0f);
}
/* :450-459 */
static inline float logit_nqt(float value, float inverseAlpha, float x0)
{
    float z = value / (1.0f - value);
    int32_t temp = f2i(z);
    int32_t e = temp & 0x7f800000;
    float p = (float)((e >> 23) - 128);
    float m = i2f((temp & 0x007fffff) + 0x3f800000);
    return fmaf(m + p, inverseAlpha, x0);
}
/* :461-464 */
static inline float scaled_logistic(float v, float growth, float mid, float scale, float bias)
{
    float y = logistic_nqt(v, growth, mid);
    return (y - bias) * (1.0f / scale);
}
/* :466-469 */
static inline float scaled_logit_nqt(float v, float invGrowth, float mid, float scale, float bias)
{
    float sv = fmaf(v, scale, bias);
    return logit_nqt(sv, invGrowth, mid);
}

/* the five derived numbers every nvq* function starts with (:386-391 and its copies) */
void jvo_nvq_derive(float growthRate, float midpoint, float minValue, float maxValue, float levels, float *out5)
{
    float delta = maxValue - minValue;
    float sgr = growthRate / delta;
    float smid = midpoint * delta;
    float inv = 1.0f / sgr;
    float bias = logistic_nqt(minValue, sgr, smid);
    float scale = (logistic_nqt(maxValue, sgr, smid) - bias) / levels;
    out5[0] = sgr; out5[1] = smid; out5[2] = inv; out5[3] = bias; out5[4] = scale;
}

float jvo_nvq_min(const float *v, int n)   /* :377-383 */
{
    float m = 3.4028234663852886e38f;
    for (int i = 0; i < n; i++) m = java_min(m, v[i]);
    return m;
}
float jvo_nvq_max(const float *v, int n)   /* :368-374 */
{
    float m = -3.4028234663852886e38f;
    for (int i = 0; i < n; i++) m = java_max(m, v[i]);
    return m;
}

/* nvqQuantize8bit :471-488 */
void jvo_nvq_quantize_8bit(const float *v, int n, float growthRate, float midpoint, float minValue, float maxValue, uint8_t *dst)
{
    float p[5];
    jvo_nvq_derive(growthRate, midpoint, minValue, maxValue, 255.0f, p);
    for (int d = 0; d < n; d++) {
        float value = scaled_logistic(v[d], p[0], p[1], p[4], p[3]);
        dst[d] = (uint8_t)((uint32_t)java_round(value) & 0xffu);
    }
}
/* nvqLoss :490-516 */
float jvo_nvq_loss(const float *v, int n, float growthRate, float midpoint, float minValue, float maxValue, int nBits)
{
    float p[5];
    jvo_nvq_derive(growthRate, midpoint, minValue, maxValue, (float)((1 << nBits) - 1), p);
    float sq = 0.0f;
    for (int d = 0; d < n; d++) {
        float r = scaled_logistic(v[d], p[0], p[1], p[4], p[3]);
        r = (float)java_round(r);
        r = scaled_logit_nqt(r, p[2], p[1], p[4], p[3]);
        float diff = v[d] - r;
        sq = fmaf(diff, diff, sq);
    }
    return sq;
}
/* nvqUniformLoss :518-536 */
float jvo_nvq_uniform_loss(const float *v, int n, float minValue, float maxValue, int nBits)
{
    float constant = (float)((1 << nBits) - 1), sq = 0.0f;
    for (int d = 0; d < n; d++) {
        float r = (v[d] - minValue) / (maxValue - minValue);
        r = (float)java_round(constant * r) / constant;
        r = r * (maxValue - minValue) + minValue;
        float diff = v[d] - r;
        sq = fmaf(diff, diff, sq);
    }
    return sq;
}
/* nvqDotProduct8bit :385-404 */
float jvo_nvq_dot_8bit(const float *q, const uint8_t *bytes, int n, float growthRate, float midpoint, float minValue, float maxValue)
{
    float p[5];
    jvo_nvq_derive(growthRate, midpoint, minValue, maxValue, 255.0f, p);
    float dp = 0.0f;
    for (int d = 0; d < n; d++) dp = fmaf(q[d], scaled_logit_nqt((float)bytes[d], p[2], p[1], p[4], p[3]), dp);
    return dp;
}
/* nvqSquareL2Distance8bit :406-428 */
float jvo_nvq_l2_8bit(const float *q, const uint8_t *bytes, int n, float growthRate, float midpoint, float minValue, float maxValue)
{
    float p[5];
    jvo_nvq_derive(growthRate, midpoint, minValue, maxValue, 255.0f, p);
    float sq = 0.0f;
    for (int d = 0; d < n; d++) {
        float t = scaled_logit_nqt((float)bytes[d], p[2], p[1], p[4], p[3]) - q[d];
        sq = fmaf(t, t, sq);
    }
    return sq;
}
/* nvqCosine8bit :430-454 -> {sum, normDQ} */
void jvo_nvq_cosine_8bit(const float *q, const uint8_t *bytes, int n, float growthRate, float midpoint, float minValue, float maxValue,
                         const float *centroid, float *out2)
{
    float p[5];
    jvo_nvq_derive(growthRate, midpoint, minValue, maxValue, 255.0f, p);
    float sum = 0.0f, norm = 0.0f;
    for (int d = 0; d < n; d++) {
        float e = scaled_logit_nqt((float)bytes[d], p[2], p[1], p[4], p[3]);
        e += centroid[d];
        sum = fmaf(q[d], e, sum);
        norm = fmaf(e, e, norm);
    }
    out2[0] = sum;
    out2[1] = norm;
}
/* the de-quantized value of one byte — what the three functions above feed their chains with */
float jvo_nvq_dequantize(uint8_t b, float growthRate, float midpoint, float minValue, float maxValue)
{
    float p[5];
    jvo_nvq_derive(growthRate, midpoint, minValue, maxValue, 255.0f, p);
    return scaled_logit_nqt((float)b, p[2], p[1], p[4], p[3]);
}

/* NVQuantization.compute :153-163: addInPlace over the vectors in order, then scale by 1.0f / n */
void jvo_nvq_global_mean(const float *X, int64_t n, int D, float *out)
{
    for (int j = 0; j < D; j++) out[j] = 0.0f;
    for (int64_t i = 0; i < n; i++)
        for (int j = 0; j < D; j++) out[j] = out[j] + X[i * D + j];
    float mult = 1.0f / (float)n;   /* int size -> float */
    for (int j = 0; j < D; j++) out[j] = out[j] * mult;
}

/* QuantizedSubVector.quantizeTo :508-557.  params = {minValue, maxValue, growthRate, midpoint} in the order
 * QuantizedSubVector.write :577-587 serialises them. */
void jvo_nvq_encode_sub(const float *v, int n, int learn, uint8_t *bytes, float *params)
{
    float minValue = jvo_nvq_min(v, n), maxValue = jvo_nvq_max(v, n);
    float growthRate = 1e-2f, midpoint = 0.0f;
    if (learn) {
        float baseline = jvo_nvq_uniform_loss(v, n, minValue, maxValue, 8);   /* setVector :672-677 */
        float coarse = 1e-2f;
        float best = 1.401298464324817e-45f;                                  /* Float.MIN_VALUE */
        for (float gr = 1e-6f; gr < 20.0f; gr += 1.0f) {
            float loss = baseline / jvo_nvq_loss(v, n, gr, 0.0f, minValue, maxValue, 8);   /* compute :683-685 */
            if (loss > best) { best = loss; coarse = gr; }
        }
        float fine = coarse;
        for (float gr = coarse - 1.0f; gr < coarse + 1.0f; gr += 0.1f) {
            float loss = baseline / jvo_nvq_loss(v, n, gr, 0.0f, minValue, maxValue, 8);
            if (loss > best) { best = loss; fine = gr; }
        }
        growthRate = fine;
    }
    jvo_nvq_quantize_8bit(v, n, growthRate, midpoint, minValue, maxValue, bytes);
    params[0] = minValue; params[1] = maxValue; params[2] = growthRate; params[3] = midpoint;
}

/* the growth rates the two loops above visit, for a device implementation that wants them as a table:
 * coarse[20]; fine[c][0..fine_n[c]) for the c-th coarse value.  Returns the number of coarse values. */
int jvo_nvq_growth_grid(float *coarse, float *fine, int *fine_n, int fine_stride)
{
    int nc = 0;
    for (float gr = 1e-6f; gr < 20.0f; gr += 1.0f) {
        coarse[nc] = gr;
        int nf = 0;
        for (float g2 = gr - 1.0f; g2 < gr + 1.0f; g2 += 0.1f) {
            if (nf < fine_stride) fine[nc * fine_stride + nf] = g2;
            nf++;
        }
        fine_n[nc] = nf;
        nc++;
    }
    return nc;
}

/* NVQuantization.encodeTo :213-216: v - globalMean (VectorUtil.sub), then every sub-vector on its own.
 * bytes: D (sub-vectors concatenated), params: S x 4 */
void jvo_nvq_encode(const float *mean, int D, int S, const float *vec, int learn, uint8_t *bytes, float *params)
{
    int *sizes = (int *)malloc(sizeof(int) * 2 * (size_t)S), *offs = sizes + S;
    float *tmp = (float *)malloc(sizeof(float) * (size_t)D);
    jvo_subvector_sizes_offsets(D, S, sizes, offs);   /* NVQuantization.getSubvectorSizesAndOffsets :236-252: same split */
    for (int j = 0; j < D; j++) tmp[j] = vec[j] - mean[j];
    for (int s = 0; s < S; s++) jvo_nvq_encode_sub(tmp + offs[s], sizes[s], learn, bytes + offs[s], params + 4 * s);
    free(tmp);
    free(sizes);
}

typedef struct { const float *mean; int D, S, learn; const float *X; int64_t lo, hi; uint8_t *bytes; float *params; } enc_job;
static void *enc_worker(void *a)
{
    enc_job *j = (enc_job *)a;
    for (int64_t i = j->lo; i < j->hi; i++)
        jvo_nvq_encode(j->mean, j->D, j->S, j->X + i * j->D, j->learn, j->bytes + i * j->D, j->params + i * 4 * j->S);
    return NULL;
}
/* NVQuantization.encodeAll :182-195 (one vector per task; the result does not depend on the schedule) */
void jvo_nvq_encode_all(const float *mean, int D, int S, const float *X, int64_t n, int learn, uint8_t *bytes, float *params, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    pthread_t th[64];
    enc_job jobs[64];
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = (enc_job){mean, D, S, learn, X, n * t / nthreads, n * (t + 1) / nthreads, bytes, params};
        pthread_create(&th[t], NULL, enc_worker, &jobs[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
}

/* NVQScorer.scoreFunctionFor(query, vsf).similarityTo(vector2) :33-137 */
float jvo_nvq_score(int vsf, const float *mean, int D, int S, const float *query, const uint8_t *bytes, const float *params)
{
    int *sizes = (int *)malloc(sizeof(int) * 2 * (size_t)S), *offs = sizes + S;
    jvo_subvector_sizes_offsets(D, S, sizes, offs);
    float result;
    if (vsf == JVO_DOT_PRODUCT) {
        float bias = jvo_dot(query, mean, D);   /* VectorUtil.dotProduct(query, globalMean) :55 */
        float nvqDot = 0.0f;
        for (int s = 0; s < S; s++) {
            const float *p = params + 4 * s;
            nvqDot += jvo_nvq_dot_8bit(query + offs[s], bytes + offs[s], sizes[s], p[2], p[3], p[0], p[1]);
        }
        result = (1.0f + nvqDot + bias) / 2.0f;
    } else if (vsf == JVO_EUCLIDEAN) {
        float *sh = (float *)malloc(sizeof(float) * (size_t)D);
        for (int j = 0; j < D; j++) sh[j] = query[j] - mean[j];   /* VectorUtil.sub :82 */
        float dist = 0.0f;
        for (int s = 0; s < S; s++) {
            const float *p = params + 4 * s;
            dist += jvo_nvq_l2_8bit(sh + offs[s], bytes + offs[s], sizes[s], p[2], p[3], p[0], p[1]);
        }
        free(sh);
        result = 1.0f / (1.0f + dist);
    } else {
        float queryNorm = (float)sqrt((double)jvo_dot(query, query, D));   /* :109 */
        float cos = 0.0f, sqn = 0.0f;
        for (int s = 0; s < S; s++) {
            const float *p = params + 4 * s;
            float part[2];
            jvo_nvq_cosine_8bit(query + offs[s], bytes + offs[s], sizes[s], p[2], p[3], p[0], p[1], mean + offs[s], part);
            cos += part[0];
            sqn += part[1];
        }
        float cosine = (cos / queryNorm) / (float)sqrt((double)sqn);
        result = (1.0f + cosine) / 2.0f;
    }
    free(sizes);
    return result;
}

/* out[q][b] = score of query q against row ids[q*B + b] (ids outside [0, n) -> -inf, the engine's convention for padding) */
void jvo_nvq_scores(int vsf, const float *mean, int D, int S, const float *queries, int Q, const uint8_t *bytes, const float *params,
                    int64_t n, const int32_t *ids, int B, float *out)
{
    for (int q = 0; q < Q; q++)
        for (int b = 0; b < B; b++) {
            int32_t id = ids[(size_t)q * B + b];
            out[(size_t)q * B + b] = (id < 0 || id >= n) ? -INFINITY
                : jvo_nvq_score(vsf, mean, D, S, queries + (size_t)q * D, bytes + (size_t)id * D, params + (size_t)id * 4 * S);
        }
}

/* NVQuantization.reconstructionError :381-408 */
double jvo_nvq_reconstruction_error(const float *mean, int D, int S, const float *vec, int learn)
{
    int *sizes = (int *)malloc(sizeof(int) * 2 * (size_t)S), *offs = sizes + S;
    jvo_subvector_sizes_offsets(D, S, sizes, offs);
    float *tmp = (float *)malloc(sizeof(float) * (size_t)D);
    uint8_t *bytes = (uint8_t *)malloc((size_t)D);
    float *params = (float *)malloc(sizeof(float) * 4 * (size_t)S);
    for (int j = 0; j < D; j++) tmp[j] = vec[j] - mean[j];
    for (int s = 0; s < S; s++) jvo_nvq_encode_sub(tmp + offs[s], sizes[s], learn, bytes + offs[s], params + 4 * s);
    float dist = 0.0f;
    for (int s = 0; s < S; s++) {
        const float *p = params + 4 * s;
        dist += jvo_nvq_l2_8bit(tmp + offs[s], bytes + offs[s], sizes[s], p[2], p[3], p[0], p[1]);
    }
    free(params); free(bytes); free(tmp); free(sizes);
    return (double)(dist / (float)D);
}

/* ---- the reranker switch of the search entry points in jv_oracle.c (NVQ.rerankerFor, B/graph/disk/feature/NVQ.java:96-110) ---- */
static const uint8_t *g_rr_bytes;
static const float *g_rr_params, *g_rr_mean;
static int g_rr_D, g_rr_S;
void jvo_set_nvq_reranker(const uint8_t *bytes, const float *params, const float *mean, int D, int S)
{
    g_rr_bytes = bytes; g_rr_params = params; g_rr_mean = mean; g_rr_D = D; g_rr_S = S;
}
int jvo_nvq_reranker_active(void) { return g_rr_bytes != NULL; }
float jvo_nvq_rerank_score(int vsf, const float *query, int32_t node)
{
    return jvo_nvq_score(vsf, g_rr_mean, g_rr_D, g_rr_S, query, g_rr_bytes + (size_t)node * g_rr_D, g_rr_params + (size_t)node * 4 * g_rr_S);
}
