/* jv_oracle_simd.c — TEST INFRASTRUCTURE ONLY (like everything under oracle/): an x86 SIMD restatement of the
 * reference's NATIVE kernels, used for ONE thing: bench.py's `cpu_baseline` leg ("C++ restatement of
 * the reference's native SIMD algorithm ... same loop/accumulator structure").  It is NOT the parity checker — the
 * scalar functions in jv_oracle.c are; these differ from them in the last bits because they accumulate lane-wise with
 * fused multiply-adds, exactly as the reference's native library differs from its own scalar Java path
 * (tolerance 1e-4 relative in the reference's tests, NC/tests/test_similarity.cpp:54-80).
 *
 * The reference's native library is written against google/highway (un-vendored submodule, absent here) and cannot
 * be built, so the Highway ops are restated with the AVX2 / AVX-512 intrinsics they lower to:
 *   DotProductImpl / L2SquareDistanceImpl   NC/src/jvector_simd_kernels.cpp:208-258  4 accumulators x MulAdd, one-vector
 *                                           loop, masked tail (LoadN), ReduceSum
 *   CosineDistanceImpl                      :262-286  three accumulators, sum_ab / sqrtf(sum_aa * sum_bb)
 *   calculate_partial_sums_f32, size == 8   :609-633  LoadDup256(query), lanes/8 centroids per vector, SwapAdjacentBlocks +
 *                                           Shuffle1032 + Shuffle2301 horizontal adds; other sizes: per-centroid distance
 *   AssembleAndSumImpl                      :670-704  u8 -> i32 promote, GatherIndex, lane-wise Add, ReduceSum, scalar tail
 *   pq_decoded_cosine_similarity_f32        :821-879  two gathers per step, sum / sqrtf(aMag * bMag)
 *   ISA tier chosen once at first use       NC/src/jvector_simd.cpp:120-167 (AVX3 > AVX2 > baseline); JVO_SIMD_TIER=avx2|scalar
 *                                           caps it like the reference's JVECTOR_MAX_ISA
 * The reference's HWY_CAPPED short-vector paths (lengths <= 4 / 8) only change which register width is used; they are
 * not restated.  Compiled with -ffp-contract=off like the rest of the oracle: every FMA here is an explicit intrinsic. */
#include "jv_oracle.h"

#include <immintrin.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { TIER_SCALAR = 0, TIER_AVX2 = 2, TIER_AVX512 = 3 };

static int detect_tier(void)
{
    int t = TIER_SCALAR;
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) t = TIER_AVX2;
    if (t == TIER_AVX2 && __builtin_cpu_supports("avx512f")) t = TIER_AVX512;
    const char *cap = getenv("JVO_SIMD_TIER");
    if (cap) {
        if (!strcmp(cap, "scalar")) t = TIER_SCALAR;
        else if (!strcmp(cap, "avx2") && t > TIER_AVX2) t = TIER_AVX2;
    }
    return t;
}

int jvs_tier(void)
{
    static int tier = -1;
    if (tier < 0) tier = detect_tier();
    return tier;
}

const char *jvs_tier_name(void)
{
    switch (jvs_tier()) {
    case TIER_AVX512: return "avx512";
    case TIER_AVX2: return "avx2";
    default: return "scalar";
    }
}

/* ------------------------------------------------------------------------------------------ AVX2 (8 lanes) */
#define T2 __attribute__((target("avx2,fma")))

static const int32_t k_mask_tab[16] = {-1, -1, -1, -1, -1, -1, -1, -1, 0, 0, 0, 0, 0, 0, 0, 0};

T2 static inline __m256i tail_mask8(int rem) { return _mm256_loadu_si256((const __m256i *)(k_mask_tab + 8 - rem)); }

T2 static inline float hsum8(__m256 v)
{
    __m128 s = _mm_add_ps(_mm256_castps256_ps128(v), _mm256_extractf128_ps(v, 1));
    s = _mm_add_ps(s, _mm_movehl_ps(s, s));
    s = _mm_add_ss(s, _mm_shuffle_ps(s, s, 0x55));
    return _mm_cvtss_f32(s);
}

T2 static float dot_avx2(const float *a, const float *b, int n)
{
    __m256 a0 = _mm256_setzero_ps(), a1 = a0, a2 = a0, a3 = a0;
    int i = 0;
    for (; i + 32 <= n; i += 32) {
        a0 = _mm256_fmadd_ps(_mm256_loadu_ps(a + i), _mm256_loadu_ps(b + i), a0);
        a1 = _mm256_fmadd_ps(_mm256_loadu_ps(a + i + 8), _mm256_loadu_ps(b + i + 8), a1);
        a2 = _mm256_fmadd_ps(_mm256_loadu_ps(a + i + 16), _mm256_loadu_ps(b + i + 16), a2);
        a3 = _mm256_fmadd_ps(_mm256_loadu_ps(a + i + 24), _mm256_loadu_ps(b + i + 24), a3);
    }
    __m256 acc = _mm256_add_ps(_mm256_add_ps(a0, a1), _mm256_add_ps(a2, a3));
    for (; i + 8 <= n; i += 8) acc = _mm256_fmadd_ps(_mm256_loadu_ps(a + i), _mm256_loadu_ps(b + i), acc);
    if (i < n) {
        const __m256i m = tail_mask8(n - i);
        acc = _mm256_fmadd_ps(_mm256_maskload_ps(a + i, m), _mm256_maskload_ps(b + i, m), acc);
    }
    return hsum8(acc);
}

T2 static float l2_avx2(const float *a, const float *b, int n)
{
    __m256 a0 = _mm256_setzero_ps(), a1 = a0, a2 = a0, a3 = a0;
    int i = 0;
    for (; i + 32 <= n; i += 32) {
        const __m256 d0 = _mm256_sub_ps(_mm256_loadu_ps(a + i), _mm256_loadu_ps(b + i));
        const __m256 d1 = _mm256_sub_ps(_mm256_loadu_ps(a + i + 8), _mm256_loadu_ps(b + i + 8));
        const __m256 d2 = _mm256_sub_ps(_mm256_loadu_ps(a + i + 16), _mm256_loadu_ps(b + i + 16));
        const __m256 d3 = _mm256_sub_ps(_mm256_loadu_ps(a + i + 24), _mm256_loadu_ps(b + i + 24));
        a0 = _mm256_fmadd_ps(d0, d0, a0);
        a1 = _mm256_fmadd_ps(d1, d1, a1);
        a2 = _mm256_fmadd_ps(d2, d2, a2);
        a3 = _mm256_fmadd_ps(d3, d3, a3);
    }
    __m256 acc = _mm256_add_ps(_mm256_add_ps(a0, a1), _mm256_add_ps(a2, a3));
    for (; i + 8 <= n; i += 8) {
        const __m256 d = _mm256_sub_ps(_mm256_loadu_ps(a + i), _mm256_loadu_ps(b + i));
        acc = _mm256_fmadd_ps(d, d, acc);
    }
    if (i < n) {
        const __m256i m = tail_mask8(n - i);
        const __m256 d = _mm256_sub_ps(_mm256_maskload_ps(a + i, m), _mm256_maskload_ps(b + i, m));
        acc = _mm256_fmadd_ps(d, d, acc);
    }
    return hsum8(acc);
}

T2 static float cosine_avx2(const float *a, const float *b, int n)
{
    __m256 ab = _mm256_setzero_ps(), aa = ab, bb = ab;
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        const __m256 va = _mm256_loadu_ps(a + i), vb = _mm256_loadu_ps(b + i);
        ab = _mm256_fmadd_ps(va, vb, ab);
        aa = _mm256_fmadd_ps(va, va, aa);
        bb = _mm256_fmadd_ps(vb, vb, bb);
    }
    if (i < n) {
        const __m256i m = tail_mask8(n - i);
        const __m256 va = _mm256_maskload_ps(a + i, m), vb = _mm256_maskload_ps(b + i, m);
        ab = _mm256_fmadd_ps(va, vb, ab);
        aa = _mm256_fmadd_ps(va, va, aa);
        bb = _mm256_fmadd_ps(vb, vb, bb);
    }
    return hsum8(ab) / sqrtf(hsum8(aa) * hsum8(bb));
}

/* size == 8: one centroid per 256-bit vector; the three shuffle+add steps leave the sum in every lane */
T2 static void partial_sums8_avx2(const float *cb, int k, const float *q, int l2, float *out)
{
    const __m256 qv = _mm256_loadu_ps(q);
    for (int c = 0; c < k; ++c) {
        const __m256 cv = _mm256_loadu_ps(cb + (size_t)c * 8);
        __m256 s;
        if (l2) {
            const __m256 d = _mm256_sub_ps(cv, qv);
            s = _mm256_mul_ps(d, d);
        } else {
            s = _mm256_mul_ps(cv, qv);
        }
        s = _mm256_add_ps(s, _mm256_permute2f128_ps(s, s, 0x01)); /* SwapAdjacentBlocks */
        s = _mm256_add_ps(s, _mm256_shuffle_ps(s, s, 0x4E));      /* Shuffle1032 */
        s = _mm256_add_ps(s, _mm256_shuffle_ps(s, s, 0xB1));      /* Shuffle2301 */
        out[c] = _mm256_cvtss_f32(s);
    }
}

T2 static float assemble_avx2(const float *data, int dataBase, const uint8_t *offs, int len)
{
    const __m256i scale = _mm256_mullo_epi32(_mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7), _mm256_set1_epi32(dataBase));
    __m256 sum = _mm256_setzero_ps();
    int i = 0;
    for (; i + 8 <= len; i += 8) {
        const __m256i off = _mm256_cvtepu8_epi32(_mm_loadl_epi64((const __m128i *)(offs + i)));
        const __m256i idx = _mm256_add_epi32(_mm256_add_epi32(_mm256_set1_epi32(i * dataBase), scale), off);
        sum = _mm256_add_ps(sum, _mm256_i32gather_ps(data, idx, 4));
    }
    float res = hsum8(sum);
    for (; i < len; ++i) res += data[dataBase * i + offs[i]];
    return res;
}

T2 static float pq_cosine_avx2(const uint8_t *offs, int len, int k, const float *lut, const float *amag, float bmag)
{
    const __m256i scale = _mm256_mullo_epi32(_mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7), _mm256_set1_epi32(k));
    __m256 sum = _mm256_setzero_ps(), mag = sum;
    int i = 0;
    for (; i + 8 <= len; i += 8) {
        const __m256i off = _mm256_cvtepu8_epi32(_mm_loadl_epi64((const __m128i *)(offs + i)));
        const __m256i idx = _mm256_add_epi32(_mm256_add_epi32(_mm256_set1_epi32(i * k), scale), off);
        sum = _mm256_add_ps(sum, _mm256_i32gather_ps(lut, idx, 4));
        mag = _mm256_add_ps(mag, _mm256_i32gather_ps(amag, idx, 4));
    }
    float s = hsum8(sum), a = hsum8(mag);
    for (; i < len; ++i) {
        s += lut[k * i + offs[i]];
        a += amag[k * i + offs[i]];
    }
    return s / sqrtf(a * bmag);
}

/* ------------------------------------------------------------------------------------------ AVX-512 (16 lanes) */
#define T3 __attribute__((target("avx512f,avx2,fma")))

T3 static inline float hsum16(__m512 v)
{
    const __m256 lo = _mm512_castps512_ps256(v);
    const __m256 hi = _mm256_castpd_ps(_mm512_extractf64x4_pd(_mm512_castps_pd(v), 1));
    return hsum8(_mm256_add_ps(lo, hi));
}

T3 static float dot_avx512(const float *a, const float *b, int n)
{
    __m512 a0 = _mm512_setzero_ps(), a1 = a0, a2 = a0, a3 = a0;
    int i = 0;
    for (; i + 64 <= n; i += 64) {
        a0 = _mm512_fmadd_ps(_mm512_loadu_ps(a + i), _mm512_loadu_ps(b + i), a0);
        a1 = _mm512_fmadd_ps(_mm512_loadu_ps(a + i + 16), _mm512_loadu_ps(b + i + 16), a1);
        a2 = _mm512_fmadd_ps(_mm512_loadu_ps(a + i + 32), _mm512_loadu_ps(b + i + 32), a2);
        a3 = _mm512_fmadd_ps(_mm512_loadu_ps(a + i + 48), _mm512_loadu_ps(b + i + 48), a3);
    }
    __m512 acc = _mm512_add_ps(_mm512_add_ps(a0, a1), _mm512_add_ps(a2, a3));
    for (; i + 16 <= n; i += 16) acc = _mm512_fmadd_ps(_mm512_loadu_ps(a + i), _mm512_loadu_ps(b + i), acc);
    if (i < n) {
        const __mmask16 m = (__mmask16)((1u << (n - i)) - 1u);
        acc = _mm512_fmadd_ps(_mm512_maskz_loadu_ps(m, a + i), _mm512_maskz_loadu_ps(m, b + i), acc);
    }
    return hsum16(acc);
}

T3 static float l2_avx512(const float *a, const float *b, int n)
{
    __m512 a0 = _mm512_setzero_ps(), a1 = a0, a2 = a0, a3 = a0;
    int i = 0;
    for (; i + 64 <= n; i += 64) {
        const __m512 d0 = _mm512_sub_ps(_mm512_loadu_ps(a + i), _mm512_loadu_ps(b + i));
        const __m512 d1 = _mm512_sub_ps(_mm512_loadu_ps(a + i + 16), _mm512_loadu_ps(b + i + 16));
        const __m512 d2 = _mm512_sub_ps(_mm512_loadu_ps(a + i + 32), _mm512_loadu_ps(b + i + 32));
        const __m512 d3 = _mm512_sub_ps(_mm512_loadu_ps(a + i + 48), _mm512_loadu_ps(b + i + 48));
        a0 = _mm512_fmadd_ps(d0, d0, a0);
        a1 = _mm512_fmadd_ps(d1, d1, a1);
        a2 = _mm512_fmadd_ps(d2, d2, a2);
        a3 = _mm512_fmadd_ps(d3, d3, a3);
    }
    __m512 acc = _mm512_add_ps(_mm512_add_ps(a0, a1), _mm512_add_ps(a2, a3));
    for (; i + 16 <= n; i += 16) {
        const __m512 d = _mm512_sub_ps(_mm512_loadu_ps(a + i), _mm512_loadu_ps(b + i));
        acc = _mm512_fmadd_ps(d, d, acc);
    }
    if (i < n) {
        const __mmask16 m = (__mmask16)((1u << (n - i)) - 1u);
        const __m512 d = _mm512_sub_ps(_mm512_maskz_loadu_ps(m, a + i), _mm512_maskz_loadu_ps(m, b + i));
        acc = _mm512_fmadd_ps(d, d, acc);
    }
    return hsum16(acc);
}

T3 static float cosine_avx512(const float *a, const float *b, int n)
{
    __m512 ab = _mm512_setzero_ps(), aa = ab, bb = ab;
    int i = 0;
    for (; i + 16 <= n; i += 16) {
        const __m512 va = _mm512_loadu_ps(a + i), vb = _mm512_loadu_ps(b + i);
        ab = _mm512_fmadd_ps(va, vb, ab);
        aa = _mm512_fmadd_ps(va, va, aa);
        bb = _mm512_fmadd_ps(vb, vb, bb);
    }
    if (i < n) {
        const __mmask16 m = (__mmask16)((1u << (n - i)) - 1u);
        const __m512 va = _mm512_maskz_loadu_ps(m, a + i), vb = _mm512_maskz_loadu_ps(m, b + i);
        ab = _mm512_fmadd_ps(va, vb, ab);
        aa = _mm512_fmadd_ps(va, va, aa);
        bb = _mm512_fmadd_ps(vb, vb, bb);
    }
    return hsum16(ab) / sqrtf(hsum16(aa) * hsum16(bb));
}

/* size == 8: two centroids per 512-bit vector, query duplicated into both halves; sums land in lanes 0 and 8 */
T3 static void partial_sums8_avx512(const float *cb, int k, const float *q, int l2, float *out)
{
    const __m256 q8 = _mm256_loadu_ps(q);
    const __m512 qv = _mm512_castpd_ps(_mm512_insertf64x4(_mm512_castps_pd(_mm512_castps256_ps512(q8)), _mm256_castps_pd(q8), 1));
    float tmp[16] __attribute__((aligned(64)));
    int c = 0;
    for (; c + 2 <= k; c += 2) {
        const __m512 cv = _mm512_loadu_ps(cb + (size_t)c * 8);
        __m512 s;
        if (l2) {
            const __m512 d = _mm512_sub_ps(cv, qv);
            s = _mm512_mul_ps(d, d);
        } else {
            s = _mm512_mul_ps(cv, qv);
        }
        s = _mm512_add_ps(s, _mm512_shuffle_f32x4(s, s, 0xB1)); /* SwapAdjacentBlocks */
        s = _mm512_add_ps(s, _mm512_shuffle_ps(s, s, 0x4E));    /* Shuffle1032 */
        s = _mm512_add_ps(s, _mm512_shuffle_ps(s, s, 0xB1));    /* Shuffle2301 */
        _mm512_store_ps(tmp, s);
        out[c] = tmp[0];
        out[c + 1] = tmp[8];
    }
    for (; c < k; ++c) out[c] = l2 ? l2_avx512(cb + (size_t)c * 8, q, 8) : dot_avx512(cb + (size_t)c * 8, q, 8);
}

T3 static float assemble_avx512(const float *data, int dataBase, const uint8_t *offs, int len)
{
    const __m512i scale = _mm512_mullo_epi32(_mm512_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15),
                                             _mm512_set1_epi32(dataBase));
    __m512 sum = _mm512_setzero_ps();
    int i = 0;
    for (; i + 16 <= len; i += 16) {
        const __m512i off = _mm512_cvtepu8_epi32(_mm_loadu_si128((const __m128i *)(offs + i)));
        const __m512i idx = _mm512_add_epi32(_mm512_add_epi32(_mm512_set1_epi32(i * dataBase), scale), off);
        sum = _mm512_add_ps(sum, _mm512_i32gather_ps(idx, data, 4));
    }
    float res = hsum16(sum);
    for (; i < len; ++i) res += data[dataBase * i + offs[i]];
    return res;
}

T3 static float pq_cosine_avx512(const uint8_t *offs, int len, int k, const float *lut, const float *amag, float bmag)
{
    const __m512i scale = _mm512_mullo_epi32(_mm512_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15),
                                             _mm512_set1_epi32(k));
    __m512 sum = _mm512_setzero_ps(), mag = sum;
    int i = 0;
    for (; i + 16 <= len; i += 16) {
        const __m512i off = _mm512_cvtepu8_epi32(_mm_loadu_si128((const __m128i *)(offs + i)));
        const __m512i idx = _mm512_add_epi32(_mm512_add_epi32(_mm512_set1_epi32(i * k), scale), off);
        sum = _mm512_add_ps(sum, _mm512_i32gather_ps(idx, lut, 4));
        mag = _mm512_add_ps(mag, _mm512_i32gather_ps(idx, amag, 4));
    }
    float s = hsum16(sum), a = hsum16(mag);
    for (; i < len; ++i) {
        s += lut[k * i + offs[i]];
        a += amag[k * i + offs[i]];
    }
    return s / sqrtf(a * bmag);
}

/* ------------------------------------------------------------------------------------------ dispatch */
float jvs_dot(const float *a, const float *b, int n)
{
    switch (jvs_tier()) {
    case TIER_AVX512: return dot_avx512(a, b, n);
    case TIER_AVX2: return dot_avx2(a, b, n);
    default: return jvo_dot_off(a, 0, b, 0, n);
    }
}

float jvs_l2(const float *a, const float *b, int n)
{
    switch (jvs_tier()) {
    case TIER_AVX512: return l2_avx512(a, b, n);
    case TIER_AVX2: return l2_avx2(a, b, n);
    default: return jvo_l2_off(a, 0, b, 0, n);
    }
}

float jvs_cosine(const float *a, const float *b, int n)
{
    switch (jvs_tier()) {
    case TIER_AVX512: return cosine_avx512(a, b, n);
    case TIER_AVX2: return cosine_avx2(a, b, n);
    default: return jvo_cosine_off(a, 0, b, 0, n);
    }
}

/* VectorSimilarityFunction.compare on the native kernels */
float jvs_compare(int vsf, const float *a, const float *b, int n)
{
    if (vsf == JVO_EUCLIDEAN) return jvo_score_from_raw(vsf, jvs_l2(a, b, n));
    if (vsf == JVO_DOT_PRODUCT) return jvo_score_from_raw(vsf, jvs_dot(a, b, n));
    return jvo_score_from_raw(vsf, jvs_cosine(a, b, n));
}

/* calculate_partial_sums_{dot,euclidean}_f32: out[cbIndex * k + c] for c in [0, k) */
void jvs_calculate_partial_sums(const float *codebook, int cbIndex, int size, int k, const float *query, int qoff, int vsf,
                                float *out)
{
    const int tier = jvs_tier();
    const int l2 = vsf == JVO_EUCLIDEAN;
    float *dst = out + (size_t)cbIndex * k;
    if (tier == TIER_SCALAR) {
        jvo_calculate_partial_sums(codebook, cbIndex, size, k, query, qoff, vsf, out);
    } else if (size == 8) {
        if (tier == TIER_AVX512) partial_sums8_avx512(codebook, k, query + qoff, l2, dst);
        else partial_sums8_avx2(codebook, k, query + qoff, l2, dst);
    } else {
        for (int c = 0; c < k; ++c)
            dst[c] = l2 ? jvs_l2(codebook + (size_t)c * size, query + qoff, size) : jvs_dot(codebook + (size_t)c * size, query + qoff, size);
    }
}

float jvs_assemble_and_sum(const float *data, int dataBase, const uint8_t *offs, int len)
{
    switch (jvs_tier()) {
    case TIER_AVX512: return assemble_avx512(data, dataBase, offs, len);
    case TIER_AVX2: return assemble_avx2(data, dataBase, offs, len);
    default: return jvo_assemble_and_sum(data, dataBase, offs, 0, len);
    }
}

float jvs_pq_decoded_cosine(const uint8_t *offs, int len, int k, const float *lut, const float *amag, float bmag)
{
    switch (jvs_tier()) {
    case TIER_AVX512: return pq_cosine_avx512(offs, len, k, lut, amag, bmag);
    case TIER_AVX2: return pq_cosine_avx2(offs, len, k, lut, amag, bmag);
    default: return jvo_pq_decoded_cosine(offs, 0, len, k, lut, amag, bmag);
    }
}

/* PQDecoder / FusedPQDecoder similarityTo on the native kernels */
float jvs_adc_score(int vsf, int M, int k, const float *lut, const float *amag, float bmag, const uint8_t *code)
{
    if (vsf == JVO_COSINE) return jvo_score_from_raw(vsf, jvs_pq_decoded_cosine(code, M, k, lut, amag, bmag));
    return jvo_score_from_raw(vsf, jvs_assemble_and_sum(lut, k, code, M));
}
