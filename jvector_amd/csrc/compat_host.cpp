// compat_host.cpp — the reference's per-pair native SPI (include/jvector_simd_compat.h) implemented on the
// host in the scalar DefaultVectorUtilSupport order.  Boundary completeness only; see the header.
// Citations: B/ = /root/reference/jvector-base/src/main/java/io/github/jbellis/jvector/
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "../../include/jvector_simd_compat.h"

namespace {

// B/vector/DefaultVectorUtilSupport.java:38-105 — full-vector dot: first n%8 one by one, then 8-blocks
inline float dot_full(const float *a, const float *b, size_t n)
{
    float res = 0.0f;
    size_t i = 0;
    for (; i < n % 8; ++i) res += b[i] * a[i];
    if (n < 8) return res;
    for (; i + 7 < n; i += 8) {
        float t = b[i] * a[i] + b[i + 1] * a[i + 1];
        for (int j = 2; j < 8; ++j) t = t + b[i + j] * a[i + j];
        res += t;
    }
    return res;
}

// :158-193 — 8-blocks of squared differences, sequential tail
inline float l2_full(const float *a, const float *b, size_t n)
{
    float sq = 0.0f;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        float d0 = a[i] - b[i], d1 = a[i + 1] - b[i + 1];
        float t = d0 * d0 + d1 * d1;
        for (int j = 2; j < 8; ++j) {
            float d = a[i + j] - b[i + j];
            t = t + d * d;
        }
        sq += t;
    }
    for (; i < n; ++i) {
        float d = a[i] - b[i];
        sq += d * d;
    }
    return sq;
}

inline float dot_seq(const float *a, const float *b, size_t n)  // :107-119
{
    float s = 0.0f;
    for (size_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

inline float l2_seq(const float *a, const float *b, size_t n)  // :195-208
{
    float s = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        float d = a[i] - b[i];
        s += d * d;
    }
    return s;
}

// NVQ helpers — DefaultVectorUtilSupport.java:441-474 (logistic / logit "NQT" approximations, Math.fma)
inline int java_round(float x)   // Math.round(float): floor(x + 1/2), NaN -> 0, saturating
{
    if (x != x) return 0;
    const double r = std::floor((double)x + 0.5);
    if (r <= -2147483648.0) return INT32_MIN;
    if (r >= 2147483647.0) return INT32_MAX;
    return (int)r;
}
inline float bits_to_float(int32_t b) { float f; std::memcpy(&f, &b, 4); return f; }
inline int32_t float_to_bits(float f)   // Float.floatToIntBits: NaN canonical
{
    if (f != f) return 0x7fc00000;
    int32_t b;
    std::memcpy(&b, &f, 4);
    return b;
}

inline float logistic_nqt(float value, float alpha, float x0)
{
    float temp = std::fmaf(value, alpha, -alpha * x0);
    int p = java_round(temp + 0.5f);
    int32_t m = float_to_bits(std::fmaf(temp - (float)p, 0.5f, 1.0f));
    temp = bits_to_float((int32_t)((uint32_t)m + ((uint32_t)p << 23)));
    return temp / (temp + 1.0f);
}
inline float logit_nqt(float value, float inverseAlpha, float x0)
{
    float z = value / (1.0f - value);
    int32_t temp = float_to_bits(z);
    int32_t e = temp & 0x7f800000;
    float p = (float)((e >> 23) - 128);
    float m = bits_to_float((temp & 0x007fffff) + 0x3f800000);
    return std::fmaf(m + p, inverseAlpha, x0);
}
inline float scaled_logistic(float v, float growth, float mid, float scale, float bias)
{
    return (logistic_nqt(v, growth, mid) - bias) * (1.0f / scale);
}
inline float scaled_logit_nqt(float v, float invGrowth, float mid, float scale, float bias)
{
    return logit_nqt(std::fmaf(v, scale, bias), invGrowth, mid);
}
struct NvqParams { float sgr, smid, inv, bias, scale; };
inline NvqParams nvq_params(float alpha, float x0, float minV, float maxV, float levels)
{
    NvqParams p;
    float delta = maxV - minV;
    p.sgr = alpha / delta;
    p.smid = x0 * delta;
    p.inv = 1.0f / p.sgr;
    p.bias = logistic_nqt(minV, p.sgr, p.smid);
    p.scale = (logistic_nqt(maxV, p.sgr, p.smid) - p.bias) / levels;
    return p;
}

}  // namespace

extern "C" {

float dot_product_f32(const float *a, size_t ao, const float *b, size_t bo, size_t n)
{
    // NativeVectorUtilSupport routes both the full (:198-202) and the offset (:205-209) form here; the
    // full form has offsets 0 and the vector's whole length — indistinguishable at this ABI, so the
    // sequential offset order is used whenever an offset is non-zero and the 8-block order otherwise.
    return (ao == 0 && bo == 0) ? dot_full(a, b, n) : dot_seq(a + ao, b + bo, n);
}
float euclidean_f32(const float *a, size_t ao, const float *b, size_t bo, size_t n)
{
    return (ao == 0 && bo == 0) ? l2_full(a, b, n) : l2_seq(a + ao, b + bo, n);
}
float cosine_f32(const float *a, size_t ao, const float *b, size_t bo, size_t n)
{
    a += ao;
    b += bo;
    float sum = 0.0f, n1 = 0.0f, n2 = 0.0f;  // DefaultVectorUtilSupport.java:121-156
    for (size_t i = 0; i < n; ++i) {
        sum += a[i] * b[i];
        n1 += a[i] * a[i];
        n2 += b[i] * b[i];
    }
    float prod = n1 * n2;
    return (float)((double)sum / std::sqrt((double)prod));
}

void add_in_place_f32(float *v1, const float *v2, size_t n) { for (size_t i = 0; i < n; ++i) v1[i] = v1[i] + v2[i]; }
void add_scalar_in_place_f32(float *v1, float x, size_t n) { for (size_t i = 0; i < n; ++i) v1[i] = v1[i] + x; }
void sub_in_place_f32(float *v1, const float *v2, size_t n) { for (size_t i = 0; i < n; ++i) v1[i] = v1[i] - v2[i]; }
void sub_scalar_in_place_f32(float *v1, float x, size_t n) { for (size_t i = 0; i < n; ++i) v1[i] = v1[i] - x; }
float max_f32(const float *v, size_t n)
{
    float m = -3.4028234663852886e+38f;  // -Float.MAX_VALUE, DefaultVectorUtilSupport.java:368-374
    for (size_t i = 0; i < n; ++i) m = (v[i] != v[i]) ? v[i] : (v[i] > m ? v[i] : m);
    return m;
}
void min_in_place_f32(float *v1, const float *v2, size_t n)
{
    for (size_t i = 0; i < n; ++i) v1[i] = (v2[i] != v2[i] || v2[i] < v1[i]) ? v2[i] : v1[i];
}

float assemble_and_sum_f32(const float *data, int dataBase, const unsigned char *offs, int off, size_t len)
{
    float sum = 0.0f;  // DefaultVectorUtilSupport.java:302-309
    for (size_t i = 0; i < len; ++i) sum += data[(size_t)dataBase * i + offs[i + off]];
    return sum;
}

float assemble_and_sum_pq_f32(const float *data, size_t M, const unsigned char *o1, int off1, const unsigned char *o2,
                              int off2, int k)
{
    const int blockSize = k * (k + 1) / 2;  // DefaultVectorUtilSupport.java:311-339
    float res = 0.0f;
    for (size_t i = 0; i < M; ++i) {
        int c1 = o1[i + off1], c2 = o2[i + off2];
        int r = c1 < c2 ? c1 : c2, c = c1 < c2 ? c2 : c1;
        res += data[i * blockSize + (r * k - (r * (r - 1) / 2)) + (c - r)];
    }
    return res;
}

float pq_decoded_cosine_similarity_f32(const unsigned char *offs, int off, size_t len, int k, const float *partialSums,
                                       const float *aMagnitude, float bMagnitude)
{
    float sum = 0.0f, aMag = 0.0f;  // VectorUtilSupport.java:152-165
    for (size_t m = 0; m < len; ++m) {
        size_t idx = m * (size_t)k + offs[m + off];
        sum += partialSums[idx];
        aMag += aMagnitude[idx];
    }
    float prod = aMag * bMagnitude;
    return (float)((double)sum / std::sqrt((double)prod));
}

void calculate_partial_sums_dot_f32(const float *cb, int cbIndex, size_t size, int k, const float *q, int qoff,
                                    float *out)
{
    for (int i = 0; i < k; ++i) out[cbIndex * k + i] = dot_seq(cb + (size_t)i * size, q + qoff, size);
}
void calculate_partial_sums_euclidean_f32(const float *cb, int cbIndex, size_t size, int k, const float *q, int qoff,
                                          float *out)
{
    for (int i = 0; i < k; ++i) out[cbIndex * k + i] = l2_seq(cb + (size_t)i * size, q + qoff, size);
}
void calculate_partial_sums_self_magnitude_f32(const float *cb, int cbIndex, size_t size, int k, float *out)
{
    for (int i = 0; i < k; ++i) out[cbIndex * k + i] = dot_seq(cb + (size_t)i * size, cb + (size_t)i * size, size);
}

// ---- NVQ (out of GPU scope; scalar DefaultVectorUtilSupport.java:376-548 semantics, unshuffled) ----
void nvq_quantize_8bit(const float *v, size_t n, float alpha, float x0, float minV, float maxV, unsigned char *dst)
{
    NvqParams p = nvq_params(alpha, x0, minV, maxV, 255.0f);
    for (size_t d = 0; d < n; ++d) dst[d] = (unsigned char)java_round(scaled_logistic(v[d], p.sgr, p.smid, p.scale, p.bias));
}
float nvq_loss(const float *v, size_t n, float alpha, float x0, float minV, float maxV, int nBits)
{
    NvqParams p = nvq_params(alpha, x0, minV, maxV, (float)((1 << nBits) - 1));
    float sq = 0.0f;
    for (size_t d = 0; d < n; ++d) {
        float r = scaled_logistic(v[d], p.sgr, p.smid, p.scale, p.bias);
        r = (float)java_round(r);
        r = scaled_logit_nqt(r, p.inv, p.smid, p.scale, p.bias);
        float diff = v[d] - r;
        sq = std::fmaf(diff, diff, sq);
    }
    return sq;
}
float nvq_uniform_loss(const float *v, size_t n, float minV, float maxV, int nBits)
{
    float constant = (float)((1 << nBits) - 1), sq = 0.0f;
    for (size_t d = 0; d < n; ++d) {
        float r = (v[d] - minV) / (maxV - minV);
        r = (float)java_round(constant * r) / constant;
        r = r * (maxV - minV) + minV;
        float diff = v[d] - r;
        sq = std::fmaf(diff, diff, sq);
    }
    return sq;
}
float nvq_square_l2_distance_8bit(const float *v, const unsigned char *qz, size_t n, float alpha, float x0, float minV,
                                  float maxV)
{
    NvqParams p = nvq_params(alpha, x0, minV, maxV, 255.0f);
    float sq = 0.0f;
    for (size_t d = 0; d < n; ++d) {
        float val = scaled_logit_nqt((float)qz[d], p.inv, p.smid, p.scale, p.bias);
        float t = val - v[d];
        sq = std::fmaf(t, t, sq);
    }
    return sq;
}
float nvq_dot_product_8bit(const float *v, const unsigned char *qz, size_t n, float alpha, float x0, float minV,
                           float maxV)
{
    NvqParams p = nvq_params(alpha, x0, minV, maxV, 255.0f);
    float dp = 0.0f;
    for (size_t d = 0; d < n; ++d) dp = std::fmaf(v[d], scaled_logit_nqt((float)qz[d], p.inv, p.smid, p.scale, p.bias), dp);
    return dp;
}
int64_t nvq_cosine_8bit_packed(const float *v, const unsigned char *qz, size_t n, float alpha, float x0, float minV,
                               float maxV, const float *centroid)
{
    NvqParams p = nvq_params(alpha, x0, minV, maxV, 255.0f);
    float sum = 0.0f, norm = 0.0f;
    for (size_t d = 0; d < n; ++d) {
        float e = scaled_logit_nqt((float)qz[d], p.inv, p.smid, p.scale, p.bias) + centroid[d];
        sum = std::fmaf(v[d], e, sum);
        norm = std::fmaf(e, e, norm);
    }
    // two floats packed lo/hi, unpacked at NativeVectorUtilSupport.java:289-297
    return (int64_t)(((uint64_t)(uint32_t)float_to_bits(norm) << 32) | (uint64_t)(uint32_t)float_to_bits(sum));
}
void nvq_shuffle_query_in_place_8bit(float *, size_t) {}  // DefaultVectorUtilSupport.java:439 (no-op, unshuffled layout)

const char *jvector_simd_get_active_isa(void) { return "gfx950-host"; }
const char *jvector_simd_get_max_isa_env(void)
{
    static const char *v = std::getenv("JVECTOR_MAX_ISA");
    return v;
}

}  // extern "C"
