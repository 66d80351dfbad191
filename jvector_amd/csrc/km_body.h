// km_body.h — PQ training on the device: KMeansPlusPlusClusterer's unweighted path and
// ProductQuantization.compute / refine (B/quantization/KMeansPlusPlusClusterer.java:131-150,171-272,330-372,437-446;
// B/quantization/ProductQuantization.java:109-139,194-221,487-495).
//
// The reference's result depends on the ORDER of its float accumulations (centroidNums is updated point by point,
// with subtractions when a point changes cluster), so a parallel reduction would only be "statistically" equal.
// Instead every accumulator is owned by exactly one thread that replays ITS OWN history in point order:
//   km_replay     thread (m, c) walks the points 0..n-1 and applies exactly the addInPlace / subInPlace calls
//                 updateAssignedPointsUnweighted would apply to centroidNums[c] — same operands, same order, hence the
//                 same bits; all M x k threads run concurrently, wavefront lanes = clusters of one subspace so the
//                 assignment bytes they scan are broadcast loads.
//   km_assign     thread (point, m): getNearestCluster (strict <, first minimum).
//   km_centroids  thread (m, c): scale(centroidNums, 1.0f / denom).
//   km_pp_init    one wavefront per subspace: k-means++ seeding; the distance refresh is lane-parallel, the
//                 sequential prefix scan that picks the next centroid (sum, then r -= d[j] until r < 1e-6) is one chain in
//                 the reference's order that every lane runs on values broadcast 64 at a time (gs_gather64).
// ThreadLocalRandom cannot be reproduced; a seeded splitmix64 stream per subspace replaces it (the CPU
// checker under tests/ makes the same substitution), so both agree bit for bit for a given seed.
// Per-thread bodies need KM_FN; km_pp_init additionally needs the wave API of gs_body.h (GS_FN, gs_lane, gs_barrier,
// gs_gather64, gs_fence).  The anisotropic k-means variants are not built.
#pragma once

#include <cstdint>

namespace jv {

struct KmParams {
    const float *X;            // [n][D] centred training vectors
    float *C;                  // codebooks, concatenated [m][k][size_m] (in/out)
    const int64_t *cb_offsets;
    const int32_t *sizes, *offsets;
    uint8_t *assign_old, *assign_new;  // [n][M]
    float *nums;               // centroidNums, layout of C
    int32_t *denoms;           // [M][k]
    int32_t *active;           // [M] 1 while the subspace's clusterer is still iterating
    int32_t *changed;          // [M]
    uint64_t *rng;             // [M] splitmix64 state
    float *dist;               // [M][n] k-means++ distances
    float *cnorm;              // [M][k] centroid norms (anisotropic rounds)
    const float *pcm;          // [M] parallel cost multiplier per subspace (anisotropic rounds), computed on the host
    int64_t n;
    int32_t D, M, k;
};

constexpr float KM_FLT_MAX = 3.4028234663852886e38f;

KM_FN uint64_t km_rng_next(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
KM_FN int64_t km_rng_int(uint64_t *s, int64_t bound) { return (int64_t)((km_rng_next(s) >> 33) % (uint64_t)bound); }
KM_FN float km_rng_float(uint64_t *s) { return (float)(km_rng_next(s) >> 40) * (1.0f / 16777216.0f); }
inline uint64_t km_stream(uint64_t seed, int m) { return seed * 0x9E3779B97F4A7C15ull + (uint64_t)m * 0xD1B54A32D192ED03ull + 1; }

// centroidOf: thread d sums dimension d over the points in order, then scales by 1.0f / n
KM_FN void km_centroid_dim(const float *X, int64_t n, int D, int64_t d, float *out)
{
    float s = 0.0f;
    for (int64_t i = 0; i < n; ++i) s = s + X[i * D + d];
    out[d] = s * (1.0f / (float)n);
}

// VectorUtil.sub(v, centroid): thread t over n*D
KM_FN void km_center(const float *X, const float *centroid, int D, int64_t t, float *Xc)
{
    Xc[t] = centroid ? X[t] - centroid[t % D] : X[t];
}

// getNearestCluster :330-343 — thread t = i*M + m
KM_FN void km_assign(const KmParams &p, int64_t t)
{
    const int m = (int)(t % p.M);
    if (!p.active[m]) {  // a converged subspace keeps its assignments current in both buffers (they alternate every round)
        p.assign_new[t] = p.assign_old[t];
        return;
    }
    const int64_t i = t / p.M;
    const int len = p.sizes[m];
    const float *x = p.X + i * p.D + p.offsets[m];
    const float *C = p.C + p.cb_offsets[m];
    float minDistance = KM_FLT_MAX;
    int nearest = 0;
    for (int c = 0; c < p.k; ++c) {
        const float *cc = C + (int64_t)c * len;
        float d = 0.0f;
        for (int j = 0; j < len; ++j) {  // squareL2Distance offsets form, a = point, b = centroid
            const float diff = x[j] - cc[j];
            d += diff * diff;
        }
        if (d < minDistance) {
            minDistance = d;
            nearest = c;
        }
    }
    p.assign_new[t] = (uint8_t)nearest;
}

// initializeAssignedPoints (first_pass) / updateAssignedPointsUnweighted: thread t = m*k + c replays cluster c's history.
// LEN > 0: the sub-vector length is the compile-time LEN (accumulator in registers); LEN == 0: any length <= 64.
template <int LEN>
KM_FN void km_replay_len(const KmParams &p, int first_pass, int64_t t, int m, int c)
{
    const int len = LEN > 0 ? LEN : p.sizes[m], off = p.offsets[m];
    float *num = p.nums + p.cb_offsets[m] + (int64_t)c * len;
    int32_t denom = first_pass ? 0 : p.denoms[t];
    float acc[LEN > 0 ? LEN : 64];  // sub-vector length <= 64 (checked by the host)
    for (int j = 0; j < len; ++j) acc[j] = first_pass ? 0.0f : num[j];
    int32_t changed = 0;
    for (int64_t i = 0; i < p.n; ++i) {
        const int a = p.assign_new[i * p.M + m];
        const int o = first_pass ? -1 : (int)p.assign_old[i * p.M + m];
        if (a == o) continue;
        ++changed;
        const float *x = p.X + i * p.D + off;
        if (o == c) {
            --denom;
            for (int j = 0; j < len; ++j) acc[j] = acc[j] - x[j];
        }
        if (a == c) {
            ++denom;
            for (int j = 0; j < len; ++j) acc[j] = acc[j] + x[j];
        }
    }
    for (int j = 0; j < len; ++j) num[j] = acc[j];
    p.denoms[t] = denom;
    if (c == 0) p.changed[m] = changed;
}

KM_FN void km_replay(const KmParams &p, int first_pass, int64_t t)
{
    const int m = (int)(t / p.k), c = (int)(t % p.k);
    if (m >= p.M || !p.active[m]) return;
    if (p.sizes[m] == 8) km_replay_len<8>(p, first_pass, t, m, c);  // every BASELINE config: D / M = 8
    else km_replay_len<0>(p, first_pass, t, m, c);
}

// updateCentroidsUnweighted, non-empty clusters: thread t = m*k + c
KM_FN void km_centroids(const KmParams &p, int64_t t)
{
    const int m = (int)(t / p.k), c = (int)(t % p.k);
    if (m >= p.M || !p.active[m]) return;
    const int32_t denom = p.denoms[t];
    if (denom == 0) return;  // km_fill_empties
    const int len = p.sizes[m];
    const float inv = 1.0f / (float)denom;
    const float *num = p.nums + p.cb_offsets[m] + (int64_t)c * len;
    float *cc = p.C + p.cb_offsets[m] + (int64_t)c * len;
    for (int j = 0; j < len; ++j) cc[j] = num[j] * inv;
}

// updateCentroidsUnweighted, empty clusters (initializeCentroidToRandomPoint) in cluster order: thread m
KM_FN void km_fill_empties(const KmParams &p, int64_t m)
{
    if (!p.active[m]) return;
    uint64_t s = p.rng[m];
    const int len = p.sizes[m];
    for (int c = 0; c < p.k; ++c) {
        if (p.denoms[m * p.k + c] != 0) continue;
        const float *x = p.X + km_rng_int(&s, p.n) * p.D + p.offsets[m];
        float *cc = p.C + p.cb_offsets[m] + (int64_t)c * len;
        for (int j = 0; j < len; ++j) cc[j] = x[j];
    }
    p.rng[m] = s;
}

// cluster() :131-150: stop iterating a subspace once <= 1 % of its points moved: thread m
KM_FN void km_finish_round(const KmParams &p, int64_t m)
{
    if (p.active[m] && (double)p.changed[m] <= 0.01 * (double)p.n) p.active[m] = 0;
}

// ---- anisotropic rounds (clusterOnceAnisotropic :163-166), sub-vectors of at most KM_ANISO_MAX_LEN dimensions ----
constexpr int KM_ANISO_MAX_LEN = 16;

// VectorUtil.dotProduct(a, b) full-vector form
KM_FN float km_full_dot(const float *a, const float *b, int n)
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

// updateCentroidsAnisotropic :380-432 for cluster c of subspace m (thread t = m*k + c): mean and normalised outer-product
// sum over the cluster's points in point order, (1 - ocm)/|L| scaling, + ocm on the diagonal, Matrix.invert
// (B/vector/Matrix.java:70-118: Gauss-Jordan, partial pivoting), inverse x mean.  Writes the point count to denoms[t]
// (0 = empty: km_fill_empties re-seeds it).
KM_FN void km_centroids_aniso(const KmParams &p, int64_t t)
{
    const int m = (int)(t / p.k), c = (int)(t % p.k);
    if (m >= p.M || !p.active[m]) return;
    const int len = p.sizes[m], off = p.offsets[m];
    const float pcm = p.pcm[m], ocm = 1.0f / pcm;
    float mean[KM_ANISO_MAX_LEN], outer[KM_ANISO_MAX_LEN * KM_ANISO_MAX_LEN], aug[KM_ANISO_MAX_LEN * 2 * KM_ANISO_MAX_LEN];
    for (int d = 0; d < len; ++d) mean[d] = 0.0f;
    for (int d = 0; d < len * len; ++d) outer[d] = 0.0f;
    int32_t cnt = 0;
    for (int64_t i = 0; i < p.n; ++i) {
        if (p.assign_old[i * p.M + m] != c) continue;  // the assignments of the previous round
        const float *x = p.X + i * p.D + off;
        ++cnt;
        for (int d = 0; d < len; ++d) mean[d] = mean[d] + x[d];
        const float denom = km_full_dot(x, x, len);
        if (denom > 0) {
            const float invd = 1.0f / denom;
            for (int r = 0; r < len; ++r)
                for (int d = 0; d < len; ++d) outer[r * len + d] = outer[r * len + d] + (x[d] * x[r]) * invd;
        }
    }
    p.denoms[t] = cnt;
    if (cnt == 0) return;
    const float sc = (1 - ocm) / (float)cnt, invc = 1.0f / (float)cnt;
    for (int d = 0; d < len * len; ++d) outer[d] = outer[d] * sc;
    for (int d = 0; d < len; ++d) mean[d] = mean[d] * invc;
    for (int d = 0; d < len; ++d) outer[d * len + d] = outer[d * len + d] + ocm;
    const int W = 2 * len;
    for (int i = 0; i < len; ++i)
        for (int j = 0; j < len; ++j) {
            aug[i * W + j] = outer[i * len + j];
            aug[i * W + j + len] = (i == j) ? 1.0f : 0.0f;
        }
    for (int i = 0; i < len; ++i) {
        int maxRow = i;
        for (int r = i + 1; r < len; ++r) {
            const float a = aug[r * W + i], b = aug[maxRow * W + i];
            if ((a < 0 ? -a : a) > (b < 0 ? -b : b)) maxRow = r;
        }
        if (maxRow != i)
            for (int j = 0; j < W; ++j) {
                const float tmp = aug[i * W + j];
                aug[i * W + j] = aug[maxRow * W + j];
                aug[maxRow * W + j] = tmp;
            }
        const float s = 1 / aug[i * W + i];
        for (int j = 0; j < W; ++j) aug[i * W + j] = aug[i * W + j] * s;
        for (int r = 0; r < len; ++r) {
            if (r == i) continue;
            const float factor = aug[r * W + i];
            for (int j = 0; j < W; ++j) aug[r * W + j] = aug[r * W + j] + (-factor * aug[i * W + j]);
        }
    }
    float *cc = p.C + p.cb_offsets[m] + (int64_t)c * len;
    for (int r = 0; r < len; ++r) {  // Matrix.multiply: dotProduct(inverse row, mean), full-vector form
        float row[KM_ANISO_MAX_LEN];
        for (int j = 0; j < len; ++j) row[j] = aug[r * W + j + len];
        cc[r] = km_full_dot(row, mean, len);
    }
}

// cNormSquared of updateAssignedPointsAnisotropic :279-285 — thread t = m*k + c
KM_FN void km_cnorm(const KmParams &p, int64_t t)
{
    const int m = (int)(t / p.k), c = (int)(t % p.k);
    if (m >= p.M || !p.active[m]) return;
    const int len = p.sizes[m];
    const float *cc = p.C + p.cb_offsets[m] + (int64_t)c * len;
    float s = 0.0f;
    for (int j = 0; j < len; ++j) s += cc[j] * cc[j];
    p.cnorm[t] = s;
}

// updateAssignedPointsAnisotropic :274-306 + weightedDistance :311-320 — thread t = i*M + m
KM_FN void km_assign_aniso(const KmParams &p, int64_t t)
{
    const int m = (int)(t % p.M);
    if (!p.active[m]) {
        p.assign_new[t] = p.assign_old[t];
        return;
    }
    const int64_t i = t / p.M;
    const int len = p.sizes[m];
    const float *x = p.X + i * p.D + p.offsets[m];
    const float *C = p.C + p.cb_offsets[m];
    const float pcm = p.pcm[m];
    const float xNorm = km_full_dot(x, x, len);
    int index = p.assign_old[t];
    float minDist = KM_FLT_MAX;
    for (int j = 0; j < p.k; ++j) {
        const float *cc = C + (int64_t)j * len;
        float cDotX = 0.0f;
        for (int d = 0; d < len; ++d) cDotX += cc[d] * x[d];
        const float pes = cDotX - xNorm;
        const float two = 2 * cDotX;
        const float rsn = p.cnorm[m * p.k + j] - two + xNorm;
        const float parErr = pes * pes;
        const float perp = rsn - parErr;
        const float dist = pcm * parErr + perp;
        if (dist < minDist) {
            minDist = dist;
            index = j;
        }
    }
    p.assign_new[t] = (uint8_t)index;
}

// changedCount of an anisotropic round — thread m
KM_FN void km_count_changed(const KmParams &p, int64_t m)
{
    if (!p.active[m]) return;
    int32_t changed = 0;
    for (int64_t i = 0; i < p.n; ++i) changed += p.assign_new[i * p.M + m] != p.assign_old[i * p.M + m];
    p.changed[m] = changed;
}

// every subspace iterates again when the anisotropic phase starts — thread m
KM_FN void km_reactivate(const KmParams &p, int64_t m) { p.active[m] = 1; }

#ifdef GS_FN
// VectorUtil.squareL2Distance(a, b) full-vector form: blocks of eight summed left to right, then the tail
GS_FN float km_full_l2(const float *a, const float *b, int n)
{
    float sq = 0.0f;
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        const float d0 = a[i] - b[i], d1 = a[i + 1] - b[i + 1], d2 = a[i + 2] - b[i + 2], d3 = a[i + 3] - b[i + 3];
        const float d4 = a[i + 4] - b[i + 4], d5 = a[i + 5] - b[i + 5], d6 = a[i + 6] - b[i + 6], d7 = a[i + 7] - b[i + 7];
        float t = d0 * d0 + d1 * d1;
        t = t + d2 * d2;
        t = t + d3 * d3;
        t = t + d4 * d4;
        t = t + d5 * d5;
        t = t + d6 * d6;
        t = t + d7 * d7;
        sq += t;
    }
    for (; i < n; ++i) {
        const float d = a[i] - b[i];
        sq += d * d;
    }
    return sq;
}

// chooseInitialCentroids :171-226 for subspace m, one wavefront
GS_FN void km_pp_init(const KmParams &p, int m)
{
    const int lane = gs_lane();
    const int len = p.sizes[m], off = p.offsets[m];
    float *dist = p.dist + (int64_t)m * p.n;
    float *C = p.C + p.cb_offsets[m];
    for (int64_t j = lane; j < p.n; j += 64) dist[j] = KM_FLT_MAX;
    // The sequential part (sum of the distances, then r -= d[j] until r < 1e-6, KMeansPlusPlusClusterer.java:139-156) is
    // order dependent, so it stays one chain — but every lane runs the same chain on values broadcast 64 at a time
    // (gs_gather64: one coalesced load per 64 distances instead of 64 dependent single-lane loads), which keeps all state
    // wave-uniform: no lane 0 special case, no shuffle of the result.
    uint64_t s = p.rng[m];
    int64_t sel = km_rng_int(&s, p.n);
    for (int c = 0; c < p.k; ++c) {
        gs_fence();
        gs_barrier();
        if (c > 0) {
            float total = 0.0f;
            for (int64_t base = 0; base < p.n; base += 64) {
                float blk[64];
                gs_gather64(base + lane < p.n ? dist[base + lane] : 0.0f, blk);
                const int cnt = (int)(p.n - base < 64 ? p.n - base : 64);
#pragma unroll
                for (int i = 0; i < 64; ++i)
                    if (i < cnt) total += blk[i];
            }
            float r = km_rng_float(&s) * total;
            sel = -1;
            for (int64_t base = 0; base < p.n && sel < 0; base += 64) {
                float blk[64];
                gs_gather64(base + lane < p.n ? dist[base + lane] : 0.0f, blk);
                const int cnt = (int)(p.n - base < 64 ? p.n - base : 64);
#pragma unroll
                for (int i = 0; i < 64; ++i) {
                    if (sel < 0 && i < cnt) {
                        r -= blk[i];
                        if ((double)r < 1e-6) sel = base + i;
                    }
                }
            }
            if (sel == -1) sel = km_rng_int(&s, p.n);
        }
        const float *x = p.X + sel * p.D + off;
        float *cc = C + (int64_t)c * len;
        for (int j = lane; j < len; j += 64) cc[j] = x[j];
        gs_fence();
        gs_barrier();
        for (int64_t j = lane; j < p.n; j += 64) {
            const float d = km_full_l2(p.X + j * p.D + off, cc, len);
            if (d < dist[j]) dist[j] = d;
        }
    }
    if (lane == 0) p.rng[m] = s;
    gs_fence();
    gs_barrier();
}
#endif

}  // namespace jv
