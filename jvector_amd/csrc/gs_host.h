// gs_host.h — host-side preparation for the device-resident graph search (pure C++, no HIP): the open-addressing
// node -> row maps of the sparse upper levels that gs_body.h's gs_level_row probes, and the sizing rules.
#pragma once

#include <cstdint>
#include <vector>

namespace jv {

struct GsLevelMap {
    std::vector<int32_t> keys, vals;  // keys: node id or -1
    uint32_t mask = 0;
    int32_t shift = 32;
};

// nodes: the level's node ids (any order, unique).  Load factor <= 1/2, same hash as the device probe.
inline GsLevelMap gs_build_level_map(const int32_t *nodes, int count)
{
    GsLevelMap m;
    int log2 = 4;
    while ((1ll << log2) < 2ll * count) ++log2;
    const uint32_t size = 1u << log2;
    m.mask = size - 1;
    m.shift = 32 - log2;
    m.keys.assign(size, -1);
    m.vals.assign(size, -1);
    for (int i = 0; i < count; ++i) {
        uint32_t h = ((uint32_t)nodes[i] * 0x9E3779B1u) >> m.shift;
        while (m.keys[h] != -1) h = (h + 1) & m.mask;
        m.keys[h] = nodes[i];
        m.vals[h] = i;
    }
    return m;
}

// ---- UBR (gs_body.h "UBR"): the 8-bit upper-bound table of ONE query's ADC entries, the way ubr_table_kernel (k_gsearch_ubr.hip)
// builds it for a batch — this restatement serves the lane emulator, the mock device and the GPU test that compares the kernel's
// bytes with it.  Dot product / cosine entries (calculatePartialSums' chain, mul and add separate); per subspace lo_m / hi_m = the
// extreme entries; ONE scale S = max_m (hi_m - lo_m) / 255; entry -> bucket b with lo_m + S (b + 1) > entry (see `bucket` below).
// tab: M x 64 dwords in the register layout (register k of lane s at ((k / 4) * 64 + s) * 4 + k % 4; for step r < M / 2 register 2r
// holds {t[r][s], t[r][s + 64], t[r + M/2][s], t[r + M/2][s + 64]} and 2r + 1 the same for codes s + 128 / s + 192);
// meta4 = {sum_m lo_m + slack, S, usable (1 / 0), 0}.  Compile with -ffp-contract=off.
// l2 (round 6): the EUCLIDEAN form — entries are squared distances (sequential t = c - q; ent += t * t, squareDistance's offset form),
// the score 1 / (1 + d) FALLS with the sum, so the table holds LOWER bucket edges: b = floor((e - lo) / S - 2^-10) (the three roundings of the
// quotient stay below 5e-5: lo + S b <= e in real arithmetic), a row's bound is base + S sum b_m with base = sum lo - slack, and a
// neighbour is dropped when 1 / (1 + that) < T.
inline void gs_ubr_build_ref(const float *codebooks /* [M][256][8] */, const float *cq /* [8 M] centred query */, int M, uint32_t *tab, float *meta4,
                             bool l2 = false)
{
    std::vector<float> e((size_t)M * 256), lo((size_t)M), hi((size_t)M);
    bool ok = true;
    float range = 0.0f;
    for (int m = 0; m < M; ++m) {
        float mn = 0.0f, mx = 0.0f;
        for (int c = 0; c < 256; ++c) {
            const float *row = codebooks + ((size_t)m * 256 + (size_t)c) * 8;
            const float *q = cq + (size_t)m * 8;
            float ent = 0.0f;
            for (int j = 0; j < 8; ++j) {
                if (l2) {
                    const float t = row[j] - q[j];
                    const float pr = t * t;
                    ent += pr;
                } else {
                    const float pr = row[j] * q[j];
                    ent += pr;
                }
            }
            e[(size_t)m * 256 + c] = ent;
            if (!(ent - ent == 0.0f)) ok = false;
            if (c == 0 || ent < mn) mn = ent;
            if (c == 0 || ent > mx) mx = ent;
        }
        lo[m] = mn;
        hi[m] = mx;
        const float r = mx - mn;
        if (r > range) range = r;
    }
    float S = range / 255.0f;
    if (!(S > 1e-30f)) S = 1e-30f;
    const float inv = 1.0f / S;
    float sum_lo = 0.0f, sum_abs = 0.0f, max_abs = 0.0f;
    for (int m = 0; m < M; ++m) {
        sum_lo += lo[m];
        const float amn = lo[m] < 0.0f ? -lo[m] : lo[m], amx = hi[m] < 0.0f ? -hi[m] : hi[m];
        const float a = amn > amx ? amn : amx;
        sum_abs += a + 256.0f * S;
        if (a > max_abs) max_abs = a;
    }
    // usable only if a bucket is not lost in the rounding of an edge: then lo + 256 S >= every entry of the subspace in f32 as well
    ok = ok && (sum_abs - sum_abs == 0.0f) && S * 1e6f >= max_abs;
    auto bucket = [&](int m, int c) -> uint32_t {
        const float ent = e[(size_t)m * 256 + c];
        // b = floor(t + 2^-10), t = (ent - lo) / S as f32 evaluates it: three roundings put t within 256 x 3.1 x 2^-24 < 5e-5 of the real
        // quotient (and the addition within 8e-6 more), so b + 1 exceeds the real quotient and lo + S (b + 1) > ent in real arithmetic;
        // a clamped b = 255 bounds too: 256 S >= (256 / 255)(1 - 2^-24) x the widest range > hi - lo.  (The traversal's f32 evaluation
        // of sum lo + S sum (b + 1) is covered by the slack in meta4[0], as before.)  One entry in a thousand lands a bucket higher
        // than the tightest choice.
        int b = l2 ? (int)((ent - lo[m]) * inv - 0x1p-10f) : (int)((ent - lo[m]) * inv + 0x1p-10f);
        b = b < 0 ? 0 : (b > 255 ? 255 : b);
        return (uint32_t)b;
    };
    const int H = M / 2;
    for (int r = 0; r < H; ++r)
        for (int s = 0; s < 64; ++s)
            for (int u = 0; u < 2; ++u) {   // u = 0: codes s, s + 64; u = 1: codes s + 128, s + 192
                const int k = 2 * r + u, c0 = s + 128 * u;
                uint32_t v = 0;
                if (ok) v = bucket(r, c0) | (bucket(r, c0 + 64) << 8) | (bucket(r + H, c0) << 16) | (bucket(r + H, c0 + 64) << 24);
                tab[((size_t)(k / 4) * 64 + (size_t)s) * 4 + (size_t)(k % 4)] = v;
            }
    meta4[0] = l2 ? sum_lo - 4e-5f * sum_abs : sum_lo + 4e-5f * sum_abs;
    meta4[1] = S;
    meta4[2] = ok ? 1.0f : 0.0f;
    meta4[3] = 0.0f;
}

// visited-table size for a search that keeps rerankK results: a search marks ~18 x rerankK nodes at maxDegree 32
// (measured, 10M x 768); 64 x leaves the table under half full for all but pathological queries (those overflow
// and are re-run on the host).
inline int gs_vcap_log2(int rerankK)
{
    int log2 = 12;
    while ((1ll << log2) < 64ll * rerankK && log2 < 22) ++log2;
    return log2;
}

}  // namespace jv
