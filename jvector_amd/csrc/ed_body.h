// ed_body.h — body of exact_dense_kernel (k_exact_dense.hip): the MFMA tile form of full-resolution scoring
// (dense Q x N form: brute force / ground truth; north_star: "MFMA only for the batched
// query x candidates GEMM form of full-resolution rerank").  Shared source: compiled for gfx950 through gs_wave_hip.h
// and for the CPU lane emulator (tests/emu/ed_emu.cpp, tests/mock/mock_kernels.cpp).
//
// NOT the bit-exact path (that is exact_scan_kernel in k_exact.hip, which keeps DefaultVectorUtilSupport's separate
// multiply / add order).  This form is the reference's *native* flavour — fused multiply-adds, like
// jvector_simd_kernels.cpp:208-286 (MulAdd) — with a fully specified order, so it is still reproducible bit for bit:
//   dot(q, v)  = fma chain over k ascending from +0:  acc = fmaf(q[k], v[k], acc)
//                (v_mfma_f32_32x32x2_f32 is bitwise a k-ordered f32 fmaf chain with no wider internal accumulation,
//                cdna_hip_programming.md §3 "FP32-input MFMA"; operand k of lane l is l >> 5, so step s feeds k = 2s, 2s+1)
//   |q|^2, |v|^2 = the same chain on (x, x), on the vector ALU from the staged tile
//   DOT     (1 + dot) / 2
//   COSINE  (1 + (float)((double)dot / sqrt((double)(|q|^2 * |v|^2)))) / 2          (cosine_finish of jv_device.h)
//   L2      d2 = fmaf(-2, dot, |q|^2 + |v|^2), clamped at 0;  1 / (1 + d2)          (the GEMM identity: loses relative
//           precision in d2 for near-identical pairs, not in the similarity)
// It agrees with the bit-exact path to ~1e-7 relative (f32 round-off class; tests hold it to 1e-5, the north_star bound).
//
// One wavefront per block computes a 32-query x 128-vector tile: 4 accumulators of 32x32 (64 VGPRs), K staged through
// LDS 32 columns at a time (rows padded to 33 floats: the operand reads As[row][k] are bank-conflict free).
// Wave API used: gs_lane, gs_barrier, gs_f32x16, gs_mfma_32x32x2, gs_fmaf, gs_sqrt.
#pragma once

#include <cstdint>

namespace jv {

struct EdParams {
    const float *vecs;     // [*, D] device rows; the scan covers rows first .. first + count
    const float *queries;  // [Q, D]
    float *out;            // [Q, count]
    int64_t first, count;
    int D, Q;
};

constexpr int ED_TQ = 32, ED_TN = 128, ED_KB = 32, ED_LD = ED_KB + 1;
constexpr int ED_LDS_FLOATS = (ED_TQ + ED_TN) * ED_LD + ED_TQ + ED_TN;  // staged tiles + the two norm arrays

template <int VSF>
GS_FN float ed_finish(float dot, float qn, float vn)
{
    if (VSF == 0 /* L2 */) {
        float d2 = gs_fmaf(-2.0f, dot, qn + vn);
        if (d2 < 0.0f) d2 = 0.0f;
        return 1.0f / (1.0f + d2);
    }
    if (VSF == 2 /* cosine */) {
        const float prod = qn * vn;
        dot = (float)((double)dot / gs_sqrt((double)prod));
    }
    return (1.0f + dot) / 2.0f;
}

// Global loads of K columns kb .. kb + 32 of both tiles into registers (zero beyond D / Q / count: a zero product leaves a
// chain as is).  Element (row = (lane >> 5) + 2 i, column = lane & 31): a half wave reads one 128-byte row segment; addresses
// are a block-uniform base plus one 32-bit lane offset, and row pairs advance by a uniform 2 D.
GS_FN void ed_load_chunk(const EdParams &p, int64_t n0, int q0, int kb, int lane, float (&ra)[16], float (&rb)[64])
{
    const int rl = lane >> 5, cl = lane & 31;
    const bool k_ok = kb + cl < p.D;
    const int lane_off = rl * p.D + cl;
    const float *qbase = p.queries + (int64_t)q0 * p.D + kb;
    const float *vbase = p.vecs + (p.first + n0) * p.D + kb;
#pragma unroll
    for (int i = 0; i < 16; ++i)  // query tile: 32 rows
        ra[i] = (k_ok && q0 + rl + 2 * i < p.Q) ? (qbase + (int64_t)(2 * i) * p.D)[lane_off] : 0.0f;
#pragma unroll
    for (int i = 0; i < 64; ++i)  // vector tile: 128 rows
        rb[i] = (k_ok && n0 + rl + 2 * i < p.count) ? (vbase + (int64_t)(2 * i) * p.D)[lane_off] : 0.0f;
}

// tile (n0 .. n0 + 128) x (q0 .. q0 + 32); n0 is relative to p.first.  lds: ED_LDS_FLOATS floats.
// Two-stage pipeline: the loads of chunk c + 1 are issued right after chunk c has been stored to LDS, so they are in flight
// while chunk c's norms and 64 MFMAs run; one resident wave per SIMD already overlaps memory latency with the matrix pipe.
template <int VSF>
GS_FN void ed_tile(const EdParams &p, int64_t n0, int q0, float *lds)
{
    const int lane = gs_lane();
    const int lo = lane & 31, hi = lane >> 5;
    float *As = lds, *Bs = As + ED_TQ * ED_LD, *qn = Bs + ED_TN * ED_LD, *vn = qn + ED_TQ;
    gs_f32x16 acc[4] = {};
    float nq = 0.0f, nv0 = 0.0f, nv1 = 0.0f;  // |q|^2 of tile row `lane` (lanes < 32), |v|^2 of tile rows lane, lane + 64
    float ra[16], rb[64];
    ed_load_chunk(p, n0, q0, 0, lane, ra, rb);

    for (int kb = 0; kb < p.D; kb += ED_KB) {
        {
            const int rl = lane >> 5, cl = lane & 31;
#pragma unroll
            for (int i = 0; i < 16; ++i) As[(rl + 2 * i) * ED_LD + cl] = ra[i];
#pragma unroll
            for (int i = 0; i < 64; ++i) Bs[(rl + 2 * i) * ED_LD + cl] = rb[i];
        }
        gs_barrier();
        if (kb + ED_KB < p.D) ed_load_chunk(p, n0, q0, kb + ED_KB, lane, ra, rb);  // prefetch: consumed after the MFMAs
        if (VSF != 1 /* not DOT: the norms */) {
#pragma unroll 4
            for (int c = 0; c < ED_KB; ++c) {
                if (lane < ED_TQ) {
                    const float a = As[lane * ED_LD + c];
                    nq = gs_fmaf(a, a, nq);
                }
                const float b0 = Bs[lane * ED_LD + c], b1 = Bs[(lane + 64) * ED_LD + c];
                nv0 = gs_fmaf(b0, b0, nv0);
                nv1 = gs_fmaf(b1, b1, nv1);
            }
        }
        // ---- 16 K-steps of 2: lane l supplies A[i = l & 31][k = 2s + (l >> 5)] and B[k][j = l & 31] of each column tile ----
#pragma unroll 4
        for (int s = 0; s < ED_KB / 2; ++s) {
            const int k = 2 * s + hi;
            const float a = As[lo * ED_LD + k];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = gs_mfma_32x32x2(a, Bs[(32 * t + lo) * ED_LD + k], acc[t]);
        }
        gs_barrier();
    }

    if (VSF != 1) {
        if (lane < ED_TQ) qn[lane] = nq;
        vn[lane] = nv0;
        vn[lane + 64] = nv1;
        gs_barrier();
    }
    // ---- C/D layout (dtype independent on gfx950): col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) ----
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int j = 32 * t + lo;
        const int64_t n = n0 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int q = q0 + i;
            if (q < p.Q && n < p.count)
                p.out[(int64_t)q * p.count + n] = ed_finish<VSF>(acc[t][r], VSF != 1 ? qn[i] : 0.0f, VSF != 1 ? vn[j] : 0.0f);
        }
    }
}

// XCD-aware block -> tile map: the 8 XCDs take consecutive block ids round-robin and each has its own L2, so give every
// XCD one contiguous range of tiles with the query tiles of one vector tile adjacent (they re-read the same 128 rows).
// Returns false for the padding blocks of the last partial round.
GS_FN bool ed_block_to_tile(int64_t block, int64_t blocks_padded, int64_t n_tiles, int q_tiles, int64_t *n_tile, int *q_tile)
{
    const int64_t per_xcd = blocks_padded / 8;
    const int64_t logical = (block % 8) * per_xcd + block / 8;
    if (logical >= n_tiles * q_tiles) return false;
    *n_tile = logical / q_tiles;
    *q_tile = (int)(logical % q_tiles);
    return true;
}

}  // namespace jv
