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
// A workgroup of 4 wavefronts computes a 128-query x 128-vector tile; wavefront w owns queries 32 w .. 32 w + 31 against all
// 128 vectors (4 accumulators of 32x32 = 64 VGPRs).  K is staged through LDS 32 columns at a time, K-MAJOR: element (row, k)
// of a tile lives at [k * 129 + row], so that
//   * the MFMA operand reads (32 lanes = 32 consecutive rows at one k) are bank-conflict free,
//   * the staging writes are too: a lane loads 4 consecutive k of one row with ONE 16-byte global load (8 lanes cover a row's
//     128-byte segment: whole lines) and scatters them with 4 ds_write_b32 whose bank is (4 kq + c + row) mod 32 — distinct
//     over the 32 lanes of a write group (rows 0..3 x kq 0..7).
// Per 32-column chunk the block loads 32 KB for 1.05 Mflop (32 flop/B; round 1's one-wave 32 x 128 tile: 12.8 flop/B with
// eighty 4-byte loads per lane — it reached 0.35 of the MFMA peak).  The next chunk's 8 loads per lane are in flight while
// the 64 MFMAs of the current one run; ~120 VGPRs and 34 KB of LDS allow 4 blocks = 16 waves per CU.
// Wave API used: gs_tid (0..255), gs_block_barrier, gs_sched_fence, gs_f32x16, gs_mfma_32x32x2, gs_fmaf, gs_sqrt.
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

constexpr int ED_WAVES = 4, ED_THREADS = 64 * ED_WAVES;
constexpr int ED_TQ = 32 * ED_WAVES, ED_TN = 128, ED_KB = 32, ED_LDW = 129;
constexpr int ED_LDS_FLOATS = 2 * ED_KB * ED_LDW + ED_TQ + ED_TN;  // staged tiles (K-major) + the two norm arrays

struct alignas(16) ed_f4 { float x, y, z, w; };

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

// Global loads of K columns kb .. kb + 32 of both tiles into registers: thread t fetches columns 4 (t & 7) .. + 3 of rows
// (t >> 3) + 32 i, i = 0..3, of each tile (zero beyond D / Q / count: a zero product leaves a chain as is).
// vec4: rows are 16-byte aligned and D % 4 == 0, so the four columns are one 16-byte load.
GS_FN void ed_load_chunk(const EdParams &p, int64_t n0, int q0, int kb, int tid, bool vec4, ed_f4 (&ra)[4], ed_f4 (&rb)[4])
{
    const int r0 = tid >> 3, k = kb + 4 * (tid & 7);
    // interior tile and chunk (block-uniform; all but the edge tiles): eight unconditional 16-byte loads, no per-row branches
    if (vec4 && q0 + ED_TQ <= p.Q && n0 + ED_TN <= p.count && kb + ED_KB <= p.D) {
        const float *qa = p.queries + (int64_t)(q0 + r0) * p.D + k;
        const float *vb = p.vecs + (p.first + n0 + r0) * p.D + k;
        const int64_t step = (int64_t)32 * p.D;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = *reinterpret_cast<const ed_f4 *>(qa + i * step);
            rb[i] = *reinterpret_cast<const ed_f4 *>(vb + i * step);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = r0 + 32 * i;
        ed_f4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (q0 + row < p.Q && k < p.D) {
            const float *src = p.queries + (int64_t)(q0 + row) * p.D + k;
            if (vec4) a = *reinterpret_cast<const ed_f4 *>(src);
            else {
                a.x = src[0];
                if (k + 1 < p.D) a.y = src[1];
                if (k + 2 < p.D) a.z = src[2];
                if (k + 3 < p.D) a.w = src[3];
            }
        }
        if (n0 + row < p.count && k < p.D) {
            const float *src = p.vecs + (p.first + n0 + row) * p.D + k;
            if (vec4) b = *reinterpret_cast<const ed_f4 *>(src);
            else {
                b.x = src[0];
                if (k + 1 < p.D) b.y = src[1];
                if (k + 2 < p.D) b.z = src[2];
                if (k + 3 < p.D) b.w = src[3];
            }
        }
        ra[i] = a;
        rb[i] = b;
    }
}

// tile (n0 .. n0 + 128) x (q0 .. q0 + 128); n0 is relative to p.first.  lds: ED_LDS_FLOATS floats.  Runs on ED_THREADS threads.
template <int VSF>
GS_FN void ed_tile(const EdParams &p, int64_t n0, int q0, float *lds)
{
    const int tid = gs_tid();
    const int wave = tid >> 6, lane = tid & 63;
    const int lo = lane & 31, hi = lane >> 5;
    float *As = lds, *Bs = As + ED_KB * ED_LDW, *qn = Bs + ED_KB * ED_LDW, *vn = qn + ED_TQ;
    const bool vec4 = (p.D & 3) == 0 && ((reinterpret_cast<uintptr_t>(p.vecs) | reinterpret_cast<uintptr_t>(p.queries)) & 15) == 0;
    const bool wave_active = q0 + 32 * wave < p.Q;  // wave-uniform
    gs_f32x16 acc[4] = {};
    float nrm = 0.0f;  // threads 0..127: |v|^2 of tile row tid; threads 128..255: |q|^2 of tile row tid - 128
    ed_f4 ra[4], rb[4];
    ed_load_chunk(p, n0, q0, 0, tid, vec4, ra, rb);

    for (int kb = 0; kb < p.D; kb += ED_KB) {
        {
            const int r0 = tid >> 3, kq = 4 * (tid & 7);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = r0 + 32 * i;
                As[(kq + 0) * ED_LDW + row] = ra[i].x;
                As[(kq + 1) * ED_LDW + row] = ra[i].y;
                As[(kq + 2) * ED_LDW + row] = ra[i].z;
                As[(kq + 3) * ED_LDW + row] = ra[i].w;
                Bs[(kq + 0) * ED_LDW + row] = rb[i].x;
                Bs[(kq + 1) * ED_LDW + row] = rb[i].y;
                Bs[(kq + 2) * ED_LDW + row] = rb[i].z;
                Bs[(kq + 3) * ED_LDW + row] = rb[i].w;
            }
        }
        gs_block_barrier();
        if (kb + ED_KB < p.D) ed_load_chunk(p, n0, q0, kb + ED_KB, tid, vec4, ra, rb);  // prefetch: consumed after the MFMAs
        if (VSF != 1 /* not DOT: the norms, k ascending */) {
            const float *src = (tid < ED_TN) ? Bs + tid : As + (tid - ED_TN);
#pragma unroll 8
            for (int c = 0; c < ED_KB; ++c) {
                const float x = src[c * ED_LDW];
                nrm = gs_fmaf(x, x, nrm);
            }
        }
        // ---- 16 K-steps of 2: lane l supplies A[i = l & 31][k = 2s + (l >> 5)] and B[k][j = l & 31] of each column tile ----
        // Operands of step s + 1 are read from LDS BEFORE the four MFMAs of step s are issued (two register sets), so that an
        // MFMA never waits for the read issued right in front of it: with reads and MFMAs back to back the matrix pipe idled
        // ~27 % of the time on LDS latency (SQ_WAIT_INST_LDS, profiles/r3_m).
        const float *a_col = As + 32 * wave + lo + hi * ED_LDW;   // k = 2 s + hi
        const float *b_col = Bs + lo + hi * ED_LDW;
        if (wave_active) {  // a wavefront whose 32 query rows lie beyond Q only helps with the staging (small batches)
            float a_cur = a_col[0], b_cur[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) b_cur[t] = b_col[32 * t];
#pragma unroll
            for (int s = 0; s < ED_KB / 2; ++s) {
                float a_nxt = 0.0f, b_nxt[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (s + 1 < ED_KB / 2) {
                    a_nxt = a_col[(2 * s + 2) * ED_LDW];
#pragma unroll
                    for (int t = 0; t < 4; ++t) b_nxt[t] = b_col[(2 * s + 2) * ED_LDW + 32 * t];
                }
                gs_sched_fence();
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = gs_mfma_32x32x2(a_cur, b_cur[t], acc[t]);
                gs_sched_fence();
                a_cur = a_nxt;
#pragma unroll
                for (int t = 0; t < 4; ++t) b_cur[t] = b_nxt[t];
            }
        }
        gs_block_barrier();
    }

    if (VSF != 1) {
        if (tid < ED_TN) vn[tid] = nrm;
        else qn[tid - ED_TN] = nrm;
        gs_block_barrier();
    }
    // ---- C/D layout (dtype independent on gfx950): col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) ----
    if (!wave_active) return;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int j = 32 * t + lo;
        const int64_t n = n0 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * hi;
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
