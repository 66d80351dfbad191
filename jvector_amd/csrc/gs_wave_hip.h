// gs_wave_hip.h — the wave API of gs_body.h / an_body.h on the GPU (wave64, one wavefront per block).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#define GS_FN __device__ __forceinline__
#define GS_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// a device function that stays a CALL: its registers are allocated on their own (gs_rr_round — the fused rerank's 130 staging registers
// must not weigh on the allocation of the expansion loop, which sits at the 256-register wall).  Callable from the 2-waves-per-SIMD kernels.
#define GS_NOINLINE __device__ __attribute__((noinline))
// a pointer into the workgroup's LDS block that went through a call is a flat pointer to the compiler; cast back, the accesses are ds_ again
#define GS_LDS_AS __attribute__((address_space(3)))
// likewise a pointer into device memory: global_load instead of flat_load (a flat load counts on the LDS counter too: every wait for an
// LDS read would wait for the row requests in flight)
#define GS_GLOBAL_AS __attribute__((address_space(1)))
__device__ __forceinline__ int gs_lane() { return (int)threadIdx.x; }
#ifdef GS_WAVE_SCOPE_BARRIER
// the workgroup form (gx_body.h): gs_body.h runs in ONE wave of a larger workgroup, so its sync points are wave-scope — LDS
// operations of a wave are performed in program order, what is needed is that the compiler neither moves nor caches accesses
// across the point (the fences emit no instruction at this scope) and that the lanes have reconverged
__device__ __forceinline__ void gs_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#else
__device__ __forceinline__ void gs_barrier() { __syncthreads(); }
#endif
// multi-wave workgroups (ed_body.h): thread index inside the block and the workgroup barrier
__device__ __forceinline__ int gs_tid() { return (int)threadIdx.x; }
__device__ __forceinline__ void gs_block_barrier() { __syncthreads(); }
__device__ __forceinline__ int gs_block_threads() { return (int)blockDim.x; }
// LDS flags between the waves of a workgroup (gx_body.h): acquire / release at workgroup scope, an LDS atomic add, and the pause
// inside a spin-wait (the waiting wave gives its issue slots to the others)
// (The flags order LDS accesses only — a slot's keys against its READY flag, a request's fields against the ring tail.  LDS
// operations of one wave are performed in issue order, so "every earlier LDS operation has been issued and the compiler moves nothing
// across" is all a release / acquire needs here; the workgroup-scope atomics of the memory model would also wait for the wave's
// outstanding GLOBAL accesses — s_waitcnt vmcnt(0) — which costs an expander nothing but stalls the control wave behind its
// fire-and-forget stores.)
__device__ __forceinline__ int32_t gs_lds_load(const int32_t *p)
{
    const int32_t v = *(const volatile __attribute__((address_space(3))) int32_t *)p;
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return v;
}
__device__ __forceinline__ void gs_lds_store(int32_t *p, int32_t v)
{
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    *(volatile __attribute__((address_space(3))) int32_t *)p = v;
}
__device__ __forceinline__ int32_t gs_lds_add(int32_t *p, int32_t v)
{
    return __hip_atomic_fetch_add((__attribute__((address_space(3))) int32_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void gs_spin_pause() { __builtin_amdgcn_s_sleep(1); }
// keeps the instruction scheduler from moving anything across this point (software pipelines written in source order)
__device__ __forceinline__ void gs_sched_fence() { __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ uint64_t gs_ballot(bool p) { return (uint64_t)__ballot(p ? 1 : 0); }
#ifdef GS_UNIFORM_SHFL
// every gs_shfl of gs_body.h / gx_body.h reads ONE lane for the whole wave (src is wave-uniform): v_readlane_b32 through the scalar
// unit instead of two ds_bpermute_b32 round trips through the LDS crossbar
__device__ __forceinline__ long long gs_shfl(long long v, int src)
{
    const int lo = __builtin_amdgcn_readlane((int)(unsigned long long)v, src), hi = __builtin_amdgcn_readlane((int)((unsigned long long)v >> 32), src);
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
// max / min of a 64-bit key over the wave in DPP steps (row_shr 1, 2, 4, 8 inside the rows of 16 lanes, row_bcast15 / row_bcast31
// across them; lane 63 ends up with the result): 6 steps of two v_mov_dpp + compare + select instead of 12 dependent ds_bpermute
// round trips.  All 64 lanes must be active (every call site of gs_body.h is wave-uniform).
#define GS_HAVE_WAVE_REDUCE 1
template <bool MAX>
__device__ __forceinline__ long long gs_wave_reduce64(long long v)
{
#define JV_DPP_STEP(CTRL, ROWS)                                                                                       \
    do {                                                                                                              \
        const int lo_ = __builtin_amdgcn_update_dpp((int)(unsigned long long)v, (int)(unsigned long long)v, CTRL, ROWS, 0xf, false);              \
        const int hi_ = __builtin_amdgcn_update_dpp((int)((unsigned long long)v >> 32), (int)((unsigned long long)v >> 32), CTRL, ROWS, 0xf, false); \
        const long long o_ = (long long)(((unsigned long long)(unsigned)hi_ << 32) | (unsigned long long)(unsigned)lo_);                          \
        v = MAX ? (o_ > v ? o_ : v) : (o_ < v ? o_ : v);                                                              \
    } while (0)
    JV_DPP_STEP(0x111, 0xf);  // row_shr:1
    JV_DPP_STEP(0x112, 0xf);  // row_shr:2
    JV_DPP_STEP(0x114, 0xf);  // row_shr:4
    JV_DPP_STEP(0x118, 0xf);  // row_shr:8   -> lane 15 of every row holds its row's result
    JV_DPP_STEP(0x142, 0xa);  // row_bcast15 -> rows 1 and 3 take in rows 0 and 2
    JV_DPP_STEP(0x143, 0xc);  // row_bcast31 -> rows 2 and 3 take in lane 31: lane 63 holds the wave's result
#undef JV_DPP_STEP
    const int lo = __builtin_amdgcn_readlane((int)(unsigned long long)v, 63), hi = __builtin_amdgcn_readlane((int)((unsigned long long)v >> 32), 63);
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
__device__ __forceinline__ long long gs_wave_max(long long v) { return gs_wave_reduce64<true>(v); }
__device__ __forceinline__ long long gs_wave_min(long long v) { return gs_wave_reduce64<false>(v); }
// the same over 32-bit values: one v_max / v_min with a DPP operand per step
#define GS_HAVE_WAVE_REDUCE32 1
#define JV_DPP_RED32(T, NAME, EXPR)                                                                  \
    __device__ __forceinline__ T NAME(T v)                                                           \
    {                                                                                                \
        _Pragma("unroll") for (int s_ = 0; s_ < 6; ++s_)                                             \
        {                                                                                            \
            T o_;                                                                                    \
            switch (s_) {                                                                            \
            case 0: o_ = (T)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xf, 0xf, false); break; \
            case 1: o_ = (T)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xf, 0xf, false); break; \
            case 2: o_ = (T)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xf, 0xf, false); break; \
            case 3: o_ = (T)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118, 0xf, 0xf, false); break; \
            case 4: o_ = (T)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false); break; \
            default: o_ = (T)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false); break; \
            }                                                                                        \
            v = EXPR;                                                                                \
        }                                                                                            \
        return (T)__builtin_amdgcn_readlane((int)v, 63);                                             \
    }
JV_DPP_RED32(int32_t, gs_wave_max_i32, (o_ > v ? o_ : v))
JV_DPP_RED32(int32_t, gs_wave_min_i32, (o_ < v ? o_ : v))
JV_DPP_RED32(uint32_t, gs_wave_max_u32, (o_ > v ? o_ : v))
#undef JV_DPP_RED32
__device__ __forceinline__ int32_t gs_uniform(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
#else
__device__ __forceinline__ long long gs_shfl(long long v, int src) { return __shfl(v, src, 64); }
#endif
// the 32-bit value of ONE lane, `src` wave-uniform: v_readlane_b32 — the result is a scalar register (gs_rr_round: the operand of a multiply)
__device__ __forceinline__ uint32_t gs_bcast32(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
__device__ __forceinline__ long long gs_shfl_xor(long long v, int m) { return __shfl_xor(v, m, 64); }
// the value is what it was, but the compiler may not reason about where it came from (keeps per-lane address arithmetic inside the
// loop that uses it instead of hoisting dozens of 64-bit pointers out of the search loop and spilling them)
#define GS_OPAQUE_I32(x) asm volatile("" : "+v"(x))
__device__ __forceinline__ int32_t gs_shfl32(int32_t v, int src) { return __builtin_amdgcn_ds_bpermute(src << 2, v); }
// packed f32 pairs (v_pk_mul_f32 / v_pk_add_f32; -ffp-contract=off keeps the multiply and the add apart)
#define GS_HAVE_PK_F32 1
typedef float gs_pk2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t gs_perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
__device__ __forceinline__ int32_t gs_cas(int32_t *p, int32_t expect, int32_t desired) { return atomicCAS(p, expect, desired); }
// LDS atomic (ds_cmpst_rtn_b32): p points into the workgroup's LDS block
__device__ __forceinline__ uint32_t gs_lds_cas(uint32_t *p, uint32_t expect, uint32_t desired)
{
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    lds_u32 *lp = (lds_u32 *)p;
    __hip_atomic_compare_exchange_strong(lp, &expect, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return expect;
}
// LDS-DMA load (global_load_lds_dword): 4 bytes per active lane from its own global address straight into LDS at
// lds + 4 * lane — no destination register, so nothing in the wave ever waits for it explicitly
__device__ __forceinline__ void gs_prefetch_lds(const void *g, void *lds)
{
    typedef __attribute__((address_space(1))) const void gptr;
    typedef __attribute__((address_space(3))) void lptr;
    __builtin_amdgcn_global_load_lds((gptr *)g, (lptr *)lds, 4, 0, 0);
}
__device__ __forceinline__ uint32_t gs_fetch_add(uint32_t *p, uint32_t v) { return atomicAdd(p, v); }
__device__ __forceinline__ void gs_fetch_add64(unsigned long long *p, unsigned long long v) { (void)atomicAdd(p, v); }
#define GS_CLOCK() ((unsigned long long)__builtin_readcyclecounter())
__device__ __forceinline__ void gs_fence() { __threadfence(); }
__device__ __forceinline__ double gs_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float gs_rsq_approx(float x) { return __builtin_amdgcn_rsqf(x); }
// ed_body.h: f32-input MFMA (bitwise a k-ordered fmaf chain; A: lane l holds A[l & 31][l >> 5], B: B[l >> 5][l & 31]) and fmaf
typedef float gs_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ gs_f32x16 gs_mfma_32x32x2(float a, float b, gs_f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float gs_fmaf(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// km_body.h: every lane receives the 64 values of the wave (64 v_readlane_b32; the results are wave-uniform)
__device__ __forceinline__ void gs_gather64(float v, float (&out)[64])
{
#pragma unroll
    for (int i = 0; i < 64; ++i) out[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i));
}
