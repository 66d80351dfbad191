// TEST INFRASTRUCTURE — part of the recipe that builds oracle/_ref/libjvector_ref.so (see build.sh).  The reference's library
// exports ONE symbol per kernel, dispatched by CPUID at load time (jvector_simd.cpp:160-180); the tests want every lane width
// on every host, so this file adds `jvref_<isa>_<kernel>` entry points that call the AVX3:: / AVX2:: / SSE42:: instances
// directly.  The kernel list and the namespace declarations come from the reference's own headers at compile time
// (jvector_simd_kernel_list.h:36-62, jvector_simd_kernels.h:33-45); nothing of them is copied here.
#include "jvector_simd.h"
#include "jvector_simd_kernels.h"

#define JVREF_EXPORT extern "C" __attribute__((visibility("default")))

#define KERNEL_ENTRY(ret_type, name, params, names)                                        \
    JVREF_EXPORT ret_type jvref_avx3_##name params { return AVX3::name names; }            \
    JVREF_EXPORT ret_type jvref_avx2_##name params { return AVX2::name names; }            \
    JVREF_EXPORT ret_type jvref_sse42_##name params { return SSE42::name names; }
JVECTOR_SIMD_KERNEL_LIST
#undef KERNEL_ENTRY

// which stand-in this build used, for the tests' report
JVREF_EXPORT const char *jvref_build_info(void)
{
    return "reference jvector_simd_kernels.cpp + jvector_simd.cpp, unmodified, over oracle/ref_build/hwy/highway.h "
           "(scalar lane emulation: avx3 = 16 f32 lanes + fma, avx2 = 8 lanes + fma, sse42 = 4 lanes, no fma)";
}
