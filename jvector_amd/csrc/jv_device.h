// jv_device.h — device-side helpers shared by the kernels.
// All translation units are compiled with -ffp-contract=off: the reference (Java) never fuses a*b+c,
// so bit-exact parity with DefaultVectorUtilSupport needs separate v_mul_f32 / v_add_f32.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace jv {

enum : int { VSF_L2 = 0, VSF_DOT = 1, VSF_COS = 2, VSF_RAW = 3 /* internal: untransformed table sum */ };

// VectorSimilarityFunction.compare transforms (VectorSimilarityFunction.java:40,54,67;
// PQDecoder.java:68,79,126; FusedPQDecoder.java:125,139,217)
__device__ __forceinline__ float score_from_raw(int vsf, float raw)
{
    if (vsf == VSF_L2) return 1.0f / (1.0f + raw);
    if (vsf == VSF_RAW) return raw;
    return (1.0f + raw) / 2.0f;
}

// (float)(sum / Math.sqrt(aMag * bMag)): float product, double sqrt and divide, narrowed
// (DefaultVectorUtilSupport.java:138,155; VectorUtilSupport.java:164)
__device__ __forceinline__ float cosine_finish(float sum, float amag, float bmag)
{
    float prod = amag * bmag;
    return (float)((double)sum / sqrt((double)prod));
}

// NumericUtils.floatToSortableInt (NumericUtils.java:49-65) mapped to an unsigned order-preserving key
__device__ __forceinline__ uint32_t float_to_ordered_u32(float v)
{
    int32_t bits = (v != v) ? 0x7fc00000 : __float_as_int(v);  // Float.floatToIntBits canonical NaN
    int32_t s = bits ^ ((bits >> 31) & 0x7fffffff);            // sortable signed int
    return (uint32_t)s ^ 0x80000000u;                          // signed order -> unsigned order
}
__device__ __forceinline__ float ordered_u32_to_float(uint32_t u)
{
    int32_t s = (int32_t)(u ^ 0x80000000u);
    int32_t bits = s ^ ((s >> 31) & 0x7fffffff);
    return __int_as_float(bits);
}

}  // namespace jv
