/*
 * jvector_simd_compat.h — the reference's per-pair native SPI, exported unchanged by libjvector_hip.so so
 * that a NativeVectorUtilSupport-style Java binding (one downcall per pair) keeps linking against it.
 *
 * Same 22 kernel symbols + 2 introspection symbols, same signatures, as
 *   /root/reference/jvector-native/src/main/native/src/jvector_simd_kernel_list.h:36-62
 *   /root/reference/jvector-native/src/main/native/src/jvector_simd.h:47,53
 *
 * These are HOST functions on HOST pointers by construction of that SPI (one 8..1536-float pair per call
 * cannot amortise a kernel launch); they are boundary completeness, not the accelerated path and not a
 * fallback for it — the batched jv_hip_* entry points in jvector_hip.h never route through them.
 * Arithmetic follows the scalar DefaultVectorUtilSupport order (the same order the HIP kernels reproduce),
 * so per-pair and batched results are bit-identical.
 */
#ifndef JVECTOR_SIMD_COMPAT_H
#define JVECTOR_SIMD_COMPAT_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#define JVC_API __attribute__((visibility("default")))

JVC_API float cosine_f32(const float *a, size_t aoffset, const float *b, size_t boffset, size_t length);
JVC_API float dot_product_f32(const float *a, size_t aoffset, const float *b, size_t boffset, size_t length);
JVC_API float euclidean_f32(const float *a, size_t aoffset, const float *b, size_t boffset, size_t length);
JVC_API void add_in_place_f32(float *v1, const float *v2, size_t length);
JVC_API void add_scalar_in_place_f32(float *v1, float value, size_t length);
JVC_API void sub_in_place_f32(float *v1, const float *v2, size_t length);
JVC_API void sub_scalar_in_place_f32(float *v1, float value, size_t length);
JVC_API float max_f32(const float *v, size_t length);
JVC_API void min_in_place_f32(float *v1, const float *v2, size_t length);
JVC_API float assemble_and_sum_f32(const float *data, int dataBase, const unsigned char *baseOffsets,
                                   int baseOffsetsOffset, size_t baseOffsetsLength);
JVC_API float assemble_and_sum_pq_f32(const float *data, size_t subspaceCount, const unsigned char *baseOffsets1,
                                      int baseOffsetsOffset1, const unsigned char *baseOffsets2,
                                      int baseOffsetsOffset2, int clusterCount);
JVC_API float pq_decoded_cosine_similarity_f32(const unsigned char *baseOffsets, int baseOffsetsOffset,
                                               size_t baseOffsetsLength, int clusterCount, const float *partialSums,
                                               const float *aMagnitude, float bMagnitude);
JVC_API void calculate_partial_sums_dot_f32(const float *codebook, int codebookIndex, size_t size, int clusterCount,
                                            const float *query, int queryOffset, float *partialSums);
JVC_API void calculate_partial_sums_euclidean_f32(const float *codebook, int codebookIndex, size_t size,
                                                  int clusterCount, const float *query, int queryOffset,
                                                  float *partialSums);
JVC_API void calculate_partial_sums_self_magnitude_f32(const float *codebook, int codebookIndex, size_t size,
                                                       int clusterCount, float *partialSums);
JVC_API void nvq_quantize_8bit(const float *vector, size_t length, float alpha, float x0, float minValue,
                               float maxValue, unsigned char *destination);
JVC_API float nvq_loss(const float *vector, size_t length, float alpha, float x0, float minValue, float maxValue,
                       int nBits);
JVC_API float nvq_uniform_loss(const float *vector, size_t length, float minValue, float maxValue, int nBits);
JVC_API float nvq_square_l2_distance_8bit(const float *vector, const unsigned char *quantized, size_t length,
                                          float alpha, float x0, float minValue, float maxValue);
JVC_API float nvq_dot_product_8bit(const float *vector, const unsigned char *quantized, size_t length, float alpha,
                                   float x0, float minValue, float maxValue);
JVC_API int64_t nvq_cosine_8bit_packed(const float *vector, const unsigned char *quantized, size_t length,
                                       float alpha, float x0, float minValue, float maxValue, const float *centroid);
JVC_API void nvq_shuffle_query_in_place_8bit(float *vector, size_t length);

/* "gfx950-host" — the tier string the Java side logs (VectorizationProvider.java:131-136) */
JVC_API const char *jvector_simd_get_active_isa(void);
/* value of JVECTOR_MAX_ISA at load time or NULL; kept for binding compatibility, has no effect here */
JVC_API const char *jvector_simd_get_max_isa_env(void);

#ifdef __cplusplus
}
#endif
#endif
