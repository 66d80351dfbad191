/*
 * Per-pair SPI of the HIP provider: the 34 VectorUtilSupport methods (B/vector/VectorUtilSupport.java:36-245).
 *
 * Same construction as the reference's native provider (jvector-native/.../vector/NativeVectorUtilSupport.java:36-299):
 * extend jvector-twenty's PanamaVectorUtilSupport (package-private, NOT final: PanamaVectorUtilSupport.java:32), override its
 * six storage hooks for MemorySegment-backed vectors (:45-67), and route the kernels that have a native symbol to it.  The
 * reference's own NativeVectorUtilSupport cannot be subclassed (it is `final`, :36) and its NativeSimdOps resolves symbols
 * in libjvector; here they come from libjvector_hip.so through HipCompatOps (include/jvector_simd_compat.h: same 24
 * signatures).  Methods without a native symbol (sum, scale, normalize-free helpers, hamming, bulk shuffles ...) are inherited
 * from the Panama implementation, exactly as in the reference.
 * Module dependencies: jvector-base, jvector-twenty (PanamaVectorUtilSupport), jvector-native (MemorySegmentVectorFloat /
 * MemorySegmentByteSequence / MemorySegmentVectorProvider, all public).
 * NOT compiled in this repository (no JDK in the build image).
 */
package io.github.jbellis.jvector.vector;

import java.lang.foreign.MemorySegment;
import java.nio.ByteOrder;

import io.github.jbellis.jvector.vector.hip.HipBatchScorer;
import io.github.jbellis.jvector.vector.hip.HipCompatOps;
import io.github.jbellis.jvector.vector.hip.HipOps;
import io.github.jbellis.jvector.vector.types.ByteSequence;
import io.github.jbellis.jvector.vector.types.VectorFloat;
import jdk.incubator.vector.ByteVector;
import jdk.incubator.vector.FloatVector;
import jdk.incubator.vector.VectorMask;
import jdk.incubator.vector.VectorSpecies;

final class HipVectorUtilSupport extends PanamaVectorUtilSupport {
    HipVectorUtilSupport() {}

    private static MemorySegment f(VectorFloat<?> v) { return ((MemorySegmentVectorFloat) v).get(); }
    private static MemorySegment b(ByteSequence<?> s) { return ((MemorySegmentByteSequence) s).get(); }

    /** "gfx950:sramecc+:xnack-" — the analogue of NativeVectorUtilSupport.getActiveIsa() (:45-49) */
    public String getActiveArch() { return HipOps.activeArch(0); }
    /** what the host-side compat kernels report (jvector_simd_get_active_isa, jvector_simd.h:47) */
    public String getActiveIsa() { return HipCompatOps.activeIsa(); }
    public String getMaxIsaEnv() { return HipCompatOps.maxIsaEnv(); }

    /** Batched scorer bound to one device context; one per searcher thread (contexts are not thread-safe). */
    public HipBatchScorer newBatchScorer(int device) { return new HipBatchScorer(device); }

    // ---- storage hooks of the Panama base class, for MemorySegment-backed vectors (little-endian, NativeVectorUtilSupport :62-92)
    @Override
    protected FloatVector fromVectorFloat(VectorSpecies<Float> SPEC, VectorFloat<?> vector, int offset) {
        return FloatVector.fromMemorySegment(SPEC, f(vector), vector.offset(offset), ByteOrder.LITTLE_ENDIAN);
    }

    @Override
    protected FloatVector fromVectorFloat(VectorSpecies<Float> SPEC, VectorFloat<?> vector, int offset, int[] indices, int indicesOffset) {
        throw new UnsupportedOperationException("Assembly not supported with memory segments.");
    }

    @Override
    protected void intoVectorFloat(FloatVector vector, VectorFloat<?> v, int offset) {
        vector.intoMemorySegment(f(v), v.offset(offset), ByteOrder.LITTLE_ENDIAN);
    }

    @Override
    protected ByteVector fromByteSequence(VectorSpecies<Byte> SPEC, ByteSequence<?> vector, int offset) {
        return ByteVector.fromMemorySegment(SPEC, b(vector), offset, ByteOrder.LITTLE_ENDIAN);
    }

    @Override
    protected void intoByteSequence(ByteVector vector, ByteSequence<?> v, int offset) {
        vector.intoMemorySegment(b(v), offset, ByteOrder.LITTLE_ENDIAN);
    }

    @Override
    protected void intoByteSequence(ByteVector vector, ByteSequence<?> v, int offset, VectorMask<Byte> mask) {
        vector.intoMemorySegment(b(v), offset, ByteOrder.LITTLE_ENDIAN, mask);
    }

    // ---- row 1: exact dot / L2 / cosine ----
    @Override
    public float dotProduct(VectorFloat<?> v1, VectorFloat<?> v2) { return HipCompatOps.dotProduct(f(v1), 0, f(v2), 0, v1.length()); }

    @Override
    public float dotProduct(VectorFloat<?> v1, int v1offset, VectorFloat<?> v2, int v2offset, int length) {
        return HipCompatOps.dotProduct(f(v1), v1offset, f(v2), v2offset, length);
    }

    @Override
    public float squareDistance(VectorFloat<?> v1, VectorFloat<?> v2) { return HipCompatOps.euclidean(f(v1), 0, f(v2), 0, v1.length()); }

    @Override
    public float squareDistance(VectorFloat<?> v1, int v1offset, VectorFloat<?> v2, int v2offset, int length) {
        return HipCompatOps.euclidean(f(v1), v1offset, f(v2), v2offset, length);
    }

    @Override
    public float cosine(VectorFloat<?> v1, VectorFloat<?> v2) { return HipCompatOps.cosine(f(v1), 0, f(v2), 0, v1.length()); }

    @Override
    public float cosine(VectorFloat<?> v1, int v1offset, VectorFloat<?> v2, int v2offset, int length) {
        return HipCompatOps.cosine(f(v1), v1offset, f(v2), v2offset, length);
    }

    // ---- element-wise helpers ----
    @Override
    public void addInPlace(VectorFloat<?> v1, VectorFloat<?> v2) { HipCompatOps.addInPlace(f(v1), f(v2), v1.length()); }

    @Override
    public void addInPlace(VectorFloat<?> v1, float value) { HipCompatOps.addScalarInPlace(f(v1), value, v1.length()); }

    @Override
    public void subInPlace(VectorFloat<?> v1, VectorFloat<?> v2) { HipCompatOps.subInPlace(f(v1), f(v2), v1.length()); }

    @Override
    public void subInPlace(VectorFloat<?> vector, float value) { HipCompatOps.subScalarInPlace(f(vector), value, vector.length()); }

    @Override
    public float max(VectorFloat<?> v) { return HipCompatOps.max(f(v), v.length()); }

    @Override
    public void minInPlace(VectorFloat<?> v1, VectorFloat<?> v2) { HipCompatOps.minInPlace(f(v1), f(v2), v1.length()); }

    // ---- rows 2 / 5 / 6: PQ tables and ADC ----
    @Override
    public float assembleAndSum(VectorFloat<?> data, int dataBase, ByteSequence<?> baseOffsets) {
        return assembleAndSum(data, dataBase, baseOffsets, 0, baseOffsets.length());
    }

    @Override
    public float assembleAndSum(VectorFloat<?> data, int dataBase, ByteSequence<?> baseOffsets, int baseOffsetsOffset, int baseOffsetsLength) {
        assert baseOffsets.offset() == 0 : "Base offsets are expected to have an offset of 0. Found: " + baseOffsets.offset();
        return HipCompatOps.assembleAndSum(f(data), dataBase, b(baseOffsets), baseOffsetsOffset, baseOffsetsLength);
    }

    @Override
    public float assembleAndSumPQ(VectorFloat<?> codebookPartialSums, int subspaceCount, ByteSequence<?> vector1Ordinals, int vector1OrdinalOffset,
                                  ByteSequence<?> vector2Ordinals, int vector2OrdinalOffset, int clusterCount) {
        assert vector1Ordinals.offset() == 0 && vector2Ordinals.offset() == 0 : "ordinal sequences must have offset 0";
        return HipCompatOps.assembleAndSumPQ(f(codebookPartialSums), subspaceCount, b(vector1Ordinals), vector1OrdinalOffset,
                                             b(vector2Ordinals), vector2OrdinalOffset, clusterCount);
    }

    @Override
    public float pqDecodedCosineSimilarity(ByteSequence<?> encoded, int clusterCount, VectorFloat<?> partialSums, VectorFloat<?> aMagnitude, float bMagnitude) {
        return pqDecodedCosineSimilarity(encoded, 0, encoded.length(), clusterCount, partialSums, aMagnitude, bMagnitude);
    }

    @Override
    public float pqDecodedCosineSimilarity(ByteSequence<?> encoded, int encodedOffset, int encodedLength, int clusterCount,
                                           VectorFloat<?> partialSums, VectorFloat<?> aMagnitude, float bMagnitude) {
        assert encoded.offset() == 0 : "encoded is expected to have an offset of 0. Found: " + encoded.offset();
        return HipCompatOps.pqDecodedCosine(b(encoded), encodedOffset, encodedLength, clusterCount, f(partialSums), f(aMagnitude), bMagnitude);
    }

    @Override
    public void calculatePartialSums(VectorFloat<?> codebook, int codebookIndex, int size, int clusterCount, VectorFloat<?> query, int queryOffset,
                                     VectorSimilarityFunction vsf, VectorFloat<?> partialSums) {
        switch (vsf) {
            case EUCLIDEAN -> HipCompatOps.partialSumsEuclidean(f(codebook), codebookIndex, size, clusterCount, f(query), queryOffset, f(partialSums));
            case DOT_PRODUCT -> HipCompatOps.partialSumsDot(f(codebook), codebookIndex, size, clusterCount, f(query), queryOffset, f(partialSums));
            default -> throw new UnsupportedOperationException("Unsupported similarity function " + vsf);
        }
    }

    @Override
    public void calculatePartialSelfMagnitudes(VectorFloat<?> codebook, int codebookIndex, int size, int clusterCount, VectorFloat<?> partialMagnitudes) {
        HipCompatOps.partialSelfMagnitudes(f(codebook), codebookIndex, size, clusterCount, f(partialMagnitudes));
    }

    // ---- NVQ (host compat symbols; NVQ is outside the GPU hot path, SURVEY §8) ----
    @Override
    public void nvqShuffleQueryInPlace8bit(VectorFloat<?> vector) { HipCompatOps.nvqShuffleQueryInPlace8bit(f(vector), vector.length()); }

    @Override
    public void nvqQuantize8bit(VectorFloat<?> vector, float alpha, float x0, float minValue, float maxValue, ByteSequence<?> destination) {
        HipCompatOps.nvqQuantize8bit(f(vector), vector.length(), alpha, x0, minValue, maxValue, b(destination));
    }

    @Override
    public float nvqLoss(VectorFloat<?> vector, float alpha, float x0, float minValue, float maxValue, int nBits) {
        return HipCompatOps.nvqLoss(f(vector), vector.length(), alpha, x0, minValue, maxValue, nBits);
    }

    @Override
    public float nvqUniformLoss(VectorFloat<?> vector, float minValue, float maxValue, int nBits) {
        return HipCompatOps.nvqUniformLoss(f(vector), vector.length(), minValue, maxValue, nBits);
    }

    @Override
    public float nvqSquareL2Distance8bit(VectorFloat<?> vector, ByteSequence<?> quantizedVector, float alpha, float x0, float minValue, float maxValue) {
        return HipCompatOps.nvqSquareL2Distance8bit(f(vector), b(quantizedVector), vector.length(), alpha, x0, minValue, maxValue);
    }

    @Override
    public float nvqDotProduct8bit(VectorFloat<?> vector, ByteSequence<?> quantizedVector, float alpha, float x0, float minValue, float maxValue) {
        return HipCompatOps.nvqDotProduct8bit(f(vector), b(quantizedVector), vector.length(), alpha, x0, minValue, maxValue);
    }

    @Override
    public float[] nvqCosine8bit(VectorFloat<?> vector, ByteSequence<?> quantizedVector, float alpha, float x0, float minValue, float maxValue,
                                 VectorFloat<?> centroid) {
        // two floats packed lo / hi in one int64 (jvector_simd_kernel_list.h:62; unpacked as NativeVectorUtilSupport.java:289-297 does)
        long packed = HipCompatOps.nvqCosine8bitPacked(f(vector), b(quantizedVector), vector.length(), alpha, x0, minValue, maxValue, f(centroid));
        return new float[] {Float.intBitsToFloat((int) (packed & 0xFFFFFFFFL)), Float.intBitsToFloat((int) (packed >>> 32))};
    }
}
