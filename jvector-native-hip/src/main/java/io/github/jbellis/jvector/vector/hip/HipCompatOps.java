/*
 * FFM downcall handles for the reference's 24 per-pair symbols as exported by libjvector_hip.so
 * (include/jvector_simd_compat.h; signatures identical to jvector-native/src/main/native/src/jvector_simd_kernel_list.h:36-62
 * and jvector_simd.h:47,53).  Hand-written in the shape jextract generates for the reference's NativeSimdOps
 * (cnative/NativeSimdOps.java:59-60,1152-1211), INCLUDING Linker.Option.critical(true): these are the short host-code
 * kernels the per-pair SPI calls with on-heap segments (MemorySegment.ofArray), exactly like the reference's.
 * Symbols are resolved through HipOps' own lookup (libjvector_hip.so) — libjvector is never loaded.
 * NOT compiled in this repository (no JDK in the build image).
 */
package io.github.jbellis.jvector.vector.hip;

import java.lang.foreign.FunctionDescriptor;
import java.lang.foreign.Linker;
import java.lang.foreign.MemorySegment;
import java.lang.invoke.MethodHandle;

import static java.lang.foreign.ValueLayout.*;

public final class HipCompatOps {
    private HipCompatOps() {}

    private static MethodHandle k(String name, FunctionDescriptor d) {
        return HipOps.downcall(name, d, Linker.Option.critical(true));
    }

    private static final FunctionDescriptor PAIR = FunctionDescriptor.of(JAVA_FLOAT, ADDRESS, JAVA_LONG, ADDRESS, JAVA_LONG, JAVA_LONG);
    private static final MethodHandle COSINE = k("cosine_f32", PAIR);
    private static final MethodHandle DOT = k("dot_product_f32", PAIR);
    private static final MethodHandle EUCLIDEAN = k("euclidean_f32", PAIR);
    private static final MethodHandle ADD = k("add_in_place_f32", FunctionDescriptor.ofVoid(ADDRESS, ADDRESS, JAVA_LONG));
    private static final MethodHandle ADD_SCALAR = k("add_scalar_in_place_f32", FunctionDescriptor.ofVoid(ADDRESS, JAVA_FLOAT, JAVA_LONG));
    private static final MethodHandle SUB = k("sub_in_place_f32", FunctionDescriptor.ofVoid(ADDRESS, ADDRESS, JAVA_LONG));
    private static final MethodHandle SUB_SCALAR = k("sub_scalar_in_place_f32", FunctionDescriptor.ofVoid(ADDRESS, JAVA_FLOAT, JAVA_LONG));
    private static final MethodHandle MAX = k("max_f32", FunctionDescriptor.of(JAVA_FLOAT, ADDRESS, JAVA_LONG));
    private static final MethodHandle MIN_IN_PLACE = k("min_in_place_f32", FunctionDescriptor.ofVoid(ADDRESS, ADDRESS, JAVA_LONG));
    private static final MethodHandle ASSEMBLE = k("assemble_and_sum_f32", FunctionDescriptor.of(JAVA_FLOAT, ADDRESS, JAVA_INT, ADDRESS, JAVA_INT, JAVA_LONG));
    private static final MethodHandle ASSEMBLE_PQ = k("assemble_and_sum_pq_f32", FunctionDescriptor.of(JAVA_FLOAT, ADDRESS, JAVA_LONG, ADDRESS, JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT));
    private static final MethodHandle PQ_COSINE = k("pq_decoded_cosine_similarity_f32", FunctionDescriptor.of(JAVA_FLOAT, ADDRESS, JAVA_INT, JAVA_LONG, JAVA_INT, ADDRESS, ADDRESS, JAVA_FLOAT));
    private static final FunctionDescriptor PARTIAL = FunctionDescriptor.ofVoid(ADDRESS, JAVA_INT, JAVA_LONG, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS);
    private static final MethodHandle PARTIAL_DOT = k("calculate_partial_sums_dot_f32", PARTIAL);
    private static final MethodHandle PARTIAL_L2 = k("calculate_partial_sums_euclidean_f32", PARTIAL);
    private static final MethodHandle PARTIAL_SELF = k("calculate_partial_sums_self_magnitude_f32", FunctionDescriptor.ofVoid(ADDRESS, JAVA_INT, JAVA_LONG, JAVA_INT, ADDRESS));
    private static final MethodHandle NVQ_SHUFFLE = k("nvq_shuffle_query_in_place_8bit", FunctionDescriptor.ofVoid(ADDRESS, JAVA_LONG));
    private static final MethodHandle NVQ_QUANTIZE = k("nvq_quantize_8bit", FunctionDescriptor.ofVoid(ADDRESS, JAVA_LONG, JAVA_FLOAT, JAVA_FLOAT, JAVA_FLOAT, JAVA_FLOAT, ADDRESS));
    private static final MethodHandle NVQ_LOSS = k("nvq_loss", FunctionDescriptor.of(JAVA_FLOAT, ADDRESS, JAVA_LONG, JAVA_FLOAT, JAVA_FLOAT, JAVA_FLOAT, JAVA_FLOAT, JAVA_INT));
    private static final MethodHandle NVQ_UNIFORM_LOSS = k("nvq_uniform_loss", FunctionDescriptor.of(JAVA_FLOAT, ADDRESS, JAVA_LONG, JAVA_FLOAT, JAVA_FLOAT, JAVA_INT));
    private static final FunctionDescriptor NVQ_PAIR = FunctionDescriptor.of(JAVA_FLOAT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_FLOAT, JAVA_FLOAT, JAVA_FLOAT, JAVA_FLOAT);
    private static final MethodHandle NVQ_L2 = k("nvq_square_l2_distance_8bit", NVQ_PAIR);
    private static final MethodHandle NVQ_DOT = k("nvq_dot_product_8bit", NVQ_PAIR);
    private static final MethodHandle NVQ_COSINE = k("nvq_cosine_8bit_packed", FunctionDescriptor.of(JAVA_LONG, ADDRESS, ADDRESS, JAVA_LONG, JAVA_FLOAT, JAVA_FLOAT, JAVA_FLOAT, JAVA_FLOAT, ADDRESS));
    private static final MethodHandle ACTIVE_ISA = k("jvector_simd_get_active_isa", FunctionDescriptor.of(ADDRESS));
    private static final MethodHandle MAX_ISA_ENV = k("jvector_simd_get_max_isa_env", FunctionDescriptor.of(ADDRESS));

    private static AssertionError wrap(Throwable t) { return new AssertionError("should not reach here", t); }

    public static float cosine(MemorySegment a, long ao, MemorySegment b, long bo, long n) {
        try { return (float) COSINE.invokeExact(a, ao, b, bo, n); } catch (Throwable t) { throw wrap(t); }
    }
    public static float dotProduct(MemorySegment a, long ao, MemorySegment b, long bo, long n) {
        try { return (float) DOT.invokeExact(a, ao, b, bo, n); } catch (Throwable t) { throw wrap(t); }
    }
    public static float euclidean(MemorySegment a, long ao, MemorySegment b, long bo, long n) {
        try { return (float) EUCLIDEAN.invokeExact(a, ao, b, bo, n); } catch (Throwable t) { throw wrap(t); }
    }
    public static void addInPlace(MemorySegment a, MemorySegment b, long n) {
        try { ADD.invokeExact(a, b, n); } catch (Throwable t) { throw wrap(t); }
    }
    public static void addScalarInPlace(MemorySegment a, float v, long n) {
        try { ADD_SCALAR.invokeExact(a, v, n); } catch (Throwable t) { throw wrap(t); }
    }
    public static void subInPlace(MemorySegment a, MemorySegment b, long n) {
        try { SUB.invokeExact(a, b, n); } catch (Throwable t) { throw wrap(t); }
    }
    public static void subScalarInPlace(MemorySegment a, float v, long n) {
        try { SUB_SCALAR.invokeExact(a, v, n); } catch (Throwable t) { throw wrap(t); }
    }
    public static float max(MemorySegment a, long n) {
        try { return (float) MAX.invokeExact(a, n); } catch (Throwable t) { throw wrap(t); }
    }
    public static void minInPlace(MemorySegment a, MemorySegment b, long n) {
        try { MIN_IN_PLACE.invokeExact(a, b, n); } catch (Throwable t) { throw wrap(t); }
    }
    public static float assembleAndSum(MemorySegment data, int dataBase, MemorySegment offsets, int offsetsOffset, long n) {
        try { return (float) ASSEMBLE.invokeExact(data, dataBase, offsets, offsetsOffset, n); } catch (Throwable t) { throw wrap(t); }
    }
    public static float assembleAndSumPQ(MemorySegment table, long m, MemorySegment c1, int o1, MemorySegment c2, int o2, int k) {
        try { return (float) ASSEMBLE_PQ.invokeExact(table, m, c1, o1, c2, o2, k); } catch (Throwable t) { throw wrap(t); }
    }
    public static float pqDecodedCosine(MemorySegment enc, int off, long len, int k, MemorySegment sums, MemorySegment aMag, float bMag) {
        try { return (float) PQ_COSINE.invokeExact(enc, off, len, k, sums, aMag, bMag); } catch (Throwable t) { throw wrap(t); }
    }
    public static void partialSumsDot(MemorySegment cb, int idx, long size, int k, MemorySegment q, int qo, MemorySegment out) {
        try { PARTIAL_DOT.invokeExact(cb, idx, size, k, q, qo, out); } catch (Throwable t) { throw wrap(t); }
    }
    public static void partialSumsEuclidean(MemorySegment cb, int idx, long size, int k, MemorySegment q, int qo, MemorySegment out) {
        try { PARTIAL_L2.invokeExact(cb, idx, size, k, q, qo, out); } catch (Throwable t) { throw wrap(t); }
    }
    public static void partialSelfMagnitudes(MemorySegment cb, int idx, long size, int k, MemorySegment out) {
        try { PARTIAL_SELF.invokeExact(cb, idx, size, k, out); } catch (Throwable t) { throw wrap(t); }
    }
    public static void nvqShuffleQueryInPlace8bit(MemorySegment v, long n) {
        try { NVQ_SHUFFLE.invokeExact(v, n); } catch (Throwable t) { throw wrap(t); }
    }
    public static void nvqQuantize8bit(MemorySegment v, long n, float alpha, float x0, float lo, float hi, MemorySegment dst) {
        try { NVQ_QUANTIZE.invokeExact(v, n, alpha, x0, lo, hi, dst); } catch (Throwable t) { throw wrap(t); }
    }
    public static float nvqLoss(MemorySegment v, long n, float alpha, float x0, float lo, float hi, int bits) {
        try { return (float) NVQ_LOSS.invokeExact(v, n, alpha, x0, lo, hi, bits); } catch (Throwable t) { throw wrap(t); }
    }
    public static float nvqUniformLoss(MemorySegment v, long n, float lo, float hi, int bits) {
        try { return (float) NVQ_UNIFORM_LOSS.invokeExact(v, n, lo, hi, bits); } catch (Throwable t) { throw wrap(t); }
    }
    public static float nvqSquareL2Distance8bit(MemorySegment v, MemorySegment q, long n, float alpha, float x0, float lo, float hi) {
        try { return (float) NVQ_L2.invokeExact(v, q, n, alpha, x0, lo, hi); } catch (Throwable t) { throw wrap(t); }
    }
    public static float nvqDotProduct8bit(MemorySegment v, MemorySegment q, long n, float alpha, float x0, float lo, float hi) {
        try { return (float) NVQ_DOT.invokeExact(v, q, n, alpha, x0, lo, hi); } catch (Throwable t) { throw wrap(t); }
    }
    public static long nvqCosine8bitPacked(MemorySegment v, MemorySegment q, long n, float alpha, float x0, float lo, float hi, MemorySegment centroid) {
        try { return (long) NVQ_COSINE.invokeExact(v, q, n, alpha, x0, lo, hi, centroid); } catch (Throwable t) { throw wrap(t); }
    }
    public static String activeIsa() {
        try { return ((MemorySegment) ACTIVE_ISA.invokeExact()).reinterpret(Long.MAX_VALUE).getString(0); } catch (Throwable t) { throw wrap(t); }
    }
    public static String maxIsaEnv() {
        try {
            MemorySegment p = (MemorySegment) MAX_ISA_ENV.invokeExact();
            return p.address() == 0L ? null : p.reinterpret(Long.MAX_VALUE).getString(0);
        } catch (Throwable t) { throw wrap(t); }
    }
}
