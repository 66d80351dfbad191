/*
 * FFM (Java 22) downcall handles for include/jvector_hip.h — hand-written in the shape jextract generates for
 * the reference (jvector-native/.../vector/cnative/NativeSimdOps.java:59-60,1152-1211), with two differences:
 *   - NO Linker.Option.critical(true): batched GPU calls block for milliseconds and must not pin the GC;
 *   - every buffer handed to a jv_hip_* call is an OFF-HEAP MemorySegment (Arena.ofShared / ofConfined).
 * Type map: pointer -> ADDRESS, size_t/int64_t -> JAVA_LONG, int/enum -> JAVA_INT.
 * NOT compiled in this repository (no JDK in the build image).
 */
package io.github.jbellis.jvector.vector.hip;

import java.io.File;
import java.lang.foreign.*;
import java.lang.invoke.MethodHandle;
import java.nio.file.Files;

import static java.lang.foreign.ValueLayout.*;

public final class HipOps {
    private HipOps() {}

    private static final Linker LINKER = Linker.nativeLinker();
    private static SymbolLookup LOOKUP;

    /** Same two-step strategy as LibraryLoader.loadJvector (cnative/LibraryLoader.java:28-55). */
    public static synchronized boolean load() {
        if (LOOKUP != null) return true;
        try {
            System.loadLibrary("jvector_hip");
        } catch (UnsatisfiedLinkError e) {
            try {
                String libName = System.mapLibraryName("jvector_hip");
                File tmp = File.createTempFile("libjvector_hip", ".so");
                try (var in = HipOps.class.getResourceAsStream("/" + libName); var out = Files.newOutputStream(tmp.toPath())) {
                    if (in == null) return false;
                    in.transferTo(out);
                }
                System.load(tmp.getAbsolutePath());
            } catch (Exception | UnsatisfiedLinkError e2) {
                return false;
            }
        }
        LOOKUP = SymbolLookup.loaderLookup().or(LINKER.defaultLookup());
        return true;
    }

    private static MethodHandle h(String name, FunctionDescriptor d) {
        return downcall(name, d);
    }

    /** Downcall handle for a symbol of libjvector_hip.so (HipCompatOps passes Linker.Option.critical(true), like the
     *  reference's sed-patched jextract output, jvector-native/src/main/native/src/jextract_vector_simd.sh). */
    static MethodHandle downcall(String name, FunctionDescriptor d, Linker.Option... options) {
        if (LOOKUP == null && !load()) throw new UnsatisfiedLinkError("libjvector_hip");
        return LINKER.downcallHandle(LOOKUP.find(name).orElseThrow(() -> new UnsatisfiedLinkError(name)), d, options);
    }

    private static final class H {
        static final MethodHandle lastError = h("jv_hip_last_error", FunctionDescriptor.of(ADDRESS));
        static final MethodHandle deviceCount = h("jv_hip_device_count", FunctionDescriptor.of(JAVA_INT));
        static final MethodHandle activeArch = h("jv_hip_active_arch", FunctionDescriptor.of(ADDRESS, JAVA_INT));
        static final MethodHandle ctxCreate = h("jv_hip_ctx_create", FunctionDescriptor.of(JAVA_INT, JAVA_INT, ADDRESS, ADDRESS));
        static final MethodHandle ctxDestroy = h("jv_hip_ctx_destroy", FunctionDescriptor.of(JAVA_INT, ADDRESS));
        static final MethodHandle ctxSync = h("jv_hip_ctx_sync", FunctionDescriptor.of(JAVA_INT, ADDRESS));
        static final MethodHandle pqCreate = h("jv_hip_pq_create", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS));
        static final MethodHandle pqLoad = h("jv_hip_pq_load", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS, ADDRESS));
        static final MethodHandle pqDestroy = h("jv_hip_pq_destroy", FunctionDescriptor.of(JAVA_INT, ADDRESS));
        static final MethodHandle codesCreate = h("jv_hip_codes_create", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS));
        static final MethodHandle codesUpload = h("jv_hip_codes_upload", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, ADDRESS));
        static final MethodHandle codesDestroy = h("jv_hip_codes_destroy", FunctionDescriptor.of(JAVA_INT, ADDRESS));
        static final MethodHandle vectorsCreate = h("jv_hip_vectors_create", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_LONG, JAVA_INT, ADDRESS));
        static final MethodHandle vectorsUpload = h("jv_hip_vectors_upload", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, ADDRESS));
        static final MethodHandle vectorsDestroy = h("jv_hip_vectors_destroy", FunctionDescriptor.of(JAVA_INT, ADDRESS));
        static final MethodHandle pqEncode = h("jv_hip_pq_encode", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS));
        static final MethodHandle lutsCreate = h("jv_hip_luts_create", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS));
        static final MethodHandle lutsBuild = h("jv_hip_luts_build", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT));
        static final MethodHandle lutsDestroy = h("jv_hip_luts_destroy", FunctionDescriptor.of(JAVA_INT, ADDRESS));
        static final MethodHandle adcScores = h("jv_hip_adc_scores", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, ADDRESS));
        static final MethodHandle adcScan = h("jv_hip_adc_scan", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, ADDRESS));
        static final MethodHandle fusedCreate = h("jv_hip_fused_create", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_INT, ADDRESS));
        static final MethodHandle fusedUpload = h("jv_hip_fused_upload", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, ADDRESS, ADDRESS));
        static final MethodHandle fusedDestroy = h("jv_hip_fused_destroy", FunctionDescriptor.of(JAVA_INT, ADDRESS));
        static final MethodHandle fusedBuild = h("jv_hip_fused_build", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, ADDRESS));
        static final MethodHandle fusedScores = h("jv_hip_fused_scores", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS));
        static final MethodHandle exactScores = h("jv_hip_exact_scores", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS));
        static final MethodHandle exactScanDense = h("jv_hip_exact_scan_dense", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, JAVA_LONG, JAVA_LONG, ADDRESS));
        static final MethodHandle topk = h("jv_hip_topk", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, JAVA_LONG, JAVA_LONG, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS));
        static final MethodHandle graphCreate = h("jv_hip_graph_create", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_LONG, JAVA_INT, ADDRESS));
        static final MethodHandle graphSetLevel = h("jv_hip_graph_set_level", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS, JAVA_INT));
        static final MethodHandle graphSetEntry = h("jv_hip_graph_set_entry", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT));
        static final MethodHandle graphDestroy = h("jv_hip_graph_destroy", FunctionDescriptor.of(JAVA_INT, ADDRESS));
        static final MethodHandle graphSearch = h("jv_hip_graph_search", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS, ADDRESS));
        static final MethodHandle graphSetTraversal = h("jv_hip_graph_set_traversal", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_INT));
        // build-time scoring (BuildScoreProvider.pqBuildScoreProvider), PQ training / serialization, anisotropic encode
        static final MethodHandle pairTableCreate = h("jv_hip_pair_table_create", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS));
        static final MethodHandle pairTableDestroy = h("jv_hip_pair_table_destroy", FunctionDescriptor.of(JAVA_INT, ADDRESS));
        static final MethodHandle codePairScores = h("jv_hip_code_pair_scores", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS));
        static final MethodHandle pqDecode = h("jv_hip_pq_decode", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, ADDRESS));
        static final MethodHandle directScores = h("jv_hip_direct_scores", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS));
        static final MethodHandle pqTrain = h("jv_hip_pq_train", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_LONG, ADDRESS));
        static final MethodHandle pqRefine = h("jv_hip_pq_refine", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_LONG, JAVA_INT, JAVA_LONG, ADDRESS));
        static final MethodHandle pqWrite = h("jv_hip_pq_write", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS, JAVA_LONG, ADDRESS));
        static final MethodHandle pqSetAnisotropicThreshold = h("jv_hip_pq_set_anisotropic_threshold", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_FLOAT));
        // host-side readers of JVector's own byte formats (include/jvector_formats.h)
        static final MethodHandle odgiDescribe = h("jv_fmt_odgi_describe", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_LONG, ADDRESS));
        static final MethodHandle odgiReadL0 = h("jv_fmt_odgi_read_l0", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS, ADDRESS));
        static final MethodHandle odgiReadLevel = h("jv_fmt_odgi_read_level", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_LONG, ADDRESS, JAVA_INT, ADDRESS, ADDRESS));
        static final MethodHandle pqvectorsDescribe = h("jv_fmt_pqvectors_describe", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS, ADDRESS));
        static final MethodHandle graphSearchFiltered = h("jv_hip_graph_search_filtered", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS));
        static final MethodHandle exactPairScores = h("jv_hip_exact_pair_scores", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS));
        // GraphSearcher objects: threshold / rerankFloor / resume
        static final MethodHandle searcherCreate = h("jv_hip_searcher_create", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS));
        static final MethodHandle searcherSearch = h("jv_hip_searcher_search", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_FLOAT, JAVA_FLOAT, ADDRESS, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS));
        static final MethodHandle searcherResume = h("jv_hip_searcher_resume", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS));
        static final MethodHandle searcherDestroy = h("jv_hip_searcher_destroy", FunctionDescriptor.of(JAVA_INT, ADDRESS));
        static final MethodHandle searchFlat = h("jv_hip_search_flat", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS));
        // construction: batched robust prune, caller-owned mutable adjacency
        static final MethodHandle retainDiverse = h("jv_hip_retain_diverse", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, JAVA_FLOAT, ADDRESS, ADDRESS, ADDRESS));
        static final MethodHandle deviceAlloc = h("jv_hip_device_alloc", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_LONG, ADDRESS));
        static final MethodHandle deviceFree = h("jv_hip_device_free", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS));
        static final MethodHandle graphSetLevel0Device = h("jv_hip_graph_set_level0_device", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_INT));
        // sharded index: one rank (thread + context) per GPU, RCCL inside the library
        static final MethodHandle commUniqueId = h("jv_hip_comm_unique_id", FunctionDescriptor.of(JAVA_INT, ADDRESS));
        static final MethodHandle commCreate = h("jv_hip_comm_create", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS));
        static final MethodHandle commDestroy = h("jv_hip_comm_destroy", FunctionDescriptor.of(JAVA_INT, ADDRESS));
        static final MethodHandle shardedTopk = h("jv_hip_sharded_topk", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS));
        static final MethodHandle commCount = h("jv_hip_comm_count", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS));
        static final MethodHandle commAllGather = h("jv_hip_comm_all_gather", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS));
        // batched Vamana construction of one level (GraphIndexBuilder.addGraphNode / cleanup as batch calls)
        static final MethodHandle builderCreate = h("jv_hip_builder_create", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_FLOAT, JAVA_FLOAT, ADDRESS));
        static final MethodHandle builderSeed = h("jv_hip_builder_seed", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT));
        static final MethodHandle builderInsertBatch = h("jv_hip_builder_insert_batch", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_INT));
        static final MethodHandle builderImproveBatch = h("jv_hip_builder_improve_batch", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_INT));
        static final MethodHandle builderFinish = h("jv_hip_builder_finish", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS));
        static final MethodHandle builderStats = h("jv_hip_builder_stats", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS));
        static final MethodHandle builderNeighborsDevice = h("jv_hip_builder_neighbors_device", FunctionDescriptor.of(ADDRESS, ADDRESS, ADDRESS));
        static final MethodHandle builderWorkingLists = h("jv_hip_builder_working_lists", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS));
        static final MethodHandle builderDestroy = h("jv_hip_builder_destroy", FunctionDescriptor.of(JAVA_INT, ADDRESS));
        static final MethodHandle buildLayered = h("jv_hip_build_layered", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_FLOAT, JAVA_FLOAT, JAVA_INT, JAVA_INT, JAVA_LONG, JAVA_INT, ADDRESS));
        static final MethodHandle layeredInfo = h("jv_hip_layered_info", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS));
        static final MethodHandle layeredLevel = h("jv_hip_layered_level", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS, ADDRESS));
        static final MethodHandle layeredLevel0Device = h("jv_hip_layered_level0_device", FunctionDescriptor.of(ADDRESS, ADDRESS));
        static final MethodHandle layeredStats = h("jv_hip_layered_stats", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS));
        static final MethodHandle layeredDestroy = h("jv_hip_layered_destroy", FunctionDescriptor.of(JAVA_INT, ADDRESS));
        // per-context options (a JVM cannot set the JVECTOR_HIP_* environment per context) and counters
        static final MethodHandle ctxSetOption = h("jv_hip_ctx_set_option", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG));
        static final MethodHandle ctxClearOption = h("jv_hip_ctx_clear_option", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS));
        static final MethodHandle ctxGetStat = h("jv_hip_ctx_get_stat", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS));
        static final MethodHandle ctxResetStats = h("jv_hip_ctx_reset_stats", FunctionDescriptor.of(JAVA_INT, ADDRESS));
        static final MethodHandle vectorsInvalidate = h("jv_hip_vectors_invalidate", FunctionDescriptor.of(JAVA_INT, ADDRESS));
        // NVQ: NVQuantization / NVQVectors / NVQScorer and the graph's NVQ reranker
        static final MethodHandle nvqCreate = h("jv_hip_nvq_create", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS));
        static final MethodHandle nvqCompute = h("jv_hip_nvq_compute", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS));
        static final MethodHandle nvqSetLearn = h("jv_hip_nvq_set_learn", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_INT));
        static final MethodHandle nvqGlobalMean = h("jv_hip_nvq_global_mean", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS));
        static final MethodHandle nvqDestroy = h("jv_hip_nvq_destroy", FunctionDescriptor.of(JAVA_INT, ADDRESS));
        static final MethodHandle nvqVectorsCreate = h("jv_hip_nvq_vectors_create", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS));
        static final MethodHandle nvqEncode = h("jv_hip_nvq_encode", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, ADDRESS, JAVA_LONG));
        static final MethodHandle nvqVectorsUpload = h("jv_hip_nvq_vectors_upload", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, ADDRESS, ADDRESS));
        static final MethodHandle nvqVectorsDownload = h("jv_hip_nvq_vectors_download", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, ADDRESS, ADDRESS));
        static final MethodHandle nvqVectorsDestroy = h("jv_hip_nvq_vectors_destroy", FunctionDescriptor.of(JAVA_INT, ADDRESS));
        static final MethodHandle nvqScores = h("jv_hip_nvq_scores", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS));
        static final MethodHandle vectorsFromNvq = h("jv_hip_vectors_from_nvq", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS));
        static final MethodHandle commCreateExternal = h("jv_hip_comm_create_external", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS, ADDRESS));
        static final MethodHandle shardedMergeRerank = h("jv_hip_sharded_merge_rerank", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS));
        static final MethodHandle shardedSearchFlat = h("jv_hip_sharded_search_flat", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS));
    }

    /** jv_status -> Java exception, mirroring the reference's exception types (include/jvector_hip.h:38-45). */
    static void check(int status) {
        if (status == 0) return;
        String msg = lastError();
        switch (status) {
            case -1: throw new IllegalArgumentException(msg);
            case -2: throw new UnsupportedOperationException(msg);   // no device: provider lookup falls back
            case -4: throw new OutOfMemoryError(msg);
            case -5: throw new UnsupportedOperationException(msg);
            default: throw new IllegalStateException(msg);
        }
    }

    public static String lastError() {
        try {
            return ((MemorySegment) H.lastError.invokeExact()).reinterpret(Long.MAX_VALUE).getString(0);
        } catch (Throwable t) { throw new AssertionError(t); }
    }

    public static int deviceCount() {
        try { return (int) H.deviceCount.invokeExact(); } catch (Throwable t) { throw new AssertionError(t); }
    }

    public static String activeArch(int device) {
        try {
            return ((MemorySegment) H.activeArch.invokeExact(device)).reinterpret(Long.MAX_VALUE).getString(0);
        } catch (Throwable t) { throw new AssertionError(t); }
    }

    // --- thin typed wrappers (only the ones HipBatchScorer needs are spelled out; the rest follow the same shape) ---
    public static MemorySegment ctxCreate(Arena arena, int device) {
        MemorySegment out = arena.allocate(ADDRESS);
        try { check((int) H.ctxCreate.invokeExact(device, MemorySegment.ofAddress(-1L) /* JV_STREAM_PRIVATE */, out)); }
        catch (RuntimeException | Error e) { throw e; } catch (Throwable t) { throw new AssertionError(t); }
        return out.get(ADDRESS, 0);
    }

    public static void ctxDestroy(MemorySegment ctx) {
        try { check((int) H.ctxDestroy.invokeExact(ctx)); } catch (RuntimeException | Error e) { throw e; } catch (Throwable t) { throw new AssertionError(t); }
    }

    public static MemorySegment pqLoad(Arena arena, MemorySegment ctx, MemorySegment bytes) {
        MemorySegment out = arena.allocate(ADDRESS);
        try { check((int) H.pqLoad.invokeExact(ctx, bytes, bytes.byteSize(), MemorySegment.NULL, out)); }
        catch (RuntimeException | Error e) { throw e; } catch (Throwable t) { throw new AssertionError(t); }
        return out.get(ADDRESS, 0);
    }

    public static MemorySegment codesCreate(Arena arena, MemorySegment ctx, MemorySegment pq, long count) {
        MemorySegment out = arena.allocate(ADDRESS);
        try { check((int) H.codesCreate.invokeExact(ctx, pq, count, out)); }
        catch (RuntimeException | Error e) { throw e; } catch (Throwable t) { throw new AssertionError(t); }
        return out.get(ADDRESS, 0);
    }

    public static void codesUpload(MemorySegment ctx, MemorySegment codes, long first, long count, MemorySegment src) {
        try { check((int) H.codesUpload.invokeExact(ctx, codes, first, count, src)); }
        catch (RuntimeException | Error e) { throw e; } catch (Throwable t) { throw new AssertionError(t); }
    }

    public static MemorySegment lutsCreate(Arena arena, MemorySegment ctx, MemorySegment pq, int maxQueries) {
        MemorySegment out = arena.allocate(ADDRESS);
        try { check((int) H.lutsCreate.invokeExact(ctx, pq, maxQueries, out)); }
        catch (RuntimeException | Error e) { throw e; } catch (Throwable t) { throw new AssertionError(t); }
        return out.get(ADDRESS, 0);
    }

    public static void lutsBuild(MemorySegment ctx, MemorySegment luts, MemorySegment queries, int q, int vsf, int kind) {
        try { check((int) H.lutsBuild.invokeExact(ctx, luts, queries, q, vsf, kind)); }
        catch (RuntimeException | Error e) { throw e; } catch (Throwable t) { throw new AssertionError(t); }
    }

    public static void adcScores(MemorySegment ctx, MemorySegment luts, MemorySegment codes, MemorySegment ordinals, int b, MemorySegment out) {
        try { check((int) H.adcScores.invokeExact(ctx, luts, codes, ordinals, b, out)); }
        catch (RuntimeException | Error e) { throw e; } catch (Throwable t) { throw new AssertionError(t); }
    }

    public static void fusedScores(MemorySegment ctx, MemorySegment luts, MemorySegment fused, MemorySegment origins, MemorySegment out, MemorySegment neighborsOut) {
        try { check((int) H.fusedScores.invokeExact(ctx, luts, fused, origins, out, neighborsOut)); }
        catch (RuntimeException | Error e) { throw e; } catch (Throwable t) { throw new AssertionError(t); }
    }

    /** GraphSearcher.search for a whole batch (host traversal inside the library, GPU frontier scoring). */
    public static void graphSearch(MemorySegment ctx, MemorySegment graph, MemorySegment luts, MemorySegment codes, MemorySegment fusedOrNull,
                                   MemorySegment vectorsOrNull, MemorySegment queries, int q, int vsf, int topK, int rerankK,
                                   MemorySegment outIds, MemorySegment outScores, MemorySegment statsOrNull) {
        try { check((int) H.graphSearch.invokeExact(ctx, graph, luts, codes, fusedOrNull, vectorsOrNull, queries, q, vsf, topK, rerankK, outIds, outScores, statsOrNull)); }
        catch (RuntimeException | Error e) { throw e; } catch (Throwable t) { throw new AssertionError(t); }
    }

    public static void exactScores(MemorySegment ctx, MemorySegment vectors, MemorySegment queries, int q, int vsf, MemorySegment ordinals, int b, MemorySegment out) {
        try { check((int) H.exactScores.invokeExact(ctx, vectors, queries, q, vsf, ordinals, b, out)); }
        catch (RuntimeException | Error e) { throw e; } catch (Throwable t) { throw new AssertionError(t); }
    }

    /** MFMA tile form of the brute-force scan (ground truth / candidate generation over many queries): fused k-ascending
     *  chains, within 1e-5 of {@link #exactScores}' scalar-order arithmetic; out is [q][count] row-major. */
    public static void exactScanDense(MemorySegment ctx, MemorySegment vectors, MemorySegment queries, int q, int vsf, long first, long count, MemorySegment out) {
        try { check((int) H.exactScanDense.invokeExact(ctx, vectors, queries, q, vsf, first, count, out)); }
        catch (RuntimeException | Error e) { throw e; } catch (Throwable t) { throw new AssertionError(t); }
    }

    // ---- the remaining entry points, same shape: status -> exception, out-pointers through a caller arena ----
    private static MemorySegment outHandle(Arena arena, java.util.function.ToIntFunction<MemorySegment> call) {
        MemorySegment out = arena.allocate(ADDRESS);
        check(call.applyAsInt(out));
        return out.get(ADDRESS, 0);
    }
    private interface Call { int run() throws Throwable; }
    private static int st(Call c) {
        try { return c.run(); } catch (RuntimeException | Error e) { throw e; } catch (Throwable t) { throw new AssertionError(t); }
    }

    public static void ctxSync(MemorySegment ctx) { check(st(() -> (int) H.ctxSync.invokeExact(ctx))); }
    public static void pqDestroy(MemorySegment pq) { check(st(() -> (int) H.pqDestroy.invokeExact(pq))); }
    public static void codesDestroy(MemorySegment c) { check(st(() -> (int) H.codesDestroy.invokeExact(c))); }
    public static void lutsDestroy(MemorySegment l) { check(st(() -> (int) H.lutsDestroy.invokeExact(l))); }

    public static MemorySegment vectorsCreate(Arena arena, MemorySegment ctx, long count, int dim) {
        return outHandle(arena, out -> st(() -> (int) H.vectorsCreate.invokeExact(ctx, count, dim, out)));
    }
    public static void vectorsUpload(MemorySegment ctx, MemorySegment vectors, long first, long count, MemorySegment src) {
        check(st(() -> (int) H.vectorsUpload.invokeExact(ctx, vectors, first, count, src)));
    }
    public static void vectorsDestroy(MemorySegment v) { check(st(() -> (int) H.vectorsDestroy.invokeExact(v))); }

    /** ProductQuantization.encodeAll (PQVectors.encodeAndBuild, PQVectors.java:109-152): count x D floats -> count x M code bytes. */
    public static void pqEncode(MemorySegment ctx, MemorySegment pq, MemorySegment vectors, long count, MemorySegment codesOut) {
        check(st(() -> (int) H.pqEncode.invokeExact(ctx, pq, vectors, count, codesOut)));
    }

    public static MemorySegment fusedCreate(Arena arena, MemorySegment ctx, MemorySegment pq, long count, int maxDegree) {
        return outHandle(arena, out -> st(() -> (int) H.fusedCreate.invokeExact(ctx, pq, count, maxDegree, out)));
    }
    public static void fusedUpload(MemorySegment ctx, MemorySegment fused, long first, long count, MemorySegment blocks, MemorySegment neighbors) {
        check(st(() -> (int) H.fusedUpload.invokeExact(ctx, fused, first, count, blocks, neighbors)));
    }
    /** FusedPQ.writeInline on the device: blocks gathered from the code store for the given neighbour rows. */
    public static void fusedBuild(MemorySegment ctx, MemorySegment fused, MemorySegment codes, long first, long count, MemorySegment neighbors) {
        check(st(() -> (int) H.fusedBuild.invokeExact(ctx, fused, codes, first, count, neighbors)));
    }

    public static void fusedDestroy(MemorySegment fused) { check(st(() -> (int) H.fusedDestroy.invokeExact(fused))); }

    /** NodeQueue-order top-k (NodeQueue.java:125-129): ids may be NULL (id = idBase + column). */
    public static void topk(MemorySegment ctx, MemorySegment scores, MemorySegment idsOrNull, int q, long n, long stride, int idBase, int k,
                            MemorySegment outIds, MemorySegment outScores) {
        check(st(() -> (int) H.topk.invokeExact(ctx, scores, idsOrNull, q, n, stride, idBase, k, outIds, outScores)));
    }

    public static MemorySegment graphCreate(Arena arena, MemorySegment ctx, long nNodes, int nLevels) {
        return outHandle(arena, out -> st(() -> (int) H.graphCreate.invokeExact(ctx, nNodes, nLevels, out)));
    }
    public static void graphSetLevel(MemorySegment ctx, MemorySegment graph, int level, int count, MemorySegment nodeIdsOrNull, MemorySegment neighbors, int degree) {
        check(st(() -> (int) H.graphSetLevel.invokeExact(ctx, graph, level, count, nodeIdsOrNull, neighbors, degree)));
    }
    public static void graphSetEntry(MemorySegment graph, int node, int level) { check(st(() -> (int) H.graphSetEntry.invokeExact(graph, node, level))); }
    public static void graphSetTraversal(MemorySegment graph, int mode) { check(st(() -> (int) H.graphSetTraversal.invokeExact(graph, mode))); }
    public static void graphDestroy(MemorySegment graph) { check(st(() -> (int) H.graphDestroy.invokeExact(graph))); }

    /** GraphSearcher.search(scoreProvider, topK, rerankK, threshold = 0, rerankFloor = 0, acceptOrds) for a batch. */
    public static void graphSearchFiltered(MemorySegment ctx, MemorySegment graph, MemorySegment luts, MemorySegment codes, MemorySegment fusedOrNull,
                                           MemorySegment vectorsOrNull, MemorySegment queries, int q, int vsf, int topK, int rerankK,
                                           MemorySegment acceptBitsOrNull, long acceptStrideWords, MemorySegment outIds, MemorySegment outScores,
                                           MemorySegment statsOrNull) {
        check(st(() -> (int) H.graphSearchFiltered.invokeExact(ctx, graph, luts, codes, fusedOrNull, vectorsOrNull, queries, q, vsf, topK, rerankK,
                                                               acceptBitsOrNull, acceptStrideWords, outIds, outScores, statsOrNull)));
    }

    /** Q GraphSearcher objects over one graph: search with every option, then resume (GraphSearcher.java:222-243,538-547). */
    public static MemorySegment searcherCreate(Arena arena, MemorySegment ctx, MemorySegment graph, MemorySegment luts, MemorySegment codes,
                                               MemorySegment fusedOrNull, MemorySegment vectorsOrNull) {
        return outHandle(arena, out -> st(() -> (int) H.searcherCreate.invokeExact(ctx, graph, luts, codes, fusedOrNull, vectorsOrNull, out)));
    }
    /** search(scoreProvider, topK, rerankK, threshold, rerankFloor, acceptOrds) per query; outCounts Q, stats Q x 4
     *  {visitedCount, expandedCount, expandedCountBaseLayer, rerankedCount}, worstApprox Q (all nullable). */
    public static void searcherSearch(MemorySegment ctx, MemorySegment searcher, MemorySegment queries, int q, int vsf, int topK, int rerankK,
                                      float threshold, float rerankFloor, MemorySegment acceptBitsOrNull, long acceptStrideWords,
                                      MemorySegment outIds, MemorySegment outScores, MemorySegment outCountsOrNull, MemorySegment statsOrNull,
                                      MemorySegment worstApproxOrNull) {
        check(st(() -> (int) H.searcherSearch.invokeExact(ctx, searcher, queries, q, vsf, topK, rerankK, threshold, rerankFloor, acceptBitsOrNull,
                                                          acceptStrideWords, outIds, outScores, outCountsOrNull, statsOrNull, worstApproxOrNull)));
    }
    /** resume(additionalK, rerankK) for every query of the last searcherSearch */
    public static void searcherResume(MemorySegment ctx, MemorySegment searcher, int additionalK, int rerankK, MemorySegment outIds,
                                      MemorySegment outScores, MemorySegment outCountsOrNull, MemorySegment statsOrNull,
                                      MemorySegment worstApproxOrNull) {
        check(st(() -> (int) H.searcherResume.invokeExact(ctx, searcher, additionalK, rerankK, outIds, outScores, outCountsOrNull, statsOrNull,
                                                          worstApproxOrNull)));
    }
    public static void searcherDestroy(MemorySegment searcher) { check(st(() -> (int) H.searcherDestroy.invokeExact(searcher))); }

    /** two-pass flat search over one shard (jv_hip_search_flat) */
    public static void searchFlat(MemorySegment ctx, MemorySegment luts, MemorySegment codes, MemorySegment vectorsOrNull, MemorySegment queries,
                                  int q, int vsf, int topK, int rerankK, int idBase, MemorySegment outIds, MemorySegment outScores) {
        check(st(() -> (int) H.searchFlat.invokeExact(ctx, luts, codes, vectorsOrNull, queries, q, vsf, topK, rerankK, idBase, outIds, outScores)));
    }

    // build-time scoring (BuildScoreProvider.pqBuildScoreProvider, BuildScoreProvider.java:167-212)
    public static MemorySegment pairTableCreate(Arena arena, MemorySegment ctx, MemorySegment pq, int vsf) {
        return outHandle(arena, out -> st(() -> (int) H.pairTableCreate.invokeExact(ctx, pq, vsf, out)));
    }
    public static void pairTableDestroy(MemorySegment t) { check(st(() -> (int) H.pairTableDestroy.invokeExact(t))); }
    /** diversityFunctionFor(node1).similarityTo(node2) for P x B candidate x selected blocks */
    public static void codePairScores(MemorySegment ctx, MemorySegment table, MemorySegment codes, MemorySegment node1, int p, MemorySegment node2, int b, MemorySegment out) {
        check(st(() -> (int) H.codePairScores.invokeExact(ctx, table, codes, node1, p, node2, b, out)));
    }

    /** randomAccessScoreProvider.diversityScoreFunctionFor(node1).similarityTo(node2) (BuildScoreProvider.java:151-157), P x B blocks */
    public static void exactPairScores(MemorySegment ctx, MemorySegment vectors, int vsf, MemorySegment node1, int p, MemorySegment node2, int b,
                                       MemorySegment out) {
        check(st(() -> (int) H.exactPairScores.invokeExact(ctx, vectors, vsf, node1, p, node2, b, out)));
    }

    /** VamanaDiversityProvider.retainDiverse for P NodeArrays at once (candidates sorted by score descending per row);
     *  selectedOut: P x maxDegree candidate indices ascending, -1 padded — the set bits of the reference's BitSet. */
    public static void retainDiverse(MemorySegment ctx, MemorySegment table, MemorySegment codes, int p, int c, MemorySegment candNodes,
                                     MemorySegment candScores, MemorySegment candCountOrNull, MemorySegment diverseBeforeOrNull, int maxDegree,
                                     float alpha, MemorySegment selectedOut, MemorySegment nSelectedOut, MemorySegment shortEdgesOutOrNull) {
        check(st(() -> (int) H.retainDiverse.invokeExact(ctx, table, codes, p, c, candNodes, candScores, candCountOrNull, diverseBeforeOrNull,
                                                         maxDegree, alpha, selectedOut, nSelectedOut, shortEdgesOutOrNull)));
    }
    // ---- NVQ: NVQuantization.compute / encodeAll (NVQuantization.java:153-216), NVQScorer (NVQScorer.java:33-137) and
    //      NVQ.rerankerFor (graph/disk/feature/NVQ.java:96-110) as batch calls ----
    /** NVQuantization.create(globalMean, nSubVectors) */
    public static MemorySegment nvqCreate(Arena arena, MemorySegment ctx, int dimension, int nSubVectors, MemorySegment globalMean) {
        return outHandle(arena, out -> st(() -> (int) H.nvqCreate.invokeExact(ctx, dimension, nSubVectors, globalMean, out)));
    }
    /** NVQuantization.compute(ravv, nSubVectors): the mean is accumulated over the device-resident rows in order */
    public static MemorySegment nvqCompute(Arena arena, MemorySegment ctx, MemorySegment vectors, int nSubVectors) {
        return outHandle(arena, out -> st(() -> (int) H.nvqCompute.invokeExact(ctx, vectors, nSubVectors, out)));
    }
    public static void nvqSetLearn(MemorySegment nvq, boolean learn) { check(st(() -> (int) H.nvqSetLearn.invokeExact(nvq, learn ? 1 : 0))); }
    public static void nvqGlobalMean(MemorySegment ctx, MemorySegment nvq, MemorySegment dst) { check(st(() -> (int) H.nvqGlobalMean.invokeExact(ctx, nvq, dst))); }
    public static void nvqDestroy(MemorySegment nvq) { check(st(() -> (int) H.nvqDestroy.invokeExact(nvq))); }
    public static MemorySegment nvqVectorsCreate(Arena arena, MemorySegment ctx, MemorySegment nvq, long count) {
        return outHandle(arena, out -> st(() -> (int) H.nvqVectorsCreate.invokeExact(ctx, nvq, count, out)));
    }
    /** encodeAll: rows [first, first + count) of `vectors` into rows [dstFirst, ...) of `rows` */
    public static void nvqEncode(MemorySegment ctx, MemorySegment nvq, MemorySegment vectors, long first, long count, MemorySegment rows, long dstFirst) {
        check(st(() -> (int) H.nvqEncode.invokeExact(ctx, nvq, vectors, first, count, rows, dstFirst)));
    }
    /** bytes: count x D; params: count x S x {minValue, maxValue, growthRate, midpoint} (QuantizedSubVector.write order) */
    public static void nvqVectorsUpload(MemorySegment ctx, MemorySegment rows, long first, long count, MemorySegment bytes, MemorySegment params) {
        check(st(() -> (int) H.nvqVectorsUpload.invokeExact(ctx, rows, first, count, bytes, params)));
    }
    public static void nvqVectorsDownload(MemorySegment ctx, MemorySegment rows, long first, long count, MemorySegment bytes, MemorySegment params) {
        check(st(() -> (int) H.nvqVectorsDownload.invokeExact(ctx, rows, first, count, bytes, params)));
    }
    public static void nvqVectorsDestroy(MemorySegment rows) { check(st(() -> (int) H.nvqVectorsDestroy.invokeExact(rows))); }
    /** scores[q][b] = NVQVectors.scoreFunctionFor(query q, vsf).similarityTo(ordinals[q*B + b]) */
    public static void nvqScores(MemorySegment ctx, MemorySegment rows, MemorySegment queries, int q, int vsf, MemorySegment ordinals, int b, MemorySegment scores) {
        check(st(() -> (int) H.nvqScores.invokeExact(ctx, rows, queries, q, vsf, ordinals, b, scores)));
    }
    /** a jv_vectors whose rerank scores the NVQ rows: pass it wherever a search takes `vectors` */
    public static MemorySegment vectorsFromNvq(Arena arena, MemorySegment ctx, MemorySegment rows) {
        return outHandle(arena, out -> st(() -> (int) H.vectorsFromNvq.invokeExact(ctx, rows, out)));
    }
    // ---- batched construction: the GraphIndexBuilder.addGraphNode loop (GraphIndexBuilder.java:605-659) as batch calls ----
    /** one graph level over the nodes of `codes` / `vectors`; alpha / neighborOverflow as GraphIndexBuilder's constructor takes them */
    public static MemorySegment builderCreate(Arena arena, MemorySegment ctx, MemorySegment pq, MemorySegment codes, MemorySegment vectors, int vsf,
                                              int maxDegree, int beamWidth, float alpha, float neighborOverflow) {
        return outHandle(arena, out -> st(() -> (int) H.builderCreate.invokeExact(ctx, pq, codes, vectors, vsf, maxDegree, beamWidth, alpha, neighborOverflow, out)));
    }
    public static void builderSeed(MemorySegment ctx, MemorySegment builder, int node) { check(st(() -> (int) H.builderSeed.invokeExact(ctx, builder, node))); }
    /** B concurrent inserts that do not see each other (nodes: int32 ordinals, off-heap or device memory); callers grow the batch with the graph */
    public static void builderInsertBatch(MemorySegment ctx, MemorySegment builder, MemorySegment nodes, int b) {
        check(st(() -> (int) H.builderInsertBatch.invokeExact(ctx, builder, nodes, b)));
    }
    /** improveConnections for nodes already in the graph: search, merge with the node's neighbours, robust prune, backlink */
    public static void builderImproveBatch(MemorySegment ctx, MemorySegment builder, MemorySegment nodes, int b) {
        check(st(() -> (int) H.builderImproveBatch.invokeExact(ctx, builder, nodes, b)));
    }
    /** cleanup(): enforceDegree on every list; neighborsOutOrNull receives count x maxDegree int32, rows packed, -1 padded */
    public static void builderFinish(MemorySegment ctx, MemorySegment builder, MemorySegment neighborsOutOrNull) {
        check(st(() -> (int) H.builderFinish.invokeExact(ctx, builder, neighborsOutOrNull)));
    }
    /** seconds3 = {search, prune, backlink}; counts5 = {batches, re-pruned lists, inserted, visitedCount, expandedCount} */
    public static void builderStats(MemorySegment builder, MemorySegment seconds3, MemorySegment counts5) {
        check(st(() -> (int) H.builderStats.invokeExact(builder, seconds3, counts5)));
    }
    public static MemorySegment builderNeighborsDevice(MemorySegment builder, MemorySegment rowWidthOutOrNull) {
        try { return (MemorySegment) H.builderNeighborsDevice.invokeExact(builder, rowWidthOutOrNull); } catch (Throwable t) { throw new AssertionError(t); }
    }
    /** the lists as they stand: idsOut int32[n x rowWidth] (-1 padded); with ctxSetOption("bl_ref_order", 1) before builderCreate also
     *  scoresOut float[n x rowWidth] (the score every entry was inserted under, NodeArray order) and diverseBeforeOut int32[n] —
     *  ConcurrentNeighborMap.Neighbors' state, for a comparison against OnHeapGraphIndex.getNeighbors of a one-thread GraphIndexBuilder */
    public static void builderWorkingLists(MemorySegment ctx, MemorySegment builder, MemorySegment idsOut, MemorySegment scoresOutOrNull,
                                           MemorySegment diverseBeforeOutOrNull) {
        check(st(() -> (int) H.builderWorkingLists.invokeExact(ctx, builder, idsOut, scoresOutOrNull, diverseBeforeOutOrNull)));
    }
    public static void builderDestroy(MemorySegment builder) { check(st(() -> (int) H.builderDestroy.invokeExact(builder))); }
    /** GraphIndexBuilder with addHierarchy in one call: seeded level draws, one Vamana graph per level, improveConnections passes, enforceDegree; outHandle receives the jv_layered */
    public static void buildLayered(MemorySegment ctx, MemorySegment pq, MemorySegment codes, MemorySegment vectors, int vsf, int maxDegree, int beamWidth,
                                    float alpha, float neighborOverflow, int maxBatch, int improvePasses, long seed, int minTop, MemorySegment outHandle) {
        check(st(() -> (int) H.buildLayered.invokeExact(ctx, pq, codes, vectors, vsf, maxDegree, beamWidth, alpha, neighborOverflow, maxBatch, improvePasses, seed, minTop, outHandle)));
    }
    /** nLevels / entryNode / entryLevel: one int each; levelCounts: int64[nLevels] (any may be NULL) */
    public static void layeredInfo(MemorySegment layered, MemorySegment nLevels, MemorySegment entryNode, MemorySegment entryLevel, MemorySegment levelCounts) {
        check(st(() -> (int) H.layeredInfo.invokeExact(layered, nLevels, entryNode, entryLevel, levelCounts)));
    }
    /** level >= 1: nodesOut int32[count] + neighborsOut int32[count x maxDegree] (off-heap); level 0: nodesOut NULL, neighborsOut off-heap or device */
    public static void layeredLevel(MemorySegment ctx, MemorySegment layered, int level, MemorySegment nodesOut, MemorySegment neighborsOut) {
        check(st(() -> (int) H.layeredLevel.invokeExact(ctx, layered, level, nodesOut, neighborsOut)));
    }
    /** the level-0 rows in device memory (for fusedBuild) */
    public static MemorySegment layeredLevel0Device(MemorySegment layered) {
        try { return (MemorySegment) H.layeredLevel0Device.invokeExact(layered); } catch (Throwable t) { throw new AssertionError(t); }
    }
    /** seconds4 = {search, prune, backlink, total}; counts5 as builderStats, summed over the levels */
    public static void layeredStats(MemorySegment layered, MemorySegment seconds4, MemorySegment counts5) {
        check(st(() -> (int) H.layeredStats.invokeExact(layered, seconds4, counts5)));
    }
    public static void layeredDestroy(MemorySegment layered) { check(st(() -> (int) H.layeredDestroy.invokeExact(layered))); }

    // ---- per-context options / counters, communicator introspection ----
    public static void ctxSetOption(Arena arena, MemorySegment ctx, String name, long value) {
        MemorySegment n = arena.allocateFrom(name);
        check(st(() -> (int) H.ctxSetOption.invokeExact(ctx, n, value)));
    }
    public static void ctxClearOption(Arena arena, MemorySegment ctx, String name) {
        MemorySegment n = arena.allocateFrom(name);
        check(st(() -> (int) H.ctxClearOption.invokeExact(ctx, n)));
    }
    /** e.g. "gs_calls_host_auto": how often JV_TRAVERSAL_AUTO fell back to the (13x slower) host searcher on this context */
    public static long ctxGetStat(Arena arena, MemorySegment ctx, String name) {
        MemorySegment n = arena.allocateFrom(name), out = arena.allocate(JAVA_LONG);
        check(st(() -> (int) H.ctxGetStat.invokeExact(ctx, n, out)));
        return out.get(JAVA_LONG, 0);
    }
    public static void ctxResetStats(MemorySegment ctx) { check(st(() -> (int) H.ctxResetStats.invokeExact(ctx))); }
    /** wrapped vectors edited in place: drop the cached cosine norms */
    public static void vectorsInvalidate(MemorySegment vectors) { check(st(() -> (int) H.vectorsInvalidate.invokeExact(vectors))); }
    /** ncclCommCount of the communicator's RCCL object (1 for a local communicator) */
    public static int commCount(Arena arena, MemorySegment comm) {
        MemorySegment out = arena.allocate(JAVA_INT);
        check(st(() -> (int) H.commCount.invokeExact(comm, out)));
        return out.get(JAVA_INT, 0);
    }
    /** all-gather of `bytes` opaque bytes per rank (small host records); recv holds world x bytes, rank-major */
    public static void commAllGather(MemorySegment ctx, MemorySegment comm, MemorySegment send, long bytes, MemorySegment recv) {
        check(st(() -> (int) H.commAllGather.invokeExact(ctx, comm, send, bytes, recv)));
    }

    /** raw device memory (e.g. the mutable adjacency of graphSetLevel0Device); a zero-length segment carrying the device address */
    public static MemorySegment deviceAlloc(Arena arena, MemorySegment ctx, long bytes) {
        return outHandle(arena, out -> st(() -> (int) H.deviceAlloc.invokeExact(ctx, bytes, out)));
    }
    public static void deviceFree(MemorySegment ctx, MemorySegment ptr) { check(st(() -> (int) H.deviceFree.invokeExact(ctx, ptr))); }
    /** level 0 read in place from caller-owned device memory: the graph a builder is still writing can be searched */
    public static void graphSetLevel0Device(MemorySegment ctx, MemorySegment graph, MemorySegment deviceNeighbors, int degree) {
        check(st(() -> (int) H.graphSetLevel0Device.invokeExact(ctx, graph, deviceNeighbors, degree)));
    }

    // sharded index
    public static MemorySegment commUniqueId(Arena arena) {
        MemorySegment id = arena.allocate(128);
        check(st(() -> (int) H.commUniqueId.invokeExact(id)));
        return id;
    }
    public static MemorySegment commCreate(Arena arena, MemorySegment ctx, MemorySegment idOrNull, int rank, int world) {
        return outHandle(arena, out -> st(() -> (int) H.commCreate.invokeExact(ctx, idOrNull, rank, world, out)));
    }
    /** a communicator over the HOST's transport: allGatherFn = upcall stub int (void* user, void* send, size_t bytes, void* recv) filling recv with every rank's bytes (a JVM with one thread per GPU exchanges through its own memory) */
    public static MemorySegment commCreateExternal(Arena arena, MemorySegment ctx, int rank, int world, MemorySegment allGatherFn, MemorySegment user) {
        return outHandle(arena, out -> st(() -> (int) H.commCreateExternal.invokeExact(ctx, rank, world, allGatherFn, user, out)));
    }
    /** the sharded exchange for partial lists the caller produced (one graph index per shard): partIds / partScores [nLocal][q][rerankK], GLOBAL ids */
    public static void shardedMergeRerank(MemorySegment ctx, MemorySegment comm, int nLocal, MemorySegment luts, MemorySegment vectorsArrayOrNull,
                                          MemorySegment idBases, MemorySegment counts, MemorySegment queries, int q, int vsf, int topK, int rerankK,
                                          MemorySegment partIds, MemorySegment partScores, MemorySegment outIds, MemorySegment outScores) {
        check(st(() -> (int) H.shardedMergeRerank.invokeExact(ctx, comm, nLocal, luts, vectorsArrayOrNull, idBases, counts, queries, q, vsf, topK, rerankK,
                                                              partIds, partScores, outIds, outScores)));
    }
    public static void commDestroy(MemorySegment comm) { check(st(() -> (int) H.commDestroy.invokeExact(comm))); }
    public static void shardedTopk(MemorySegment ctx, MemorySegment comm, MemorySegment scores, MemorySegment ids, int q, int kIn, int kOut,
                                   MemorySegment outIds, MemorySegment outScores) {
        check(st(() -> (int) H.shardedTopk.invokeExact(ctx, comm, scores, ids, q, kIn, kOut, outIds, outScores)));
    }
    public static void shardedSearchFlat(MemorySegment ctx, MemorySegment comm, int nLocal, MemorySegment luts, MemorySegment codesArray,
                                         MemorySegment vectorsArrayOrNull, MemorySegment idBases, MemorySegment queries, int q, int vsf, int topK,
                                         int rerankK, MemorySegment outIds, MemorySegment outScores) {
        check(st(() -> (int) H.shardedSearchFlat.invokeExact(ctx, comm, nLocal, luts, codesArray, vectorsArrayOrNull, idBases, queries, q, vsf, topK,
                                                             rerankK, outIds, outScores)));
    }
}
