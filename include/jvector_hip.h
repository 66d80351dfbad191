/*
 * jvector_hip.h — C ABI of libjvector_hip.so, the MI355X (gfx950) distance / quantization
 * engine that sits behind JVector's VectorizationProvider / VectorUtilSupport plugin surface.
 *
 * Boundary rules (mirroring the reference's native boundary):
 *   - extern "C", plain pointers and sizes, no C++/torch types;
 *   - the CALLER owns every user buffer; the library never retains a user pointer beyond the call
 *     (device-resident copies are explicit objects: jv_pq, jv_codes, jv_vectors, jv_fused, jv_luts);
 *   - unlike the reference's per-pair, never-failing kernels
 *     (jvector-native/src/main/native/src/jvector_simd.h:30-57), a GPU call can fail, so every entry
 *     point returns a jv_status and jv_hip_last_error() returns a thread-local message;
 *   - thread-safety: a jv_ctx is owned by ONE host thread (it holds that thread's HIP stream and
 *     scratch, the analogue of the reference's ThreadLocal scratch, ProductQuantization.java:237);
 *     jv_pq / jv_codes / jv_vectors / jv_fused are immutable after upload and may be shared by all
 *     contexts of the same device.
 *   - user pointers may be HOST (pageable or pinned) or DEVICE pointers; the library classifies them
 *     with hipPointerGetAttributes.  Host buffers are staged through the context's pinned scratch on
 *     the context's stream; device buffers are used in place (zero copy).
 *   - every call is asynchronous on the context's stream unless it returns data to a HOST buffer (then it
 *     synchronises the stream before returning).  jv_hip_ctx_sync() is the explicit barrier.
 *   - there is NO CPU fallback: without a usable gfx950 device every call fails with JV_ERR_NO_DEVICE.
 *
 * Each entry point cites the reference interface it batches/replaces (paths relative to /root/reference;
 * B/ = jvector-base/src/main/java/io/github/jbellis/jvector/).
 */
#ifndef JVECTOR_HIP_H
#define JVECTOR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JV_API __attribute__((visibility("default")))

typedef enum {
    JV_OK = 0,
    JV_ERR_INVALID = -1,      /* bad argument (the reference would throw IllegalArgumentException / assert) */
    JV_ERR_NO_DEVICE = -2,    /* no usable gfx950 device / HIP runtime */
    JV_ERR_HIP = -3,          /* a HIP runtime call failed; see jv_hip_last_error() */
    JV_ERR_OOM = -4,          /* device or pinned-host allocation failed */
    JV_ERR_UNSUPPORTED = -5   /* e.g. clusterCount > 256, anisotropic PQ with clusterCount != 256, NVQ bit widths other than 8 */
} jv_status;

/* VectorSimilarityFunction ordinals — B/vector/VectorSimilarityFunction.java:34-69 */
typedef enum { JV_EUCLIDEAN = 0, JV_DOT_PRODUCT = 1, JV_COSINE = 2 } jv_vsf;

/* Which reference decoder's query-side arithmetic to reproduce (they differ only in how the cosine
 * query magnitude is accumulated): PQDecoder (B/quantization/PQDecoder.java:88-122, full-vector
 * dotProduct) or FusedPQDecoder (B/quantization/FusedPQDecoder.java:178-191, per-subspace sums). */
typedef enum { JV_DECODER_PQ = 0, JV_DECODER_FUSED = 1 } jv_decoder_kind;

typedef struct jv_ctx jv_ctx;
typedef struct jv_pq jv_pq;
typedef struct jv_codes jv_codes;
typedef struct jv_vectors jv_vectors;
typedef struct jv_fused jv_fused;
typedef struct jv_luts jv_luts;

/* ---------------------------------------------------------------------------------------------
 * Library / context
 * ------------------------------------------------------------------------------------------- */
JV_API const char *jv_hip_version(void);
/* Thread-local description of the last failure on the calling thread ("" if none). Static storage. */
JV_API const char *jv_hip_last_error(void);
/* Number of visible gfx950 devices (0 when there is no GPU / no HIP runtime). Never fails. */
JV_API int jv_hip_device_count(void);
/* Diagnostic twin of jvector_simd_get_active_isa (jvector_simd.h:47): e.g. "gfx950:sramecc+:xnack-". */
JV_API const char *jv_hip_active_arch(int device);

/* stream: the hipStream_t every call of this context enqueues on.
 *   - an existing stream handle (e.g. the caller framework's current stream) — the engine's work is then
 *     ordered with the caller's own work on that stream;
 *   - NULL: the HIP null (legacy default) stream, which is ordered with all blocking streams;
 *   - JV_STREAM_PRIVATE: the context creates (and owns) its own non-blocking stream; the caller must then
 *     order its own producers/consumers of device buffers with jv_hip_ctx_sync() — device memory written by
 *     another stream and handed to the engine without that ordering is a data race. */
#define JV_STREAM_PRIVATE ((void *)(intptr_t)-1)
JV_API int jv_hip_ctx_create(int device, void *stream, jv_ctx **out);
JV_API int jv_hip_ctx_destroy(jv_ctx *ctx);
JV_API int jv_hip_ctx_sync(jv_ctx *ctx);
JV_API void *jv_hip_ctx_stream(jv_ctx *ctx);
/* Device-side timing of the engine's kernel regions with HIP events recorded on the context's stream
 * (the analogue of the reference's per-phase bench diagnostics, EX/benchmarks/diagnostics/).
 * regions: "adc" (ADC scan/gather/fused kernels), "topk", "exact", "lut", "encode", "norms".
 * profile(ctx, 1) clears the counters and starts recording; profile_read synchronises the stream and returns
 * the summed elapsed milliseconds and the number of recorded regions (one per API-level launch group). */
JV_API int jv_hip_ctx_profile(jv_ctx *ctx, int enable);
/* Per-context tuning options and event counters.  Every option also has a process-wide default in the environment variable
 * JVECTOR_HIP_<NAME IN CAPITALS>; a value set here wins for this context (a JVM cannot change its environment, and two
 * contexts may want different settings).  Options (all integers):
 *   graph_traversal  0 auto / 1 host / 2 device — overrides jv_hip_graph_set_traversal for searches on this context
 *   gs_vcap_log2     log2 of the device traversal's per-worker visited table (tier 2); pinning it also turns the in-kernel
 *                    growth pool and the retry passes off unless gs_grow / gs_retry = 1 ask for them
 *   gs_v1_log2       log2(slots) of the visited set's LDS tier; 0 = off, unset = the largest that keeps 8 waves per CU
 *   gs_grow, gs_retry, gs_tie_check, gs_push_log, gs_push_log_cap, gs_occ, gs_pair, gs_cand_cap, gs_waves_per_cu
 *                    device-traversal internals
 *   gs_wgx           the WORKGROUP form of the traversal (one query per workgroup, ADC table in LDS): unset = batches of up to 4
 *                    queries per CU, 1 / 0 = always / never; gs_wgx_waves, gs_wgx_slots, gs_wgx_depth, gs_wgx_per_cu, gs_wgx_lut_m tune it
 *   gs_pairc         0 = rows of 33 ... 64 neighbours (the builder's working rows) are scored one lane per neighbour instead of
 *                    pair lanes over the compacted fresh list
 *   gs_quad          a measured-and-lost form of the one-wave kernel (four lanes per neighbour in short expansions), experimental builds
 *                    only.  (gs_lutr and gs_ub8 — register-resident ADC table, in-kernel 8-bit bound table — left the source in round 6.)
 *   rd_split, rd_wide_stage (default 1), rd_chunk, rd_table_free (default 0)   forms of the robust-prune kernel;
 *                    selections are identical for every setting
 *   bl_insert_alpha_x100, bl_improve_beam   jv_hip_build_layered experiments: another alpha (x 100) for the insert phase, another
 *                    beam for the improveConnections passes
 *   gs_prof, graph_timing   1 = developer diagnostics on stderr
 *   no_filter        1 = jv_hip_search_flat materialises all scores instead of threshold-filtering them
 *   quiet            1 = no one-line notices on stderr (e.g. when AUTO traversal takes the host searcher)
 * Counters (jv_hip_ctx_get_stat; unknown names read 0), accumulated over the graph searches of this context:
 *   gs_calls_device, gs_calls_host, gs_calls_host_auto (AUTO fell back to the host searcher: the shape is outside the device
 *   traversal's coverage), gs_queries_device, gs_queries_retried (re-run on the device with a bigger visited table),
 *   gs_queries_host_fallback (finished by the host searcher after the device passes), gs_ties_resolved_device,
 *   gs_ties_to_host, gs_last_v1_log2, gs_last_workers_per_cu, gs_calls_wgx, gs_last_wgx (1: the last search ran in the workgroup
 *   form), gs_last_pair (1: pair lanes over the row, 2: over the compacted fresh list, 0: one lane per neighbour), gs_last_ubr (1: the
 *   last search ran the register-table bound form — option gs_ubr, on by default where it applies: pair-lane kernels, dot product /
 *   cosine, PQ-96; gs_ubrc, also on by default: the same form over the compacted fresh list of rows 33 ... 64 wide read by ordinal, i.e. the
 *   builder's own searches; gs_ubr_trim = candidates pushed between two trims of its queue), gs_ubr_dropped (neighbours that form dropped
 *   behind their bound, unscored), gs_last_rr_rows (rows [0, r) of every query's kept approximate results whose exact rerank score the
 *   traversal wave computed itself in the last search — option gs_fused_rerank, on by default where the rerank's transposing kernel
 *   applies: 16-byte aligned rows, D % 8 == 0, lists of <= 256, one-wave forms; the same scalar-order chain, the same bits;
 *   0: the rerank was a kernel of its own), gs_deferred / gs_defer_restarts (option gs_defer, on by default with that form over the
 *   row: fresh neighbours met at a level >= gs_defer_min_level (2) whose bound lies below the layer's best result are not scored —
 *   only the largest upper bound of their scores is kept; a pop such a node might outrank makes the query start over without
 *   deferral (gs_defer_restarts); an index on which more than a tenth of a batch started over is searched without deferral from
 *   then on unless gs_defer is set explicitly (gs_defer_switched_off); ids, scores and both counters are unchanged:
 *   GraphSearcher.java:263-282,324-331); experimental_build (1: the library was built with make EXPERIMENTAL=1 and also holds the
 *   measured-and-switched-off variants gs_quad, rd_table_free, rd_chunk, rd_square — the default build accepts
 *   and ignores their options). */
JV_API int jv_hip_ctx_set_option(jv_ctx *ctx, const char *name, int64_t value);
JV_API int jv_hip_ctx_clear_option(jv_ctx *ctx, const char *name);
JV_API int jv_hip_ctx_get_stat(jv_ctx *ctx, const char *name, int64_t *out);
JV_API int jv_hip_ctx_reset_stats(jv_ctx *ctx);
JV_API int jv_hip_ctx_profile_read(jv_ctx *ctx, const char *region, double *total_ms, int64_t *count);

/* ---------------------------------------------------------------------------------------------
 * ProductQuantization (device-resident codebooks)
 *   replaces: the codebook arguments of calculate_partial_sums_*_f32 (jvector_simd_kernel_list.h:53-55)
 *   and ProductQuantization's fields (B/quantization/ProductQuantization.java:66-75).
 * codebooks: concatenation over m of k*sizes[m] floats, centroid-major — the order
 *   ProductQuantization.write emits (:593-598).  sizes==NULL => getSubvectorSizesAndOffsets(D, M) (:535-550).
 * centroid: globalCentroid (D floats) or NULL.  k = clusterCount, 1..256 (ProductQuantization.checkClusterCount; one code byte).
 * A quantizer with fewer than 256 clusters is kept padded to 256 rows per sub-space with copies of centroid 0 (never chosen:
 * closestCentroidIndex keeps the first minimum), so every kernel keeps its table stride; jv_hip_pq_info / _write /
 * _self_magnitudes speak the caller's count.  jv_hip_pq_train / _refine train with the caller's count; FusedPQ
 * (FusedPQ.java:57-59) and anisotropic training / encoding need 256.
 * ------------------------------------------------------------------------------------------- */
JV_API int jv_hip_pq_create(jv_ctx *ctx, int D, int M, int k, const int *sizes, const float *codebooks,
                            const float *centroid, jv_pq **out);
/* Parses the reference's big-endian wire format (ProductQuantization.load :649-693; v0..v6). */
JV_API int jv_hip_pq_load(jv_ctx *ctx, const uint8_t *buf, size_t len, size_t *consumed, jv_pq **out);
/* ProductQuantization.anisotropicThreshold (ProductQuantization.java:72; encodeTo :439-449): t > -1 makes every encode call
 * use encodeAnisotropic (:269-306, coordinate descent on the parallel / perpendicular residual cost; vectors must be unit
 * length); -1 (the default, UNWEIGHTED) is the plain nearest-centroid encode.  jv_hip_pq_load takes the threshold from the
 * serialized form (version >= 3).  Valid range -1 <= t < 1 (KMeansPlusPlusClusterer.java:87-92). */
JV_API int jv_hip_pq_set_anisotropic_threshold(jv_pq *pq, float threshold);
JV_API float jv_hip_pq_anisotropic_threshold(const jv_pq *pq);
/* PQ training, unweighted k-means only.
 *   pq_train  = ProductQuantization.compute(ravv, M, k, globallyCenter) (ProductQuantization.java:109-139): optional global
 *               centring (centroidOf), per subspace k-means++ seeding + K_MEANS_ITERATIONS = 6 Lloyd rounds with the
 *               reference's early stop (<= 1 % of the points moved).  `vectors` is the training set [n][D] the caller
 *               sampled (the reference caps it at 128 000 vectors, extractTrainingVectors :141-175); n >= k.
 *   pq_refine = ProductQuantization.refine(ravv, lloydsRounds) (:194-221): Lloyd rounds on new data from pq's codebooks.
 * The reference draws from ThreadLocalRandom, so its codebooks are not reproducible; here a splitmix64 stream per subspace
 * derived from `seed` replaces it and everything else — including the order of every float accumulation — is the
 * reference's, so the result is a deterministic function of (vectors, seed).  Both return a NEW jv_pq. */
JV_API int jv_hip_pq_train(jv_ctx *ctx, const float *vectors, int64_t n, int D, int M, int k, int globally_center,
                           uint64_t seed, jv_pq **out);
/* ProductQuantization.compute(ravv, M, k, globallyCenter, anisotropicThreshold): after the unweighted rounds, 6 rounds of
 * anisotropic k-means (KMeansPlusPlusClusterer.java:274-320,380-432: weighted reassignment; centroids from the per-cluster
 * normalised outer-product sums through an explicit Gauss-Jordan inverse), sub-vectors of 2..16 dimensions, unit-length
 * input.  The result carries the threshold (its encode calls are anisotropic).  jv_hip_pq_refine on such a PQ runs
 * anisotropic rounds only, as the reference does (:212-214). */
JV_API int jv_hip_pq_train_anisotropic(jv_ctx *ctx, const float *vectors, int64_t n, int D, int M, int k, int globally_center,
                                       float anisotropic_threshold, uint64_t seed, jv_pq **out);
JV_API int jv_hip_pq_refine(jv_ctx *ctx, const jv_pq *pq, const float *vectors, int64_t n, int lloyds_rounds, uint64_t seed,
                            jv_pq **out);
/* ProductQuantization.write(out, version) (ProductQuantization.java:560-599; big-endian, versions 0..6): *len_out = bytes
 * needed; the block is written only when buf != NULL and cap >= *len_out (call once with buf = NULL to size the buffer).
 * jv_hip_pq_load / ProductQuantization.load read it back. */
JV_API int jv_hip_pq_write(jv_ctx *ctx, const jv_pq *pq, int version, uint8_t *buf, size_t cap, size_t *len_out);
JV_API int jv_hip_pq_destroy(jv_pq *pq);
JV_API int jv_hip_pq_info(const jv_pq *pq, int *D, int *M, int *k, int *has_centroid);

/* ---------------------------------------------------------------------------------------------
 * PQVectors (device-resident code store, ordinal-major: code of ordinal o at bytes [o*M, (o+1)*M))
 *   replaces: PQVectors.compressedDataChunks + getChunk/getOffsetInChunk (B/quantization/PQVectors.java:377-395);
 *   the 2 GiB chunking (PQLayout :515-540) is a JVM-array artefact and does not exist on the device.
 * ------------------------------------------------------------------------------------------- */
JV_API int jv_hip_codes_create(jv_ctx *ctx, const jv_pq *pq, int64_t count, jv_codes **out);
/* wrap caller-owned DEVICE memory (count*M bytes, 16-byte aligned) without copying */
JV_API int jv_hip_codes_wrap(jv_ctx *ctx, const jv_pq *pq, int64_t count, void *device_codes, jv_codes **out);
JV_API int jv_hip_codes_upload(jv_ctx *ctx, jv_codes *codes, int64_t first, int64_t count, const uint8_t *src);
JV_API int jv_hip_codes_download(jv_ctx *ctx, const jv_codes *codes, int64_t first, int64_t count, uint8_t *dst);
JV_API int jv_hip_codes_destroy(jv_codes *codes);
JV_API int64_t jv_hip_codes_count(const jv_codes *codes);
JV_API void *jv_hip_codes_device_ptr(const jv_codes *codes);

/* ---------------------------------------------------------------------------------------------
 * Full-resolution vectors (device-resident, row-major N x D float32)
 *   replaces: RandomAccessVectorValues.getVector / getVectorInto as used by rerankerFor
 *   (B/graph/RandomAccessVectorValues.java:114-124; OnDiskGraphIndex.java:565-581).
 * ------------------------------------------------------------------------------------------- */
JV_API int jv_hip_vectors_create(jv_ctx *ctx, int64_t count, int D, jv_vectors **out);
JV_API int jv_hip_vectors_wrap(jv_ctx *ctx, int64_t count, int D, void *device_vectors, jv_vectors **out);
JV_API int jv_hip_vectors_upload(jv_ctx *ctx, jv_vectors *v, int64_t first, int64_t count, const float *src);
/* Wrapped (caller-owned) vectors edited in place: the engine caches one float per row for the cosine rerank (the reference's
 * norm2 accumulator, DefaultVectorUtilSupport.java:131-137); tell it that the rows changed.  jv_hip_vectors_upload does this itself. */
JV_API int jv_hip_vectors_invalidate(jv_vectors *v);
JV_API int jv_hip_vectors_destroy(jv_vectors *v);

/* ---------------------------------------------------------------------------------------------
 * ProductQuantization.encode — SURVEY §8a row 3
 *   replaces: PQVectors.encodeAndBuild's parallel forEach (B/quantization/PQVectors.java:137-149) ->
 *   ProductQuantization.encodeTo/encodeUnweighted/closestCentroidIndex (:422-449,:507-520).
 *   Bit-exact: centring by float subtraction, sequential non-fused (v-c)^2 sums, strict '<' first-min.
 * vectors: count x D floats (host or device); codes_out: count x M bytes (host or device).
 * ------------------------------------------------------------------------------------------- */
JV_API int jv_hip_pq_encode(jv_ctx *ctx, const jv_pq *pq, const float *vectors, int64_t count, uint8_t *codes_out);
/* encode rows [first, first+count) of a device-resident vector set straight into a code store */
JV_API int jv_hip_pq_encode_into(jv_ctx *ctx, const jv_pq *pq, const jv_vectors *v, int64_t first, int64_t count,
                                 jv_codes *codes);

/* ---------------------------------------------------------------------------------------------
 * ADC look-up tables for a batch of queries — SURVEY §8a row 2
 *   replaces: PQDecoder.CachingDecoder / CosineDecoder constructors (B/quantization/PQDecoder.java:41-54,
 *   88-122) and FusedPQDecoder constructors (B/quantization/FusedPQDecoder.java:49-77,146-192), i.e.
 *   M calls of VectorUtil.calculatePartialSums (VectorUtilSupport.java:135; native
 *   calculate_partial_sums_{dot,euclidean}_f32) per query, + calculatePartialSelfMagnitudes once per PQ.
 * queries: Q x D floats (host or device), NOT pre-centred (the library subtracts globalCentroid).
 * The returned object owns Q x M x 256 floats on the device and is reusable: calling build again on the
 * same object with Q' <= capacity re-fills it without allocation.
 * ------------------------------------------------------------------------------------------- */
JV_API int jv_hip_luts_create(jv_ctx *ctx, const jv_pq *pq, int max_queries, jv_luts **out);
JV_API int jv_hip_luts_build(jv_ctx *ctx, jv_luts *luts, const float *queries, int Q, jv_vsf vsf,
                             jv_decoder_kind kind);
JV_API int jv_hip_luts_destroy(jv_luts *luts);
/* test/diagnostic access: copies query q's table (M*256 floats) and bMagnitude to host */
JV_API int jv_hip_luts_download(jv_ctx *ctx, const jv_luts *luts, int q, float *lut_out, float *bmag_out);
/* test/diagnostic access: the 8-bit upper-bound tables the register-table traversal (option gs_ubr) loads for the queries last
 * staged by jv_hip_luts_build (dot product / cosine, uniform 8-dim sub-vectors): tab_out = Q x M x 64 dwords in the register
 * layout of gs_host.h gs_ubr_build_ref, meta_out = Q x 4 floats {sum of low edges + slack, scale, usable, 0} */
JV_API int jv_hip_luts_bound_tables(jv_ctx *ctx, const jv_luts *luts, uint32_t *tab_out, float *meta_out);
/* copies the PQ's cosine self-magnitude table (M*256 floats; calculatePartialSelfMagnitudes) to host */
JV_API int jv_hip_pq_self_magnitudes(jv_ctx *ctx, const jv_pq *pq, float *out);

/* ---------------------------------------------------------------------------------------------
 * ADC scoring — SURVEY §8a rows 5 and 6
 *   replaces: one assemble_and_sum_f32 / pq_decoded_cosine_similarity_f32 call per candidate
 *   (jvector_simd_kernel_list.h:50,52; PQDecoder.similarityTo B/quantization/PQDecoder.java:65-80,124-135)
 *   including the score transform 1/(1+d), (1+s)/2, (1+c)/2.
 * scan:   scores_out[q*count + i] = similarityTo(first + i)   for i in [0,count), all Q queries of `luts`
 * gather: scores_out[q*B + j]     = similarityTo(ordinals[q*B + j]); an ordinal < 0 yields -INFINITY
 * scores_out / ordinals: host or device.
 * ------------------------------------------------------------------------------------------- */
JV_API int jv_hip_adc_scan(jv_ctx *ctx, const jv_luts *luts, const jv_codes *codes, int64_t first, int64_t count,
                           float *scores_out);
JV_API int jv_hip_adc_scores(jv_ctx *ctx, const jv_luts *luts, const jv_codes *codes, const int32_t *ordinals,
                             int B, float *scores_out);

/* ---------------------------------------------------------------------------------------------
 * Fused-PQ ("FusedADC") L0 blocks — SURVEY §8a row 7
 *   replaces: FusedPQDecoder.enableSimilarityToNeighbors + similarityToNeighbor
 *   (B/quantization/FusedPQDecoder.java:85-111,206-213) over the block written by FusedPQ.writeInline
 *   (B/graph/disk/feature/FusedPQ.java:146-161): neighbour i's code at bytes [i*M,(i+1)*M) of the origin
 *   node's block, zero-padded to maxDegree*M.
 * blocks:    count x (maxDegree*M) bytes;  neighbors: count x maxDegree int32 (pad -1), as in the L0 record
 *            (B/graph/disk/OnDiskGraphIndex.java:538-547).
 * scores_out[q*maxDegree + i] = similarityToNeighbor(origins[q], i); slots with neighbour id -1 get -INFINITY.
 * ------------------------------------------------------------------------------------------- */
JV_API int jv_hip_fused_create(jv_ctx *ctx, const jv_pq *pq, int64_t count, int maxDegree, jv_fused **out);
JV_API int jv_hip_fused_upload(jv_ctx *ctx, jv_fused *f, int64_t first, int64_t count, const uint8_t *blocks,
                               const int32_t *neighbors);
JV_API int jv_hip_fused_destroy(jv_fused *f);
JV_API int jv_hip_fused_scores(jv_ctx *ctx, const jv_luts *luts, const jv_fused *f, const int32_t *origins,
                               float *scores_out, int32_t *neighbors_out /* nullable: Q x maxDegree ids */);

/* ---------------------------------------------------------------------------------------------
 * Full-resolution scoring — SURVEY §8a row 1
 *   replaces: VectorSimilarityFunction.compare (B/vector/VectorSimilarityFunction.java:37-69) ->
 *   dot_product_f32 / euclidean_f32 / cosine_f32 (jvector_simd_kernel_list.h:38-40), one call per pair, as
 *   driven by NodeQueue.rerank (B/graph/NodeQueue.java:160-195).
 *   Accumulation order is the scalar DefaultVectorUtilSupport order (bit-exact to it).
 * gather: scores_out[q*B + j] = compare(queries[q], vectors[ordinals[q*B+j]]); ordinal < 0 -> -INFINITY
 * scan:   scores_out[q*count + i] = compare(queries[q], vectors[first+i])  (brute force / ground truth)
 * ------------------------------------------------------------------------------------------- */
JV_API int jv_hip_exact_scores(jv_ctx *ctx, const jv_vectors *v, const float *queries, int Q, jv_vsf vsf,
                               const int32_t *ordinals, int B, float *scores_out);
JV_API int jv_hip_exact_scan(jv_ctx *ctx, const jv_vectors *v, const float *queries, int Q, jv_vsf vsf,
                             int64_t first, int64_t count, float *scores_out);
/* The exact build-score provider — BuildScoreProvider.randomAccessScoreProvider (B/graph/similarity/BuildScoreProvider.java:
 * 106-160): diversityScoreFunctionFor(node1).similarityTo(node2) = similarityFunction.compare(vectors[node1], vectors[node2])
 * for P x B (node, candidate) blocks — the full-resolution counterpart of jv_hip_code_pair_scores, same arithmetic as
 * jv_hip_exact_scores (searchProviderFor(node1) is jv_hip_exact_scores with the node's own row as the query).
 * scores_out[p*B + b]; an ordinal outside [0, count) on either side gives -INFINITY.  Buffers: host or device memory. */
JV_API int jv_hip_exact_pair_scores(jv_ctx *ctx, const jv_vectors *v, jv_vsf vsf, const int32_t *node1, int P, const int32_t *node2,
                                    int B, float *scores_out);
/* MFMA tile form of the scan (north_star: "MFMA only for the batched query x candidates GEMM form of full-resolution
 * rerank"; SURVEY §8d: the dense Q x N form of row 1 — brute force, ground truth).  Same arguments and output layout as
 * jv_hip_exact_scan, but NOT bit-identical to it: dot products and norms are k-ascending f32 fused-multiply-add chains (what
 * v_mfma_f32_32x32x2_f32 computes; the reference's native library fuses as well, jvector_simd_kernels.cpp:208-286), cosine
 * finishes as jv_hip_exact_scan does, L2 uses |q|^2 + |v|^2 - 2 q.v.  Scores agree with jv_hip_exact_scan to 1e-5 relative
 * and are themselves reproducible bit for bit (jvector_amd/csrc/ed_body.h states the order).  Use it for candidate
 * generation / ground truth at Q >= ~32, and jv_hip_exact_scores / jv_hip_exact_scan where the bit-exact order matters. */
JV_API int jv_hip_exact_scan_dense(jv_ctx *ctx, const jv_vectors *v, const float *queries, int Q, jv_vsf vsf,
                                   int64_t first, int64_t count, float *scores_out);

/* ---------------------------------------------------------------------------------------------
 * NVQ ("NuVeQ", non-uniform vector quantization) — the reference's compressed RERANK codec; SURVEY 8 f.4 names it as what
 * follows the format readers.  One byte per dimension plus four floats per sub-vector: a reranked candidate costs
 * D + 16 S bytes of HBM instead of 4 D.
 *   replaces: NVQuantization.compute / create / encodeAll / encode (B/quantization/NVQuantization.java:153-216,
 *             QuantizedSubVector.quantizeTo :508-557 incl. the growth-rate search over nvqLoss / nvqUniformLoss),
 *             NVQVectors.scoreFunctionFor / NVQScorer (B/quantization/NVQScorer.java:33-137),
 *             NVQ.rerankerFor / SeparatedNVQ.rerankerFor (B/graph/disk/feature/NVQ.java:96-110),
 *             VectorUtilSupport.nvqQuantize8bit / nvqLoss / nvqUniformLoss / nvqDotProduct8bit / nvqSquareL2Distance8bit /
 *             nvqCosine8bit / nvqShuffleQueryInPlace8bit (B/vector/VectorUtilSupport.java; scalar bodies
 *             DefaultVectorUtilSupport.java:385-548, native NC/src/jvector_simd_kernels.cpp:1029-1643).
 * Arithmetic = the scalar provider's (Math.fma chains over each sub-vector in dimension order); encoded bytes, parameters
 * and scores are bit-identical to it.  The query is NOT shuffled (DefaultVectorUtilSupport's no-op): callers hand over
 * plain queries.
 *   params   : per vector S x 4 floats {minValue, maxValue, growthRate, midpoint} — the order QuantizedSubVector.write
 *              serialises them (:577-587)
 *   bytes    : per vector D bytes, the sub-vectors' bytes concatenated (sub-vector split = getSubvectorSizesAndOffsets)
 * jv_hip_vectors_from_nvq wraps NVQ rows as a jv_vectors: every search entry point that takes `vectors` for its rerank
 * (jv_hip_search_flat, jv_hip_graph_search, jv_hip_searcher_*, jv_hip_sharded_search, jv_hip_exact_scores) then reranks
 * with the NVQ score function, as a graph whose features are FUSED_PQ + NVQ_VECTORS does; entry points that need the
 * float rows themselves (encode, scans, construction) return JV_ERR_UNSUPPORTED for such a set.
 * ------------------------------------------------------------------------------------------- */
typedef struct jv_nvq jv_nvq;
typedef struct jv_nvq_vectors jv_nvq_vectors;
/* NVQuantization.create(globalMean, nSubVectors); global_mean: D floats (host or device) */
JV_API int jv_hip_nvq_create(jv_ctx *ctx, int D, int n_subvectors, const float *global_mean, jv_nvq **out);
/* NVQuantization.compute(ravv, nSubVectors): the mean is accumulated over the rows in order, on the device */
JV_API int jv_hip_nvq_compute(jv_ctx *ctx, const jv_vectors *v, int n_subvectors, jv_nvq **out);
JV_API int jv_hip_nvq_set_learn(jv_nvq *nvq, int learn); /* NVQuantization.learn (default 1) */
JV_API int jv_hip_nvq_dimension(const jv_nvq *nvq);
JV_API int jv_hip_nvq_subvectors(const jv_nvq *nvq);
JV_API int jv_hip_nvq_global_mean(jv_ctx *ctx, const jv_nvq *nvq, float *dst);
JV_API int jv_hip_nvq_destroy(jv_nvq *nvq);
JV_API int jv_hip_nvq_vectors_create(jv_ctx *ctx, const jv_nvq *nvq, int64_t count, jv_nvq_vectors **out);
/* encodeAll: rows [first, first + count) of v into rows [dst_first, ...) of dst */
JV_API int jv_hip_nvq_encode(jv_ctx *ctx, const jv_nvq *nvq, const jv_vectors *v, int64_t first, int64_t count,
                             jv_nvq_vectors *dst, int64_t dst_first);
/* rows decoded from a file (jv_fmt_nvq_unpack); bytes: count x D, params: count x S x 4 (host or device) */
JV_API int jv_hip_nvq_vectors_upload(jv_ctx *ctx, jv_nvq_vectors *nv, int64_t first, int64_t count, const uint8_t *bytes,
                                     const float *params);
JV_API int jv_hip_nvq_vectors_download(jv_ctx *ctx, const jv_nvq_vectors *nv, int64_t first, int64_t count, uint8_t *bytes,
                                       float *params);
JV_API int64_t jv_hip_nvq_vectors_count(const jv_nvq_vectors *nv);
JV_API int jv_hip_nvq_vectors_destroy(jv_nvq_vectors *nv);
/* scores_out[q][b] = NVQVectors.scoreFunctionFor(query q, vsf).similarityTo(ordinals[q*B + b]); ordinals outside
 * [0, count) give -inf */
JV_API int jv_hip_nvq_scores(jv_ctx *ctx, const jv_nvq_vectors *nv, const float *queries, int Q, jv_vsf vsf,
                             const int32_t *ordinals, int B, float *scores_out);
/* a jv_vectors whose rows are `nv`; destroy with jv_hip_vectors_destroy.
 * OWNERSHIP ORDER of the NVQ handles (nothing is reference-counted): a jv_nvq must outlive every jv_nvq_vectors created from it, a
 * jv_nvq_vectors must outlive every jv_vectors view of it — destroy views, then rows, then the NVQuantization (the Python mirror keeps
 * the parents alive for you).  Every handle of a call must live on the context's device (JV_ERR_INVALID otherwise). */
JV_API int jv_hip_vectors_from_nvq(jv_ctx *ctx, jv_nvq_vectors *nv, jv_vectors **out);

/* ---------------------------------------------------------------------------------------------
 * Top-k under the NodeQueue total order — SURVEY §8a row 9
 *   replaces: NodeQueue.encode + BoundedLongHeap.push (B/graph/NodeQueue.java:125-129,
 *   B/util/BoundedLongHeap.java:59-69, B/util/NumericUtils.java:49-65): higher score first, ties -> smaller id.
 * scores: Q rows of n floats (row stride `stride` elements);  ids: matching int32 rows, or NULL meaning
 *   id = id_base + column.  Entries whose id < 0 (explicit ids) are ignored.
 * out_ids / out_scores: Q x k, best first; when fewer than k valid entries exist the tail is (-1, -INFINITY).
 * Also the merge step of the sharded configuration: feed the all-gathered partial lists with global ids.
 * ------------------------------------------------------------------------------------------- */
JV_API int jv_hip_topk(jv_ctx *ctx, const float *scores, const int32_t *ids, int Q, int64_t n, int64_t stride,
                       int32_t id_base, int k, int32_t *out_ids, float *out_scores);

/* ---------------------------------------------------------------------------------------------
 * Two-pass flat search: LUT build -> ADC scan of all codes -> top-rerankK -> exact rerank -> top-K.
 *   Arithmetic per candidate identical to GraphSearcher's scoring calls (B/graph/GraphSearcher.java:443-450,
 *   471-507); the candidate SET is the whole shard instead of a graph frontier.
 *   id_base is added to every returned id (shard offset in the sharded configuration).
 *   vectors == NULL or rerankK == 0  => no rerank, results are the ADC top-K.
 * out_ids / out_scores: Q x topK (host or device).
 * ------------------------------------------------------------------------------------------- */
JV_API int jv_hip_search_flat(jv_ctx *ctx, jv_luts *luts, const jv_codes *codes, const jv_vectors *vectors,
                              const float *queries, int Q, jv_vsf vsf, int topK, int rerankK, int32_t id_base,
                              int32_t *out_ids, float *out_scores);

/* ---------------------------------------------------------------------------------------------
 * Batched graph searcher — SURVEY §8f rank 1 ("next" row): a multi-query restatement of
 *   GraphSearcher.search / searchOneLayer / reranking (B/graph/GraphSearcher.java:222-507) +
 *   View.processNeighbors (B/graph/disk/OnDiskGraphIndex.java:639-661, B/graph/OnHeapGraphIndex.java:475-483).
 * Two traversals behind the same call (jv_hip_graph_set_traversal below).  DEVICE: one wavefront per query keeps the
 * candidate / result queues in LDS and the visited set in device memory and runs the whole loop on the GPU.  HOST: the
 * traversal state lives on the host and every round ships one expanded node per live query to the GPU.  Either way layer 0
 * with FusedPQ scores the origin's packed block (FusedPQDecoder.similarityToNeighbor), otherwise the neighbours' own codes
 * (PQDecoder.similarityTo), and per query the visit order, scores, results, visitedCount and expandedCount equal the
 * reference's sequential search on the same graph.
 *
 * Graph: level 0 holds every node (count == n_nodes, node_ids == NULL); upper levels list their node ids in
 * ascending order.  neighbors: count x degree int32, packed, padded with -1 (the L0 record layout,
 * OnDiskGraphIndex.java:538-547).  Adjacency is HOST memory and is copied.
 * search: `luts` is (re)built inside for the Q queries with the decoder kind implied by `fused`;
 *   vectors == NULL => no rerank (results are the approximate top-K);
 *   stats (nullable): Q x 2 int64 = {visitedCount, expandedCount} per query (SearchResult counters).
 * ------------------------------------------------------------------------------------------- */
typedef struct jv_graph jv_graph;
JV_API int jv_hip_graph_create(jv_ctx *ctx, int64_t n_nodes, int n_levels, jv_graph **out);
JV_API int jv_hip_graph_set_level(jv_ctx *ctx, jv_graph *g, int level, int count, const int32_t *node_ids,
                                  const int32_t *neighbors, int degree);
/* Raw device memory for callers that have no allocator of their own (a Java host; torch users pass their tensors instead):
 * e.g. the mutable adjacency of jv_hip_graph_set_level0_device.  Freed with jv_hip_device_free. */
JV_API int jv_hip_device_alloc(jv_ctx *ctx, size_t bytes, void **out);
JV_API int jv_hip_device_free(jv_ctx *ctx, void *ptr);
/* Level 0 in CALLER-owned device memory: n_nodes x degree int32, packed rows padded with -1, read in place by the device
 * traversal on every search — the owner may rewrite rows between searches (incremental construction: search the partial
 * graph, prune, write the new rows, search again).  No host copy is kept, so such a graph is searched by the device
 * traversal only, without FusedPQ blocks (scores come from the code store: PQDecoder.similarityTo).  Upper levels, if any,
 * are set with jv_hip_graph_set_level as usual.  Ids must stay inside [-1, n_nodes): the kernel does not check them. */
JV_API int jv_hip_graph_set_level0_device(jv_ctx *ctx, jv_graph *g, const int32_t *d_neighbors, int degree);
JV_API int jv_hip_graph_set_entry(jv_graph *g, int32_t node, int level);
JV_API int jv_hip_graph_destroy(jv_graph *g);
/* Where the traversal state (candidate / result queues, visited set) lives.
 *   HOST   : the host batched searcher (C++ worker pool; the GPU scores each round's frontier).
 *   DEVICE : one wavefront per query keeps the queues in LDS / L2 and runs the whole loop on the GPU (256-cluster codebooks,
 *            degree <= 512 — rows wider than 64 are walked 64 neighbours at a time —: kernels specialised for uniform 8-dim sub-vectors at M = 16, 32, 48, 64, 96, 128, 192, a generic
 *            build for every other quantizer — ragged or other sub-vector sizes, any M); queries that outgrow its
 *            fixed-size structures are re-run on the host.  Results, visitedCount and expandedCount are identical either way.
 *   AUTO   : DEVICE wherever it applies (as above and the queues fit LDS), else HOST.
 *            The environment variable JVECTOR_HIP_GRAPH_TRAVERSAL=host|device overrides the setting.
 *   Queries that outgrow the device kernel's fixed-size structures are first retried on the device with an 8x / 64x
 *   larger visited table; only what still overflows is re-run on the host. */
enum { JV_TRAVERSAL_AUTO = 0, JV_TRAVERSAL_HOST = 1, JV_TRAVERSAL_DEVICE = 2 };
JV_API int jv_hip_graph_set_traversal(jv_graph *g, int mode);
JV_API int jv_hip_graph_search(jv_ctx *ctx, const jv_graph *g, jv_luts *luts, const jv_codes *codes,
                               const jv_fused *fused, const jv_vectors *vectors, const float *queries, int Q,
                               jv_vsf vsf, int topK, int rerankK, int32_t *out_ids, float *out_scores, int64_t *stats);

/* ---------------------------------------------------------------------------------------------
 * Build-time scoring: the PQ-only score functions graph construction uses
 * (BuildScoreProvider.pqBuildScoreProvider, B/graph/similarity/BuildScoreProvider.java:167-212), batched.
 *   pair table   = ProductQuantization.createCodebookPartialSums(vsf) (ProductQuantization.java:609-628): per subspace
 *                  the upper triangle of centroid x centroid dot products (DOT_PRODUCT, COSINE) or squared distances
 *                  (EUCLIDEAN); M * k(k+1)/2 floats, built once per (pq, vsf) and kept on the device.
 *   code_pair_scores[p][b] = ImmutablePQVectors.diversityFunctionFor(node1[p], vsf).similarityTo(node2[p*B + b])
 *                  (ImmutablePQVectors.java:61-104 -> VectorUtil.assembleAndSumPQ, DefaultVectorUtilSupport.java:312-335,
 *                  native jvector_simd_kernels.cpp:729-815): the candidate x selected score blocks of
 *                  VamanaDiversityProvider.retainDiverse.  Ordinals outside [0, count) (e.g. -1 padding) give -inf.
 *   pq_decode    = ProductQuantization.decode (ProductQuantization.java:454-471) of `count` codes (the rows `ordinals`
 *                  lists, or first..first+count when ordinals is NULL): pqBuildScoreProvider.searchProviderFor(node1)
 *                  searches from the DECODED vector (feed the result to jv_hip_luts_build).
 *   direct_scores[q][b] = PQVectors.scoreFunctionFor(query q, vsf).similarityTo(ordinals[q*B + b])
 *                  (PQVectors.java:223-281): query vs code without a look-up table.
 * Pointers may be host or device memory, as everywhere in this header.
 * ------------------------------------------------------------------------------------------- */
typedef struct jv_pair_table jv_pair_table;
JV_API int jv_hip_pair_table_create(jv_ctx *ctx, const jv_pq *pq, jv_vsf vsf, jv_pair_table **out);
JV_API int64_t jv_hip_pair_table_size(const jv_pair_table *t); /* number of floats */
JV_API int jv_hip_pair_table_download(jv_ctx *ctx, const jv_pair_table *t, float *dst);
JV_API int jv_hip_pair_table_destroy(jv_pair_table *t);
JV_API int jv_hip_code_pair_scores(jv_ctx *ctx, const jv_pair_table *t, const jv_codes *codes, const int32_t *node1, int P,
                                   const int32_t *node2, int B, float *scores_out);
/* FusedPQ.writeInline for nodes [first, first + count) (FusedPQ.java:146-161): stores the neighbour rows (count x maxDegree
 * int32, -1 padded, host or device memory) and gathers each neighbour's code out of `codes` into the node's packed block,
 * zero padded — the producer of the layout jv_hip_fused_scores / the graph searcher read.  Neighbour ids index `codes`. */
/* Batched robust prune — VamanaDiversityProvider.retainDiverse (B/graph/diversity/VamanaDiversityProvider.java:43-96) for P
 * nodes at once with the PQ diversity score above as scoreProvider.diversityScoreFunctionFor (BuildScoreProvider.java:181-186):
 * BASELINE config 5's "GPU-batched neighbor scoring".  Per node p: the NodeArray is cand_nodes / cand_scores[p*C .. p*C +
 * cand_count[p]) (sorted by score descending by the caller, as NodeArray keeps it; cand_count NULL = C entries each);
 * diverse_before[p] (NULL = 0) candidates at the front are taken as already diverse; alpha is GraphIndexBuilder's alpha
 * (the prune ramps 1.0, 1.2, ... <= alpha + 1e-6).
 *   selected_out   : P x maxDegree candidate INDICES (positions in the node's list) in ascending order, -1 padded — the set
 *                    bits of the reference's `selected` BitSet
 *   n_selected_out : P (nSelected)      short_edges_out : P floats or NULL (retainDiverse's return value; NaN if the loop never ran)
 * maxDegree <= 64; C * M bytes of candidate codes must fit LDS next to the bookkeeping (C <= ~600 at M = 96), else
 * JV_ERR_UNSUPPORTED.  Buffers may be host or device memory.  Selections are identical to the reference's sequential loop. */
JV_API int jv_hip_retain_diverse(jv_ctx *ctx, const jv_pair_table *t, const jv_codes *codes, int P, int C,
                                 const int32_t *cand_nodes, const float *cand_scores, const int32_t *cand_count,
                                 const int32_t *diverse_before, int maxDegree, float alpha, int32_t *selected_out,
                                 int32_t *n_selected_out, float *short_edges_out);
JV_API int jv_hip_fused_build(jv_ctx *ctx, jv_fused *f, const jv_codes *codes, int64_t first, int64_t count,
                              const int32_t *neighbors);

/* Batched Vamana construction of one graph level (BASELINE config 5) — the driver of the calls above, so that a host needs no
 * array plumbing of its own:  replaces the per-node GraphIndexBuilder.addGraphNode loop (B/graph/GraphIndexBuilder.java:605-659:
 * search -> VamanaDiversityProvider.retainDiverse -> ConcurrentNeighborMap.insertDiverse / backlink, :104-163) and
 * cleanup()'s enforceDegree (:472-508) with batch calls.  Scores are the PQ build-score provider's (BuildScoreProvider.java:
 * 167-212) except that the insert query is the node's full-resolution vector, not its decoded code.
 *   create       : nodes = the `codes->count` ordinals of `codes` / `vectors`; the adjacency (count x floor(maxDegree *
 *                  neighborOverflow) int32, device memory owned by the builder) starts empty.  beamWidth = candidates per
 *                  insert; alpha = the robust prune's relaxation (the prune ramps 1.0, 1.2, ... <= alpha).
 *   seed         : the first node (entry point of the construction-time searches)
 *   insert_batch : inserts `nodes[0..B)` (host or device memory; none of them inserted before) as B concurrent inserts that
 *                  do not see each other: candidate search over the graph so far (device traversal), robust prune, rows,
 *                  backlinks, re-prune of the lists that outgrow the working width.  Callers grow the batch with the graph
 *                  (prefix doubling: batch <= nodes already inserted).
 *   improve_batch: improveConnections (GraphIndexBuilder.java:510-560, what cleanup() :472-508 runs before enforceDegree) for
 *                  `nodes[0..B)`, all of them IN the graph: search the graph as it stands for each node, MERGE the results with the
 *                  neighbours the node has (ConcurrentNeighborMap.insertDiverse), robust-prune the merged list scored with the PQ
 *                  diversity function, rewrite the row, backlink its members.  A pass over every level-0 node after the last
 *                  insert is what the bench's --build-improve does.
 *   finish       : enforceDegree on every list; neighbors_out (nullable; host or device) receives count x maxDegree int32,
 *                  rows packed, -1 padded.  The builder can keep inserting afterwards.
 *   stats        : seconds3 = {search, prune, backlink}; counts5 = {batches, re-pruned lists, inserted nodes, visitedCount and
 *                  expandedCount summed over the construction-time searches}
 *   neighbors_device : the working adjacency in place (row width in *row_width) — e.g. for jv_hip_fused_build.
 * The result depends on the insertion order and batch boundaries only (no atomics decide an edge), so a build is reproducible. */
typedef struct jv_builder jv_builder;
JV_API int jv_hip_builder_create(jv_ctx *ctx, const jv_pq *pq, const jv_codes *codes, const jv_vectors *vectors, jv_vsf vsf,
                                 int max_degree, int beam_width, float alpha, float neighbor_overflow, jv_builder **out);
JV_API int jv_hip_builder_seed(jv_ctx *ctx, jv_builder *b, int32_t node);
JV_API int jv_hip_builder_insert_batch(jv_ctx *ctx, jv_builder *b, const int32_t *nodes, int B);
JV_API int jv_hip_builder_improve_batch(jv_ctx *ctx, jv_builder *b, const int32_t *nodes, int B);
JV_API int jv_hip_builder_finish(jv_ctx *ctx, jv_builder *b, int32_t *neighbors_out);
JV_API int jv_hip_builder_stats(const jv_builder *b, double *seconds3, int64_t *counts5);
JV_API const int32_t *jv_hip_builder_neighbors_device(const jv_builder *b, int *row_width);
/* REFERENCE ORDER (context option bl_ref_order = 1, read by jv_hip_builder_create): the builder keeps its lists the way
 * ConcurrentNeighborMap.Neighbors does — every entry with the score it was inserted under, NodeArray order, the diverseBefore mark —
 * and performs insertDiverse / backlink -> Neighbors.insert / retainDiverse(diverseBefore) / enforceDegree as the reference does
 * (ConcurrentNeighborMap.java:139-146,190-200,222-296; NodeArray.java:166-228).  With ONE node per batch that is addGraphNode
 * (GraphIndexBuilder.java:605-659) operation for operation: the adjacency equals the reference's one-thread build (oracle:
 * jvo_builder_*, tests/test_builder_reference_order.py).  working_lists copies the lists as they stand: ids_out [n x row_width] (-1
 * padded), scores_out [n x row_width] and diverse_before_out [n] (each nullable; host or device memory; the last two only in
 * reference order / with sorted lists: JV_ERR_INVALID otherwise).
 * LIMITS of the byte-for-byte claim: it holds for one node per batch, without re-inserts, with maxDegree x overflow <= 64.
 * Three places differ from Neighbors.insert beyond that: (1) back edges and improve candidates are de-duplicated BY ID, the reference
 * rejects a duplicate only at an equal score (insertionPoint == -1) and can list a node twice under two PQ scores — it does so on
 * improve passes; (2) a back-link merge drops what does not fit a working list of R + 2R entries, the reference's list grows;
 * (3) the hard maximum is clamped to 64 entries, the reference uses (int)(overflow x maxDegree) uncapped.
 * bl_sorted_lists = 1: the DEFAULT build's own scores (the symmetric PQ diversity function of node and member) stored and the lists
 * kept sorted instead of re-scored and re-sorted at every re-prune — the identical graph (rows as sets); = 2 adds the diverseBefore
 * shortcut, bl_ref_order = 2 removes it from reference order: the two ablations of DESIGN.md §7. */
JV_API int jv_hip_builder_working_lists(jv_ctx *ctx, const jv_builder *b, int32_t *ids_out, float *scores_out, int32_t *diverse_before_out);
JV_API int jv_hip_builder_destroy(jv_builder *b);
/* The whole LAYERED build in one call (GraphIndexBuilder with addHierarchy): levels drawn per node like getRandomGraphLevel (:562-575:
 * floor(-ln(U) / ln(maxDegree)), U from a splitmix64 seeded with `seed`; levels with fewer than min_top nodes fold into the one
 * below), every level a jv_builder of its own (inserts in a seeded random order with prefix-doubling batches of at most max_batch,
 * then improve_passes passes of improveConnections over every node of the level, then enforceDegree), entry point = the top level's
 * node most similar to the mean of the top level's vectors.  A function of (data, parameters, seed) only.
 *   info   : n_levels, entry node / level, level_counts[n_levels] (level 0 = every ordinal)
 *   level  : level >= 1: nodes_out[count] ascending node ids, neighbors_out[count x maxDegree] global ids (host memory);
 *            level 0: nodes_out must be NULL, neighbors_out[n x maxDegree] may be host or device memory
 *   level0_device : the level-0 rows in place, e.g. for jv_hip_fused_build
 *   stats  : seconds4 = {search, prune, backlink, total}; counts5 as jv_hip_builder_stats, summed over the levels */
typedef struct jv_layered jv_layered;
JV_API int jv_hip_build_layered(jv_ctx *ctx, const jv_pq *pq, const jv_codes *codes, const jv_vectors *vectors, jv_vsf vsf, int max_degree,
                                int beam_width, float alpha, float neighbor_overflow, int max_batch, int improve_passes, uint64_t seed,
                                int min_top, jv_layered **out);
JV_API int jv_hip_layered_info(const jv_layered *l, int *n_levels, int32_t *entry_node, int *entry_level, int64_t *level_counts);
JV_API int jv_hip_layered_level(jv_ctx *ctx, const jv_layered *l, int level, int32_t *nodes_out, int32_t *neighbors_out);
JV_API const int32_t *jv_hip_layered_level0_device(const jv_layered *l);
JV_API int jv_hip_layered_stats(const jv_layered *l, double *seconds4, int64_t *counts5);
JV_API int jv_hip_layered_destroy(jv_layered *l);
/* copy blocks (count x maxDegree*M bytes) and / or neighbour rows back; either output may be NULL */
JV_API int jv_hip_fused_download(jv_ctx *ctx, const jv_fused *f, int64_t first, int64_t count, uint8_t *blocks_out,
                                 int32_t *neighbors_out);
JV_API int jv_hip_pq_decode(jv_ctx *ctx, const jv_codes *codes, const int32_t *ordinals, int64_t first, int64_t count,
                            float *vectors_out);
JV_API int jv_hip_direct_scores(jv_ctx *ctx, const jv_codes *codes, const float *queries, int Q, jv_vsf vsf,
                                const int32_t *ordinals, int B, float *scores_out);

/* GraphSearcher.search(scoreProvider, topK, threshold = 0, acceptOrds) (GraphSearcher.java:222-243): the Bits filter as a
 * little-endian bit array over node ids — bit n of 64-bit word n / 64 set = node n may be RETURNED.  Filtered-out nodes are
 * still traversed (their neighbours are scored and expanded); only layer 0 consults the filter (upper layers run with
 * Bits.ALL, :276).  accept_stride_words = 0: one mask for the whole batch; otherwise query q uses the words starting at
 * accept_bits + q * accept_stride_words (>= ceil(n_nodes / 64)).  accept_bits may be host or device memory; NULL = Bits.ALL,
 * i.e. jv_hip_graph_search. */
JV_API int jv_hip_graph_search_filtered(jv_ctx *ctx, const jv_graph *g, jv_luts *luts, const jv_codes *codes,
                                        const jv_fused *fused, const jv_vectors *vectors, const float *queries, int Q,
                                        jv_vsf vsf, int topK, int rerankK, const uint64_t *accept_bits,
                                        int64_t accept_stride_words, int32_t *out_ids, float *out_scores, int64_t *stats);

/* GraphSearcher OBJECTS — every option of GraphSearcher.search(scoreProvider, topK, rerankK, threshold, rerankFloor, acceptOrds)
 * (GraphSearcher.java:222-243) and resume(additionalK, rerankK) (:538-547), for a batch of Q independent searchers that keep
 * their candidate queue, visited set, evictedResults and CachingReranker (:554-581) between calls.
 *   threshold > 0   : layer 0 admits only nodes whose approximate similarity reaches it (:437) and stops early through
 *                     ScoreTracker.TwoPhaseTracker (ScoreTracker.java:80-140; commons-math3 3.6.1's LEGACY percentile);
 *                     typically used with a large topK = rerankK to find (approximately) everything above the threshold.
 *   rerankFloor     : only approximate results >= rerankFloor are scored exactly (the best one if none is) and can be
 *                     returned (NodeQueue.rerank, NodeQueue.java:160-230); the others wait in evictedResults for a resume.
 *   resume          : continue layer 0 where the last search / resume of this object stopped and return the next
 *                     additionalK results (never a node returned before); threshold = rerankFloor = 0 (:538-547).
 * Outputs: out_ids / out_scores Q x topK best first, (-1, -inf) padded; out_counts (nullable) Q results per query;
 * stats (nullable) Q x 4 int64 = {visitedCount, expandedCount, expandedCountBaseLayer, rerankedCount};
 * worst_approx (nullable) Q floats = worstApproximateScoreInTopK (+inf when fewer than topK results or no reranker).
 * Exact-score ties at the K-th place resolve as in the reference (its result heap's array order).
 * Where they run: search() runs on the DEVICE traversal wherever its session kernels apply (256-cluster codebooks — specialised builds at M = 16 … 192, a generic one otherwise —, degree <= 512, the rerankK results fit LDS; graph traversal not pinned to the host): threshold admission, the
 * TwoPhaseTracker stop and acceptOrds inside the kernel, then the host rebuilds approximateResults' heap array from the kernel's
 * addTopCandidate log and runs the reference's rerank (floor, caching reranker, worst approximate score).  resume() needs the
 * candidate queue / visited set of every searcher, which never left the device: the session kernel replays the searcher's earlier
 * calls (the traversal is deterministic; up to 7 of them) and continues in the same launch; a longer history, or a query that
 * outgrows the device structures, replays on the host batched searcher instead.  Every other shape runs both calls on the host
 * searcher.  Counters: gs_session_calls_device / gs_session_resume_device / gs_session_resume_replays /
 * gs_session_calls_host_overflow / _unsupported.
 * graph / luts / codes / fused / vectors must outlive the object; one object serves one batch at a time (a new search() discards the previous state).  accept_bits as in
 * jv_hip_graph_search_filtered (copied: resume uses the same filter).  Buffers may be host or device memory. */
typedef struct jv_searcher jv_searcher;
JV_API int jv_hip_searcher_create(jv_ctx *ctx, const jv_graph *g, jv_luts *luts, const jv_codes *codes, const jv_fused *fused,
                                  const jv_vectors *vectors, jv_searcher **out);
JV_API int jv_hip_searcher_search(jv_ctx *ctx, jv_searcher *s, const float *queries, int Q, jv_vsf vsf, int topK, int rerankK,
                                  float threshold, float rerankFloor, const uint64_t *accept_bits, int64_t accept_stride_words,
                                  int32_t *out_ids, float *out_scores, int32_t *out_counts, int64_t *stats, float *worst_approx);
JV_API int jv_hip_searcher_resume(jv_ctx *ctx, jv_searcher *s, int additionalK, int rerankK, int32_t *out_ids, float *out_scores,
                                  int32_t *out_counts, int64_t *stats, float *worst_approx);
JV_API int jv_hip_searcher_destroy(jv_searcher *s);

/* ---------------------------------------------------------------------------------------------
 * Sharded index (BASELINE config 4; SURVEY §8b "jv_hip_sharded_topk (RCCL)", §8e): PQ codes and base vectors are
 * partitioned by contiguous ordinal range (the analogue of PQVectors' chunking, B/quantization/PQVectors.java:515-540),
 * ONE RANK PER GPU — a process, or one host thread of a JVM, each with its own jv_ctx.  The reference has no sharded
 * search; the contract is equality with the single index: ids and scores bit-identical to jv_hip_search_flat over the
 * concatenated index, engineered score ties included (NodeQueue order, B/graph/NodeQueue.java:125-129).
 * Collectives: one all-gather of the partial top-rerankK (Q x rerankK x (i32, f32) per shard) and one all-gather of the
 * owners' exact scores — RCCL over xGMI, bound at run time (dlopen librccl.so; JVECTOR_HIP_RCCL_PATH overrides).
 *
 *   jv_hip_comm_unique_id : rank 0 makes the 128-byte rendezvous id (ncclGetUniqueId); the HOST distributes it to the other
 *                           ranks by its own means (Java: a shared field; torch.distributed: a broadcast).
 *   jv_hip_comm_create    : collective over all ranks (ncclCommInitRank) on ctx's device.  id == NULL with world == 1 makes a
 *                           purely local communicator (no RCCL loaded): all shards live on this context.
 *   jv_hip_comm_count     : ncclCommCount of the communicator's RCCL object — how many ranks RCCL itself says it joined (1 for a
 *                           local communicator).  A launcher uses it to verify that an N-GPU job really is N RCCL ranks.
 *   jv_hip_comm_all_gather: all-gather of `bytes` opaque bytes per rank (host or device buffers; recv holds world x bytes,
 *                           rank-major) over the communicator, blocking — for small host-side records (per-rank timings,
 *                           shard tables), not a data-path primitive.
 *   jv_hip_sharded_topk   : every rank hands its partial list (Q x k_in scores + GLOBAL ids, host or device); every rank
 *                           receives the same merged top-k_out (best first; (-1, -inf) padded).
 *   jv_hip_sharded_search_flat : the whole two-pass search.  This rank holds n_local shards (normally 1; every rank the same
 *                           number): codes[s] (+ vectors[s], or vectors == NULL for no rerank) own global ordinals
 *                           [id_base[s], id_base[s] + count).  luts: capacity >= Q, same jv_pq on every rank.
 *                           Every rank receives the identical Q x topK result.  The call opens with one small all-gather
 *                           of {n_local, Q, topK, rerankK, rerank?, vsf, D, argument status, shard ranges}: every rank takes
 *                           the same decisions from the same table, so ranks that disagree (one shard without vectors, a
 *                           different Q, a rejected argument) make EVERY rank return an error instead of hanging in a
 *                           collective the others never issue.  (A HIP failure inside one rank's scan is not covered.)
 * ------------------------------------------------------------------------------------------- */
/*   jv_hip_comm_create_external : a communicator whose all-gathers are carried by the HOST's transport instead of RCCL — gloo, MPI, a
 *                           JVM's own channel between its per-GPU threads: all_gather_fn(user, send, bytes, recv) must fill recv
 *                           (world x bytes, rank-major, host memory) with every rank's `bytes` of send and return 0; the library
 *                           stages its device buffers through host memory around the call.  Everything else — agreement header,
 *                           NodeQueue-order merge, owner selection, kernels — is the same code as over RCCL.
 *   jv_hip_sharded_merge_rerank : the exchange of jv_hip_sharded_search_flat for partial lists the CALLER produced (one graph index
 *                           per shard, the way JVector deployments shard): part_ids / part_scores [n_local][Q][rerankK] with GLOBAL
 *                           ids (host or device), counts[s] ordinals owned from id_base[s]; vectors == NULL: no exact rerank.
 *                           An id of list s outside [id_base[s], id_base[s] + counts[s]) is dropped ((-1, -inf)) before the merge.
 *                           After the agreement header a rank-local failure (a HIP error in one rank's scan, a non-zero return of
 *                           the external all-gather on one rank) is NOT recoverable: the other ranks are inside the next collective.
 *                           Such failures must be collective — tear the communicator down on every rank.  */
#define JV_COMM_ID_BYTES 128
typedef struct jv_comm jv_comm;
typedef int (*jv_all_gather_fn)(void *user, const void *send, size_t bytes, void *recv);
JV_API int jv_hip_comm_unique_id(uint8_t *id_out /* JV_COMM_ID_BYTES */);
JV_API int jv_hip_comm_create(jv_ctx *ctx, const uint8_t *id, int rank, int world, jv_comm **out);
JV_API int jv_hip_comm_create_external(jv_ctx *ctx, int rank, int world, jv_all_gather_fn all_gather_fn, void *user, jv_comm **out);
JV_API int jv_hip_comm_destroy(jv_comm *comm);
JV_API int jv_hip_comm_rank(const jv_comm *comm);
JV_API int jv_hip_comm_world(const jv_comm *comm);
JV_API int jv_hip_comm_count(const jv_comm *comm, int *out);
JV_API int jv_hip_comm_all_gather(jv_ctx *ctx, jv_comm *comm, const void *send, size_t bytes, void *recv);
JV_API int jv_hip_sharded_topk(jv_ctx *ctx, jv_comm *comm, const float *scores, const int32_t *ids, int Q, int k_in, int k_out,
                               int32_t *out_ids, float *out_scores);
JV_API int jv_hip_sharded_search_flat(jv_ctx *ctx, jv_comm *comm, int n_local, jv_luts *luts, const jv_codes *const *codes,
                                      const jv_vectors *const *vectors, const int64_t *id_base, const float *queries, int Q,
                                      jv_vsf vsf, int topK, int rerankK, int32_t *out_ids, float *out_scores);

JV_API int jv_hip_sharded_merge_rerank(jv_ctx *ctx, jv_comm *comm, int n_local, jv_luts *luts, const jv_vectors *const *vectors,
                                       const int64_t *id_base, const int64_t *counts, const float *queries, int Q, jv_vsf vsf, int topK,
                                       int rerankK, const int32_t *part_ids, const float *part_scores, int32_t *out_ids, float *out_scores);

#ifdef __cplusplus
}
#endif
#endif /* JVECTOR_HIP_H */
