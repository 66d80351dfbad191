/*
 * jv_oracle.h — CPU oracle for the JVector distance / quantization hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library.  The product
 * (jvector_amd/, libjvector_hip.so) never links, imports or calls it.
 *
 * It is a plain-C restatement of the *scalar* reference implementation
 * (DefaultVectorUtilSupport + ProductQuantization + PQDecoder + FusedPQDecoder +
 * NodeQueue ordering) with Java's floating-point semantics reproduced exactly:
 * strict IEEE-754 binary32, left-to-right evaluation, NO fused multiply-add
 * (compile with -ffp-contract=off, never -ffast-math).
 *
 * Path abbreviations in citations (relative to /root/reference):
 *   B/  = jvector-base/src/main/java/io/github/jbellis/jvector/
 *   NC/ = jvector-native/src/main/native/
 *   TS/ = jvector-tests/src/test/java/io/github/jbellis/jvector/
 *
 * Parity pin status:
 *   dot / L2 / cosine ........ pinned by the reference's native known-answer generator
 *                              (NC/tests/test_helpers.cpp:78-87 make_vec + 19 lengths, tol 1e-4 rel)
 *   PQ file format ........... pinned by the binary fixture jvector-tests/resources/version0.pq
 *                              (TS/quantization/TestProductQuantization.java:215-248, byte-exact re-save)
 *   PQLayout chunk math ...... pinned by the literal table in TestProductQuantization.java:305-340
 *   ADC / LUT / encode ....... pinned only through restated Java-test *properties*
 *                              (ADC == direct 1e-6, perfect reconstruction, fused == unfused);
 *                              the reference holds no literal golden values for these and cannot be
 *                              run here (no JDK, Highway submodule empty) => "parity unpinned at
 *                              the literal-value level" for PQ code bytes and ADC sums.
 */
#ifndef JV_ORACLE_H
#define JV_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* VectorSimilarityFunction ordinal order: B/vector/VectorSimilarityFunction.java:34-69 */
enum { JVO_EUCLIDEAN = 0, JVO_DOT_PRODUCT = 1, JVO_COSINE = 2 };

/* ---- row 1: full-resolution distances (DefaultVectorUtilSupport) ---- */
float jvo_dot(const float *a, const float *b, int n);
float jvo_dot_off(const float *a, int aoff, const float *b, int boff, int n);
float jvo_l2(const float *a, const float *b, int n);
float jvo_l2_off(const float *a, int aoff, const float *b, int boff, int n);
float jvo_cosine(const float *a, const float *b, int n);
float jvo_cosine_off(const float *a, int aoff, const float *b, int boff, int n);
/* VectorSimilarityFunction.compare: raw distance -> similarity score */
float jvo_compare(int vsf, const float *a, const float *b, int n);
float jvo_score_from_raw(int vsf, float raw);
void  jvo_sub(const float *a, const float *b, float *out, int n);

/* ---- rows 2,5,6: ADC tables and lookups ---- */
float jvo_assemble_and_sum(const float *data, int dataBase, const uint8_t *offs, int offsOff, int len);
float jvo_assemble_and_sum_pq(const float *tri, int M, const uint8_t *c1, int o1,
                              const uint8_t *c2, int o2, int k);
void  jvo_calculate_partial_sums(const float *codebook, int cbIndex, int size, int k,
                                 const float *query, int qoff, int vsf, float *out);
void  jvo_calculate_partial_self_magnitudes(const float *codebook, int cbIndex, int size, int k,
                                            float *out);
float jvo_pq_decoded_cosine(const uint8_t *enc, int encOff, int encLen, int k,
                            const float *partialSums, const float *aMag, float bMag);

/* ---- ProductQuantization as flat arrays ----
 * codebooks: concatenation over m of k*size_m floats (centroid-major), exactly the
 * order ProductQuantization.write emits them (B/quantization/ProductQuantization.java:593-598). */
typedef struct {
    int D, M, k;
    const int *sizes;      /* [M] */
    const int *offsets;    /* [M] */
    const float *codebooks;/* [sum_m k*sizes[m]] */
    const float *centroid; /* [D] or NULL */
    const float *self_magnitudes; /* optional [M*k]: the cached partialSquaredMagnitudes table (ProductQuantization.java:75,
                                   * 238) for the SEARCH entry points; NULL = rebuild it per query (same values) */
} jvo_pq;

void jvo_subvector_sizes_offsets(int D, int M, int *sizes, int *offsets);
int  jvo_closest_centroid(const jvo_pq *pq, const float *vec /*already centred*/, int m);
void jvo_pq_encode(const jvo_pq *pq, const float *vec, uint8_t *dst);
/* PQ training with a seeded RNG substituted for ThreadLocalRandom (see jv_oracle.c) */
uint64_t jvo_kmeans_stream(uint64_t seed, int m);
void jvo_kmeans_pp_init(const float *X, int64_t n, int stride, int off, int len, int k, uint64_t *rng, float *C);
int  jvo_kmeans_lloyd(const float *X, int64_t n, int stride, int off, int len, int k, float *C, int rounds, uint64_t *rng);
void jvo_centroid_of(const float *X, int64_t n, int D, float *out);
void jvo_pq_train(const float *X, int64_t n, int D, int M, int k, int globallyCenter, uint64_t seed, int rounds,
                  float *codebooks, float *centroid, int *rounds_run);
void jvo_pq_refine(const jvo_pq *pq, const float *X, int64_t n, int rounds, uint64_t seed, float *codebooks);
void jvo_pq_train_aniso(const float *X, int64_t n, int D, int M, int k, int globallyCenter, float threshold, uint64_t seed, int rounds,
                        float *codebooks, float *centroid, int *rounds_run);
void jvo_pq_refine_aniso(const jvo_pq *pq, float threshold, const float *X, int64_t n, int rounds, uint64_t seed, float *codebooks);
int  jvo_nodequeue_push(int64_t *heap, int *size, int cap, int order, int32_t node, float score);
void jvo_nodequeue_top(const int64_t *heap, int order, int32_t *node, float *score);
void jvo_nodequeue_pop(int64_t *heap, int *size);
float jvo_parallel_cost_multiplier(float threshold, int dimensions);
void jvo_pq_encode_anisotropic(const jvo_pq *pq, float threshold, const float *vec, uint8_t *dst);
void jvo_pq_encode_all(const jvo_pq *pq, const float *vecs, int64_t n, uint8_t *dst, int nthreads);
void jvo_pq_decode(const jvo_pq *pq, const uint8_t *code, float *dst);
/* createCodebookPartialSums: M*k*(k+1)/2 floats */
void jvo_pq_codebook_partial_sums(const jvo_pq *pq, int vsf, float *out);
float jvo_pq_diversity_score(const float *tri, int M, int k, int vsf, const uint8_t *code1, const uint8_t *code2);
float jvo_pq_diversity_score_direct(const jvo_pq *pq, int vsf, const uint8_t *code1, const uint8_t *code2);
/* VamanaDiversityProvider.retainDiverse with the PQ diversity score; selected: n bytes out; returns nSelected */
int jvo_retain_diverse(const float *tri, int M, int k, int vsf, const uint8_t *codes, const int32_t *nodes, const float *scores,
                       int n, int maxDegree, int diverseBefore, float alpha, uint8_t *selected, double *short_edges);

/* PQDecoder (precomputedScoreFunctionFor): builds LUT (M*k floats), for cosine also the
 * aMagnitude table and bMagnitude.  lut/amag caller-allocated; amag/bmag may be NULL unless cosine. */
void  jvo_pqdecoder_init(const jvo_pq *pq, const float *query, int vsf,
                         float *lut, float *amag, float *bmag);
/* FusedPQDecoder variant: identical tables, but the cosine query magnitude is accumulated
 * per subspace (B/quantization/FusedPQDecoder.java:178-191). */
void  jvo_fuseddecoder_init(const jvo_pq *pq, const float *query, int vsf,
                            float *lut, float *amag, float *bmag);
/* similarityTo(node): score of one code row under the prepared tables */
float jvo_adc_score(int vsf, int M, int k, const float *lut, const float *amag, float bmag,
                    const uint8_t *code);
/* batched convenience: scores[i] = jvo_adc_score(codes + ord[i]*M) ; ord==NULL => i */
void  jvo_adc_scores(int vsf, int M, int k, const float *lut, const float *amag, float bmag,
                     const uint8_t *codes, const int32_t *ord, int64_t n, float *scores);
/* PQVectors.scoreFunctionFor (non-precomputed, decode-free direct path):
 * B/quantization/PQVectors.java:222-280 — used by the ADC==direct property test */
float jvo_pq_direct_score(const jvo_pq *pq, const float *query, int vsf, const uint8_t *code);

/* ---- row 9: NodeQueue total order ---- */
int32_t jvo_float_to_sortable_int(float v);
float   jvo_sortable_int_to_float(int32_t v);
int64_t jvo_nodequeue_encode(int32_t node, float score);
/* top-k of (ids[i], scores[i]) under the NodeQueue order, best first.  returns count written */
int     jvo_topk(const int32_t *ids, const float *scores, int64_t n, int k,
                 int32_t *out_ids, float *out_scores);

/* CPU-baseline driver: two-pass flat search (ADC scan of all codes -> top rerankK -> exact rerank -> topK),
 * one query per worker thread.  vecs may be NULL (then no rerank).  Used by bench.py's cpu_baseline leg and
 * by tests as the end-to-end checker. */
void jvo_search_flat(const jvo_pq *pq, const uint8_t *codes, const float *vecs, int64_t n, const float *queries,
                     int Q, int vsf, int topK, int rerankK, int32_t *out_ids, float *out_scores, int nthreads);

/* ---- GraphSearcher restatement (checker for the host batched searcher) ----
 * Multi-level graph: level l has level_count[l] nodes (level_nodes[l] sorted ascending; NULL = all nodes, level 0),
 * rows of level_degree[l] neighbour ids, packed, padded with -1. */
typedef struct {
    int64_t n_nodes;
    int n_levels;
    int32_t entry_node;
    int entry_level;
    const int *level_count;
    const int *level_degree;
    const int32_t *const *level_nodes;
    const int32_t *const *level_neighbors;
} jvo_graph;
void jvo_graph_search(const jvo_graph *g, const jvo_pq *pq, const uint8_t *codes, const float *vecs,
                      const float *query, int vsf, int fused, int topK, int rerankK,
                      int32_t *out_ids, float *out_scores, int64_t *stats /* [visited, expanded] or NULL */);
void jvo_graph_search_filtered(const jvo_graph *g, const jvo_pq *pq, const uint8_t *codes, const float *vecs,
                               const float *query, int vsf, int fused, int topK, int rerankK, const uint64_t *accept,
                               int32_t *out_ids, float *out_scores, int64_t *stats);

/* analysis aid: the calling thread's next jvo_graph_search* call records the ids of the nodes it scores, in order */
void jvo_set_visit_log(int32_t *buf, int64_t cap);
int64_t jvo_visit_log_count(void);

/* GraphSearcher as an object: threshold > 0 (TwoPhaseTracker), rerankFloor, resume(), rerankedCount and
 * worstApproximateInTopK (GraphSearcher.java:222-243,355-369,406-547; NodeQueue.java:160-230; ScoreTracker.java:38-140).
 * The graph / pq / codes / vecs pointers must outlive the searcher.  stats (nullable): {visitedCount, expandedCount,
 * expandedCountBaseLayer, rerankedCount}.  Both calls return the number of results written (<= topK; the rest of the topK
 * output slots are (-1, -inf)), or -1 for an illegal argument (rerankK < topK; resume before search). */
typedef struct jvo_searcher jvo_searcher;
jvo_searcher *jvo_searcher_new(const jvo_graph *g, const jvo_pq *pq, const uint8_t *codes, const float *vecs, int vsf,
                               int fused);
void jvo_searcher_free(jvo_searcher *s);
int jvo_searcher_search(jvo_searcher *s, const float *query, int topK, int rerankK, float threshold, float rerankFloor,
                        const uint64_t *accept, int32_t *out_ids, float *out_scores, int64_t *stats, float *worst_out);
int jvo_searcher_resume(jvo_searcher *s, int additionalK, int rerankK, int32_t *out_ids, float *out_scores, int64_t *stats,
                        float *worst_out);
/* org.apache.commons.math3 (3.6.1) StatUtils.percentile = Percentile, EstimationType.LEGACY, restated from its documentation */
double jvo_percentile_legacy(const double *values, int n, double p);

/* ---- GraphIndexBuilder with one thread (jv_oracle.c "GraphIndexBuilder, one thread"): java.util.Random(0) level draws, addGraphNode,
 * insertDiverse / backlink / enforceDegree, improveConnections, cleanup — the checker of the engine's builder in reference order ---- */
void   jvo_java_random_seed(int64_t *state, int64_t seed);
double jvo_java_random_next_double(int64_t *state);
int    jvo_random_graph_level(int64_t *state, int degree0, int addHierarchy);
typedef struct jvo_builder jvo_builder;
jvo_builder *jvo_builder_new(const jvo_pq *pq, const uint8_t *codes, const float *vecs, int64_t n, int vsf, int maxDegree, int beamWidth,
                             float alpha, float neighborOverflow, int addHierarchy, int refineFinalGraph);
void jvo_builder_free(jvo_builder *b);
void jvo_builder_set_levels(jvo_builder *b, const int8_t *levels);
void jvo_builder_set_deviations(jvo_builder *b, int dedupe_ids, int full_vectors, int sorted_candidates);
int  jvo_builder_add(jvo_builder *b, int32_t node);
void jvo_builder_improve(jvo_builder *b, int32_t node);
void jvo_builder_enforce_degree(jvo_builder *b, int32_t node);
void jvo_builder_cleanup(jvo_builder *b);
int  jvo_builder_row(const jvo_builder *b, int level, int32_t node, int32_t *ids, float *scores, int *diverseBefore);
void jvo_builder_info(const jvo_builder *b, int32_t *entry_node, int *entry_level, int *n_levels, int64_t *reprunes);
int  jvo_nodearray_insert_sorted(int32_t *nodes, float *scores, int *size, int32_t node, float score);
int  jvo_nodearray_merge(const int32_t *n1, const float *s1, int size1, const int32_t *n2, const float *s2, int size2, int32_t *out_n, float *out_s);

/* exact rerank of pre-gathered candidate rows (Q x R x D), one query per worker thread */
void jvo_rerank(const float *queries, const float *cand_vecs, const int32_t *cand_ids, int Q, int R, int D, int vsf,
                int topK, int32_t *out_ids, float *out_scores, int nthreads);

/* ---- PQVectors.PQLayout chunk math: B/quantization/PQVectors.java:515-540 ---- */
typedef struct {
    int fullChunkVectors, lastChunkVectors, fullSizeChunks, totalChunks, fullChunkBytes, lastChunkBytes;
} jvo_pq_layout;
int jvo_pq_layout_compute(int vectorCount, int compressedDimension, jvo_pq_layout *out);

/* ---- ProductQuantization.load / write (big-endian) ----
 * Parses header; returns 0 on success.  Pointers into caller's output arrays. */
int  jvo_pq_parse(const uint8_t *buf, size_t len, int *version, int *D, int *M, int *k,
                  int *centroidLen, float *anisoThreshold,
                  int *sizes /*cap maxM*/, int maxM,
                  float *centroid /*cap D*/, float *codebooks /*cap*/, size_t codebookCap,
                  size_t *consumed);
size_t jvo_pq_serialize(int version, int D, int M, int k, const int *sizes, const float *centroid,
                        float anisoThreshold, const float *codebooks, uint8_t *out, size_t cap);

/* ---- the reference's native known-answer generator: NC/tests/test_helpers.cpp:78-87 ---- */
void jvo_make_vec(float *v, size_t n, float seed);

/* ---- cpu_baseline only (jv_oracle_simd.c): x86 SIMD restatement of the reference's native kernels.  NOT the parity
 * checker: results differ from the scalar functions above in the last bits (lane-wise FMA accumulation). ---- */
int   jvs_tier(void);            /* 0 scalar, 2 avx2+fma, 3 avx512f; JVO_SIMD_TIER=avx2|scalar caps it */
const char *jvs_tier_name(void);
float jvs_dot(const float *a, const float *b, int n);
float jvs_l2(const float *a, const float *b, int n);
float jvs_cosine(const float *a, const float *b, int n);
float jvs_compare(int vsf, const float *a, const float *b, int n);
void  jvs_calculate_partial_sums(const float *codebook, int cbIndex, int size, int k, const float *query, int qoff, int vsf,
                                 float *out);
float jvs_assemble_and_sum(const float *data, int dataBase, const uint8_t *offs, int len);
float jvs_pq_decoded_cosine(const uint8_t *offs, int len, int k, const float *lut, const float *amag, float bmag);
float jvs_adc_score(int vsf, int M, int k, const float *lut, const float *amag, float bmag, const uint8_t *code);
/* 0 = scalar arithmetic in jvo_search_flat / jvo_rerank / jvo_graph_search* (default; what parity tests use), non-zero =
 * the SIMD kernels above.  Returns the tier in effect.  Process-wide: set it before starting search threads. */
int   jvo_set_simd(int on);

/* ---- specification of the engine's MFMA tile form (ed_body.h): k-ascending fmaf chains; see jv_oracle.c ---- */
float jvo_dense_compare(int vsf, const float *q, const float *v, int n);
void  jvo_dense_scan(int vsf, const float *queries, int Q, const float *vecs, int64_t n, int D, float *out);

/* ---- NVQ (jv_nvq.c): the reference's compressed rerank codec — B/quantization/NVQuantization.java, NVQScorer.java,
 * DefaultVectorUtilSupport.java:385-548.  params per sub-vector = {minValue, maxValue, growthRate, midpoint} (the order
 * QuantizedSubVector.write serialises them); bytes = the sub-vectors' bytes concatenated (D per vector). ---- */
void  jvo_nvq_derive(float growthRate, float midpoint, float minValue, float maxValue, float levels, float *out5);
float jvo_nvq_min(const float *v, int n);
float jvo_nvq_max(const float *v, int n);
void  jvo_nvq_quantize_8bit(const float *v, int n, float growthRate, float midpoint, float minValue, float maxValue, uint8_t *dst);
float jvo_nvq_loss(const float *v, int n, float growthRate, float midpoint, float minValue, float maxValue, int nBits);
float jvo_nvq_uniform_loss(const float *v, int n, float minValue, float maxValue, int nBits);
float jvo_nvq_dot_8bit(const float *q, const uint8_t *bytes, int n, float growthRate, float midpoint, float minValue, float maxValue);
float jvo_nvq_l2_8bit(const float *q, const uint8_t *bytes, int n, float growthRate, float midpoint, float minValue, float maxValue);
void  jvo_nvq_cosine_8bit(const float *q, const uint8_t *bytes, int n, float growthRate, float midpoint, float minValue, float maxValue,
                          const float *centroid, float *out2);
float jvo_nvq_dequantize(uint8_t b, float growthRate, float midpoint, float minValue, float maxValue);
void  jvo_nvq_global_mean(const float *X, int64_t n, int D, float *out);
void  jvo_nvq_encode_sub(const float *v, int n, int learn, uint8_t *bytes, float *params);
int   jvo_nvq_growth_grid(float *coarse, float *fine, int *fine_n, int fine_stride);
void  jvo_nvq_encode(const float *mean, int D, int S, const float *vec, int learn, uint8_t *bytes, float *params);
void  jvo_nvq_encode_all(const float *mean, int D, int S, const float *X, int64_t n, int learn, uint8_t *bytes, float *params, int nthreads);
float jvo_nvq_score(int vsf, const float *mean, int D, int S, const float *query, const uint8_t *bytes, const float *params);
void  jvo_nvq_scores(int vsf, const float *mean, int D, int S, const float *queries, int Q, const uint8_t *bytes, const float *params,
                     int64_t n, const int32_t *ids, int B, float *out);
double jvo_nvq_reconstruction_error(const float *mean, int D, int S, const float *vec, int learn);
/* search entry points rerank with NVQ rows instead of `vecs` while this is set (bytes == NULL clears it) */
void  jvo_set_nvq_reranker(const uint8_t *bytes, const float *params, const float *mean, int D, int S);
int   jvo_nvq_reranker_active(void);
float jvo_nvq_rerank_score(int vsf, const float *query, int32_t node);

#ifdef __cplusplus
}
#endif
#endif
