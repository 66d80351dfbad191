/* jvector_formats.h — readers for the byte formats a real JVector produces, so that indexes / codebooks / code
 * tables written elsewhere can be handed to the device objects of jvector_hip.h without a JVM.
 *
 * HOST code only (no device is touched, callable on a machine without a GPU): every function takes a caller-owned
 * byte range (a read or mmap of the file) and either describes it or unpacks one section into caller-owned arrays,
 * converting the reference's BIG-endian ints / floats (B/disk/IndexWriter.java:36-42) to host order.
 * Return value: JV_OK or a jv_status error (message via jv_hip_last_error()); nothing is allocated or retained.
 *
 * Replaces, for ingestion only:
 *   ProductQuantization.load        B/quantization/ProductQuantization.java:649-693   (block length / shape)
 *   PQVectors.load                  B/quantization/PQVectors.java:54-75
 *   OnDiskGraphIndex.load           B/graph/disk/OnDiskGraphIndex.java:235-316  (header first, or footer for v5+)
 *     CommonHeader.load             B/graph/disk/CommonHeader.java:116-152
 *     Header.load                   B/graph/disk/Header.java:96-124
 *     L0 record layout              B/graph/disk/OnDiskGraphIndex.java:514-547
 *     in-memory upper layers        B/graph/disk/OnDiskGraphIndex.java:132-161
 *     v6 hierarchy source features  B/graph/disk/OnDiskGraphIndex.java:183-231
 *   SiftLoader.readFvecs / readIvecs  EX/util/SiftLoader.java:37-83   (little-endian dataset files)
 */
#ifndef JVECTOR_FORMATS_H
#define JVECTOR_FORMATS_H

#include <stddef.h>
#include <stdint.h>

#include "jvector_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- ProductQuantization block ------------------------------------------------------------------------------ */
/* Describe the PQ block that starts at buf (any version 0..6) without building it: its byte length and shape.
 * anisotropic_threshold is -1 when absent (version < 3) or isotropic. Any out pointer may be NULL. */
JV_API int jv_fmt_pq_describe(const uint8_t *buf, size_t len, size_t *block_len, int *version, int *D, int *M, int *k,
                              int *has_centroid, float *anisotropic_threshold);

/* ---- PQVectors ---------------------------------------------------------------------------------------------- */
/* [PQ block][int count][int M][count*M code bytes, ordinal-major; chunk boundaries are invisible on disk].
 * codes_off is the offset of the first code byte: pass buf+codes_off to jv_hip_codes_upload / PQVectors(...). */
JV_API int jv_fmt_pqvectors_describe(const uint8_t *buf, size_t len, size_t *pq_block_len, int64_t *count, int *M,
                                     size_t *codes_off);

/* ---- NVQuantization / NVQVectors ---------------------------------------------------------------------------- */
/* NVQuantization.write (B/quantization/NVQuantization.java:260-277): [int version][int D][D floats][int bitsPerDimension = 8]
 * [int S][S ints].  vector_stride = NVQuantization.compressedVectorSize (:357-363), the bytes of one serialised
 * QuantizedVector.  The stored sub-vector sizes must be the split NVQuantization.create(D, S) makes, else
 * JV_ERR_UNSUPPORTED.  Any out pointer may be NULL. */
JV_API int jv_fmt_nvq_describe(const uint8_t *buf, size_t len, size_t *block_len, int *version, int *D, int *n_subvectors,
                               size_t *mean_off, int64_t *vector_stride);
JV_API int jv_fmt_nvq_read_mean(const uint8_t *buf, size_t len, float *global_mean /* D floats, host order */);
/* NVQVectors.write (B/quantization/NVQVectors.java:50-62): [NVQuantization block][int count][count QuantizedVectors]. */
JV_API int jv_fmt_nvqvectors_describe(const uint8_t *buf, size_t len, size_t *nvq_block_len, int64_t *count, size_t *vectors_off,
                                      int64_t *vector_stride);
/* Decode `count` serialised QuantizedVectors (QuantizedVector.write :437-443, QuantizedSubVector.write :577-587), record r at
 * src + r * stride, into what jv_hip_nvq_vectors_upload takes:
 *   bytes  : count x D (the sub-vectors' bytes concatenated);  params : count x S x {minValue, maxValue, growthRate, midpoint} */
JV_API int jv_fmt_nvq_unpack(const uint8_t *src, size_t len, int64_t stride, int64_t count, int D, int n_subvectors, uint8_t *bytes,
                             float *params);

/* ---- OnDiskGraphIndex --------------------------------------------------------------------------------------- */
#define JV_ODGI_MAX_LAYERS 32 /* CommonHeader.V4_MAX_LAYERS */
#define JV_ODGI_MAX_FEATURES 8

/* FeatureId ordinals (B/graph/disk/feature/FeatureId.java:31-36) */
enum {
    JV_FEATURE_INLINE_VECTORS = 0,
    JV_FEATURE_FUSED_PQ = 1,
    JV_FEATURE_NVQ_VECTORS = 2,
    JV_FEATURE_SEPARATED_VECTORS = 3,
    JV_FEATURE_SEPARATED_NVQ = 4
};

typedef struct jv_odgi_info {
    int32_t version;        /* 2..6 */
    int32_t dimension;
    int32_t entry_node;     /* -1 = ENTRY_NODE_ABSENT */
    int32_t entry_level;    /* n_layers - 1 */
    int32_t id_upper_bound; /* number of L0 records (max ordinal + 1) */
    int32_t n_layers;
    int32_t layer_size[JV_ODGI_MAX_LAYERS];
    int32_t layer_degree[JV_ODGI_MAX_LAYERS];
    int32_t n_features;
    int32_t feature_id[JV_ODGI_MAX_FEATURES]; /* in record order (v6: header order; <= v5: FeatureId order) */
    int64_t header_off;          /* where the authoritative header was read (footer copy for v5+) */
    int64_t l0_off;              /* first L0 record */
    int64_t record_stride;       /* 4 + inline block + 4 * (1 + degree0) */
    int64_t inline_vectors_off;  /* offset of the D floats inside a record, -1 if the feature is absent */
    int64_t fused_off;           /* offset of the degree0*M code block inside a record, -1 if absent */
    int64_t neighbors_off;       /* offset of the int degree inside a record */
    int64_t pq_off, pq_len;      /* FusedPQ's header = a ProductQuantization block inside buf; -1/0 if absent */
    int32_t pq_M;                /* subspace count of that PQ (bytes per code), 0 if absent */
    int64_t upper_off;           /* first sparse-level record (levels 1..n_layers-1, back to back) */
    int64_t hierarchy_off;       /* v6 + FusedPQ: (int node, M code bytes) x hierarchy_count; -1 if absent */
    int32_t hierarchy_count;     /* layer_size[1], or 1 (the entry node) for a single-layer graph */
    int64_t separated_vectors_off; /* SEPARATED_VECTORS: id_upper_bound x D floats; -1 if absent */
    int64_t nvq_off, nvq_len;    /* NVQ_VECTORS / SEPARATED_NVQ header = an NVQuantization block inside buf; -1/0 if absent */
    int32_t nvq_S;               /* its sub-vector count, 0 if absent */
    int32_t reserved0;
    int64_t nvq_stride;          /* bytes of one serialised QuantizedVector (NVQuantization.compressedVectorSize) */
    int64_t nvq_inline_off;      /* NVQ_VECTORS: offset of the QuantizedVector inside a record, -1 if absent */
    int64_t separated_nvq_off;   /* SEPARATED_NVQ: id_upper_bound x nvq_stride bytes; -1 if absent */
} jv_odgi_info;

/* Parse the header(s) of the index that starts at buf[0] and spans len bytes (a slice that holds the index and nothing
 * else, as OnDiskGraphIndex.loadFromFooter requires) and validate every section against len.
 * NVQ features: header and offsets are reported here, rows come out of jv_fmt_odgi_read_nvq. */
JV_API int jv_fmt_odgi_describe(const uint8_t *buf, size_t len, jv_odgi_info *info);

/* Unpack layer 0. Any output may be NULL.
 *   neighbors : id_upper_bound x degree0 int32, packed, padded with -1 (what jv_hip_graph_set_level takes)
 *   vectors   : id_upper_bound x D float32 (inline, else separated vectors; JV_ERR_INVALID if neither is present)
 *   fused     : id_upper_bound x degree0 x M bytes (what jv_hip_fused_upload takes)
 * The stored ordinal is a writer-side sanity value the reference's reader never reads: a record is accepted when it
 * carries its own position (NodeRecordTask, and the sequential writer for live nodes) or -1, the sequential writer's
 * placeholder for an ordinal its OrdinalMapper OMITTED (OnDiskGraphIndexWriter.java:101-110; the feature bytes of such a
 * record are seek-skipped, i.e. unspecified) — a placeholder yields no neighbours and zeroed vector / fused bytes. */
JV_API int jv_fmt_odgi_read_l0(const uint8_t *buf, size_t len, const jv_odgi_info *info, int32_t *neighbors,
                               float *vectors, uint8_t *fused);

/* Unpack sparse level `level` (1..n_layers-1): node ids sorted ascending with their rows permuted alike
 * (jv_hip_graph_set_level wants ascending ids; the file order is the writer's iteration order).
 *   node_ids : layer_size[level] int32;  neighbors : layer_size[level] x layer_degree[level] int32, -1 padded */
JV_API int jv_fmt_odgi_read_level(const uint8_t *buf, size_t len, const jv_odgi_info *info, int level,
                                  int32_t *node_ids, int32_t *neighbors);

/* v6 + FusedPQ: the PQ codes of the hierarchy nodes (FusedPQ.loadSourceFeature, FusedPQ.java:214-220), file order.
 *   node_ids : hierarchy_count int32;  codes : hierarchy_count x M bytes */
JV_API int jv_fmt_odgi_read_hierarchy_codes(const uint8_t *buf, size_t len, const jv_odgi_info *info, int32_t *node_ids,
                                            uint8_t *codes);

/* NVQ_VECTORS (inline) or SEPARATED_NVQ rows of every L0 node: bytes id_upper_bound x D, params id_upper_bound x S x 4
 * (either may be NULL); a placeholder record yields a zeroed row. */
JV_API int jv_fmt_odgi_read_nvq(const uint8_t *buf, size_t len, const jv_odgi_info *info, uint8_t *bytes, float *params);

/* ---- fvecs / ivecs ------------------------------------------------------------------------------------------ */
/* Little-endian rows of [int32 dim][dim x 4-byte element]; every row must carry the same dim. */
JV_API int jv_fmt_xvecs_describe(const uint8_t *buf, size_t len, int64_t *rows, int *dim);
/* out: rows x dim 4-byte elements (float32 for fvecs, int32 for ivecs), row headers stripped. */
JV_API int jv_fmt_xvecs_read(const uint8_t *buf, size_t len, void *out);

#ifdef __cplusplus
}
#endif
#endif /* JVECTOR_FORMATS_H */
