// formats.cpp — host-side readers for JVector's byte formats (include/jvector_formats.h).
// No device code: these run on a machine without a GPU and are covered by the "not gpu" tests.
// Everything JVector writes is BIG-endian (B/disk/IndexWriter.java:36-42); fvecs/ivecs datasets are little-endian.
#include <algorithm>
#include <numeric>

#include "../../include/jvector_formats.h"
#include "jv_internal.h"

using namespace jv;

namespace {

inline int32_t be32(const uint8_t *p)
{
    return (int32_t)(((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]);
}
inline int64_t be64(const uint8_t *p) { return (int64_t)(((uint64_t)(uint32_t)be32(p) << 32) | (uint32_t)be32(p + 4)); }
inline int32_t le32(const uint8_t *p)
{
    return (int32_t)(((uint32_t)p[3] << 24) | ((uint32_t)p[2] << 16) | ((uint32_t)p[1] << 8) | (uint32_t)p[0]);
}
// n big-endian 4-byte words -> host order (works for int32 and float32 alike)
inline void copy_be32(void *dst, const uint8_t *src, size_t n)
{
    uint32_t *d = (uint32_t *)dst;
    for (size_t i = 0; i < n; ++i) d[i] = (uint32_t)be32(src + 4 * i);
}

constexpr int32_t kPqMagic = 0x75EC4012;          // ProductQuantization.MAGIC (ProductQuantization.java:60)
constexpr int32_t kOdgiMagic = (int32_t)0xFFFF0D61;  // OnDiskGraphIndex.MAGIC (OnDiskGraphIndex.java:72)
constexpr int32_t kFooterMagic = 0x4a564244;      // AbstractGraphIndexWriter.FOOTER_MAGIC (:47)

// A cursor over a byte range that refuses to run past the end.
struct Cursor {
    const uint8_t *buf;
    size_t len, pos;
    bool ok(size_t nbytes) const { return nbytes <= len && pos <= len - nbytes; }
};

#define RD_NEED(c, nbytes, what) JV_REQUIRE((c).ok(nbytes), "%s: truncated input at byte %zu (need %zu more)", what, (c).pos, (size_t)(nbytes))

struct PqShape {
    size_t block_len;
    int version, D, M, k, has_centroid;
    float aniso;
};

// ProductQuantization.load :649-693 — walk the block, validate, do not copy the codebooks.
int pq_walk(const uint8_t *buf, size_t len, PqShape *s)
{
    Cursor c{buf, len, 0};
    RD_NEED(c, 4, "pq");
    int32_t maybeMagic = be32(buf + c.pos);
    c.pos += 4;
    int version = 0, gcl;
    if (maybeMagic != kPqMagic) {
        gcl = maybeMagic;  // versions 0-2 start with the global-centroid length
    } else {
        RD_NEED(c, 8, "pq");
        version = be32(buf + c.pos);
        gcl = be32(buf + c.pos + 4);
        c.pos += 8;
    }
    JV_REQUIRE(version >= 0 && version <= 6, "pq: unsupported version %d", version);
    JV_REQUIRE(gcl >= 0 && gcl < (1 << 24), "pq: implausible centroid length %d", gcl);
    RD_NEED(c, (size_t)gcl * 4, "pq");
    c.pos += (size_t)gcl * 4;
    RD_NEED(c, 4, "pq");
    int M = be32(buf + c.pos);
    c.pos += 4;
    JV_REQUIRE(M > 0 && M < (1 << 20), "pq: implausible M %d", M);
    RD_NEED(c, (size_t)M * 4, "pq");
    int64_t D = 0;
    for (int m = 0; m < M; ++m, c.pos += 4) {
        int sz = be32(buf + c.pos);
        JV_REQUIRE(sz > 0 && sz < (1 << 20), "pq: bad subvector size %d", sz);
        D += sz;
    }
    JV_REQUIRE(D < (1 << 24), "pq: implausible dimension %lld", (long long)D);
    float aniso = -1.0f;
    if (version >= 3) {
        RD_NEED(c, 4, "pq");
        int32_t b = be32(buf + c.pos);
        memcpy(&aniso, &b, 4);
        c.pos += 4;
    }
    RD_NEED(c, 4, "pq");
    int k = be32(buf + c.pos);
    c.pos += 4;
    JV_REQUIRE(k > 0 && k <= 65536, "pq: implausible cluster count %d", k);
    size_t cb = (size_t)k * (size_t)D * 4;
    RD_NEED(c, cb, "pq");
    c.pos += cb;
    JV_REQUIRE(gcl == 0 || gcl == D, "Global centroid length %d does not match vector dimensionality %lld", gcl, (long long)D);
    s->block_len = c.pos;
    s->version = version;
    s->D = (int)D;
    s->M = M;
    s->k = k;
    s->has_centroid = gcl > 0;
    s->aniso = aniso;
    return JV_OK;
}

struct NvqShape {
    size_t block_len, mean_off, sizes_off;
    int version, D, S;
    int64_t vector_stride;   // NVQuantization.compressedVectorSize :357-363
};

// NVQuantization.load :322-344 — [int version][int D][D floats][int bitsPerDimension][int S][S ints].  The device layout
// derives the sub-vector split from (D, S) as NVQuantization.create does (getSubvectorSizesAndOffsets :236-252, the only way
// the reference makes one); a file whose stored sizes differ from that split is refused.
int nvq_walk(const uint8_t *buf, size_t len, NvqShape *s)
{
    Cursor c{buf, len, 0};
    RD_NEED(c, 8, "nvq");
    s->version = be32(buf);
    const int D = be32(buf + 4);
    c.pos = 8;
    JV_REQUIRE(s->version >= 0 && s->version <= 6, "nvq: unsupported version %d", s->version);
    JV_REQUIRE(D > 0 && D < (1 << 24), "nvq: implausible global mean length %d", D);
    s->mean_off = c.pos;
    RD_NEED(c, (size_t)D * 4, "nvq");
    c.pos += (size_t)D * 4;
    RD_NEED(c, 8, "nvq");
    const int bits = be32(buf + c.pos), S = be32(buf + c.pos + 4);
    c.pos += 8;
    if (bits != 8) {   // BitsPerDimension.load :96-103
        set_error("Unsupported BitsPerDimension %d", bits);
        return JV_ERR_UNSUPPORTED;
    }
    JV_REQUIRE(S > 0 && S <= D, "nvq: %d sub-vectors for %d dimensions", S, D);
    s->sizes_off = c.pos;
    RD_NEED(c, (size_t)S * 4, "nvq");
    int64_t total = 0;
    s->vector_stride = 4;
    for (int i = 0; i < S; ++i, c.pos += 4) {
        const int sz = be32(buf + c.pos), want = D / S + (i < D % S ? 1 : 0);
        if (sz != want) {
            set_error("nvq: sub-vector %d has %d dimensions, NVQuantization.create(%d, %d) gives %d", i, sz, D, S, want);
            return JV_ERR_UNSUPPORTED;
        }
        total += sz;
        s->vector_stride += (int64_t)sz + 4 * 4 + 3 * 4;   // QuantizedSubVector.compressedVectorSize :497-503
    }
    JV_REQUIRE(total == D, "Global mean length %d does not match vector dimensionality %lld", D, (long long)total);
    s->D = D;
    s->S = S;
    s->block_len = c.pos;
    return JV_OK;
}

// QuantizedVector.load :449-458 / QuantizedSubVector.load :605-617 for `count` records `stride` bytes apart
int nvq_unpack(const uint8_t *src, size_t len, int64_t stride, int64_t count, int D, int S, uint8_t *bytes, float *params)
{
    JV_REQUIRE(stride > 0 && count >= 0 && (count == 0 || ((uint64_t)(count - 1) * (uint64_t)stride <= len)), "nvq_unpack: records out of range");
    for (int64_t r = 0; r < count; ++r) {
        Cursor c{src, len, (size_t)(r * stride)};
        RD_NEED(c, 4, "nvq vector");
        const int ns = be32(src + c.pos);
        c.pos += 4;
        JV_REQUIRE(ns == S, "nvq vector %lld has %d sub-vectors, the quantizer %d", (long long)r, ns, S);
        int off = 0;
        for (int i = 0; i < S; ++i) {
            const int want = D / S + (i < D % S ? 1 : 0);
            RD_NEED(c, 28, "nvq vector");
            const int bits = be32(src + c.pos);
            if (bits != 8) {
                set_error("Unsupported BitsPerDimension %d", bits);
                return JV_ERR_UNSUPPORTED;
            }
            if (params) copy_be32(params + (r * S + i) * 4, src + c.pos + 4, 4);   // minValue, maxValue, growthRate, midpoint
            const int od = be32(src + c.pos + 20), nb = be32(src + c.pos + 24);
            c.pos += 28;
            JV_REQUIRE(od == want && nb == want, "nvq vector %lld sub-vector %d: %d dimensions in %d bytes, expected %d", (long long)r, i, od,
                       nb, want);
            RD_NEED(c, (size_t)nb, "nvq vector");
            if (bytes) memcpy(bytes + r * D + off, src + c.pos, (size_t)nb);
            c.pos += (size_t)nb;
            off += nb;
        }
    }
    return JV_OK;
}

// CommonHeader.load :116-152
int common_header(Cursor &c, jv_odgi_info *o)
{
    RD_NEED(c, 4, "odgi header");
    int32_t maybeMagic = be32(c.buf + c.pos);
    c.pos += 4;
    int32_t size;
    if (maybeMagic == kOdgiMagic) {
        RD_NEED(c, 8, "odgi header");
        o->version = be32(c.buf + c.pos);
        size = be32(c.buf + c.pos + 4);
        c.pos += 8;
    } else {
        o->version = 2;
        size = maybeMagic;
    }
    JV_REQUIRE(o->version >= 2 && o->version <= 6, "odgi: unsupported version %d", o->version);
    RD_NEED(c, 12, "odgi header");
    o->dimension = be32(c.buf + c.pos);
    o->entry_node = be32(c.buf + c.pos + 4);
    int32_t degree0 = be32(c.buf + c.pos + 8);
    c.pos += 12;
    o->id_upper_bound = size;
    if (o->version < 4) {
        o->n_layers = 1;
        o->layer_size[0] = size;
        o->layer_degree[0] = degree0;
    } else {
        RD_NEED(c, 8 + 8 * JV_ODGI_MAX_LAYERS, "odgi header");
        o->id_upper_bound = be32(c.buf + c.pos);
        o->n_layers = be32(c.buf + c.pos + 4);
        c.pos += 8;
        JV_REQUIRE(o->n_layers >= 1 && o->n_layers <= JV_ODGI_MAX_LAYERS, "odgi: bad layer count %d", o->n_layers);
        for (int i = 0; i < JV_ODGI_MAX_LAYERS; ++i, c.pos += 8) {  // unused entries are zero padding
            o->layer_size[i] = i < o->n_layers ? be32(c.buf + c.pos) : 0;
            o->layer_degree[i] = i < o->n_layers ? be32(c.buf + c.pos + 4) : 0;
        }
    }
    o->entry_level = o->n_layers - 1;
    JV_REQUIRE(o->dimension > 0 && o->dimension < (1 << 24), "odgi: implausible dimension %d", o->dimension);
    JV_REQUIRE(o->id_upper_bound >= 0, "odgi: negative id upper bound");
    for (int i = 0; i < o->n_layers; ++i) {
        JV_REQUIRE(o->layer_size[i] >= 0 && o->layer_size[i] <= o->id_upper_bound, "odgi: layer %d size %d out of range", i,
                   o->layer_size[i]);
        JV_REQUIRE(o->layer_degree[i] >= 0 && o->layer_degree[i] < (1 << 20), "odgi: layer %d degree %d out of range", i,
                   o->layer_degree[i]);
    }
    JV_REQUIRE(o->entry_node >= -1 && o->entry_node < std::max(o->id_upper_bound, 1), "odgi: entry node %d out of range",
               o->entry_node);
    return JV_OK;
}

// Header.load :96-124 and the inline-offset bookkeeping of the OnDiskGraphIndex constructor (:91-116)
int full_header(Cursor &c, jv_odgi_info *o)
{
    JV_TRY(common_header(c, o));
    o->n_features = 0;
    if (o->version >= 6) {
        RD_NEED(c, 4, "odgi features");
        int nf = be32(c.buf + c.pos);
        c.pos += 4;
        JV_REQUIRE(nf >= 0 && nf <= JV_ODGI_MAX_FEATURES, "odgi: bad feature count %d", nf);
        o->n_features = nf;
    } else if (o->version >= 3) {
        RD_NEED(c, 4, "odgi features");
        int32_t flags = be32(c.buf + c.pos);  // FeatureId.deserialize :47-54
        c.pos += 4;
        JV_REQUIRE((flags & ~0x1F) == 0, "odgi: unknown feature flags 0x%x", flags);
        for (int n = 0; n < 5; ++n)
            if (flags & (1 << n)) o->feature_id[o->n_features++] = n;
    } else {
        o->feature_id[o->n_features++] = JV_FEATURE_INLINE_VECTORS;
    }
    int64_t inline_bytes = 0;
    o->inline_vectors_off = o->fused_off = -1;
    o->pq_off = -1;
    o->pq_len = 0;
    o->pq_M = 0;
    o->separated_vectors_off = -1;
    o->nvq_off = o->nvq_inline_off = o->separated_nvq_off = -1;
    o->nvq_len = o->nvq_stride = 0;
    o->nvq_S = 0;
    for (int i = 0; i < o->n_features; ++i) {
        if (o->version >= 6) {
            RD_NEED(c, 4, "odgi features");
            o->feature_id[i] = be32(c.buf + c.pos);
            c.pos += 4;
        }
        switch (o->feature_id[i]) {
        case JV_FEATURE_INLINE_VECTORS:  // no header (InlineVectors.java:45-64)
            JV_REQUIRE(o->inline_vectors_off < 0, "odgi: duplicate INLINE_VECTORS");
            o->inline_vectors_off = 4 + inline_bytes;
            inline_bytes += (int64_t)o->dimension * 4;
            break;
        case JV_FEATURE_FUSED_PQ: {  // header = the PQ block (FusedPQ.java:108-114,139-141)
            JV_REQUIRE(o->fused_off < 0, "odgi: duplicate FUSED_PQ");
            PqShape s;
            JV_TRY(pq_walk(c.buf + c.pos, c.len - c.pos, &s));
            JV_REQUIRE(s.D == o->dimension, "odgi: FusedPQ dimension %d != index dimension %d", s.D, o->dimension);
            o->pq_off = (int64_t)c.pos;
            o->pq_len = (int64_t)s.block_len;
            o->pq_M = s.M;
            c.pos += s.block_len;
            o->fused_off = 4 + inline_bytes;
            inline_bytes += (int64_t)s.M * o->layer_degree[0];
            break;
        }
        case JV_FEATURE_SEPARATED_VECTORS:  // header = long offset (SeparatedVectors.java:54-66,83-86)
            RD_NEED(c, 8, "odgi features");
            o->separated_vectors_off = be64(c.buf + c.pos);
            c.pos += 8;
            break;
        case JV_FEATURE_NVQ_VECTORS:      // header = the NVQuantization block (NVQ.java:64-76), inline = one QuantizedVector
        case JV_FEATURE_SEPARATED_NVQ: {  // header = the block + long offset (SeparatedNVQ.java:78-82,100-108)
            JV_REQUIRE(o->nvq_off < 0, "odgi: more than one NVQ feature");
            NvqShape ns;
            JV_TRY(nvq_walk(c.buf + c.pos, c.len - c.pos, &ns));
            JV_REQUIRE(ns.D == o->dimension, "odgi: NVQ dimension %d != index dimension %d", ns.D, o->dimension);
            o->nvq_off = (int64_t)c.pos;
            o->nvq_len = (int64_t)ns.block_len;
            o->nvq_S = ns.S;
            o->nvq_stride = ns.vector_stride;
            c.pos += ns.block_len;
            if (o->feature_id[i] == JV_FEATURE_NVQ_VECTORS) {
                o->nvq_inline_off = 4 + inline_bytes;
                inline_bytes += ns.vector_stride;
            } else {
                RD_NEED(c, 8, "odgi features");
                o->separated_nvq_off = be64(c.buf + c.pos);
                c.pos += 8;
            }
            break;
        }
        default:
            JV_REQUIRE(false, "odgi: unknown feature id %d", o->feature_id[i]);
        }
    }
    o->neighbors_off = 4 + inline_bytes;
    o->record_stride = 4 + inline_bytes + 4 * (1 + (int64_t)o->layer_degree[0]);
    return JV_OK;
}

bool range_ok(int64_t off, int64_t bytes, size_t len) { return off >= 0 && bytes >= 0 && (uint64_t)off + (uint64_t)bytes <= (uint64_t)len; }

int check_info(const uint8_t *buf, size_t len, const jv_odgi_info *o, const char *what)
{
    JV_REQUIRE(buf && o, "%s: NULL argument", what);
    JV_REQUIRE(o->n_layers >= 1 && o->n_layers <= JV_ODGI_MAX_LAYERS && o->id_upper_bound >= 0 && o->record_stride > 0 &&
                   range_ok(o->l0_off, o->record_stride * (int64_t)o->id_upper_bound, len),
               "%s: info does not describe this buffer (call jv_fmt_odgi_describe first)", what);
    return JV_OK;
}

}  // namespace

extern "C" {

int jv_fmt_pq_describe(const uint8_t *buf, size_t len, size_t *block_len, int *version, int *D, int *M, int *k,
                       int *has_centroid, float *anisotropic_threshold)
{
    clear_error();
    JV_REQUIRE(buf, "pq_describe: NULL buffer");
    PqShape s;
    JV_TRY(pq_walk(buf, len, &s));
    if (block_len) *block_len = s.block_len;
    if (version) *version = s.version;
    if (D) *D = s.D;
    if (M) *M = s.M;
    if (k) *k = s.k;
    if (has_centroid) *has_centroid = s.has_centroid;
    if (anisotropic_threshold) *anisotropic_threshold = s.aniso;
    return JV_OK;
}

int jv_fmt_pqvectors_describe(const uint8_t *buf, size_t len, size_t *pq_block_len, int64_t *count, int *M, size_t *codes_off)
{
    clear_error();
    JV_REQUIRE(buf, "pqvectors_describe: NULL buffer");
    PqShape s;
    JV_TRY(pq_walk(buf, len, &s));
    Cursor c{buf, len, s.block_len};
    RD_NEED(c, 8, "pqvectors");
    int32_t n = be32(buf + c.pos), cd = be32(buf + c.pos + 4);  // PQVectors.load :58-60
    c.pos += 8;
    JV_REQUIRE(n >= 0, "pqvectors: negative vector count %d", n);
    JV_REQUIRE(cd == s.M, "pqvectors: compressed dimension %d does not match the codebook's %d subspaces", cd, s.M);
    RD_NEED(c, (size_t)n * (size_t)cd, "pqvectors");
    if (pq_block_len) *pq_block_len = s.block_len;
    if (count) *count = n;
    if (M) *M = cd;
    if (codes_off) *codes_off = c.pos;
    return JV_OK;
}

int jv_fmt_odgi_describe(const uint8_t *buf, size_t len, jv_odgi_info *info)
{
    clear_error();
    JV_REQUIRE(buf && info, "odgi_describe: NULL argument");
    memset(info, 0, sizeof(*info));
    // OnDiskGraphIndex.load :248-268 — the leading header fixes where the L0 records start ...
    Cursor c{buf, len, 0};
    jv_odgi_info lead;
    memset(&lead, 0, sizeof(lead));
    JV_TRY(full_header(c, &lead));
    int64_t l0_off = (int64_t)c.pos;
    *info = lead;
    info->header_off = 0;
    if (lead.version >= 5) {
        // ... and from v5 on the authoritative copy is the one the footer points at (loadFromFooter :287-316):
        // [header'] [long headerOffset] [int FOOTER_MAGIC] at the very end of the slice.
        JV_REQUIRE(len >= 12, "odgi: too short for a footer");
        int32_t magic = be32(buf + len - 4);
        JV_REQUIRE(magic == kFooterMagic, "odgi: footer magic 0x%x does not match 0x%x", magic, kFooterMagic);
        int64_t hoff = be64(buf + len - 12);
        JV_REQUIRE(hoff >= 0 && (uint64_t)hoff < len - 12, "odgi: footer header offset %lld out of range", (long long)hoff);
        Cursor f{buf, len - 12, (size_t)hoff};
        jv_odgi_info foot;
        memset(&foot, 0, sizeof(foot));
        JV_TRY(full_header(f, &foot));
        JV_REQUIRE(f.pos == len - 12, "odgi: footer header ends at byte %zu, expected %zu", f.pos, len - 12);
        JV_REQUIRE(foot.record_stride == lead.record_stride, "odgi: leading and footer headers disagree on the record layout");
        *info = foot;
        info->header_off = hoff;
    }
    info->l0_off = l0_off;
    jv_odgi_info *o = info;
    int64_t l0_bytes = o->record_stride * (int64_t)o->id_upper_bound;
    JV_REQUIRE(range_ok(o->l0_off, l0_bytes, len), "odgi: %d L0 records of %lld bytes do not fit in %zu bytes", o->id_upper_bound,
               (long long)o->record_stride, len);
    // sparse levels: (int node, int count, degree x int) per node (loadInMemoryLayers :132-161)
    o->upper_off = o->l0_off + l0_bytes;
    int64_t upper_bytes = 0;
    for (int lvl = 1; lvl < o->n_layers; ++lvl) upper_bytes += 4 * (int64_t)o->layer_size[lvl] * (2 + (int64_t)o->layer_degree[lvl]);
    JV_REQUIRE(range_ok(o->upper_off, upper_bytes, len), "odgi: sparse levels run past the end of the input");
    // v6 + fused: source codes of the hierarchy nodes (loadInMemoryFeatures :183-231)
    o->hierarchy_off = -1;
    o->hierarchy_count = 0;
    if (o->version == 6 && o->fused_off >= 0) {
        o->hierarchy_off = o->upper_off + upper_bytes;
        o->hierarchy_count = o->n_layers >= 2 ? o->layer_size[1] : 1;
        JV_REQUIRE(range_ok(o->hierarchy_off, (int64_t)o->hierarchy_count * (4 + (int64_t)o->pq_M), len),
                   "odgi: hierarchy source features run past the end of the input");
    }
    if (o->separated_vectors_off >= 0)
        JV_REQUIRE(range_ok(o->separated_vectors_off, (int64_t)o->id_upper_bound * o->dimension * 4, len),
                   "odgi: separated vectors run past the end of the input");
    if (o->separated_nvq_off >= 0)
        JV_REQUIRE(range_ok(o->separated_nvq_off, (int64_t)o->id_upper_bound * o->nvq_stride, len),
                   "odgi: separated NVQ vectors run past the end of the input");
    return JV_OK;
}

int jv_fmt_nvq_describe(const uint8_t *buf, size_t len, size_t *block_len, int *version, int *D, int *n_subvectors, size_t *mean_off,
                        int64_t *vector_stride)
{
    clear_error();
    JV_REQUIRE(buf, "nvq_describe: NULL buffer");
    NvqShape s;
    JV_TRY(nvq_walk(buf, len, &s));
    if (block_len) *block_len = s.block_len;
    if (version) *version = s.version;
    if (D) *D = s.D;
    if (n_subvectors) *n_subvectors = s.S;
    if (mean_off) *mean_off = s.mean_off;
    if (vector_stride) *vector_stride = s.vector_stride;
    return JV_OK;
}

int jv_fmt_nvq_read_mean(const uint8_t *buf, size_t len, float *global_mean)
{
    clear_error();
    JV_REQUIRE(buf && global_mean, "nvq_read_mean: NULL argument");
    NvqShape s;
    JV_TRY(nvq_walk(buf, len, &s));
    copy_be32(global_mean, buf + s.mean_off, (size_t)s.D);
    return JV_OK;
}

int jv_fmt_nvqvectors_describe(const uint8_t *buf, size_t len, size_t *nvq_block_len, int64_t *count, size_t *vectors_off,
                               int64_t *vector_stride)
{
    clear_error();
    JV_REQUIRE(buf, "nvqvectors_describe: NULL buffer");
    NvqShape s;
    JV_TRY(nvq_walk(buf, len, &s));
    Cursor c{buf, len, s.block_len};
    RD_NEED(c, 4, "nvqvectors");
    const int32_t n = be32(buf + c.pos);   // NVQVectors.load :73-77
    c.pos += 4;
    JV_REQUIRE(n >= 0, "Invalid compressed vector count %d", n);
    JV_REQUIRE(range_ok((int64_t)c.pos, (int64_t)n * s.vector_stride, len), "nvqvectors: %d vectors of %lld bytes run past the end of the input",
               n, (long long)s.vector_stride);
    if (nvq_block_len) *nvq_block_len = s.block_len;
    if (count) *count = n;
    if (vectors_off) *vectors_off = c.pos;
    if (vector_stride) *vector_stride = s.vector_stride;
    return JV_OK;
}

int jv_fmt_nvq_unpack(const uint8_t *src, size_t len, int64_t stride, int64_t count, int D, int n_subvectors, uint8_t *bytes, float *params)
{
    clear_error();
    JV_REQUIRE(src, "nvq_unpack: NULL buffer");
    JV_REQUIRE(D > 0 && n_subvectors > 0 && n_subvectors <= D, "nvq_unpack: bad shape %d / %d", D, n_subvectors);
    return nvq_unpack(src, len, stride, count, D, n_subvectors, bytes, params);
}

int jv_fmt_odgi_read_nvq(const uint8_t *buf, size_t len, const jv_odgi_info *o, uint8_t *bytes, float *params)
{
    clear_error();
    JV_TRY(check_info(buf, len, o, "odgi_read_nvq"));
    JV_REQUIRE(o->nvq_off >= 0 && (o->nvq_inline_off >= 0 || o->separated_nvq_off >= 0), "odgi_read_nvq: the index has no NVQ feature");
    const int64_t N = o->id_upper_bound;
    if (o->nvq_inline_off < 0) {
        JV_REQUIRE(range_ok(o->separated_nvq_off, N * o->nvq_stride, len), "odgi_read_nvq: separated NVQ vectors out of range");
        // AbstractGraphIndexWriter.writeSeparatedFeatures writes featureSize ZERO bytes for every ordinal the OrdinalMapper omitted:
        // such a record (its level-0 record is the -1 placeholder, or — whatever the writer — every byte is zero, i.e. a sub-vector
        // count of 0) is a hole, not a damaged vector: zeroed row, like the inline branch below
        const int D = o->dimension, S = o->nvq_S;
        for (int64_t i = 0; i < N; ++i) {
            const uint8_t *rec = buf + o->separated_nvq_off + i * o->nvq_stride;
            bool hole = be32(buf + o->l0_off + i * o->record_stride) == -1;
            if (!hole) {
                hole = true;
                for (int64_t b = 0; b < o->nvq_stride && hole; ++b) hole = rec[b] == 0;
            }
            if (hole) {
                if (bytes) memset(bytes + i * D, 0, (size_t)D);
                if (params) memset(params + i * 4 * S, 0, sizeof(float) * 4 * (size_t)S);
                continue;
            }
            JV_TRY(nvq_unpack(rec, (size_t)o->nvq_stride, o->nvq_stride, 1, D, S, bytes ? bytes + i * D : nullptr,
                              params ? params + i * 4 * S : nullptr));
        }
        return JV_OK;
    }
    // inline: a placeholder record (ordinal -1, jv_fmt_odgi_read_l0) carries unspecified feature bytes -> zeroed row
    const int D = o->dimension, S = o->nvq_S;
    for (int64_t i = 0; i < N; ++i) {
        const uint8_t *rec = buf + o->l0_off + i * o->record_stride;
        if (be32(rec) == -1) {
            if (bytes) memset(bytes + i * D, 0, (size_t)D);
            if (params) memset(params + i * 4 * S, 0, sizeof(float) * 4 * (size_t)S);
            continue;
        }
        JV_TRY(nvq_unpack(rec + o->nvq_inline_off, (size_t)o->nvq_stride, o->nvq_stride, 1, D, S, bytes ? bytes + i * D : nullptr,
                          params ? params + i * 4 * S : nullptr));
    }
    return JV_OK;
}

int jv_fmt_odgi_read_l0(const uint8_t *buf, size_t len, const jv_odgi_info *o, int32_t *neighbors, float *vectors, uint8_t *fused)
{
    clear_error();
    JV_TRY(check_info(buf, len, o, "odgi_read_l0"));
    const int deg = o->layer_degree[0], D = o->dimension;
    const int64_t N = o->id_upper_bound;
    JV_REQUIRE(!fused || o->fused_off >= 0, "odgi_read_l0: the index has no FUSED_PQ feature");
    JV_REQUIRE(!vectors || o->inline_vectors_off >= 0 || o->separated_vectors_off >= 0,
               "odgi_read_l0: the index stores no full-resolution vectors");
    if (vectors && o->inline_vectors_off < 0) {
        JV_REQUIRE(range_ok(o->separated_vectors_off, N * D * 4, len), "odgi_read_l0: separated vectors out of range");
        copy_be32(vectors, buf + o->separated_vectors_off, (size_t)N * D);
    }
    const size_t fused_bytes = (size_t)deg * (size_t)o->pq_M;
    for (int64_t i = 0; i < N; ++i) {
        const uint8_t *rec = buf + o->l0_off + i * o->record_stride;
        // The stored ordinal is a writer-side sanity value the reference's reader never looks at.  The sequential
        // writer emits -1 for an ordinal the OrdinalMapper OMITTED (OnDiskGraphIndexWriter.java:101-110: feature bytes
        // seek-skipped = unspecified, count 0, -1 padding): such a record is a hole — no neighbours, zeroed features.
        int32_t ord = be32(rec);
        JV_REQUIRE(ord == (int32_t)i || ord == -1, "odgi_read_l0: record %lld carries ordinal %d", (long long)i, ord);
        if (ord == -1) {
            if (vectors && o->inline_vectors_off >= 0) memset(vectors + i * D, 0, (size_t)D * sizeof(float));
            if (fused) memset(fused + (size_t)i * fused_bytes, 0, fused_bytes);
            if (neighbors)
                for (int j = 0; j < deg; ++j) neighbors[i * deg + j] = -1;
            continue;
        }
        if (vectors && o->inline_vectors_off >= 0) copy_be32(vectors + i * D, rec + o->inline_vectors_off, (size_t)D);
        if (fused) memcpy(fused + (size_t)i * fused_bytes, rec + o->fused_off, fused_bytes);
        if (neighbors) {
            int32_t cnt = be32(rec + o->neighbors_off);
            JV_REQUIRE(cnt >= 0 && cnt <= deg, "odgi_read_l0: node %lld has %d neighbours, max degree %d", (long long)i, cnt, deg);
            int32_t *row = neighbors + i * deg;
            copy_be32(row, rec + o->neighbors_off + 4, (size_t)cnt);
            for (int j = 0; j < cnt; ++j)
                JV_REQUIRE(row[j] >= 0 && row[j] < N, "odgi_read_l0: node %lld neighbour %d out of range", (long long)i, row[j]);
            for (int j = cnt; j < deg; ++j) row[j] = -1;
        }
    }
    return JV_OK;
}

int jv_fmt_odgi_read_level(const uint8_t *buf, size_t len, const jv_odgi_info *o, int level, int32_t *node_ids, int32_t *neighbors)
{
    clear_error();
    JV_TRY(check_info(buf, len, o, "odgi_read_level"));
    JV_REQUIRE(level >= 1 && level < o->n_layers, "odgi_read_level: level %d outside 1..%d", level, o->n_layers - 1);
    JV_REQUIRE(node_ids && neighbors, "odgi_read_level: NULL output");
    int64_t off = o->upper_off;
    for (int lvl = 1; lvl < level; ++lvl) off += 4 * (int64_t)o->layer_size[lvl] * (2 + (int64_t)o->layer_degree[lvl]);
    const int n = o->layer_size[level], deg = o->layer_degree[level];
    const int64_t stride = 4 * (2 + (int64_t)deg);
    JV_REQUIRE(range_ok(off, stride * n, len), "odgi_read_level: level %d runs past the end of the input", level);
    // file order is the writer's iteration order; hand back ascending ids
    std::vector<int32_t> perm(n), ids(n);
    for (int i = 0; i < n; ++i) {
        ids[i] = be32(buf + off + i * stride);
        JV_REQUIRE(ids[i] >= 0 && ids[i] < o->id_upper_bound, "Node ID %d out of bounds for layer %d", ids[i], level);
    }
    std::iota(perm.begin(), perm.end(), 0);
    std::sort(perm.begin(), perm.end(), [&](int a, int b) { return ids[a] < ids[b]; });
    for (int r = 0; r < n; ++r) {
        const uint8_t *rec = buf + off + perm[r] * stride;
        JV_REQUIRE(r == 0 || ids[perm[r]] != ids[perm[r - 1]], "odgi_read_level: node %d listed twice in layer %d", ids[perm[r]], level);
        node_ids[r] = ids[perm[r]];
        int32_t cnt = be32(rec + 4);
        JV_REQUIRE(cnt >= 0 && cnt <= deg, "Node %d neighborCount %d > M %d", node_ids[r], cnt, deg);
        int32_t *row = neighbors + (int64_t)r * deg;
        copy_be32(row, rec + 8, (size_t)cnt);
        for (int j = 0; j < cnt; ++j)
            JV_REQUIRE(row[j] >= 0 && row[j] < o->id_upper_bound, "odgi_read_level: node %d neighbour %d out of range", node_ids[r], row[j]);
        for (int j = cnt; j < deg; ++j) row[j] = -1;
    }
    return JV_OK;
}

int jv_fmt_odgi_read_hierarchy_codes(const uint8_t *buf, size_t len, const jv_odgi_info *o, int32_t *node_ids, uint8_t *codes)
{
    clear_error();
    JV_TRY(check_info(buf, len, o, "odgi_read_hierarchy_codes"));
    JV_REQUIRE(o->hierarchy_off >= 0, "odgi_read_hierarchy_codes: the index carries no hierarchy source features");
    JV_REQUIRE(node_ids && codes, "odgi_read_hierarchy_codes: NULL output");
    const int64_t stride = 4 + (int64_t)o->pq_M;
    JV_REQUIRE(range_ok(o->hierarchy_off, stride * o->hierarchy_count, len), "odgi_read_hierarchy_codes: out of range");
    for (int i = 0; i < o->hierarchy_count; ++i) {
        const uint8_t *rec = buf + o->hierarchy_off + i * stride;
        node_ids[i] = be32(rec);
        JV_REQUIRE(node_ids[i] >= 0 && node_ids[i] < std::max(o->id_upper_bound, 1), "odgi_read_hierarchy_codes: node %d out of range",
                   node_ids[i]);
        memcpy(codes + (size_t)i * o->pq_M, rec + 4, (size_t)o->pq_M);
    }
    return JV_OK;
}

int jv_fmt_xvecs_describe(const uint8_t *buf, size_t len, int64_t *rows, int *dim)
{
    clear_error();
    JV_REQUIRE(buf || len == 0, "xvecs_describe: NULL buffer");
    if (len == 0) {
        if (rows) *rows = 0;
        if (dim) *dim = 0;
        return JV_OK;
    }
    JV_REQUIRE(len >= 4, "xvecs: truncated row header");
    int32_t d = le32(buf);  // SiftLoader.readFvecs :41-43
    JV_REQUIRE(d > 0 && d < (1 << 24), "xvecs: implausible dimension %d", d);
    size_t stride = 4 + (size_t)d * 4;
    JV_REQUIRE(len % stride == 0, "xvecs: %zu bytes is not a whole number of %d-dimensional rows", len, d);
    if (rows) *rows = (int64_t)(len / stride);
    if (dim) *dim = d;
    return JV_OK;
}

int jv_fmt_xvecs_read(const uint8_t *buf, size_t len, void *out)
{
    int64_t rows;
    int dim;
    JV_TRY(jv_fmt_xvecs_describe(buf, len, &rows, &dim));
    JV_REQUIRE(out || rows == 0, "xvecs_read: NULL output");
    size_t stride = 4 + (size_t)dim * 4;
    uint32_t *o = (uint32_t *)out;
    for (int64_t r = 0; r < rows; ++r) {
        const uint8_t *row = buf + (size_t)r * stride;
        JV_REQUIRE(le32(row) == dim, "xvecs: row %lld has dimension %d, expected %d", (long long)r, le32(row), dim);
        for (int j = 0; j < dim; ++j) o[(size_t)r * dim + j] = (uint32_t)le32(row + 4 + 4 * (size_t)j);
    }
    return JV_OK;
}

}  // extern "C"
