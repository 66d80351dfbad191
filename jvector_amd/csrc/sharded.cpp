// sharded.cpp — the sharded-index search behind the C ABI (SURVEY §8b export list: jv_hip_sharded_topk; §8e).
//
// PQ codes and base vectors are partitioned by contiguous ordinal range over the GPUs of one node (the analogue of
// PQVectors' chunking, B/quantization/PQVectors.java:515-540).  One RANK per GPU — a process (torch.distributed style) or
// a host thread of one JVM, each with its own jv_ctx; the only data-path exchange is
//   1. one all-gather of every shard's partial top-rerankK (Q x rerankK x (i32 global id, f32 score)) and
//   2. one all-gather of the owners' exact scores (Q x rerankK x f32),
// both latency-bound messages (Q = 1024, rerankK = 100: 0.8 MB per rank) that RCCL sends point-to-point over xGMI.
// The merge is the NodeQueue-order top-k (k_topk.hip) over the union — keys are unique because global ids are disjoint —
// so the result is bit-identical, ids and scores, to the single-index two-pass search (tests/test_sharded*.py).
// Step 2 is a SELECTION of the owner's value, not a MAX all-reduce: NaN / -inf exact scores arrive unchanged.
//
// RCCL is bound at run time (dlopen of librccl.so, path override JVECTOR_HIP_RCCL_PATH): the library keeps no link-time
// dependency on it, a process that never shards never loads it, and a host that already carries an RCCL (torch) shares it.
// comm == NULL means "all shards are local to this context" (single rank): the same code path minus the collectives.
#include <dlfcn.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "jv_internal.h"

namespace jv {

int launch_shard_interleave(hipStream_t s, const int32_t *d_ids, const float *d_sc, int P, int Q, int k, int32_t *d_out_ids,
                            float *d_out_sc);
int launch_shard_sanitize(hipStream_t s, int32_t *d_ids, float *d_sc, int64_t n, int64_t lo, int64_t hi);
int launch_shard_localize(hipStream_t s, const int32_t *d_gids, int64_t n, int64_t base, int64_t count, int32_t *d_local);
int launch_shard_select(hipStream_t s, const int32_t *d_gids, const float *d_exact, const long long *d_ranges, int P, int64_t n,
                        float *d_out);

namespace {

// ---- the eight RCCL entry points this file uses, declared here (rccl.h: ncclUniqueId is 128 opaque bytes passed BY VALUE;
//      ncclInt32 = 2, ncclInt64 = 4, ncclFloat32 = 7) ----
struct RcclId {
    char internal[JV_COMM_ID_BYTES];
};
typedef void *RcclComm;
struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(RcclId *) = nullptr;
    int (*CommInitRank)(RcclComm *, int, RcclId, int) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    int (*CommCount)(RcclComm, int *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, RcclComm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
constexpr int kNcclInt8 = 0, kNcclInt32 = 2, kNcclInt64 = 4, kNcclFloat32 = 7;

std::mutex g_rccl_mu;
Rccl g_rccl;

int load_rccl(const Rccl **out)
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (!g_rccl.handle) {
        const char *cands[] = {getenv("JVECTOR_HIP_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void *h = nullptr;
        for (const char *c : cands)
            if (c && *c && (h = dlopen(c, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) {
            set_error("sharded: cannot load RCCL (librccl.so; set JVECTOR_HIP_RCCL_PATH): %s", dlerror());
            return JV_ERR_UNSUPPORTED;
        }
        Rccl r;
        r.handle = h;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
        r.CommCount = (decltype(r.CommCount))dlsym(h, "ncclCommCount");
        r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
        r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.CommCount || !r.AllGather || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
            set_error("sharded: the RCCL library lacks a required symbol");
            dlclose(h);
            return JV_ERR_UNSUPPORTED;
        }
        g_rccl = r;
    }
    *out = &g_rccl;
    return JV_OK;
}

#define JV_RCCL_CHECK(r, expr)                                                                  \
    do {                                                                                        \
        const int _rc = (expr);                                                                 \
        if (_rc != 0) {                                                                         \
            set_error("RCCL: %s failed: %s", #expr, (r)->GetErrorString(_rc));                  \
            return JV_ERR_HIP;                                                                  \
        }                                                                                       \
    } while (0)

}  // namespace
}  // namespace jv

using namespace jv;

struct jv_comm {
    const Rccl *rccl = nullptr;
    RcclComm comm = nullptr;
    // an EXTERNAL transport instead of RCCL (jv_hip_comm_create_external): the host's own all-gather over host memory — gloo / MPI /
    // a JVM's channel; the merge, the owner selection and every kernel stay the library's
    jv_all_gather_fn ext_fn = nullptr;
    void *ext_user = nullptr;
    std::vector<char> ext_send, ext_recv;
    int rank = 0, world = 1, device = 0;
    std::vector<long long> h_ranges;  // host copy of the shard table (source of an async upload: must outlive the call)
    // device staging owned by the communicator (jv_hip_search_flat uses the context's scratch for itself)
    Buffer part_ids, part_sc, all_ids, all_sc, row_ids, row_sc, cand, cand_sc, local, exact, all_exact, ranges, all_ranges, bytes_in,
        bytes_out;
    ~jv_comm()
    {
        for (Buffer *b : {&part_ids, &part_sc, &all_ids, &all_sc, &row_ids, &row_sc, &cand, &cand_sc, &local, &exact, &all_exact, &ranges,
                          &all_ranges, &bytes_in, &bytes_out})
            b->release();
    }
};

namespace {

// all-gather `count` elements of type `dt` per rank; local communicator: a device copy
int all_gather(jv_ctx *ctx, jv_comm *c, const void *send, void *recv, size_t count, int dt, size_t elem)
{
    if (c->ext_fn) {   // the host's transport: device -> host, its all-gather, host -> device (latency-bound messages of < 1 MB)
        const size_t bytes = count * elem;
        c->ext_send.resize(bytes);
        c->ext_recv.resize(bytes * (size_t)c->world);
        JV_HIP_CHECK(hipMemcpyAsync(c->ext_send.data(), send, bytes, hipMemcpyDeviceToHost, ctx->stream));
        JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        const int rc = c->ext_fn(c->ext_user, c->ext_send.data(), bytes, c->ext_recv.data());
        if (rc != 0) {
            set_error("sharded: the external all-gather returned %d", rc);
            return JV_ERR_HIP;
        }
        JV_HIP_CHECK(hipMemcpyAsync(recv, c->ext_recv.data(), bytes * (size_t)c->world, hipMemcpyHostToDevice, ctx->stream));
        JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));   // ext_recv is reused by the next exchange
        return JV_OK;
    }
    if (!c->comm) {  // local communicator; an RCCL communicator of one rank still goes through RCCL
        if (send != recv) JV_HIP_CHECK(hipMemcpyAsync(recv, send, count * elem, hipMemcpyDeviceToDevice, ctx->stream));
        return JV_OK;
    }
    JV_RCCL_CHECK(c->rccl, c->rccl->AllGather(send, recv, count, dt, c->comm, ctx->stream));
    return JV_OK;
}

// merge the gathered pieces [P][Q][k] into [Q][k_out] under the NodeQueue order
int merge_pieces(jv_ctx *ctx, jv_comm *c, const int32_t *all_ids, const float *all_sc, int P, int Q, int k, int k_out, int32_t *d_out_ids,
                 float *d_out_sc)
{
    const size_t cells = (size_t)P * Q * k;
    JV_TRY(c->row_ids.reserve(sizeof(int32_t) * cells));
    JV_TRY(c->row_sc.reserve(sizeof(float) * cells));
    JV_TRY(launch_shard_interleave(ctx->stream, all_ids, all_sc, P, Q, k, (int32_t *)c->row_ids.ptr, (float *)c->row_sc.ptr));
    JV_TRY(ctx->d_scratch.reserve(topk_scratch_bytes(Q, k_out)));
    ProfScope ps(ctx, R_TOPK);
    return launch_topk(ctx->stream, ctx, (const float *)c->row_sc.ptr, (const int32_t *)c->row_ids.ptr, Q, (int64_t)P * k, (int64_t)P * k, 0,
                       k_out, d_out_ids, d_out_sc, ctx->d_scratch.ptr);
}

}  // namespace

extern "C" {

int jv_hip_comm_unique_id(uint8_t *id_out)
{
    clear_error();
    JV_REQUIRE(id_out, "comm_unique_id: NULL argument");
    const Rccl *r;
    JV_TRY(load_rccl(&r));
    RcclId id;
    JV_RCCL_CHECK(r, r->GetUniqueId(&id));
    memcpy(id_out, id.internal, JV_COMM_ID_BYTES);
    return JV_OK;
}

int jv_hip_comm_create(jv_ctx *ctx, const uint8_t *id, int rank, int world, jv_comm **out)
{
    clear_error();
    JV_REQUIRE(ctx && out, "comm_create: NULL argument");
    JV_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_create: rank %d outside world %d", rank, world);
    JV_REQUIRE(world == 1 || id, "comm_create: a communicator of %d ranks needs the unique id of jv_hip_comm_unique_id", world);
    JV_TRY(use_device(ctx->device));
    jv_comm *c = new jv_comm();
    c->rank = rank;
    c->world = world;
    c->device = ctx->device;
    if (id) {  // world 1 without an id: purely local communicator, RCCL is not even loaded
        int rc = load_rccl(&c->rccl);
        if (rc != JV_OK) {
            delete c;
            return rc;
        }
        RcclId rid;
        memcpy(rid.internal, id, JV_COMM_ID_BYTES);
        const int nrc = c->rccl->CommInitRank(&c->comm, world, rid, rank);
        if (nrc != 0) {
            set_error("RCCL: ncclCommInitRank(rank %d of %d) failed: %s", rank, world, c->rccl->GetErrorString(nrc));
            delete c;
            return JV_ERR_HIP;
        }
    }
    *out = c;
    return JV_OK;
}

int jv_hip_comm_create_external(jv_ctx *ctx, int rank, int world, jv_all_gather_fn all_gather_fn, void *user, jv_comm **out)
{
    clear_error();
    JV_REQUIRE(ctx && out && all_gather_fn, "comm_create_external: NULL argument");
    JV_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_create_external: rank %d outside world %d", rank, world);
    jv_comm *c = new jv_comm();
    c->rank = rank;
    c->world = world;
    c->device = ctx->device;
    c->ext_fn = all_gather_fn;
    c->ext_user = user;
    *out = c;
    return JV_OK;
}

int jv_hip_comm_destroy(jv_comm *c)
{
    if (!c) return JV_OK;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)c->rccl->CommDestroy(c->comm);
    delete c;
    return JV_OK;
}

int jv_hip_comm_rank(const jv_comm *c) { return c ? c->rank : 0; }
int jv_hip_comm_world(const jv_comm *c) { return c ? c->world : 1; }

int jv_hip_comm_count(const jv_comm *c, int *out)
{
    clear_error();
    JV_REQUIRE(c && out, "comm_count: NULL argument");
    if (c->ext_fn) {  // the host's transport: what the host said
        *out = c->world;
        return JV_OK;
    }
    if (!c->comm) {  // local communicator: no RCCL object behind it
        *out = 1;
        return JV_OK;
    }
    int n = 0;
    JV_RCCL_CHECK(c->rccl, c->rccl->CommCount(c->comm, &n));
    *out = n;
    return JV_OK;
}

int jv_hip_comm_all_gather(jv_ctx *ctx, jv_comm *comm, const void *send, size_t bytes, void *recv)
{
    clear_error();
    JV_REQUIRE(ctx && comm, "comm_all_gather: NULL argument");
    if (bytes == 0) return JV_OK;
    JV_REQUIRE(send && recv, "comm_all_gather: NULL buffer");
    JV_TRY(use_device(ctx->device));
    const size_t W = (size_t)comm->world;
    JV_TRY(comm->bytes_in.reserve(bytes));
    JV_TRY(comm->bytes_out.reserve(bytes * W));
    JV_HIP_CHECK(hipMemcpyAsync(comm->bytes_in.ptr, send, bytes, hipMemcpyDefault, ctx->stream));
    JV_TRY(all_gather(ctx, comm, comm->bytes_in.ptr, comm->bytes_out.ptr, bytes, kNcclInt8, 1));
    JV_HIP_CHECK(hipMemcpyAsync(recv, comm->bytes_out.ptr, bytes * W, hipMemcpyDefault, ctx->stream));
    JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return JV_OK;
}

int jv_hip_sharded_topk(jv_ctx *ctx, jv_comm *comm, const float *scores, const int32_t *ids, int Q, int k_in, int k_out,
                        int32_t *out_ids, float *out_scores)
{
    clear_error();
    JV_REQUIRE(ctx && comm, "sharded_topk: NULL argument");
    JV_REQUIRE(Q >= 0 && k_in > 0 && k_out > 0, "sharded_topk: bad sizes");
    if (Q == 0) return JV_OK;
    JV_REQUIRE(scores && ids && out_ids && out_scores, "sharded_topk: NULL buffer");
    JV_TRY(use_device(ctx->device));
    const size_t cells = (size_t)Q * k_in;
    const void *d_sc = nullptr, *d_ids = nullptr;
    JV_TRY(comm->part_sc.reserve(sizeof(float) * cells));
    JV_TRY(comm->part_ids.reserve(sizeof(int32_t) * cells));
    // host or device partial lists -> the communicator's send buffers
    JV_HIP_CHECK(hipMemcpyAsync(comm->part_sc.ptr, scores, sizeof(float) * cells, hipMemcpyDefault, ctx->stream));
    JV_HIP_CHECK(hipMemcpyAsync(comm->part_ids.ptr, ids, sizeof(int32_t) * cells, hipMemcpyDefault, ctx->stream));
    d_sc = comm->part_sc.ptr;
    d_ids = comm->part_ids.ptr;
    const int W = comm->world;
    JV_TRY(comm->all_sc.reserve(sizeof(float) * cells * W));
    JV_TRY(comm->all_ids.reserve(sizeof(int32_t) * cells * W));
    if (comm->comm) JV_RCCL_CHECK(comm->rccl, comm->rccl->GroupStart());
    JV_TRY(all_gather(ctx, comm, d_ids, comm->all_ids.ptr, cells, kNcclInt32, 4));
    JV_TRY(all_gather(ctx, comm, d_sc, comm->all_sc.ptr, cells, kNcclFloat32, 4));
    if (comm->comm) JV_RCCL_CHECK(comm->rccl, comm->rccl->GroupEnd());
    OutStage oi, os;
    JV_TRY(stage_out_begin(ctx, out_ids, sizeof(int32_t) * (size_t)Q * k_out, ctx->d_scratch2, &oi));
    JV_TRY(stage_out_begin(ctx, out_scores, sizeof(float) * (size_t)Q * k_out, ctx->d_scratch3, &os));
    JV_TRY(merge_pieces(ctx, comm, (const int32_t *)comm->all_ids.ptr, (const float *)comm->all_sc.ptr, W, Q, k_in, k_out, (int32_t *)oi.dev,
                        (float *)os.dev));
    JV_TRY(stage_out_end(ctx, oi));
    return stage_out_end(ctx, os);
}

}  // extern "C"

// The exchange every sharded search ends with, whatever produced the partial lists (an exhaustive ADC scan per shard, a graph search
// per shard): agreement header -> all-gather of the partial top-rerankK (global ids) -> NodeQueue-order merge -> exact scores by the
// owning shard -> all-gather + owner selection -> top-K.  counts[s] = ordinals shard s owns from id_base[s]; vectors == NULL: no rerank.
// part_ids / part_sc: [n_local][Q][rerankK] in the communicator's part_* buffers (device).  Queries must be staged in `luts`
// (luts_prepare) when reranking.
static int sharded_exchange(jv_ctx *ctx, jv_comm *comm, int n_local, jv_luts *luts, const jv_vectors *const *vectors, const int64_t *id_base,
                            const int64_t *counts, int Q, jv_vsf vsf, int topK, int rerankK, int local_status, const char *local_msg,
                            bool rerank, int (*produce)(void *), void *produce_arg, int32_t *out_ids, float *out_scores)
{
    constexpr int kHdr = 8, kMaxLocal = 64, kRec = kHdr + 2 * kMaxLocal;
    const int W = comm->world;
    std::vector<long long> h_rec(kRec, 0), h_all((size_t)kRec * W, 0);
    h_rec[0] = n_local;
    h_rec[1] = Q;
    h_rec[2] = topK;
    h_rec[3] = rerankK;
    h_rec[4] = rerank ? 1 : 0;
    h_rec[5] = (long long)vsf;
    h_rec[6] = luts->pq->D;
    h_rec[7] = local_status;
    for (int s = 0; s < n_local && s < kMaxLocal && local_status == 0; ++s) {
        h_rec[kHdr + 2 * s] = id_base[s];
        h_rec[kHdr + 2 * s + 1] = counts[s];
    }
    if (comm->comm || comm->ext_fn) {
        JV_TRY(comm->ranges.reserve(sizeof(long long) * kRec));
        JV_TRY(comm->all_ranges.reserve(sizeof(long long) * (size_t)kRec * W));
        JV_HIP_CHECK(hipMemcpyAsync(comm->ranges.ptr, h_rec.data(), sizeof(long long) * kRec, hipMemcpyHostToDevice, ctx->stream));
        JV_TRY(all_gather(ctx, comm, comm->ranges.ptr, comm->all_ranges.ptr, kRec, kNcclInt64, 8));
        JV_HIP_CHECK(hipMemcpyAsync(h_all.data(), comm->all_ranges.ptr, sizeof(long long) * h_all.size(), hipMemcpyDeviceToHost, ctx->stream));
        JV_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    } else {
        h_all = h_rec;
    }
    if (local_status != 0) {
        set_error("%s", local_msg);
        return JV_ERR_INVALID;
    }
    for (int r = 0; r < W; ++r) {
        const long long *o = h_all.data() + (size_t)kRec * r;
        JV_REQUIRE(o[7] == 0, "sharded search: rank %d rejected its arguments (see that rank's error)", r);
        JV_REQUIRE(o[0] == h_rec[0] && o[1] == h_rec[1] && o[2] == h_rec[2] && o[3] == h_rec[3] && o[5] == h_rec[5] && o[6] == h_rec[6],
                   "sharded search: rank %d disagrees with rank %d on (n_local, Q, topK, rerankK, vsf, D) = (%lld, %lld, %lld, %lld, %lld, %lld) "
                   "vs (%lld, %lld, %lld, %lld, %lld, %lld)", r, comm->rank, o[0], o[1], o[2], o[3], o[5], o[6], h_rec[0], h_rec[1], h_rec[2],
                   h_rec[3], h_rec[5], h_rec[6]);
        JV_REQUIRE(o[4] == h_rec[4], "sharded search: rank %d %s full-resolution vectors for every shard, rank %d %s — either every "
                   "rank reranks or none", r, o[4] ? "has" : "lacks", comm->rank, h_rec[4] ? "has" : "lacks");
    }
    if (Q == 0) return JV_OK;
    const int P = W * n_local, k = rerankK;
    const size_t cells = (size_t)Q * k;
    const bool grouped = comm->comm != nullptr;
    // the shard table [P][2] = {first global ordinal, count} in rank-major order, for the owner selection of step 3
    comm->h_ranges.assign(2 * (size_t)P, 0);
    for (int r = 0; r < W; ++r)
        for (int s = 0; s < n_local; ++s) {
            comm->h_ranges[2 * ((size_t)r * n_local + s)] = h_all[(size_t)kRec * r + kHdr + 2 * s];
            comm->h_ranges[2 * ((size_t)r * n_local + s) + 1] = h_all[(size_t)kRec * r + kHdr + 2 * s + 1];
        }
    JV_TRY(comm->all_ranges.reserve(sizeof(long long) * std::max<size_t>(2 * (size_t)P, (size_t)kRec * W)));
    JV_HIP_CHECK(hipMemcpyAsync(comm->all_ranges.ptr, comm->h_ranges.data(), sizeof(long long) * 2 * (size_t)P, hipMemcpyHostToDevice, ctx->stream));

    // 1. every local shard's partial top-rerankK, GLOBAL ids, into part_ids / part_sc [n_local][Q][k]
    JV_TRY(comm->part_ids.reserve(sizeof(int32_t) * cells * n_local));
    JV_TRY(comm->part_sc.reserve(sizeof(float) * cells * n_local));
    JV_TRY(produce(produce_arg));
    // 2. all-gather (ids, scores), merge -> global top-rerankK
    JV_TRY(comm->all_ids.reserve(sizeof(int32_t) * cells * P));
    JV_TRY(comm->all_sc.reserve(sizeof(float) * cells * P));
    if (grouped) JV_RCCL_CHECK(comm->rccl, comm->rccl->GroupStart());
    JV_TRY(all_gather(ctx, comm, comm->part_ids.ptr, comm->all_ids.ptr, cells * n_local, kNcclInt32, 4));
    JV_TRY(all_gather(ctx, comm, comm->part_sc.ptr, comm->all_sc.ptr, cells * n_local, kNcclFloat32, 4));
    if (grouped) JV_RCCL_CHECK(comm->rccl, comm->rccl->GroupEnd());
    JV_TRY(comm->cand.reserve(sizeof(int32_t) * cells));
    JV_TRY(comm->cand_sc.reserve(sizeof(float) * cells));
    OutStage oi, os;
    JV_TRY(stage_out_begin(ctx, out_ids, sizeof(int32_t) * (size_t)Q * topK, ctx->d_scratch2, &oi));
    JV_TRY(stage_out_begin(ctx, out_scores, sizeof(float) * (size_t)Q * topK, ctx->d_scratch3, &os));
    if (!rerank) {  // no full-resolution vectors: the merged approximate top-K is the answer
        JV_TRY(merge_pieces(ctx, comm, (const int32_t *)comm->all_ids.ptr, (const float *)comm->all_sc.ptr, P, Q, k, topK, (int32_t *)oi.dev,
                            (float *)os.dev));
        JV_TRY(stage_out_end(ctx, oi));
        return stage_out_end(ctx, os);
    }
    JV_TRY(merge_pieces(ctx, comm, (const int32_t *)comm->all_ids.ptr, (const float *)comm->all_sc.ptr, P, Q, k, k, (int32_t *)comm->cand.ptr,
                        (float *)comm->cand_sc.ptr));
    // 3. exact scores by the owning shard (same kernel, same arithmetic as the single index), all-gather, owner selection
    JV_TRY(comm->local.reserve(sizeof(int32_t) * cells));
    JV_TRY(comm->exact.reserve(sizeof(float) * cells * n_local));
    JV_TRY(comm->all_exact.reserve(sizeof(float) * cells * P));
    JV_TRY(ctx->d_in.reserve(sizeof(float) * (size_t)Q));
    for (int s = 0; s < n_local; ++s) {
        JV_TRY(launch_shard_localize(ctx->stream, (const int32_t *)comm->cand.ptr, (int64_t)cells, id_base[s], counts[s],
                                     (int32_t *)comm->local.ptr));
        JV_TRY(rerank_gather(ctx, vectors[s], luts->d_raw_queries, Q, vsf, (const int32_t *)comm->local.ptr, k,
                             (float *)comm->exact.ptr + cells * s, (float *)ctx->d_in.ptr));
    }
    JV_TRY(all_gather(ctx, comm, comm->exact.ptr, comm->all_exact.ptr, cells * n_local, kNcclFloat32, 4));
    JV_TRY(launch_shard_select(ctx->stream, (const int32_t *)comm->cand.ptr, (const float *)comm->all_exact.ptr,
                               (const long long *)comm->all_ranges.ptr, P, (int64_t)cells, (float *)comm->cand_sc.ptr));
    // 4. final top-K under the NodeQueue order on the exact scores
    JV_TRY(ctx->d_scratch.reserve(topk_scratch_bytes(Q, topK)));
    {
        ProfScope ps(ctx, R_TOPK);
        JV_TRY(launch_topk(ctx->stream, ctx, (const float *)comm->cand_sc.ptr, (const int32_t *)comm->cand.ptr, Q, k, k, 0, topK,
                           (int32_t *)oi.dev, (float *)os.dev, ctx->d_scratch.ptr));
    }
    JV_TRY(stage_out_end(ctx, oi));
    return stage_out_end(ctx, os);
}

namespace {
struct Msg {   // a rank's own argument check: recorded, exchanged, and only then acted upon (every rank issues the same collectives)
    char text[256] = {0};
    int status = 0;
    template <typename... A>
    void fail(const char *fmt, A... a)
    {
        if (status == 0) {
            if constexpr (sizeof...(a) == 0) snprintf(text, sizeof(text), "%s", fmt);
            else snprintf(text, sizeof(text), fmt, a...);
            status = 1;
        }
    }
};
struct FlatProduce {
    jv_ctx *ctx;
    jv_comm *comm;
    jv_luts *luts;
    const jv_codes *const *codes;
    const int64_t *id_base;
    const float *queries;
    int n_local, Q, k;
    jv_vsf vsf;
};
int flat_produce(void *a)
{
    const FlatProduce &f = *(const FlatProduce *)a;
    const size_t cells = (size_t)f.Q * f.k;
    for (int s = 0; s < f.n_local; ++s)
        JV_TRY(jv_hip_search_flat(f.ctx, f.luts, f.codes[s], nullptr, f.queries, f.Q, f.vsf, f.k, 0, (int32_t)f.id_base[s],
                                  (int32_t *)f.comm->part_ids.ptr + cells * s, (float *)f.comm->part_sc.ptr + cells * s));
    return JV_OK;
}
struct GivenProduce {
    jv_ctx *ctx;
    jv_comm *comm;
    jv_luts *luts;
    const float *queries;
    const int32_t *part_ids;
    const float *part_sc;
    int n_local, Q, k;
    jv_vsf vsf;
    bool rerank;
    const int64_t *id_base, *counts;
};
int given_produce(void *a)
{
    const GivenProduce &g = *(const GivenProduce *)a;
    const size_t cells = (size_t)g.Q * g.k * g.n_local;
    JV_HIP_CHECK(hipMemcpyAsync(g.comm->part_ids.ptr, g.part_ids, sizeof(int32_t) * cells, hipMemcpyDefault, g.ctx->stream));
    JV_HIP_CHECK(hipMemcpyAsync(g.comm->part_sc.ptr, g.part_sc, sizeof(float) * cells, hipMemcpyDefault, g.ctx->stream));
    // the caller's lists are taken at their word only inside the shard's own range: an id nobody owns leaves the exchange here
    const size_t per = (size_t)g.Q * g.k;
    for (int s = 0; s < g.n_local; ++s)
        JV_TRY(launch_shard_sanitize(g.ctx->stream, (int32_t *)g.comm->part_ids.ptr + per * s, (float *)g.comm->part_sc.ptr + per * s, (int64_t)per,
                                     g.id_base[s], g.id_base[s] + g.counts[s]));
    // the raw queries the exact rerank scores against (jv_hip_search_flat stages them itself on the flat path)
    if (g.rerank) JV_TRY(luts_prepare(g.ctx, g.luts, g.queries, g.Q, g.vsf, JV_DECODER_PQ, false));
    return JV_OK;
}
}  // namespace

extern "C" {

int jv_hip_sharded_search_flat(jv_ctx *ctx, jv_comm *comm, int n_local, jv_luts *luts, const jv_codes *const *codes,
                               const jv_vectors *const *vectors, const int64_t *id_base, const float *queries, int Q, jv_vsf vsf,
                               int topK, int rerankK, int32_t *out_ids, float *out_scores)
{
    clear_error();
    JV_REQUIRE(ctx && comm && luts && codes && id_base, "sharded_search_flat: NULL argument");
    JV_TRY(use_device(ctx->device));
    // ---- 0. agreement.  The collectives must be issued by every rank the same number of times with the same sizes, so nothing
    //      rank-local may decide whether a rank takes part: each rank validates its own arguments into a status word, all ranks
    //      exchange one fixed-size header and then take the SAME decision from the same table (sharded_exchange).
    Msg m;
    if (n_local < 1 || n_local > 64) m.fail("sharded_search_flat: %d local shards (1..64)", n_local);
    if (!(topK > 0 && rerankK >= topK)) m.fail("rerankK %d must be >= topK %d", rerankK, topK);  // GraphSearcher.java:233
    if (Q < 0 || (Q > 0 && !(queries && out_ids && out_scores))) m.fail("sharded_search_flat: NULL buffer");
    bool rerank = vectors != nullptr;
    std::vector<int64_t> counts((size_t)std::max(n_local, 0), 0);
    for (int s = 0; s < n_local && m.status == 0; ++s) {
        if (!codes[s]) {
            m.fail("sharded_search_flat: shard %d has no codes", s);
            break;
        }
        counts[(size_t)s] = codes[s]->count;
        if (!(id_base[s] >= 0 && id_base[s] + codes[s]->count <= 0x7fffffffLL)) m.fail("sharded_search_flat: shard %d id range overflows int32", s);
        if (codes[s]->M != luts->pq->M) m.fail("sharded_search_flat: shard %d codes have M = %d, the tables M = %d", s, codes[s]->M, luts->pq->M);
        if (vectors && vectors[s]) {
            if (vectors[s]->count < codes[s]->count) m.fail("sharded_search_flat: shard %d has fewer vectors than codes", s);
            if (vectors[s]->D != luts->pq->D) m.fail("sharded_search_flat: shard %d vectors have D = %d, the quantizer D = %d", s, vectors[s]->D, luts->pq->D);
        } else {
            rerank = false;
        }
    }
    FlatProduce f{ctx, comm, luts, codes, id_base, queries, n_local, Q, rerankK, vsf};
    return sharded_exchange(ctx, comm, n_local, luts, vectors, id_base, counts.data(), Q, vsf, topK, rerankK, m.status, m.text, rerank, flat_produce,
                            &f, out_ids, out_scores);
}

// The same exchange for partial lists the CALLER produced (one graph index per shard — the way JVector deployments shard: a segment
// index per partition — or anything else that yields a top-rerankK with global ids): part_ids / part_scores [n_local][Q][rerankK],
// host or device memory.  vectors == NULL: the merged approximate top-K is the answer.
int jv_hip_sharded_merge_rerank(jv_ctx *ctx, jv_comm *comm, int n_local, jv_luts *luts, const jv_vectors *const *vectors, const int64_t *id_base,
                                const int64_t *counts, const float *queries, int Q, jv_vsf vsf, int topK, int rerankK, const int32_t *part_ids,
                                const float *part_scores, int32_t *out_ids, float *out_scores)
{
    clear_error();
    JV_REQUIRE(ctx && comm && luts && id_base && counts, "sharded_merge_rerank: NULL argument");
    JV_TRY(use_device(ctx->device));
    Msg m;
    if (n_local < 1 || n_local > 64) m.fail("sharded_merge_rerank: %d local shards (1..64)", n_local);
    if (!(topK > 0 && rerankK >= topK)) m.fail("rerankK %d must be >= topK %d", rerankK, topK);
    if (Q < 0 || Q > luts->capacity) m.fail("sharded_merge_rerank: %d queries, the tables hold %d", Q, luts->capacity);
    if (Q > 0 && !(part_ids && part_scores && out_ids && out_scores)) m.fail("sharded_merge_rerank: NULL buffer");
    bool rerank = vectors != nullptr;
    for (int s = 0; s < n_local && m.status == 0; ++s) {
        if (!(id_base[s] >= 0 && counts[s] >= 0 && id_base[s] + counts[s] <= 0x7fffffffLL)) m.fail("sharded_merge_rerank: shard %d id range overflows int32", s);
        if (vectors && vectors[s]) {
            if (vectors[s]->count < counts[s]) m.fail("sharded_merge_rerank: shard %d has fewer vectors than ordinals", s);
            if (vectors[s]->D != luts->pq->D) m.fail("sharded_merge_rerank: shard %d vectors have D = %d, the quantizer D = %d", s, vectors[s]->D, luts->pq->D);
        } else {
            rerank = false;
        }
    }
    if (rerank && Q > 0 && !queries) m.fail("sharded_merge_rerank: the exact rerank needs the queries");
    GivenProduce g{ctx, comm, luts, queries, part_ids, part_scores, n_local, Q, rerankK, vsf, rerank, id_base, counts};
    return sharded_exchange(ctx, comm, n_local, luts, vectors, id_base, counts, Q, vsf, topK, rerankK, m.status, m.text, rerank, given_produce, &g,
                            out_ids, out_scores);
}

}  // extern "C"
