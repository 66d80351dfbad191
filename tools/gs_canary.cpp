// gs_canary.cpp — standalone check + timing of the two graph traversals through the C ABI, without Python or torch
// (a fresh GPU box spends 1-2 minutes importing torch; this starts in a second and is what a first hardware run of a new
// traversal kernel should be: small, bounded, comparing against the verified host searcher).
//   build : make -C jvector_amd/csrc canary      (g++, links libjvector_hip.so; no HIP headers needed)
//   run   : build/gs_canary [N] [Q] [degree] [rerankK] [iters] [vsf 0|1|2]
// Random unit vectors are not needed: codes are uniform random bytes, the graph is a random regular digraph, queries are
// Gaussian.  Every search runs once on the host traversal and `iters` times on the device traversal; ids, scores and the
// visited / expanded counters must be identical.  Prints one JSON line; exit code 0 = identical.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../include/jvector_hip.h"

#define CK(expr)                                                                      \
    do {                                                                              \
        int _s = (expr);                                                              \
        if (_s != JV_OK) {                                                            \
            fprintf(stderr, "%s failed (%d): %s\n", #expr, _s, jv_hip_last_error()); \
            return 2;                                                                 \
        }                                                                             \
    } while (0)

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    const int64_t N = argc > 1 ? atoll(argv[1]) : 200000;
    const int Q = argc > 2 ? atoi(argv[2]) : 2048;
    const int deg = argc > 3 ? atoi(argv[3]) : 32;
    const int rerankK = argc > 4 ? atoi(argv[4]) : 100;
    const int iters = argc > 5 ? atoi(argv[5]) : 3;
    const jv_vsf vsf = (jv_vsf)(argc > 6 ? atoi(argv[6]) : 2);
    const int D = 768, M = 96, topK = 10;
    std::mt19937_64 rng(12345);
    std::normal_distribution<float> gauss(0.0f, 1.0f);

    jv_ctx *ctx = nullptr;
    CK(jv_hip_ctx_create(0, nullptr, &ctx));
    std::vector<float> cb((size_t)256 * D);
    for (auto &x : cb) x = 0.05f * gauss(rng);
    jv_pq *pq = nullptr;
    CK(jv_hip_pq_create(ctx, D, M, 256, nullptr, cb.data(), nullptr, &pq));
    std::vector<uint8_t> codes((size_t)N * M);
    for (auto &c : codes) c = (uint8_t)(rng() & 0xFF);
    jv_codes *cv = nullptr;
    CK(jv_hip_codes_create(ctx, pq, N, &cv));
    CK(jv_hip_codes_upload(ctx, cv, 0, N, codes.data()));
    std::vector<int32_t> nbrs((size_t)N * deg);
    for (auto &n : nbrs) n = (int32_t)(rng() % (uint64_t)N);
    jv_fused *fused = nullptr;
    CK(jv_hip_fused_create(ctx, pq, N, deg, &fused));
    CK(jv_hip_fused_build(ctx, fused, cv, 0, N, nbrs.data()));
    jv_graph *g = nullptr;
    CK(jv_hip_graph_create(ctx, N, 1, &g));
    CK(jv_hip_graph_set_level(ctx, g, 0, (int)N, nullptr, nbrs.data(), deg));
    CK(jv_hip_graph_set_entry(g, 0, 0));
    jv_luts *luts = nullptr;
    CK(jv_hip_luts_create(ctx, pq, Q, &luts));
    std::vector<float> queries((size_t)Q * D);
    for (auto &x : queries) x = gauss(rng);

    std::vector<int32_t> ids_h((size_t)Q * topK), ids_d((size_t)Q * topK);
    std::vector<float> sc_h((size_t)Q * topK), sc_d((size_t)Q * topK);
    std::vector<int64_t> st_h((size_t)Q * 2), st_d((size_t)Q * 2);

    CK(jv_hip_graph_set_traversal(g, JV_TRAVERSAL_HOST));
    CK(jv_hip_graph_search(ctx, g, luts, cv, fused, nullptr, queries.data(), Q, vsf, topK, rerankK, ids_h.data(), sc_h.data(), st_h.data()));
    double t0 = now_ms();
    CK(jv_hip_graph_search(ctx, g, luts, cv, fused, nullptr, queries.data(), Q, vsf, topK, rerankK, ids_h.data(), sc_h.data(), st_h.data()));
    const double host_ms = now_ms() - t0;

    CK(jv_hip_graph_set_traversal(g, JV_TRAVERSAL_DEVICE));
    CK(jv_hip_graph_search(ctx, g, luts, cv, fused, nullptr, queries.data(), Q, vsf, topK, rerankK, ids_d.data(), sc_d.data(), st_d.data()));
    double best = 1e300;
    for (int it = 0; it < iters; ++it) {
        t0 = now_ms();
        CK(jv_hip_graph_search(ctx, g, luts, cv, fused, nullptr, queries.data(), Q, vsf, topK, rerankK, ids_d.data(), sc_d.data(),
                               st_d.data()));
        const double ms = now_ms() - t0;
        if (ms < best) best = ms;
    }
    if (iters <= 0) best = 0.0;
    const bool same = ids_h == ids_d && !memcmp(sc_h.data(), sc_d.data(), sizeof(float) * sc_h.size()) && st_h == st_d;
    double expanded = 0;
    for (int q = 0; q < Q; ++q) expanded += (double)st_h[2 * q + 1];
    printf("{\"identical\": %s, \"N\": %lld, \"Q\": %d, \"degree\": %d, \"rerankK\": %d, \"vsf\": %d, \"avg_expanded\": %.1f, "
           "\"host_ms\": %.2f, \"device_ms\": %.2f, \"host_qps\": %.0f, \"device_qps\": %.0f, \"arch\": \"%s\"}\n",
           same ? "true" : "false", (long long)N, Q, deg, rerankK, (int)vsf, expanded / Q, host_ms, best, Q / host_ms * 1e3, best > 0 ? Q / best * 1e3 : 0.0,
           jv_hip_active_arch(0));
    jv_hip_luts_destroy(luts);
    jv_hip_graph_destroy(g);
    jv_hip_fused_destroy(fused);
    jv_hip_codes_destroy(cv);
    jv_hip_pq_destroy(pq);
    jv_hip_ctx_destroy(ctx);
    return same ? 0 : 1;
}
