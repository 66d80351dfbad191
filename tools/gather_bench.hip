// gather_bench.hip — how fast can a CU gather 32-byte codebook rows from an L2-resident 768 KB table?
// (the inner loop of the table-free FusedPQ scoring).  Variants:
//   0: one lane per row, two dwordx4 loads (what round 1's kernels do)
//   1: two adjacent lanes per row, one dwordx4 each (the pair shares a 128-byte line)
//   2: four adjacent lanes per row, one dwordx2 each
//   3: one lane per row, rows forced to distinct lines but SAME row for all lanes of a quad (upper bound of coalescing)
//   4: variant 0 with only 34 of the 64 lanes active (the traversal kernel's average: 17 fresh neighbours x 2 lanes) — is the
//      cost per instruction or per active lane?
//   5: variant 0 with 16 of 64 lanes active
// build: hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o build/gather_bench ; run: build/gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int M = 96, K = 256, ROWS = M * K;  // 24576 rows of 8 floats

template <int VAR>
__global__ __launch_bounds__(64) void gather(const float *__restrict__ cb, const unsigned char *__restrict__ codes, int iters,
                                             float *__restrict__ out)
{
    const int lane = threadIdx.x;
    const unsigned char *my = codes + ((size_t)blockIdx.x * 64 + (VAR == 1 ? (lane >> 1) : VAR == 2 ? (lane >> 2) : VAR == 3 ? (lane & ~3) : lane)) * M;
    float acc = 0.f;
    if ((VAR == 4 && lane >= 34) || (VAR == 5 && lane >= 16)) {
        out[(size_t)blockIdx.x * 64 + lane] = 0.f;
        return;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll 8
        for (int m = 0; m < M; ++m) {
            const unsigned code = my[m] ^ (it & 255);
            const float *row = cb + ((size_t)m * K + (code & 255)) * 8;
            if (VAR == 0 || VAR == 3 || VAR == 4 || VAR == 5) {
                const float4 a = *reinterpret_cast<const float4 *>(row);
                const float4 b = *reinterpret_cast<const float4 *>(row + 4);
                acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
            } else if (VAR == 1) {
                const float4 a = *reinterpret_cast<const float4 *>(row + 4 * (lane & 1));
                acc += a.x + a.y + a.z + a.w;
            } else {
                const float2 a = *reinterpret_cast<const float2 *>(row + 2 * (lane & 3));
                acc += a.x + a.y;
            }
        }
    }
    out[(size_t)blockIdx.x * 64 + lane] = acc;
}

template <int VAR>
void run(const float *cb, const unsigned char *codes, float *out, int blocks, int iters, const char *name)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(gather<VAR>, dim3(blocks), dim3(64), 0, 0, cb, codes, 1, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(gather<VAR>, dim3(blocks), dim3(64), 0, 0, cb, codes, iters, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double rows_per_lane_group = (double)M * iters;
    const double groups = (double)blocks * (VAR == 1 ? 32 : VAR == 2 ? 16 : VAR == 4 ? 34 : VAR == 5 ? 16 : 64);
    const double rows = rows_per_lane_group * groups;
    printf("%-44s %8.3f ms  %7.1f G rows/s  %6.2f TB/s  %6.3f rows/clk/CU (2.4 GHz, 256 CUs)\n", name, ms, rows / ms / 1e6,
           rows * 32 / ms / 1e9, rows / (ms * 1e-3) / 2.4e9 / 256);
}

int main()
{
    const int blocks = 256 * 8, iters = 40;  // 8 waves per CU, like the traversal kernel
    std::vector<float> h_cb((size_t)ROWS * 8);
    for (auto &x : h_cb) x = (float)rand() / RAND_MAX;
    std::vector<unsigned char> h_codes((size_t)blocks * 64 * M);
    for (auto &c : h_codes) c = (unsigned char)(rand() & 255);
    float *cb, *out;
    unsigned char *codes;
    CHECK(hipMalloc((void **)&cb, h_cb.size() * 4));
    CHECK(hipMalloc((void **)&codes, h_codes.size()));
    CHECK(hipMalloc((void **)&out, (size_t)blocks * 64 * 4));
    CHECK(hipMemcpy(cb, h_cb.data(), h_cb.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(codes, h_codes.data(), h_codes.size(), hipMemcpyHostToDevice));
    run<0>(cb, codes, out, blocks, iters, "0: lane per row, 2 x dwordx4");
    run<1>(cb, codes, out, blocks, iters, "1: lane PAIR per row, 1 x dwordx4 each");
    run<2>(cb, codes, out, blocks, iters, "2: lane QUAD per row, 1 x dwordx2 each");
    run<3>(cb, codes, out, blocks, iters, "3: quad reads the same row (2 x dwordx4)");
    run<4>(cb, codes, out, blocks, iters, "4: variant 0, 34 of 64 lanes active");
    run<5>(cb, codes, out, blocks, iters, "5: variant 0, 16 of 64 lanes active");
    return 0;
}
