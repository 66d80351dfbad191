// k_pq_train.hip — PQ training kernels; bodies in km_body.h, shared with the CPU tests.
// Flat launches except the k-means++ seeding (one wavefront per subspace).  Training touches <= 128 000 vectors
// (ProductQuantization.MAX_PQ_TRAINING_SET_SIZE): everything here is small next to the 10M-vector encode, the point is
// bit-reproducibility of the reference's order-dependent accumulations, not throughput.
#include "gs_wave_hip.h"
#include "jv_device.h"
#include "jv_internal.h"

#define KM_FN __device__ __forceinline__
#include "km_body.h"

namespace jv {

__global__ __launch_bounds__(256) void km_centroid_kernel(const float *X, int64_t n, int D, float *out)
{
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d < D) km_centroid_dim(X, n, D, d, out);
}
__global__ __launch_bounds__(256) void km_center_kernel(const float *X, const float *centroid, int D, int64_t total, float *Xc)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < total) km_center(X, centroid, D, t, Xc);
}
__global__ __launch_bounds__(256) void km_assign_kernel(KmParams p, int64_t total)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < total) km_assign(p, t);
}
__global__ __launch_bounds__(256) void km_replay_kernel(KmParams p, int first_pass, int64_t total)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < total) km_replay(p, first_pass, t);
}
__global__ __launch_bounds__(256) void km_centroids_kernel(KmParams p, int64_t total)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < total) km_centroids(p, t);
}
__global__ __launch_bounds__(64) void km_fill_empties_kernel(KmParams p)
{
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m < p.M) km_fill_empties(p, m);
}
__global__ __launch_bounds__(64) void km_finish_round_kernel(KmParams p)
{
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m < p.M) km_finish_round(p, m);
}
__global__ __launch_bounds__(64) void km_pp_init_kernel(KmParams p) { km_pp_init(p, (int)blockIdx.x); }
__global__ __launch_bounds__(64) void km_centroids_aniso_kernel(KmParams p, int64_t total)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < total) km_centroids_aniso(p, t);
}
__global__ __launch_bounds__(256) void km_cnorm_kernel(KmParams p, int64_t total)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < total) km_cnorm(p, t);
}
__global__ __launch_bounds__(256) void km_assign_aniso_kernel(KmParams p, int64_t total)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < total) km_assign_aniso(p, t);
}
__global__ __launch_bounds__(64) void km_count_changed_kernel(KmParams p)
{
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m < p.M) km_count_changed(p, m);
}
__global__ __launch_bounds__(64) void km_reactivate_kernel(KmParams p)
{
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m < p.M) km_reactivate(p, m);
}

static dim3 grid_for(int64_t total, int block) { return dim3((unsigned)((total + block - 1) / block)); }

int launch_km_centroid(hipStream_t s, const float *d_X, int64_t n, int D, float *d_out)
{
    hipLaunchKernelGGL(km_centroid_kernel, grid_for(D, 256), dim3(256), 0, s, d_X, n, D, d_out);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}
int launch_km_center(hipStream_t s, const float *d_X, const float *d_centroid, int64_t n, int D, float *d_Xc)
{
    const int64_t total = n * D;
    hipLaunchKernelGGL(km_center_kernel, grid_for(total, 256), dim3(256), 0, s, d_X, d_centroid, D, total, d_Xc);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}
int launch_km_pp_init(hipStream_t s, const KmParams &p)
{
    hipLaunchKernelGGL(km_pp_init_kernel, dim3(p.M), dim3(64), 0, s, p);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}
int launch_km_assign(hipStream_t s, const KmParams &p)
{
    const int64_t total = p.n * p.M;
    hipLaunchKernelGGL(km_assign_kernel, grid_for(total, 256), dim3(256), 0, s, p, total);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}
int launch_km_replay(hipStream_t s, const KmParams &p, int first_pass)
{
    const int64_t total = (int64_t)p.M * p.k;
    hipLaunchKernelGGL(km_replay_kernel, grid_for(total, 256), dim3(256), 0, s, p, first_pass, total);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}
int launch_km_update_centroids(hipStream_t s, const KmParams &p)
{
    const int64_t total = (int64_t)p.M * p.k;
    hipLaunchKernelGGL(km_centroids_kernel, grid_for(total, 256), dim3(256), 0, s, p, total);
    hipLaunchKernelGGL(km_fill_empties_kernel, grid_for(p.M, 64), dim3(64), 0, s, p);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}
// one anisotropic round: centroids (+ empty-cluster re-seeding) -> centroid norms -> weighted reassignment -> changed counts
int launch_km_aniso_round(hipStream_t s, const KmParams &p)
{
    const int64_t mk = (int64_t)p.M * p.k, nm = p.n * p.M;
    hipLaunchKernelGGL(km_centroids_aniso_kernel, grid_for(mk, 64), dim3(64), 0, s, p, mk);
    hipLaunchKernelGGL(km_fill_empties_kernel, grid_for(p.M, 64), dim3(64), 0, s, p);
    hipLaunchKernelGGL(km_cnorm_kernel, grid_for(mk, 256), dim3(256), 0, s, p, mk);
    hipLaunchKernelGGL(km_assign_aniso_kernel, grid_for(nm, 256), dim3(256), 0, s, p, nm);
    hipLaunchKernelGGL(km_count_changed_kernel, grid_for(p.M, 64), dim3(64), 0, s, p);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}
int launch_km_reactivate(hipStream_t s, const KmParams &p)
{
    hipLaunchKernelGGL(km_reactivate_kernel, grid_for(p.M, 64), dim3(64), 0, s, p);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}
int launch_km_finish_round(hipStream_t s, const KmParams &p)
{
    hipLaunchKernelGGL(km_finish_round_kernel, grid_for(p.M, 64), dim3(64), 0, s, p);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

}  // namespace jv
