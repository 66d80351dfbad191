// k_pq_aniso.hip — anisotropic PQ encode: one wavefront per vector, body in an_body.h (shared with
// the CPU lane-emulator tests).  Compute-bound on the recomputed residuals ((1 + sweeps) x M x 256 x 8 products per
// vector); the codebook is read through L2, the vector and its code live in LDS.
#include <algorithm>
#include <cmath>

#include "gs_wave_hip.h"
#include "jv_device.h"
#include "jv_internal.h"

#include "an_body.h"

namespace jv {

__global__ __launch_bounds__(64) void pq_encode_aniso_kernel(AnParams p)
{
    extern __shared__ __attribute__((aligned(16))) char an_lds[];
    an_worker(p, (int)blockIdx.x, (int)gridDim.x, an_lds);
}

// KMeansPlusPlusClusterer.computeParallelCostMultiplier :116-124
static float parallel_cost_multiplier(float threshold, int dimensions)
{
    const double t = (double)threshold;
    const double parallelCost = t * t;
    const double perpendicularCost = (1 - parallelCost) / (dimensions - 1);
    return (float)std::max(1.0, parallelCost / perpendicularCost);
}

int launch_pq_encode_anisotropic(hipStream_t s, const jv_pq *pq, const float *d_vecs, int64_t count, uint8_t *d_codes)
{
    if (count == 0) return JV_OK;
    AnParams p{pq->d_codebooks, pq->d_cb_offsets, pq->d_sizes, pq->d_offsets, pq->d_centroid, pq->d_self_mag, d_vecs, d_codes,
               count, pq->D, pq->M, pq->k, parallel_cost_multiplier(pq->aniso, pq->D)};
    const int workers = (int)std::min<int64_t>(count, 256 * 16);
    hipLaunchKernelGGL(pq_encode_aniso_kernel, dim3(workers), dim3(64), an_lds_bytes(pq->D, pq->M), s, p);
    JV_HIP_CHECK(hipGetLastError());
    return JV_OK;
}

}  // namespace jv
