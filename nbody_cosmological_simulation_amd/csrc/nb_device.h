// nb_device.h -- device-side helpers shared by the force kernels (nb_force.hip, nb_force_sym.hip).
#pragma once
#include "nb_internal.h"

namespace nbdev {

__device__ __forceinline__ float round_bf16(float x) { return (float)(__bf16)x; }
__device__ __forceinline__ float round_f16(float x) { return (float)(_Float16)x; }

// r2 exactly as the reference's fp32 tensors produce it: (dx*dx + dy*dy [+ dz*dz]) + eps2, one
// rounding per operation, no fused multiply-add (simulation.py:86).
template <int D>
__device__ __forceinline__ float r2_f32_exact(const float *d, float eps2)
{
    float s = __fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1]));
    if (D == 3) s = __fadd_rn(s, __fmul_rn(d[2], d[2]));
    return __fadd_rn(s, eps2);
}

// bin index = number of thresholds <= r2 (branch-free binary search over the LDS table)
template <int LP>
__device__ __forceinline__ int grid_bin_lookup(const float *thr, float r2)
{
    int k = 0;
#pragma unroll
    for (int step = LP / 2; step >= 1; step >>= 1)
        k += (thr[k + step] <= r2) ? step : 0;
    return k;
}

// Exact bin from an estimate: k0 = rint(log2(r2)*a + b) is within one bin of the true index
// (grid_tables_kernel guarantees it when tab->use_est), thr[k0] / thr[k0+1] settle it.
__device__ __forceinline__ int grid_bin_estimate(const float *thr, float r2, float est_a, float est_b, int lmax_bin)
{
    const float ne = __builtin_fmaf(__builtin_amdgcn_logf(r2), est_a, est_b);
    int k0 = (int)(ne + 0.5f);
    k0 = min(max(k0, 0), lmax_bin);
    const float lo = thr[k0], hi = thr[k0 + 1];
    return k0 - ((r2 < lo) ? 1 : 0) + ((r2 >= hi) ? 1 : 0);
}

}  // namespace nbdev
