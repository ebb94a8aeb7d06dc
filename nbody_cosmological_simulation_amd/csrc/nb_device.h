// nb_device.h -- device-side helpers shared by the force kernels (nb_force.hip, nb_force_sym.hip).
#pragma once
#include "nb_internal.h"

namespace nbdev {

__device__ __forceinline__ float round_bf16(float x) { return (float)(__bf16)x; }
__device__ __forceinline__ float round_f16(float x) { return (float)(_Float16)x; }

// mass_prod = masses[i] * masses[j] keeps the masses' dtype upstream (simulation.py:185): fp32-typed masses give an
// fp32-rounded product even when the positions are fp64, half-typed masses (omega_point_test.py:722-733 keeps them
// half while the rest of the state is promoted) a product rounded to that half type.  mass_dt: nb_dtype of the masses.
__device__ __forceinline__ float mass_prod_f32(float mi, float mj, int mass_dt)
{
    float p = __fmul_rn(mi, mj);                       // exact for half inputs (11 + 11 bits)
    if (mass_dt == NB_F16) p = (float)(_Float16)p;
    else if (mass_dt == NB_BF16) p = (float)(__bf16)p;
    return p;
}

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
// lp: table size (power of two, entries beyond the levels hold +inf)
__device__ __forceinline__ int grid_bin_lookup(const float *thr, float r2, int lp)
{
    int k = 0;
    for (int step = lp >> 1; step >= 1; step >>= 1)
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

// Floor form of the estimate: with k0 = floor(estimate) the exact bin is k0 or k0 + 1, so ONE
// threshold (thr[k0+1]) settles it.  kmax = levels - 2.
__device__ __forceinline__ int grid_bin_floor_estimate(const float *thr, float r2, float est_a, float est_b, int kmax)
{
    const float ne = __builtin_fmaf(__builtin_amdgcn_logf(r2), est_a, est_b);
    int k0 = (int)ne;
    k0 = min(max(k0, 0), kmax);
    return k0 + ((r2 >= thr[k0 + 1]) ? 1 : 0);
}

// Same guarantee, one LDS access: with k0 = floor(estimate) the exact bin is k0 or k0 + 1 (the exact
// index is rint() of a value the estimate tracks to << 0.5), so a record {thr[k0+1], lut[k0], lut[k0+1]}
// decides it with one compare.  Returns the force factor (1/q^1.5)*G of the exact bin.
__device__ __forceinline__ float grid_w_estimate(const float4 *rec, float r2, float est_a, float est_b, int kmax)
{
    const float ne = __builtin_fmaf(__builtin_amdgcn_logf(r2), est_a, est_b);
    int k0 = (int)ne;                       // truncation; negative estimates clamp to bin 0 below
    k0 = min(max(k0, 0), kmax);
    const float4 rc = rec[k0];
    return (r2 >= rc.x) ? rc.z : rc.y;
}

}  // namespace nbdev
