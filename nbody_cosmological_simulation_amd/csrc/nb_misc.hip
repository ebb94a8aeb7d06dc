// nb_misc.hip -- O(N) kernels around the force sum: slab reduction + leapfrog kicks
// (simulation.py:132-141), dtype conversion, linear force quantisation (quantization.py:74-88),
// energies (simulation.py:170-192) and the tensor-level precision hooks.  gfx950 only.
#include "nb_device.h"

#include <hip/hip_fp16.h>

namespace {

constexpr int EW_BLOCK = 256;

inline int ew_grid(int64_t count, int per_thread = 1)
{
    int64_t b = (count + (int64_t)EW_BLOCK * per_thread - 1) / ((int64_t)EW_BLOCK * per_thread);
    if (b < 1) b = 1;
    if (b > 2048 * 8) b = 2048 * 8;
    return (int)b;
}

// torch semantics: `a + b * s` is two rounded operations (no fused multiply-add), with the
// Python scalar cast to the tensor dtype first.
__device__ __forceinline__ float axpy1(float a, float b, float s) { return __fadd_rn(a, __fmul_rn(b, s)); }
__device__ __forceinline__ double axpy1(double a, double b, double s) { return __dadd_rn(a, __dmul_rn(b, s)); }

// acc = sum_s partial[s] (fixed order -> reproducible); optional closing half kick v += a*h
template <typename T>
__global__ void __launch_bounds__(EW_BLOCK)
reduce_kernel(const double *__restrict__ partial, int nchunks, int64_t count, T *__restrict__ acc,
              T *__restrict__ vel, T half_dt, int do_kick, T *__restrict__ pos, T dt)
{
    for (int64_t idx = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; idx < count;
         idx += (int64_t)gridDim.x * EW_BLOCK) {
        double s = partial[idx];
        for (int c = 1; c < nchunks; ++c) s += partial[(size_t)c * count + idx];
        const T a = (T)s;
        acc[idx] = a;
        if (do_kick == 1) {
            vel[idx] = axpy1(vel[idx], a, half_dt);
        } else if (do_kick == 2) {
            // closing kick of this step and the opening kick + drift of the next one (simulation.py:141,
            // then :132,:135 of the following step()): the same operations the separate launches perform
            T v = axpy1(vel[idx], a, half_dt);
            v = axpy1(v, a, half_dt);
            vel[idx] = v;
            pos[idx] = axpy1(pos[idx], v, dt);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(EW_BLOCK)
axpy_kernel(T *__restrict__ y, const T *__restrict__ x, T s, int64_t count)
{
    for (int64_t idx = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; idx < count;
         idx += (int64_t)gridDim.x * EW_BLOCK)
        y[idx] = axpy1(y[idx], x[idx], s);
}

// v += a*(dt/2); x += v*dt   (simulation.py:132,135)
template <typename T>
__global__ void __launch_bounds__(EW_BLOCK)
kick_drift_kernel(T *__restrict__ pos, T *__restrict__ vel, const T *__restrict__ acc, T half_dt, T dt,
                  int64_t count)
{
    for (int64_t idx = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; idx < count;
         idx += (int64_t)gridDim.x * EW_BLOCK) {
        const T v = axpy1(vel[idx], acc[idx], half_dt);
        vel[idx] = v;
        pos[idx] = axpy1(pos[idx], v, dt);
    }
}

// ---- dtype conversion -----------------------------------------------------------------------
template <typename T> __device__ __forceinline__ double load_as_double(const T *p, int64_t i) { return (double)p[i]; }
template <> __device__ __forceinline__ double load_as_double<_Float16>(const _Float16 *p, int64_t i) { return (double)(float)p[i]; }
template <> __device__ __forceinline__ double load_as_double<__bf16>(const __bf16 *p, int64_t i) { return (double)(float)p[i]; }

template <typename TI, typename TO>
__global__ void __launch_bounds__(EW_BLOCK)
convert_kernel(const TI *__restrict__ in, TO *__restrict__ out, int64_t count)
{
    for (int64_t idx = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; idx < count;
         idx += (int64_t)gridDim.x * EW_BLOCK) {
        const double v = load_as_double<TI>(in, idx);
        if constexpr (sizeof(TO) == 2)
            out[idx] = (TO)(float)v;     // double -> half/bfloat16 goes through float like torch
        else
            out[idx] = (TO)v;
    }
}

template <typename TI>
hipError_t convert_out(const TI *in, void *out, int out_dt, int64_t count, hipStream_t st)
{
    const int grid = ew_grid(count);
    switch (out_dt) {
    case NB_F16: hipLaunchKernelGGL((convert_kernel<TI, _Float16>), dim3(grid), dim3(EW_BLOCK), 0, st, in, (_Float16 *)out, count); break;
    case NB_BF16: hipLaunchKernelGGL((convert_kernel<TI, __bf16>), dim3(grid), dim3(EW_BLOCK), 0, st, in, (__bf16 *)out, count); break;
    case NB_F32: hipLaunchKernelGGL((convert_kernel<TI, float>), dim3(grid), dim3(EW_BLOCK), 0, st, in, (float *)out, count); break;
    case NB_F64: hipLaunchKernelGGL((convert_kernel<TI, double>), dim3(grid), dim3(EW_BLOCK), 0, st, in, (double *)out, count); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---- NaN-propagating min/max helpers (torch.min()/max() return NaN if any element is NaN) ----
__device__ __forceinline__ float nan_min(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : fminf(a, b); }
__device__ __forceinline__ float nan_max(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : fmaxf(a, b); }
__device__ __forceinline__ double nan_min(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : fmin(a, b); }
__device__ __forceinline__ double nan_max(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : fmax(a, b); }

template <typename T> __device__ __forceinline__ T shfl_xor_t(T v, int off);
template <> __device__ __forceinline__ float shfl_xor_t<float>(float v, int off) { return __shfl_xor(v, off, 64); }
template <> __device__ __forceinline__ double shfl_xor_t<double>(double v, int off) { return __shfl_xor(v, off, 64); }

// min/max over `count` elements in two short stages (a single block needs ~40 us for 131 072
// forces): stage 1 = up to MM_BLOCKS blocks write per-block partials, stage 2 = one block folds them.
constexpr int MM_BLOCKS = NB_MINMAX_BLOCKS;

template <typename T, bool LOGC>
__device__ __forceinline__ void minmax_block(const T *__restrict__ in, int64_t begin, int64_t end, int64_t stride,
                                             T min_val, T &mn, T &mx)
{
    // four independent loads in flight per thread
    int64_t i = begin;
    if (!LOGC)
        for (; i + 3 * stride < end; i += 4 * stride) {
            const T v0 = in[i], v1 = in[i + stride], v2 = in[i + 2 * stride], v3 = in[i + 3 * stride];
            mn = nan_min(nan_min(mn, v0), nan_min(v1, nan_min(v2, v3)));
            mx = nan_max(nan_max(mx, v0), nan_max(v1, nan_max(v2, v3)));
        }
    for (; i < end; i += stride) {
        T v = in[i];
        if (LOGC) {
            v = (v < min_val) ? min_val : v;
            if constexpr (sizeof(T) == 4) v = (float)log((double)v); else v = log(v);
        }
        mn = nan_min(mn, v);
        mx = nan_max(mx, v);
    }
}

template <typename T>
__device__ __forceinline__ void minmax_fold(T &mn, T &mx, T *s_mn, T *s_mx)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mn = nan_min(mn, shfl_xor_t<T>(mn, off));
        mx = nan_max(mx, shfl_xor_t<T>(mx, off));
    }
    if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { mn = nan_min(mn, s_mn[w]); mx = nan_max(mx, s_mx[w]); }
}

template <typename T, bool LOGC>
__global__ void __launch_bounds__(256)
minmax_stage1_kernel(const T *__restrict__ in, int64_t count, T min_val, double *__restrict__ partials)
{
    __shared__ T s_mn[4], s_mx[4];
    T mn = (T)__builtin_inf(), mx = -(T)__builtin_inf();
    minmax_block<T, LOGC>(in, (int64_t)blockIdx.x * 256 + threadIdx.x, count, (int64_t)gridDim.x * 256, min_val, mn, mx);
    minmax_fold<T>(mn, mx, s_mn, s_mx);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = (double)mn; partials[2 * blockIdx.x + 1] = (double)mx; }
}

__global__ void __launch_bounds__(256)
minmax_stage2_kernel(const double *__restrict__ partials, int nblocks, double *__restrict__ out)
{
    __shared__ double s_mn[4], s_mx[4];
    double mn = __builtin_inf(), mx = -__builtin_inf();
    for (int i = threadIdx.x; i < nblocks; i += 256) {
        mn = nan_min(mn, partials[2 * i]);
        mx = nan_max(mx, partials[2 * i + 1]);
    }
    minmax_fold<double>(mn, mx, s_mn, s_mx);
    if (threadIdx.x == 0) { out[0] = mn; out[1] = mx; }
}

// quantization.py:74-88 elementwise part, T arithmetic, one rounding per op
template <typename T>
__device__ __forceinline__ T lin_quant(T v, T mn, T range, T lm1, int *bin)
{
    T x, k;
    if constexpr (sizeof(T) == 4) {
        x = __fmul_rn(__fdiv_rn(__fsub_rn(v, mn), range), lm1);
        k = rintf(x);
        x = __fadd_rn(__fmul_rn(__fdiv_rn(k, lm1), range), mn);
    } else {
        x = __dmul_rn(__ddiv_rn(__dsub_rn(v, mn), range), lm1);
        k = rint(x);
        x = __dadd_rn(__dmul_rn(__ddiv_rn(k, lm1), range), mn);
    }
    if (bin) *bin = (k != k) ? -2 : (int)k;
    return x;
}

template <typename T>
__global__ void __launch_bounds__(EW_BLOCK)
grid_quantize_kernel(const T *in, T *out /* may alias in */, int64_t count, int levels,
                     const double *__restrict__ mn_mx, int16_t *__restrict__ bins)
{
    const T mn = (T)mn_mx[0], mx = (T)mn_mx[1];
    T range;
    if constexpr (sizeof(T) == 4) range = __fsub_rn(mx, mn); else range = __dsub_rn(mx, mn);
    const bool passthrough = range < (T)1e-10;
    const T lm1 = (T)(levels - 1);
    for (int64_t idx = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; idx < count;
         idx += (int64_t)gridDim.x * EW_BLOCK) {
        int b = -1;
        T v = in[idx];
        if (!passthrough) v = lin_quant<T>(v, mn, range, lm1, &b);
        out[idx] = v;
        if (bins) bins[idx] = (int16_t)b;
    }
}

// Force quantisation inside a step (simulation.py:115-116 -> quantization.py:74-88), second half: every block
// folds the stage-1 min/max partials itself (min/max are exact, so every block gets the same bounds), then
// quantises its elements; optionally the closing half kick (simulation.py:141) and the next step's opening kick +
// drift (:132,:135) ride along -- the same operations the separate launches perform.
__global__ void __launch_bounds__(256)
force_quant_finish_kernel(float *__restrict__ acc, int64_t count, int levels, const double *__restrict__ partials,
                          int nblocks, double *__restrict__ mn_mx, int16_t *__restrict__ bins, float *__restrict__ vel,
                          float *__restrict__ pos, float half_dt, float dt, int kick, float *__restrict__ packed, int np,
                          int dim)
{
    __shared__ double s_mn[4], s_mx[4];
    __shared__ double s_out[2];
    double dmn = __builtin_inf(), dmx = -__builtin_inf();
    for (int i = threadIdx.x; i < nblocks; i += 256) {
        dmn = nan_min(dmn, partials[2 * i]);
        dmx = nan_max(dmx, partials[2 * i + 1]);
    }
    minmax_fold<double>(dmn, dmx, s_mn, s_mx);
    if (threadIdx.x == 0) {
        s_out[0] = dmn;
        s_out[1] = dmx;
        if (blockIdx.x == 0) { mn_mx[0] = dmn; mn_mx[1] = dmx; }
    }
    __syncthreads();
    const float mn = (float)s_out[0], mx = (float)s_out[1];
    const float range = __fsub_rn(mx, mn);
    const bool passthrough = range < 1e-10f;
    const float lm1 = (float)(levels - 1);
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < count; idx += (int64_t)gridDim.x * 256) {
        int b = -1;
        float a = acc[idx];
        if (!passthrough) a = lin_quant<float>(a, mn, range, lm1, &b);
        acc[idx] = a;
        if (bins) bins[idx] = (int16_t)b;
        if (kick) {
            float v = axpy1(vel[idx], a, half_dt);
            if (kick == 2) {
                v = axpy1(v, a, half_dt);
                const float x = axpy1(pos[idx], v, dt);
                pos[idx] = x;
                // pair-symmetric path: the next evaluation's packed positions (component arrays) in the same pass
                if (packed) packed[(size_t)(idx % dim) * np + idx / dim] = x;
            }
            vel[idx] = v;
        }
    }
}

// quantization.py:91-127 elementwise part (tensor-level hook: log/exp per element)
template <typename T>
__global__ void __launch_bounds__(EW_BLOCK)
grid_quantize_safe_kernel(const T *__restrict__ in, T *__restrict__ out, int64_t count, int levels, T min_val,
                          const double *__restrict__ mn_mx)
{
    const T lmin = (T)mn_mx[0], lmax = (T)mn_mx[1];
    T range;
    if constexpr (sizeof(T) == 4) range = __fsub_rn(lmax, lmin); else range = __dsub_rn(lmax, lmin);
    const bool passthrough = range < (T)1e-10;
    const T lm1 = (T)(levels - 1);
    for (int64_t idx = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; idx < count;
         idx += (int64_t)gridDim.x * EW_BLOCK) {
        T v = in[idx];
        v = (v < min_val) ? min_val : v;
        if (!passthrough) {
            if constexpr (sizeof(T) == 4) {
                const float lt = (float)log((double)v);
                float x = __fmul_rn(__fdiv_rn(__fsub_rn(lt, lmin), range), lm1);
                const float k = rintf(x);
                x = __fadd_rn(__fmul_rn(__fdiv_rn(k, lm1), range), lmin);
                v = (float)exp((double)x);
            } else {
                const double lt = log(v);
                double x = __dmul_rn(__ddiv_rn(__dsub_rn(lt, lmin), range), lm1);
                const double k = rint(x);
                x = __dadd_rn(__dmul_rn(__ddiv_rn(k, lm1), range), lmin);
                v = exp(x);
            }
            v = (v < min_val) ? min_val : v;
        }
        out[idx] = v;
    }
}

// cast hooks: FLOAT64 -> double, FLOAT32 -> float, BF16/F16 -> round trip through the half type
template <typename TI, typename TO, int MODE>
__global__ void __launch_bounds__(EW_BLOCK)
cast_hook_kernel(const TI *__restrict__ in, TO *__restrict__ out, int64_t count)
{
    for (int64_t idx = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; idx < count;
         idx += (int64_t)gridDim.x * EW_BLOCK) {
        const TI v = in[idx];
        if constexpr (MODE == NB_FLOAT64) out[idx] = (TO)v;
        else if constexpr (MODE == NB_FLOAT32) out[idx] = (TO)(float)v;
        else if constexpr (MODE == NB_BFLOAT16) out[idx] = (TO)(float)(__bf16)(float)v;
        else out[idx] = (TO)(float)(_Float16)(float)v;
    }
}

// ---- energies ----------------------------------------------------------------------------------
// HP: state tensors typed float16 / bfloat16 (values stored as fp32): every op rounds to that type;
// HP = -1: no extra rounding (note NB_F16 == 0, so "none" must not be 0)
template <int HP> __device__ __forceinline__ float round_hp(float x)
{
    if (HP == NB_F16) return (float)(_Float16)x;
    if (HP == NB_BF16) return (float)(__bf16)x;
    return x;
}

template <typename T>
__device__ __forceinline__ double block_sum(double v, double *s_red)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_red[w];
    return t;
}

// 0.5 * sum m * |v|^2 (simulation.py:170-174).  VF32: velocities hold fp32-typed values.
template <typename T, bool VF32, int HP = -1>
__global__ void __launch_bounds__(NB_BLOCK)
kinetic_kernel(const T *__restrict__ vel, const T *__restrict__ mass, int n, int dim, double *__restrict__ part)
{
    __shared__ double s_red[NB_BLOCK / 64];
    double s = 0.0;
    for (int i = blockIdx.x * NB_BLOCK + threadIdx.x; i < n; i += gridDim.x * NB_BLOCK) {
        if (VF32 || sizeof(T) == 4) {
            float v2 = 0.0f;
            for (int k = 0; k < dim; ++k) {
                const float v = (float)vel[(size_t)i * dim + k];
                const float sq = round_hp<HP>(__fmul_rn(v, v));
                v2 = (k == 0) ? sq : __fadd_rn(v2, sq);
            }
            v2 = round_hp<HP>(v2);
            s += (double)round_hp<HP>(__fmul_rn((float)mass[i], v2));
        } else {
            double v2 = 0.0;
            for (int k = 0; k < dim; ++k) {
                const double v = (double)vel[(size_t)i * dim + k];
                v2 += v * v;
            }
            s += (double)mass[i] * v2;
        }
    }
    const double t = block_sum<T>(s, s_red);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

__global__ void __launch_bounds__(NB_BLOCK)
final_sum_kernel(const double *__restrict__ part, int count, double *__restrict__ out)
{
    __shared__ double s_red[NB_BLOCK / 64];
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += NB_BLOCK) s += part[i];
    const double t = block_sum<double>(s, s_red);
    if (threadIdx.x == 0) *out = t;
}

// sum_{i<j} m_i m_j / sqrt(r2 + eps2) over this rank's sources (simulation.py:176-192)
template <typename T, int D, bool PA_F32, int HP = -1>
__global__ void __launch_bounds__(NB_BLOCK)
potential_kernel(const T *__restrict__ pos, const T *__restrict__ mass, ForceGeom g, double eps2, float eps2_f,
                 int mass_dt, double *__restrict__ part)
{
    __shared__ T sj[D + 1][NB_TJ];
    __shared__ double s_red[NB_BLOCK / 64];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * NB_BLOCK + tid;
    const int ic = i < g.n ? i : g.n - 1;
    T xi[D];
#pragma unroll
    for (int k = 0; k < D; ++k) xi[k] = pos[(size_t)ic * D + k];
    const T mi = (i < g.n) ? mass[ic] : (T)0;

    const int j_lo = g.j_begin + blockIdx.y * g.chunk_len;
    const int j_hi = min(j_lo + g.chunk_len, g.j_end);
    const int i_first = blockIdx.x * NB_BLOCK;
    double s = 0.0;
    for (int jt = j_lo; jt < j_hi; jt += NB_TJ) {
        const int cnt = min(NB_TJ, j_hi - jt);
        if (jt + cnt - 1 <= i_first) continue;   // whole tile at or below the diagonal (block-uniform)
        {
            int j = jt + tid;
            j = j < j_hi ? j : j_hi - 1;
#pragma unroll
            for (int k = 0; k < D; ++k) sj[k][tid] = pos[(size_t)j * D + k];
            sj[D][tid] = mass[j];
        }
        __syncthreads();
#pragma unroll 4
        for (int jj = 0; jj < cnt; ++jj) {
            const bool take = (jt + jj) > i;
            double term;
            if (PA_F32 || sizeof(T) == 4) {
                float d2 = 0.0f;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const float df = round_hp<HP>(__fsub_rn((float)sj[k][jj], (float)xi[k]));
                    const float sq = round_hp<HP>(__fmul_rn(df, df));
                    d2 = (k == 0) ? sq : __fadd_rn(d2, sq);
                }
                d2 = round_hp<HP>(d2);
                const float dist = round_hp<HP>(__builtin_sqrtf(round_hp<HP>(__fadd_rn(d2, eps2_f))));
                term = (double)round_hp<HP>(__fdiv_rn(round_hp<HP>(nbdev::mass_prod_f32((float)mi, (float)sj[D][jj], mass_dt)), dist));
            } else {
                double q = eps2;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const double df = (double)sj[k][jj] - (double)xi[k];
                    q = __builtin_fma(df, df, q);
                }
                // 1/sqrt(q): v_rsq_f64 seed + one third-order step (error < 1 ulp)
                const double y0 = __builtin_amdgcn_rsq(q);
                const double e = __builtin_fma(-q * y0, y0, 1.0);
                const double y = __builtin_fma(y0 * e, __builtin_fma(e, 0.375, 0.5), y0);
                // mass_prod keeps the masses' dtype upstream (simulation.py:185): fp32-typed masses
                // give an fp32-rounded product even when positions are already fp64
                const double mp = mass_dt != NB_F64 ? (double)nbdev::mass_prod_f32((float)mi, (float)sj[D][jj], mass_dt)
                                                    : (double)mi * (double)sj[D][jj];
                term = mp * y;
            }
            s += take ? term : 0.0;
        }
        __syncthreads();
    }
    const double t = block_sum<T>(s, s_red);
    if (tid == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
hipError_t nb_launch_reduce(const double *partial, int nchunks, int64_t count, void *acc, int is_f64, void *vel,
                            double half_dt, int do_kick, void *pos, double dt, hipStream_t st)
{
    const int grid = ew_grid(count);
    if (is_f64)
        hipLaunchKernelGGL((reduce_kernel<double>), dim3(grid), dim3(EW_BLOCK), 0, st, partial, nchunks, count,
                           (double *)acc, (double *)vel, half_dt, do_kick, (double *)pos, dt);
    else
        hipLaunchKernelGGL((reduce_kernel<float>), dim3(grid), dim3(EW_BLOCK), 0, st, partial, nchunks, count,
                           (float *)acc, (float *)vel, (float)half_dt, do_kick, (float *)pos, (float)dt);
    return hipGetLastError();
}

hipError_t nb_launch_axpy(void *y, const void *x, double scalar, int64_t count, int is_f64, hipStream_t st)
{
    const int grid = ew_grid(count);
    if (is_f64)
        hipLaunchKernelGGL((axpy_kernel<double>), dim3(grid), dim3(EW_BLOCK), 0, st, (double *)y, (const double *)x,
                           scalar, count);
    else
        hipLaunchKernelGGL((axpy_kernel<float>), dim3(grid), dim3(EW_BLOCK), 0, st, (float *)y, (const float *)x,
                           (float)scalar, count);
    return hipGetLastError();
}

hipError_t nb_launch_kick_drift(void *pos, void *vel, const void *acc, double half_dt, double dt, int64_t count,
                                int is_f64, hipStream_t st)
{
    const int grid = ew_grid(count);
    if (is_f64)
        hipLaunchKernelGGL((kick_drift_kernel<double>), dim3(grid), dim3(EW_BLOCK), 0, st, (double *)pos,
                           (double *)vel, (const double *)acc, half_dt, dt, count);
    else
        hipLaunchKernelGGL((kick_drift_kernel<float>), dim3(grid), dim3(EW_BLOCK), 0, st, (float *)pos,
                           (float *)vel, (const float *)acc, (float)half_dt, (float)dt, count);
    return hipGetLastError();
}

hipError_t nb_launch_convert(const void *in, int in_dt, void *out, int out_dt, int64_t count, hipStream_t st)
{
    switch (in_dt) {
    case NB_F16: return convert_out<_Float16>((const _Float16 *)in, out, out_dt, count, st);
    case NB_BF16: return convert_out<__bf16>((const __bf16 *)in, out, out_dt, count, st);
    case NB_F32: return convert_out<float>((const float *)in, out, out_dt, count, st);
    case NB_F64: return convert_out<double>((const double *)in, out, out_dt, count, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t nb_launch_minmax_generic(const void *in, int is_f64, int64_t count, int log_clamped, double min_val,
                                    double *mn_mx, double *partials, hipStream_t st)
{
    int blocks = (int)((count + 1023) / 1024);          // >= 4 elements per thread
    blocks = blocks < 1 ? 1 : (blocks > MM_BLOCKS ? MM_BLOCKS : blocks);
#define NB_MM(TT, LL) \
    hipLaunchKernelGGL((minmax_stage1_kernel<TT, LL>), dim3(blocks), dim3(256), 0, st, (const TT *)in, count, (TT)min_val, partials)
    if (is_f64) { if (log_clamped) NB_MM(double, true); else NB_MM(double, false); }
    else        { if (log_clamped) NB_MM(float, true); else NB_MM(float, false); }
#undef NB_MM
    hipLaunchKernelGGL(minmax_stage2_kernel, dim3(1), dim3(256), 0, st, partials, blocks, mn_mx);
    return hipGetLastError();
}

hipError_t nb_launch_grid_quantize(const void *in, void *out, int is_f64, int64_t count, int levels,
                                   const double *mn_mx, hipStream_t st)
{
    const int grid = ew_grid(count);
    if (is_f64)
        hipLaunchKernelGGL((grid_quantize_kernel<double>), dim3(grid), dim3(EW_BLOCK), 0, st, (const double *)in,
                           (double *)out, count, levels, mn_mx, (int16_t *)nullptr);
    else
        hipLaunchKernelGGL((grid_quantize_kernel<float>), dim3(grid), dim3(EW_BLOCK), 0, st, (const float *)in,
                           (float *)out, count, levels, mn_mx, (int16_t *)nullptr);
    return hipGetLastError();
}

// same kernel, fp32, with optional bin output (force quantisation inside the step)
hipError_t nb_launch_force_quant_bins(const float *in, float *out, int64_t count, int levels, const double *mn_mx,
                                      int16_t *bins, hipStream_t st)
{
    const int grid = ew_grid(count);
    hipLaunchKernelGGL((grid_quantize_kernel<float>), dim3(grid), dim3(EW_BLOCK), 0, st, in, out, count, levels,
                       mn_mx, bins);
    return hipGetLastError();
}

// min/max of the summed forces + quantisation (+ kicks) of the step: two launches
hipError_t nb_launch_force_quant_step(float *acc, int64_t count, int levels, double *mn_mx, double *partials,
                                      int16_t *bins, float *vel, float *pos, double half_dt, double dt, int kick,
                                      float *packed, int np, int dim, hipStream_t st)
{
    int blocks = (int)((count + 1023) / 1024);
    blocks = blocks < 1 ? 1 : (blocks > MM_BLOCKS ? MM_BLOCKS : blocks);
    hipLaunchKernelGGL((minmax_stage1_kernel<float, false>), dim3(blocks), dim3(256), 0, st, (const float *)acc, count,
                       0.0f, partials);
    int grid = (int)((count + 255) / 256);
    grid = grid > 2048 ? 2048 : grid;
    hipLaunchKernelGGL(force_quant_finish_kernel, dim3(grid), dim3(256), 0, st, acc, count, levels, partials, blocks,
                       mn_mx, bins, vel, pos, (float)half_dt, (float)dt, kick, packed, np, dim);
    return hipGetLastError();
}

hipError_t nb_launch_force_quant_finish(float *acc, int64_t count, int levels, const double *partials, int nblocks,
                                        double *mn_mx, int16_t *bins, float *vel, float *pos, double half_dt, double dt,
                                        int kick, hipStream_t st, float *packed, int np, int dim)
{
    int grid = (int)((count + 255) / 256);
    grid = grid > 2048 ? 2048 : grid;
    hipLaunchKernelGGL(force_quant_finish_kernel, dim3(grid), dim3(256), 0, st, acc, count, levels, partials, nblocks, mn_mx,
                       bins, vel, pos, (float)half_dt, (float)dt, kick, packed, np, dim);
    return hipGetLastError();
}

hipError_t nb_launch_grid_quantize_safe(const void *in, void *out, int is_f64, int64_t count, int levels,
                                        double min_val, const double *mn_mx, hipStream_t st)
{
    const int grid = ew_grid(count);
    if (is_f64)
        hipLaunchKernelGGL((grid_quantize_safe_kernel<double>), dim3(grid), dim3(EW_BLOCK), 0, st,
                           (const double *)in, (double *)out, count, levels, min_val, mn_mx);
    else
        hipLaunchKernelGGL((grid_quantize_safe_kernel<float>), dim3(grid), dim3(EW_BLOCK), 0, st, (const float *)in,
                           (float *)out, count, levels, (float)min_val, mn_mx);
    return hipGetLastError();
}

hipError_t nb_launch_cast_hook(const void *in, int in_dt, void *out, int mode, int64_t count, hipStream_t st)
{
    const int grid = ew_grid(count);
#define NB_CAST(TI, TO, MODE) \
    hipLaunchKernelGGL((cast_hook_kernel<TI, TO, MODE>), dim3(grid), dim3(EW_BLOCK), 0, st, (const TI *)in, (TO *)out, count)
    if (in_dt == NB_F32) {
        switch (mode) {
        case NB_FLOAT64: NB_CAST(float, double, NB_FLOAT64); break;
        case NB_FLOAT32: NB_CAST(float, float, NB_FLOAT32); break;
        case NB_BFLOAT16: NB_CAST(float, float, NB_BFLOAT16); break;
        case NB_FLOAT16: NB_CAST(float, float, NB_FLOAT16); break;
        default: return hipErrorInvalidValue;
        }
    } else if (in_dt == NB_F64) {
        switch (mode) {
        case NB_FLOAT64: NB_CAST(double, double, NB_FLOAT64); break;
        case NB_FLOAT32: NB_CAST(double, float, NB_FLOAT32); break;
        case NB_BFLOAT16: NB_CAST(double, float, NB_BFLOAT16); break;
        case NB_FLOAT16: NB_CAST(double, float, NB_FLOAT16); break;
        default: return hipErrorInvalidValue;
        }
    } else {
        return hipErrorInvalidValue;
    }
#undef NB_CAST
    return hipGetLastError();
}

hipError_t nb_launch_final_sum(const double *part, int count, double *out, hipStream_t st)
{
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(NB_BLOCK), 0, st, part, count, out);
    return hipGetLastError();
}

hipError_t nb_launch_kinetic(const void *vel, const void *mass, int n, int dim, int is_f64, int vel_f32_logical,
                             int half_pa, double *scratch, double *out, hipStream_t st)
{
    int blocks = (n + NB_BLOCK - 1) / NB_BLOCK;
    if (blocks > 1024) blocks = 1024;
    if (is_f64) {
        if (half_pa == NB_F16)
            hipLaunchKernelGGL((kinetic_kernel<double, true, NB_F16>), dim3(blocks), dim3(NB_BLOCK), 0, st, (const double *)vel, (const double *)mass, n, dim, scratch);
        else if (half_pa == NB_BF16)
            hipLaunchKernelGGL((kinetic_kernel<double, true, NB_BF16>), dim3(blocks), dim3(NB_BLOCK), 0, st, (const double *)vel, (const double *)mass, n, dim, scratch);
        else if (vel_f32_logical)
            hipLaunchKernelGGL((kinetic_kernel<double, true>), dim3(blocks), dim3(NB_BLOCK), 0, st, (const double *)vel, (const double *)mass, n, dim, scratch);
        else
            hipLaunchKernelGGL((kinetic_kernel<double, false>), dim3(blocks), dim3(NB_BLOCK), 0, st, (const double *)vel, (const double *)mass, n, dim, scratch);
    } else if (half_pa == NB_F16) {
        hipLaunchKernelGGL((kinetic_kernel<float, true, NB_F16>), dim3(blocks), dim3(NB_BLOCK), 0, st, (const float *)vel, (const float *)mass, n, dim, scratch);
    } else if (half_pa == NB_BF16) {
        hipLaunchKernelGGL((kinetic_kernel<float, true, NB_BF16>), dim3(blocks), dim3(NB_BLOCK), 0, st, (const float *)vel, (const float *)mass, n, dim, scratch);
    } else {
        hipLaunchKernelGGL((kinetic_kernel<float, true>), dim3(blocks), dim3(NB_BLOCK), 0, st, (const float *)vel, (const float *)mass, n, dim, scratch);
    }
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(NB_BLOCK), 0, st, scratch, blocks, out);
    return hipGetLastError();
}

hipError_t nb_launch_potential(const void *pos, const void *mass, const ForceGeom &g, int dim, int is_f64,
                               int pa_f32, int mass_dt, int half_pa, double eps2_py, float eps2_half,
                               double *scratch, double *out, hipStream_t st)
{
    const dim3 grid((g.n + NB_BLOCK - 1) / NB_BLOCK, g.nchunks);
    const float e32 = (half_pa >= 0) ? eps2_half : (float)eps2_py;
#define NB_PE(T, D, PA) \
    hipLaunchKernelGGL((potential_kernel<T, D, PA>), grid, dim3(NB_BLOCK), 0, st, (const T *)pos, (const T *)mass, g, eps2_py, e32, mass_dt, scratch)
    if (dim != 2 && dim != 3) return hipErrorInvalidValue;
    if (is_f64 && half_pa == NB_F16) {
        if (dim == 2) hipLaunchKernelGGL((potential_kernel<double, 2, true, NB_F16>), grid, dim3(NB_BLOCK), 0, st, (const double *)pos, (const double *)mass, g, eps2_py, e32, mass_dt, scratch);
        else hipLaunchKernelGGL((potential_kernel<double, 3, true, NB_F16>), grid, dim3(NB_BLOCK), 0, st, (const double *)pos, (const double *)mass, g, eps2_py, e32, mass_dt, scratch);
    } else if (is_f64 && half_pa == NB_BF16) {
        if (dim == 2) hipLaunchKernelGGL((potential_kernel<double, 2, true, NB_BF16>), grid, dim3(NB_BLOCK), 0, st, (const double *)pos, (const double *)mass, g, eps2_py, e32, mass_dt, scratch);
        else hipLaunchKernelGGL((potential_kernel<double, 3, true, NB_BF16>), grid, dim3(NB_BLOCK), 0, st, (const double *)pos, (const double *)mass, g, eps2_py, e32, mass_dt, scratch);
    } else if (is_f64) {
        if (pa_f32) { if (dim == 2) NB_PE(double, 2, true); else NB_PE(double, 3, true); }
        else        { if (dim == 2) NB_PE(double, 2, false); else NB_PE(double, 3, false); }
    } else if (half_pa == NB_F16) {
        if (dim == 2) hipLaunchKernelGGL((potential_kernel<float, 2, true, NB_F16>), grid, dim3(NB_BLOCK), 0, st, (const float *)pos, (const float *)mass, g, eps2_py, e32, mass_dt, scratch);
        else hipLaunchKernelGGL((potential_kernel<float, 3, true, NB_F16>), grid, dim3(NB_BLOCK), 0, st, (const float *)pos, (const float *)mass, g, eps2_py, e32, mass_dt, scratch);
    } else if (half_pa == NB_BF16) {
        if (dim == 2) hipLaunchKernelGGL((potential_kernel<float, 2, true, NB_BF16>), grid, dim3(NB_BLOCK), 0, st, (const float *)pos, (const float *)mass, g, eps2_py, e32, mass_dt, scratch);
        else hipLaunchKernelGGL((potential_kernel<float, 3, true, NB_BF16>), grid, dim3(NB_BLOCK), 0, st, (const float *)pos, (const float *)mass, g, eps2_py, e32, mass_dt, scratch);
    } else {
        if (dim == 2) NB_PE(float, 2, true); else NB_PE(float, 3, true);
    }
#undef NB_PE
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(NB_BLOCK), 0, st, scratch, (int)(grid.x * grid.y), out);
    return hipGetLastError();
}
