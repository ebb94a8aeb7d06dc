// nb_generic.hip -- dtype-faithful all-pairs force evaluation for the combinations the tuned kernels do not cover.
//
// The reference is dtype-polymorphic PyTorch (simulation.py:74-118, quantization.py:21-127): every tensor op rounds
// to its tensor's dtype and dtypes promote through the chain
//     P (positions) -> Q (hook output) -> W = promote(Q, M) (x masses) -> W2 = promote(W, f32) (x (1 - eye))
//                  -> A = promote(W2, P) (x diff, summed).
// The tuned kernels implement the chains every script of the reference builds (fp32 or fp64 state under the seven
// modes, half-typed state under the cast modes).  Everything else the stock class ACCEPTS -- fp64 masses or
// velocities beside fp32 positions, the grid modes (INT8 / INT4 / CUSTOM) on fp64 or float16 / bfloat16 state --
// runs here: one-sided, LDS-tiled, every elementary operation evaluated in fp64 and rounded once to the dtype torch
// would hold it in (identical to native arithmetic in that dtype for + - * / sqrt: 53 >= 2*24 + 2 bits), log / exp /
// pow in fp64 then rounded.  Slow (a log, an exp and a pow per pair in the grid modes) and only reached by unusual
// inputs; correct by construction rather than fast.  Golden: tests/golden/g15_dtype_combos.npz.
#include "nb_internal.h"

namespace {

__device__ __forceinline__ double rnd_small(double x, int mant, int emin, int emax)
{
    if (x == 0.0 || x != x || __builtin_isinf(x)) return x;
    int e;
    (void)frexp(fabs(x), &e);
    const int ue = e - 1;
    const int q = ue < emin ? emin : ue;
    const double ulp = ldexp(1.0, q - mant);
    double r = rint(fabs(x) / ulp) * ulp;
    if (r > ldexp(2.0 - ldexp(1.0, -mant), emax)) r = __builtin_inf();
    return x < 0 ? -r : r;
}
__device__ __forceinline__ double rnd(int T, double x)
{
    switch (T) {
    case NB_F64: return x;
    case NB_F32: return (double)(float)x;
    case NB_F16: return rnd_small(x, 10, -14, 15);
    default: return rnd_small(x, 7, -126, 127);
    }
}
__device__ __forceinline__ int opmath(int T) { return T == NB_F64 ? NB_F64 : NB_F32; }
__device__ __forceinline__ double clamp_min(double x, double lo) { return (x != x) ? x : (x < lo ? lo : x); }

// simulation.py:83-86
template <int D>
__device__ __forceinline__ double pair_r2(int P, const double *xi, const double *xj, double eps2_P, double *diff)
{
    const int O = opmath(P);
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        diff[k] = rnd(P, xj[k] - xi[k]);
        const double sq = rnd(P, diff[k] * diff[k]);
        s = (k == 0) ? sq : rnd(O, s + sq);
    }
    s = rnd(P, s);
    return rnd(P, s + eps2_P);
}

// quantization.py:91-127, scalar pieces in dtype T
__device__ __forceinline__ double gqs_log(int T, double t, double min_T) { return rnd(T, log(clamp_min(t, min_T))); }

struct GenScalars {          // device scalars of one evaluation
    double r2max;            // max over all pairs of r2 (dtype P); NaN if any r2 is NaN
    unsigned long long r2max_bits;   // atomicMax target (non-negative doubles order as unsigned integers)
    int nan_flag;
};

// all-pairs maximum of r2 (the grid's log_max, quantization.py:113); the minimum is the diagonal: r2 = eps2
template <typename S, int D>
__global__ void __launch_bounds__(NB_BLOCK)
generic_r2max_kernel(const S *__restrict__ pos, int n, int P, double eps2_py, GenScalars *__restrict__ sc)
{
    const double eps2_P = rnd(P, eps2_py);
    __shared__ double sj[D][NB_TJ];
    __shared__ unsigned long long s_max;
    __shared__ int s_nan;
    const int tid = threadIdx.x;
    if (tid == 0) { s_max = 0ull; s_nan = 0; }
    int i = blockIdx.x * NB_BLOCK + tid;
    const bool live = i < n;
    i = live ? i : n - 1;
    double xi[D], diff[D];
#pragma unroll
    for (int k = 0; k < D; ++k) xi[k] = (double)pos[(size_t)i * D + k];
    double mx = 0.0;
    bool nan = false;
    for (int jt = 0; jt < n; jt += NB_TJ) {
        int j = jt + tid;
        j = j < n ? j : n - 1;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < D; ++k) sj[k][tid] = (double)pos[(size_t)j * D + k];
        __syncthreads();
        const int cnt = min(NB_TJ, n - jt);
        for (int jj = 0; jj < cnt; ++jj) {
            double xj[D];
#pragma unroll
            for (int k = 0; k < D; ++k) xj[k] = sj[k][jj];
            const double r2 = pair_r2<D>(P, xi, xj, eps2_P, diff);
            nan |= (r2 != r2);
            mx = r2 > mx ? r2 : mx;
        }
    }
    if (live) {
        atomicMax(&s_max, (unsigned long long)__double_as_longlong(mx));
        if (nan) atomicOr(&s_nan, 1);
    }
    __syncthreads();
    if (tid == 0) {
        atomicMax(&sc->r2max_bits, s_max);
        if (s_nan) atomicOr(&sc->nan_flag, 1);
    }
}

struct GenArgs {
    int n, j_begin, j_end, chunk_len;
    int P, M, mode, levels;
    double G, eps2_py;
};

// partial[chunk][n][D]: sums over this chunk's sources of rnd(A, w * diff), accumulated in fp64
template <typename S, int D>
__global__ void __launch_bounds__(NB_BLOCK)
generic_force_kernel(const S *__restrict__ pos, const S *__restrict__ mass, double *__restrict__ partial, GenArgs g,
                     const GenScalars *__restrict__ sc)
{
    __shared__ double sj[D + 1][NB_TJ];
    const int tid = threadIdx.x;
    int i = blockIdx.x * NB_BLOCK + tid;
    const bool live = i < g.n;
    i = live ? i : g.n - 1;
    const int P = g.P, M = g.M, mode = g.mode;
    const bool grid = mode >= NB_INT8_SIM;
    // dtype chain (see the header comment)
    int Q = P;
    if (mode == NB_FLOAT64) Q = NB_F64;
    else if (mode <= NB_FLOAT16) Q = NB_F32;
    const int W = (Q == M) ? Q : ((Q == NB_F64 || M == NB_F64) ? NB_F64 : NB_F32);
    const int W2 = (W == NB_F64) ? NB_F64 : NB_F32;
    const int A = (W2 == NB_F64 || P == NB_F64) ? NB_F64 : NB_F32;
    const double eps2_P = rnd(P, g.eps2_py);
    const double min_P = rnd(P, 0.01);
    // A Python scalar that MULTIPLIES / DIVIDES a half tensor stays in float (torch's CPU mul / div kernels take the
    // scalar operand in opmath precision; one that is ADDED is rounded to the tensor's dtype first -- measured on
    // torch 2.10 by test_torch_scalar_semantics_on_half_tensors in the CPU test suite).  Round 2 rounded G and
    // (levels - 1) to the half type here: G = 0.001 became 0.0010004 in float16, a 4e-4 error of every force, which
    // showed up as 4 % of the INT8 force values in a neighbouring force bin.
    const double Gs = rnd(opmath(Q), g.G);
    const int L = g.levels;
    double lmin = 0.0, lmax = 0.0, range = 0.0, lm1 = 0.0;
    bool degenerate = false;
    if (grid) {
        const double r2max = sc->nan_flag ? __builtin_nan("") : __longlong_as_double((long long)sc->r2max_bits);
        lmin = gqs_log(P, eps2_P, min_P);         // the diagonal (r2 = eps2) is part of the N x N tensor
        lmax = gqs_log(P, r2max, min_P);
        range = rnd(P, lmax - lmin);
        lm1 = rnd(opmath(P), (double)(L - 1));    // scalar of `* (levels - 1)` / `/ (levels - 1)`: opmath, see Gs
        degenerate = range < 1e-10;               // NaN compares false: quantised values become NaN like upstream
    }
    double xi[D], acc[D];
#pragma unroll
    for (int k = 0; k < D; ++k) { xi[k] = (double)pos[(size_t)i * D + k]; acc[k] = 0.0; }

    const int j_lo = g.j_begin + blockIdx.y * g.chunk_len;
    const int j_hi = min(j_lo + g.chunk_len, g.j_end);
    for (int jt = j_lo; jt < j_hi; jt += NB_TJ) {
        int j = jt + tid;
        j = j < j_hi ? j : j_hi - 1;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < D; ++k) sj[k][tid] = (double)pos[(size_t)j * D + k];
        sj[D][tid] = (double)mass[j];
        __syncthreads();
        const int cnt = min(NB_TJ, j_hi - jt);
        for (int jj = 0; jj < cnt; ++jj) {
            double xj[D], diff[D];
#pragma unroll
            for (int k = 0; k < D; ++k) xj[k] = sj[k][jj];
            const double r2 = pair_r2<D>(P, xi, xj, eps2_P, diff);
            double q;
            if (!grid) {
                // quantization.py:43-56; torch converts double -> half through float
                if (mode == NB_FLOAT64) q = r2;
                else if (mode == NB_FLOAT32) q = (double)(float)r2;
                else if (mode == NB_BFLOAT16) q = rnd(NB_BF16, (double)(float)r2);
                else q = rnd(NB_F16, (double)(float)r2);
            } else if (degenerate) {
                q = clamp_min(r2, min_P);
            } else {
                const double lt = gqs_log(P, r2, min_P);
                double nrm = rnd(P, rnd(P, lt - lmin) / range);
                nrm = rnd(P, nrm * lm1);
                const double kq = rint(nrm);                              // torch.round: half to even
                double v = rnd(P, kq / lm1);
                v = rnd(P, v * range);
                v = rnd(P, v + lmin);
                q = clamp_min(rnd(P, exp(v)), min_P);
            }
            const double p = rnd(Q, pow(q, 1.5));                         // simulation.py:97
            double w = rnd(Q, rnd(Q, 1.0 / p) * Gs);                      // :101 reciprocal() * G
            w = rnd(W, w * sj[D][jj]);                                    // :105
            w = rnd(W2, w * ((jt + jj == i) ? 0.0 : 1.0));                // :108 (inf * 0 = NaN like upstream)
#pragma unroll
            for (int k = 0; k < D; ++k) acc[k] += rnd(A, w * diff[k]);    // :112, summed in fp64
        }
    }
    if (live) {
        double *out = partial + (size_t)blockIdx.y * g.n * D;
#pragma unroll
        for (int k = 0; k < D; ++k) out[(size_t)i * D + k] = acc[k];
    }
}

// acc[i] = rnd(A, sum over chunks) into storage S
template <typename S>
__global__ void __launch_bounds__(NB_BLOCK)
generic_finish_kernel(const double *__restrict__ partial, int nchunks, int64_t count, int A, S *__restrict__ acc)
{
    const int64_t e = (int64_t)blockIdx.x * NB_BLOCK + threadIdx.x;
    if (e >= count) return;
    double s = 0.0;
    for (int c = 0; c < nchunks; ++c) s += partial[(size_t)c * count + e];
    acc[e] = (S)rnd(A, s);
}

__global__ void generic_reset_kernel(GenScalars *sc)
{
    sc->r2max_bits = 0ull;
    sc->nan_flag = 0;
    sc->r2max = 0.0;
}

}  // namespace

size_t nb_generic_scalars_bytes() { return sizeof(GenScalars); }

// r2max -> *sc (device); the caller may all-reduce sc->r2max_bits (max) across source shards before the force launch
hipError_t nb_launch_generic_r2max(const void *pos, int storage_f64, int n, int dim, int P, double eps2_py, void *sc,
                                   hipStream_t st)
{
    hipLaunchKernelGGL(generic_reset_kernel, dim3(1), dim3(1), 0, st, (GenScalars *)sc);
    const int blocks = (n + NB_BLOCK - 1) / NB_BLOCK;
#define NB_GR(SS, DD) hipLaunchKernelGGL((generic_r2max_kernel<SS, DD>), dim3(blocks), dim3(NB_BLOCK), 0, st, (const SS *)pos, n, P, eps2_py, (GenScalars *)sc)
    if (storage_f64) { if (dim == 2) NB_GR(double, 2); else NB_GR(double, 3); }
    else             { if (dim == 2) NB_GR(float, 2); else NB_GR(float, 3); }
#undef NB_GR
    return hipGetLastError();
}

hipError_t nb_launch_generic_force(const void *pos, const void *mass, int storage_f64, double *partial, const ForceGeom &geom,
                                   int dim, int P, int M, int mode, int levels, double G, double eps2_py, const void *sc,
                                   void *acc, int A, hipStream_t st)
{
    GenArgs g{geom.n, geom.j_begin, geom.j_end, geom.chunk_len, P, M, mode, levels, G, eps2_py};
    const dim3 grid((geom.n + NB_BLOCK - 1) / NB_BLOCK, geom.nchunks);
#define NB_GF(SS, DD) hipLaunchKernelGGL((generic_force_kernel<SS, DD>), grid, dim3(NB_BLOCK), 0, st, (const SS *)pos, (const SS *)mass, partial, g, (const GenScalars *)sc)
    if (storage_f64) { if (dim == 2) NB_GF(double, 2); else NB_GF(double, 3); }
    else             { if (dim == 2) NB_GF(float, 2); else NB_GF(float, 3); }
#undef NB_GF
    const int64_t count = (int64_t)geom.n * dim;
    const int blocks = (int)((count + NB_BLOCK - 1) / NB_BLOCK);
    if (storage_f64)
        hipLaunchKernelGGL((generic_finish_kernel<double>), dim3(blocks), dim3(NB_BLOCK), 0, st, partial, geom.nchunks, count, A, (double *)acc);
    else
        hipLaunchKernelGGL((generic_finish_kernel<float>), dim3(blocks), dim3(NB_BLOCK), 0, st, partial, geom.nchunks, count, A, (float *)acc);
    return hipGetLastError();
}
