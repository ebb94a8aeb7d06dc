// nb_force_sym.hip -- pair-symmetric all-pairs force kernel for gfx950 (fp64 state).
//
// Same mathematics as force_f64_kernel (reference simulation.py:74-118, FLOAT64 mode), but
// every UNORDERED pair {a, b} is evaluated once and applied to both particles
// (a_a += G m_b w d,  a_b -= G m_a w d), which halves the q^(-3/2) evaluations -- the dominant
// cost of this VALU-bound kernel (DESIGN.md "instruction budget": 16 fp64 ops per unordered
// pair instead of 13 per ordered pair).  The reference treats the summation order as free
// (SURVEY.md section 2, row 20), so the result is the same sum in a different, FIXED order.
//
// Scheme (no LDS, no barriers in the pair loop):
//   particles are cut into tiles of B = 64*R; a wavefront keeps one target tile I in registers
//   (R particles per lane) for its whole life and walks source tiles J >= I.  The J tile is held
//   one particle-set per lane too, together with ITS accumulators; after each of 64 steps the J
//   data and the J accumulators rotate by one lane (ds_bpermute: the LDS crossbar, not the
//   VALU), so every lane meets every J particle once and the accumulators return home.
//   J == I (diagonal tile) is evaluated one-sided, so each ordered pair is counted once.
//   Outputs go to per-(row chunk) and per-(row) slabs that reduce_sym_kernel adds in a fixed
//   order: no atomics, run-to-run bit-identical.
#include "nb_internal.h"

namespace {

template <typename V>
__device__ __forceinline__ V rot1(V v, int src_lane_addr);

template <>
__device__ __forceinline__ double rot1<double>(double v, int addr)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_bpermute(addr, (int)(b & 0xffffffffll));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, (int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// s = q^(-3/2) without any mass factor (see inv_r3_f64 in nb_force.hip for the derivation).
// c15 / c1875 hold 1.5 and 1.875 in registers chosen by the caller: as literals the compiler
// re-materialises 1.5 with two v_mov per pair (v_fmac needs it in the destination).
__device__ __forceinline__ double inv_r3_sym(double q, double c15, double c1875)
{
    const double y0 = __builtin_amdgcn_rsq(q);
    const double y02 = y0 * y0;
    const double e = __builtin_fma(-q, y02, 1.0);
    const double v = y0 * y02;
    const double c = __builtin_fma(e, c1875, c15);
    const double ce = c * e;
    return __builtin_fma(v, ce, v);
}

// One tile-vs-tile sweep: 64 steps, R*R pairs per lane per step, J data rotating by one lane.
// DIAG: J is the target tile itself -> one-sided (each ordered pair once, mirror images dropped).
template <int D, int R, bool DIAG>
__device__ __forceinline__ void sweep(const double (&xi)[R][D], const double (&gi)[R], double (&ai)[R][D],
                                      double (&xj)[R][D], double (&gj)[R], double (&aj)[R][D], double eps2,
                                      int rot_addr)
{
    double c15 = 1.5, c1875 = 1.875;
    asm volatile("" : "+v"(c15), "+s"(c1875));     // opaque: keep them in a VGPR / SGPR pair
#pragma unroll 1
    for (int s = 0; s < 64; ++s) {
#pragma unroll
        for (int ri = 0; ri < R; ++ri) {
#pragma unroll
            for (int rj = 0; rj < R; ++rj) {
                double d[D];
#pragma unroll
                for (int k = 0; k < D; ++k) d[k] = xj[rj][k] - xi[ri][k];
                double q = __builtin_fma(d[D - 1], d[D - 1], eps2);
#pragma unroll
                for (int k = D - 2; k >= 0; --k) q = __builtin_fma(d[k], d[k], q);
                const double w = inv_r3_sym(q, c15, c1875);
                const double wj = w * gj[rj];
#pragma unroll
                for (int k = 0; k < D; ++k) ai[ri][k] = __builtin_fma(wj, d[k], ai[ri][k]);
                if (!DIAG) {
                    const double wi = w * gi[ri];
#pragma unroll
                    for (int k = 0; k < D; ++k) aj[rj][k] = __builtin_fma(-wi, d[k], aj[rj][k]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int k = 0; k < D; ++k) {
                xj[r][k] = rot1<double>(xj[r][k], rot_addr);
                if (!DIAG) aj[r][k] = rot1<double>(aj[r][k], rot_addr);
            }
            gj[r] = rot1<double>(gj[r], rot_addr);
        }
    }
}

template <int D, int R>
__global__ void __launch_bounds__(NB_BLOCK)
force_sym_f64_kernel(const double *__restrict__ packed,   // [D+1][NP]: x, y, (z), G*m ; padded with m = 0
                     const SymWork *__restrict__ work, double *__restrict__ rowslab,
                     double *__restrict__ colslab, int np, double eps2)
{
    constexpr int B = 64 * R;
    __shared__ double s_ai[NB_BLOCK / 64][R][D][64];

    const SymWork wk = work[blockIdx.x];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rot_addr = ((lane + 1) & 63) << 2;

    double xi[R][D], gi[R], ai[R][D];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int p = wk.tile_i * B + r * 64 + lane;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            xi[r][k] = packed[(size_t)k * np + p];
            ai[r][k] = 0.0;
        }
        gi[r] = packed[(size_t)D * np + p];
    }

    for (int J = wk.jt_begin + wave; J < wk.jt_end; J += NB_BLOCK / 64) {
        double xj[R][D], gj[R], aj[R][D];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int p = J * B + r * 64 + lane;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                xj[r][k] = packed[(size_t)k * np + p];
                aj[r][k] = 0.0;
            }
            gj[r] = packed[(size_t)D * np + p];
        }
        if (J == wk.tile_i) {                   // wave-uniform
            sweep<D, R, true>(xi, gi, ai, xj, gj, aj, eps2, rot_addr);
        } else {
            sweep<D, R, false>(xi, gi, ai, xj, gj, aj, eps2, rot_addr);
            // column contributions of row I to the particles of tile J (accumulators are home again)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int p = J * B + r * 64 + lane;
#pragma unroll
                for (int k = 0; k < D; ++k)
                    colslab[((size_t)wk.row_ord * D + k) * np + p] = aj[r][k];
            }
        }
    }

    // combine the four waves' row sums in a fixed order and write one slab slot per workgroup
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < D; ++k) s_ai[wave][r][k][lane] = ai[r][k];
    __syncthreads();
    for (int idx = threadIdx.x; idx < R * D * 64; idx += NB_BLOCK) {
        const int l = idx & 63, rk = idx >> 6;
        const int r = rk / D, k = rk % D;
        double v = s_ai[0][r][k][l];
#pragma unroll
        for (int w = 1; w < NB_BLOCK / 64; ++w) v += s_ai[w][r][k][l];
        const int p = wk.tile_i * B + r * 64 + l;
        rowslab[((size_t)wk.slot * D + k) * np + p] = v;
    }
}

// x, y, (z), G*m as padded component arrays
template <int D>
__global__ void __launch_bounds__(NB_BLOCK)
pack_kernel(const double *__restrict__ pos, const double *__restrict__ mass, double *__restrict__ packed, int n,
            int np, double G)
{
    const int p = blockIdx.x * NB_BLOCK + threadIdx.x;
    if (p >= np) return;
    const bool real = p < n;
#pragma unroll
    for (int k = 0; k < D; ++k) packed[(size_t)k * np + p] = real ? pos[(size_t)p * D + k] : 0.0;
    packed[(size_t)D * np + p] = real ? G * mass[p] : 0.0;
}

// acc[p] = sum of the row slots of p's tile + sum over rows I < tile(p) of the column slabs,
// always in the same order.  Optionally fuses the closing half kick (simulation.py:141).
template <int D>
__global__ void __launch_bounds__(NB_BLOCK)
reduce_sym_kernel(const double *__restrict__ rowslab, const double *__restrict__ colslab,
                  const int *__restrict__ row_slot0, const int *__restrict__ row_nslots,
                  const int *__restrict__ row_ord, int tile_b, int n, int np, double *__restrict__ acc,
                  double *__restrict__ vel, double half_dt, int do_kick)
{
    const int p = blockIdx.x * NB_BLOCK + threadIdx.x;
    if (p >= n) return;
    const int J = p / tile_b;
    double s[D];
#pragma unroll
    for (int k = 0; k < D; ++k) s[k] = 0.0;
    const int s0 = row_slot0[J], ns = row_nslots[J];
    for (int c = 0; c < ns; ++c)
#pragma unroll
        for (int k = 0; k < D; ++k) s[k] += rowslab[((size_t)(s0 + c) * D + k) * np + p];
#pragma unroll 4
    for (int I = 0; I < J; ++I) {
        const int ord = row_ord[I];
        if (ord >= 0) {
#pragma unroll
            for (int k = 0; k < D; ++k) s[k] += colslab[((size_t)ord * D + k) * np + p];
        }
    }
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const size_t idx = (size_t)p * D + k;
        acc[idx] = s[k];
        if (do_kick) vel[idx] = __dadd_rn(vel[idx], __dmul_rn(s[k], half_dt));
    }
}

}  // namespace

hipError_t nb_launch_pack_f64(const double *pos, const double *mass, double *packed, int n, int np, int dim,
                              double G, hipStream_t st)
{
    const int grid = (np + NB_BLOCK - 1) / NB_BLOCK;
    if (dim == 2) hipLaunchKernelGGL((pack_kernel<2>), dim3(grid), dim3(NB_BLOCK), 0, st, pos, mass, packed, n, np, G);
    else if (dim == 3) hipLaunchKernelGGL((pack_kernel<3>), dim3(grid), dim3(NB_BLOCK), 0, st, pos, mass, packed, n, np, G);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t nb_launch_force_sym_f64(const double *packed, const SymWork *work, int nwork, double *rowslab,
                                   double *colslab, int np, int dim, int r, double eps2, hipStream_t st)
{
#define NB_SYM(DD, RR) \
    hipLaunchKernelGGL((force_sym_f64_kernel<DD, RR>), dim3(nwork), dim3(NB_BLOCK), 0, st, packed, work, rowslab, colslab, np, eps2)
    if (dim == 2 && r == 1) NB_SYM(2, 1);
    else if (dim == 2 && r == 2) NB_SYM(2, 2);
    else if (dim == 2 && r == 4) NB_SYM(2, 4);
    else if (dim == 3 && r == 1) NB_SYM(3, 1);
    else if (dim == 3 && r == 2) NB_SYM(3, 2);
    else return hipErrorInvalidValue;
#undef NB_SYM
    return hipGetLastError();
}

hipError_t nb_launch_reduce_sym_f64(const double *rowslab, const double *colslab, const int *row_slot0,
                                    const int *row_nslots, const int *row_ord, int tile_b, int n, int np, int dim,
                                    double *acc, double *vel, double half_dt, int do_kick, hipStream_t st)
{
    const int grid = (n + NB_BLOCK - 1) / NB_BLOCK;
    if (dim == 2)
        hipLaunchKernelGGL((reduce_sym_kernel<2>), dim3(grid), dim3(NB_BLOCK), 0, st, rowslab, colslab, row_slot0,
                           row_nslots, row_ord, tile_b, n, np, acc, vel, half_dt, do_kick);
    else if (dim == 3)
        hipLaunchKernelGGL((reduce_sym_kernel<3>), dim3(grid), dim3(NB_BLOCK), 0, st, rowslab, colslab, row_slot0,
                           row_nslots, row_ord, tile_b, n, np, acc, vel, half_dt, do_kick);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
