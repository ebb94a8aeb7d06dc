// nb_force.hip -- all-pairs softened-gravity kernels for gfx950 (MI355X, CDNA4).
//
// Replaces GalaxySimulation._compute_accelerations (reference simulation.py:74-118) and the
// per-pair part of quantize_distance_squared (quantization.py:21-71).
//
// Decomposition (DESIGN.md "force kernel"):
//   grid  = (target tiles) x (source chunks); block = 256 threads = 4 wavefronts of 64.
//   Each thread keeps R targets in registers; the block walks its source chunk in tiles of
//   NB_TJ particles staged in LDS with one coalesced load per thread; the inner loop reads the
//   tile with wave-uniform (broadcast, conflict-free) ds_reads.
//   Every (tile, chunk) block writes an fp64 partial sum per target to its own slab;
//   reduce_kernel (nb_misc.hip) adds the slabs in fixed order -> run-to-run bit reproducible,
//   no atomics.
//   The path is VALU-bound (arithmetic intensity ~0.135*N flop/B), so the
//   inner loop is written for minimum instruction count: one v_rsq + a fused second-order
//   correction instead of sqrt/div/pow, no self-interaction mask (d == 0 kills the diagonal
//   term exactly as the reference's (1 - eye) multiply does, NaN cases included).
#include "nb_device.h"

#include <hip/hip_fp16.h>

#include <cstdlib>

namespace {

using namespace nbdev;

// ------------------------------------------------------------------------------------------
// q^(-3/2) kernels
// ------------------------------------------------------------------------------------------

// fp64:  returns gm * q^(-3/2).  y0 = v_rsq_f64(q) (about 2^-26 relative), then with
// e = 1 - q*y0^2 :  q^(-3/2) = y0^3 * (1-e)^(-3/2) = y0^3 * (1 + e*(3/2 + 15/8 e) + O(e^3)).
// The neglected term is < 2^-70; the result carries ~1.5 ulp of rounding, the same order as the
// reference's pow -> reciprocal -> *G -> *m chain (simulation.py:97-105).
__device__ __forceinline__ double inv_r3_f64(double q, double gm)
{
    const double y0 = __builtin_amdgcn_rsq(q);
    const double y02 = y0 * y0;
    const double e = __builtin_fma(-q, y02, 1.0);
    const double u = y0 * gm;
    const double v = u * y02;
    const double c = __builtin_fma(e, 1.875, 1.5);
    const double ce = c * e;
    return __builtin_fma(v, ce, v);
}

// fp32:  q^(-3/2) from v_rsq_f32 (1 ulp) plus one Newton step, then cubed.
__device__ __forceinline__ float inv_r3_f32(float q)
{
    const float y0 = __builtin_amdgcn_rsqf(q);
    const float t = q * y0;
    const float e = __builtin_fmaf(-t, y0, 1.0f);
    const float y = __builtin_fmaf(0.5f * y0, e, y0);
    return (y * y) * y;
}

// PA: arithmetic type of diff / r2.  NB_F32 normally; NB_F16 / NB_BF16 for the first evaluation on
// half-typed state tensors (omega_point_test.py:722-733): every op of simulation.py:83-86 rounds to
// the half type, the D-term sum accumulates in float and rounds once (torch opmath), eps2 enters
// as a half scalar.
template <int PA> __device__ __forceinline__ float round_pa(float x)
{
    if (PA == NB_F16) return round_f16(x);
    if (PA == NB_BF16) return round_bf16(x);
    return x;
}
template <int D, int PA>
__device__ __forceinline__ float r2_half_state(const float *xi, const float *xj, float eps2_pa, float *d)
{
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        d[k] = round_pa<PA>(__fsub_rn(xj[k], xi[k]));
        const float sq = round_pa<PA>(__fmul_rn(d[k], d[k]));
        s = (k == 0) ? sq : __fadd_rn(s, sq);
    }
    s = round_pa<PA>(s);
    return round_pa<PA>(__fadd_rn(s, eps2_pa));
}

// ------------------------------------------------------------------------------------------
// fp64 state (FLOAT64 mode).  PA_F32: diff and r2 in fp32 (first evaluation on fp32-typed
// positions), everything after the hook in fp64.
// ------------------------------------------------------------------------------------------
// QHOOK != HOOK_NONE: fp64 positions under FLOAT32 / BFLOAT16 / FLOAT16 mode (omega_point_test.py
// :722-733 builds such sims): diff and r2 in fp64 in the reference's op order, the hook casts r2
// to fp32 (and through the half type), q^1.5 and G/. stay fp32, the product with the fp64 mass and
// everything after it is fp64 again (torch promotion).
// PAIR: -1 = fp64 pair arithmetic; NB_F32 / NB_F16 / NB_BF16 = the dtype the positions are typed as
// (first evaluation before the state has been promoted to fp64).
template <int D, int R, int PAIR, int QHOOK = -1>
__global__ void __launch_bounds__(NB_BLOCK)
force_f64_kernel(const double *__restrict__ pos, const double *__restrict__ mass,
                 double *__restrict__ partial, ForceGeom g, double G, double eps2, float eps2_f)
{
    __shared__ double sj[D + 1][NB_TJ];

    const int tid = threadIdx.x;
    const int ibase = blockIdx.x * (NB_BLOCK * R);

    double xi[R][D];
    double acc[R][D];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int i = ibase + r * NB_BLOCK + tid;
        i = i < g.n ? i : g.n - 1;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            xi[r][k] = pos[(size_t)i * D + k];
            acc[r][k] = 0.0;
        }
    }

    const int j_lo = g.j_begin + blockIdx.y * g.chunk_len;
    const int j_hi = min(j_lo + g.chunk_len, g.j_end);

    for (int jt = j_lo; jt < j_hi; jt += NB_TJ) {
        {
            int j = jt + tid;
            j = j < j_hi ? j : j_hi - 1;
#pragma unroll
            for (int k = 0; k < D; ++k) sj[k][tid] = pos[(size_t)j * D + k];
            sj[D][tid] = (QHOOK >= 0) ? mass[j] : G * mass[j];
        }
        __syncthreads();
        const int cnt = min(NB_TJ, j_hi - jt);
#pragma unroll 4
        for (int jj = 0; jj < cnt; ++jj) {
            double xj[D];
#pragma unroll
            for (int k = 0; k < D; ++k) xj[k] = sj[k][jj];
            const double gm = sj[D][jj];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                double d[D];
                double q;
                if (QHOOK >= 0) {
#pragma unroll
                    for (int k = 0; k < D; ++k) d[k] = __dsub_rn(xj[k], xi[r][k]);
                    double s2 = __dadd_rn(__dmul_rn(d[0], d[0]), __dmul_rn(d[1], d[1]));
                    if (D == 3) s2 = __dadd_rn(s2, __dmul_rn(d[2], d[2]));
                    float q32 = (float)__dadd_rn(s2, eps2);
                    if (QHOOK == HOOK_BF16) q32 = round_bf16(q32);
                    if (QHOOK == HOOK_F16) q32 = round_f16(q32);
                    float wq = inv_r3_f32(q32) * (float)G;
                    if (QHOOK == HOOK_F16) wq = (q32 == __builtin_inff()) ? 0.0f : wq;
                    const double w = __dmul_rn((double)wq, gm);
#pragma unroll
                    for (int k = 0; k < D; ++k) acc[r][k] = __builtin_fma(w, d[k], acc[r][k]);
                    continue;
                }
                if (PAIR == NB_F32) {
                    float df[D];
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        df[k] = __fsub_rn((float)xj[k], (float)xi[r][k]);
                        d[k] = (double)df[k];
                    }
                    q = (double)r2_f32_exact<D>(df, eps2_f);
                } else if (PAIR == NB_F16 || PAIR == NB_BF16) {
                    float xif[D], xjf[D], df[D];
#pragma unroll
                    for (int k = 0; k < D; ++k) { xif[k] = (float)xi[r][k]; xjf[k] = (float)xj[k]; }
                    q = (double)r2_half_state<D, PAIR>(xif, xjf, eps2_f, df);     // eps2_f already half-rounded
#pragma unroll
                    for (int k = 0; k < D; ++k) d[k] = (double)df[k];
                } else {
#pragma unroll
                    for (int k = 0; k < D; ++k) d[k] = xj[k] - xi[r][k];
                    q = __builtin_fma(d[D - 1], d[D - 1], eps2);
#pragma unroll
                    for (int k = D - 2; k >= 0; --k) q = __builtin_fma(d[k], d[k], q);
                }
                const double w = inv_r3_f64(q, gm);
#pragma unroll
                for (int k = 0; k < D; ++k) acc[r][k] = __builtin_fma(w, d[k], acc[r][k]);
            }
        }
        __syncthreads();
    }

    double *out = partial + (size_t)blockIdx.y * g.n * D;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = ibase + r * NB_BLOCK + tid;
        if (i < g.n) {
#pragma unroll
            for (int k = 0; k < D; ++k) out[(size_t)i * D + k] = acc[r][k];
        }
    }
}

// ------------------------------------------------------------------------------------------
// fp32 state (FLOAT32 / BFLOAT16 / FLOAT16 / INT8 / INT4 / CUSTOM modes)
// ------------------------------------------------------------------------------------------
// BINS: the same body with the quant-bin read-out (per-target integer checksums s1 = sum_j k, s2 = sum_j k ((j mod
// 65521) + 1), see nb_force_sym_kernel.h BinDbg), added to bin_out = {s1[n], s2[n], {0, table pairs}} with atomics.
template <int D, int R, int HOOK, int PA = NB_F32, bool BINS = false>
__global__ void __launch_bounds__(NB_BLOCK)
force_f32_kernel(const float *__restrict__ pos, const float *__restrict__ mass,
                 double *__restrict__ partial, ForceGeom g, float G, float eps2,
                 const GridTables *__restrict__ tab, int lp, unsigned long long *__restrict__ bin_out)
{
    __shared__ float sj[D + 1][NB_TJ];
    extern __shared__ float s_tables[];        // grid hook: thr[lp + 1], lut[lp + 1] (nb_lut_lds_bytes)
    float *s_thr = s_tables, *s_lut = s_tables + lp + 1;

    const int tid = threadIdx.x;
    const int ibase = blockIdx.x * (NB_BLOCK * R);
    bool degenerate = false, use_est = false;
    float est_a = 0.0f, est_b = 0.0f;
    int est_kmax = 0;

    if (HOOK == HOOK_GRID) {
        for (int k = tid; k < lp; k += NB_BLOCK) {
            s_thr[k] = (k < tab->levels) ? tab->thr[k] : __builtin_inff();
            s_lut[k] = (k < tab->levels) ? tab->lut[k] : 0.0f;
        }
        degenerate = tab->degenerate != 0;
        use_est = tab->use_est != 0;
        est_a = tab->est_a;
        est_b = tab->est_b;
        est_kmax = tab->levels - 2;
    }

    float xi[R][D];
    double acc[R][D];
    long long b1[BINS ? R : 1], b2[BINS ? R : 1], bcount = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int i = ibase + r * NB_BLOCK + tid;
        i = i < g.n ? i : g.n - 1;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            xi[r][k] = pos[(size_t)i * D + k];
            acc[r][k] = 0.0;
        }
        if constexpr (BINS) b1[r] = b2[r] = 0;
    }

    const int j_lo = g.j_begin + blockIdx.y * g.chunk_len;
    const int j_hi = min(j_lo + g.chunk_len, g.j_end);

    for (int jt = j_lo; jt < j_hi; jt += NB_TJ) {
        {
            int j = jt + tid;
            j = j < j_hi ? j : j_hi - 1;
#pragma unroll
            for (int k = 0; k < D; ++k) sj[k][tid] = pos[(size_t)j * D + k];
            sj[D][tid] = mass[j];
        }
        __syncthreads();
        const int cnt = min(NB_TJ, j_hi - jt);
        // fp32 running sums over one tile only; the tile sums are folded into fp64 accumulators
        // so the error of an N-term fp32 sum stays at the level of torch's cascade summation.
        float tacc[R][D];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int k = 0; k < D; ++k) tacc[r][k] = 0.0f;

#pragma unroll 4
        for (int jj = 0; jj < cnt; ++jj) {
            float xj[D];
#pragma unroll
            for (int k = 0; k < D; ++k) xj[k] = sj[k][jj];
            const float mj = sj[D][jj];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float d[D];
                float r2;
                if (PA == NB_F32) {
#pragma unroll
                    for (int k = 0; k < D; ++k) d[k] = __fsub_rn(xj[k], xi[r][k]);
                    r2 = r2_f32_exact<D>(d, eps2);
                } else {
                    r2 = r2_half_state<D, PA>(xi[r], xj, eps2, d);
                }
                float wq;   // (1 / q^1.5) * G   (simulation.py:97-101)
                if (HOOK == HOOK_GRID) {
                    if (!degenerate) {
                        // floor estimate + one threshold compare (nb_device.h) when the tables allow it,
                        // else the 8-step binary search: both give the exact bin
                        const int kb = use_est ? grid_bin_floor_estimate(s_thr, r2, est_a, est_b, est_kmax)
                                               : grid_bin_lookup(s_thr, r2, lp);
                        wq = s_lut[kb];
                        if constexpr (BINS) {
                            b1[r] += kb;
                            b2[r] += (long long)kb * ((jt + jj) % 65521 + 1);
                            ++bcount;
                        }
                    } else {
                        const float q = (r2 < 0.01f) ? 0.01f : r2;     // clamp keeps NaN
                        wq = inv_r3_f32(q) * G;
                    }
                } else {
                    float q = r2;
                    if (HOOK == HOOK_BF16) q = round_bf16(r2);
                    if (HOOK == HOOK_F16) q = round_f16(r2);
                    wq = inv_r3_f32(q) * G;
                    // fp16 overflow: q = +inf -> pow = inf -> 1/inf = 0 upstream (rsq-based form gives NaN)
                    if (HOOK == HOOK_F16) wq = (q == __builtin_inff()) ? 0.0f : wq;
                }
                const float w = wq * mj;                                // simulation.py:105
#pragma unroll
                for (int k = 0; k < D; ++k) tacc[r][k] = __builtin_fmaf(w, d[k], tacc[r][k]);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int k = 0; k < D; ++k) acc[r][k] += (double)tacc[r][k];
        __syncthreads();
    }

    double *out = partial + (size_t)blockIdx.y * g.n * D;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = ibase + r * NB_BLOCK + tid;
        if (i < g.n) {
#pragma unroll
            for (int k = 0; k < D; ++k) out[(size_t)i * D + k] = acc[r][k];
            if constexpr (BINS) {
                atomicAdd(&bin_out[i], (unsigned long long)b1[r]);
                atomicAdd(&bin_out[(size_t)g.n + i], (unsigned long long)b2[r]);
            }
        }
    }
    if constexpr (BINS) {
        // clamped duplicate targets (i >= n) counted pairs too: report only this thread's live ones
        int live = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) live += (ibase + r * NB_BLOCK + tid < g.n) ? 1 : 0;
        atomicAdd(&bin_out[2 * (size_t)g.n + 1], (unsigned long long)(bcount / R * live));
    }
}

// ------------------------------------------------------------------------------------------
// K2: global max of the fp32 r2 over (all targets) x (this rank's sources).
// quantization.py:113 needs log(max r2); the minimum is analytic (diagonal).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned int r2_order_bits(float v)
{
    // r2 >= 0 always, so the IEEE bit pattern orders like an unsigned int; any NaN maps above +inf
    return (v != v) ? 0x7fc00000u : __float_as_uint(v);
}

// all-pairs max of the fp32 r2 over this workgroup's targets and source chunk; thread 0 adds it to tab->r2max_bits
// SPLIT > 1 (small systems, R = 1): SPLIT neighbouring lanes share a target and deal a tile's sources among them --
// a thread then walks NB_TJ / SPLIT sources instead of NB_TJ, and there are SPLIT times as many workgroups.
template <int D, int R, int SPLIT = 1>
__device__ __forceinline__ void r2max_block(const float *__restrict__ pos, const ForceGeom &g, float eps2,
                                            GridTables *__restrict__ tab)
{
    static_assert(SPLIT == 1 || R == 1, "lane splitting is for one target per thread");
    __shared__ float sj[D][NB_TJ];
    __shared__ unsigned int s_red[NB_BLOCK / 64];
    const int tid = threadIdx.x;
    const int ibase = blockIdx.x * (NB_BLOCK * R / SPLIT);
    const int sub = tid % SPLIT;

    float xi[R][D];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int i = ibase + r * NB_BLOCK + tid / SPLIT;
        i = i < g.n ? i : g.n - 1;
#pragma unroll
        for (int k = 0; k < D; ++k) xi[r][k] = pos[(size_t)i * D + k];
    }
    const int j_lo = g.j_begin + blockIdx.y * g.chunk_len;
    const int j_hi = min(j_lo + g.chunk_len, g.j_end);

    unsigned int best = 0;
    for (int jt = j_lo; jt < j_hi; jt += NB_TJ) {
        // r2 is symmetric: source tiles entirely below this block's targets are covered by the
        // mirrored pairs of another block (block-uniform test, so the barriers stay matched)
        if (jt + NB_TJ <= ibase) continue;
        {
            int j = jt + tid;
            j = j < j_hi ? j : j_hi - 1;
#pragma unroll
            for (int k = 0; k < D; ++k) sj[k][tid] = pos[(size_t)j * D + k];
        }
        __syncthreads();
        const int cnt = min(NB_TJ, j_hi - jt);
#pragma unroll 8
        for (int jj = sub; jj < cnt; jj += SPLIT) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float d[D];
#pragma unroll
                for (int k = 0; k < D; ++k) d[k] = __fsub_rn(sj[k][jj], xi[r][k]);
                best = max(best, r2_order_bits(r2_f32_exact<D>(d, eps2)));
            }
        }
        __syncthreads();
    }
    // wavefront (64-lane) shuffle reduction, then one atomic per block
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) best = max(best, (unsigned int)__shfl_xor((int)best, off, 64));
    if ((tid & 63) == 0) s_red[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        unsigned int b = s_red[0];
#pragma unroll
        for (int w = 1; w < NB_BLOCK / 64; ++w) b = max(b, s_red[w]);
        atomicMax(&tab->r2max_bits, b);
    }
}

template <int D, int R>
__global__ void __launch_bounds__(NB_BLOCK)
r2max_kernel(const float *__restrict__ pos, ForceGeom g, float eps2, GridTables *__restrict__ tab)
{
    r2max_block<D, R>(pos, g, eps2, tab);
}

// ------------------------------------------------------------------------------------------
// K2 with pruning.  The pair with the largest r2 lies on the outside of the particle cloud, so a
// brute-force scan of all N^2 pairs is wasteful.  Exactness argument (all quantities fp32, margins
// 1e-5 >> the ~4e-7 rounding of r2 and rho):
//   * c = bounding-box centre, rho_i = |x_i - c|; for any pair dist(a,b) <= rho_a + rho_b <= rho_a + rho_max;
//   * LB = r2 of a real pair found by two farthest-point hops (far -> g -> h), so max r2 >= LB;
//   * hence both members of the maximal pair satisfy rho >= sqrt(LB - eps2) - rho_max (up to margins).
// Candidates passing that test are compacted and scanned exhaustively with the exact r2 formula.
// For a disk galaxy this keeps a few percent of the particles (~0.1 % of the pairs).
// ------------------------------------------------------------------------------------------
// Stage 1 of the bounding box: per-block min/max of every coordinate (+ a NaN flag) as ordered
// integer keys folded with atomics; the rho kernel decodes them (two launches, a few us each).
__device__ __forceinline__ unsigned int order_key(float v)
{
    const unsigned int u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);      // monotone in v for all finite values
}
__device__ __forceinline__ float order_unkey(unsigned int k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

template <int D>
__global__ void __launch_bounds__(NB_BLOCK)
prune_bbox_kernel(const float *__restrict__ pos, int n, PruneState *__restrict__ ps)
{
    __shared__ unsigned int s_mn[D][NB_BLOCK / 64], s_mx[D][NB_BLOCK / 64];
    __shared__ int s_nan;
    if (threadIdx.x == 0) s_nan = 0;
    __syncthreads();
    unsigned int mn[D], mx[D];
#pragma unroll
    for (int k = 0; k < D; ++k) { mn[k] = 0xffffffffu; mx[k] = 0u; }
    bool bad = false;
    for (int i = blockIdx.x * NB_BLOCK + threadIdx.x; i < n; i += gridDim.x * NB_BLOCK)
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const float v = pos[(size_t)i * D + k];
            bad |= (v != v);
            const unsigned int key = order_key(v);
            mn[k] = min(mn[k], key);
            mx[k] = max(mx[k], key);
        }
    if (bad) s_nan = 1;
#pragma unroll
    for (int k = 0; k < D; ++k) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            mn[k] = min(mn[k], (unsigned int)__shfl_xor((int)mn[k], off, 64));
            mx[k] = max(mx[k], (unsigned int)__shfl_xor((int)mx[k], off, 64));
        }
        if ((threadIdx.x & 63) == 0) { s_mn[k][threadIdx.x >> 6] = mn[k]; s_mx[k][threadIdx.x >> 6] = mx[k]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            unsigned int a = s_mn[k][0], b = s_mx[k][0];
            for (int w = 1; w < NB_BLOCK / 64; ++w) { a = min(a, s_mn[k][w]); b = max(b, s_mx[k][w]); }
            atomicMin(&ps->box_min[k], a);
            atomicMax(&ps->box_max[k], b);
        }
        if (s_nan) atomicOr(&ps->nan_flag, 1);
    }
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// one atomic per block instead of one per wave
__device__ __forceinline__ void block_atomic_max_u64(unsigned long long key, unsigned long long *dst)
{
    __shared__ unsigned long long s_key[NB_BLOCK / 64];
    key = wave_max_u64(key);
    if ((threadIdx.x & 63) == 0) s_key[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long m = s_key[0];
#pragma unroll
        for (int w = 1; w < NB_BLOCK / 64; ++w) m = s_key[w] > m ? s_key[w] : m;
        if (m) atomicMax(dst, m);
    }
}

template <int D>
__global__ void __launch_bounds__(NB_BLOCK)
prune_rho_kernel(const float *__restrict__ pos, int n, float *__restrict__ rho, PruneState *__restrict__ ps)
{
    const int i = blockIdx.x * NB_BLOCK + threadIdx.x;
    unsigned long long key = 0ull;
    if (i == 0) {            // the centre stays put for the tracked searches that follow (PruneState::c)
#pragma unroll
        for (int k = 0; k < D; ++k) ps->c[k] = 0.5f * order_unkey(ps->box_min[k]) + 0.5f * order_unkey(ps->box_max[k]);
    }
    if (i < n) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const float c = 0.5f * order_unkey(ps->box_min[k]) + 0.5f * order_unkey(ps->box_max[k]);
            const float d = pos[(size_t)i * D + k] - c;
            s += d * d;
        }
        const float r = sqrtf(s);
        rho[i] = r;
        key = ((unsigned long long)__float_as_uint(r) << 32) | (unsigned int)i;
    }
    block_atomic_max_u64(key, &ps->far);
}

// farthest partner (by the exact fp32 r2) of the particle whose index is stored in *src_key
template <int D>
__global__ void __launch_bounds__(NB_BLOCK)
prune_hop_kernel(const float *__restrict__ pos, int n, float eps2, const unsigned long long *__restrict__ src_key,
                 unsigned long long *__restrict__ dst_key)
{
    const int f = (int)(*src_key & 0xffffffffull);
    const int j = blockIdx.x * NB_BLOCK + threadIdx.x;
    unsigned long long key = 0ull;
    if (j < n) {
        float d[D];
#pragma unroll
        for (int k = 0; k < D; ++k) d[k] = __fsub_rn(pos[(size_t)j * D + k], pos[(size_t)f * D + k]);
        key = ((unsigned long long)r2_order_bits(r2_f32_exact<D>(d, eps2)) << 32) | (unsigned int)j;
    }
    block_atomic_max_u64(key, dst_key);
}

template <int D>
__global__ void __launch_bounds__(NB_BLOCK)
prune_compact_kernel(const float *__restrict__ pos, const float *__restrict__ rho, int n, float eps2,
                     float *__restrict__ cand, PruneState *__restrict__ ps)
{
    const int i = blockIdx.x * NB_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float lb = __uint_as_float((unsigned int)(ps->lb[1] >> 32));
    const float rho_max = __uint_as_float((unsigned int)(ps->far >> 32));
    const float need = sqrtf(fmaxf(lb - eps2, 0.0f)) * (1.0f - 1e-5f) - rho_max * (1.0f + 1e-5f);
    if (rho[i] * (1.0f + 1e-5f) >= need) {
        const int slot = atomicAdd(&ps->count, 1);
#pragma unroll
        for (int k = 0; k < D; ++k) cand[(size_t)slot * D + k] = pos[(size_t)i * D + k];
    }
}

// exhaustive exact max over the compacted candidates (count read on the device)
template <int D>
__global__ void __launch_bounds__(NB_BLOCK)
prune_scan_kernel(const float *__restrict__ cand, float eps2, const PruneState *__restrict__ ps,
                  GridTables *__restrict__ tab)
{
    __shared__ float sj[D][NB_TJ];
    __shared__ unsigned int s_red[NB_BLOCK / 64];
    const int m = ps->count;
    const int tid = threadIdx.x;
    const int ibase = blockIdx.x * NB_BLOCK;
    if (ps->nan_flag) {                          // a NaN coordinate: torch's max() would be NaN
        if (blockIdx.x == 0 && tid == 0) atomicMax(&tab->r2max_bits, 0x7fc00000u);
        return;
    }
    if (ibase >= m) return;                      // block-uniform
    const int i = min(ibase + tid, m - 1);
    float xi[D];
#pragma unroll
    for (int k = 0; k < D; ++k) xi[k] = cand[(size_t)i * D + k];
    unsigned int best = 0;
    for (int jt = ibase; jt < m; jt += NB_TJ) {  // r2 is symmetric: tiles below the block are mirrored elsewhere
        {
            const int j = min(jt + tid, m - 1);
#pragma unroll
            for (int k = 0; k < D; ++k) sj[k][tid] = cand[(size_t)j * D + k];
        }
        __syncthreads();
        const int cnt = min(NB_TJ, m - jt);
#pragma unroll 8
        for (int jj = 0; jj < cnt; ++jj) {
            float d[D];
#pragma unroll
            for (int k = 0; k < D; ++k) d[k] = __fsub_rn(sj[k][jj], xi[k]);
            best = max(best, r2_order_bits(r2_f32_exact<D>(d, eps2)));
        }
        __syncthreads();
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) best = max(best, (unsigned int)__shfl_xor((int)best, off, 64));
    if ((tid & 63) == 0) s_red[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        unsigned int b = s_red[0];
#pragma unroll
        for (int w = 1; w < NB_BLOCK / 64; ++w) b = max(b, s_red[w]);
        atomicMax(&tab->r2max_bits, b);
    }
}

// ------------------------------------------------------------------------------------------
// Grid tables: exact scalar form of _grid_quantize_safe (quantization.py:91-127) evaluated on
// the device once per force evaluation.  Every stage (clamp, log, -lmin, /range, *(L-1), round)
// is monotone in r2, so the bin index is a step function of the fp32 r2: thread k finds the
// smallest fp32 value whose bin is >= k by bisection over bit patterns.  Bit-identical to
// evaluating the formula per pair, with no log/div/exp in the pair loop.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float logf_cr(float t) { return (float)log((double)t); }

__device__ __forceinline__ float grid_bin_exact(float t, float min_val, float lmin, float range, float lm1)
{
    const float ts = (t < min_val) ? min_val : t;
    const float lt = logf_cr(ts);
    const float nrm = __fmul_rn(__fdiv_rn(__fsub_rn(lt, lmin), range), lm1);
    return rintf(nrm);   // half-to-even like torch.round
}

// The same bin for a t within ~1e-5 (relative) of a point c whose fp64 logarithm is known (c = exp(c_log): the result
// of exp carries the same 1e-16 the library log of t would): log t = c_log + log1p(u), u = (t - c) / c, three terms
// of the series (u^4 / 4 < 1e-21).  A dozen fp64 operations instead of a library log (~1400 cycles for a lone wave);
// the threshold search spends all its evaluations inside such a bracket.
__device__ __forceinline__ float grid_bin_near(float t, double c_log, double c_inv, float lmin, float range, float lm1)
{
    const double u = __builtin_fma((double)t, c_inv, -1.0);
    const double l1p = u * (1.0 + u * (-0.5 + u * (1.0 / 3.0)));
    const float lt = (float)(c_log + l1p);
    const float nrm = __fmul_rn(__fdiv_rn(__fsub_rn(lt, lmin), range), lm1);
    return rintf(nrm);
}

// One thread per level, NB_LUT_MIN threads per block; the block that arrives last (all others have read
// tab->r2max_bits and written their entries) writes the scalars and resets the scratch for the next evaluation.
// FUSED: called by the last workgroup of r2max_tables_kernel (single-block tables, levels <= NB_LUT_MIN; the
// caller passes the finished maximum).  Otherwise the body of grid_tables_kernel (one workgroup per NB_LUT_MIN levels).
template <bool FUSED>
__device__ __forceinline__ void grid_tables_body(GridTables *__restrict__ tab, int levels, float G, float eps2, float min_val,
                                                 PruneState *__restrict__ ps, int allow_fast, unsigned int r2max_bits,
                                                 const double *__restrict__ bounds = nullptr)
{
    if (bounds) {                  // tensor-level hook: minimum / maximum of the tensor, found on the device
        eps2 = (float)bounds[0];
        r2max_bits = __float_as_uint((float)bounds[1]);
    }
    __shared__ int s_last;
    __shared__ float s_thr[NB_LUT_MIN], s_lut[NB_LUT_MIN], s_red[NB_LUT_MIN], s_par[4];
    __shared__ double s_lg2[2];          // log2 of the first / last factor, taken by their own threads (off the serial tail)
    const int k = (FUSED ? 0 : blockIdx.x) * NB_LUT_MIN + threadIdx.x;
    const int tables_blocks = FUSED ? 1 : gridDim.x;
    const float r2max = __uint_as_float(r2max_bits);
    const float tmin = (eps2 < min_val) ? min_val : eps2;        // diagonal entries: r2 == eps2
    const float tmax = (r2max < min_val) ? min_val : r2max;
    // the two correctly rounded logarithms head every thread's dependency chain: even lanes take one, odd lanes the
    // other, and neighbours swap (all 256 threads are active here)
    const bool odd = (threadIdx.x & 1) != 0;
    const float lg_mine = logf_cr(odd ? tmax : tmin);
    const float lg_other = __shfl_xor(lg_mine, 1, 64);
    const float lmin = odd ? lg_other : lg_mine;
    const float lmax = odd ? lg_mine : lg_other;
    const float range = __fsub_rn(lmax, lmin);
    const float lm1 = (float)(levels - 1);
    const bool degenerate = range < 1e-10f || levels > NB_MAX_LUT;

    // Small single-block tables (INT4, CUSTOM 64): the value / factor of a level and its threshold are independent
    // chains of comparable length, so they go to different lanes (threads [0, L) and [128, 128 + L)).
    const bool two_roles = tables_blocks == 1 && levels <= NB_LUT_MIN / 2;
    const int tid_l = threadIdx.x;
    const int ka = two_roles ? (tid_l < levels ? tid_l : -1) : (k < levels ? k : -1);
    const int kb = two_roles ? ((tid_l >= NB_LUT_MIN / 2 && tid_l - NB_LUT_MIN / 2 < levels) ? tid_l - NB_LUT_MIN / 2 : -1) : ka;
    const int base_l = two_roles ? 0 : (k - tid_l);          // first level of this workgroup
    if (ka >= 0) {
        // value of bin k (quantization.py:121-127) and its force factor (simulation.py:97-101)
        float v = __fdiv_rn((float)ka, lm1);
        v = __fmul_rn(v, range);
        v = __fadd_rn(v, lmin);
        float q = (float)exp((double)v);
        q = (q < min_val) ? min_val : q;
        // q^1.5 correctly rounded to fp32: q * sqrt(q) in fp64 is good to 2 ulp of fp64, 2^-29 of an fp32 spacing
        const double qd = (double)q;
        const float p = (float)(qd * __dsqrt_rn(qd));
        const float w = __fmul_rn(__fdiv_rn(1.0f, p), G);
        tab->qval[ka] = q;
        tab->lut[ka] = w;
        s_lut[ka - base_l] = w;
        if (ka == 0) s_lg2[0] = log2((double)w);
        if (ka == levels - 1) s_lg2[1] = log2((double)w);
    }
    if (kb >= 0) {
        const int k = kb;
        float thr = -__builtin_inff();
        if (k > 0 && !degenerate) {
            if (r2max != r2max) {
                thr = __builtin_nanf("");
            } else {
                unsigned int lo = __float_as_uint(tmin);   // bin(tmin) == 0 < k
                unsigned int hi = __float_as_uint(tmax);   // bin(tmax) == L-1 >= k
                // The edge of bin k sits near exp(lmin + (k - 1/2) range / (L-1)): bracket it within +-4e-6
                // (~70 fp32 values; the fp32 roundings of the reference's formula move the edge by ~1e-6 at most)
                // when both ends check out with the exact formula, so the bisection needs ~7 instead of ~31
                // evaluations; otherwise keep the full interval.  The result is the
                // same either way (bin is monotone in t).
                bool near = false;          // both ends of the bracket hold: every further t is within 4e-6 of `edge`
                double edge = 1.0, edge_log = 0.0, edge_inv = 1.0;
                if (range > 1e-6f) {
                    edge_log = (double)lmin + ((double)k - 0.5) * (double)range / (double)lm1;
                    edge = exp(edge_log);
                    edge_inv = 1.0 / edge;
                    const float a_lo = (float)(edge * (1.0 - 4e-6)), a_hi = (float)(edge * (1.0 + 4e-6));
                    const bool in_lo = a_lo > tmin && a_lo < tmax, in_hi = a_hi > tmin && a_hi < tmax;
                    const bool ok_lo = in_lo && grid_bin_near(a_lo, edge_log, edge_inv, lmin, range, lm1) < (float)k;
                    const bool ok_hi = in_hi && grid_bin_near(a_hi, edge_log, edge_inv, lmin, range, lm1) >= (float)k;
                    if (ok_lo) lo = __float_as_uint(a_lo);
                    if (ok_hi) hi = __float_as_uint(a_hi);
                    near = ok_lo && ok_hi;
                }
                while (hi - lo > 1u) {
                    const unsigned int mid = lo + ((hi - lo) >> 1);
                    const float b = near ? grid_bin_near(__uint_as_float(mid), edge_log, edge_inv, lmin, range, lm1)
                                         : grid_bin_exact(__uint_as_float(mid), min_val, lmin, range, lm1);
                    if (b >= (float)k) hi = mid; else lo = mid;
                }
                thr = __uint_as_float(hi);
            }
        }
        tab->thr[k] = thr;
        s_thr[k - base_l] = thr;
    }
    __syncthreads();               // every thread of this block has read tab->r2max_bits and written its entry
    if (!FUSED && gridDim.x > 1) {       // (single-block tables -- every grid up to 256 levels -- need no arrival count)
        if (threadIdx.x == 0) {
            __threadfence();
            s_last = (atomicAdd(&tab->blocks_done, 1u) == gridDim.x - 1) ? 1 : 0;
        }
        __syncthreads();
        if (!s_last) return;
    }
    const bool fin = threadIdx.x == 0;
    // ---- table-free pair path: parameters + validation (single-block tables only: levels <= NB_LUT_MIN) ------
    const double est_a_d = 0.6931471805599453 * (double)lm1 / (double)range;
    const float est_a = (float)est_a_d, est_b = (float)(-(double)lmin * (double)lm1 / (double)range);
    const bool est_ok = (range >= 1e-10f && est_a_d < 1.0e4);
    const int kc = levels / 2;
    const float est_bc = (float)(-(double)lmin * (double)lm1 / (double)range - (double)kc);
    bool fast_try = allow_fast && est_ok && tables_blocks == 1 && levels >= 2 && r2max == r2max && r2max < 1e30f;
    if (fin) {
        // log2 of the force factor is affine in the bin index: fit it to the table's end points
        const double w0 = (double)s_lut[0], w1 = (double)s_lut[min(levels, NB_LUT_MIN) - 1];
        double c1 = 0.0, cm = 0.0;
        if (fast_try && w0 > 0.0 && w1 > 0.0 && w0 < 1e30 && w1 < 1e30) {
            c1 = (s_lg2[1] - s_lg2[0]) / (double)(levels - 1);
            cm = s_lg2[0] + c1 * (double)kc;            // log2 of the factor of the middle bin
        } else {
            fast_try = false;
        }
        const double tm = rint(cm);
        if (!(fabs(tm) < 100.0) || !(fabs(c1) * (double)levels < 100.0)) fast_try = false;
        s_par[0] = (float)c1;
        s_par[1] = (float)(cm - tm);
        s_par[2] = (float)tm;
        s_last = fast_try ? 2 : 1;
    }
    __syncthreads();
    fast_try = (s_last == 2);
    float my_dev = 0.0f, my_rel = 0.0f;
    if (fast_try && k < levels) {
        // the estimate exactly as the pair loop evaluates it, on both sides of the lower edge of bin k
        if (k > 0) {
            const float r = s_thr[k], rp = __uint_as_float(__float_as_uint(r) - 1u);
            const float edge = (float)(k - kc) - 0.5f;
            const float e1 = __builtin_fmaf(__builtin_amdgcn_logf(r), est_a, est_bc) - edge;
            const float e0 = __builtin_fmaf(__builtin_amdgcn_logf(rp), est_a, est_bc) - edge;
            my_dev = fmaxf(fabsf(e1), fabsf(e0));
            if (!(my_dev == my_dev)) my_dev = 1.0f;
        }
        const float wf = __builtin_amdgcn_exp2f(__builtin_fmaf((float)(k - kc), s_par[0], s_par[1]));
        const float wt = ldexpf(s_lut[k], -(int)s_par[2]);
        my_rel = fabsf(wf - wt) / wt;
        if (!(my_rel == my_rel)) my_rel = 1.0f;
    }
    float maxdev = 0.0f, maxrel = 0.0f;
    if (fast_try) {                 // block-uniform: both maxima by wave shuffles + one exchange through LDS
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            my_dev = fmaxf(my_dev, __shfl_xor(my_dev, off, 64));
            my_rel = fmaxf(my_rel, __shfl_xor(my_rel, off, 64));
        }
        if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6] = my_dev; s_red[8 + (threadIdx.x >> 6)] = my_rel; }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NB_LUT_MIN / 64; ++w) { maxdev = fmaxf(maxdev, s_red[w]); maxrel = fmaxf(maxrel, s_red[8 + w]); }
    }
    if (fin && ps) {
        // the pruned max-r2 search is finished: seed the tracked search of the next evaluation with its far pair
        // (the second hop's pair g -> h is a real pair; the exact maximum is re-derived from the positions then) ...
        ps->pair_i = (int)(ps->lb[0] & 0xffffffffull);
        ps->pair_j = (int)(ps->lb[1] & 0xffffffffull);
        ps->rho_m = __uint_as_float((unsigned int)(ps->far >> 32)) * 1.005f;
        ps->rho_prev = 0.0f;
        ps->seeded = (ps->nan_flag == 0) ? 1 : 0;
        ps->rho_cur = 0u;
        ps->scan_done = 0u;
        ps->best_i = 0ull;
        ps->best_j = 0ull;
        // ... and reset its scratch
        for (int c = 0; c < 3; ++c) { ps->box_min[c] = 0xffffffffu; ps->box_max[c] = 0u; }
        ps->far = 0ull;
        ps->lb[0] = 0ull;
        ps->lb[1] = 0ull;
        ps->count = 0;
        ps->nan_flag = 0;
    }
    if (fin) {
        tab->blocks_done = 0u;
        // sentinel above the last bin: NaN compares false, so a lookup can never step past L-1
        tab->thr[levels] = __builtin_nanf("");
        // fast bin estimate for the pair loop: n ~ log2(r2) * a + b with a = ln2*(L-1)/range.
        // v_log_f32 is good to ~1e-6 absolute, so for a < 1e4 the estimate is within 0.01 bins of the
        // exact (monotone) formula and rint() of it is off by at most one bin, which two threshold
        // compares repair exactly.  Narrow grids (a >= 1e4) keep the binary search.
        tab->est_a = est_a;
        tab->est_b = est_b;
        tab->use_est = est_ok ? 1 : 0;
        tab->lmin = lmin;
        tab->lmax = lmax;
        tab->range = range;
        tab->r2max = r2max;
        tab->degenerate = (range < 1e-10f) ? 1 : 0;
        tab->levels = levels;
        tab->uniform_ok = (tab->use_est && r2max < 1e30f) ? 1 : 0;   // false for NaN / inf r2max too
        // v_log_f32 is good to 1 ulp of its result; inside a bin the estimate can therefore dip below its value at
        // the bin's lower edge by 2 * a * ulp (taken twice here) plus the rounding of the fma (half an ulp of L)
        const float lmag = fmaxf(fabsf(__builtin_amdgcn_logf(tmin)), fabsf(__builtin_amdgcn_logf(tmax)));
        const float ulp_log = __uint_as_float((__float_as_uint(fmaxf(lmag, 1.0f)) & 0x7f800000u)) * 1.1920929e-7f;
        const float ulp_ne = __uint_as_float((__float_as_uint((float)levels) & 0x7f800000u)) * 1.1920929e-7f;
        // ... and the pair loop's estimate takes r2 from fused multiply-adds: within 4 ulp (4.8e-7 relative, 6.9e-7 in
        // log2) of the reference's separately rounded r2
        const float eta = 4.0f * est_a * ulp_log + ulp_ne + 8.0e-7f * est_a;
        const float delta = 2.0f * (maxdev + eta);
        tab->fast_ok = (fast_try && delta < 0.05f && maxrel <= 2.0e-6f) ? 1 : 0;
        tab->sure_lim = 0.5f - delta;
        tab->kc = kc;
        tab->tm = (int)s_par[2];
        tab->est_bc = est_bc;
        tab->c1 = s_par[0];
        tab->c0c = s_par[1];
        tab->fast_maxdev = maxdev;
        tab->fast_maxrel = maxrel;
        tab->r2max_bits = 0u;      // consumed (every thread read it on entry): ready for the next atomicMax round
    }
}

__global__ void __launch_bounds__(NB_LUT_MIN)
grid_tables_kernel(GridTables *__restrict__ tab, int levels, float G, float eps2, float min_val,
                   PruneState *__restrict__ ps, int allow_fast)
{
    grid_tables_body<false>(tab, levels, G, eps2, min_val, ps, allow_fast, tab->r2max_bits);
}

// ---- tensor-level _grid_quantize_safe (quantization.py:91-127) through the same tables ----------------------------
// The hook used to evaluate a library log twice and an exp once PER ELEMENT in fp64 (fp32 tensors: correctly rounded
// logf / expf) -- ~600 fp64 instructions per element, 200 us for a 4096 x 4096 tensor.  The bin of an element is a step
// function of its value, so: plain min / max of the tensor (log is monotone: lmin / lmax are the logs of the clamped
// extremes), the threshold / value tables for exactly these bounds (the kernel above, bounds read on the device),
// and one pass that looks every element's bin up (v_log_f32 estimate + two threshold compares, or the binary search
// for narrow grids) and writes the bin's value.  Non-finite bounds and a degenerate range keep the elementwise formula.
__global__ void __launch_bounds__(NB_LUT_MIN)
grid_tables_bounds_kernel(GridTables *__restrict__ tab, int levels, float min_val, const double *__restrict__ bounds)
{
    grid_tables_body<false>(tab, levels, 1.0f, 0.0f, min_val, nullptr, 0, 0u, bounds);
}

__global__ void __launch_bounds__(256)
grid_quantize_safe_tab_kernel(const float *__restrict__ in, float *__restrict__ out, int64_t count, int levels, float min_val,
                              const double *__restrict__ bounds, const GridTables *__restrict__ tab, int lp)
{
    extern __shared__ float s_tab[];
    float *s_thr = s_tab, *s_q = s_tab + lp + 1;
    const float mn = (float)bounds[0], mx = (float)bounds[1];
    const bool finite = fabsf(mn) < 1e30f && fabsf(mx) < 1e30f;          // false for NaN / inf as well
    if (!finite) {
        // the formula element by element, as torch evaluates it (NaN / inf propagate through lmin / lmax)
        const float tmin = (mn < min_val) ? min_val : mn, tmax = (mx < min_val) ? min_val : mx;
        const float lmin = (float)log((double)tmin), lmax = (float)log((double)tmax);
        const float range = __fsub_rn(lmax, lmin), lm1 = (float)(levels - 1);
        const bool passthrough = range < 1e-10f;
        for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < count; idx += (int64_t)gridDim.x * 256) {
            float v = in[idx];
            v = (v < min_val) ? min_val : v;
            if (!passthrough) {
                const float lt = (float)log((double)v);
                float x = __fmul_rn(__fdiv_rn(__fsub_rn(lt, lmin), range), lm1);
                const float k = rintf(x);
                x = __fadd_rn(__fmul_rn(__fdiv_rn(k, lm1), range), lmin);
                v = (float)exp((double)x);
                v = (v < min_val) ? min_val : v;
            }
            out[idx] = v;
        }
        return;
    }
    for (int k = threadIdx.x; k <= lp; k += 256) s_thr[k] = k <= levels ? tab->thr[k] : __builtin_inff();
    for (int k = threadIdx.x; k < levels; k += 256) s_q[k] = tab->qval[k];
    __syncthreads();
    if (threadIdx.x == 0) s_thr[levels] = __builtin_inff();      // the table's NaN sentinel would stop the binary search
    __syncthreads();
    const bool passthrough = tab->degenerate != 0;
    const bool use_est = tab->use_est != 0;
    const float est_a = tab->est_a, est_b = tab->est_b;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < count; idx += (int64_t)gridDim.x * 256) {
        float v = in[idx];
        v = (v < min_val) ? min_val : v;
        if (!passthrough) {
            const int k = use_est ? grid_bin_estimate(s_thr, v, est_a, est_b, levels - 1) : grid_bin_lookup(s_thr, v, lp);
            v = s_q[k];
        }
        out[idx] = v;
    }
}

// Small systems (one launch instead of two per grid evaluation): every workgroup adds its part of the all-pairs
// maximum; the LAST one to arrive (device-scope counter) reads the finished maximum and builds the tables.
constexpr int NB_R2MAX_SPLIT = 8;
template <int D, int R>
__global__ void __launch_bounds__(NB_BLOCK)
r2max_tables_kernel(const float *__restrict__ pos, ForceGeom g, float eps2, GridTables *__restrict__ tab, int levels,
                    float G, float min_val, int allow_fast)
{
    static_assert(NB_BLOCK == NB_LUT_MIN, "the last workgroup runs the single-block tables code");
    __shared__ unsigned int s_bits;
    __shared__ int s_mine;
    r2max_block<D, R, NB_R2MAX_SPLIT>(pos, g, eps2, tab);
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned int total = gridDim.x * gridDim.y;
        s_mine = (atomicAdd(&tab->blocks_done, 1u) == total - 1) ? 1 : 0;
        // the maximum itself through the same atomic unit that every workgroup's atomicMax went to
        s_bits = s_mine ? atomicMax(&tab->r2max_bits, 0u) : 0u;
    }
    __syncthreads();
    if (!s_mine) return;
    grid_tables_body<true>(tab, levels, G, eps2, min_val, nullptr, allow_fast, s_bits);
}

// ------------------------------------------------------------------------------------------
// Tracked max-r2 search (round 3): two launches per evaluation instead of six (+ tables).
// Exactness argument as for the pruned search above, with the lower bound LB taken from the far pair of the LAST
// evaluation re-evaluated at the new positions, a fixed centre c, and rho_M = (last measured rho_max) + twice its last
// growth + 1e-4 of it as the bound on every rho: both members of the maximal pair satisfy rho >= sqrt(LB - eps2) - rho_M.  The filter pass
// measures the actual rho_max; if it exceeds rho_M (a particle left the margin) the scan simply takes ALL particles
// -- slower, still exact.  NaN coordinates poison the maximum exactly as torch's max() does.
// ------------------------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(NB_BLOCK)
track_filter_kernel(const float *__restrict__ pos, int n, float eps2, float *__restrict__ cand, int *__restrict__ cand_idx,
                    PruneState *__restrict__ ps)
{
    __shared__ float s_need;
    __shared__ unsigned int s_rho[NB_BLOCK / 64];
    __shared__ int s_nan;
    if (threadIdx.x == 0) {
        float d[D];
#pragma unroll
        for (int k = 0; k < D; ++k) d[k] = __fsub_rn(pos[(size_t)ps->pair_j * D + k], pos[(size_t)ps->pair_i * D + k]);
        const float lb = r2_f32_exact<D>(d, eps2);
        s_need = sqrtf(fmaxf(lb - eps2, 0.0f)) * (1.0f - 1e-5f) - ps->rho_m * (1.0f + 1e-5f);
        s_nan = 0;
    }
    __syncthreads();
    const int i = blockIdx.x * NB_BLOCK + threadIdx.x;
    unsigned int rb = 0u;
    if (i < n) {
        float x[D], s = 0.0f;
        bool bad = false;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            x[k] = pos[(size_t)i * D + k];
            bad |= (x[k] != x[k]);
            const float d = x[k] - ps->c[k];
            s += d * d;
        }
        const float r = sqrtf(s);
        if (bad) s_nan = 1;
        else rb = __float_as_uint(r);              // r >= 0 (or +inf): bit patterns order like the values
        if (!bad && r * (1.0f + 1e-5f) >= s_need) {
            const int slot = atomicAdd(&ps->count, 1);
#pragma unroll
            for (int k = 0; k < D; ++k) cand[(size_t)slot * D + k] = x[k];
            cand_idx[slot] = i;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) rb = max(rb, (unsigned int)__shfl_xor((int)rb, off, 64));
    if ((threadIdx.x & 63) == 0) s_rho[threadIdx.x >> 6] = rb;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int m = s_rho[0];
#pragma unroll
        for (int w = 1; w < NB_BLOCK / 64; ++w) m = max(m, s_rho[w]);
        atomicMax(&ps->rho_cur, m);
        if (s_nan) atomicOr(&ps->nan_flag, 1);
    }
}

// Tail shared by the two tracked-search kernels: block maximum of the scanned pairs -> device-wide maximum (two 64-bit
// atomics per workgroup) -> the LAST workgroup to arrive publishes the maximum, prepares the next evaluation's search
// (far pair, rho bound with its margin, clean counters) and returns true with the maximum's order bits in *bits_out.
// m: particles scanned; rho_cur / nan_flag: measured by the filter pass of this evaluation.
__device__ __forceinline__ bool track_finish(unsigned long long best_i, unsigned long long best_j, int m, float eps2,
                                             unsigned int rho_cur, int nan_flag, PruneState *__restrict__ ps,
                                             unsigned long long *s_ki, unsigned long long *s_kj, int *s_mine,
                                             unsigned int *bits_out)
{
    const int tid = threadIdx.x;
    {
        // block maximum; the partner index travels with the lane that holds the maximum
        unsigned long long ki = best_i, kj = best_j;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned long long oi = __shfl_xor(ki, off, 64), oj = __shfl_xor(kj, off, 64);
            if (oi > ki) { ki = oi; kj = oj; }
        }
        if ((tid & 63) == 0) { s_ki[tid >> 6] = ki; s_kj[tid >> 6] = kj; }
    }
    __syncthreads();
    if (tid == 0) {
        unsigned long long ki = s_ki[0], kj = s_kj[0];
#pragma unroll
        for (int w = 1; w < NB_BLOCK / 64; ++w)
            if (s_ki[w] > ki) { ki = s_ki[w]; kj = s_kj[w]; }
        if (ki) {
            // two independent maxima: on an exact tie of r2 they may name particles of different pairs -- harmless, the
            // pair only provides the next evaluation's lower bound, which is re-evaluated from the positions
            atomicMax(&ps->best_i, ki);
            atomicMax(&ps->best_j, kj);
        }
        __threadfence();
        *s_mine = (atomicAdd(&ps->scan_done, 1u) == gridDim.x - 1) ? 1 : 0;
        if (*s_mine) {
            const unsigned long long fi = atomicMax(&ps->best_i, 0ull), fj = atomicMax(&ps->best_j, 0ull);
            unsigned int bits = (unsigned int)(fi >> 32);
            if (m <= 0) bits = 0u;
            if (nan_flag) bits = 0x7fc00000u;                     // a NaN coordinate: torch's max() would be NaN
            // a single particle (or none scanned): the diagonal pair r2 = eps2 is the maximum
            if (bits == 0u) bits = r2_order_bits(eps2);
            *bits_out = bits;
            // next evaluation: this maximum's pair, the measured rho_max plus margin, clean counters
            if (fi) { ps->pair_i = (int)(fi & 0xffffffffull); ps->pair_j = (int)(fj & 0xffffffffull); }
            {
                // margin for the next evaluation: twice the growth rho_max showed over this step (an escaper keeps its
                // speed) + 1e-4 relative; the first tracked evaluation has no growth figure yet and keeps 0.5 %
                const float rc = __uint_as_float(rho_cur);
                const float grow = (ps->rho_prev > 0.0f) ? fmaxf(rc - ps->rho_prev, 0.0f) : 0.0025f * rc;
                ps->rho_m = rc + 2.0f * grow + 1e-4f * rc;
                ps->rho_prev = rc;
            }
            ps->seeded = (nan_flag == 0) ? 1 : 0;
            ps->rho_cur = 0u;
            ps->count = 0;
            ps->nan_flag = 0;
            ps->best_i = 0ull;
            ps->best_j = 0ull;
            ps->scan_done = 0u;
        }
    }
    __syncthreads();
    return *s_mine != 0;
}

constexpr int NB_TRACK_BLOCKS = 128;     // scan workgroups: every one ends in up to three same-address atomics (~17 ns each)

template <int D>
__global__ void __launch_bounds__(NB_BLOCK)
track_scan_kernel(const float *__restrict__ pos, const float *__restrict__ cand, const int *__restrict__ cand_idx, int n,
                  float eps2, PruneState *__restrict__ ps, GridTables *__restrict__ tab, int levels, float G, float min_val,
                  int allow_fast, int fuse_tables)
{
    static_assert(NB_BLOCK == NB_LUT_MIN, "the last workgroup runs the single-block tables code");
    __shared__ float sj[D][NB_TJ];
    __shared__ int sji[NB_TJ];
    __shared__ unsigned long long s_ki[NB_BLOCK / 64], s_kj[NB_BLOCK / 64];
    __shared__ unsigned int s_bits;
    __shared__ int s_mine;
    const int tid = threadIdx.x;
    // the filter pass measured the real rho_max: inside the assumed bound the candidates are complete, else take everyone
    const bool valid = ps->rho_cur <= __float_as_uint(ps->rho_m);
    const int m = valid ? ps->count : n;
    const float *src = valid ? cand : pos;
    const int T = (m + NB_TJ - 1) / NB_TJ;
    const long long npairs = (long long)T * (T + 1) / 2;        // tile pairs (it <= jt), row-major
    unsigned long long best_i = 0ull, best_j = 0ull;
    for (long long p = blockIdx.x; p < npairs; p += gridDim.x) {
        // p -> (it, jt): rows of lengths T, T - 1, ...; first[it] = it * T - it * (it - 1) / 2
        const double tt = 2.0 * T + 1.0;
        int it = (int)((tt - sqrt(tt * tt - 8.0 * (double)p)) * 0.5);
        it = max(0, min(it, T - 1));
        while (it > 0 && (long long)it * T - (long long)it * (it - 1) / 2 > p) --it;
        while ((long long)(it + 1) * T - (long long)(it + 1) * it / 2 <= p) ++it;
        const int jt = it + (int)(p - ((long long)it * T - (long long)it * (it - 1) / 2));
        const int i = min(it * NB_TJ + tid, m - 1), j = min(jt * NB_TJ + tid, m - 1);
        float xi[D];
#pragma unroll
        for (int k = 0; k < D; ++k) xi[k] = src[(size_t)i * D + k];
        const int ii = valid ? cand_idx[i] : i;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < D; ++k) sj[k][tid] = src[(size_t)j * D + k];
        sji[tid] = valid ? cand_idx[j] : j;
        __syncthreads();
        const int cnt = min(NB_TJ, m - jt * NB_TJ);
        unsigned int best = 0u;
        int bj = 0;
#pragma unroll 8
        for (int jj = 0; jj < cnt; ++jj) {
            float d[D];
#pragma unroll
            for (int k = 0; k < D; ++k) d[k] = __fsub_rn(sj[k][jj], xi[k]);
            const unsigned int b = r2_order_bits(r2_f32_exact<D>(d, eps2));
            bj = b > best ? jj : bj;
            best = max(best, b);
        }
        const unsigned long long ki = ((unsigned long long)best << 32) | (unsigned int)ii;
        const unsigned long long kj = ((unsigned long long)best << 32) | (unsigned int)sji[bj];
        if (ki > best_i) { best_i = ki; best_j = kj; }
    }
    const bool mine = track_finish(best_i, best_j, m, eps2, ps->rho_cur, ps->nan_flag, ps, s_ki, s_kj, &s_mine, &s_bits);
    if (!mine) return;
    if (!fuse_tables) {
        if (tid == 0) tab->r2max_bits = s_bits;                  // the multi-block tables kernel follows
        return;
    }
    grid_tables_body<true>(tab, levels, G, eps2, min_val, nullptr, allow_fast, s_bits);
}

// debug / parity: bin index of every pair with the tables the force kernel used
template <int D>
__global__ void __launch_bounds__(NB_BLOCK)
d2bins_kernel(const float *__restrict__ pos, int n, float eps2, const GridTables *__restrict__ tab,
              int16_t *__restrict__ bins, int i0)
{
    const int j = blockIdx.x * NB_BLOCK + threadIdx.x;
    const int i = i0 + blockIdx.y;
    if (j >= n) return;
    float d[D];
#pragma unroll
    for (int k = 0; k < D; ++k) d[k] = __fsub_rn(pos[(size_t)j * D + k], pos[(size_t)i * D + k]);
    const float r2 = r2_f32_exact<D>(d, eps2);
    int b = -1;
    if (!tab->degenerate) {
        b = 0;
        for (int k = 1; k < tab->levels; ++k) b += (tab->thr[k] <= r2) ? 1 : 0;
    }
    bins[(size_t)(i - i0) * n + j] = (int16_t)b;
}

template <typename F>
hipError_t dispatch_dim(int dim, F &&f)
{
    if (dim == 2) return f(std::integral_constant<int, 2>{});
    if (dim == 3) return f(std::integral_constant<int, 3>{});
    return hipErrorInvalidValue;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
constexpr int R_F32 = 2;

hipError_t nb_launch_force_f64(const double *pos, const double *mass, double *partial, const ForceGeom &g,
                               int dim, int pair_dt, int qhook, double G, double eps2_py, float eps2_pair,
                               hipStream_t st)
{
    // pair_dt: -1 (fp64 pairs) or the dtype the positions are typed as (NB_F32 / NB_F16 / NB_BF16)
    const int r = (pair_dt >= 0 || qhook >= 0) ? 2 : g.r;   // the special variants are compiled for R = 2 only
    const dim3 grid((g.n + NB_BLOCK * r - 1) / (NB_BLOCK * r), g.nchunks);
#define NB_F64(DD, RR, PA, QH) \
    hipLaunchKernelGGL((force_f64_kernel<DD, RR, PA, QH>), grid, dim3(NB_BLOCK), 0, st, pos, mass, partial, g, G, eps2_py, eps2_pair)
    return dispatch_dim(dim, [&](auto D) {
        constexpr int DD = decltype(D)::value;
        if (qhook == HOOK_NONE) NB_F64(DD, 2, -1, HOOK_NONE);
        else if (qhook == HOOK_BF16) NB_F64(DD, 2, -1, HOOK_BF16);
        else if (qhook == HOOK_F16) NB_F64(DD, 2, -1, HOOK_F16);
        else if (qhook >= 0) return hipErrorInvalidValue;
        else if (pair_dt == NB_F32) NB_F64(DD, 2, NB_F32, -1);
        else if (pair_dt == NB_F16) NB_F64(DD, 2, NB_F16, -1);
        else if (pair_dt == NB_BF16) NB_F64(DD, 2, NB_BF16, -1);
        else if (g.r == 1) NB_F64(DD, 1, -1, -1);
        else if (g.r == 2) NB_F64(DD, 2, -1, -1);
        else NB_F64(DD, 4, -1, -1);
        return hipGetLastError();
    });
#undef NB_F64
}

hipError_t nb_launch_force_f32(const float *pos, const float *mass, double *partial, const ForceGeom &g,
                               int dim, int hook, int pa, float G, float eps2, const GridTables *tab, int levels,
                               hipStream_t st, unsigned long long *bin_out)
{
    const int lp = hook == HOOK_GRID ? nb_lut_pad(levels) : 0;
    const size_t lds = hook == HOOK_GRID ? nb_lut_lds_bytes(levels) : 0;
    // small systems are parallelism-bound: one target per thread doubles the workgroups (like the fp64 kernel)
    const int r = g.n <= 8192 ? 1 : R_F32;
    const dim3 grid((g.n + NB_BLOCK * r - 1) / (NB_BLOCK * r), g.nchunks);
#define NB_F32KB(DD, HH, PP, BB)                                                                                         \
    do {                                                                                                                 \
        if (r == 1)                                                                                                      \
            hipLaunchKernelGGL((force_f32_kernel<DD, 1, HH, PP, BB>), grid, dim3(NB_BLOCK), lds, st, pos, mass, partial, g, \
                               G, eps2, tab, lp, bin_out);                                                               \
        else                                                                                                             \
            hipLaunchKernelGGL((force_f32_kernel<DD, R_F32, HH, PP, BB>), grid, dim3(NB_BLOCK), lds, st, pos, mass, partial, \
                               g, G, eps2, tab, lp, bin_out);                                                            \
    } while (0)
#define NB_F32K(DD, HH, PP) NB_F32KB(DD, HH, PP, false)
    return dispatch_dim(dim, [&](auto D) {
        constexpr int DD = decltype(D)::value;
        if (bin_out) {           // bin read-out of the grid hook (nb_quant_bin_sums): same body, BINS = true
            if (hook != HOOK_GRID || pa != NB_F32) return hipErrorInvalidValue;
            NB_F32KB(DD, HOOK_GRID, NB_F32, true);
            return hipGetLastError();
        }
        if (pa != NB_F32) {
            // half-typed state: cast hooks only (a grid over a half tensor is not implemented)
            if (hook == HOOK_GRID) return hipErrorInvalidValue;
            if (pa == NB_F16) {
                if (hook == HOOK_NONE) NB_F32K(DD, HOOK_NONE, NB_F16);
                else if (hook == HOOK_BF16) NB_F32K(DD, HOOK_BF16, NB_F16);
                else NB_F32K(DD, HOOK_F16, NB_F16);
            } else {
                if (hook == HOOK_NONE) NB_F32K(DD, HOOK_NONE, NB_BF16);
                else if (hook == HOOK_BF16) NB_F32K(DD, HOOK_BF16, NB_BF16);
                else NB_F32K(DD, HOOK_F16, NB_BF16);
            }
            return hipGetLastError();
        }
        switch (hook) {
        case HOOK_NONE: NB_F32K(DD, HOOK_NONE, NB_F32); break;
        case HOOK_BF16: NB_F32K(DD, HOOK_BF16, NB_F32); break;
        case HOOK_F16: NB_F32K(DD, HOOK_F16, NB_F32); break;
        case HOOK_GRID: NB_F32K(DD, HOOK_GRID, NB_F32); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    });
#undef NB_F32K
#undef NB_F32KB
}

hipError_t nb_launch_r2max(const float *pos, const ForceGeom &g, int dim, float eps2, GridTables *tab,
                           hipStream_t st)
{
    // small systems (the only ones that take this path by default): one target per thread, four times the
    // workgroups
    const int r = g.n <= 8192 ? 1 : 4;
    const dim3 grid((g.n + NB_BLOCK * r - 1) / (NB_BLOCK * r), g.nchunks);
    return dispatch_dim(dim, [&](auto D) {
        constexpr int DD = decltype(D)::value;
        if (r == 1) hipLaunchKernelGGL((r2max_kernel<DD, 1>), grid, dim3(NB_BLOCK), 0, st, pos, g, eps2, tab);
        else hipLaunchKernelGGL((r2max_kernel<DD, 4>), grid, dim3(NB_BLOCK), 0, st, pos, g, eps2, tab);
        return hipGetLastError();
    });
}

hipError_t nb_launch_r2max_tables(const float *pos, const ForceGeom &g, int dim, float eps2, GridTables *tab, int levels,
                                  float G, float min_val, int allow_fast, hipStream_t st)
{
    if (levels > NB_LUT_MIN) return hipErrorInvalidValue;
    const int per_block = NB_BLOCK / NB_R2MAX_SPLIT;
    const int gx = (g.n + per_block - 1) / per_block;
    // every workgroup ends in two atomics on one address (they serialise in the L2: ~25 ns each), so about one
    // workgroup per CU is the right number: fewer, longer source chunks than the force kernels use
    ForceGeom g2 = g;
    const int njr = g.j_end - g.j_begin > 0 ? g.j_end - g.j_begin : 1;
    int nch = 256 / gx;
    nch = nch < 1 ? 1 : (nch > g.nchunks ? g.nchunks : nch);
    int chunk = (njr + nch - 1) / nch;
    chunk = (chunk + NB_TJ - 1) / NB_TJ * NB_TJ;
    g2.chunk_len = chunk;
    g2.nchunks = (njr + chunk - 1) / chunk;
    const dim3 grid(gx, g2.nchunks);
    return dispatch_dim(dim, [&](auto D) {
        constexpr int DD = decltype(D)::value;
        hipLaunchKernelGGL((r2max_tables_kernel<DD, 1>), grid, dim3(NB_BLOCK), 0, st, pos, g2, eps2, tab, levels, G, min_val,
                           allow_fast);
        return hipGetLastError();
    });
}

hipError_t nb_launch_r2max_pruned(const float *pos, int n, int dim, float eps2, float *cand, float *rho,
                                  PruneState *ps, GridTables *tab, hipStream_t st)
{
    const int blocks = (n + NB_BLOCK - 1) / NB_BLOCK;
    const int bbox_blocks = blocks < 128 ? blocks : 128;
    // *ps is in its reset state on entry: nb_api.cpp initialises it once, grid_tables_kernel (which
    // always follows) puts it back after use -- no per-step memset / copy launches
    return dispatch_dim(dim, [&](auto D) {
        constexpr int DD = decltype(D)::value;
        hipLaunchKernelGGL((prune_bbox_kernel<DD>), dim3(bbox_blocks), dim3(NB_BLOCK), 0, st, pos, n, ps);
        hipLaunchKernelGGL((prune_rho_kernel<DD>), dim3(blocks), dim3(NB_BLOCK), 0, st, pos, n, rho, ps);
        hipLaunchKernelGGL((prune_hop_kernel<DD>), dim3(blocks), dim3(NB_BLOCK), 0, st, pos, n, eps2, &ps->far, &ps->lb[0]);
        hipLaunchKernelGGL((prune_hop_kernel<DD>), dim3(blocks), dim3(NB_BLOCK), 0, st, pos, n, eps2, &ps->lb[0], &ps->lb[1]);
        hipLaunchKernelGGL((prune_compact_kernel<DD>), dim3(blocks), dim3(NB_BLOCK), 0, st, pos, rho, n, eps2, cand, ps);
        hipLaunchKernelGGL((prune_scan_kernel<DD>), dim3(blocks), dim3(NB_BLOCK), 0, st, cand, eps2, ps, tab);
        return hipGetLastError();
    });
}

hipError_t nb_launch_r2max_tracked(const float *pos, int n, int dim, float eps2, float *cand, int *cand_idx, PruneState *ps,
                                   GridTables *tab, int levels, float G, float min_val, int allow_fast, hipStream_t st)
{
    const int blocks = (n + NB_BLOCK - 1) / NB_BLOCK;
    const int fuse = levels <= NB_LUT_MIN ? 1 : 0;
    // (one launch instead of two -- every scan workgroup filtering for itself -- was measured slower at N = 6000 / 12 000:
    // 52.7 / 96.9 vs 41.8 / 73.2 us per INT8 step, profiles/r03_tracked_fused_one_launch_ab.txt)
    return dispatch_dim(dim, [&](auto D) {
        constexpr int DD = decltype(D)::value;
        hipLaunchKernelGGL((track_filter_kernel<DD>), dim3(blocks), dim3(NB_BLOCK), 0, st, pos, n, eps2, cand, cand_idx, ps);
        hipLaunchKernelGGL((track_scan_kernel<DD>), dim3(NB_TRACK_BLOCKS), dim3(NB_BLOCK), 0, st, pos, cand, cand_idx, n, eps2, ps,
                           tab, levels, G, min_val, allow_fast, fuse);
        return hipGetLastError();
    });
}

hipError_t nb_launch_grid_quantize_safe_tab(const float *in, float *out, int64_t count, int levels, float min_val,
                                            const double *bounds, GridTables *tab, hipStream_t st)
{
    if (levels < 2 || levels > NB_MAX_LUT) return hipErrorInvalidValue;
    // the scalar tail of the tables (arrival counter, flags) starts from zero; the arrays are fully rewritten
    hipError_t e = hipMemsetAsync(&tab->lmin, 0, sizeof(GridTables) - offsetof(GridTables, lmin), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(grid_tables_bounds_kernel, dim3((levels + NB_LUT_MIN - 1) / NB_LUT_MIN), dim3(NB_LUT_MIN), 0, st, tab,
                       levels, min_val, bounds);
    const int lp = nb_lut_pad(levels);
    int grid = (int)((count + 1023) / 1024);
    grid = grid < 1 ? 1 : (grid > 4096 ? 4096 : grid);
    hipLaunchKernelGGL(grid_quantize_safe_tab_kernel, dim3(grid), dim3(256), (size_t)(lp + 1 + levels) * sizeof(float), st, in,
                       out, count, levels, min_val, bounds, tab, lp);
    return hipGetLastError();
}

hipError_t nb_launch_grid_tables(GridTables *tab, int levels, float G, float eps2, float min_val, PruneState *ps,
                                 hipStream_t st, int allow_fast)
{
    hipLaunchKernelGGL(grid_tables_kernel, dim3((levels + NB_LUT_MIN - 1) / NB_LUT_MIN), dim3(NB_LUT_MIN), 0, st, tab,
                       levels, G, eps2, min_val, ps, allow_fast);
    return hipGetLastError();
}

hipError_t nb_launch_d2bins(const float *pos, int n, int dim, float eps2, const GridTables *tab, int16_t *bins,
                            hipStream_t st, int i0, int i1)
{
    if (i1 < 0) i1 = n;
    const dim3 grid((n + NB_BLOCK - 1) / NB_BLOCK, i1 - i0);
    return dispatch_dim(dim, [&](auto D) {
        constexpr int DD = decltype(D)::value;
        hipLaunchKernelGGL((d2bins_kernel<DD>), grid, dim3(NB_BLOCK), 0, st, pos, n, eps2, tab, bins, i0);
        return hipGetLastError();
    });
}
