// nb_small.hip -- one launch per leapfrog step for small systems (N up to a few thousand).
//
// Small systems are bound by LATENCY, not throughput: a force launch, a reduction launch and the kernel boundaries
// between them (~1.5 us each, /opt/skills/guides/MI355X_MICROARCH.md "boundary") cost more than the arithmetic
// (N = 1024: 0.5 M pairs = 0.3 us of the chip's VALU time).  A grid barrier inside a persistent kernel costs 4-10 us
// on this machine (same guide, "barrier-xcd"), i.e. MORE than a kernel boundary, so the step is not made persistent;
// instead the whole step is ONE launch with no cross-workgroup reduction at all:
//   * S lanes of a wavefront share one target and split the sources between them (S = 16 / 32 / 64 by size), so a
//     target's sum is finished by a fixed butterfly of wave shuffles -- no slabs, no second kernel;
//   * every workgroup stages the sources through LDS in tiles of 1024 (coalesced loads; lanes of a group read
//     consecutive entries, groups read the same entries: conflict-free broadcasts);
//   * the lane that holds a finished sum applies the closing half kick and, inside nb_step, the NEXT step's opening
//     kick + drift, writing the new positions to a second buffer (other workgroups still read the old ones): the
//     positions ping-pong between two buffers, one launch per step.
// One-sided (each ordered pair evaluated), which at these sizes is free: the chip is mostly idle.
// Arithmetic per pair is that of the tuned kernels (fp64: v_rsq_f64 + second-order correction; fp32: reference op
// order for r2 without fma, cast hooks, v_rsq_f32 + first-order correction, fp32 products summed in fp64).
#include "nb_device.h"

#include <cstdlib>

int nb_small_block(int n);

namespace {

using namespace nbdev;

constexpr int SM_TILE = 1024;     // sources per LDS tile (2048 measured: slower -- fewer workgroups per CU)

__device__ __forceinline__ double inv_r3_d(double q)
{
    const double y0 = __builtin_amdgcn_rsq(q);
    const double y02 = y0 * y0;
    const double e = __builtin_fma(-q, y02, 1.0);
    const double v = y0 * y02;
    const double c = __builtin_fma(e, 1.875, 1.5);
    return __builtin_fma(v, c * e, v);
}
__device__ __forceinline__ float inv_r3_f(float q)
{
    const float y0 = __builtin_amdgcn_rsqf(q);
    const float y02 = y0 * y0;
    const float e = __builtin_fmaf(-q, y02, 1.0f);
    const float v = y0 * y02;
    return __builtin_fmaf(v * e, 1.5f, v);
}

template <typename T> __device__ __forceinline__ T axpy_sep(T a, T b, T s);      // a + b*s, two roundings like torch
template <> __device__ __forceinline__ double axpy_sep<double>(double a, double b, double s) { return __dadd_rn(a, __dmul_rn(b, s)); }
template <> __device__ __forceinline__ float axpy_sep<float>(float a, float b, float s) { return __fadd_rn(a, __fmul_rn(b, s)); }

// do_kick: 0 force only; 1 + closing half kick; 2 + next step's opening kick + drift (positions -> pos_out); 3 closing
// kick + the next step's drifted positions speculatively into pos_out; | 4: vel holds the previous step's closing state,
// this step's opening kick is applied on read with the previous accelerations
// BINS (grid hook only): the same body with the quant-bin read-out -- per-target integer checksums s1 = sum_j k,
// s2 = sum_j k ((j mod 65521) + 1) of the bin every pair was given, by whichever route the production code took
// (table-free estimate / wave ballot / threshold fallback); bin_out = {s1[n], s2[n], {table-free pairs, table pairs}}.
// BS: threads per workgroup.  Every workgroup streams ALL sources through its LDS, so the L2 -> LDS traffic of a step is
// N^2 * 24 B / (targets per workgroup): 512 threads (8 targets of 64 lanes) halve it against 256; used up to N = 2048,
// where all workgroups still run in one round (nb_small_block).
template <typename T, int D, int HOOK, int S, bool BINS = false, int BS = NB_BLOCK>
__global__ void __launch_bounds__(BS)
small_step_kernel(const T *__restrict__ pos_in, T *__restrict__ pos_out, T *__restrict__ vel, T *__restrict__ acc,
                  const T *__restrict__ mass, int n, T G, T eps2, T half_dt, T dt, int do_kick,
                  const GridTables *__restrict__ tab, double *__restrict__ part, unsigned long long *__restrict__ bin_out)
{
    constexpr bool F64 = sizeof(T) == 8;
    constexpr int TG = BS / S;                       // targets per workgroup
    __shared__ T sx[D][SM_TILE];
    __shared__ T sg[SM_TILE];                        // G * m_j (fp32: the reference's (1/p * G) * m_j order is kept below)
    // grid hook (INT8 / INT4 / CUSTOM up to 256 levels): the evaluation's tables (nb_force.hip grid_tables_kernel)
    __shared__ float s_thr[HOOK == HOOK_GRID ? NB_LUT_MIN + 1 : 1], s_lut[HOOK == HOOK_GRID ? NB_LUT_MIN + 1 : 1];
    const int tid = threadIdx.x;
    bool g_fast = false, g_est = false, g_deg = false;
    float est_a = 0.0f, est_b = 0.0f, est_bc = 0.0f, sure_lim = 0.0f, c1 = 0.0f, c0c = 0.0f, kcf = 0.0f;
    int g_levels = 0, g_tm = 0;
    if constexpr (HOOK == HOOK_GRID) {
        g_levels = tab->levels;
        for (int k = tid; k <= NB_LUT_MIN; k += BS) {
            s_thr[k] = (k <= g_levels) ? tab->thr[k] : __builtin_inff();     // thr[levels] = NaN sentinel, +inf padding
            s_lut[k] = (k < g_levels) ? tab->lut[k] : 0.0f;
        }
        g_fast = tab->fast_ok != 0;
        g_est = tab->use_est != 0;
        g_deg = tab->degenerate != 0;
        est_a = tab->est_a; est_b = tab->est_b; est_bc = tab->est_bc; sure_lim = tab->sure_lim;
        c1 = tab->c1; c0c = tab->c0c; kcf = (float)tab->kc; g_tm = tab->tm;
    }
    const int grp = tid / S, l = tid % S;
    const int i_raw = blockIdx.x * TG + grp;
    const bool live = i_raw < n;
    const int i = live ? i_raw : n - 1;
    T xi[D];
#pragma unroll
    for (int k = 0; k < D; ++k) xi[k] = pos_in[(size_t)i * D + k];
    double a[D];
#pragma unroll
    for (int k = 0; k < D; ++k) a[k] = 0.0;
    long long b1 = 0, b2 = 0, bfast = 0, bexact = 0;     // BINS only

    for (int j0 = 0; j0 < n; j0 += SM_TILE) {
        __syncthreads();
        // entries past the end, up to the pair loop's stride: padding (far away, massless)
        constexpr int STRIDE = (HOOK == HOOK_GRID) ? 4 * S : S;
        const int cnt_ld = min(SM_TILE, (min(SM_TILE, n - j0) + STRIDE - 1) / STRIDE * STRIDE);
        // (a "flat" variant -- consecutive threads reading consecutive elements of the (N, D) array and scattering them
        // into the component arrays -- measured slower on the same box: 6.8 vs 5.4 us per step at N = 1024 fp64)
        for (int t = tid; t < cnt_ld; t += BS) {
            const int j = j0 + t;
            if (j < n) {
#pragma unroll
                for (int k = 0; k < D; ++k) sx[k][t] = pos_in[(size_t)j * D + k];
                sg[t] = F64 ? (T)(G * mass[j]) : mass[j];
            } else {
#pragma unroll
                for (int k = 0; k < D; ++k) sx[k][t] = F64 ? (T)1e150 : (T)1e18;
                sg[t] = (T)0;
            }
        }
        __syncthreads();
        const int cnt_up = cnt_ld;                   // padding entries are harmless
        if constexpr (HOOK == HOOK_GRID) {
            // four sources per iteration: independent log / exp chains in flight, ONE edge test for all of them
            constexpr int U = 4;
            for (int jj = l; jj < cnt_up; jj += U * S) {
                float dd[U][D], q[U], w[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int k = 0; k < D; ++k) dd[u][k] = __fsub_rn(sx[k][jj + u * S], xi[k]);
                    q[u] = r2_f32_exact<D>(dd[u], eps2);
                }
                // (1 / q_k^1.5) * G of each pair's bin: table-free when no pair of the wave sits on a bin edge
                // (DESIGN.md section 4.3), else floor(estimate) + one threshold compare, else binary search
                int kb[U];                 // BINS: the bin each pair was given
                bool kexact = true;
                if (g_deg) {
#pragma unroll
                    for (int u = 0; u < U; ++u) { w[u] = __fmul_rn(inv_r3_f(q[u] < 0.01f ? 0.01f : q[u]), G); kb[u] = 0; }
                } else if (g_fast) {
                    float kf[U], dev = 0.0f;
                    bool bad = false;
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const float ne = __builtin_fmaf(__builtin_amdgcn_logf(q[u]), est_a, est_bc);
                        kf[u] = __builtin_rintf(ne);
                        const float dv = __builtin_fabsf(ne - kf[u]);
                        bad |= !(dv <= sure_lim);                                 // also true for NaN
                        dev = __builtin_fmaxf(dev, dv);
                    }
                    if (__builtin_amdgcn_ballot_w64(bad) != 0ull) {
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            kb[u] = grid_bin_floor_estimate(s_thr, q[u], est_a, est_b, g_levels - 2);
                            w[u] = s_lut[kb[u]];
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < U; ++u) {        // max(): softening^2 below the grid's floor
                            const float kc_ = __builtin_fmaxf(kf[u], -kcf);
                            w[u] = ldexpf(__builtin_amdgcn_exp2f(__builtin_fmaf(kc_, c1, c0c)), g_tm);
                            if constexpr (BINS) kb[u] = (int)__builtin_fminf(kc_ + kcf, 1e6f);
                        }
                        kexact = false;
                    }
                } else if (g_est) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        kb[u] = grid_bin_floor_estimate(s_thr, q[u], est_a, est_b, g_levels - 2);
                        w[u] = s_lut[kb[u]];
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        kb[u] = grid_bin_lookup(s_thr, q[u], NB_LUT_MIN);
                        w[u] = s_lut[kb[u]];
                    }
                }
                if constexpr (BINS) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int j = j0 + jj + u * S;
                        if (j < n) {             // padding entries of the tile take part in no pair
                            b1 += kb[u];
                            b2 += (long long)kb[u] * (j % 65521 + 1);
                            if (kexact) ++bexact; else ++bfast;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float wm = __fmul_rn(w[u], sg[jj + u * S]);
#pragma unroll
                    for (int k = 0; k < D; ++k) a[k] += (double)__fmul_rn(wm, dd[u][k]);
                }
            }
        } else {
#pragma unroll 4
        for (int jj = l; jj < cnt_up; jj += S) {
            T d[D];
            if constexpr (F64) {
#pragma unroll
                for (int k = 0; k < D; ++k) d[k] = sx[k][jj] - xi[k];
                double q = __builtin_fma(d[D - 1], d[D - 1], eps2);
#pragma unroll
                for (int k = D - 2; k >= 0; --k) q = __builtin_fma(d[k], d[k], q);
                const double w = inv_r3_d(q) * sg[jj];
#pragma unroll
                for (int k = 0; k < D; ++k) a[k] = __builtin_fma(w, d[k], a[k]);
            } else {
#pragma unroll
                for (int k = 0; k < D; ++k) d[k] = __fsub_rn(sx[k][jj], xi[k]);
                float q = r2_f32_exact<D>(d, eps2);
                if (HOOK == HOOK_BF16) q = round_bf16(q);
                if (HOOK == HOOK_F16) q = round_f16(q);
                float w = __fmul_rn(inv_r3_f(q), G);
                if (HOOK == HOOK_F16) w = (q == __builtin_inff()) ? 0.0f : w;      // pow(inf) = inf -> G / inf = 0 upstream
                w = __fmul_rn(w, sg[jj]);
#pragma unroll
                for (int k = 0; k < D; ++k) a[k] += (double)__fmul_rn(w, d[k]);
            }
        }
        }
    }
    // the S lanes of a target: fixed butterfly
#pragma unroll
    for (int off = S / 2; off >= 1; off >>= 1) {
#pragma unroll
        for (int k = 0; k < D; ++k) a[k] += __shfl_xor(a[k], off, 64);
    }
    if constexpr (BINS) {
#pragma unroll
        for (int off = S / 2; off >= 1; off >>= 1) {
            b1 += __shfl_xor(b1, off, 64);
            b2 += __shfl_xor(b2, off, 64);
            bfast += __shfl_xor(bfast, off, 64);
            bexact += __shfl_xor(bexact, off, 64);
        }
        if (live && l == 0) {
            bin_out[i] = (unsigned long long)b1;
            bin_out[(size_t)n + i] = (unsigned long long)b2;
            atomicAdd(&bin_out[2 * (size_t)n], (unsigned long long)bfast);
            atomicAdd(&bin_out[2 * (size_t)n + 1], (unsigned long long)bexact);
        }
    }
    __shared__ double s_mm[BS / 16][2];        // INT8 / INT4: min / max of this workgroup's force components
    double lo = __builtin_inf(), hi = -__builtin_inf();
    if (live && l == 0) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const size_t idx = (size_t)i * D + k;
            const T ak = (T)a[k];
            const T a_prev = (do_kick & 4) ? acc[idx] : (T)0;     // (read before this evaluation's force replaces it)
            acc[idx] = ak;
            const double av = (double)ak;              // NaN-propagating like torch's min() / max()
            lo = (av != av || lo != lo) ? __builtin_nan("") : (av < lo ? av : lo);
            hi = (av != av || hi != hi) ? __builtin_nan("") : (av > hi ? av : hi);
            const int kmode = do_kick & 3;
            if (kmode >= 1) {
                T v = vel[idx];
                if (do_kick & 4) v = axpy_sep<T>(v, a_prev, half_dt);     // this step's opening kick, deferred (see below)
                v = axpy_sep<T>(v, ak, half_dt);                          // closing kick (simulation.py:141)
                if (kmode == 2) {
                    v = axpy_sep<T>(v, ak, half_dt);                      // next step's opening kick (:132)
                    pos_out[idx] = axpy_sep<T>(xi[k], v, dt);             // ... and drift (:135)
                } else if (kmode == 3) {
                    // last step of a native call: velocities stay at the closing kick (what a reader must see), but the
                    // positions the NEXT step would drift to go to pos_out -- if the next nb_step finds the state
                    // untouched it takes them and applies its opening kick here on read (flag 4): a Python loop of
                    // step() costs one launch per step instead of two
                    const T vo = axpy_sep<T>(v, ak, half_dt);
                    pos_out[idx] = axpy_sep<T>(xi[k], vo, dt);
                }
                vel[idx] = v;
            }
        }
    }
    if (part) {                                      // kernel-uniform
        if (l == 0) { s_mm[grp][0] = lo; s_mm[grp][1] = hi; }    // dead targets hold (+inf, -inf): neutral
        __syncthreads();
        if (tid == 0) {
            double mn = s_mm[0][0], mx = s_mm[0][1];
            for (int g = 1; g < TG; ++g) {
                const double a0 = s_mm[g][0], a1 = s_mm[g][1];
                mn = (a0 != a0 || mn != mn) ? __builtin_nan("") : (a0 < mn ? a0 : mn);
                mx = (a1 != a1 || mx != mx) ? __builtin_nan("") : (a1 > mx ? a1 : mx);
            }
            part[2 * (size_t)blockIdx.x] = mn;
            part[2 * (size_t)blockIdx.x + 1] = mx;
        }
    }
}

template <typename T, int D, int HOOK, bool BINS = false>
hipError_t launch_s(const T *pos_in, T *pos_out, T *vel, T *acc, const T *mass, int n, double G, double eps2, double half_dt,
                    double dt, int do_kick, int lanes, hipStream_t st, const GridTables *tab = nullptr, double *part = nullptr,
                    unsigned long long *bin_out = nullptr)
{
#define NB_SMALL(SS)                                                                                                       \
    do {                                                                                                                   \
        if (nb_small_block(n) == 512)                                                                                      \
            hipLaunchKernelGGL((small_step_kernel<T, D, HOOK, SS, BINS, 512>), dim3((n + 512 / SS - 1) / (512 / SS)), dim3(512), 0, \
                               st, pos_in, pos_out, vel, acc, mass, n, (T)G, (T)eps2, (T)half_dt, (T)dt, do_kick, tab, part, bin_out); \
        else                                                                                                               \
            hipLaunchKernelGGL((small_step_kernel<T, D, HOOK, SS, BINS, 256>), dim3((n + 256 / SS - 1) / (256 / SS)), dim3(256), 0, \
                               st, pos_in, pos_out, vel, acc, mass, n, (T)G, (T)eps2, (T)half_dt, (T)dt, do_kick, tab, part, bin_out); \
    } while (0)
    if (lanes == 64) NB_SMALL(64);
    else if (lanes == 32) NB_SMALL(32);
    else NB_SMALL(16);
#undef NB_SMALL
    return hipGetLastError();
}

}  // namespace

// lanes per target by size: enough lanes to keep a lane's source loop short, few enough that the source tiles every
// workgroup re-reads from L2 stay small (N^2 * S * 0.1 bytes per step)
// threads per workgroup of the one-launch step (see small_step_kernel): NB_SMALL_BLOCK overrides (A/B)
int nb_small_block(int n)
{
    static const int forced = getenv("NB_SMALL_BLOCK") ? atoi(getenv("NB_SMALL_BLOCK")) : 0;
    if (forced == 256 || forced == 512) return forced;
    // measured fp64 us per step, 256 / 512 / 1024 threads: N = 1024 5.4 / 5.2 / -, 2048 8.4 / 7.5 / -, 2500 10.9 / 11.3 / 11.3,
    // 3000 11.9 / 12.5 / 12.4, 4096 17.2 / 17.4 / 24.8: the larger workgroup wins while all of them fit the chip in one round
    // ... and again above N = 3072 (fp64 only gets there), with 64 lanes per target: N = 4096 15.7 (512 x 64) vs 17.2
    // (256 x 32) / 17.9 (256 x 64)
    return (n <= 2048 || n > 3072) ? 512 : 256;
}

int nb_small_lanes(int n)
{
    // measured (fp32, us per step at N = 1024 / 2048 / 3000 / 4096): 16 lanes 6.8 / 11.4 / 15.9 / 20.5, 32 lanes
    // 5.4 / 8.5 / 12.5 / 15.5, 64 lanes 4.7 / 7.5 / 10.3 / 15.8 (two-launch path: 8.0 / 9.5 / 11.5 / 16.0);
    // fp64: 64 lanes 4.9 / 7.9 / 11.4 / 17.2, 32 lanes 5.5 / 8.8 / 13.3 / 16.9 (two-launch path: 8.3 / 12.5 / 19.1 / 21.3)
    return 64;       // (32 lanes above N = 3072 with 256-thread workgroups until round 3; 512 x 64 is ahead there now)
}

hipError_t nb_launch_small_step(const void *pos_in, void *pos_out, void *vel, void *acc, const void *mass, int n, int dim,
                                int is_f64, int hook, double G, double eps2, double half_dt, double dt, int do_kick, int lanes,
                                hipStream_t st, const GridTables *tab, double *part, unsigned long long *bin_out)
{
    if (dim != 2 && dim != 3) return hipErrorInvalidValue;
    if (bin_out && hook != HOOK_GRID) return hipErrorInvalidValue;
    if (hook == HOOK_GRID) {
        if (is_f64 || !tab) return hipErrorInvalidValue;
        const float g32 = (float)G, e32 = (float)eps2;
        if (bin_out) {          // bin read-out (nb_quant_bin_sums): the same kernel body, BINS = true
            if (dim == 2) return launch_s<float, 2, HOOK_GRID, true>((const float *)pos_in, (float *)pos_out, (float *)vel, (float *)acc, (const float *)mass, n, g32, e32, half_dt, dt, do_kick, lanes, st, tab, part, bin_out);
            return launch_s<float, 3, HOOK_GRID, true>((const float *)pos_in, (float *)pos_out, (float *)vel, (float *)acc, (const float *)mass, n, g32, e32, half_dt, dt, do_kick, lanes, st, tab, part, bin_out);
        }
        if (dim == 2) return launch_s<float, 2, HOOK_GRID>((const float *)pos_in, (float *)pos_out, (float *)vel, (float *)acc, (const float *)mass, n, g32, e32, half_dt, dt, do_kick, lanes, st, tab, part);
        return launch_s<float, 3, HOOK_GRID>((const float *)pos_in, (float *)pos_out, (float *)vel, (float *)acc, (const float *)mass, n, g32, e32, half_dt, dt, do_kick, lanes, st, tab, part);
    }
    if (is_f64) {
        if (dim == 2) return launch_s<double, 2, HOOK_NONE>((const double *)pos_in, (double *)pos_out, (double *)vel, (double *)acc, (const double *)mass, n, G, eps2, half_dt, dt, do_kick, lanes, st);
        return launch_s<double, 3, HOOK_NONE>((const double *)pos_in, (double *)pos_out, (double *)vel, (double *)acc, (const double *)mass, n, G, eps2, half_dt, dt, do_kick, lanes, st);
    }
#define NB_SF(DD, HH) launch_s<float, DD, HH>((const float *)pos_in, (float *)pos_out, (float *)vel, (float *)acc, (const float *)mass, n, (double)(float)G, (double)(float)eps2, half_dt, dt, do_kick, lanes, st)
    if (dim == 2) {
        if (hook == HOOK_BF16) return NB_SF(2, HOOK_BF16);
        if (hook == HOOK_F16) return NB_SF(2, HOOK_F16);
        return NB_SF(2, HOOK_NONE);
    }
    if (hook == HOOK_BF16) return NB_SF(3, HOOK_BF16);
    if (hook == HOOK_F16) return NB_SF(3, HOOK_F16);
    return NB_SF(3, HOOK_NONE);
#undef NB_SF
}
