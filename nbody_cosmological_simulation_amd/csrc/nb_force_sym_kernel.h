// nb_force_sym_kernel.h -- device templates of the pair-symmetric force kernel (see nb_force_sym.hip for the scheme):
// the sweeps of one target tile against one source tile (scalar and packed fp32), force_sym_kernel itself and its
// launch templates.  Included by nb_force_sym.hip (production instantiations) and nb_force_sym_bins.hip (the SAME
// kernel bodies instantiated with BINS = true: every pair's quant-bin decision is read out of the pair loop that
// makes it -- table-free estimate, wave ballot, threshold fallback -- as per-particle integer checksums).
#pragma once
#include "nb_device.h"
#ifdef NB_GRID_R2_EXACT
#define NB_GRID_R2_EXACT_V 1
#else
#define NB_GRID_R2_EXACT_V 0
#endif
// Table-free pair path, force factor of a pair whose bin the estimate settled: 0 = v_exp_f32((k - kc) c1 + c0c) (round 2:
// no LDS access at all), 1 = the table entry lut[k] from LDS (one ds_read_b32 on the LDS pipe instead of a packed fma and
// a quarter-rate v_exp_f32 on the VALU port; the factor is then the exact table value).
#ifndef NB_GRID_FACTOR_LUT
#define NB_GRID_FACTOR_LUT 0
#endif
// General-mass grid kernel, 2-D, R = 4: 0 = scalar sweep (round 2: 1.065 ms per INT8 step at N = 65 536), 1 = packed sweep
// over the whole source tile (37 VGPRs spilled: 1.085 ms), 2 = packed sweep over the source tile in two halves (116 VGPRs,
// no spill: 0.905 ms) -- the default (profiles/r03_grid_step_timing.txt).  Bins identical in all three.
#ifndef NB_GRID_GENERAL_PACKED
#define NB_GRID_GENERAL_PACKED 2
#endif

#include <hip/hip_ext.h>

#include <type_traits>

namespace {

using namespace nbdev;

template <typename T>
__device__ __forceinline__ T rot1(T v, int addr);

template <>
__device__ __forceinline__ double rot1<double>(double v, int addr)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_bpermute(addr, (int)(b & 0xffffffffll));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, (int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <>
__device__ __forceinline__ float rot1<float>(float v, int addr)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v)));
}

// q^(-3/2) without any mass factor.
// fp64: v_rsq_f64 seed (2^-24) + second-order correction, see inv_r3_f64 in nb_force.hip.
//       c15 / c1875 hold 1.5 and 1.875 in registers chosen by the caller: as literals the
//       compiler re-materialises 1.5 with two v_mov per pair (v_fmac needs it in the destination).
__device__ __forceinline__ double inv_r3_sym(double q, double c15, double c1875)
{
#ifdef NB_EXP_RSQ_F32
    const double y0 = (double)__builtin_amdgcn_rsqf((float)q);   // experiment: fp32 seed via two converts
#else
    const double y0 = __builtin_amdgcn_rsq(q);
#endif
    const double y02 = y0 * y0;
    const double e = __builtin_fma(-q, y02, 1.0);
    const double v = y0 * y02;
    const double c = __builtin_fma(e, c1875, c15);
    const double ce = c * e;
    return __builtin_fma(v, ce, v);
}
// fp32: v_rsq_f32 seed (1 ulp) + first-order correction of the cube: y0^3 (1 + 1.5 e), residual
//       ~2 e^2 < 1e-13, leaving only the ~1.5 ulp of the three roundings.
__device__ __forceinline__ float inv_r3_sym(float q, float c15, float)
{
    const float y0 = __builtin_amdgcn_rsqf(q);
    const float y02 = y0 * y0;
    const float e = __builtin_fmaf(-q, y02, 1.0f);
    const float v = y0 * y02;
    const float ve = v * e;
    return __builtin_fmaf(ve, c15, v);
}

// Source slots per sweep: the whole tile (R) in 2-D; in 3-D four targets and four sources per lane do not fit
// 128 VGPRs, so the source tile is swept in two halves of two slots (8 pairs per lane per rotation step
// instead of the 4 of an R = 2 tiling).
constexpr int sym_rj(int d, int r) { return (d == 3 && r == 4) ? 2 : r; }

struct GridArgs {
    const float *thr, *lut;
    float est_a, est_b, gfac;
    int lmax_bin;
    int lp;                 // table size (power of two) for the binary-search fallback
    // table-free path (GridTables::fast_ok): bin - kc = rint(centred estimate) when the estimate is further than
    // sure_lim from a bin edge, scaled force factor = v_exp_f32((bin - kc) c1 + c0c)
    float est_bc, sure_lim, c1, c0c, kcf;
};
// grid variants of the sweeps (template parameter EST): how a pair finds its force factor
enum { GRID_SEARCH = 0,   // binary search over the thresholds (estimate unusable: very narrow grids)
       GRID_EST = 2,      // floor(estimate) + one threshold compare + table value: two dependent LDS reads
       GRID_FAST = 3,     // table-free (see GridArgs); falls back to GRID_EST for a wave with a pair near a bin edge
       GRID_FAST_CLAMP = 4,   // ... with softening^2 below the grid's floor 0.01: estimates below bin 0 clamp to it
       GRID_DEGENERATE = 5 }; // lmax - lmin < 1e-10: clamped values pass through (quantization.py:115-116)

// ---- bin read-out (BINS = true instantiations only: nb_force_sym_bins.hip) --------------------------------------
// Per-particle integer checksums of the quant-bin index k (quantization.py:119-121) of every pair a particle takes
// part in:   s1[p] = sum_q k(p, q),   s2[p] = sum_q k(p, q) * ((q mod 65521) + 1)      (exact in int64, order-free)
// accumulated INSIDE the pair loop, at the point where the production code decides the bin (table-free estimate,
// wave ballot, threshold fallback -- the control flow is the production one, the integers ride along).  The
// symmetric sweep adds the shared k to both particles; the source-side sums rotate with the source data exactly
// like the force accumulators do.  Padding particles (index >= n) take part in no pair.
template <int R, int RJ, bool ON> struct BinDbg {};
template <int R, int RJ> struct BinDbg<R, RJ, true> {
    long long i1[R], i2[R], j1[RJ], j2[RJ];
    int ip[R];               // particle index of target slot r (this lane)
    int jp[RJ];              // particle index of the source held in slot r right now (rotates with the J data)
    int n;
    long long fast_pairs;    // pairs whose bin came from the table-free estimate alone
    long long exact_pairs;   // pairs whose bin came from a threshold table (fallback wave, GRID_EST, GRID_SEARCH)
    template <bool DIAG> __device__ __forceinline__ void add(int ri, int rj, int k, bool exact)
    {
        const bool real = ip[ri] < n && jp[rj] < n;
        const long long kk = real ? (long long)k : 0ll;
        i1[ri] += kk;
        i2[ri] += kk * (long long)(jp[rj] % 65521 + 1);
        if (!DIAG) {
            j1[rj] += kk;
            j2[rj] += kk * (long long)(ip[ri] % 65521 + 1);
        }
        if (real) { if (exact) ++exact_pairs; else ++fast_pairs; }
    }
    __device__ __forceinline__ void rotate(int rj, int rot_addr, bool diag)
    {
        jp[rj] = __builtin_amdgcn_ds_bpermute(rot_addr, jp[rj]);
        if (!diag) {
            j1[rj] = __double_as_longlong(rot1<double>(__longlong_as_double(j1[rj]), rot_addr));
            j2[rj] = __double_as_longlong(rot1<double>(__longlong_as_double(j2[rj]), rot_addr));
        }
    }
};

// One sweep of a target tile (R particles per lane) against RJ source slots: `nsteps` rotation steps (64 = all
// lanes), R*RJ pairs per lane per step, J data rotating by one lane per step.  RJ = R covers the whole
// source tile; D = 3 sweeps it in two halves (RJ = 2) to stay inside 128 VGPRs with R = 4 targets.
// DIAG:    J is the target tile itself -> one-sided (each ordered pair once, mirrors dropped).
// UNIFORM: all masses equal -> the mass factor is applied once to the finished sums
//          (reduce_sym_kernel), saving both mass multiplies and the rotation of the masses.
// HOOK:    precision hook applied to the fp32 r2 (HOOK_NONE for fp64); EST: grid bins by estimate.
template <typename T, int D, int R, int RJ, bool DIAG, bool UNIFORM, int HOOK, int EST, bool BINS = false>
__device__ __forceinline__ void sweep(const T (&xi)[R][D], const T (&gi)[R], T (&ai)[R][D], T (&xj)[RJ][D],
                                      T (&gj)[RJ], T (&aj)[RJ][D], T eps2, int rot_addr, const GridArgs &ga,
                                      int nsteps, BinDbg<R, RJ, BINS> &bd)
{
    T c15 = (T)1.5, c1875 = (T)1.875;
    if constexpr (std::is_same_v<T, double>) asm volatile("" : "+v"(c15), "+s"(c1875));
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
        // source slot outermost: once all targets have met source rj its data and accumulator are
        // final for this step, so their rotation is issued at once and overlaps the remaining slots
#pragma unroll
        for (int rj = 0; rj < RJ; ++rj) {
#pragma unroll
            for (int ri = 0; ri < R; ++ri) {
                T d[D];
                T w;
                if constexpr (std::is_same_v<T, double>) {
                    double q;
                    if (HOOK == HOOK_F32PAIR) {
                        // FLOAT64 mode on fp32-typed positions (first evaluation): diff and
                        // r2 in fp32 in the reference's op order, everything after the hook in fp64
                        float df[D];
#pragma unroll
                        for (int k = 0; k < D; ++k) {
                            df[k] = __fsub_rn((float)xj[rj][k], (float)xi[ri][k]);
                            d[k] = (double)df[k];
                        }
                        q = (double)r2_f32_exact<D>(df, ga.gfac);       // gfac carries eps2 as fp32 here
                    } else {
#pragma unroll
                        for (int k = 0; k < D; ++k) d[k] = xj[rj][k] - xi[ri][k];
                        q = __builtin_fma(d[D - 1], d[D - 1], eps2);
#pragma unroll
                        for (int k = D - 2; k >= 0; --k) q = __builtin_fma(d[k], d[k], q);
                    }
                    w = inv_r3_sym(q, c15, c1875);
                } else {
                    // reference op order, one rounding per op, no FMA (bit-identical r2)
#pragma unroll
                    for (int k = 0; k < D; ++k) d[k] = __fsub_rn(xj[rj][k], xi[ri][k]);
                    float r2;
#ifndef NB_GRID_R2_EXACT
                    if (HOOK == HOOK_GRID && (EST == GRID_FAST || EST == GRID_FAST_CLAMP)) {
                        // the ESTIMATE may take r2 from fused multiply-adds (D ops instead of 2 D; within 4 ulp of the
                        // reference's r2 = 7e-7 * est_a bins, which sure_lim allows for): a pair far from every bin edge
                        // has the same bin either way, and the rare wave with a pair near an edge recomputes the
                        // reference's r2 bit for bit before it consults the thresholds
                        r2 = __builtin_fmaf(d[D - 1], d[D - 1], eps2);
#pragma unroll
                        for (int k = D - 2; k >= 0; --k) r2 = __builtin_fmaf(d[k], d[k], r2);
                    } else
#endif
                    {
                        r2 = __fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1]));
                        if (D == 3) r2 = __fadd_rn(r2, __fmul_rn(d[2], d[2]));
                        r2 = __fadd_rn(r2, eps2);
                    }
                    if (HOOK == HOOK_GRID) {
                        int kbin = 0;              // BINS: the bin this pair was given, and by which route
                        bool kexact = true;
                        if (EST == GRID_DEGENERATE) {
                            w = inv_r3_sym((r2 < 0.01f) ? 0.01f : r2, c15, c1875) * ga.gfac;
                        } else if (EST == GRID_FAST || EST == GRID_FAST_CLAMP) {
                            // table-free unless one of the wave's 64 pairs sits near a bin edge (see sweep_pk)
                            const float ne = __builtin_fmaf(__builtin_amdgcn_logf(r2), ga.est_a, ga.est_bc);
                            float kf = __builtin_rintf(ne);
                            const float dev = __builtin_fabsf(ne - kf);
                            if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(dev <= ga.sure_lim)) != 0ull, 0)) {
                                float r2e = __fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1]));
                                if (D == 3) r2e = __fadd_rn(r2e, __fmul_rn(d[2], d[2]));
                                r2e = __fadd_rn(r2e, eps2);
                                const int kb = grid_bin_floor_estimate(ga.thr, r2e, ga.est_a, ga.est_b, ga.lmax_bin - 1);
                                w = ga.lut[kb];
                                if constexpr (BINS) kbin = kb;
                            } else {
                                if (EST == GRID_FAST_CLAMP) kf = __builtin_amdgcn_fmed3f(kf, -ga.kcf, 1e30f);
                                w = __builtin_amdgcn_exp2f(__builtin_fmaf(kf, ga.c1, ga.c0c));
                                if constexpr (BINS) { kbin = (int)__builtin_fminf(kf + ga.kcf, 1e6f); kexact = false; }
                            }
                        } else if (EST == GRID_EST) {
                            const int kb = grid_bin_floor_estimate(ga.thr, r2, ga.est_a, ga.est_b, ga.lmax_bin - 1);
                            w = ga.lut[kb];
                            if constexpr (BINS) kbin = kb;
                        } else {
                            const int kb = grid_bin_lookup(ga.thr, r2, ga.lp);
                            w = ga.lut[kb];   // (1/q^1.5)*G
                            if constexpr (BINS) kbin = kb;
                        }
                        if constexpr (BINS) bd.template add<DIAG>(ri, rj, kbin, kexact);
                    } else {
                        float q = r2;
                        if (HOOK == HOOK_BF16) q = (float)(__bf16)r2;
                        if (HOOK == HOOK_F16) q = (float)(_Float16)r2;
                        w = inv_r3_sym(q, c15, c1875);
                        // fp16 overflow: q = +inf -> pow = inf -> 1/inf = 0 upstream (rsq-based form gives NaN)
                        if (HOOK == HOOK_F16) w = (q == __builtin_inff()) ? 0.0f : w;
                    }
                }
                const T wj = UNIFORM ? w : w * gj[rj];
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    if constexpr (std::is_same_v<T, double>) ai[ri][k] = __builtin_fma(wj, d[k], ai[ri][k]);
                    else ai[ri][k] = __builtin_fmaf(wj, d[k], ai[ri][k]);
                }
                if (!DIAG) {
                    const T wi = UNIFORM ? w : w * gi[ri];
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        if constexpr (std::is_same_v<T, double>) aj[rj][k] = __builtin_fma(-wi, d[k], aj[rj][k]);
                        else aj[rj][k] = __builtin_fmaf(-wi, d[k], aj[rj][k]);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < D; ++k) {
                xj[rj][k] = rot1<T>(xj[rj][k], rot_addr);
                if (!DIAG) aj[rj][k] = rot1<T>(aj[rj][k], rot_addr);
            }
            if (!UNIFORM) gj[rj] = rot1<T>(gj[rj], rot_addr);
            if constexpr (BINS) bd.rotate(rj, rot_addr, DIAG);
        }
    }
}

// fp32 sweep on packed pairs.  A lone wave issues one VALU instruction per ~4.6 cycles, i.e. half the
// fp32 rate, so the scalar fp32 loop needs two ready waves at all times and loses ~25 % to stalls;
// v_pk_{add,mul,fma}_f32 do two lanes' worth per issue.  Source slots (rj, rj+1) ride in the two halves
// of a float2, so d, r2, the correction, the mass factors and both accumulations are packed; only
// v_rsq_f32, the half-precision converts and the grid lookups stay per component.  Arithmetic is
// identical to the scalar form (same operations, same order; -ffp-contract=off keeps r2 unfused).
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 rot1_f2(f2 v, int addr) { return f2{rot1<float>(v.x, addr), rot1<float>(v.y, addr)}; }

template <int D, int R, int RJ, bool DIAG, bool UNIFORM, int HOOK, int EST, bool BINS = false>
__device__ __forceinline__ void sweep_pk(const float (&xi)[R][D], const float (&gi)[R], f2 (&ai2)[R][D],
                                         f2 (&xj2)[RJ / 2][D], f2 (&gj2)[RJ / 2], f2 (&aj2)[RJ / 2][D], float eps2,
                                         int rot_addr, const GridArgs &ga, int nsteps, BinDbg<R, RJ, BINS> &bd)
{
#ifdef NB_F32_CORR
    const f2 c15 = {1.5f, 1.5f}, one = {1.0f, 1.0f};
#endif
    // table-free grid path: the two additive constants of its packed fmas live in VGPR pairs for the whole sweep
    // (a packed op reads at most one SGPR pair: from SGPRs they cost a v_mov_b64 per use, 2 of ~23 VALU ops per unit)
    f2 est_bc2 = {ga.est_bc, ga.est_bc}, c0c2 = {ga.c0c, ga.c0c};
#ifndef NB_GRID_CONST_SGPR
    if (HOOK == HOOK_GRID && (EST == GRID_FAST || EST == GRID_FAST_CLAMP)) asm volatile("" : "+v"(est_bc2), "+v"(c0c2));
#endif
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
#pragma unroll
        for (int h = 0; h < RJ / 2; ++h) {
#pragma unroll
            for (int ri = 0; ri < R; ++ri) {
                f2 d[D];
#pragma unroll
                for (int k = 0; k < D; ++k) d[k] = xj2[h][k] - xi[ri][k];
                f2 r2;
#ifndef NB_F32_R2_EXACT
                if constexpr (HOOK == HOOK_NONE || (HOOK == HOOK_GRID && (EST == GRID_FAST || EST == GRID_FAST_CLAMP) && !NB_GRID_R2_EXACT_V)) {
                    // grid modes, table-free path: only the ESTIMATE uses this r2 (see the scalar sweep above); the
                    // fallback below recomputes the reference's r2.
                    // FLOAT32 mode has no rounding DECISION hanging on r2 (no bins, no half-type cast), so r2 is built with
                    // fused multiply-adds: D packed ops instead of 2 D, and closer to the exact r2 than the reference's
                    // separately rounded sum (summed forces vs exact: rms 1.12e-8 against 1.21e-8 with the unfused form and
                    // 1.51e-8 for the reference's own fp32 arithmetic; tests/tools/f32_accuracy.py).  0.581 -> 0.532 ms per
                    // launch on the same box.  Every hook that DOES round r2 keeps the reference's r2 bit for bit, below.
                    r2 = __builtin_elementwise_fma(d[D - 1], d[D - 1], f2{eps2, eps2});
#pragma unroll
                    for (int k = D - 2; k >= 0; --k) r2 = __builtin_elementwise_fma(d[k], d[k], r2);
                } else
#endif
                {
                    r2 = d[0] * d[0] + d[1] * d[1];         // one rounding per op (contraction is off): the reference's r2, bit for bit
                    if constexpr (D == 3) r2 = r2 + d[2] * d[2];
                    r2 = r2 + eps2;
                }
                f2 w;
                if (HOOK == HOOK_GRID) {
                    int kbx = 0, kby = 0;          // BINS: the bins these two pairs were given, and by which route
                    bool kexact = true;
                    if (EST == GRID_DEGENERATE) {
                        w.x = inv_r3_sym((r2.x < 0.01f) ? 0.01f : r2.x, 1.5f, 0.0f) * ga.gfac;
                        w.y = inv_r3_sym((r2.y < 0.01f) ? 0.01f : r2.y, 1.5f, 0.0f) * ga.gfac;
                    } else if (EST == GRID_FAST || EST == GRID_FAST_CLAMP) {
                        // no table access for a wave whose 128 pairs all sit clear of the bin edges (the common case:
                        // an edge zone is ~1e-4 of a bin wide); the estimate is the one grid_tables_kernel validated
                        const f2 lg = {__builtin_amdgcn_logf(r2.x), __builtin_amdgcn_logf(r2.y)};
                        const f2 ne = __builtin_elementwise_fma(lg, f2{ga.est_a, ga.est_a}, est_bc2);
                        f2 kf = {__builtin_rintf(ne.x), __builtin_rintf(ne.y)};
                        const f2 fr = ne - kf;
                        const float dev = __builtin_fmaxf(__builtin_fabsf(fr.x), __builtin_fabsf(fr.y));
                        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(dev <= ga.sure_lim)) != 0ull, 0)) {   // also taken for NaN
                            f2 r2e = d[0] * d[0] + d[1] * d[1];     // the reference's r2, bit for bit (one rounding per op)
                            if constexpr (D == 3) r2e = r2e + d[2] * d[2];
                            r2e = r2e + eps2;
                            const int k0 = grid_bin_floor_estimate(ga.thr, r2e.x, ga.est_a, ga.est_b, ga.lmax_bin - 1);
                            const int k1 = grid_bin_floor_estimate(ga.thr, r2e.y, ga.est_a, ga.est_b, ga.lmax_bin - 1);
                            w.x = ga.lut[k0];
                            w.y = ga.lut[k1];
                            if constexpr (BINS) { kbx = k0; kby = k1; }
                        } else {
                            if (EST == GRID_FAST_CLAMP)
                                kf = f2{__builtin_amdgcn_fmed3f(kf.x, -ga.kcf, 1e30f), __builtin_amdgcn_fmed3f(kf.y, -ga.kcf, 1e30f)};
#if NB_GRID_FACTOR_LUT
                            // (padding pairs: the estimate of r2 ~ 1e36 lies far above the table -- clamp to the extra
                            // zero-weight entry lut[levels])
                            const int i0 = min((int)(kf.x + ga.kcf), ga.lmax_bin), i1 = min((int)(kf.y + ga.kcf), ga.lmax_bin);
                            w = f2{ga.lut[i0], ga.lut[i1]};
#else
                            const f2 th = __builtin_elementwise_fma(kf, f2{ga.c1, ga.c1}, c0c2);
                            w = f2{__builtin_amdgcn_exp2f(th.x), __builtin_amdgcn_exp2f(th.y)};
#endif
                            if constexpr (BINS) {
                                kbx = (int)__builtin_fminf(kf.x + ga.kcf, 1e6f);
                                kby = (int)__builtin_fminf(kf.y + ga.kcf, 1e6f);
                                kexact = false;
                            }
                        }
                    } else if (EST == GRID_EST) {
                        const int k0 = grid_bin_floor_estimate(ga.thr, r2.x, ga.est_a, ga.est_b, ga.lmax_bin - 1);
                        const int k1 = grid_bin_floor_estimate(ga.thr, r2.y, ga.est_a, ga.est_b, ga.lmax_bin - 1);
                        w.x = ga.lut[k0];
                        w.y = ga.lut[k1];
                        if constexpr (BINS) { kbx = k0; kby = k1; }
                    } else {
                        const int k0 = grid_bin_lookup(ga.thr, r2.x, ga.lp), k1 = grid_bin_lookup(ga.thr, r2.y, ga.lp);
                        w.x = ga.lut[k0];
                        w.y = ga.lut[k1];
                        if (UNIFORM) {      // the search stops at the last real bin when the table is full: padding particles weigh 0
                            w.x = (r2.x >= 1e35f) ? 0.0f : w.x;
                            w.y = (r2.y >= 1e35f) ? 0.0f : w.y;
                        }
                        if constexpr (BINS) { kbx = k0; kby = k1; }
                    }
                    if constexpr (BINS) {
                        bd.template add<DIAG>(ri, 2 * h, kbx, kexact);
                        bd.template add<DIAG>(ri, 2 * h + 1, kby, kexact);
                    }
                } else {
                    if constexpr (HOOK == HOOK_BF16 || HOOK == HOOK_F16) {
                        // BFLOAT16 / FLOAT16 hooks: q carries 8 / 11 significant bits, so the 1-ulp v_rsq_f32 cubed
                        // (<= 3 ulp of fp32, 2e-7) is already four orders of magnitude below the hook's own rounding
                        // of r2 -- no Newton correction.  That also serves fp16 overflow for free: q = +inf gives
                        // v_rsq_f32 = 0, w = 0, which is what upstream's G / inf**1.5 yields (the corrected form would
                        // produce inf * 0).  Packed conversion to the half type (v_cvt_pk_*; round to nearest even).
                        f2 q;
                        if constexpr (HOOK == HOOK_BF16) {
                            typedef __bf16 b2 __attribute__((ext_vector_type(2)));
                            q = __builtin_convertvector(__builtin_convertvector(r2, b2), f2);
                        } else {
                            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                            q = __builtin_convertvector(__builtin_convertvector(r2, h2), f2);
                        }
                        const f2 y0 = {__builtin_amdgcn_rsqf(q.x), __builtin_amdgcn_rsqf(q.y)};
                        w = y0 * (y0 * y0);
                    } else {
                        // FLOAT32 hook: the 1-ulp v_rsq_f32 cubed, no Newton correction (round 2).  Measured against the
                        // exact fp64 forces at N = 30 000 (tests/tools/f32_accuracy.py, relative to the force scale): the
                        // reference's own fp32 arithmetic (its CPU restatement) max 1.05e-7 / rms 1.51e-8; this kernel WITH the
                        // first-order correction 1.26e-7 / 1.61e-8, WITHOUT 1.23e-7 / 1.21e-8 -- the error of the summed
                        // forces is set by the fp32 roundings of the differences, r2 and the products, not by the last
                        // ulp of q^-3/2, so the three packed ops of the correction bought nothing (0.657 -> 0.556 ms per
                        // launch at N = 65 536).  NB_F32_CORR restores it for A/B measurements.
                        const f2 q = r2;
                        const f2 y0 = {__builtin_amdgcn_rsqf(q.x), __builtin_amdgcn_rsqf(q.y)};
                        const f2 y02 = y0 * y0;
#ifdef NB_F32_CORR
                        const f2 e = __builtin_elementwise_fma(-q, y02, one);
                        const f2 v = y0 * y02;
                        const f2 ve = v * e;
                        w = __builtin_elementwise_fma(ve, c15, v);
#else
                        w = y0 * y02;
#endif
                    }
                }
                const f2 wj = UNIFORM ? w : w * gj2[h];
#pragma unroll
                for (int k = 0; k < D; ++k) ai2[ri][k] = __builtin_elementwise_fma(wj, d[k], ai2[ri][k]);
                if (!DIAG) {
                    const f2 wi = UNIFORM ? w : w * gi[ri];
#pragma unroll
                    for (int k = 0; k < D; ++k) aj2[h][k] = __builtin_elementwise_fma(-wi, d[k], aj2[h][k]);
                }
            }
#pragma unroll
            for (int k = 0; k < D; ++k) {
                xj2[h][k] = rot1_f2(xj2[h][k], rot_addr);
                if (!DIAG) aj2[h][k] = rot1_f2(aj2[h][k], rot_addr);
            }
            if (!UNIFORM) gj2[h] = rot1_f2(gj2[h], rot_addr);
            if constexpr (BINS) { bd.rotate(2 * h, rot_addr, DIAG); bd.rotate(2 * h + 1, rot_addr, DIAG); }
        }
    }
}

#ifdef NB_WG_TRACE
// experimental builds only (make exp EXPFLAGS=-DNB_WG_TRACE; tools/wg_trace.py): where and when every workgroup of the
// last force launch ran -- {start, end} of the constant 100 MHz clock, XCC_ID, and per wave HW_ID | (end - start) << 32
constexpr int NB_WG_TRACE_MAX = 16384;
__device__ unsigned long long nb_wg_trace_buf[NB_WG_TRACE_MAX][8];
#endif

// T = double: FLOAT64 mode on fp64 state.  T = float: every fp32-state mode (HOOK selects it).
// packed  [D+1][NP] of T : x, y, (z), mass factor (G*m, or m for HOOK_GRID whose LUT carries G);
//                          padding particles sit far away (see pack_kernel).
// rowslab [slot][D][B] fp64 (one target tile per workgroup slot), colslab [row][D][NP] of T.
// <= 128 VGPRs: four waves per SIMD.  (Five waves -- 96 VGPRs -- were measured too: no gain at any
// shard count, and the general-mass kernel starts to spill.)
// (The general-mass row-split instantiations need 134-140 VGPRs: three waves per SIMD instead of spilling.)
// BINS (nb_force_sym_bins.hip only): the same body with the bin read-out of BinDbg; bin_out = {s1[n], s2[n],
// {table-free pairs, table pairs}} as unsigned 64-bit integers, added with atomics (integer sums: order-free).
template <typename T, int D, int R, bool UNIFORM, int HOOK, int LPC = NB_LUT_MIN, bool BINS = false, bool RSPLIT = false>
__global__ void __launch_bounds__(NB_BLOCK, BINS ? 2 : ((HOOK == HOOK_GRID && LPC > NB_LUT_MIN) || (RSPLIT && !UNIFORM) ? 3 : 4))
force_sym_kernel(const T *__restrict__ packed, const SymWork *__restrict__ work, double *__restrict__ rowslab,
                 T *__restrict__ colslab, int np, T eps2, const GridTables *__restrict__ tab, float gfac, float g_newton,
                 unsigned long long *__restrict__ bin_out, int bin_n)
{
    constexpr int B = 64 * R;
    constexpr bool F32_ = std::is_same_v<T, float>;
    // source slots per sweep; the packed general-mass grid kernel (A/B builds) also sweeps the source tile in two halves
    constexpr int RJ = (NB_GRID_GENERAL_PACKED == 2 && F32_ && HOOK == HOOK_GRID && !UNIFORM && R == 4 && D == 2) ? 2 : sym_rj(D, R);
    constexpr int W = NB_BLOCK / 64;
    constexpr bool F32 = std::is_same_v<T, float>;
    __shared__ T s_aj[W][RJ][D][64];
    // grid hook: threshold and LUT tables with compile-time offsets (a run-time table base costs an address add per
    // lookup: +4..8 % on the INT8 / CUSTOM kernels).  LPC = 256 serves INT8 / INT4 / CUSTOM <= 256, LPC = NB_MAX_LUT
    // the larger CUSTOM grids (33 KB: three workgroups per CU instead of four).
    constexpr int lp = LPC;
    __shared__ float s_thr[HOOK == HOOK_GRID ? LPC + 1 : 1];
    __shared__ float s_lut[HOOK == HOOK_GRID ? LPC + 1 : 1];

#ifdef NB_WG_TRACE
    const unsigned long long trace_t0 = wall_clock64();
    const unsigned long long trace_c0 = __builtin_readcyclecounter();
#endif
    const SymWork wk = work[blockIdx.x];
    // (row-split: the wave index in an SGPR, so that the per-wave step range below is scalar)
    const int wave = RSPLIT ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // Row-split work items (RSPLIT instantiations; fp64, D = 2, R = 4 plans of mid-sized systems, nb_plan.cpp): all four
    // waves hold the SAME target tile and share the item's rotation steps among them, so the workgroup leaves ONE row
    // slot (the waves' row sums are added through LDS in wave order) + one column entry instead of four + one.  A
    // separate instantiation: carried as a run-time flag the extra code cost the headline launch 0.5 % (same-box A/B
    // 1189.2 vs 1183.5 us per step, profiles/r03_rowsplit_sweep.txt).
    static_assert(!RSPLIT || (!std::is_same_v<T, float> && D == 2 && R == 4), "row-split: fp64, 2-D, four targets per lane");
    constexpr bool RS_OK = RSPLIT;
    constexpr bool rowsplit = RSPLIT;
    const int I = wk.tile_i + (rowsplit ? 0 : wave);            // this wave's target tile (wave-uniform)
    const int wk_s_begin = rowsplit ? wk.s_begin + wk.s_count * wave / (NB_BLOCK / 64) : wk.s_begin;
    const int wk_s_count = rowsplit ? wk.s_begin + wk.s_count * (wave + 1) / (NB_BLOCK / 64) - wk_s_begin : wk.s_count;
    const int rot_addr = ((lane + 1) & 63) << 2;
    GridArgs ga{s_thr, s_lut, 0.0f, 0.0f, gfac, 0, lp, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    bool use_est = false, fast = false, degenerate = false;
    float gscale = gfac;      // uniform-mass grid kernel: factor applied to the finished sums (the common mass)
    int mass_exp = 0;
    if (HOOK == HOOK_GRID) {
        const int levels = tab->levels;
        // (round 2 launched the uniform-mass grid kernel and a general-mass twin as a pair and let the tables decide on
        // the device which of the two worked; the uniform kernel now serves every state of the tables itself --
        // table-free, estimate + threshold, binary search, degenerate pass-through -- so there is ONE launch)
        // table-free pairs produce factors scaled by 2^-tm (GridTables): the table copies are scaled alike
        // (exact: a power of two) and the power of two goes back in with the common mass
        fast = tab->fast_ok != 0;
        const int tm = fast ? tab->tm : 0;
        mass_exp = UNIFORM ? 0 : tm;          // general masses: the power of two rides on the mass factors instead
        for (int k = threadIdx.x; k <= lp; k += NB_BLOCK) {
            // binary search pads with +inf; the estimate path needs the NaN sentinel at thr[levels]
            s_thr[k] = (k <= levels) ? tab->thr[k] : __builtin_inff();
            // uniform masses: padding particles (r2 >= 1e36) are caught by one more "bin" of weight 0
            if (UNIFORM && k == levels) s_thr[k] = 1e35f;
            s_lut[k] = (k < levels) ? ldexpf(tab->lut[k], -tm) : 0.0f;
        }
        degenerate = tab->degenerate != 0;
        ga.est_a = tab->est_a;
        ga.est_b = tab->est_b;
        ga.lmax_bin = UNIFORM ? levels : levels - 1;
        use_est = tab->use_est != 0;
        ga.est_bc = tab->est_bc;
        ga.sure_lim = tab->sure_lim;
        ga.c1 = tab->c1;
        ga.c0c = tab->c0c;
        ga.kcf = (float)tab->kc;
        gscale = ldexpf(gfac, tm);
        if (UNIFORM) ga.gfac = g_newton;      // degenerate pass-through evaluates (1 / q^1.5) * G itself; the common mass follows
        __syncthreads();
    }

    T xi[R][D], gi[R];
    // fp64 running sums over the whole chunk (fp32: folded per tile).  The packed uniform-mass grid kernel keeps them
    // in LDS instead of 8 (12 in 3-D) register pairs: they are touched once per 64-step sweep, and with them in
    // registers the table-free sweep does not fit 128 VGPRs (round 2: 25-27 VGPRs spilled to scratch).
    constexpr bool ROW_LDS = F32 && HOOK == HOOK_GRID && R == 4 && LPC == NB_LUT_MIN;
    __shared__ double s_row[ROW_LDS ? W : 1][ROW_LDS ? R * D : 1][ROW_LDS ? 64 : 1];
    double ai_sum[ROW_LDS ? 1 : R][ROW_LDS ? 1 : D];
    auto row_ref = [&](int r, int k) -> double & {
        if constexpr (ROW_LDS) return s_row[wave][r * D + k][lane];
        else return ai_sum[r][k];
    };
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int p = I * B + r * 64 + lane;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            xi[r][k] = packed[(size_t)k * np + p];
            row_ref(r, k) = 0.0;
        }
        gi[r] = UNIFORM ? (T)1 : packed[(size_t)D * np + p];
        if constexpr (F32 && HOOK == HOOK_GRID && !UNIFORM) gi[r] = ldexpf(gi[r], mass_exp);
    }
    BinDbg<R, RJ, BINS> bd;
    if constexpr (BINS) {
        bd.n = bin_n;
        bd.fast_pairs = bd.exact_pairs = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) { bd.ip[r] = I * B + r * 64 + lane; bd.i1[r] = bd.i2[r] = 0; }
    }

    for (int J = wk.jt_begin; J < wk.jt_end; ++J) {     // all four waves take the same source tile
      for (int half = 0; half < R / RJ; ++half) {       // ... in R / RJ parts of RJ source slots each
        const int hs = half * RJ;                       // first source slot of this part
        T aj[RJ][D];
#pragma unroll
        for (int r = 0; r < RJ; ++r)
#pragma unroll
            for (int k = 0; k < D; ++k) aj[r][k] = (T)0;
        if constexpr (BINS) {
#pragma unroll
            for (int r = 0; r < RJ; ++r) {
                bd.jp[r] = J * B + (hs + r) * 64 + ((lane + wk_s_begin) & 63);
                bd.j1[r] = bd.j2[r] = 0;
            }
        }
        if (J >= I) {                                   // wave-uniform; tiles below the diagonal belong to other rows
            const bool diag = (J == I);
            // fp32 modes: source slots (2h, 2h+1) packed in float2 halves (sweep_pk).  The general-mass grid kernel
            // keeps the scalar loop (its packed form needs more than 128 VGPRs: measured 1.50 vs 1.24 ms per launch
            // at N = 65536 with the spills inside the pair loop).
            constexpr bool use_packed = F32 && (RJ % 2 == 0) && (HOOK != HOOK_GRID || UNIFORM || (NB_GRID_GENERAL_PACKED && D == 2 && R == 4));
            if constexpr (use_packed) {
                f2 xj2[RJ / 2][D], gj2[RJ / 2], aj2[RJ / 2][D], ai2[R][D];
#pragma unroll
                for (int h = 0; h < RJ / 2; ++h) {
                    // a split sweep starts s_begin rotation steps in: lane l meets particle (l + s_begin) first
                    const int p0 = J * B + (hs + 2 * h) * 64 + ((lane + wk_s_begin) & 63);
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        xj2[h][k] = f2{packed[(size_t)k * np + p0], packed[(size_t)k * np + p0 + 64]};
                        aj2[h][k] = f2{0.0f, 0.0f};
                    }
                    gj2[h] = UNIFORM ? f2{1.0f, 1.0f} : f2{packed[(size_t)D * np + p0], packed[(size_t)D * np + p0 + 64]};
                    // general-mass grid kernel: the table-free factors are scaled by 2^-tm, the power of two rides on the masses
                    if constexpr (HOOK == HOOK_GRID && !UNIFORM) gj2[h] = f2{ldexpf(gj2[h].x, mass_exp), ldexpf(gj2[h].y, mass_exp)};
                }
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int k = 0; k < D; ++k) ai2[r][k] = f2{0.0f, 0.0f};
#define NB_SWEEP_PK(EE)                                                                                                   \
    do {                                                                                                                 \
        if (diag) sweep_pk<D, R, RJ, true, UNIFORM, HOOK, EE, BINS>(xi, gi, ai2, xj2, gj2, aj2, eps2, rot_addr, ga, wk_s_count, bd); \
        else sweep_pk<D, R, RJ, false, UNIFORM, HOOK, EE, BINS>(xi, gi, ai2, xj2, gj2, aj2, eps2, rot_addr, ga, wk_s_count, bd);     \
    } while (0)
                if (HOOK == HOOK_GRID && fast) {
                    if (eps2 < 0.01f) NB_SWEEP_PK(GRID_FAST_CLAMP);
                    else NB_SWEEP_PK(GRID_FAST);
                } else if (HOOK == HOOK_GRID && degenerate) {
                    NB_SWEEP_PK(GRID_DEGENERATE);
                } else if (HOOK == HOOK_GRID && use_est) {
                    NB_SWEEP_PK(GRID_EST);
                } else if (HOOK == HOOK_GRID) {
                    NB_SWEEP_PK(GRID_SEARCH);        // very narrow grids / NaN bounds: no usable estimate
                } else {
                    NB_SWEEP_PK(0);
                }
#undef NB_SWEEP_PK
#pragma unroll
                for (int h = 0; h < RJ / 2; ++h)
#pragma unroll
                    for (int k = 0; k < D; ++k) { aj[2 * h][k] = aj2[h][k].x; aj[2 * h + 1][k] = aj2[h][k].y; }
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int k = 0; k < D; ++k) row_ref(r, k) += (double)(ai2[r][k].x + ai2[r][k].y);
            } else {
                T xj[RJ][D], gj[RJ], ai[R][D];
#pragma unroll
                for (int r = 0; r < RJ; ++r) {
                    const int p = J * B + (hs + r) * 64 + ((lane + wk_s_begin) & 63);
#pragma unroll
                    for (int k = 0; k < D; ++k) xj[r][k] = packed[(size_t)k * np + p];
                    gj[r] = UNIFORM ? (T)1 : packed[(size_t)D * np + p];
                    if constexpr (F32 && HOOK == HOOK_GRID && !UNIFORM) gj[r] = ldexpf(gj[r], mass_exp);
                }
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int k = 0; k < D; ++k) ai[r][k] = F32 ? (T)0 : (T)row_ref(r, k);
#define NB_SWEEP(EE)                                                                                                  \
    do {                                                                                                                 \
        if (diag) sweep<T, D, R, RJ, true, UNIFORM, HOOK, EE, BINS>(xi, gi, ai, xj, gj, aj, eps2, rot_addr, ga, wk_s_count, bd);     \
        else sweep<T, D, R, RJ, false, UNIFORM, HOOK, EE, BINS>(xi, gi, ai, xj, gj, aj, eps2, rot_addr, ga, wk_s_count, bd);         \
    } while (0)
                if (HOOK == HOOK_GRID && degenerate) NB_SWEEP(GRID_DEGENERATE);
                else if (HOOK == HOOK_GRID && fast && eps2 < (T)0.01) NB_SWEEP(GRID_FAST_CLAMP);
                else if (HOOK == HOOK_GRID && fast) NB_SWEEP(GRID_FAST);
                else if (HOOK == HOOK_GRID && use_est) NB_SWEEP(GRID_EST);
                else NB_SWEEP(GRID_SEARCH);
#undef NB_SWEEP
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int k = 0; k < D; ++k) row_ref(r, k) = F32 ? row_ref(r, k) + (double)ai[r][k] : (double)ai[r][k];
            }
        }
        if constexpr (BINS) {
            // source-side checksums: after the sweep they sit in the lane that holds their particle's index
            if (J > I) {
#pragma unroll
                for (int r = 0; r < RJ; ++r)
                    if (bd.jp[r] < bin_n) {
                        atomicAdd(&bin_out[bd.jp[r]], (unsigned long long)bd.j1[r]);
                        atomicAdd(&bin_out[(size_t)bin_n + bd.jp[r]], (unsigned long long)bd.j2[r]);
                    }
            }
        }
        // column contributions of the super-row to these source slots of tile J: the diagonal sweep leaves aj
        // untouched (0), skipped waves hold 0; add the four waves in a fixed order and write ONE slab entry.
        // After s_count rotations lane l holds the accumulators of particle (l + s_begin + s_count).
        const int home = (lane + wk_s_begin + wk_s_count) & 63;
#pragma unroll
        for (int r = 0; r < RJ; ++r)
#pragma unroll
            for (int k = 0; k < D; ++k) s_aj[wave][r][k][home] = aj[r][k];
        __syncthreads();
        if (J > wk.tile_i) {                            // block-uniform: at least the first row lies below J
            for (int idx = threadIdx.x; idx < RJ * D * 64; idx += NB_BLOCK) {
                const int l = idx & 63, rk = idx >> 6;
                const int r = rk / D, k = rk % D;
                T v = s_aj[0][r][k][l];
#pragma unroll
                for (int w = 1; w < W; ++w) v += s_aj[w][r][k][l];
                if (UNIFORM && HOOK == HOOK_GRID) v *= (T)gscale;    // the common mass (times 2^tm on the table-free path)
                colslab[((size_t)wk.col_ord * D + k) * np + (size_t)J * B + (hs + r) * 64 + l] = v;
            }
        }
        __syncthreads();
      }
    }

    if constexpr (BINS) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (bd.ip[r] < bin_n) {
                atomicAdd(&bin_out[bd.ip[r]], (unsigned long long)bd.i1[r]);
                atomicAdd(&bin_out[(size_t)bin_n + bd.ip[r]], (unsigned long long)bd.i2[r]);
            }
        atomicAdd(&bin_out[2 * (size_t)bin_n], (unsigned long long)bd.fast_pairs);
        atomicAdd(&bin_out[2 * (size_t)bin_n + 1], (unsigned long long)bd.exact_pairs);
    }
    if constexpr (RS_OK) {
        if (rowsplit) {
            // the four waves' sums of the one target tile, added in wave order (s_aj is free after the last barrier of
            // the source loop and has exactly W x R x D x 64 elements of T = double)
            static_assert(!RS_OK || sizeof(s_aj) == sizeof(double) * W * R * D * 64, "row-split reuses the column buffer");
            double *s_rc = reinterpret_cast<double *>(&s_aj[0][0][0][0]);
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int k = 0; k < D; ++k) s_rc[((wave * R + r) * D + k) * 64 + lane] = row_ref(r, k);
            __syncthreads();
            for (int idx = threadIdx.x; idx < R * D * 64; idx += NB_BLOCK) {
                const int l = idx & 63, rk = idx >> 6;         // rk = r * D + k
                double v = s_rc[rk * 64 + l];
#pragma unroll
                for (int w = 1; w < W; ++w) v += s_rc[(w * R * D + rk) * 64 + l];
                rowslab[((size_t)wk.slot * D + (rk % D)) * B + (rk / D) * 64 + l] = v;
            }
            return;
        }
    }
    // row sums: one compact slot per (row, chunk)
    const int slot = wk.slot + wave * wk.slot_stride;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < D; ++k)
            rowslab[((size_t)slot * D + k) * B + r * 64 + lane] =
                (UNIFORM && HOOK == HOOK_GRID) ? row_ref(r, k) * (double)gscale : row_ref(r, k);
#ifdef NB_WG_TRACE
    if (lane == 0 && blockIdx.x < NB_WG_TRACE_MAX)
        nb_wg_trace_buf[blockIdx.x][4 + wave] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |     // HW_REG_HW_ID
                                                ((wall_clock64() - trace_t0) << 32);
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x < NB_WG_TRACE_MAX) {
        nb_wg_trace_buf[blockIdx.x][0] = trace_t0;
        nb_wg_trace_buf[blockIdx.x][1] = wall_clock64();
        nb_wg_trace_buf[blockIdx.x][2] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
        nb_wg_trace_buf[blockIdx.x][3] = __builtin_readcyclecounter() - trace_c0;       // shader clocks
    }
#endif
}

template <typename T, int D, int R, int HOOK, int LPC, bool BINS = false>
hipError_t launch_sym_lpc(const T *packed, const SymWork *work, int nwork, double *rowslab, T *colslab, int np,
                          int uniform, T eps2, const GridTables *tab, float gfac, hipStream_t st, NbKernelEvents ev,
                          float mass_value, unsigned long long *bin_out = nullptr, int bin_n = 0)
{
    // g_newton: G for the uniform-mass grid kernel (its gfac carries the common mass; its tables carry G already)
    if (uniform && HOOK == HOOK_GRID)
        hipExtLaunchKernelGGL((force_sym_kernel<T, D, R, true, HOOK, LPC, BINS>), dim3(nwork), dim3(NB_BLOCK), 0, st, ev.start,
                              ev.stop, 0, packed, work, rowslab, colslab, np, eps2, tab, mass_value, gfac, bin_out, bin_n);
    else if (uniform)
        hipExtLaunchKernelGGL((force_sym_kernel<T, D, R, true, HOOK, LPC, BINS>), dim3(nwork), dim3(NB_BLOCK), 0, st, ev.start,
                              ev.stop, 0, packed, work, rowslab, colslab, np, eps2, tab, gfac, 0.0f, bin_out, bin_n);
    else
        hipExtLaunchKernelGGL((force_sym_kernel<T, D, R, false, HOOK, LPC, BINS>), dim3(nwork), dim3(NB_BLOCK), 0, st, ev.start,
                              ev.stop, 0, packed, work, rowslab, colslab, np, eps2, tab, gfac, 0.0f, bin_out, bin_n);
    return hipGetLastError();
}

template <typename T, int D, int R, int HOOK, bool BINS = false>
hipError_t launch_sym_u(const T *packed, const SymWork *work, int nwork, double *rowslab, T *colslab, int np,
                        int uniform, T eps2, const GridTables *tab, float gfac, hipStream_t st, NbKernelEvents ev,
                        float mass_value = 0.0f, int levels = 0, unsigned long long *bin_out = nullptr, int bin_n = 0)
{
    if constexpr (HOOK == HOOK_GRID) {
        if (levels > NB_LUT_MIN)
            return launch_sym_lpc<T, D, R, HOOK, NB_MAX_LUT, BINS>(packed, work, nwork, rowslab, colslab, np, uniform, eps2,
                                                                   tab, gfac, st, ev, mass_value, bin_out, bin_n);
    }
    return launch_sym_lpc<T, D, R, HOOK, NB_LUT_MIN, BINS>(packed, work, nwork, rowslab, colslab, np, uniform, eps2, tab,
                                                           gfac, st, ev, mass_value, bin_out, bin_n);
}

// row-split instantiations (fp64, D = 2, R = 4, no grid hook): see force_sym_kernel
template <int HOOK>
hipError_t launch_sym_rowsplit(const double *packed, const SymWork *work, int nwork, double *rowslab, double *colslab, int np,
                               int uniform, double eps2, float gfac, hipStream_t st, NbKernelEvents ev)
{
    if (uniform)
        hipExtLaunchKernelGGL((force_sym_kernel<double, 2, 4, true, HOOK, NB_LUT_MIN, false, true>), dim3(nwork), dim3(NB_BLOCK), 0, st,
                              ev.start, ev.stop, 0, packed, work, rowslab, colslab, np, eps2, nullptr, gfac, 0.0f, nullptr, 0);
    else
        hipExtLaunchKernelGGL((force_sym_kernel<double, 2, 4, false, HOOK, NB_LUT_MIN, false, true>), dim3(nwork), dim3(NB_BLOCK), 0, st,
                              ev.start, ev.stop, 0, packed, work, rowslab, colslab, np, eps2, nullptr, gfac, 0.0f, nullptr, 0);
    return hipGetLastError();
}

}  // namespace
