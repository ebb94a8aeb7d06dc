// nb_force_sym.hip -- pair-symmetric all-pairs force kernels for gfx950 (fp64 and fp32 state).
//
// Same mathematics as the one-sided kernels of nb_force.hip (reference simulation.py:74-118 with
// the hooks of quantization.py:21-71), but every UNORDERED pair {a, b} is evaluated once and
// applied to both particles (a_a += m_b w d,  a_b -= m_a w d): r2 and the hook are symmetric in
// (a, b) -- d_ab = -d_ba exactly, also in fp32 without FMA -- so the quantised distance and its
// bin are the same for both directions.  This halves the q^(-3/2) evaluations, the dominant cost
// of this VALU-bound path (DESIGN.md "instruction budget").  The reference treats the summation
// order as free; here it is a different but FIXED order.
//
// Scheme (no LDS staging, no barrier in the pair loop):
//   particles are cut into tiles of B = 64*R; a wavefront keeps one target tile I in registers
//   (R particles per lane) for its whole life and walks source tiles J >= I.  The J tile is held
//   one particle-set per lane too, together with ITS accumulators; after each of 64 steps the J
//   data and the J accumulators rotate by one lane (ds_bpermute_b32: the LDS crossbar, no VALU
//   cycles), so every lane meets every J particle once and the accumulators return home.
//   J == I (diagonal tile) is evaluated one-sided, so each ordered pair is counted once.
//   A workgroup = four waves = four consecutive target tiles (a "super-row") walking the SAME
//   source tiles in step; after each source tile the four waves' column sums are added through
//   LDS in a fixed order, so one column slab per super-row leaves the chip (4x less slab traffic
//   than one per row).  Row sums go to per-(row, chunk) slots.  reduce_sym_kernel adds slots and
//   slabs in a fixed order: no atomics, run-to-run bit-identical.
#include "nb_device.h"
#ifdef NB_GRID_R2_EXACT
#define NB_GRID_R2_EXACT_V 1
#else
#define NB_GRID_R2_EXACT_V 0
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

// One sweep of a target tile (R particles per lane) against RJ source slots: `nsteps` rotation steps (64 = all
// lanes), R*RJ pairs per lane per step, J data rotating by one lane per step.  RJ = R covers the whole
// source tile; D = 3 sweeps it in two halves (RJ = 2) to stay inside 128 VGPRs with R = 4 targets.
// DIAG:    J is the target tile itself -> one-sided (each ordered pair once, mirrors dropped).
// UNIFORM: all masses equal -> the mass factor is applied once to the finished sums
//          (reduce_sym_kernel), saving both mass multiplies and the rotation of the masses.
// HOOK:    precision hook applied to the fp32 r2 (HOOK_NONE for fp64); EST: grid bins by estimate.
template <typename T, int D, int R, int RJ, bool DIAG, bool UNIFORM, int HOOK, int EST>
__device__ __forceinline__ void sweep(const T (&xi)[R][D], const T (&gi)[R], T (&ai)[R][D], T (&xj)[RJ][D],
                                      T (&gj)[RJ], T (&aj)[RJ][D], T eps2, int rot_addr, const GridArgs &ga,
                                      int nsteps)
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
                                w = ga.lut[grid_bin_floor_estimate(ga.thr, r2e, ga.est_a, ga.est_b, ga.lmax_bin - 1)];
                            } else {
                                if (EST == GRID_FAST_CLAMP) kf = __builtin_amdgcn_fmed3f(kf, -ga.kcf, 1e30f);
                                w = __builtin_amdgcn_exp2f(__builtin_fmaf(kf, ga.c1, ga.c0c));
                            }
                        } else if (EST == GRID_EST) {
                            w = ga.lut[grid_bin_floor_estimate(ga.thr, r2, ga.est_a, ga.est_b, ga.lmax_bin - 1)];
                        } else {
                            w = ga.lut[grid_bin_lookup(ga.thr, r2, ga.lp)];   // (1/q^1.5)*G
                        }
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

template <int D, int R, int RJ, bool DIAG, bool UNIFORM, int HOOK, int EST>
__device__ __forceinline__ void sweep_pk(const float (&xi)[R][D], const float (&gi)[R], f2 (&ai2)[R][D],
                                         f2 (&xj2)[RJ / 2][D], f2 (&gj2)[RJ / 2], f2 (&aj2)[RJ / 2][D], float eps2,
                                         int rot_addr, const GridArgs &ga, int nsteps)
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
                            w.x = ga.lut[grid_bin_floor_estimate(ga.thr, r2e.x, ga.est_a, ga.est_b, ga.lmax_bin - 1)];
                            w.y = ga.lut[grid_bin_floor_estimate(ga.thr, r2e.y, ga.est_a, ga.est_b, ga.lmax_bin - 1)];
                        } else {
                            if (EST == GRID_FAST_CLAMP)
                                kf = f2{__builtin_amdgcn_fmed3f(kf.x, -ga.kcf, 1e30f), __builtin_amdgcn_fmed3f(kf.y, -ga.kcf, 1e30f)};
                            const f2 th = __builtin_elementwise_fma(kf, f2{ga.c1, ga.c1}, c0c2);
                            w = f2{__builtin_amdgcn_exp2f(th.x), __builtin_amdgcn_exp2f(th.y)};
                        }
                    } else if (EST == GRID_EST) {
                        w.x = ga.lut[grid_bin_floor_estimate(ga.thr, r2.x, ga.est_a, ga.est_b, ga.lmax_bin - 1)];
                        w.y = ga.lut[grid_bin_floor_estimate(ga.thr, r2.y, ga.est_a, ga.est_b, ga.lmax_bin - 1)];
                    } else {
                        w.x = ga.lut[grid_bin_lookup(ga.thr, r2.x, ga.lp)];
                        w.y = ga.lut[grid_bin_lookup(ga.thr, r2.y, ga.lp)];
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
        }
    }
}

// T = double: FLOAT64 mode on fp64 state.  T = float: every fp32-state mode (HOOK selects it).
// packed  [D+1][NP] of T : x, y, (z), mass factor (G*m, or m for HOOK_GRID whose LUT carries G);
//                          padding particles sit far away (see pack_kernel).
// rowslab [slot][D][B] fp64 (one target tile per workgroup slot), colslab [row][D][NP] of T.
// <= 128 VGPRs: four waves per SIMD.  (Five waves -- 96 VGPRs -- were measured too: no gain at any
// shard count, and the general-mass kernel starts to spill.)
template <typename T, int D, int R, bool UNIFORM, int HOOK, int LPC = NB_LUT_MIN>
__global__ void __launch_bounds__(NB_BLOCK, (HOOK == HOOK_GRID && LPC > NB_LUT_MIN) ? 3 : 4)
force_sym_kernel(const T *__restrict__ packed, const SymWork *__restrict__ work, double *__restrict__ rowslab,
                 T *__restrict__ colslab, int np, T eps2, const GridTables *__restrict__ tab, float gfac, int gate)
{
    constexpr int B = 64 * R;
    constexpr int RJ = sym_rj(D, R);            // source slots per sweep
    constexpr int W = NB_BLOCK / 64;
    constexpr bool F32 = std::is_same_v<T, float>;
    __shared__ T s_aj[W][RJ][D][64];
    // grid hook: threshold and LUT tables with compile-time offsets (a run-time table base costs an address add per
    // lookup: +4..8 % on the INT8 / CUSTOM kernels).  LPC = 256 serves INT8 / INT4 / CUSTOM <= 256, LPC = NB_MAX_LUT
    // the larger CUSTOM grids (33 KB: three workgroups per CU instead of four).
    constexpr int lp = LPC;
    __shared__ float s_thr[HOOK == HOOK_GRID ? LPC + 1 : 1];
    __shared__ float s_lut[HOOK == HOOK_GRID ? LPC + 1 : 1];

    const SymWork wk = work[blockIdx.x];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int I = wk.tile_i + wave;             // this wave's target tile (wave-uniform)
    const int rot_addr = ((lane + 1) & 63) << 2;
    GridArgs ga{s_thr, s_lut, 0.0f, 0.0f, gfac, 0, lp, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    bool use_est = false, fast = false, degenerate = false;
    float gscale = gfac;      // uniform-mass grid kernel: factor applied to the finished sums (the common mass)
    int mass_exp = 0;
    if (HOOK == HOOK_GRID) {
        const int levels = tab->levels;
        // uniform-mass grid kernel (gate 1) and its general-mass stand-in (gate 2) are launched as a
        // pair; the tables decide on the device which of the two does the work
        const int ok = tab->uniform_ok;
        if ((gate == 1 && !ok) || (gate == 2 && ok)) return;
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
        __syncthreads();
    }

    T xi[R][D], gi[R];
    double ai_sum[R][D];      // fp64 running sums over the whole chunk (fp32: folded per tile)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int p = I * B + r * 64 + lane;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            xi[r][k] = packed[(size_t)k * np + p];
            ai_sum[r][k] = 0.0;
        }
        gi[r] = UNIFORM ? (T)1 : packed[(size_t)D * np + p];
        if constexpr (F32 && HOOK == HOOK_GRID && !UNIFORM) gi[r] = ldexpf(gi[r], mass_exp);
    }

    for (int J = wk.jt_begin; J < wk.jt_end; ++J) {     // all four waves take the same source tile
      for (int half = 0; half < R / RJ; ++half) {       // ... in R / RJ parts of RJ source slots each
        const int hs = half * RJ;                       // first source slot of this part
        T aj[RJ][D];
#pragma unroll
        for (int r = 0; r < RJ; ++r)
#pragma unroll
            for (int k = 0; k < D; ++k) aj[r][k] = (T)0;
        if (J >= I) {                                   // wave-uniform; tiles below the diagonal belong to other rows
            const bool diag = (J == I);
            // fp32 modes: source slots (2h, 2h+1) packed in float2 halves (sweep_pk).  The general-mass grid kernel
            // keeps the scalar loop (its packed form needs more than 128 VGPRs: measured 1.50 vs 1.24 ms per launch
            // at N = 65536 with the spills inside the pair loop); the uniform-mass one always has a usable estimate
            // (GridTables::uniform_ok gates it).
            constexpr bool use_packed = F32 && (RJ % 2 == 0) && (HOOK != HOOK_GRID || UNIFORM);
            if constexpr (use_packed) {
                f2 xj2[RJ / 2][D], gj2[RJ / 2], aj2[RJ / 2][D], ai2[R][D];
#pragma unroll
                for (int h = 0; h < RJ / 2; ++h) {
                    // a split sweep starts s_begin rotation steps in: lane l meets particle (l + s_begin) first
                    const int p0 = J * B + (hs + 2 * h) * 64 + ((lane + wk.s_begin) & 63);
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        xj2[h][k] = f2{packed[(size_t)k * np + p0], packed[(size_t)k * np + p0 + 64]};
                        aj2[h][k] = f2{0.0f, 0.0f};
                    }
                    gj2[h] = UNIFORM ? f2{1.0f, 1.0f} : f2{packed[(size_t)D * np + p0], packed[(size_t)D * np + p0 + 64]};
                }
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int k = 0; k < D; ++k) ai2[r][k] = f2{0.0f, 0.0f};
#define NB_SWEEP_PK(EE)                                                                                                   \
    do {                                                                                                                 \
        if (diag) sweep_pk<D, R, RJ, true, UNIFORM, HOOK, EE>(xi, gi, ai2, xj2, gj2, aj2, eps2, rot_addr, ga, wk.s_count); \
        else sweep_pk<D, R, RJ, false, UNIFORM, HOOK, EE>(xi, gi, ai2, xj2, gj2, aj2, eps2, rot_addr, ga, wk.s_count);     \
    } while (0)
                if (HOOK == HOOK_GRID && fast) {
                    if (eps2 < 0.01f) NB_SWEEP_PK(GRID_FAST_CLAMP);
                    else NB_SWEEP_PK(GRID_FAST);
                } else if (HOOK == HOOK_GRID) {
                    NB_SWEEP_PK(GRID_EST);           // only reached with a usable estimate
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
                    for (int k = 0; k < D; ++k) ai_sum[r][k] += (double)(ai2[r][k].x + ai2[r][k].y);
            } else {
                T xj[RJ][D], gj[RJ], ai[R][D];
#pragma unroll
                for (int r = 0; r < RJ; ++r) {
                    const int p = J * B + (hs + r) * 64 + ((lane + wk.s_begin) & 63);
#pragma unroll
                    for (int k = 0; k < D; ++k) xj[r][k] = packed[(size_t)k * np + p];
                    gj[r] = UNIFORM ? (T)1 : packed[(size_t)D * np + p];
                    if constexpr (F32 && HOOK == HOOK_GRID && !UNIFORM) gj[r] = ldexpf(gj[r], mass_exp);
                }
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int k = 0; k < D; ++k) ai[r][k] = F32 ? (T)0 : (T)ai_sum[r][k];
#define NB_SWEEP(EE)                                                                                                  \
    do {                                                                                                                 \
        if (diag) sweep<T, D, R, RJ, true, UNIFORM, HOOK, EE>(xi, gi, ai, xj, gj, aj, eps2, rot_addr, ga, wk.s_count);     \
        else sweep<T, D, R, RJ, false, UNIFORM, HOOK, EE>(xi, gi, ai, xj, gj, aj, eps2, rot_addr, ga, wk.s_count);         \
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
                    for (int k = 0; k < D; ++k) ai_sum[r][k] = F32 ? ai_sum[r][k] + (double)ai[r][k] : (double)ai[r][k];
            }
        }
        // column contributions of the super-row to these source slots of tile J: the diagonal sweep leaves aj
        // untouched (0), skipped waves hold 0; add the four waves in a fixed order and write ONE slab entry.
        // After s_count rotations lane l holds the accumulators of particle (l + s_begin + s_count).
        const int home = (lane + wk.s_begin + wk.s_count) & 63;
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

    // row sums: one compact slot per (row, chunk)
    const int slot = wk.slot + wave * wk.slot_stride;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < D; ++k)
            rowslab[((size_t)slot * D + k) * B + r * 64 + lane] =
                (UNIFORM && HOOK == HOOK_GRID) ? ai_sum[r][k] * (double)gscale : ai_sum[r][k];
}

// Padding particles sit at `pad` in every coordinate (chosen by nb_launch_pack): r^2 stays finite and
// y0^3 underflows to exactly 0 (fp64: r2 ~ 1e300, y0^3 ~ 1e-450; fp32: r2 ~ 1e36, y0^3 ~ 1e-54), so
// they contribute exactly nothing whatever their mass; in grid modes they fall into the last bin and
// carry mass 0.  (fp32 pair arithmetic on fp64 storage: pad = 1e18, contributions ~1e-37, absorbed.)
template <typename T> __device__ __forceinline__ T axpy_rn(T a, T b, T s);
template <> __device__ __forceinline__ double axpy_rn<double>(double a, double b, double s) { return __dadd_rn(a, __dmul_rn(b, s)); }
template <> __device__ __forceinline__ float axpy_rn<float>(float a, float b, float s) { return __fadd_rn(a, __fmul_rn(b, s)); }

// x, y, (z), mass factor as padded component arrays; with KICK the opening half of a step rides
// along: v += a*(dt/2); x += v*dt (simulation.py:132,135, separate mul/add roundings like torch).
// KICK = 2 additionally applies the closing half kick of the PREVIOUS step first (simulation.py:141),
// which the multi-GPU / force-quantising paths cannot fuse into their reduction.
template <typename T, int D, int KICK>
__global__ void __launch_bounds__(NB_BLOCK)
pack_kernel(T *__restrict__ pos, T *__restrict__ vel, const T *__restrict__ acc, const T *__restrict__ mass,
            T *__restrict__ packed, int n, int np, T half_dt, T dt, T gfac, T pad, int p_begin, int p_end)
{
    const int p = p_begin + blockIdx.x * NB_BLOCK + threadIdx.x;
    if (p >= p_end) return;
    if (p < n) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const size_t idx = (size_t)p * D + k;
            T x = pos[idx];
            if (KICK) {
                T v = vel[idx];
                if (KICK == 2) v = axpy_rn<T>(v, acc[idx], half_dt);
                v = axpy_rn<T>(v, acc[idx], half_dt);
                x = axpy_rn<T>(x, v, dt);
                vel[idx] = v;
                pos[idx] = x;
            }
            packed[(size_t)k * np + p] = x;
        }
        packed[(size_t)D * np + p] = gfac * mass[p];
    } else {
#pragma unroll
        for (int k = 0; k < D; ++k) packed[(size_t)k * np + p] = pad;
        packed[(size_t)D * np + p] = (T)0;
    }
}

// acc[p] = sum of the row slots of p's tile + sum of the column-slab entries of the owned super-rows at or
// above tile(p) (a prefix [0, col_upto[J]) of the slab index space), always in the same order.
// Block = 64 particles x 16 waves: the items (row slots, then slab entries) are dealt over the waves
// (item c to wave c mod 16, four independent loads in flight per lane), the 16 partial sums are
// combined in wave order through LDS -> one fixed summation tree.  Optionally fuses the closing half
// kick (simulation.py:141) and applies the uniform-mass factor.
constexpr int NB_RED_WAVES = 16;
template <typename T, int D>
__global__ void __launch_bounds__(64 * NB_RED_WAVES)
reduce_sym_kernel(const double *__restrict__ rowslab, const T *__restrict__ colslab,
                  const int *__restrict__ row_slot0, const int *__restrict__ row_nslots,
                  const int *__restrict__ col_upto, int tile_b, int n, int np,
                  double scale, T *__restrict__ acc, T *__restrict__ vel, T half_dt, int do_kick,
                  T *__restrict__ pos, T *__restrict__ packed, T dt, int blk0, double *__restrict__ sums64,
                  double *__restrict__ mm_part)
{
    __shared__ double s_part[NB_RED_WAVES][D][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int blk = blk0 + blockIdx.x;              // blocks [blk0, blk0 + gridDim.x): one pipeline chunk's tiles
    const int p = blk * 64 + lane;
    const int pc = p < n ? p : n - 1;
    const int J = (blk * 64) / tile_b;              // block-uniform: tile_b is a multiple of 64
    const int s0 = row_slot0[J], ns = row_nslots[J], total = ns + col_upto[J];
    const int off = blk * 64 + lane - J * tile_b;
    double s[D];
#pragma unroll
    for (int k = 0; k < D; ++k) s[k] = 0.0;
#ifndef NB_RED_UNROLL
#define NB_RED_UNROLL 4
#endif
#pragma unroll NB_RED_UNROLL
    for (int c = g; c < total; c += NB_RED_WAVES) {
        if (c < ns) {                               // wave-uniform
#pragma unroll
            for (int k = 0; k < D; ++k) s[k] += rowslab[((size_t)(s0 + c) * D + k) * tile_b + off];
        } else {
#pragma unroll
            for (int k = 0; k < D; ++k) s[k] += (double)colslab[((size_t)(c - ns) * D + k) * np + pc];
        }
    }
#pragma unroll
    for (int k = 0; k < D; ++k) s_part[g][k][lane] = s[k];
    __syncthreads();
    // INT8 / INT4 on one GPU: the min / max of the finished forces for quantize_force, one pair per workgroup
    // (NaN-propagating like torch.min / max), so the step needs no reduction launch of its own
    double mm_lo = __builtin_inf(), mm_hi = -__builtin_inf();
    if (g == 0 && p < n) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            double t = s_part[0][k][lane];
#pragma unroll
            for (int w = 1; w < NB_RED_WAVES; ++w) t += s_part[w][k][lane];
            const size_t idx = (size_t)p * D + k;
            if (sums64) {                          // multi-GPU, fp32 state: the ranks exchange the unrounded fp64 sums
                sums64[idx] = t;
                continue;
            }
            const T a = (T)(t * scale);            // scale = mass factor of the uniform kernel, else 1
            acc[idx] = a;
            if (mm_part) {
                const double ad = (double)a;
                mm_lo = (ad != ad || mm_lo != mm_lo) ? __builtin_nan("") : (ad < mm_lo ? ad : mm_lo);
                mm_hi = (ad != ad || mm_hi != mm_hi) ? __builtin_nan("") : (ad > mm_hi ? ad : mm_hi);
            }
            if (do_kick == 1) {
                vel[idx] = axpy_rn<T>(vel[idx], a, half_dt);
            } else if (do_kick == 2) {
                // closing kick, then the next step's opening kick + drift and its repack (what pack_kernel<KICK=1>
                // would do in a launch of its own; mass factors and padding in `packed` do not change)
                T v = axpy_rn<T>(vel[idx], a, half_dt);
                v = axpy_rn<T>(v, a, half_dt);
                const T x = axpy_rn<T>(pos[idx], v, dt);
                vel[idx] = v;
                pos[idx] = x;
                packed[(size_t)k * np + p] = x;
            }
        }
    }
    if (mm_part && g == 0) {               // wave-uniform
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double ol = __shfl_xor(mm_lo, off, 64), oh = __shfl_xor(mm_hi, off, 64);
            mm_lo = (ol != ol || mm_lo != mm_lo) ? __builtin_nan("") : (ol < mm_lo ? ol : mm_lo);
            mm_hi = (oh != oh || mm_hi != mm_hi) ? __builtin_nan("") : (oh > mm_hi ? oh : mm_hi);
        }
        if (lane == 0) { mm_part[2 * blockIdx.x] = mm_lo; mm_part[2 * blockIdx.x + 1] = mm_hi; }
    }
}

__global__ void __launch_bounds__(NB_BLOCK)
finish_sums64_kernel(const double *__restrict__ sums64, double scale, float *__restrict__ acc, float *__restrict__ vel,
                     float *__restrict__ pos, float *__restrict__ packed, long long count, int np, int dim, int mode,
                     float half_dt, float dt)
{
    const long long e = (long long)blockIdx.x * NB_BLOCK + threadIdx.x;
    if (e >= count) return;
    const float a = (float)(sums64[e] * scale);
    acc[e] = a;
    if (mode == 1) {
        vel[e] = axpy_rn<float>(vel[e], a, half_dt);
    } else if (mode == 2) {
        float v = axpy_rn<float>(vel[e], a, half_dt);
        v = axpy_rn<float>(v, a, half_dt);
        const float x = axpy_rn<float>(pos[e], v, dt);
        vel[e] = v;
        pos[e] = x;
        if (packed) packed[(size_t)(e % dim) * np + e / dim] = x;
    }
}

// Potential energy over the same tile-pair work list: sum over unordered pairs of
// m_a m_b / sqrt(r2 + eps2) (simulation.py:176-192; the caller applies -G).  Off-diagonal tile
// pairs hold every unordered pair once; a diagonal tile holds each twice plus the self pairs, so
// those terms carry weight 1/2 and the self pairs are skipped.  F32T: term arithmetic in fp32 with
// correctly rounded sqrt / divide (fp32-typed state); otherwise fp64 with v_rsq_f64 + a third-order
// step.  Per-lane fp64 sums, wave shuffle + LDS block sum, one partial per workgroup.
template <typename T, int D, int R, bool F32T>
__global__ void __launch_bounds__(NB_BLOCK)
potential_sym_kernel(const T *__restrict__ packed, const SymWork *__restrict__ work, double *__restrict__ part,
                     int np, double eps2, float eps2_f, int mass_dt)
{
    constexpr int B = 64 * R;
    __shared__ double s_red[NB_BLOCK / 64];
    const SymWork wk = work[blockIdx.x];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int I = wk.tile_i + wave;
    const int rot_addr = ((lane + 1) & 63) << 2;

    T xi[R][D], mi[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int p = I * B + r * 64 + lane;
#pragma unroll
        for (int k = 0; k < D; ++k) xi[r][k] = packed[(size_t)k * np + p];
        mi[r] = packed[(size_t)D * np + p];
    }
    double sum = 0.0;
    for (int J = max(wk.jt_begin, I); J < wk.jt_end; ++J) {
        T xj[R][D], mj[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int p = J * B + r * 64 + ((lane + wk.s_begin) & 63);
#pragma unroll
            for (int k = 0; k < D; ++k) xj[r][k] = packed[(size_t)k * np + p];
            mj[r] = packed[(size_t)D * np + p];
        }
        const bool diag = (J == I);
        double tsum[R];             // independent chains: a single accumulator would serialise the adds
#pragma unroll
        for (int r = 0; r < R; ++r) tsum[r] = 0.0;
#pragma unroll 1
        for (int s = wk.s_begin; s < wk.s_begin + wk.s_count; ++s) {
#pragma unroll
            for (int ri = 0; ri < R; ++ri) {
#pragma unroll
                for (int rj = 0; rj < R; ++rj) {
                    double term;
                    if (F32T) {
                        float d2 = 0.0f;
#pragma unroll
                        for (int k = 0; k < D; ++k) {
                            const float df = __fsub_rn((float)xj[rj][k], (float)xi[ri][k]);
                            const float sq = __fmul_rn(df, df);
                            d2 = (k == 0) ? sq : __fadd_rn(d2, sq);
                        }
#ifdef NB_PE_F32_EXACT
                        const float dist = __builtin_sqrtf(__fadd_rn(d2, eps2_f));
                        term = (double)__fdiv_rn(mass_prod_f32((float)mi[ri], (float)mj[rj], mass_dt), dist);
#else
                        // m_a m_b / sqrt(q) as one v_rsq_f32 and a product (1.5 ulp per term, unbiased) instead of the
                        // correctly rounded sqrt and divide (about 25 instructions): the terms are summed in fp64 here,
                        // while the reference's own fp32 .sum() of N^2 terms carries ~1e-7 of rounding -- the last ulp
                        // of a term is far below what the result can show (measured: DESIGN.md 4.6)
                        const float y = __builtin_amdgcn_rsqf(__fadd_rn(d2, eps2_f));
                        term = (double)__fmul_rn(mass_prod_f32((float)mi[ri], (float)mj[rj], mass_dt), y);
#endif
                    } else {
                        double q = eps2;
#pragma unroll
                        for (int k = 0; k < D; ++k) {
                            const double df = (double)xj[rj][k] - (double)xi[ri][k];
                            q = __builtin_fma(df, df, q);
                        }
                        const double y0 = __builtin_amdgcn_rsq(q);
                        const double e = __builtin_fma(-q * y0, y0, 1.0);
                        const double y = __builtin_fma(y0 * e, __builtin_fma(e, 0.375, 0.5), y0);
                        const double mp = mass_dt != NB_F64 ? (double)mass_prod_f32((float)mi[ri], (float)mj[rj], mass_dt)
                                                            : (double)mi[ri] * (double)mj[rj];
                        term = mp * y;
                    }
                    if (ri == rj) term = (diag && s == 0) ? 0.0 : term;     // self pair
                    tsum[rj] += term;
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int k = 0; k < D; ++k) xj[r][k] = rot1<T>(xj[r][k], rot_addr);
                mj[r] = rot1<T>(mj[r], rot_addr);
            }
        }
        double tile = tsum[0];
#pragma unroll
        for (int r = 1; r < R; ++r) tile += tsum[r];
        sum += diag ? 0.5 * tile : tile;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) s_red[wave] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = s_red[0];
#pragma unroll
        for (int w = 1; w < NB_BLOCK / 64; ++w) t += s_red[w];
        part[blockIdx.x] = t;
    }
}

template <typename T, int D, int R, int HOOK, int LPC>
hipError_t launch_sym_lpc(const T *packed, const SymWork *work, int nwork, double *rowslab, T *colslab, int np,
                          int uniform, T eps2, const GridTables *tab, float gfac, hipStream_t st, NbKernelEvents ev,
                          float mass_value)
{
    if (uniform && HOOK == HOOK_GRID) {
        // uniform-mass grid kernel, valid only while the tables say so (GridTables::uniform_ok, known on the
        // device only): launch it together with the general kernel, exactly one of the two does the work
        hipExtLaunchKernelGGL((force_sym_kernel<T, D, R, true, HOOK, LPC>), dim3(nwork), dim3(NB_BLOCK), 0, st, ev.start,
                              nullptr, 0, packed, work, rowslab, colslab, np, eps2, tab, mass_value, 1);
        hipExtLaunchKernelGGL((force_sym_kernel<T, D, R, false, HOOK, LPC>), dim3(nwork), dim3(NB_BLOCK), 0, st, nullptr,
                              ev.stop, 0, packed, work, rowslab, colslab, np, eps2, tab, gfac, 2);
    } else if (uniform)
        hipExtLaunchKernelGGL((force_sym_kernel<T, D, R, true, HOOK, LPC>), dim3(nwork), dim3(NB_BLOCK), 0, st, ev.start,
                              ev.stop, 0, packed, work, rowslab, colslab, np, eps2, tab, gfac, 0);
    else
        hipExtLaunchKernelGGL((force_sym_kernel<T, D, R, false, HOOK, LPC>), dim3(nwork), dim3(NB_BLOCK), 0, st, ev.start,
                              ev.stop, 0, packed, work, rowslab, colslab, np, eps2, tab, gfac, 0);
    return hipGetLastError();
}

template <typename T, int D, int R, int HOOK>
hipError_t launch_sym_u(const T *packed, const SymWork *work, int nwork, double *rowslab, T *colslab, int np,
                        int uniform, T eps2, const GridTables *tab, float gfac, hipStream_t st, NbKernelEvents ev,
                        float mass_value = 0.0f, int levels = 0)
{
    if constexpr (HOOK == HOOK_GRID) {
        if (levels > NB_LUT_MIN)
            return launch_sym_lpc<T, D, R, HOOK, NB_MAX_LUT>(packed, work, nwork, rowslab, colslab, np, uniform, eps2, tab,
                                                             gfac, st, ev, mass_value);
    }
    return launch_sym_lpc<T, D, R, HOOK, NB_LUT_MIN>(packed, work, nwork, rowslab, colslab, np, uniform, eps2, tab, gfac,
                                                     st, ev, mass_value);
}

}  // namespace

hipError_t nb_launch_pack(void *pos, void *vel, const void *acc, const void *mass, void *packed, int n, int np,
                          int dim, int is_f64, int kick, double half_dt, double dt, double gfac, int f32_pairs,
                          hipStream_t st, int p_begin, int p_end)
{
    if (p_end < 0) p_end = np;
    if (p_end <= p_begin) return hipSuccess;
    // padding coordinate: far enough that r^-3 vanishes, close enough that r2 stays finite in the
    // arithmetic the pair loop uses (fp32 pair arithmetic on fp64 storage needs the fp32 value)
    const double pad = (is_f64 && !f32_pairs) ? 1e150 : 1e18;
    const int grid = (p_end - p_begin + NB_BLOCK - 1) / NB_BLOCK;
#define NB_PACK(TT, DD, KK) \
    hipLaunchKernelGGL((pack_kernel<TT, DD, KK>), dim3(grid), dim3(NB_BLOCK), 0, st, (TT *)pos, (TT *)vel, \
                       (const TT *)acc, (const TT *)mass, (TT *)packed, n, np, (TT)half_dt, (TT)dt, (TT)gfac, (TT)pad, \
                       p_begin, p_end)
#define NB_PACK_K(TT, DD) do { if (kick == 2) NB_PACK(TT, DD, 2); else if (kick == 1) NB_PACK(TT, DD, 1); else NB_PACK(TT, DD, 0); } while (0)
    if (dim != 2 && dim != 3) return hipErrorInvalidValue;
    if (is_f64) { if (dim == 2) NB_PACK_K(double, 2); else NB_PACK_K(double, 3); }
    else        { if (dim == 2) NB_PACK_K(float, 2); else NB_PACK_K(float, 3); }
#undef NB_PACK_K
#undef NB_PACK
    return hipGetLastError();
}

hipError_t nb_launch_force_sym_f64(const double *packed, const SymWork *work, int nwork, double *rowslab,
                                   double *colslab, int np, int dim, int r, int uniform, int pa_f32, double eps2,
                                   hipStream_t st, NbKernelEvents ev)
{
    if (pa_f32) {   // first evaluation on fp32-typed positions: default tile shapes only
        const float e32 = (float)eps2;
        if (dim == 2 && r == 4) return launch_sym_u<double, 2, 4, HOOK_F32PAIR>(packed, work, nwork, rowslab, colslab, np, uniform, eps2, nullptr, e32, st, ev);
        if (dim == 2 && r == 2) return launch_sym_u<double, 2, 2, HOOK_F32PAIR>(packed, work, nwork, rowslab, colslab, np, uniform, eps2, nullptr, e32, st, ev);
        if (dim == 3 && r == 2) return launch_sym_u<double, 3, 2, HOOK_F32PAIR>(packed, work, nwork, rowslab, colslab, np, uniform, eps2, nullptr, e32, st, ev);
        if (dim == 3 && r == 4) return launch_sym_u<double, 3, 4, HOOK_F32PAIR>(packed, work, nwork, rowslab, colslab, np, uniform, eps2, nullptr, e32, st, ev);
        return hipErrorInvalidValue;
    }
    if (dim == 2 && r == 1) return launch_sym_u<double, 2, 1, HOOK_NONE>(packed, work, nwork, rowslab, colslab, np, uniform, eps2, nullptr, 1.0f, st, ev);
    if (dim == 2 && r == 2) return launch_sym_u<double, 2, 2, HOOK_NONE>(packed, work, nwork, rowslab, colslab, np, uniform, eps2, nullptr, 1.0f, st, ev);
    if (dim == 2 && r == 4) return launch_sym_u<double, 2, 4, HOOK_NONE>(packed, work, nwork, rowslab, colslab, np, uniform, eps2, nullptr, 1.0f, st, ev);
    if (dim == 3 && r == 1) return launch_sym_u<double, 3, 1, HOOK_NONE>(packed, work, nwork, rowslab, colslab, np, uniform, eps2, nullptr, 1.0f, st, ev);
    if (dim == 3 && r == 2) return launch_sym_u<double, 3, 2, HOOK_NONE>(packed, work, nwork, rowslab, colslab, np, uniform, eps2, nullptr, 1.0f, st, ev);
    if (dim == 3 && r == 4) return launch_sym_u<double, 3, 4, HOOK_NONE>(packed, work, nwork, rowslab, colslab, np, uniform, eps2, nullptr, 1.0f, st, ev);
    return hipErrorInvalidValue;
}

hipError_t nb_launch_force_sym_f32(const float *packed, const SymWork *work, int nwork, double *rowslab,
                                   float *colslab, int np, int dim, int r, int uniform, int hook, float eps2,
                                   const GridTables *tab, float G, float mass_value, int levels, hipStream_t st,
                                   NbKernelEvents ev)
{
#define NB_SYM32(DD, RR)                                                                                              \
    switch (hook) {                                                                                                   \
    case HOOK_NONE: return launch_sym_u<float, DD, RR, HOOK_NONE>(packed, work, nwork, rowslab, colslab, np, uniform, eps2, tab, G, st, ev); \
    case HOOK_BF16: return launch_sym_u<float, DD, RR, HOOK_BF16>(packed, work, nwork, rowslab, colslab, np, uniform, eps2, tab, G, st, ev); \
    case HOOK_F16: return launch_sym_u<float, DD, RR, HOOK_F16>(packed, work, nwork, rowslab, colslab, np, uniform, eps2, tab, G, st, ev);   \
    case HOOK_GRID: return launch_sym_u<float, DD, RR, HOOK_GRID>(packed, work, nwork, rowslab, colslab, np, uniform, eps2, tab, G, st, ev, mass_value, levels); \
    default: return hipErrorInvalidValue;                                                                             \
    }
    if (dim == 2 && r == 2) { NB_SYM32(2, 2) }
    if (dim == 2 && r == 4) { NB_SYM32(2, 4) }
    if (dim == 3 && r == 2) { NB_SYM32(3, 2) }
    if (dim == 3 && r == 4) { NB_SYM32(3, 4) }
#undef NB_SYM32
    return hipErrorInvalidValue;
}

hipError_t nb_launch_potential_sym(const void *packed, const SymWork *work, int nwork, double *part, int np, int dim,
                                   int r, int is_f64, int f32_terms, int mass_dt, double eps2, hipStream_t st)
{
    const float e32 = (float)eps2;
#define NB_PES(TT, DD, RR, FF) \
    hipLaunchKernelGGL((potential_sym_kernel<TT, DD, RR, FF>), dim3(nwork), dim3(NB_BLOCK), 0, st, (const TT *)packed, \
                       work, part, np, eps2, e32, mass_dt)
    if (is_f64) {
        if (dim == 2 && r == 4) { if (f32_terms) NB_PES(double, 2, 4, true); else NB_PES(double, 2, 4, false); }
        else if (dim == 2 && r == 2) { if (f32_terms) NB_PES(double, 2, 2, true); else NB_PES(double, 2, 2, false); }
        else if (dim == 3 && r == 2) { if (f32_terms) NB_PES(double, 3, 2, true); else NB_PES(double, 3, 2, false); }
        else if (dim == 3 && r == 4) { if (f32_terms) NB_PES(double, 3, 4, true); else NB_PES(double, 3, 4, false); }
        else return hipErrorInvalidValue;
    } else {
        if (dim == 2 && r == 4) NB_PES(float, 2, 4, true);
        else if (dim == 2 && r == 2) NB_PES(float, 2, 2, true);
        else if (dim == 3 && r == 2) NB_PES(float, 3, 2, true);
        else if (dim == 3 && r == 4) NB_PES(float, 3, 4, true);
        else return hipErrorInvalidValue;
    }
#undef NB_PES
    return hipGetLastError();
}

hipError_t nb_launch_finish_sums64(const double *sums64, double scale, float *acc, float *vel, float *pos, float *packed,
                                   int n, int np, int dim, int mode, double half_dt, double dt, hipStream_t st)
{
    const long long count = (long long)n * dim;
    hipLaunchKernelGGL(finish_sums64_kernel, dim3((unsigned)((count + NB_BLOCK - 1) / NB_BLOCK)), dim3(NB_BLOCK), 0, st, sums64,
                       scale, acc, vel, pos, packed, count, np, dim, mode, (float)half_dt, (float)dt);
    return hipGetLastError();
}

hipError_t nb_launch_reduce_sym(const double *rowslab, const void *colslab, const int *row_slot0,
                                const int *row_nslots, const int *col_upto, int tile_b, int n,
                                int np, int dim, int is_f64, double scale, void *acc, void *vel, double half_dt,
                                int do_kick, void *pos, void *packed, double dt, hipStream_t st, int p_begin, int p_end,
                                double *sums64, double *mm_part)
{
    if (p_end < 0 || p_end > n) p_end = n;
    if (p_end <= p_begin) return hipSuccess;
    const int blk0 = p_begin / 64;                  // chunk boundaries are tile boundaries (multiples of 64)
    const int grid = (p_end + 63) / 64 - blk0;
#define NB_RED(TT, DD) \
    hipLaunchKernelGGL((reduce_sym_kernel<TT, DD>), dim3(grid), dim3(64 * NB_RED_WAVES), 0, st, rowslab, (const TT *)colslab, \
                       row_slot0, row_nslots, col_upto, tile_b, n, np, scale, (TT *)acc, (TT *)vel, (TT)half_dt, do_kick, \
                       (TT *)pos, (TT *)packed, (TT)dt, blk0, sums64, mm_part)
    if (dim != 2 && dim != 3) return hipErrorInvalidValue;
    if (is_f64) { if (dim == 2) NB_RED(double, 2); else NB_RED(double, 3); }
    else        { if (dim == 2) NB_RED(float, 2); else NB_RED(float, 3); }
#undef NB_RED
    return hipGetLastError();
}
