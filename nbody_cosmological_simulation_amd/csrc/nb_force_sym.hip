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
#include "nb_force_sym_kernel.h"

namespace {

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
// spread != 0 (potential energy with uniform masses, where no mass factor silences the padding): padding particle
// number k sits at pad * (1 + k) along x, so pad-pad pairs are as far apart as pad-real ones.
template <typename T, int D, int KICK>
__global__ void __launch_bounds__(NB_BLOCK)
pack_kernel(T *__restrict__ pos, T *__restrict__ vel, const T *__restrict__ acc, const T *__restrict__ mass,
            T *__restrict__ packed, int n, int np, T half_dt, T dt, T gfac, T pad, int p_begin, int p_end, int spread)
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
        for (int k = 0; k < D; ++k) packed[(size_t)k * np + p] = (spread && k == 0) ? pad * (T)(1 + (p - n)) : pad;
        packed[(size_t)D * np + p] = (T)0;
    }
}

// acc[p] = sum of the row slots of p's tile + sum of the column-slab entries of the owned super-rows at or
// above tile(p) (a prefix [0, col_upto[J]) of the slab index space), always in the same order.
// Block = 64 particles x 16 waves: the items (row slots, then slab entries) are dealt over the waves
// (item c to wave c mod 16, four independent loads in flight per lane), the 16 partial sums are
// combined in wave order through LDS -> one fixed summation tree.  Optionally fuses the closing half
// kick (simulation.py:141) and applies the uniform-mass factor.
#ifndef NB_RED_WAVES_N
#define NB_RED_WAVES_N 16
#endif
constexpr int NB_RED_WAVES = NB_RED_WAVES_N;
template <typename T, int D>
__global__ void __launch_bounds__(64 * NB_RED_WAVES)
reduce_sym_kernel(const double *__restrict__ rowslab, const T *__restrict__ colslab,
                  const int *__restrict__ row_slot0, const int *__restrict__ row_nslots,
                  const int *__restrict__ col_upto, int tile_b, int n, int np,
                  double scale, T *__restrict__ acc, T *__restrict__ vel, T half_dt, int do_kick,
                  T *__restrict__ pos, T *__restrict__ packed, T dt, int blk0, double *__restrict__ sums64,
                  double *__restrict__ mm_part, T *__restrict__ pos_next)
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
            // do_kick bit 2: the velocities hold the previous step's CLOSING state and the positions were taken from its
            // speculative drift -- this step's opening kick is applied here, with the accelerations being replaced
            const T a_prev = (do_kick & 4) ? acc[idx] : (T)0;
            acc[idx] = a;
            if (mm_part) {
                const double ad = (double)a;
                mm_lo = (ad != ad || mm_lo != mm_lo) ? __builtin_nan("") : (ad < mm_lo ? ad : mm_lo);
                mm_hi = (ad != ad || mm_hi != mm_hi) ? __builtin_nan("") : (ad > mm_hi ? ad : mm_hi);
            }
            const int kmode = do_kick & 3;
            if (kmode != 0) {
                T v = vel[idx];
                if (do_kick & 4) v = axpy_rn<T>(v, a_prev, half_dt);
                v = axpy_rn<T>(v, a, half_dt);                                   // closing kick
                if (kmode == 1) {
                    vel[idx] = v;
                } else if (kmode == 2) {
                    // ... then the next step's opening kick + drift and its repack (what pack_kernel<KICK=1>
                    // would do in a launch of its own; mass factors and padding in `packed` do not change)
                    v = axpy_rn<T>(v, a, half_dt);
                    const T x = axpy_rn<T>(pos[idx], v, dt);
                    vel[idx] = v;
                    pos[idx] = x;
                    packed[(size_t)k * np + p] = x;
                } else {
                    // last step of a native call: the state stays at the closing kick (what a reader must see); the
                    // positions the NEXT step drifts to go to pos_next and, packed, to the force kernel's input.  The next
                    // nb_step takes them if nothing wrote state, dt or `packed` in between (nb_step.cpp: step_run)
                    vel[idx] = v;
                    const T vo = axpy_rn<T>(v, a, half_dt);
                    const T x = axpy_rn<T>(pos[idx], vo, dt);
                    pos_next[idx] = x;
                    packed[(size_t)k * np + p] = x;
                }
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
// those terms carry weight 1/2 and the self pairs are skipped.
//   UNIFORM   all masses equal: the mass product is a constant the host applies to the finished sum -- no mass
//             loads, rotations or multiplies in the pair loop (padding particles are spread out by pack_kernel so
//             that pad-pad pairs are as far apart as pad-real ones: 1 / dist ~ 1e-150, absorbed).
//   F32T      fp32-typed state: term arithmetic in fp32 (one v_rsq_f32 per pair; r2 from fused multiply-adds -- no
//             rounding DECISION hangs on it here, unlike the grid modes), source slots paired in float2 halves so the
//             differences, r2 and the accumulation issue as v_pk_*_f32; fp32 running sums of one tile (R/2 x 2 chains
//             of 64 R terms) are folded into the fp64 sum after every tile.
//   otherwise fp64: v_rsq_f64 (2^-24) + ONE Newton step folded into the accumulation, y = y0 (1 + e / 2), e = 1 - q y0^2:
//             the neglected 3 e^2 / 8 is < 5e-15 relative (asserted energy bar: 1e-12).  8 VALU + 1 rsq per pair for
//             uniform masses in 2-D (the force kernel: 14 + 1).
// Per-lane fp64 sums, wave shuffle + LDS block sum, one partial per workgroup.
// TM: type the masses are carried (and rotated) in -- float when the masses are fp32- / half-typed (their product is
// rounded in that dtype upstream: one fp32 multiply and one conversion per pair instead of three conversions), else T.
template <typename T, typename TM, int D, int R, bool DIAG, bool UNIFORM>
__device__ __forceinline__ double pe_sweep_f64(const T (&xi)[R][D], const TM (&mi)[R], T (&xj)[R][D], TM (&mj)[R], double eps2,
                                               int mass_dt, int rot_addr, int s_begin, int s_count)
{
    double tsum[R];             // independent chains: a single accumulator would serialise the adds
#pragma unroll
    for (int r = 0; r < R; ++r) tsum[r] = 0.0;
    double half = 0.5, one = 1.0;
    asm volatile("" : "+v"(half), "+s"(one));
#pragma unroll 1
    for (int s = s_begin; s < s_begin + s_count; ++s) {
#pragma unroll
        for (int rj = 0; rj < R; ++rj) {
#pragma unroll
            for (int ri = 0; ri < R; ++ri) {
                double d[D];
#pragma unroll
                for (int k = 0; k < D; ++k) d[k] = (double)xj[rj][k] - (double)xi[ri][k];
                double q = __builtin_fma(d[D - 1], d[D - 1], eps2);
#pragma unroll
                for (int k = D - 2; k >= 0; --k) q = __builtin_fma(d[k], d[k], q);
                const double y0 = __builtin_amdgcn_rsq(q);
                const double e = __builtin_fma(-(q * y0), y0, one);
                const double c = __builtin_fma(e, half, one);
                if (UNIFORM) {
                    if (DIAG && ri == rj) tsum[rj] = (s == 0) ? tsum[rj] : __builtin_fma(y0, c, tsum[rj]);     // self pair
                    else tsum[rj] = __builtin_fma(y0, c, tsum[rj]);
                } else {
                    double mp;
                    if constexpr (std::is_same_v<TM, float>) mp = (double)mass_prod_f32(mi[ri], mj[rj], mass_dt);
                    else mp = (double)mi[ri] * (double)mj[rj];
                    const double term = mp * (y0 * c);
                    if (DIAG && ri == rj) tsum[rj] += (s == 0) ? 0.0 : term;
                    else tsum[rj] += term;
                }
            }
#pragma unroll
            for (int k = 0; k < D; ++k) xj[rj][k] = rot1<T>(xj[rj][k], rot_addr);
            if (!UNIFORM) mj[rj] = rot1<TM>(mj[rj], rot_addr);
        }
    }
    double tile = tsum[0];
#pragma unroll
    for (int r = 1; r < R; ++r) tile += tsum[r];
    return DIAG ? 0.5 * tile : tile;
}

template <int D, int R, bool DIAG, bool UNIFORM>
__device__ __forceinline__ double pe_sweep_f32(const float (&xi)[R][D], const float (&mi)[R], f2 (&xj2)[R / 2][D], f2 (&mj2)[R / 2],
                                               float eps2, int mass_dt, int rot_addr, int s_begin, int s_count)
{
    f2 ts2[R / 2];
#pragma unroll
    for (int h = 0; h < R / 2; ++h) ts2[h] = f2{0.0f, 0.0f};
#pragma unroll 1
    for (int s = s_begin; s < s_begin + s_count; ++s) {
#pragma unroll
        for (int h = 0; h < R / 2; ++h) {
#pragma unroll
            for (int ri = 0; ri < R; ++ri) {
                f2 d[D];
#pragma unroll
                for (int k = 0; k < D; ++k) d[k] = xj2[h][k] - xi[ri][k];
                f2 r2 = __builtin_elementwise_fma(d[D - 1], d[D - 1], f2{eps2, eps2});
#pragma unroll
                for (int k = D - 2; k >= 0; --k) r2 = __builtin_elementwise_fma(d[k], d[k], r2);
                f2 y = {__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
                if (DIAG && s == 0) {                       // self pairs: slot 2h / 2h + 1 against target slot ri
                    if (2 * h == ri) y.x = 0.0f;
                    if (2 * h + 1 == ri) y.y = 0.0f;
                }
                if (UNIFORM) {
                    ts2[h] = ts2[h] + y;
                } else {
                    f2 mp = mj2[h] * mi[ri];                 // exact for half inputs; rounded to the masses' dtype like upstream
                    if (mass_dt == NB_F16) mp = f2{round_f16(mp.x), round_f16(mp.y)};
                    else if (mass_dt == NB_BF16) mp = f2{round_bf16(mp.x), round_bf16(mp.y)};
                    ts2[h] = __builtin_elementwise_fma(mp, y, ts2[h]);
                }
            }
#pragma unroll
            for (int k = 0; k < D; ++k) xj2[h][k] = rot1_f2(xj2[h][k], rot_addr);
            if (!UNIFORM) mj2[h] = rot1_f2(mj2[h], rot_addr);
        }
    }
    double tile = 0.0;
#pragma unroll
    for (int h = 0; h < R / 2; ++h) tile += (double)ts2[h].x + (double)ts2[h].y;
    return DIAG ? 0.5 * tile : tile;
}

template <typename T, int D, int R, bool F32T, bool UNIFORM>
__global__ void __launch_bounds__(NB_BLOCK)
potential_sym_kernel(const T *__restrict__ packed, const SymWork *__restrict__ work, double *__restrict__ part,
                     int np, double eps2, float eps2_f, int mass_dt)
{
    constexpr int B = 64 * R;
    __shared__ double s_red[NB_BLOCK / 64];
    const SymWork wk = work[blockIdx.x];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // wave index in an SGPR
    // row-split work items (see force_sym_kernel): one target tile, the rotation steps shared among the four waves
    const bool rowsplit = wk.slot_stride < 0;
    const int I = wk.tile_i + (rowsplit ? 0 : wave);
    const int wk_s_begin = rowsplit ? wk.s_begin + wk.s_count * wave / (NB_BLOCK / 64) : wk.s_begin;
    const int wk_s_count = rowsplit ? wk.s_begin + wk.s_count * (wave + 1) / (NB_BLOCK / 64) - wk_s_begin : wk.s_count;
    const int rot_addr = ((lane + 1) & 63) << 2;
    using TA = std::conditional_t<F32T, float, T>;      // arithmetic type of the pair terms

    TA xi[R][D], mi[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int p = I * B + r * 64 + lane;
#pragma unroll
        for (int k = 0; k < D; ++k) xi[r][k] = (TA)packed[(size_t)k * np + p];
        mi[r] = UNIFORM ? (TA)1 : (TA)packed[(size_t)D * np + p];
    }
    double sum = 0.0;
    for (int J = max(wk.jt_begin, I); J < wk.jt_end; ++J) {
        const bool diag = (J == I);
        if constexpr (F32T) {
            f2 xj2[R / 2][D], mj2[R / 2];
#pragma unroll
            for (int h = 0; h < R / 2; ++h) {
                const int p0 = J * B + 2 * h * 64 + ((lane + wk_s_begin) & 63);
#pragma unroll
                for (int k = 0; k < D; ++k) xj2[h][k] = f2{(float)packed[(size_t)k * np + p0], (float)packed[(size_t)k * np + p0 + 64]};
                mj2[h] = UNIFORM ? f2{1.0f, 1.0f} : f2{(float)packed[(size_t)D * np + p0], (float)packed[(size_t)D * np + p0 + 64]};
            }
            sum += diag ? pe_sweep_f32<D, R, true, UNIFORM>(xi, mi, xj2, mj2, eps2_f, mass_dt, rot_addr, wk_s_begin, wk_s_count)
                        : pe_sweep_f32<D, R, false, UNIFORM>(xi, mi, xj2, mj2, eps2_f, mass_dt, rot_addr, wk_s_begin, wk_s_count);
        } else {
            T xj[R][D], mj[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int p = J * B + r * 64 + ((lane + wk_s_begin) & 63);
#pragma unroll
                for (int k = 0; k < D; ++k) xj[r][k] = packed[(size_t)k * np + p];
                mj[r] = UNIFORM ? (T)1 : packed[(size_t)D * np + p];
            }
            if (!UNIFORM && mass_dt != NB_F64) {            // kernel-uniform: masses typed narrower than the positions
                float mif[R], mjf[R];
#pragma unroll
                for (int r = 0; r < R; ++r) { mif[r] = (float)mi[r]; mjf[r] = (float)mj[r]; }
                sum += diag ? pe_sweep_f64<T, float, D, R, true, UNIFORM>(xi, mif, xj, mjf, eps2, mass_dt, rot_addr, wk_s_begin, wk_s_count)
                            : pe_sweep_f64<T, float, D, R, false, UNIFORM>(xi, mif, xj, mjf, eps2, mass_dt, rot_addr, wk_s_begin, wk_s_count);
            } else {
                sum += diag ? pe_sweep_f64<T, T, D, R, true, UNIFORM>(xi, mi, xj, mj, eps2, mass_dt, rot_addr, wk_s_begin, wk_s_count)
                            : pe_sweep_f64<T, T, D, R, false, UNIFORM>(xi, mi, xj, mj, eps2, mass_dt, rot_addr, wk_s_begin, wk_s_count);
            }
        }
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

}  // namespace

hipError_t nb_launch_pack(void *pos, void *vel, const void *acc, const void *mass, void *packed, int n, int np,
                          int dim, int is_f64, int kick, double half_dt, double dt, double gfac, int f32_pairs,
                          hipStream_t st, int p_begin, int p_end, int spread_pad)
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
                       p_begin, p_end, spread_pad)
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
                                   hipStream_t st, NbKernelEvents ev, int rowsplit)
{
    if (rowsplit) {         // row-split work items (nb_plan.cpp): their own instantiations
        if (dim != 2 || r != 4) return hipErrorInvalidValue;
        if (pa_f32) return launch_sym_rowsplit<HOOK_F32PAIR>(packed, work, nwork, rowslab, colslab, np, uniform, eps2, (float)eps2, st, ev);
        return launch_sym_rowsplit<HOOK_NONE>(packed, work, nwork, rowslab, colslab, np, uniform, eps2, 1.0f, st, ev);
    }
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
                                   int r, int is_f64, int f32_terms, int mass_dt, double eps2, int uniform, hipStream_t st)
{
    const float e32 = (float)eps2;
#define NB_PES(TT, DD, RR, FF)                                                                                           \
    do {                                                                                                                 \
        if (uniform)                                                                                                     \
            hipLaunchKernelGGL((potential_sym_kernel<TT, DD, RR, FF, true>), dim3(nwork), dim3(NB_BLOCK), 0, st,         \
                               (const TT *)packed, work, part, np, eps2, e32, mass_dt);                                  \
        else                                                                                                             \
            hipLaunchKernelGGL((potential_sym_kernel<TT, DD, RR, FF, false>), dim3(nwork), dim3(NB_BLOCK), 0, st,        \
                               (const TT *)packed, work, part, np, eps2, e32, mass_dt);                                  \
    } while (0)
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
                                double *sums64, double *mm_part, void *pos_next)
{
    if ((do_kick & 3) == 3 && !pos_next) return hipErrorInvalidValue;
    if (p_end < 0 || p_end > n) p_end = n;
    if (p_end <= p_begin) return hipSuccess;
    const int blk0 = p_begin / 64;                  // chunk boundaries are tile boundaries (multiples of 64)
    const int grid = (p_end + 63) / 64 - blk0;
#define NB_RED(TT, DD) \
    hipLaunchKernelGGL((reduce_sym_kernel<TT, DD>), dim3(grid), dim3(64 * NB_RED_WAVES), 0, st, rowslab, (const TT *)colslab, \
                       row_slot0, row_nslots, col_upto, tile_b, n, np, scale, (TT *)acc, (TT *)vel, (TT)half_dt, do_kick, \
                       (TT *)pos, (TT *)packed, (TT)dt, blk0, sums64, mm_part, (TT *)pos_next)
    if (dim != 2 && dim != 3) return hipErrorInvalidValue;
    if (is_f64) { if (dim == 2) NB_RED(double, 2); else NB_RED(double, 3); }
    else        { if (dim == 2) NB_RED(float, 2); else NB_RED(float, 3); }
#undef NB_RED
    return hipGetLastError();
}

#ifdef NB_WG_TRACE
// experimental builds only: copy the workgroup trace of the last force_sym_kernel launch of this library (tools/wg_trace.py)
extern "C" int nb_debug_wg_trace(unsigned long long *host, int nwg)
{
    if (nwg > NB_WG_TRACE_MAX) nwg = NB_WG_TRACE_MAX;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(nb_wg_trace_buf), sizeof(unsigned long long) * 8 * (size_t)nwg) == hipSuccess ? nwg : -1;
}
#endif
