// nb_internal.h -- shared declarations between the C-ABI layer (nb_api.cpp) and the HIP
// kernels (nb_force.hip, nb_misc.hip).  gfx950 only; no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nbody_amd.h"

// ---- launch geometry ---------------------------------------------------------------------
constexpr int NB_BLOCK = 256;   // 4 wavefronts of 64
constexpr int NB_TJ = 256;      // sources staged in LDS per tile (one per thread)
constexpr int NB_MAX_LUT = 4096; // grid levels served by the threshold tables (LDS: 8 bytes per level, sized per launch)
constexpr int NB_LUT_MIN = 256;   // smallest table allocation (INT8); tables are padded to a power of two
#ifndef NB_REC_LEVELS
#define NB_REC_LEVELS 32         // up to this many levels the pair loop reads one 16-byte record per pair
#endif

// hooks compiled into the pair loop (quantization.py:21-71)
enum { HOOK_NONE = 0, HOOK_BF16 = 1, HOOK_F16 = 2, HOOK_GRID = 3,
       HOOK_F32PAIR = 4 /* fp64 state, diff/r2 in fp32: FLOAT64-mode first evaluation */ };

// Device-resident grid tables written by grid_tables_kernel, read by the force kernel.
struct GridTables {
    float thr[NB_MAX_LUT + 1]; // thr[k] = smallest fp32 r2 whose bin index is >= k (thr[0] = -inf, thr[L] = NaN)
    float lut[NB_MAX_LUT];   // lut[k] = (1 / q_k^1.5) * G   in fp32 (simulation.py:97-101)
    float qval[NB_MAX_LUT];  // q_k = quantised distance-squared value of bin k
    float lmin, lmax, range; // log-grid bounds (quantization.py:109-113)
    float r2max;             // max over all pairs of fp32 r2
    float est_a, est_b;      // bin estimate: rint(log2(r2)*est_a + est_b) is within +-1 of the exact bin
    int use_est;             // 1: the estimate is safe (est_a small enough for v_log_f32's error)
    int degenerate;          // 1: lmax-lmin < 1e-10 -> values pass through clamped
    int levels;
    unsigned int r2max_bits; // atomicMax target (positive floats order as unsigned ints)
    unsigned int blocks_done; // grid_tables_kernel: arrival counter of its blocks (the last one finalises)
    int uniform_ok;          // (round 2: gate of the uniform / general pair of launches; informational since round 3 -- the
                             //  uniform-mass grid kernel serves every state of the tables itself)
    // Table-free pair path.  A pair whose bin estimate sits further than `sure_lim` from
    // a bin edge has bin rint(estimate) for certain; its force factor follows from the bin index alone:
    // log2(lut[k]) is affine in k, so lut[k] * 2^-tm = v_exp_f32((k - kc) * c1 + c0c), with kc the middle bin and
    // tm an integer that centres the exponent (both shrink the rounding of the fma; the power of two is put back
    // exactly when the finished sums are scaled).  grid_tables_kernel fits c1 / c0c to the table itself,
    // measures the estimate's deviation at EVERY bin edge and the factor's deviation from EVERY table entry
    // (the reference's own fp32 evaluation of the bin values scatters them by ~1e-6 around the affine law),
    // and only then sets fast_ok.
    int fast_ok;
    int kc, tm;
    float est_bc;            // est_b - kc: the centred estimate rounds to k - kc directly
    float sure_lim;          // 0.5 - delta: |estimate - rint(estimate)| at or below this -> bin is certain
    float c1, c0c;
    float fast_maxdev, fast_maxrel;   // measured: estimate deviation at the bin edges (bins), factor vs table (relative)
};

// table entries allocated in LDS for `levels` grid levels: next power of two, at least NB_LUT_MIN (the binary-search
// fallback walks a power-of-two table padded with +inf)
inline int nb_lut_pad(int levels)
{
    int p = NB_LUT_MIN;
    while (p < levels) p <<= 1;
    return p;
}
inline size_t nb_lut_lds_bytes(int levels) { return 2 * (size_t)(nb_lut_pad(levels) + 1) * sizeof(float); }

// Scratch of the pruned max-r2 search (nb_force.hip "K2 with pruning").
struct PruneState {
    unsigned int box_min[3];  // ordered-integer keys of the coordinate minima / maxima (bounding box)
    unsigned int box_max[3];
    unsigned long long far;   // (rho bits << 32 | index) of the particle farthest from the box centre
    unsigned long long lb[2]; // (r2 bits << 32 | index): farthest partner of `far`, then of that partner
    int count;                // candidates kept
    int nan_flag;             // a NaN coordinate was seen -> r2max is NaN
    // ---- tracked search (round 3): after a seeding evaluation every further one starts from its predecessor's result.
    // The farthest pair of the last evaluation, re-evaluated at the new positions, is the lower bound the two hops
    // used to find; the centre stays where the seeding evaluation put it (any fixed point serves the bound
    // dist(a, b) <= rho_a + rho_max), and the bound on rho_max is the last measured maximum plus a margin (twice its last
    // growth + 1e-4 of it: a RELATIVE margin alone would swallow the whole galaxy once a lone escaper is far out) that
    // the filter pass itself checks (a violated margin makes the scan fall back to all pairs: exact either way).
    int seeded;               // c / rho_m / pair are valid
    float c[3];               // centre of the candidate test
    float rho_m;              // assumed upper bound of every particle's distance from c (checked against rho_cur)
    float rho_prev;           // the measured maximum of the evaluation before (0: none yet) -- its growth sets the margin
    int pair_i, pair_j;       // a far pair of the last evaluation
    unsigned int rho_cur;     // filter pass: bits of the largest distance from c seen (atomicMax; distances are >= 0)
    unsigned int scan_done;   // scan: arrival counter of its workgroups (the last one finalises)
    unsigned long long best_i, best_j;   // scan: (r2 bits << 32 | particle index) maxima over all scanned pairs
};

struct ForceGeom {
    int n;          // particles
    int j_begin;    // first source of this rank's block
    int j_end;      // one past the last source of this rank's block
    int chunk_len;  // sources per blockIdx.y (multiple of NB_TJ)
    int nchunks;    // gridDim.y
    int r;          // targets per thread
};

// One workgroup of the pair-symmetric kernels: the super-row of target tiles
// [tile_i, tile_i + 4) (one per wave) against source tiles [jt_begin, jt_end); a wave skips the
// source tiles below its own target tile.
struct SymWork {
    int tile_i;        // first target tile of the super-row (multiple of 4)
    int jt_begin, jt_end;
    int slot;          // row-slab slot of wave 0 for this chunk; wave w uses slot + w * slot_stride
    int slot_stride;   // = number of (chunk, split) work items of this super-row
    int col_ord;       // column-slab index of this (super-row, split)
    int s_begin;       // first rotation step of this work item (0 unless the sweep is split)
    int s_count;       // rotation steps to run (64 = a full tile-vs-tile sweep)
};

// pack positions + mass factors into padded component arrays; kick != 0 fuses the opening
// Optional timing events of a launch: attached to the dispatch itself (hipExtLaunchKernelGGL), so the
// kernel's own start / end timestamps are taken without barrier packets around it on the stream.
struct NbKernelEvents {
    hipEvent_t start = nullptr, stop = nullptr;
};

// kick + drift of a step (simulation.py:132,135)
hipError_t nb_launch_pack(void *pos, void *vel, const void *acc, const void *mass, void *packed, int n, int np,
                          int dim, int is_f64, int kick, double half_dt, double dt, double gfac, int f32_pairs,
                          hipStream_t st, int p_begin = 0, int p_end = -1 /* packed entries [p_begin, p_end); -1 = np */,
                          int spread_pad = 0 /* padding particles at distinct far positions (uniform-mass potential energy) */);
hipError_t nb_launch_force_sym_f64(const double *packed, const SymWork *work, int nwork, double *rowslab,
                                   double *colslab, int np, int dim, int r, int uniform, int pa_f32, double eps2,
                                   hipStream_t st, NbKernelEvents ev = {}, int rowsplit = 0 /* row-split work items (nb_plan.cpp) */);
hipError_t nb_launch_force_sym_f32(const float *packed, const SymWork *work, int nwork, double *rowslab,
                                   float *colslab, int np, int dim, int r, int uniform, int hook, float eps2,
                                   const GridTables *tab, float G, float mass_value, int levels, hipStream_t st,
                                   NbKernelEvents ev = {});
hipError_t nb_launch_potential_sym(const void *packed, const SymWork *work, int nwork, double *part, int np, int dim,
                                   int r, int is_f64, int f32_terms, int mass_dt /* nb_dtype of the masses */, double eps2,
                                   int uniform /* all masses equal: the caller applies the mass product to the sum */, hipStream_t st);
hipError_t nb_launch_final_sum(const double *part, int count, double *out, hipStream_t st);
hipError_t nb_launch_reduce_sym(const double *rowslab, const void *colslab, const int *row_slot0,
                                const int *row_nslots, const int *col_upto, int tile_b, int n,
                                int np, int dim, int is_f64, double scale, void *acc, void *vel, double half_dt,
                                int do_kick /* 1: closing kick, 2: + next opening kick + drift + repack, 3: closing kick + the next step's positions speculatively into pos_next / packed; | 4: this step's opening kick on read */,
                                void *pos, void *packed, double dt, hipStream_t st,
                                int p_begin = 0, int p_end = -1 /* particles [p_begin, p_end); -1 = n */,
                                double *sums64 = nullptr /* instead of acc / kicks: the unscaled, unrounded fp64 sums */,
                                double *mm_part = nullptr /* per-workgroup {min, max} of the forces written (whole-range launches) */,
                                void *pos_next = nullptr /* do_kick mode 3 */);
// fp32 state, multi-GPU: acc = (float)(sums64 * scale) after the ranks' fp64 sums were added, + the kicks of mode
// (0 none, 1 closing, 2 closing + next opening + drift + repack) -- the tail of reduce_sym_kernel, after the exchange
hipError_t nb_launch_finish_sums64(const double *sums64, double scale, float *acc, float *vel, float *pos, float *packed,
                                   int n, int np, int dim, int mode, double half_dt, double dt, hipStream_t st);

// ---- kernel launchers (implemented in the .hip files) --------------------------------------
// T = storage/accumulation type of the state (float or double); pa_f32 != 0 selects fp32
// pair arithmetic for diff / r2 (always for T=float; for T=double it is the FLOAT64-mode first
// evaluation on fp32-typed positions).
// qhook >= 0: fp64 positions under a cast mode (hook output fp32); pa: NB_F32, or NB_F16 / NB_BF16 for
// the first evaluation on half-typed state (eps2 then already rounded to that type).
hipError_t nb_launch_force_f64(const double *pos, const double *mass, double *partial, const ForceGeom &g,
                               int dim, int pair_dt, int qhook, double G, double eps2_py, float eps2_pair,
                               hipStream_t st);
hipError_t nb_launch_force_f32(const float *pos, const float *mass, double *partial, const ForceGeom &g,
                               int dim, int hook, int pa, float G, float eps2, const GridTables *tab, int levels,
                               hipStream_t st, unsigned long long *bin_out = nullptr /* bin read-out, see below */);
// Bin read-out (nb_quant_bin_sums): the production grid-mode pair loops instantiated with BINS = true add, for every
// pair, the bin index they decided to per-particle integer checksums: bin_out = {s1[n], s2[n], {table-free pairs,
// table pairs}} (unsigned 64-bit, zeroed by the caller).
hipError_t nb_launch_force_sym_f32_bins(const float *packed, const SymWork *work, int nwork, double *rowslab,
                                        float *colslab, int np, int dim, int r, int uniform, float eps2,
                                        const GridTables *tab, float G, float mass_value, int levels,
                                        unsigned long long *bin_out, int bin_n, hipStream_t st);
// tensor-level _grid_quantize_safe of an fp32 tensor through threshold / value tables built for its own bounds
// (bounds: device, {min, max} of the tensor as doubles; tab: device scratch); levels <= NB_MAX_LUT
hipError_t nb_launch_grid_quantize_safe_tab(const float *in, float *out, int64_t count, int levels, float min_val,
                                            const double *bounds, GridTables *tab, hipStream_t st);
// small systems: all-pairs maximum and the tables of the evaluation in ONE launch (levels <= NB_LUT_MIN)
hipError_t nb_launch_r2max_tables(const float *pos, const ForceGeom &g, int dim, float eps2, GridTables *tab, int levels,
                                  float G, float min_val, int allow_fast, hipStream_t st);
hipError_t nb_launch_r2max(const float *pos, const ForceGeom &g, int dim, float eps2, GridTables *tab,
                           hipStream_t st);
// exact max of the fp32 r2 over all pairs via candidate pruning (every rank computes it redundantly,
// O(N) + (candidates)^2 work, no collective); cand: n*dim floats, rho: n floats, st: PruneState
hipError_t nb_launch_r2max_pruned(const float *pos, int n, int dim, float eps2, float *cand, float *rho,
                                  PruneState *ps, GridTables *tab, hipStream_t st);
// tracked search, two launches: filter (O(N): candidates of the maximal pair, compacted with their indices) and scan
// (exact max over candidate pairs; its last workgroup finalises, prepares the next evaluation's search and -- for
// levels <= NB_LUT_MIN -- builds the tables as well: no grid_tables launch).  Needs a seeded *ps (nb_launch_r2max_pruned
// followed by nb_launch_grid_tables seeds it).
hipError_t nb_launch_r2max_tracked(const float *pos, int n, int dim, float eps2, float *cand, int *cand_idx, PruneState *ps,
                                   GridTables *tab, int levels, float G, float min_val, int allow_fast, hipStream_t st);
hipError_t nb_launch_grid_tables(GridTables *tab, int levels, float G, float eps2, float min_val,
                                 PruneState *ps /* reset after use; may be null */, hipStream_t st, int allow_fast = 1);
hipError_t nb_launch_d2bins(const float *pos, int n, int dim, float eps2, const GridTables *tab,
                            int16_t *bins, hipStream_t st, int i0 = 0, int i1 = -1 /* rows [i0, i1); -1 = n */);

// reduce the S partial slabs in fixed order; optionally fuse the closing half kick
hipError_t nb_launch_reduce(const double *partial, int nchunks, int64_t count, void *acc, int is_f64,
                            void *vel, double half_dt, int do_kick /* 1: closing kick, 2: + next opening kick + drift */,
                            void *pos, double dt, hipStream_t st);
hipError_t nb_launch_axpy(void *y, const void *x, double scalar, int64_t count, int is_f64, hipStream_t st);
hipError_t nb_launch_kick_drift(void *pos, void *vel, const void *acc, double half_dt, double dt,
                                int64_t count, int is_f64, hipStream_t st);
hipError_t nb_launch_convert(const void *in, int in_dt, void *out, int out_dt, int64_t count, hipStream_t st);

// linear force grid (quantization.py:74-88) applied in place inside the step, fp32, with bin output
hipError_t nb_launch_force_quant_step(float *acc, int64_t count, int levels, double *mn_mx, double *partials,
                                      int16_t *bins, float *vel, float *pos, double half_dt, double dt,
                                      int kick /* 0 none, 1 closing kick, 2 + next opening kick + drift */,
                                      float *packed /* symmetric path: also repack the new positions, else null */,
                                      int np, int dim, hipStream_t st);
hipError_t nb_launch_force_quant_bins(const float *in, float *out, int64_t count, int levels, const double *mn_mx,
                                      int16_t *bins, hipStream_t st);

hipError_t nb_launch_kinetic(const void *vel, const void *mass, int n, int dim, int is_f64, int vel_f32_logical,
                             int half_pa, double *scratch, double *out, hipStream_t st);
hipError_t nb_launch_potential(const void *pos, const void *mass, const ForceGeom &g, int dim, int is_f64,
                               int pa_f32, int mass_dt /* nb_dtype of the masses */, int half_pa, double eps2_py, float eps2_half,
                               double *scratch, double *out, hipStream_t st);

// tensor-level hooks (quantization.py module functions)
hipError_t nb_launch_cast_hook(const void *in, int in_dt, void *out, int mode, int64_t count, hipStream_t st);
// two-stage min/max; partials: device scratch of 2 * NB_MINMAX_BLOCKS doubles
constexpr int NB_MINMAX_BLOCKS = 2048;   // enough waves for HBM speed on N x N tensors (256 ran at 0.6 TB/s)
hipError_t nb_launch_minmax_generic(const void *in, int is_f64, int64_t count, int log_clamped, double min_val,
                                    double *mn_mx /* device, 2 doubles */, double *partials, hipStream_t st);
hipError_t nb_launch_grid_quantize(const void *in, void *out, int is_f64, int64_t count, int levels,
                                   const double *mn_mx, hipStream_t st);
hipError_t nb_launch_grid_quantize_safe(const void *in, void *out, int is_f64, int64_t count, int levels,
                                        double min_val, const double *mn_mx, hipStream_t st);

// ---- diagnostics (nb_metrics.hip; reference metrics.py:25-156) ------------------------------------------------
struct NbMetricsArgs {
    const void *pos, *vel, *mass;   // device, storage type S (float or double)
    int n, dim;
    int storage_f64;                // S
    int arith_f64;                  // A: the tensors' logical dtype (per-particle arithmetic follows torch in A)
    int num_bins;                   // rotation-curve bins (<= 255)
    const float *edges;             // device, num_bins + 1 float32 bin edges, or null: linspace(0, max_radius) restated
    double max_radius;              // < 0: radii.max()
    int kth;                        // order statistic of the radii to report (min(int(N p / 100), N - 1))
    double G;
    int radius_only;                // 1: only out[0] = radii.max() (first phase of a caller that builds the edges itself)
    void *scratch;                  // nb_metrics_scratch_bytes(n, num_bins)
    double *out;                    // device, 5 + 2 num_bins doubles: max r, r_kth, bound count, dispersion, means[nb],
                                    // counts[nb], max_radius used
};
size_t nb_metrics_scratch_bytes(int n, int num_bins);
// stable radix sorts of 32- / 64-bit unsigned keys (nb_sort.hip: rocPRIM); values are particle indices
size_t nb_sort_temp_bytes(int n, int key64);
hipError_t nb_sort_keys(void *tmp, size_t tmp_bytes, const void *kin, void *kout, int n, int key64, hipStream_t st);
hipError_t nb_sort_pairs(void *tmp, size_t tmp_bytes, const void *kin, void *kout, const int *vin, int *vout, int n,
                         int key64, hipStream_t st);
hipError_t nb_launch_metrics(const NbMetricsArgs &a, hipStream_t st);

// ---- direct xGMI all-reduce of small force vectors (nb_p2p.hip) ------------------------------------------------
size_t nb_p2p_handle_bytes();
hipError_t nb_p2p_export(int device, int rank, int nranks, size_t cap_bytes, void *handle_out);
hipError_t nb_p2p_import(const void *handles);
int nb_p2p_state();                 // 0 none, 1 attached, 2 enabled
void nb_p2p_enable(bool on);
size_t nb_p2p_capacity();           // bytes of the shared input buffer (0: not attached)
void *nb_p2p_data();                // this rank's shared input buffer (device pointer)
int nb_p2p_nranks();
int nb_p2p_device();
// leapfrog work fused behind the sum: mode 1 closing half kick, 2 + the next step's opening kick + drift (+ repack)
struct NbP2PKick {
    int mode, dim, np;
    int f64_to_f32;                // the vector holds fp64 sums, the result (and vel / pos / packed) is fp32:
    double scale;                  //   result = (float)(sum * scale)
    void *vel, *pos, *packed;      // storage type of the force vector; packed may be null (one-sided kernels)
    double half_dt, dt;
};
hipError_t nb_p2p_allreduce(void *dst, size_t count, int is_f64, double timeout_s, hipStream_t st,
                            const NbP2PKick *kick = nullptr);
hipError_t nb_p2p_status(int *status);
hipError_t nb_p2p_selftest_round(void *scratch, size_t count, int is_f64, int round, double timeout_s, int *bad_dev,
                                 hipStream_t st);
void nb_p2p_shutdown();
hipError_t nb_p2p_virtual(int nranks, size_t count, int is_f64, int concurrent, int iters, double timeout_s, int *bad_total,
                          double *us_per_call);

// ---- dtype-faithful generic force evaluation (nb_generic.hip) ---------------------------------------------------
size_t nb_generic_scalars_bytes();
hipError_t nb_launch_generic_r2max(const void *pos, int storage_f64, int n, int dim, int P, double eps2_py, void *sc,
                                   hipStream_t st);
hipError_t nb_launch_generic_force(const void *pos, const void *mass, int storage_f64, double *partial, const ForceGeom &geom,
                                   int dim, int P /* positions' dtype */, int M /* masses' dtype */, int mode, int levels,
                                   double G, double eps2_py, const void *sc, void *acc, int A /* result dtype */,
                                   hipStream_t st);

// ---- one-launch step for small systems (nb_small.hip) -------------------------------------------------------------
int nb_small_lanes(int n);
int nb_small_block(int n);          // threads per workgroup of the one-launch step at this size (256 or 512)
hipError_t nb_launch_small_step(const void *pos_in, void *pos_out, void *vel, void *acc, const void *mass, int n, int dim,
                                int is_f64, int hook, double G, double eps2, double half_dt, double dt,
                                int do_kick /* 0 force only, 1 + closing kick, 2 + next opening kick + drift into pos_out, 3 closing kick + speculative next positions into pos_out; | 4 opening kick on read */,
                                int lanes /* 16 / 32 / 64 lanes per target */, hipStream_t st,
                                const GridTables *tab = nullptr /* HOOK_GRID: this evaluation's tables */,
                                double *part = nullptr /* INT8 / INT4: 2 n doubles, per-target min / max of the forces */,
                                unsigned long long *bin_out = nullptr /* HOOK_GRID: bin read-out (BINS instantiation) */);
// second half of nb_launch_force_quant_step with caller-provided min / max partials (nblocks pairs of doubles)
hipError_t nb_launch_force_quant_finish(float *acc, int64_t count, int levels, const double *partials, int nblocks,
                                        double *mn_mx, int16_t *bins, float *vel, float *pos, double half_dt, double dt,
                                        int kick, hipStream_t st, float *packed = nullptr, int np = 0, int dim = 1);
