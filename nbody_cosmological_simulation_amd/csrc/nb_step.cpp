// nb_step.cpp -- kernel selection and sequencing of the hot path behind the C-ABI (include/nbody_amd.h):
//   force_eval   one evaluation of GalaxySimulation._compute_accelerations (reference simulation.py:74-118)
//   step_run     kick-drift-kick leapfrog steps (simulation.py:120-143), launches fused as far as each path allows
//   energy_eval  kinetic / potential energy (simulation.py:170-192)
//   bin_sums_eval  the quant-bin read-out of the production grid-mode pair loops (nb_quant_bin_sums)
// Everything is queued on the handle's own HIP stream; the only host waits are the ones the callers need.
#include <algorithm>
#include <cstring>

#include "nb_state.h"

namespace nbhost {

// ---- size thresholds of the kernel selection, in one place (measured crossovers on MI355X) ----
struct NbTuning {
    int onesided_r1_max_n = 8192;       // one-sided fp64 kernel: one target per thread up to here (parallelism-bound)
    int onesided_target_wgs = 1024;     // ... and enough source chunks for >= 4 workgroups per CU
    int prune_min_n = 8192;             // grid modes, first evaluation: pruned max-r2 search above, all-pairs scan at or below
    int track_min_n = 3072;             // grid modes: above, every evaluation after the first TRACKS the farthest pair of its
                                        // predecessor (two launches incl. the tables; the seed then always prunes).  On the
                                        // one-launch small-system path the all-pairs pass costs the same (measured INT8 / CUSTOM
                                        // N = 3000: 31.3 / 27.1 vs 31.0 / 26.7 us per step: the table construction by a single
                                        // workgroup is what is left there, not the search)
    int red_mm_max_blocks = 1024;       // INT8 / INT4: the reduction also hands out force min / max partials up to
                                        // N = 65 536 (one launch fewer: INT8 step 785.8 -> 784.2 us there; beyond, every
                                        // finish workgroup would fold thousands of them)
    int small_max_f64 = 4096;           // one-launch step: fp64 4.9 / 7.9 / 11.4 / 16.9 us per step at N = 1024 ... 4096
    int small_max_f32 = 3072;           // fp32 storage: above, the tiled path is ahead (FLOAT32) or level (INT8 / INT4)
    int small_fuse_tables_max_n = 2048; // small grid steps: max-r2 launch also builds the tables up to here
};
static const NbTuning g_tune{};

void compute_geometry(nb_sim *s)
{
    const int n = s->cfg.n;
    ForceGeom g{};
    g.n = n;
    g.j_begin = (int)((int64_t)s->cfg.rank * n / s->cfg.nranks);
    g.j_end = (int)((int64_t)(s->cfg.rank + 1) * n / s->cfg.nranks);
    // targets per thread of the one-sided fp64 kernel.  Small systems are parallelism-bound, not
    // throughput-bound: R = 1 doubles the workgroups (N = 1024: 27.8 -> 17.6 us per step, N = 4096:
    // 32.5 -> 22.6 us)
    g.r = (n <= g_tune.onesided_r1_max_n) ? 1 : 2;
    if (s->knobs.r_onesided == 1 || s->knobs.r_onesided == 2 || s->knobs.r_onesided == 4)   // NB_R tuning knob
        g.r = s->knobs.r_onesided;
    const int njr = std::max(g.j_end - g.j_begin, 1);
    const int itiles = (n + NB_BLOCK * g.r - 1) / (NB_BLOCK * g.r);
    const int max_chunks = (njr + NB_TJ - 1) / NB_TJ;
    int nch = (g_tune.onesided_target_wgs + itiles - 1) / itiles;       // aim for >= 4 workgroups per CU
    nch = std::max(1, std::min(std::min(nch, max_chunks), 64));
    int chunk = (njr + nch - 1) / nch;
    chunk = (chunk + NB_TJ - 1) / NB_TJ * NB_TJ;
    g.chunk_len = chunk;
    g.nchunks = (njr + chunk - 1) / chunk;
    s->geom = g;
}

int acc_logical_dtype(const nb_sim *s)
{
    // promote(promote(Q, M), P) with Q = hook output dtype (quantization.py:43-71)
    int q = s->logical[0];
    if (s->cfg.mode == NB_FLOAT64) q = NB_F64;
    else if (s->cfg.mode <= NB_FLOAT16) q = NB_F32;
    return promote(promote(promote(q, s->logical[2]), NB_F32), s->logical[0]);
}

namespace {

int prof_begin(nb_sim *s, int *slot, bool record = true)
{
    *slot = -1;
    if (!(s->cfg.flags & NB_FLAG_PROFILE)) return NB_OK;
    if (!s->prof_init) {
        for (int i = 0; i < PROF_RING; ++i) {
            HIPCHK(hipEventCreate(&s->ev_start[i]));
            HIPCHK(hipEventCreate(&s->ev_stop[i]));
        }
        s->prof_init = true;
    }
    if (s->prof_count == PROF_RING) {   // drain
        HIPCHK(hipStreamSynchronize(s->stream));
        for (int i = 0; i < PROF_RING; ++i) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, s->ev_start[i], s->ev_stop[i]));
            s->prof_total_ms += ms;
        }
        s->prof_launches += PROF_RING;
        s->prof_count = 0;
    }
    *slot = s->prof_count++;
    if (record) HIPCHK(hipEventRecord(s->ev_start[*slot], s->stream));
    return NB_OK;
}
// events handed to a launcher that attaches them to the dispatch itself (no barrier packets on the stream)
NbKernelEvents prof_events(nb_sim *s, int slot)
{
    NbKernelEvents ev;
    if (slot >= 0) { ev.start = s->ev_start[slot]; ev.stop = s->ev_stop[slot]; }
    return ev;
}
int prof_end(nb_sim *s, int slot)
{
    if (slot >= 0) HIPCHK(hipEventRecord(s->ev_stop[slot], s->stream));
    return NB_OK;
}

// Which evaluations take the dtype-faithful generic kernel (nb_generic.hip): dtype chains no script of the
// reference builds but its stock class accepts.
bool use_generic(const nb_sim *s)
{
    const nb_config &c = s->cfg;
    if (grid_mode(c.mode) && mode_levels(c) > NB_MAX_LUT) return true;       // fused grids beyond the table capacity
    if (s->is_f64) {
        if (c.mode == NB_FLOAT64) return false;
        if (grid_mode(c.mode)) return true;                                  // grid over an fp64 (or fp64-stored) tensor
        return s->logical[0] != NB_F64;      // cast mode before the promotion: fp32 / half positions beside fp64 tensors
    }
    return grid_mode(c.mode) && is_half(s->logical[0]);                      // grid over a half tensor
}

int force_eval_generic(nb_sim *s, bool do_kick, bool *defer_kick, bool *open_next)
{
    const nb_config &c = s->cfg;
    const int64_t cnt = nd(s);
    const bool no_comm = (c.flags & NB_FLAG_NO_COMM) != 0;
    const bool multi = comm_active(s);
    if (multi && !s->comm) return fail(NB_ERR_COMM, "nranks > 1 but nb_comm_init was not called");
    const bool fq = force_quant_mode(c) && !(no_comm && c.nranks > 1);
    const int L = mode_levels(c);
    if (grid_mode(c.mode) && L < 2) return fail(NB_ERR_INVALID, "grid levels must be >= 2 (got %d)", L);
    const int A = acc_logical_dtype(s);
    if (!s->gen_scalars) HIPCHK(hipMalloc(&s->gen_scalars, nb_generic_scalars_bytes()));
    if (grid_mode(c.mode))       // every rank scans all pairs itself: no collective for the grid bounds
        HIPCHK(nb_launch_generic_r2max(s->pos, s->is_f64, c.n, c.dim, s->logical[0], c.softening_sq, s->gen_scalars, s->stream));
    HIPCHK(nb_launch_generic_force(s->pos, s->mass, s->is_f64, s->partial, s->geom, c.dim, s->logical[0], s->logical[2], c.mode,
                                   L, c.G, c.softening_sq, s->gen_scalars, s->acc, A, s->stream));
    s->last_kernel = "generic_force_kernel";
    s->last_generic = true;
    if (multi)
        if (int rc = comm_allreduce_sum(s, s->acc, (size_t)cnt, s->is_f64)) return rc;
    if (fq) {
        // quantize_force on a tensor of dtype A (quantization.py:74-88): linear grid over its global min / max
        const bool a64 = (A == NB_F64);
        if (a64 == s->is_f64) {
            HIPCHK(nb_launch_minmax_generic(s->acc, s->is_f64, cnt, 0, 0.0, s->scalars, s->scalars + 8, s->stream));
            HIPCHK(nb_launch_grid_quantize(s->acc, s->acc, s->is_f64, cnt, L, s->scalars, s->stream));
        } else {
            // fp32-typed forces held in fp64 storage: quantise in fp32 through the staging buffer
            HIPCHK(nb_launch_convert(s->acc, NB_F64, s->staging, NB_F32, cnt, s->stream));
            HIPCHK(nb_launch_minmax_generic(s->staging, 0, cnt, 0, 0.0, s->scalars, s->scalars + 8, s->stream));
            HIPCHK(nb_launch_grid_quantize(s->staging, s->staging, 0, cnt, L, s->scalars, s->stream));
            HIPCHK(nb_launch_convert(s->staging, NB_F32, s->acc, NB_F64, cnt, s->stream));
        }
    }
    if (open_next) *open_next = false;
    if (do_kick) {
        if (defer_kick) *defer_kick = true;
        else HIPCHK(nb_launch_axpy(s->vel, s->acc, c.dt / 2, cnt, s->is_f64, s->stream));
    }
    s->logical[3] = A;
    s->have_acc = true;
    return NB_OK;
}

}  // namespace

// one evaluation of simulation.py:74-118; optionally followed by the closing half kick (:141)
// defer_kick: the caller will apply the closing half kick itself (fused into the next step's
// opening launch) when this evaluation cannot fuse it into its reduction.
int force_eval(nb_sim *s, bool do_kick, bool packed_ready, bool *defer_kick, bool *open_next)
{
    if (!s->have_pos || !s->have_mass) return fail(NB_ERR_INVALID, "positions and masses must be set first");
    const nb_config &c = s->cfg;
    const int64_t cnt = nd(s);
    const double half_dt = c.dt / 2;
    const bool fq = force_quant_mode(c) && !((c.flags & NB_FLAG_NO_COMM) && c.nranks > 1);
    const bool no_comm = (c.flags & NB_FLAG_NO_COMM) != 0;
    // collectives run whenever a communicator exists (a 1-rank communicator exercises the same
    // RCCL calls on a single GPU) and must exist when the sources are really sharded
    const bool multi = (c.nranks > 1 && !no_comm) || s->comm != nullptr;
    if (multi && !s->comm) return fail(NB_ERR_COMM, "nranks > 1 but nb_comm_init was not called");
    if (multi)
        if (int rc = comm_check(s)) return rc;
    if (no_comm && c.nranks > 1 && do_kick) return fail(NB_ERR_INVALID, "NB_FLAG_NO_COMM handles cannot step");
    if (use_generic(s)) {
        if (s->req_open_on_read) return fail(NB_ERR_INVALID, "internal: speculative positions on the generic path");
        return force_eval_generic(s, do_kick, defer_kick, open_next);
    }
    s->last_generic = false;
    int slot;
    bool used_sym = false, sym_uniform = false;

    if (s->is_f64) {
        // (grid modes on fp64 storage and cast modes before the positions are promoted took the generic path above)
        int qhook = -1;                      // fp64 positions under a cast mode: hook output is fp32
        if (c.mode == NB_FLOAT32) qhook = HOOK_NONE;
        else if (c.mode == NB_BFLOAT16) qhook = HOOK_BF16;
        else if (c.mode == NB_FLOAT16) qhook = HOOK_F16;
        const int pair_dt = (qhook < 0 && s->logical[0] != NB_F64) ? s->logical[0] : -1;   // NB_F32 / F16 / BF16
        const int pa_f32 = (pair_dt == NB_F32);
        const bool sym_default_shape = s->sym.r == 4 || s->sym.r == 2;   // HOOK_F32PAIR instantiations
        used_sym = s->sym.enabled && qhook < 0 && (pair_dt < 0 || (pa_f32 && sym_default_shape));
        sym_uniform = s->mass_uniform;
        if (used_sym) {
            const auto &sp = s->sym;
            if (!packed_ready)
                HIPCHK(nb_launch_pack(s->pos, s->vel, s->acc, s->mass, sp.packed, c.n, sp.np, c.dim, 1, 0, 0.0, 0.0,
                                      c.G, pa_f32, s->stream));
            if (int rc = prof_begin(s, &slot, false)) return rc;
            HIPCHK(nb_launch_force_sym_f64((const double *)sp.packed, sp.work, sp.nwork, sp.rowslab,
                                           (double *)sp.colslab, sp.np, c.dim, sp.r, s->mass_uniform, pa_f32,
                                           c.softening_sq, s->stream, prof_events(s, slot), sp.rowsplit ? 1 : 0));
            s->last_kernel = "force_sym_kernel<double";
        } else {
            if (int rc = prof_begin(s, &slot)) return rc;
            HIPCHK(nb_launch_force_f64((const double *)s->pos, (const double *)s->mass, s->partial, s->geom, c.dim,
                                       pair_dt, qhook, c.G, c.softening_sq,
                                       (float)round_dt(pair_dt >= 0 ? pair_dt : NB_F32, c.softening_sq), s->stream));
            s->last_kernel = "force_f64_kernel";
            if (int rc = prof_end(s, slot)) return rc;
        }
    } else {
        int hook = HOOK_NONE;
        if (c.mode == NB_BFLOAT16) hook = HOOK_BF16;
        else if (c.mode == NB_FLOAT16) hook = HOOK_F16;
        else if (grid_mode(c.mode)) hook = HOOK_GRID;
        const int pa = is_half(s->logical[0]) ? s->logical[0] : NB_F32;   // half-typed positions (first evaluation)
        const float eps2 = (float)round_dt(pa, c.softening_sq);
        if (hook == HOOK_GRID) {
            const int L = mode_levels(c);
            if (L > NB_MAX_LUT || L < 2)
                return fail(NB_ERR_UNSUPPORTED, "grid levels must be in [2, %d] on the fused path (got %d)",
                            NB_MAX_LUT, L);
            // tab->r2max_bits is 0 here: zeroed at creation, put back by grid_tables_kernel after each use.
            // Small systems scan all pairs in one launch; the pruned search (six launches, O(N) + candidates^2)
            // pays off above that.
            // Tracked search: the farthest pair of the previous evaluation gives this one's lower bound, so two launches
            // (filter, scan + tables) replace six + one (round 3; exact either way, nb_force.hip).  Single GPU or every rank
            // redundantly; not for comm-less shards, whose first evaluation is their only one.
            const bool track = !s->knobs.no_prune && !s->knobs.no_track && c.n > g_tune.track_min_n && !(no_comm && c.nranks > 1);
            const bool prune = !s->knobs.no_prune && (c.n > g_tune.prune_min_n || track);
            if (track && s->prune_seeded) {
                HIPCHK(nb_launch_r2max_tracked((const float *)s->pos, c.n, c.dim, eps2, s->prune_cand, s->prune_idx, s->prune_state,
                                               s->tab, L, (float)c.G, 0.01f, s->knobs.no_grid_fast ? 0 : 1, s->stream));
                if (L > NB_LUT_MIN)      // multi-block tables: the scan left the maximum in tab->r2max_bits
                    HIPCHK(nb_launch_grid_tables(s->tab, L, (float)c.G, eps2, 0.01f, nullptr, s->stream,
                                                 s->knobs.no_grid_fast ? 0 : 1));
            } else {
            if (prune) {
                // every rank finds the global maximum itself: O(N) + (outer candidates)^2, no collective
                HIPCHK(nb_launch_r2max_pruned((const float *)s->pos, c.n, c.dim, eps2, s->prune_cand, s->prune_rho,
                                              s->prune_state, s->tab, s->stream));
            } else {
                ForceGeom gmax = s->geom;
                const bool scan_all = (no_comm && c.nranks > 1) || (multi && g_pc.direct_only);
                if (scan_all) {   // a comm-less shard (and a rank without RCCL's max) scans every source itself
                    gmax.j_begin = 0;
                    gmax.j_end = c.n;
                    gmax.nchunks = (c.n + gmax.chunk_len - 1) / gmax.chunk_len;
                }
                HIPCHK(nb_launch_r2max((const float *)s->pos, gmax, c.dim, eps2, s->tab, s->stream));
                if (multi && !scan_all)   // NB_FLAG_NO_COMM shards see only their own block's maximum
                    if (int rc = comm_allreduce_max_u32(s, &s->tab->r2max_bits)) return rc;
            }
            HIPCHK(nb_launch_grid_tables(s->tab, L, (float)c.G, eps2, 0.01f, prune ? s->prune_state : nullptr,
                                         s->stream, s->knobs.no_grid_fast ? 0 : 1));
            s->prune_seeded = prune;     // grid_tables_kernel seeded the tracked search from the pruned one's far pair
            }
        }
        used_sym = s->sym.enabled && pa == NB_F32;
        if (used_sym) {
            const auto &sp = s->sym;
            // grid LUT already carries G (simulation.py:101), so the packed factor is the bare mass there
            sym_uniform = s->mass_uniform && hook != HOOK_GRID;
            const double gfac = (hook == HOOK_GRID) ? 1.0 : (double)(float)c.G;
            if (!packed_ready)
                HIPCHK(nb_launch_pack(s->pos, s->vel, s->acc, s->mass, sp.packed, c.n, sp.np, c.dim, 0, 0, 0.0, 0.0,
                                      gfac, 0, s->stream));
            if (int rc = prof_begin(s, &slot, false)) return rc;
            // grid modes: the uniform kernel applies the common mass itself (reduce scale stays 1).  On the R = 2 tiling
            // (N < 20 480: a few hundred short work items, one wave per SIMD) a step is bound by the LATENCY of a sweep,
            // and the general-mass kernel's four independent scalar pairs per rotation step hide the log / exp chains
            // better than the packed uniform kernel does (measured INT8 / INT4 us per step, uniform vs general:
            // N = 6000 61.7 / 55.1 vs 48.3 / 47.6, N = 12 000 100 vs 88; N = 20 000 equal; N = 65 536 0.83 vs 1.24 ms)
            const bool grid_uniform = s->mass_uniform && sp.r != 2;
            if (s->bins_active && hook == HOOK_GRID)     // nb_quant_bin_sums: the same kernels, BINS = true
                HIPCHK(nb_launch_force_sym_f32_bins((const float *)sp.packed, sp.work, sp.nwork, sp.rowslab,
                                                    (float *)sp.colslab, sp.np, c.dim, sp.r, grid_uniform, eps2, s->tab,
                                                    (float)c.G, (float)s->mass_value, mode_levels(c), s->bin_out, c.n,
                                                    s->stream));
            else
            HIPCHK(nb_launch_force_sym_f32((const float *)sp.packed, sp.work, sp.nwork, sp.rowslab,
                                           (float *)sp.colslab, sp.np, c.dim, sp.r,
                                           hook == HOOK_GRID ? grid_uniform : sym_uniform, hook, eps2, s->tab,
                                           (float)c.G, (float)s->mass_value, hook == HOOK_GRID ? mode_levels(c) : 0,
                                           s->stream, prof_events(s, slot)));
            s->last_kernel = "force_sym_kernel<float";
        } else {
            if (int rc = prof_begin(s, &slot)) return rc;
            HIPCHK(nb_launch_force_f32((const float *)s->pos, (const float *)s->mass, s->partial, s->geom, c.dim, hook,
                                       pa, (float)c.G, eps2, s->tab, hook == HOOK_GRID ? mode_levels(c) : 0, s->stream,
                                       (s->bins_active && hook == HOOK_GRID) ? s->bin_out : nullptr));
            s->last_kernel = "force_f32_kernel";
            if (int rc = prof_end(s, slot)) return rc;
        }
    }

    const bool fuse_kick = do_kick && !multi && !fq;
    const bool want_open = do_kick && open_next && *open_next;   // nb_step asks: may this evaluation open the next step?
    bool opened = false;
    double x64_scale = 1.0;
    // INT8 / INT4 on one GPU, pair-symmetric path: the reduction hands quantize_force its min / max partials (one pair
    // per workgroup of 64 particles), saving the min/max launch (4.6 of 50 us per step at N = 6000)
    const int red_blocks = (c.n + 63) / 64;
    // (up to N = 32 768: beyond, every workgroup of the finish launch would fold thousands of partials -- measured
    // neutral to slightly negative at N = 65 536, where the launch it saves is 0.5 % of the step anyway)
    const bool red_mm = fq && used_sym && !multi && !s->is_f64 && red_blocks <= g_tune.red_mm_max_blocks && !s->knobs.no_red_mm;
    // multi-GPU: the rank's partial force vector goes straight into the buffer the peers read (direct xGMI
    // all-reduce), or into `acc` for the in-place RCCL all-reduce
    // multi-GPU INT8 / INT4 on the pair-symmetric path: the ranks exchange the UNROUNDED fp64 sums and round once,
    // (float)(sum * scale), exactly where the single-GPU reduction rounds, so the all-reduce itself adds no fp32
    // rounding of its own before quantize_force snaps the forces to their grid (a last-bit difference there is what
    // flips a force bin: measured against the single-GPU run after five steps at N = 9000 INT8, two ranks: positions
    // 1.2e-8 with the fp64 exchange, 1.2e-6 -- a flipped bin -- with fp32 partials).  Twice the bytes, so only where a
    // grid follows: the other fp32 modes differ across rank counts at the 1e-7 of their in-kernel fp32 running sums
    // either way (measured: identical with both exchanges).
    const bool x64 = multi && used_sym && !s->is_f64 && fq && !s->knobs.no_x64;
    bool p2p = multi && (x64 ? p2p_use_x64(s, cnt) : p2p_use(s, cnt));
    void *red_out = p2p ? nb_p2p_data() : s->acc;
    if (x64 && !p2p && !s->sums64) HIPCHK(hipMalloc((void **)&s->sums64, (size_t)cnt * sizeof(double)));
    double *sums64 = x64 ? (p2p ? (double *)nb_p2p_data() : s->sums64) : nullptr;
    if (p2p)
        if (int rc = p2p_claim_buffer(s)) return rc;
    if (used_sym) {
        const auto &sp = s->sym;
        // uniform-mass kernels leave out the mass factor: G*m in T arithmetic (fp32: (float)G * m)
        double scale = 1.0;
        if (sym_uniform) scale = s->is_f64 ? c.G * s->mass_value : (double)((float)c.G * (float)s->mass_value);
        // inside nb_step the reduction also opens the next step and repacks its positions
        const bool open = fuse_kick && want_open;
        int kmode = open ? 2 : (fuse_kick ? 1 : 0);
        // the last step of a native call leaves the NEXT step's drifted positions in pos_alt and `packed` (mode 3); a
        // call that starts from them applies its opening kick on read (bit 2) -- a Python loop of step() then costs
        // force + reduction per tick, no pack launch (see step_run)
        const bool spec = fuse_kick && !open && s->req_spec_next && !grid_mode(c.mode);
        if (spec) {
            if (!s->pos_alt) HIPCHK(hipMalloc(&s->pos_alt, (size_t)cnt * (s->is_f64 ? 8 : 4)));
            kmode = 3;
        }
        if (s->req_open_on_read) {
            if (!fuse_kick) return fail(NB_ERR_INVALID, "internal: a step started from speculative positions cannot fuse its kicks");
            kmode |= 4;
        }
        HIPCHK(nb_launch_reduce_sym(sp.rowslab, sp.colslab, sp.row_slot0, sp.row_nslots, sp.col_upto,
                                    sp.tile_b, c.n, sp.np, c.dim, s->is_f64, scale, red_out, s->vel, half_dt,
                                    kmode, s->pos, sp.packed, c.dt, s->stream, 0, -1, sums64,
                                    red_mm ? s->scalars + 8 : nullptr, s->pos_alt));
        if (spec) { s->spec_open = true; s->spec_kind = 2; s->spec_dt = c.dt; }
        x64_scale = scale;
        opened = open;
    } else {
        if (s->req_open_on_read) return fail(NB_ERR_INVALID, "internal: speculative positions on the one-sided path");
        // one-sided path inside nb_step: the reduction can also open the next step (one launch fewer per step,
        // which is what small systems are bound by)
        const bool open = fuse_kick && want_open;
        HIPCHK(nb_launch_reduce(s->partial, s->geom.nchunks, cnt, red_out, s->is_f64, s->vel, half_dt,
                                open ? 2 : (fuse_kick ? 1 : 0), s->pos, c.dt, s->stream));
        opened = open;
    }
    bool kicked = fuse_kick;
    if (p2p) {
        // every rank holds every summed element inside this kernel: the kicks (and, inside nb_step, the next step's
        // opening kick + drift + repack) ride along as they do in the single-GPU reduction -- no pack launch
        NbP2PKick kk{};
        kk.f64_to_f32 = x64 ? 1 : 0;
        kk.scale = x64_scale;
        if (do_kick && !fq && !s->knobs.no_p2p_kick) {
            const bool open = want_open;
            kk.mode = open ? 2 : 1;
            kk.dim = c.dim; kk.np = used_sym ? s->sym.np : 0;
            kk.vel = s->vel; kk.pos = s->pos; kk.packed = used_sym ? (void *)s->sym.packed : nullptr;
            kk.half_dt = half_dt; kk.dt = c.dt;
            kicked = true;
            opened = open;
        }
        HIPCHK(nb_p2p_allreduce(s->acc, (size_t)cnt, s->is_f64 || x64, p2p_step_timeout_s(), s->stream, &kk));
        s->used_p2p = true;
    } else if (multi && x64) {
        if (int rc = comm_allreduce_sum(s, s->sums64, (size_t)cnt, true)) return rc;
        const bool fin_kick = do_kick && !fq;
        const bool open = fin_kick && want_open;
        HIPCHK(nb_launch_finish_sums64(s->sums64, x64_scale, (float *)s->acc, (float *)s->vel, (float *)s->pos,
                                       (float *)s->sym.packed, c.n, s->sym.np, c.dim, fin_kick ? (open ? 2 : 1) : 0,
                                       half_dt, c.dt, s->stream));
        if (fin_kick) { kicked = true; opened = open; }
    } else if (multi) {
        if (int rc = comm_allreduce_sum(s, s->acc, (size_t)cnt, s->is_f64)) return rc;
    }
    if (fq) {
        // min/max of the summed forces, then quantisation with the closing kick (and, inside nb_step, the next
        // step's opening kick + drift) in the same launch
        const bool open = want_open;
        if (red_mm)
            HIPCHK(nb_launch_force_quant_finish((float *)s->acc, cnt, mode_levels(c), s->scalars + 8, red_blocks, s->scalars,
                                                s->fbins, (float *)s->vel, (float *)s->pos, half_dt, c.dt,
                                                do_kick ? (open ? 2 : 1) : 0, s->stream, (float *)s->sym.packed, s->sym.np,
                                                c.dim));
        else
        HIPCHK(nb_launch_force_quant_step((float *)s->acc, cnt, mode_levels(c), s->scalars, s->scalars + 8, s->fbins,
                                          (float *)s->vel, (float *)s->pos, half_dt, c.dt, do_kick ? (open ? 2 : 1) : 0,
                                          used_sym ? (float *)s->sym.packed : nullptr, s->sym.np, c.dim, s->stream));
        kicked = do_kick;
        opened = open;
    }
    if (open_next) *open_next = opened;
    if (do_kick && !kicked) {
        if (defer_kick) *defer_kick = true;
        else HIPCHK(nb_launch_axpy(s->vel, s->acc, half_dt, cnt, s->is_f64, s->stream));
    }
    s->logical[3] = acc_logical_dtype(s);
    s->have_acc = true;
    return NB_OK;
}

namespace {

// ---- small systems: one launch per step (nb_small.hip) ---------------------------------------------------------
bool small_ok(const nb_sim *s)
{
    const nb_config &c = s->cfg;
    const int sdt = s->is_f64 ? NB_F64 : NB_F32;
    // fp32 storage: above 3072 the tiled path is ahead (measured us per step with 512 x 64 workgroups, one launch vs
    // tiled: FLOAT32 N = 3300 13.2 / 10.8, 3584 13.5 / 11.0, 4096 14.7 / 13.3; INT8 3584 33.2 / 35.7, 4096 35.3 / 36.5 and
    // INT4 3840 37.2 / 34.3, 4096 42.3 / 43.8 -- within the run-to-run spread of the max-r2 search; up to 3072 one
    // launch wins or ties everywhere); fp64 keeps it to 4096 (15.7 against 18.9)
    const int nmax = s->knobs.small_max > 0 ? s->knobs.small_max : (s->is_f64 ? g_tune.small_max_f64 : g_tune.small_max_f32);
    if (s->knobs.no_smalln || c.n > nmax || comm_active(s) || c.nranks != 1 || !s->have_acc) return false;
    if (grid_mode(c.mode) && (s->is_f64 || mode_levels(c) > NB_LUT_MIN || mode_levels(c) < 2)) return false;
    if (s->is_f64 != (c.mode == NB_FLOAT64)) return false;           // fp64 state under a cast mode: tuned one-sided kernel
    // masses: fp32-typed masses in an fp64 run enter the fp64 product exactly (no rounding of their own); half-typed
    // masses round the product to the half type and stay on the tuned kernels
    const bool mass_ok = s->logical[2] == sdt || (s->is_f64 && s->logical[2] == NB_F32);
    return s->logical[0] == sdt && s->logical[1] == sdt && mass_ok && s->logical[3] == sdt;
}

// this evaluation's grid on a small system: all-pairs max of r2 and the threshold / factor tables -- one launch for
// both up to N = 2048 (measured, INT4: N = 1024 22.5 -> 18.6 us per step; N = 3000 30.8 vs 31.7: there the fused
// kernel's arrival counter and longer source chunks cost more than the launch)
int small_grid_tables(nb_sim *s)
{
    const nb_config &c = s->cfg;
    const float eps2f = (float)c.softening_sq;
    if (s->prune_seeded && !s->knobs.no_prune && !s->knobs.no_track && c.n > g_tune.track_min_n) {
        // the first evaluation (tiled path) seeded the tracked search: filter + scan (+ tables) instead of an all-pairs pass
        HIPCHK(nb_launch_r2max_tracked((const float *)s->pos, c.n, c.dim, eps2f, s->prune_cand, s->prune_idx, s->prune_state,
                                       s->tab, mode_levels(c), (float)c.G, 0.01f, s->knobs.no_grid_fast ? 0 : 1, s->stream));
    } else if (mode_levels(c) <= NB_LUT_MIN && c.n <= g_tune.small_fuse_tables_max_n && !s->knobs.no_small_fuse) {
        HIPCHK(nb_launch_r2max_tables((const float *)s->pos, s->geom, c.dim, eps2f, s->tab, mode_levels(c), (float)c.G, 0.01f,
                                      s->knobs.no_grid_fast ? 0 : 1, s->stream));
    } else {
        HIPCHK(nb_launch_r2max((const float *)s->pos, s->geom, c.dim, eps2f, s->tab, s->stream));
        HIPCHK(nb_launch_grid_tables(s->tab, mode_levels(c), (float)c.G, eps2f, 0.01f, nullptr, s->stream,
                                     s->knobs.no_grid_fast ? 0 : 1));
    }
    return NB_OK;
}

// the remaining `nsteps` steps of an nb_step call; `opened`: this step's opening kick + drift was already applied
int step_small(nb_sim *s, int nsteps, bool opened)
{
    const nb_config &c = s->cfg;
    const size_t el = s->is_f64 ? 8 : 4;
    const bool grid = grid_mode(c.mode);
    const bool fq = force_quant_mode(c);
    if (!s->pos_alt) HIPCHK(hipMalloc(&s->pos_alt, (size_t)nd(s) * el));
    if (fq && !s->small_part) HIPCHK(hipMalloc((void **)&s->small_part, 2 * (size_t)c.n * sizeof(double)));
    const int hook = grid ? HOOK_GRID : (c.mode == NB_BFLOAT16 ? HOOK_BF16 : (c.mode == NB_FLOAT16 ? HOOK_F16 : HOOK_NONE));
    const int lanes = s->knobs.small_lanes ? s->knobs.small_lanes : nb_small_lanes(c.n);
    // A step() loop driven from Python is one nb_step(1) per tick: the last step of a call leaves the next step's
    // drifted positions in pos_alt (kick mode 3); if nothing wrote state or dt since, this call takes them and applies
    // its opening kick on read -- one launch per tick instead of two (FLOAT32 us per step() call: N = 1024 9.9 -> 5.6, N = 3000 12.6 -> 10.8;
    // profiles/r03_python_step_overhead.txt)
    const bool speculate = !grid && !fq && !s->knobs.no_spec;
    bool open_on_read = false;
    if (!opened) {
        if (speculate && s->spec_open && s->spec_kind == 1 && s->spec_dt == c.dt) {
            std::swap(s->pos, s->pos_alt);
            open_on_read = true;
        } else {
            HIPCHK(nb_launch_kick_drift(s->pos, s->vel, s->acc, c.dt / 2, c.dt, nd(s), s->is_f64, s->stream));
        }
    }
    s->spec_open = false;
    for (int t = 0; t < nsteps; ++t) {
        const bool last = (t + 1 == nsteps);
        if (grid)
            if (int rc = small_grid_tables(s)) return rc;
        // INT8 / INT4: the forces are snapped to their grid (and the kicks applied) by the finish launch
        const int kick = fq ? 0 : ((last ? (speculate ? 3 : 1) : 2) | ((t == 0 && open_on_read) ? 4 : 0));
        int slot;
        if (int rc = prof_begin(s, &slot)) return rc;
        HIPCHK(nb_launch_small_step(s->pos, s->pos_alt, s->vel, s->acc, s->mass, c.n, c.dim, s->is_f64, hook, c.G,
                                    c.softening_sq, c.dt / 2, c.dt, kick, lanes, s->stream, grid ? s->tab : nullptr,
                                    fq ? s->small_part : nullptr));
        if (int rc = prof_end(s, slot)) return rc;
        if (fq)      // one min / max pair per workgroup of the force launch
            HIPCHK(nb_launch_force_quant_finish((float *)s->acc, nd(s), mode_levels(c), s->small_part,
                                                (c.n + nb_small_block(c.n) / lanes - 1) / (nb_small_block(c.n) / lanes), s->scalars, s->fbins,
                                                (float *)s->vel, (float *)s->pos, c.dt / 2, c.dt, last ? 1 : 2, s->stream));
        else if (!last)
            std::swap(s->pos, s->pos_alt);
    }
    if (speculate) { s->spec_open = true; s->spec_kind = 1; s->spec_dt = c.dt; }
    s->last_kernel = "small_step_kernel";
    s->last_generic = false;
    return NB_OK;
}

}  // namespace

// `nsteps` leapfrog steps (simulation.py:120-143): v += a dt/2; x += v dt; a = force(x); v += a dt/2.
int step_run(nb_sim *s, int nsteps)
{
    bool pending_close = false;     // closing kick of the previous step still to be applied
    bool opened = false;            // the previous step's reduction already did this step's opening kick + drift
    bool packed_by_prev = false;    // ... and repacked the positions for the symmetric kernel
    bool open_on_read = false;      // this call starts from the previous call's speculative positions (tiled path)
    for (int t = 0; t < nsteps; ++t) {
        // small systems with settled dtypes: one launch per step
        if (!pending_close && small_ok(s)) return step_small(s, nsteps - t, opened);
        if (t == 0 && s->spec_open && s->spec_kind == 2 && s->spec_dt == s->cfg.dt && s->sym.enabled && !comm_active(s) &&
            !force_quant_mode(s->cfg) && !grid_mode(s->cfg.mode) && s->pos_alt) {
            const int sd = s->is_f64 ? NB_F64 : NB_F32;
            if (s->logical[0] == sd && s->logical[1] == sd && s->logical[3] == sd) {
                std::swap(s->pos, s->pos_alt);        // positions after this step's drift; `packed` holds them as well
                opened = packed_by_prev = open_on_read = true;
            }
        }
        s->spec_open = false;
        // opening kick + drift; on the pair-symmetric path the repack rides in the same launch
        const int sdt = s->is_f64 ? NB_F64 : NB_F32;
        const bool fuse_pack = s->sym.enabled && s->logical[0] == sdt && s->logical[1] == sdt &&
                               s->logical[3] == sdt && !grid_mode(s->cfg.mode);
        const bool uniform_dt = s->logical[0] == sdt && s->logical[1] == sdt && s->logical[3] == sdt;
        if (opened) {
            // nothing to launch: positions and velocities were advanced by the previous reduction
        } else if (fuse_pack) {
            HIPCHK(nb_launch_pack(s->pos, s->vel, s->acc, s->mass, s->sym.packed, s->cfg.n, s->sym.np, s->cfg.dim,
                                  s->is_f64, pending_close ? 2 : 1, s->cfg.dt / 2, s->cfg.dt,
                                  s->is_f64 ? s->cfg.G : (double)(float)s->cfg.G, 0, s->stream));
        } else {
            if (pending_close) HIPCHK(nb_launch_axpy(s->vel, s->acc, s->cfg.dt / 2, nd(s), s->is_f64, s->stream));
            HIPCHK(nb_launch_kick_drift(s->pos, s->vel, s->acc, s->cfg.dt / 2, s->cfg.dt, nd(s), s->is_f64, s->stream));
        }
        // packed positions are current when this step's pack launch wrote them, or when the previous evaluation
        // opened this step on the symmetric path (its reduction / quantisation repacked them)
        const bool packed_ready = opened ? packed_by_prev : fuse_pack;
        pending_close = false;
        s->logical[1] = promote(s->logical[1], s->logical[3]);
        s->logical[0] = promote(s->logical[0], s->logical[1]);
        // a closing kick that cannot ride in the reduction (RCCL all-reduce / force quantisation in
        // between) is folded into the next step's opening launch when there is one
        const bool may_defer = (t + 1 < nsteps) && fuse_pack;
        opened = (t + 1 < nsteps) && uniform_dt;      // request; force_eval answers
        s->req_open_on_read = (t == 0) && open_on_read;
        s->req_spec_next = (t + 1 == nsteps) && uniform_dt && s->sym.enabled && !s->knobs.no_spec;
        const int rc_eval = force_eval(s, true, packed_ready, may_defer ? &pending_close : nullptr, &opened);
        s->req_open_on_read = s->req_spec_next = false;
        if (rc_eval) return rc_eval;
        packed_by_prev = opened && s->sym.enabled;
        s->logical[1] = promote(s->logical[1], s->logical[3]);
    }
    return NB_OK;
}

int energy_eval(nb_sim *s, double *kinetic, double *potential)
{
    const nb_config &c = s->cfg;
    double host[2] = {0, 0};
    bool pe_uniform = false;
    const int hp_v = is_half(s->logical[1]) ? s->logical[1] : -1;   // NB_F16 == 0: "none" is -1
    const int hp_x = is_half(s->logical[0]) ? s->logical[0] : -1;
    if (kinetic) {
        if (!s->have_vel || !s->have_mass) return fail(NB_ERR_INVALID, "velocities/masses not set");
        HIPCHK(nb_launch_kinetic(s->vel, s->mass, c.n, c.dim, s->is_f64, s->logical[1] != NB_F64, hp_v, s->scratch,
                                 s->scalars + 2, s->stream));
    }
    if (potential) {
        if (!s->have_pos || !s->have_mass) return fail(NB_ERR_INVALID, "positions/masses not set");
        const auto &sp = s->sym;
        const bool pe_sym = sp.enabled && hp_x < 0 && (sp.r == 2 || sp.r == 4) && (size_t)sp.nwork <= s->scratch_elems &&
                            !s->knobs.no_pe_sym;
        if (pe_sym) {
            // same tile-pair work list as the force kernel; `packed` is scratch between force evaluations
            // uniform masses: no mass factor in the pair loop; m * m (rounded like upstream's masses[i] * masses[j], in the
            // masses' dtype) multiplies the finished sum below
            pe_uniform = s->mass_uniform;
            if (s->spec_kind == 2) s->spec_open = false;      // `packed` (the speculative next positions with it) is rewritten here
            HIPCHK(nb_launch_pack(s->pos, s->vel, s->acc, s->mass, sp.packed, c.n, sp.np, c.dim, s->is_f64, 0, 0.0, 0.0,
                                  1.0, s->is_f64 && s->logical[0] != NB_F64, s->stream, 0, -1, pe_uniform ? 1 : 0));
            HIPCHK(nb_launch_potential_sym(sp.packed, sp.work, sp.nwork, s->scratch, sp.np, c.dim, sp.r, s->is_f64,
                                           s->logical[0] != NB_F64, s->logical[2], c.softening_sq, pe_uniform ? 1 : 0,
                                           s->stream));
            HIPCHK(nb_launch_final_sum(s->scratch, sp.nwork, s->scalars + 3, s->stream));
        } else {
            HIPCHK(nb_launch_potential(s->pos, s->mass, s->geom, c.dim, s->is_f64, s->logical[0] != NB_F64,
                                       s->logical[2], hp_x, c.softening_sq,
                                       (float)round_dt(hp_x >= 0 ? hp_x : NB_F32, c.softening_sq), s->scratch,
                                       s->scalars + 3, s->stream));
        }
        if ((c.nranks > 1 && !(c.flags & NB_FLAG_NO_COMM)) || s->comm) {
            if (!s->comm) return fail(NB_ERR_COMM, "nranks > 1 but nb_comm_init was not called");
            if (int rc = comm_allreduce_sum(s, s->scalars + 3, 1, true)) return rc;
        }
    }
    HIPCHK(hipMemcpyAsync(host, s->scalars + 2, 2 * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (int rc = p2p_check(s)) return rc;
    if (kinetic) {
        // ke = 0.5 * (masses * v_sq).sum() in the promoted dtype of (velocities, masses)
        const int t = promote(s->logical[1], s->logical[2]);
        *kinetic = round_dt(t, round_dt(t, 0.5) * round_dt(t, host[0]));
    }
    if (potential) {
        const int t = promote(s->logical[0], s->logical[2]);
        if (pe_uniform) host[1] *= round_dt(s->logical[2], s->mass_value * s->mass_value);
        // -G * sum: a Python scalar MULTIPLYING a half tensor stays in float (torch opmath; only added scalars are
        // rounded to the tensor's dtype first)
        *potential = round_dt(t, round_dt(is_half(t) ? NB_F32 : t, -c.G) * round_dt(t, host[1]));
        // the reference multiplies by the triu mask before dividing by dist (simulation.py:189): the masked
        // entries are 0 / dist = NaN where dist == 0, i.e. on the whole diagonal when the softening rounds to
        // zero in the positions' dtype (softening 0; 1e-4 with float16 positions).  dist > 0 otherwise.
        if (c.n > 0 && round_dt(s->logical[0], c.softening_sq) == 0.0) *potential = std::nan("");
    }
    return NB_OK;
}

// Quant-bin read-out (nb_quant_bin_sums).  Runs the evaluation once more on the CURRENT positions with the BINS
// instantiation of whichever grid-mode pair loop the production path uses here -- same launch sequence (max-r2, tables,
// uniform / general pair of launches, reduction, force quantisation: the forces are simply recomputed), same control
// flow inside the pair loop -- and returns, per particle p, s1 = sum_q k(p, q) and s2 = sum_q k(p, q) ((q mod 65521) + 1)
// over ALL q including p itself (the reference's N x N bin matrix has k = 0 on the diagonal).
int bin_sums_eval(nb_sim *s, int which, int64_t *sum_k, int64_t *sum_kw, double info[8])
{
    const nb_config &c = s->cfg;
    if (s->comm) return fail(NB_ERR_UNSUPPORTED, "quant-bin read-out: not on a handle with a communicator (single GPU or NB_FLAG_NO_COMM shards)");
    if (use_generic(s)) return fail(NB_ERR_UNSUPPORTED, "quant-bin read-out: this evaluation runs on the generic per-pair path");
    const int L = mode_levels(c);
    if (L < 2 || L > NB_MAX_LUT) return fail(NB_ERR_UNSUPPORTED, "grid levels must be in [2, %d] on the fused path (got %d)", NB_MAX_LUT, L);
    const size_t words = 2 * (size_t)c.n + 2;
    if (!s->bin_out) HIPCHK(hipMalloc((void **)&s->bin_out, words * sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(s->bin_out, 0, words * sizeof(unsigned long long), s->stream));
    const bool last_small = strcmp(s->last_kernel, "small_step_kernel") == 0;
    const bool want_small = which == 2 || (which == 0 && last_small);
    int path = 0, shape = 0;
    if (want_small) {
        if (!small_ok(s)) return fail(NB_ERR_INVALID, "quant-bin read-out: the one-launch small-system step does not apply to this handle "
                                                      "(size, dtypes, or no step taken yet)");
        // what step_small launches for one step, force only (do_kick = 0), forces into the staging buffer
        const int lanes = s->knobs.small_lanes ? s->knobs.small_lanes : nb_small_lanes(c.n);
        if (!s->pos_alt) HIPCHK(hipMalloc(&s->pos_alt, (size_t)nd(s) * 4));
        if (int rc = small_grid_tables(s)) return rc;
        HIPCHK(nb_launch_small_step(s->pos, s->pos_alt, s->vel, s->staging, s->mass, c.n, c.dim, 0, HOOK_GRID, c.G, c.softening_sq,
                                    c.dt / 2, c.dt, 0, lanes, s->stream, s->tab, nullptr, s->bin_out));
        path = 3;
        shape = lanes;
    } else {
        s->bins_active = true;
        const int rc = force_eval(s, false);
        s->bins_active = false;
        if (rc) return rc;
        const bool sym = strncmp(s->last_kernel, "force_sym_kernel", 16) == 0;
        path = sym ? 1 : 2;
        shape = sym ? s->sym.r : 0;
        if (last_small) s->last_kernel = "small_step_kernel";       // the read-out does not change what the steps run on
    }
    std::vector<unsigned long long> host(words);
    GridTables *ht = new GridTables;
    hipError_t e = hipMemcpyAsync(host.data(), s->bin_out, words * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ht, s->tab, sizeof(GridTables), hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    const int fast_ok = ht->fast_ok, degenerate = ht->degenerate;
    delete ht;
    HIPCHK(e);
    if (degenerate) return fail(NB_ERR_UNSUPPORTED, "quant-bin read-out: degenerate grid (lmax - lmin < 1e-10): values pass through, no bins");
    for (int i = 0; i < c.n; ++i) { sum_k[i] = (int64_t)host[i]; sum_kw[i] = (int64_t)host[(size_t)c.n + i]; }
    if (info) {
        info[0] = path;                   // 1 pair-symmetric tiles, 2 one-sided tiles, 3 one-launch small-system kernel
        info[1] = shape;                  // targets per lane (1) / lanes per target (3)
        // the uniform-mass packed kernel did the work (pair-symmetric path; the same rule force_eval applies)
        info[2] = (path == 1 && s->mass_uniform && s->sym.r != 2) ? 1 : 0;
        info[3] = fast_ok;                // the tables enabled the table-free pair path
        info[4] = (double)host[2 * (size_t)c.n];        // pair evaluations binned by the table-free estimate alone
        info[5] = (double)host[2 * (size_t)c.n + 1];    // pair evaluations binned through a threshold table
        info[6] = L;
        info[7] = 0;
    }
    return NB_OK;
}

int prof_collect(nb_sim *s, double *total_ms, int32_t *launches)
{
    HIPCHK(hipStreamSynchronize(s->stream));
    for (int i = 0; i < s->prof_count; ++i) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, s->ev_start[i], s->ev_stop[i]));
        s->prof_total_ms += ms;
    }
    s->prof_launches += s->prof_count;
    s->prof_count = 0;
    if (total_ms) *total_ms = s->prof_total_ms;
    if (launches) *launches = s->prof_launches;
    s->prof_total_ms = 0;
    s->prof_launches = 0;
    return NB_OK;
}

}  // namespace nbhost
