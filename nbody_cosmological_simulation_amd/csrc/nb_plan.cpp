// nb_plan.cpp -- work list of the pair-symmetric kernels (see nb_plan.h).  No HIP calls in this file.
#include "nb_plan.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace {
int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
}  // namespace

NbKnobs nb_read_knobs()
{
    NbKnobs k;
    k.sym = env_int("NB_SYM", -1);
    k.sym_r = env_int("NB_SYM_R", 0);
    k.sym_cl = std::max(0, env_int("NB_SYM_CL", 0));
    k.sym_split = env_int("NB_SYM_SPLIT", 0);
    if (k.sym_split < 1 || k.sym_split > 16) k.sym_split = 0;
    k.sym_rowsplit = env_int("NB_SYM_ROWSPLIT", -1);
    if (k.sym_rowsplit > 8) k.sym_rowsplit = -1;
    k.tail_pieces = env_int("NB_SYM_TAIL", 0);
    if (k.tail_pieces != 2 && k.tail_pieces != 4 && k.tail_pieces != 8 && k.tail_pieces != 16) k.tail_pieces = 0;
    k.r_onesided = env_int("NB_R", 0);
    k.no_prune = getenv("NB_NO_PRUNE") != nullptr;
    k.no_track = env_int("NB_NO_TRACK", 0) != 0;
    k.no_pe_sym = getenv("NB_NO_PE_SYM") != nullptr;
    k.no_uniform = getenv("NB_NO_UNIFORM") != nullptr;
    k.no_smalln = getenv("NB_NO_SMALLN") != nullptr;
    k.no_grid_fast = getenv("NB_NO_GRID_FAST") != nullptr;
    k.no_p2p = getenv("NB_NO_P2P") != nullptr;
    k.no_p2p_kick = getenv("NB_P2P_NO_KICK") != nullptr;
    k.no_small_fuse = getenv("NB_NO_SMALL_FUSE") != nullptr;
    k.no_spec = getenv("NB_NO_SPEC") != nullptr;
    k.no_x64 = getenv("NB_NO_X64") != nullptr;
    k.no_red_mm = getenv("NB_NO_RED_MM") != nullptr;
    k.small_max = std::max(0, env_int("NB_SMALL_MAX", 0));
    k.small_lanes = env_int("NB_SMALL_LANES", 0);
    if (k.small_lanes != 16 && k.small_lanes != 32 && k.small_lanes != 64) k.small_lanes = 0;
    return k;
}

// Rows (target tiles) are grouped into super-rows of four (one per wave of a workgroup); super-rows are
// dealt to the ranks in a snake pattern so every rank owns the same number of tile pairs to within one
// super-row; each owned super-row is cut into work items of `cl` source tiles = one workgroup each.
void nb_plan_sym(const PlanInput &in, const NbKnobs &knobs, SymPlanHost &sp)
{
    sp = SymPlanHost();
    const int n = in.n, dim = in.dim, P = in.nranks;
    const bool is_f64 = in.is_f64;
    // Small systems: the pair-symmetric kernel has too few work items to fill the chip (one 64-step sweep per
    // wave is its floor: ~50 us per step with tiles of 256 at any N <= 8192) and the one-sided LDS kernel wins.
    // Between the two regimes a finer tiling (R = 2: tiles of 128, four times the work items, sweeps a quarter
    // as long) fills the chip earlier: measured fp64 us per step, one-sided / R = 2 / R = 4: N = 6144 38.9 / 30.8 /
    // 51.5, 8192 47.4 / 44.5 / 52.1, 12288 113 / 66.5 / 81.0, 16384 149 / 107 / 118, 20480 - / 154 / 152, 32768 - /
    // 366 / 310; fp32: N = 4096 18.6 / 16.1 / 37.0, 8192 40.2 / 26.4 / 38.3, 16384 118 / 61.4 / 70.8, 24576 269 / 118 / 105.
    // And at the very small end (fp64, 2-D) tiles of 64 (R = 1: one 64-step sweep of single pairs per wave, ~4 us)
    // beat the one-sided kernel: 12.1 vs 16.6 us per step at N = 1024, 13.1 vs 18.3 at 2560, 18.1 vs 18.7 at 3000.
    const bool tiny = is_f64 && dim == 2 && n <= 2816;
    // fp32 family: R = 2 already wins at N = 1024 (14.2 vs 15.4 us; INT4 51 vs 62).  3-D crosses over later
    // (fp64: 8192 56.9 one-sided vs 60.5, 12288 131 vs 92.6; fp32: 3000 19.8 vs 20.0, 8192 52.3 vs 36.0).
    // round 2: with the sweeps of mid-sized systems cut into pieces (below) the 3-D fp64 crossover moves down to 8192
    // (50.5 us per step against 57 one-sided)
    const int sym_from = dim == 3 ? (is_f64 ? 8192 : 4096) : (is_f64 ? 5120 : 1024);
    int want = (tiny || n >= sym_from) ? 1 : 0;
    if (knobs.sym >= 0) want = knobs.sym;
    // comm-less shards (NB_FLAG_NO_COMM) use the one-sided kernel unless NB_SYM=2 asks for the symmetric
    // plan of their rank (tests: the partial sums of all ranks' plans must add up to the full result)
    if (!want || ((in.flags & NB_FLAG_NO_COMM) && want < 2)) return;
    if (is_f64 && in.mode != NB_FLOAT64) return;    // fp64 state under a cast / grid mode: one-sided kernel
    // targets per lane.  Measured on MI355X, N=65536, D=2: R=1 3.39 ms, R=2 1.87 ms, R=4 1.47 ms.  D=3 also keeps
    // four targets per lane and sweeps the source tile in two halves of two slots (sym_rj, nb_force_sym.hip)
    // fp64 mid sizes (5120 ... 20 479) also take R = 4 since round 2: 16 pairs per rotation step hide the rotation's
    // latency better than 4, and the missing parallelism comes from cutting every sweep into pieces instead (us per
    // step, R = 2 whole sweeps -> R = 4 in four pieces: N = 8192 44.3 -> 39.1, 12 288 68.8 -> 60.9, 16 384 113.5 -> 96.7;
    // 3-D: 12 288 96.4 -> 76.2, 16 384 158 -> 124).  fp32 keeps R = 2 there (R = 4 in pieces: 41.5 vs 40.3 at 12 288).
    sp.r = tiny ? 1 : (n >= 20480 || is_f64) ? 4 : 2;
    if ((knobs.sym_r == 1 && is_f64 && dim == 2) || knobs.sym_r == 2 || knobs.sym_r == 4) sp.r = knobs.sym_r;
    sp.tile_b = 64 * sp.r;
    const int T = (n + sp.tile_b - 1) / sp.tile_b;            // tiles that hold particles
    const int SR = (T + 3) / 4;                               // super-rows of four target tiles
    sp.tiles = SR * 4;                                        // padded tile count
    sp.np = sp.tiles * sp.tile_b;
    // The choice between this plan and the one-sided source blocks must be the same on every rank (their
    // partial sums are added): it may only depend on rank-independent quantities.  Fewer super-rows than
    // ranks would leave a rank without work; the slab budget is checked for the worst case (every owned
    // super-row cut into eight pieces -- the most the plan itself chooses; NB_SYM_SPLIT is budgeted by its own value).
    const size_t el = is_f64 ? sizeof(double) : sizeof(float);
    if (SR < P) return;
    {
        const long long rows_max = (SR + P - 1) / P;            // owned super-rows of the busiest rank
        const long long entries_max = knobs.sym_split ? rows_max * knobs.sym_split
                                                      : rows_max + 7 * std::min<long long>(rows_max, 2LL * in.cus);
        if ((size_t)dim * sp.np * el * (size_t)entries_max > (size_t)48 << 30) return;
    }

    std::vector<int> ord(SR, -1);
    sp.row_slot0.assign(sp.tiles, 0);
    sp.row_nslots.assign(sp.tiles, 0);
    long long owned_pairs = 0;
    int nrows = 0;
    for (int S = 0; S < SR; ++S) {
        const int k = S % (2 * P);
        const int owner = k < P ? k : 2 * P - 1 - k;           // snake: equal pair counts per rank
        if (owner == in.rank) {
            ord[S] = nrows++;
            for (int w = 0; w < 4; ++w) owned_pairs += std::max(0, T - (4 * S + w));
        }
    }
    // A workgroup sweeps 4 rows x cl source tiles.  ~2000 workgroups per launch: with tail smoothing
    // (below) measured at N=65536 (R=4): cl = 1/2/4/6/8/16 -> step 1.28/1.28/1.27/1.29/1.29/1.47 ms
    // (small cl pays in row-slot traffic, large cl in load balance).
    // The fp32 kernels (items half as long as fp64 ones; the grid-mode kernels also copy their tables per workgroup) want
    // twice the work items: measured per step at N = 65536, cl = 1 / 2 / 4: INT8 784 / 783 / 810 us, INT4 756 / 752 / 773,
    // CUSTOM 750 / 747 / 770, FLOAT32 514 / 504 / 506 (3-D force launch: 0.671 / 0.672 / 0.684 ms); FLOAT64 1205 / 1198 / 1198.
    const long long target_items = is_f64 ? 2048 : 4096;
    int cl = (int)(owned_pairs / target_items / 4);
    cl = std::max(1, std::min(cl, 16));
    if (knobs.sym_cl > 0) cl = knobs.sym_cl;
    sp.cl = cl;

    // ---- tail smoothing -------------------------------------------------------------------------------
    // Work items (4 rows x cl source tiles) all take the same time and the chip runs `slots` workgroups at
    // once (4 per CU at <= 128 VGPRs), so items beyond a multiple of `slots` cost a whole extra round on a
    // few CUs (measured: 1040 items on 1024 slots -> 0.21 instead of 0.16 ms).  The sweeps of just enough
    // trailing super-rows are therefore cut into pieces of 16 (or 8) rotation steps (a piece starts with the
    // source tile pre-rotated, see force_sym_kernel); only those super-rows pay the extra slab / slot traffic.
    const long long wg_slots = 4LL * in.cus;
    std::vector<int> nch_of(SR, 0), split_of(SR, 1);
    long long items = 0;
    for (int S = 0; S < SR; ++S)
        if (ord[S] >= 0) { nch_of[S] = (T - 4 * S + cl - 1) / cl; items += nch_of[S]; }
    // (Round 3 also tried DECREASING chunks towards the end of the work list -- cl / 2, cl / 4 source tiles, then half and
    // quarter sweeps for the trailing super-rows -- because a CU drains through 3, 2, 1 resident waves per SIMD when the
    // list runs dry.  The workgroup trace confirmed full residency until 1.05 of 1.15 ms instead of 0.75, and the launch
    // was no faster: fp64 N = 65 536 1203.7 -> 1204.9 us per step, 131 072 4817 -> 4771, FLOAT32 503.6 -> 509.0, INT8 785.1 ->
    // 782.3 (profiles/r03_guided_chunks_ab.txt).  One or two fp64 waves keep a SIMD's issue port nearly as busy as four.)
    int rowsplit = 0;                     // > 0: row-split work items with this many step pieces per (row, source chunk)
    if (knobs.sym_split) {
        for (int S = 0; S < SR; ++S) split_of[S] = knobs.sym_split;
    } else if (sp.r == 4 && items < 1000) {
        // Mid-sized systems on the R = 4 tiling: too few work items to fill 1024 workgroup slots with four waves per SIMD,
        // and a whole sweep (64 steps x 16 pairs) is long -- every sweep is cut into pieces (nsp need not divide 64).
        // Where the time goes (workgroup trace of the force launch, tools/wg_trace.py, fp64 N = 8192, 144 items): the
        // dispatcher spreads workgroups evenly over the CUs and a SIMD serves its resident waves oldest first, so the
        // launch lasts rounds = ceil(workgroups / CUs) pieces back to back on the fullest CUs (4 pieces: 576 workgroups,
        // 64 CUs hold three -> 31.7 us; 3 pieces: 432, two per CU -> 29.4 us); a SIMD with one resident wave issues a
        // rotation step of 16 pairs in ~0.7 us, with four in 0.55 us; and every piece costs the reduction one more row
        // slot and slab entry to read (reduction + launch gap: 6.3 / 7.9 / 9.8 / 13.2 us with 3 / 4 / 5 / 7 pieces).
        // The pieces per sweep minimise that model; its choice is within 3 % of the best measured one at every size
        // of the sweep behind it (profiles/r03_mid_split_sweep.txt; fp64 us per step, pieces = 4 -> chosen: N = 5632
        // 28.0 -> 24.2, 6144 28.7 -> 25.0, 8192 39.3 -> 35.9, 10 240 50.6 -> 47.2, 14 336 79.5 -> 76.9).
        int pieces = 4;
        {
            const double t_step[4] = {0.70, 0.60, 0.62, 0.55};      // us per rotation step with 1 .. 4 resident waves per SIMD
            double best = 1e30;
            for (int nsp : {2, 3, 4, 5, 6, 8}) {
                if (items >= 500 && nsp != 4) continue;             // (the sweep behind the model ends there: four pieces)
                const long long wgs = items * nsp, rounds = (wgs + in.cus - 1) / in.cus;
                const double cost = rounds * ((64 + nsp - 1) / nsp + 1) * t_step[std::min<long long>(rounds, 4) - 1] +
                                    0.3 * 5.0 * (double)wgs / in.cus;          // 4 row slots + 1 slab entry per workgroup
                if (cost < best) { best = cost; pieces = nsp; }
            }
            // ROW-SPLIT work items (fp64, 2-D): a workgroup = ONE target tile whose rotation steps its four waves share --
            // the granularity of a quarter sweep of a super-row, but one row slot + one slab entry to reduce instead of
            // four + one (the reduction is bound by exactly that traffic).  Same model: a wave runs 64 / (4 nsp) steps; a
            // slab entry per tile instead of per super-row makes the reduction's items dearer (0.5 against 0.3 per unit).
            // Measured fp64 us per step, classic -> chosen (profiles/r03_rowsplit_sweep.txt): N = 5120 20.4 -> 18.2, 7168
            // 30.4 -> 26.9, 8192 35.9 -> 33.7, 12 288 62.0 -> 56.1, 16 384 96.6 -> 93.3.
            if (is_f64 && dim == 2 && cl == 1 && knobs.sym_rowsplit != 0) {
                long long rows_units = 0;
                for (int S = 0; S < SR; ++S)
                    if (ord[S] >= 0) for (int w = 0; w < 4; ++w) rows_units += std::max(0, T - (4 * S + w));
                for (int nsp : {1, 2, 3, 4}) {
                    const long long wgs = rows_units * nsp, rounds = (wgs + in.cus - 1) / in.cus;
                    const double cost = rounds * ((64 + 4 * nsp - 1) / (4 * nsp) + 1) * t_step[std::min<long long>(rounds, 4) - 1] +
                                        0.5 * 2.0 * (double)wgs / in.cus;
                    if (cost < best) { best = cost; rowsplit = nsp; }
                }
            }
        }
        for (int S = 0; S < SR; ++S) split_of[S] = pieces;
    } else if (items <= 100) {
        // Very small systems are bound by the LATENCY of one 64-step sweep (~4 us: a bpermute + a dependent VALU chain
        // per step), not by throughput: a few dozen workgroups leave most of the chip idle.  Cutting every sweep into
        // pieces shortens the critical path (measured fp64 N = 1024, 40 items: 12.1 / 9.2 / 8.0 / 9.1 us per step
        // with 1 / 2 / 4 / 8 pieces; N = 2048, 144 items: no gain).
        const int pieces = items < 24 ? 8 : (items <= 64 ? 4 : 2);
        for (int S = 0; S < SR; ++S) split_of[S] = pieces;
    } else if (sp.r == 2 && !is_f64 && items < (in.mode >= NB_INT8_SIM ? 900 : 400)) {
        // fp32, R = 2, a few hundred items: two pieces (us per step 1 / 2 / 4 pieces: N = 4096 15.9 / 13.8 / 14.7,
        // 6144 20.2 / 18.6 / 21.6; 8192 (544 items) 26.3 / 26.1 / 30.8).  The grid modes' pair loop is ~1.6 x as long, so
        // halves pay up to more items (INT8 us per step, 1 / 2 pieces: N = 7168 (420 items) 48.3 / 45.6, 8192 (544) 54.8 /
        // 50.6, 10 240 (840) 62.6 / 60.0, 11 264 (1012) 63.6 / 65.2; profiles/r03_tail_pieces_sweep.txt)
        for (int S = 0; S < SR; ++S) split_of[S] = 2;
    } else if (items > wg_slots / 2) {
        long long rem = items % wg_slots;
        // a single round of workgroups has nothing to hide a straggler behind: finer pieces there
        // R = 2 (fp32 family below N = 20 480): a whole sweep is only 64 steps x 4 pairs, finer pieces than halves cost
        // more than they smooth (FLOAT32 us per step, none / 2 / 4 / 8 pieces: N = 11 776 38.9 / 35.7 / 36.8 / 40.4,
        // 12 288 39.4 / 39.4 / 40.1 / 45.1, 13 312 45.6 / 45.5 / 48.0 / 52.6, 13 824 -- 488 items over -- 46.0 / 47.9 /
        // 51.0 / 54.8, 16 384 63.6 / 59.6 / 59.8 / 63.1; profiles/r03_tail_pieces_sweep.txt)
        const int pieces = knobs.tail_pieces ? knobs.tail_pieces : (sp.r == 2 ? 2 : (items < 2 * wg_slots ? 8 : 4));
        if (rem > 0 && rem <= (sp.r == 2 ? 3 * wg_slots / 8 : wg_slots / 2)) {
            for (int S = SR - 1; S >= 0 && rem > 0; --S)      // trailing (shortest) super-rows first
                if (ord[S] >= 0) { split_of[S] = pieces; rem -= nch_of[S]; }
        }
    }

    if (knobs.sym_rowsplit > 0 && is_f64 && dim == 2 && sp.r == 4 && in.mode == NB_FLOAT64) rowsplit = knobs.sym_rowsplit;   // A/B knob
    if (in.mode != NB_FLOAT64) rowsplit = 0;

    // ---- work items, slots, slab entries ---------------------------------------------------------------
    std::vector<SymWork> items_v;
    int slots = 0, ncol = 0;
    if (rowsplit > 0) {
        // one workgroup per (target tile I, source chunk, step piece q); slab entries per (I, q), numbered by ascending
        // tile so that the entries tile J needs -- those of the owned tiles I < J -- form a prefix
        const int nsp = rowsplit;
        sp.col_upto.assign(sp.tiles, 0);
        for (int I = 0; I < sp.tiles; ++I) {
            sp.col_upto[I] = ncol;
            sp.row_slot0[I] = slots;
            sp.row_nslots[I] = 0;
            if (I >= T || ord[I >> 2] < 0) continue;
            const int nch = (T - I + cl - 1) / cl;
            sp.row_nslots[I] = nch * nsp;
            for (int ch = 0; ch < nch; ++ch)
                for (int q = 0; q < nsp; ++q) {
                    const int s0 = 64 * q / nsp, s1 = 64 * (q + 1) / nsp;
                    items_v.push_back(SymWork{I, I + ch * cl, std::min(T, I + (ch + 1) * cl), slots + ch * nsp + q, -1,
                                              ncol + q, s0, s1 - s0});
                }
            slots += nch * nsp;
            ncol += nsp;
        }
    } else {
    for (int S = 0; S < SR; ++S) {
        if (ord[S] < 0) continue;
        const int j0 = 4 * S;
        const int nch = nch_of[S], nsp = split_of[S];
        const int per_row = nch * nsp;
        for (int w = 0; w < 4; ++w) { sp.row_slot0[j0 + w] = slots + w * per_row; sp.row_nslots[j0 + w] = per_row; }
        for (int ch = 0; ch < nch; ++ch)
            for (int q = 0; q < nsp; ++q) {
                // piece q of nsp: rotation steps [64 q / nsp, 64 (q + 1) / nsp) -- nsp need not divide 64
                const int s0 = 64 * q / nsp, s1 = 64 * (q + 1) / nsp;
                SymWork wk{j0, j0 + ch * cl, std::min(T, j0 + (ch + 1) * cl), slots + ch * nsp + q, per_row,
                           ncol + q, s0, s1 - s0};
                items_v.push_back(wk);
            }
        slots += 4 * per_row;
        ncol += nsp;
    }
    // slab entries are numbered in ascending super-row order, so the entries a tile needs -- those of the
    // owned super-rows strictly above it, plus its own super-row when it is not the first tile of it --
    // form a prefix of the index space
    sp.col_upto.assign(sp.tiles, 0);
    {
        std::vector<int> first(SR + 1, 0);      // first[S] = entries of owned super-rows < S
        for (int S = 0; S < SR; ++S) first[S + 1] = first[S] + (ord[S] >= 0 ? split_of[S] : 0);
        for (int J = 0; J < sp.tiles; ++J) sp.col_upto[J] = first[(J >> 2) + ((J & 3) ? 1 : 0)];
    }
    }
    // whole sweeps first, pieces last (longest processing time first)
    std::stable_sort(items_v.begin(), items_v.end(), [](const SymWork &a, const SymWork &b) {
        return (long long)(a.jt_end - a.jt_begin) * a.s_count > (long long)(b.jt_end - b.jt_begin) * b.s_count;
    });
    sp.work = std::move(items_v);
    sp.rowsplit = rowsplit;
    sp.nslots = slots;
    sp.ncol = ncol;
    sp.col_bytes = (size_t)dim * sp.np * el * (size_t)std::max(ncol, 1);
    sp.row_bytes = (size_t)dim * sp.tile_b * sizeof(double) * (size_t)std::max(slots, 1);
    sp.packed_bytes = (size_t)(dim + 1) * sp.np * el;
    sp.enabled = !sp.work.empty();
}
