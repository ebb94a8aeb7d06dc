// nb_plan.h -- host-side planning of the pair-symmetric force evaluation (pure C++: no HIP call, no
// allocation on a device), shared by nb_api.cpp (build_sym_plan) and the device-less debug entry nb_plan_debug().
//
// What is planned: which (super-row, source-tile) work items THIS rank evaluates, where each item
// leaves its row / column sums and in which order the reduction adds them.  Everything here is a function of
// rank-independent quantities plus the rank itself, so every rank derives a consistent global plan
// without talking to the others.
#pragma once
#include <vector>

#include "nb_internal.h"

// Tuning / test knobs, read from the environment ONCE per handle (nb_create) -- never inside a step.
struct NbKnobs {
    int sym = -1;          // NB_SYM: -1 by size, 0 never, 1 always, 2 also for comm-less shards (tests)
    int sym_r = 0;         // NB_SYM_R: targets per lane (1, 2, 4); 0 = by size
    int sym_cl = 0;        // NB_SYM_CL: source tiles per work item; 0 = by work
    int sym_split = 0;     // NB_SYM_SPLIT: cut EVERY sweep into 1 ... 16 pieces; 0 = the plan's own choice
    int sym_rowsplit = -1; // NB_SYM_ROWSPLIT: row-split work items (fp64, 2-D, R = 4): -1 = the plan's choice, 0 = never, n = always, n step pieces
    int tail_pieces = 0;   // NB_SYM_TAIL: pieces per sweep of the tail-smoothed super-rows (4 or 8); 0 = auto
    int r_onesided = 0;    // NB_R: targets per thread of the one-sided fp64 kernel
    bool no_prune = false;   // NB_NO_PRUNE: all-pairs max-r2 scan at any N
    bool no_track = false;   // NB_NO_TRACK: every grid evaluation searches its farthest pair from scratch (A/B of round 3's tracked search)
    bool no_pe_sym = false;  // NB_NO_PE_SYM: one-sided potential-energy kernel
    bool no_uniform = false; // NB_NO_UNIFORM: general-mass kernels even for equal masses
    bool no_smalln = false;  // NB_NO_SMALLN: never use the single-launch small-N step
    int small_max = 0;       // NB_SMALL_MAX: largest N of the single-launch step (0 = default), NB_SMALL_LANES: lanes per target
    int small_lanes = 0;
    bool no_grid_fast = false;   // NB_NO_GRID_FAST: grid modes always read their tables (A/B of the table-free pair path)
    bool no_p2p = false;         // NB_NO_P2P: force vectors always go through RCCL (never the direct xGMI all-reduce)
    bool no_red_mm = false;      // NB_NO_RED_MM: INT8 / INT4 steps keep their own min/max launch (A/B)
    bool no_x64 = false;         // NB_NO_X64: multi-GPU fp32 modes exchange fp32 partial forces (not the fp64 sums)
    bool no_small_fuse = false;  // NB_NO_SMALL_FUSE: small grid steps launch max-r2 and tables separately (A/B)
    bool no_p2p_kick = false;    // NB_P2P_NO_KICK: the direct all-reduce does not fuse the kicks / drift / repack (A/B)
    bool no_spec = false;        // NB_NO_SPEC: the last step of a native call does not leave the next step's positions (A/B, tests)
};
NbKnobs nb_read_knobs();

struct SymPlanHost {
    bool enabled = false;
    int r = 2, tile_b = 128, tiles = 0, np = 0, nslots = 0, ncol = 0, cl = 1;
    int rowsplit = 0;                                // > 0: row-split work items (slot_stride < 0) with this many step pieces
    std::vector<SymWork> work;                       // longest first
    std::vector<int> row_slot0, row_nslots, col_upto;   // per tile (see reduce_sym_kernel)
    size_t col_bytes = 0, row_bytes = 0, packed_bytes = 0;
};

struct PlanInput {
    int n = 0, dim = 2, mode = 0, flags = 0, rank = 0, nranks = 1;
    bool is_f64 = true;
    bool multi = false;      // a communicator will sum the partial forces
    int cus = 256;           // compute units of the device
};

// Fills `out`; out.enabled == false means "use the one-sided kernels" (the decision is rank-independent).
void nb_plan_sym(const PlanInput &in, const NbKnobs &knobs, SymPlanHost &out);
