// nb_state.h -- host-side state shared by the translation units behind the C-ABI (include/nbody_amd.h):
//   nb_api.cpp    entry points of a simulation handle + argument checks, state upload / download
//   nb_step.cpp   kernel selection and the sequencing of one force evaluation / leapfrog step / energy evaluation
//   nb_comm.cpp   process communicator: RCCL (resolved with dlopen) and the direct xGMI all-reduce (nb_p2p.hip)
//   nb_hooks.cpp  handle-less tensor-level hooks (quantization.py module functions) and diagnostics on caller tensors
// Host orchestration only: no arithmetic of the hot path runs on the host; without a HIP device every entry fails.
#pragma once
#include <rccl/rccl.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "nb_internal.h"
#include "nb_plan.h"

namespace nbhost {

// ---- errors: integer status + thread-local message (nb_last_error) ----------------------------------------------
std::string &last_error_string();
int fail(int code, const char *fmt, ...);

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return ::nbhost::fail(e_ == hipErrorOutOfMemory ? NB_ERR_OOM : NB_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                                  hipGetErrorString(e_), __FILE__, __LINE__);                     \
    } while (0)

inline int promote(int a, int b)
{
    if (a == b) return a;
    if (a == NB_F64 || b == NB_F64) return NB_F64;
    return NB_F32;
}
inline size_t dt_size(int dt) { return dt == NB_F64 ? 8 : (dt == NB_F32 ? 4 : 2); }
inline bool is_half(int dt) { return dt == NB_F16 || dt == NB_BF16; }
// host-side round-to-nearest-even to the dtype (only for the O(1) scalars of a call: eps2 and the final energy
// scalings; torch casts double -> half via float)
double round_dt(int dt, double x);

// ---- RCCL, resolved lazily so single-GPU use never loads it ------------------------------------------------------
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
extern Rccl g_rccl;
int load_rccl();

#define NCCLCHK(expr)                                                                             \
    do {                                                                                          \
        ncclResult_t r_ = (expr);                                                                 \
        if (r_ != ncclSuccess)                                                                    \
            return ::nbhost::fail(NB_ERR_COMM, "%s failed: %s", #expr,                            \
                                  ::nbhost::g_rccl.GetErrorString ? ::nbhost::g_rccl.GetErrorString(r_) : "?"); \
    } while (0)

// One RCCL communicator per PROCESS (= per GPU), shared by every simulation handle of the process: a
// precision sweep builds seven simulations, not seven communicators.  Handles borrow it; only the explicit,
// collective nb_comm_shutdown() destroys it -- never nb_destroy(), which Python may run from a garbage
// collector at a different moment on every rank.  `generation` counts communicators of this process: a handle
// remembers the one it attached to, and every use after nb_comm_shutdown() fails instead of touching a destroyed
// communicator.
struct ProcComm {
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = 0, device = -1;
    // "direct only": no RCCL communicator at all -- every sum goes through the direct all-reduce of nb_p2p.hip
    // (ranks of one node, vectors up to its capacity).  `comm` then holds a sentinel that is never handed to RCCL.
    bool direct_only = false;
    unsigned generation = 0;
};
extern ProcComm g_pc;
extern std::mutex g_pc_mu;
// The direct all-reduce has ONE shared input buffer per process: a handle on another stream must not fill it while
// the previous user's kernels may still read it.  Handles alternate rarely (several simulations alive at once), so
// the hand-over is a host-side wait on the previous user's stream; a single simulation never pays for it.
extern hipStream_t g_p2p_last_stream;
extern std::mutex g_p2p_mu;

constexpr int PROF_RING = 256;
// A peer that never arrives at a barrier of the direct all-reduce raises a sticky status word after this long; every
// entry point that synchronises a handle which used the direct path reports it as NB_ERR_COMM (p2p_check).
// NB_P2P_TIMEOUT_S (read once per process) overrides it -- tests drive the error path with a short one.
double p2p_step_timeout_s();

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; (void)hipSetDevice(dev); }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int check_device(int device);      // NB_OK if `device` is a valid HIP device ordinal

}  // namespace nbhost

struct nb_sim {
    nb_config cfg{};
    hipStream_t stream = nullptr;
    bool have_storage = false;
    bool is_f64 = false;                 // storage / accumulation type of the state buffers
    int logical[4] = {NB_F32, NB_F32, NB_F32, NB_F32};   // pos, vel, mass, acc as Python sees them
    bool have_pos = false, have_vel = false, have_mass = false, have_acc = false;
    void *arena = nullptr;               // ONE device allocation behind the buffers of the first upload (nb_api.cpp: ensure_storage)
    size_t arena_bytes = 0;
    void *pos = nullptr, *vel = nullptr, *mass = nullptr, *acc = nullptr;
    double *partial = nullptr;           // nchunks slabs of n*dim fp64 partial sums
    void *staging = nullptr;             // n*dim*8 bytes, for dtype conversion on upload / download
    GridTables *tab = nullptr;
    double *scratch = nullptr;           // per-block energy partials
    size_t scratch_elems = 0;
    double *scalars = nullptr;           // device: [0,1] force min/max, [2] ke, [3] pe
    int16_t *fbins = nullptr;            // n*dim, INT8/INT4 only
    float *prune_cand = nullptr, *prune_rho = nullptr;   // grid modes: pruned max-r2 search
    int *prune_idx = nullptr;            // ... particle indices of the compacted candidates (tracked search)
    PruneState *prune_state = nullptr;
    bool prune_seeded = false;           // the last evaluation left a far pair / centre / rho bound for a tracked search
    bool mass_uniform = false;           // all masses equal (checked on the device at upload)
    double mass_value = 0.0;
    const char *last_kernel = "none";
    ForceGeom geom{};
    // pair-symmetric path (nb_force_sym.hip): device mirror of the host plan (nb_plan.h)
    struct SymPlan {
        bool enabled = false;
        bool rowsplit = false;            // row-split work items: force_sym_kernel's RSPLIT instantiations
        int r = 2, tile_b = 128, tiles = 0, np = 0, nwork = 0, nslots = 0;
        SymWork *work = nullptr;
        int *row_slot0 = nullptr, *row_nslots = nullptr, *col_upto = nullptr;
        void *packed = nullptr, *colslab = nullptr;   // storage type of the state (fp32 or fp64)
        double *rowslab = nullptr;
    } sym;
    void *pos_alt = nullptr;             // small-N single-launch step: positions ping-pong between pos and pos_alt
    bool spec_open = false;              // ... pos_alt holds the positions the NEXT step drifts to (computed with spec_dt by
    double spec_dt = 0.0;                //     the last step of the previous nb_step); dropped by every entry that writes state
    int spec_kind = 0;                   //     1: left by the small-system kernel, 2: by reduce_sym_kernel (sym.packed holds them too)
    bool req_spec_next = false;          // step_run -> force_eval: last step of a call, leave the next step's positions
    bool req_open_on_read = false;       // step_run -> force_eval: this step started from speculative positions
    double *small_part = nullptr;        // ... INT8 / INT4: per-target min / max of the forces (2 n doubles)
    void *gen_scalars = nullptr;         // generic (dtype-faithful) path: device scalars of one evaluation
    bool last_generic = false;           // the last force evaluation ran on the generic path (no threshold tables)
    bool used_p2p = false;               // a force vector of this handle went through the direct xGMI all-reduce
    double *sums64 = nullptr;            // multi-GPU, fp32 state, RCCL carrier: the fp64 sums the ranks exchange
    void *metrics_scratch = nullptr;     // nb_metrics work arrays (allocated on first use)
    size_t metrics_cap = 0;
    unsigned long long *bin_out = nullptr;   // nb_quant_bin_sums: {s1[n], s2[n], counters[2]} while a read-out runs
    bool bins_active = false;            // force_eval launches the BINS instantiations of the grid-mode kernels
    NbKnobs knobs;                       // environment knobs, read once in nb_create
    ncclComm_t comm = nullptr;
    unsigned comm_generation = 0;        // ProcComm::generation this handle attached to
    // profiling
    hipEvent_t ev_start[nbhost::PROF_RING], ev_stop[nbhost::PROF_RING];
    bool prof_init = false;
    int prof_count = 0;
    double prof_total_ms = 0.0;
    int prof_launches = 0;
};

namespace nbhost {

inline int64_t nd(const nb_sim *s) { return (int64_t)s->cfg.n * s->cfg.dim; }
inline bool grid_mode(int mode) { return mode >= NB_INT8_SIM; }
// collectives run whenever a communicator is attached (a 1-rank communicator exercises the same RCCL calls on
// a single GPU) and must exist when the pair work is really sharded
inline bool comm_active(const nb_sim *s)
{
    return (s->cfg.nranks > 1 && !(s->cfg.flags & NB_FLAG_NO_COMM)) || s->comm != nullptr;
}
inline int mode_levels(const nb_config &c)
{
    if (c.mode == NB_INT8_SIM) return 256;
    if (c.mode == NB_INT4_SIM) return 16;
    return c.levels > 0 ? c.levels : 64;
}
inline bool force_quant_mode(const nb_config &c)
{
    return c.mode == NB_INT8_SIM || c.mode == NB_INT4_SIM ||
           (c.mode == NB_CUSTOM && (c.flags & NB_FLAG_CUSTOM_FORCEQ));
}
int acc_logical_dtype(const nb_sim *s);

// ---- nb_comm.cpp ---------------------------------------------------------------------------------------------------
// NB_OK while the handle's communicator (if any) is the live process communicator; NB_ERR_COMM after nb_comm_shutdown
int comm_check(const nb_sim *s);
bool p2p_use(const nb_sim *s, int64_t cnt);      // the direct all-reduce serves this handle's force vector
bool p2p_use_x64(const nb_sim *s, int64_t cnt);  // ... the fp64 sums of an fp32 handle (INT8 / INT4 exchange)
int p2p_claim_buffer(nb_sim *s);                 // hand the shared input buffer over to this handle's stream
// sum `count` elements of `buf` over the ranks, in place, on the handle's stream (RCCL or the direct path)
int comm_allreduce_sum(nb_sim *s, void *buf, size_t count, bool f64);
int comm_allreduce_max_u32(nb_sim *s, unsigned int *buf);
// after a host-side wait on the handle's stream: NB_ERR_COMM if a barrier of the direct all-reduce timed out
int p2p_check(nb_sim *s);

// ---- nb_step.cpp ---------------------------------------------------------------------------------------------------
void compute_geometry(nb_sim *s);
// one evaluation of simulation.py:74-118; optionally followed by the closing half kick (:141)
// defer_kick: the caller will apply the closing half kick itself (fused into the next step's
// opening launch) when this evaluation cannot fuse it into its reduction.
int force_eval(nb_sim *s, bool do_kick, bool packed_ready = false, bool *defer_kick = nullptr,
               bool *open_next = nullptr);
int step_run(nb_sim *s, int nsteps);                                   // simulation.py:120-143, nsteps times
int energy_eval(nb_sim *s, double *kinetic, double *potential);        // simulation.py:170-192
int bin_sums_eval(nb_sim *s, int which, int64_t *sum_k, int64_t *sum_kw, double info[8]);
int prof_collect(nb_sim *s, double *total_ms, int32_t *launches);

}  // namespace nbhost
