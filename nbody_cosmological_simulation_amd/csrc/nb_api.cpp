// nb_api.cpp -- C-ABI of the MI355X N-body engine (include/nbody_amd.h).
//
// Host-side orchestration only: owns the per-handle device state, decides the launch geometry,
// tracks the reference's dtype state machine and sequences the kernels
// of one force evaluation / leapfrog step on the handle's own HIP stream.  No arithmetic of
// the hot path runs on the host; without a HIP device every entry point fails.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "nb_internal.h"
#include "nb_plan.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(e_ == hipErrorOutOfMemory ? NB_ERR_OOM : NB_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                               \
    } while (0)

inline int promote(int a, int b)
{
    if (a == b) return a;
    if (a == NB_F64 || b == NB_F64) return NB_F64;
    return NB_F32;
}
inline size_t dt_size(int dt) { return dt == NB_F64 ? 8 : (dt == NB_F32 ? 4 : 2); }
inline bool is_half(int dt) { return dt == NB_F16 || dt == NB_BF16; }

// host-side round-to-nearest-even to float16 / bfloat16 (only for the O(1) scalars of a call:
// eps2 and the final energy scalings of half-typed state; torch casts double -> half via float)
double round_small(double x, int mant, int emin, int emax)
{
    if (x == 0.0 || std::isnan(x) || std::isinf(x)) return x;
    int e;
    (void)std::frexp(std::fabs(x), &e);
    const int ue = e - 1;
    const int q = ue < emin ? emin : ue;
    const double ulp = std::ldexp(1.0, q - mant);
    double r = std::nearbyint(std::fabs(x) / ulp) * ulp;
    if (r > std::ldexp(2.0 - std::ldexp(1.0, -mant), emax)) r = INFINITY;
    return x < 0 ? -r : r;
}
double round_dt(int dt, double x)
{
    if (dt == NB_F64) return x;
    const double f = (double)(float)x;
    if (dt == NB_F16) return round_small(f, 10, -14, 15);
    if (dt == NB_BF16) return round_small(f, 7, -126, 127);
    return f;
}

// ---- RCCL, resolved lazily so single-GPU use never loads it ---------------------------------
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int load_rccl()
{
    if (g_rccl.lib) return NB_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(NB_ERR_COMM, "cannot load librccl: %s", dlerror());
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy)
        return fail(NB_ERR_COMM, "librccl is missing expected symbols");
    g_rccl.lib = h;
    return NB_OK;
}

#define NCCLCHK(expr)                                                                             \
    do {                                                                                          \
        ncclResult_t r_ = (expr);                                                                 \
        if (r_ != ncclSuccess)                                                                    \
            return fail(NB_ERR_COMM, "%s failed: %s", #expr,                                      \
                        g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?");                 \
    } while (0)

// One RCCL communicator per PROCESS (= per GPU), shared by every simulation handle of the process: a
// precision sweep builds seven simulations, not seven communicators.  Handles borrow it; only the explicit,
// collective nb_comm_shutdown() destroys it -- never nb_destroy(), which Python may run from a garbage
// collector at a different moment on every rank.
struct ProcComm {
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = 0, device = -1;
    // "direct only": no RCCL communicator at all -- every sum goes through the direct all-reduce of nb_p2p.hip
    // (ranks of one node, vectors up to its capacity).  `comm` then holds a sentinel that is never handed to RCCL.
    bool direct_only = false;
};
char g_direct_sentinel;
ProcComm g_pc;
std::mutex g_pc_mu;
// The direct all-reduce has ONE shared input buffer per process: a handle on another stream must not fill it while
// the previous user's kernels may still read it.  Handles alternate rarely (several simulations alive at once), so
// the hand-over is a host-side wait on the previous user's stream; a single simulation never pays for it.
hipStream_t g_p2p_last_stream = nullptr;
std::mutex g_p2p_mu;

constexpr int PROF_RING = 256;

}  // namespace

struct nb_sim {
    nb_config cfg{};
    hipStream_t stream = nullptr;
    bool have_storage = false;
    bool is_f64 = false;                 // storage / accumulation type of the state buffers
    int logical[4] = {NB_F32, NB_F32, NB_F32, NB_F32};   // pos, vel, mass, acc as Python sees them
    bool have_pos = false, have_vel = false, have_mass = false, have_acc = false;
    void *pos = nullptr, *vel = nullptr, *mass = nullptr, *acc = nullptr;
    double *partial = nullptr;           // nchunks slabs of n*dim fp64 partial sums
    void *staging = nullptr;             // n*dim*8 bytes, for dtype conversion on upload / download
    GridTables *tab = nullptr;
    double *scratch = nullptr;           // per-block energy partials
    size_t scratch_elems = 0;
    double *scalars = nullptr;           // device: [0,1] force min/max, [2] ke, [3] pe
    int16_t *fbins = nullptr;            // n*dim, INT8/INT4 only
    float *prune_cand = nullptr, *prune_rho = nullptr;   // grid modes: pruned max-r2 search
    PruneState *prune_state = nullptr;
    bool mass_uniform = false;           // all masses equal (checked on the device at upload)
    double mass_value = 0.0;
    const char *last_kernel = "none";
    ForceGeom geom{};
    // pair-symmetric path (nb_force_sym.hip): device mirror of the host plan (nb_plan.h)
    struct SymPlan {
        bool enabled = false;
        int r = 2, tile_b = 128, tiles = 0, np = 0, nwork = 0, nslots = 0;
        SymWork *work = nullptr;
        int *row_slot0 = nullptr, *row_nslots = nullptr, *col_upto = nullptr;
        void *packed = nullptr, *colslab = nullptr;   // storage type of the state (fp32 or fp64)
        void *packed_alt = nullptr;                   // second packed buffer of the chunked multi-GPU step
        double *rowslab = nullptr;
        std::vector<int> chunk_work, chunk_tile;      // pipeline chunks (host side, see nb_plan.h)
    } sym;
    void *pos_alt = nullptr;             // small-N single-launch step: positions ping-pong between pos and pos_alt
    double *small_part = nullptr;        // ... INT8 / INT4: per-target min / max of the forces (2 n doubles)
    void *gen_scalars = nullptr;         // generic (dtype-faithful) path: device scalars of one evaluation
    bool last_generic = false;           // the last force evaluation ran on the generic path (no threshold tables)
    bool used_p2p = false;               // a force vector of this handle went through the direct xGMI all-reduce
    double *sums64 = nullptr;            // multi-GPU, fp32 state, RCCL carrier: the fp64 sums the ranks exchange
    void *metrics_scratch = nullptr;     // nb_metrics work arrays (allocated on first use)
    size_t metrics_cap = 0;
    NbKnobs knobs;                       // environment knobs, read once in nb_create
    // chunked multi-GPU step: second force stream, collective stream, events
    hipStream_t fstream2 = nullptr, cstream = nullptr;
    hipEvent_t ev_ready = nullptr, ev_done = nullptr, ev_force[4] = {nullptr, nullptr, nullptr, nullptr};
    hipGraphExec_t chunk_graph_exec = nullptr;       // two chunked steps (A -> B -> A), captured once
    double chunk_graph_key[4] = {0, 0, 0, 0};        // G, softening^2, dt, uniform mass the graph was captured with
    ncclComm_t comm = nullptr;
    // profiling
    hipEvent_t ev_start[PROF_RING], ev_stop[PROF_RING];
    bool prof_init = false;
    int prof_count = 0;
    double prof_total_ms = 0.0;
    int prof_launches = 0;
};

namespace {

int64_t nd(const nb_sim *s) { return (int64_t)s->cfg.n * s->cfg.dim; }

bool grid_mode(int mode) { return mode >= NB_INT8_SIM; }
// collectives run whenever a communicator is attached (a 1-rank communicator exercises the same RCCL calls on
// a single GPU) and must exist when the pair work is really sharded
bool comm_active(const nb_sim *s)
{
    return (s->cfg.nranks > 1 && !(s->cfg.flags & NB_FLAG_NO_COMM)) || s->comm != nullptr;
}
// The direct xGMI all-reduce (nb_p2p.hip) serves this handle's force vector when every rank enabled it after the
// collective self-test, the process communicator is the one it was built for, and the vector fits its buffers.
// The decision depends only on values that are equal on all ranks.
constexpr double P2P_STEP_TIMEOUT_S = 300.0;
bool p2p_use(const nb_sim *s, int64_t cnt)
{
    if (s->knobs.no_p2p || !s->comm || nb_p2p_state() != 2) return false;
    if (nb_p2p_nranks() != g_pc.nranks || nb_p2p_device() != s->cfg.device) return false;
    if (!s->is_f64 && (cnt & 1)) return false;                 // the kernel moves 8-byte units
    return (size_t)cnt * (s->is_f64 ? 8 : 4) <= nb_p2p_capacity();
}
// Sum `count` elements of `buf` over the ranks, in place, on the handle's stream: RCCL, or -- on a direct-only
// communicator -- a copy into the shared input buffer and the direct all-reduce.
int comm_allreduce_sum(nb_sim *s, void *buf, size_t count, bool f64)
{
    if (!g_pc.direct_only) {
        NCCLCHK(g_rccl.AllReduce(buf, buf, count, f64 ? ncclDouble : ncclFloat, ncclSum, s->comm, s->stream));
        return NB_OK;
    }
    const size_t bytes = count * (f64 ? 8 : 4);
    if (nb_p2p_state() != 2 || bytes > nb_p2p_capacity() || (!f64 && (count & 1)))
        return fail(NB_ERR_COMM, "direct-only communicator: %zu %s elements do not fit the direct all-reduce (capacity %zu "
                                 "bytes, fp32 counts even); use an RCCL communicator", count, f64 ? "fp64" : "fp32",
                    nb_p2p_capacity());
    {
        std::lock_guard<std::mutex> lock(g_p2p_mu);
        if (g_p2p_last_stream && g_p2p_last_stream != s->stream) HIPCHK(hipStreamSynchronize(g_p2p_last_stream));
        g_p2p_last_stream = s->stream;
    }
    HIPCHK(hipMemcpyAsync(nb_p2p_data(), buf, bytes, hipMemcpyDeviceToDevice, s->stream));
    HIPCHK(nb_p2p_allreduce(buf, count, f64, P2P_STEP_TIMEOUT_S, s->stream));
    s->used_p2p = true;
    return NB_OK;
}
int mode_levels(const nb_config &c)
{
    if (c.mode == NB_INT8_SIM) return 256;
    if (c.mode == NB_INT4_SIM) return 16;
    return c.levels > 0 ? c.levels : 64;
}
bool force_quant_mode(const nb_config &c)
{
    return c.mode == NB_INT8_SIM || c.mode == NB_INT4_SIM ||
           (c.mode == NB_CUSTOM && (c.flags & NB_FLAG_CUSTOM_FORCEQ));
}

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
    g.r = (n <= 8192) ? 1 : 2;
    if (s->knobs.r_onesided == 1 || s->knobs.r_onesided == 2 || s->knobs.r_onesided == 4)   // NB_R tuning knob
        g.r = s->knobs.r_onesided;
    const int njr = std::max(g.j_end - g.j_begin, 1);
    const int itiles = (n + NB_BLOCK * g.r - 1) / (NB_BLOCK * g.r);
    const int max_chunks = (njr + NB_TJ - 1) / NB_TJ;
    int nch = (1024 + itiles - 1) / itiles;       // aim for >= 4 workgroups per CU
    nch = std::max(1, std::min(std::min(nch, max_chunks), 64));
    int chunk = (njr + nch - 1) / nch;
    chunk = (chunk + NB_TJ - 1) / NB_TJ * NB_TJ;
    g.chunk_len = chunk;
    g.nchunks = (njr + chunk - 1) / chunk;
    s->geom = g;
}

// Work list of the pair-symmetric kernel: planned on the host (nb_plan.cpp, device-free and testable on its
// own through nb_plan_debug), mirrored here into device buffers.
int build_sym_plan(nb_sim *s)
{
    auto &sp = s->sym;
    sp.enabled = false;
    const nb_config &c = s->cfg;
    PlanInput in;
    in.n = c.n; in.dim = c.dim; in.mode = c.mode; in.flags = c.flags; in.rank = c.rank; in.nranks = c.nranks;
    in.is_f64 = s->is_f64;
    in.multi = comm_active(s);
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, c.device) == hipSuccess) in.cus = prop.multiProcessorCount; }
    SymPlanHost h;
    nb_plan_sym(in, s->knobs, h);
    if (!h.enabled) return NB_OK;
    sp.r = h.r; sp.tile_b = h.tile_b; sp.tiles = h.tiles; sp.np = h.np;
    sp.nwork = (int)h.work.size();
    sp.nslots = h.nslots;
    sp.chunk_work = h.chunk_work;
    sp.chunk_tile = h.chunk_tile;
    HIPCHK(hipMalloc((void **)&sp.work, h.work.size() * sizeof(SymWork)));
    HIPCHK(hipMalloc((void **)&sp.row_slot0, sp.tiles * sizeof(int)));
    HIPCHK(hipMalloc((void **)&sp.row_nslots, sp.tiles * sizeof(int)));
    HIPCHK(hipMalloc((void **)&sp.col_upto, sp.tiles * sizeof(int)));
    HIPCHK(hipMalloc((void **)&sp.packed, h.packed_bytes));
    if (sp.chunk_tile.size() > 2) HIPCHK(hipMalloc((void **)&sp.packed_alt, h.packed_bytes));
    HIPCHK(hipMalloc((void **)&sp.rowslab, h.row_bytes));
    HIPCHK(hipMalloc((void **)&sp.colslab, h.col_bytes));
    HIPCHK(hipMemcpy(sp.work, h.work.data(), h.work.size() * sizeof(SymWork), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(sp.row_slot0, h.row_slot0.data(), sp.tiles * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(sp.row_nslots, h.row_nslots.data(), sp.tiles * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(sp.col_upto, h.col_upto.data(), sp.tiles * sizeof(int), hipMemcpyHostToDevice));
    sp.enabled = true;
    return NB_OK;
}

int ensure_storage(nb_sim *s, bool f64)
{
    if (s->have_storage) {
        if (s->is_f64 != f64)
            return fail(NB_ERR_UNSUPPORTED, "state storage type cannot change after the first upload "
                                            "(create a new handle for a different input dtype)");
        return NB_OK;
    }
    HIPCHK(hipSetDevice(s->cfg.device));
    s->is_f64 = f64;
    const size_t el = f64 ? 8 : 4;
    const size_t cnt = (size_t)nd(s);
    HIPCHK(hipMalloc(&s->pos, cnt * el));
    HIPCHK(hipMalloc(&s->vel, cnt * el));
    HIPCHK(hipMalloc(&s->acc, cnt * el));
    HIPCHK(hipMalloc(&s->mass, (size_t)s->cfg.n * el));
    HIPCHK(hipMalloc(&s->staging, cnt * 8));
    HIPCHK(hipMalloc((void **)&s->partial, (size_t)s->geom.nchunks * cnt * sizeof(double)));
    HIPCHK(hipMalloc((void **)&s->tab, sizeof(GridTables)));
    HIPCHK(hipMemsetAsync(s->tab, 0, sizeof(GridTables), s->stream));
    const size_t pe_blocks = (size_t)((s->cfg.n + NB_BLOCK - 1) / NB_BLOCK) * s->geom.nchunks;
    s->scratch_elems = std::max<size_t>(pe_blocks, 1024);
    HIPCHK(hipMalloc((void **)&s->scratch, s->scratch_elems * sizeof(double)));
    // scalars[0..7] results, [8 ...] partials of the two-stage min/max
    HIPCHK(hipMalloc((void **)&s->scalars, (8 + 2 * NB_MINMAX_BLOCKS) * sizeof(double)));
    HIPCHK(hipMemsetAsync(s->scalars, 0, (8 + 2 * NB_MINMAX_BLOCKS) * sizeof(double), s->stream));
    if (force_quant_mode(s->cfg)) HIPCHK(hipMalloc((void **)&s->fbins, cnt * sizeof(int16_t)));
    if (!f64 && grid_mode(s->cfg.mode)) {
        HIPCHK(hipMalloc((void **)&s->prune_cand, cnt * sizeof(float)));
        HIPCHK(hipMalloc((void **)&s->prune_rho, (size_t)s->cfg.n * sizeof(float)));
        HIPCHK(hipMalloc((void **)&s->prune_state, sizeof(PruneState)));
        const PruneState init = {{0xffffffffu, 0xffffffffu, 0xffffffffu}, {0u, 0u, 0u}, 0ull, {0ull, 0ull}, 0, 0};
        HIPCHK(hipMemcpy(s->prune_state, &init, sizeof init, hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemsetAsync(s->acc, 0, cnt * el, s->stream));
    if (int rc = build_sym_plan(s)) return rc;
    if ((size_t)s->sym.nwork > s->scratch_elems) {
        (void)hipFree(s->scratch);
        s->scratch_elems = (size_t)s->sym.nwork;
        HIPCHK(hipMalloc((void **)&s->scratch, s->scratch_elems * sizeof(double)));
    }
    s->have_storage = true;
    return NB_OK;
}

// copy `count` elements of dtype `dt` from a caller buffer into state storage (with conversion)
int upload(nb_sim *s, const void *src, int dt, int on_device, void *dst, int64_t count)
{
    const int sdt = s->is_f64 ? NB_F64 : NB_F32;
    const size_t bytes = (size_t)count * dt_size(dt);
    if (dt == sdt) {
        HIPCHK(hipMemcpyAsync(dst, src, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                              s->stream));
        // the caller owns `src` and may free or overwrite it as soon as we return (a torch temporary
        // goes back to the caching allocator): the copy must have consumed it by then
        HIPCHK(hipStreamSynchronize(s->stream));
        return NB_OK;
    }
    const void *dsrc = src;
    if (!on_device) {
        HIPCHK(hipMemcpyAsync(s->staging, src, bytes, hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        dsrc = s->staging;
    }
    HIPCHK(nb_launch_convert(dsrc, dt, dst, sdt, count, s->stream));
    if (on_device) HIPCHK(hipStreamSynchronize(s->stream));        // same ownership rule as above
    return NB_OK;
}

int download(nb_sim *s, const void *src, int logical_dt, void *dst, int on_device, int64_t count)
{
    const int sdt = s->is_f64 ? NB_F64 : NB_F32;
    const size_t bytes = (size_t)count * dt_size(logical_dt);
    if (logical_dt == sdt) {
        HIPCHK(hipMemcpyAsync(dst, src, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                              s->stream));
        return NB_OK;
    }
    if (on_device) {
        HIPCHK(nb_launch_convert(src, sdt, dst, logical_dt, count, s->stream));
        return NB_OK;
    }
    HIPCHK(nb_launch_convert(src, sdt, s->staging, logical_dt, count, s->stream));
    HIPCHK(hipMemcpyAsync(dst, s->staging, bytes, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));   // staging is reused by the next array
    return NB_OK;
}

int acc_logical_dtype(const nb_sim *s)
{
    // promote(promote(Q, M), P) with Q = hook output dtype (quantization.py:43-71)
    int q = s->logical[0];
    if (s->cfg.mode == NB_FLOAT64) q = NB_F64;
    else if (s->cfg.mode <= NB_FLOAT16) q = NB_F32;
    return promote(promote(promote(q, s->logical[2]), NB_F32), s->logical[0]);
}

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

// one evaluation of simulation.py:74-118; optionally followed by the closing half kick (:141)
// defer_kick: the caller will apply the closing half kick itself (fused into the next step's
// opening launch) when this evaluation cannot fuse it into its reduction.
int force_eval(nb_sim *s, bool do_kick, bool packed_ready = false, bool *defer_kick = nullptr,
               bool *open_next = nullptr)
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
    if (no_comm && c.nranks > 1 && do_kick) return fail(NB_ERR_INVALID, "NB_FLAG_NO_COMM handles cannot step");
    if (use_generic(s)) return force_eval_generic(s, do_kick, defer_kick, open_next);
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
                                           c.softening_sq, s->stream, prof_events(s, slot)));
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
            const bool prune = !s->knobs.no_prune && c.n > 8192;
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
                    NCCLCHK(g_rccl.AllReduce(&s->tab->r2max_bits, &s->tab->r2max_bits, 1, ncclUint32, ncclMax, s->comm,
                                             s->stream));
            }
            HIPCHK(nb_launch_grid_tables(s->tab, L, (float)c.G, eps2, 0.01f, prune ? s->prune_state : nullptr,
                                         s->stream, s->knobs.no_grid_fast ? 0 : 1));
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
            HIPCHK(nb_launch_force_sym_f32((const float *)sp.packed, sp.work, sp.nwork, sp.rowslab,
                                           (float *)sp.colslab, sp.np, c.dim, sp.r,
                                           hook == HOOK_GRID ? grid_uniform : sym_uniform, hook, eps2, s->tab,
                                           (float)c.G, (float)s->mass_value, hook == HOOK_GRID ? mode_levels(c) : 0,
                                           s->stream, prof_events(s, slot)));
            s->last_kernel = "force_sym_kernel<float";
        } else {
            if (int rc = prof_begin(s, &slot)) return rc;
            HIPCHK(nb_launch_force_f32((const float *)s->pos, (const float *)s->mass, s->partial, s->geom, c.dim, hook,
                                       pa, (float)c.G, eps2, s->tab, hook == HOOK_GRID ? mode_levels(c) : 0, s->stream));
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
    const bool red_mm = fq && used_sym && !multi && !s->is_f64 && red_blocks <= 512 && !s->knobs.no_red_mm;
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
    bool p2p = multi && (x64 ? (s->comm && !s->knobs.no_p2p && nb_p2p_state() == 2 && nb_p2p_nranks() == g_pc.nranks &&
                               nb_p2p_device() == c.device && (size_t)cnt * 8 <= nb_p2p_capacity())
                             : p2p_use(s, cnt));
    void *red_out = p2p ? nb_p2p_data() : s->acc;
    if (x64 && !p2p && !s->sums64) HIPCHK(hipMalloc((void **)&s->sums64, (size_t)cnt * sizeof(double)));
    double *sums64 = x64 ? (p2p ? (double *)nb_p2p_data() : s->sums64) : nullptr;
    if (p2p) {
        std::lock_guard<std::mutex> lock(g_p2p_mu);
        if (g_p2p_last_stream && g_p2p_last_stream != s->stream) HIPCHK(hipStreamSynchronize(g_p2p_last_stream));
        g_p2p_last_stream = s->stream;
    }
    if (used_sym) {
        const auto &sp = s->sym;
        // uniform-mass kernels leave out the mass factor: G*m in T arithmetic (fp32: (float)G * m)
        double scale = 1.0;
        if (sym_uniform) scale = s->is_f64 ? c.G * s->mass_value : (double)((float)c.G * (float)s->mass_value);
        // inside nb_step the reduction also opens the next step and repacks its positions
        const bool open = fuse_kick && want_open;
        HIPCHK(nb_launch_reduce_sym(sp.rowslab, sp.colslab, sp.row_slot0, sp.row_nslots, sp.col_upto,
                                    sp.tile_b, c.n, sp.np, c.dim, s->is_f64, scale, red_out, s->vel, half_dt,
                                    open ? 2 : (fuse_kick ? 1 : 0), s->pos, sp.packed, c.dt, s->stream, 0, -1, sums64,
                                    red_mm ? s->scalars + 8 : nullptr));
        x64_scale = scale;
        opened = open;
    } else {
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
        HIPCHK(nb_p2p_allreduce(s->acc, (size_t)cnt, s->is_f64 || x64, P2P_STEP_TIMEOUT_S, s->stream, &kk));
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

// ---- chunked multi-GPU step ------------------------------------------------------
// The owned super-rows are swept in C chunks of ascending super-row index.  After chunk c the sums of the
// tiles below chunk_tile[c+1] are complete on this rank, so their reduction, their slice of the per-step RCCL
// all-reduce and their kicks + drift + repack run on a separate high-priority stream while the force
// streams sweep the remaining super-rows.  Chunk boundaries are identical on every rank (nb_plan.cpp).
bool chunked_ok(const nb_sim *s)
{
    const int sdt = s->is_f64 ? NB_F64 : NB_F32;
    return s->comm && !g_pc.direct_only && s->sym.enabled && s->sym.chunk_tile.size() > 2 && s->sym.packed_alt && !grid_mode(s->cfg.mode) &&
           s->logical[0] == sdt && s->logical[1] == sdt && s->logical[3] == sdt && s->have_acc;
}

int ensure_chunk_streams(nb_sim *s)
{
    if (s->cstream) return NB_OK;
    int least = 0, greatest = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    // stream priorities HURT here (measured: P = 2 stand-in 0.84 ms per step with a high-priority collective stream and
    // a low-priority second force stream against 0.68 with equal priorities): off unless NB_CHUNK_PRIO=1
    if (!s->knobs.chunk_prio) least = greatest = 0;
    HIPCHK(hipStreamCreateWithPriority(&s->cstream, hipStreamNonBlocking, greatest));
    HIPCHK(hipStreamCreateWithPriority(&s->fstream2, hipStreamNonBlocking, least));
    HIPCHK(hipEventCreateWithFlags(&s->ev_ready, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&s->ev_done, hipEventDisableTiming));
    for (auto &e : s->ev_force) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return NB_OK;
}

int launch_sym_range(nb_sim *s, const void *packed, int w0, int nw, hipStream_t st)
{
    if (nw <= 0) return NB_OK;
    const nb_config &c = s->cfg;
    const auto &sp = s->sym;
    if (s->is_f64) {
        HIPCHK(nb_launch_force_sym_f64((const double *)packed, sp.work + w0, nw, sp.rowslab, (double *)sp.colslab, sp.np,
                                       c.dim, sp.r, s->mass_uniform, 0, c.softening_sq, st));
        s->last_kernel = "force_sym_kernel<double";
    } else {
        const int hook = c.mode == NB_BFLOAT16 ? HOOK_BF16 : (c.mode == NB_FLOAT16 ? HOOK_F16 : HOOK_NONE);
        HIPCHK(nb_launch_force_sym_f32((const float *)packed, sp.work + w0, nw, sp.rowslab, (float *)sp.colslab, sp.np,
                                       c.dim, sp.r, s->mass_uniform, hook, (float)c.softening_sq, s->tab, (float)c.G,
                                       (float)s->mass_value, 0, st));
        s->last_kernel = "force_sym_kernel<float";
    }
    return NB_OK;
}

// `nsteps` leapfrog steps; on entry the accelerations are complete and, if pending_close, the closing half
// kick of the previous step is still due.  On exit everything is applied and the main stream has joined.
// one chunked step: forces of all chunks on the force streams, per chunk (in order) reduction + all-reduce slice + kicks /
// drift / repack into `pk_next` on the collective stream; joins back into the main stream
int enqueue_chunk_step(nb_sim *s, bool last, void *pk_cur, void *pk_next)
{
    const nb_config &c = s->cfg;
    auto &sp = s->sym;
    const int C = (int)sp.chunk_tile.size() - 1;
    const double half_dt = c.dt / 2;
    const double gfac = s->is_f64 ? c.G : (double)(float)c.G;
    const size_t el = s->is_f64 ? 8 : 4;
    double scale = 1.0;
    if (s->mass_uniform) scale = s->is_f64 ? c.G * s->mass_value : (double)((float)c.G * (float)s->mass_value);
    HIPCHK(hipEventRecord(s->ev_ready, s->stream));
    HIPCHK(hipStreamWaitEvent(s->fstream2, s->ev_ready, 0));
    for (int ch = 0; ch < C; ++ch) {
        hipStream_t fs = ((ch & 1) && !s->knobs.chunk_serial) ? s->fstream2 : s->stream;
        if (int rc = launch_sym_range(s, pk_cur, sp.chunk_work[ch], sp.chunk_work[ch + 1] - sp.chunk_work[ch], fs)) return rc;
        HIPCHK(hipEventRecord(s->ev_force[ch], fs));
    }
    for (int ch = 0; ch < C; ++ch) {
        const int p0 = sp.chunk_tile[ch] * sp.tile_b;
        const int p1 = std::min(sp.chunk_tile[ch + 1] * sp.tile_b, c.n);
        HIPCHK(hipStreamWaitEvent(s->cstream, s->ev_force[ch], 0));
        if (p1 > p0) {
            HIPCHK(nb_launch_reduce_sym(sp.rowslab, sp.colslab, sp.row_slot0, sp.row_nslots, sp.col_upto, sp.tile_b, c.n,
                                        sp.np, c.dim, s->is_f64, scale, s->acc, s->vel, half_dt, 0, s->pos, pk_cur,
                                        c.dt, s->cstream, p0, p1));
            char *a = (char *)s->acc + (size_t)p0 * c.dim * el;
            NCCLCHK(g_rccl.AllReduce(a, a, (size_t)(p1 - p0) * c.dim, s->is_f64 ? ncclDouble : ncclFloat, ncclSum, s->comm,
                                     s->cstream));
        }
        if (!last) {
            // closing kick, the next step's opening kick + drift, repack into the other buffer (the force
            // streams may still be reading this step's positions from pk_cur)
            const int pe = (ch == C - 1) ? sp.np : sp.chunk_tile[ch + 1] * sp.tile_b;
            HIPCHK(nb_launch_pack(s->pos, s->vel, s->acc, s->mass, pk_next, c.n, sp.np, c.dim, s->is_f64, 2, half_dt,
                                  c.dt, gfac, 0, s->cstream, p0, pe));
        } else if (p1 > p0) {
            HIPCHK(nb_launch_axpy((char *)s->vel + (size_t)p0 * c.dim * el, (char *)s->acc + (size_t)p0 * c.dim * el,
                                  half_dt, (int64_t)(p1 - p0) * c.dim, s->is_f64, s->cstream));
        }
    }
    HIPCHK(hipEventRecord(s->ev_done, s->cstream));
    HIPCHK(hipStreamWaitEvent(s->stream, s->ev_done, 0));
    return NB_OK;
}

// `nsteps` leapfrog steps; on entry the accelerations are complete and, if pending_close, the closing half
// kick of the previous step is still due.  On exit everything is applied and the main stream has joined.
// Pairs of steps (A -> B -> A buffers) are captured ONCE into a hipGraph and replayed: the fork / join of the three
// streams then costs one graph launch instead of a dozen event records / waits per step (NB_CHUNK_GRAPH=0: eager).
int step_chunked(nb_sim *s, int nsteps, bool pending_close)
{
    if (int rc = ensure_chunk_streams(s)) return rc;
    const nb_config &c = s->cfg;
    auto &sp = s->sym;
    const double half_dt = c.dt / 2;
    const double gfac = s->is_f64 ? c.G : (double)(float)c.G;
    // opening kick + drift + pack of the first step (whole range, main stream)
    HIPCHK(nb_launch_pack(s->pos, s->vel, s->acc, s->mass, sp.packed, c.n, sp.np, c.dim, s->is_f64, pending_close ? 2 : 1,
                          half_dt, c.dt, gfac, 0, s->stream));
    int t = 0;
    if (s->knobs.chunk_graph && nsteps >= 3) {
        const double key[4] = {c.G, c.softening_sq, c.dt, s->mass_uniform ? s->mass_value : -1.0};
        if (s->chunk_graph_exec && memcmp(key, s->chunk_graph_key, sizeof key) != 0) {
            (void)hipGraphExecDestroy(s->chunk_graph_exec);
            s->chunk_graph_exec = nullptr;
        }
        if (!s->chunk_graph_exec) {
            hipGraph_t graph = nullptr;
            HIPCHK(hipStreamBeginCapture(s->stream, hipStreamCaptureModeRelaxed));
            int rc = enqueue_chunk_step(s, false, sp.packed, sp.packed_alt);
            if (!rc) rc = enqueue_chunk_step(s, false, sp.packed_alt, sp.packed);
            const hipError_t e = hipStreamEndCapture(s->stream, &graph);
            if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
            HIPCHK(e);
            HIPCHK(hipGraphInstantiate(&s->chunk_graph_exec, graph, nullptr, nullptr, 0));
            (void)hipGraphDestroy(graph);
            memcpy(s->chunk_graph_key, key, sizeof key);
        }
        for (; t + 2 < nsteps; t += 2) HIPCHK(hipGraphLaunch(s->chunk_graph_exec, s->stream));
    }
    for (; t < nsteps; ++t) {
        const bool last = (t + 1 == nsteps);
        if (int rc = enqueue_chunk_step(s, last, sp.packed, sp.packed_alt)) return rc;
        if (!last) std::swap(sp.packed, sp.packed_alt);
    }
    s->logical[3] = acc_logical_dtype(s);
    return NB_OK;
}

// ---- small systems: one launch per step (nb_small.hip) ---------------------------------------------------------
constexpr int NB_SMALL_MAX_DEFAULT = 4096;      // fp64: 4.9 / 7.9 / 11.4 / 16.9 us per step at N = 1024 / 2048 / 3000 / 4096
bool small_ok(const nb_sim *s)
{
    const nb_config &c = s->cfg;
    const int sdt = s->is_f64 ? NB_F64 : NB_F32;
    // fp32 storage: above 3072 the one-launch kernel runs with 32 lanes per target and the tiled path is ahead
    // (measured us per step, one launch vs tiled: FLOAT32 N = 3200 13.7 / 10.7, 3584 14.9 / 10.9, 4096 15.9 / 13.4;
    // CUSTOM 3584 32.8 / 30.9; INT4 4096 39.5 / 40.1 -- equal; up to 3072 one launch wins or ties everywhere);
    // fp64 keeps it to 4096 (17.2 against 18.9)
    const int nmax = s->knobs.small_max > 0 ? s->knobs.small_max : (s->is_f64 ? NB_SMALL_MAX_DEFAULT : 3072);
    if (s->knobs.no_smalln || c.n > nmax || comm_active(s) || c.nranks != 1 || !s->have_acc) return false;
    if (grid_mode(c.mode) && (s->is_f64 || mode_levels(c) > NB_LUT_MIN || mode_levels(c) < 2)) return false;
    if (s->is_f64 != (c.mode == NB_FLOAT64)) return false;           // fp64 state under a cast mode: tuned one-sided kernel
    // masses: fp32-typed masses in an fp64 run enter the fp64 product exactly (no rounding of their own); half-typed
    // masses round the product to the half type and stay on the tuned kernels
    const bool mass_ok = s->logical[2] == sdt || (s->is_f64 && s->logical[2] == NB_F32);
    return s->logical[0] == sdt && s->logical[1] == sdt && mass_ok && s->logical[3] == sdt;
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
    const float eps2f = (float)c.softening_sq;
    if (!opened)
        HIPCHK(nb_launch_kick_drift(s->pos, s->vel, s->acc, c.dt / 2, c.dt, nd(s), s->is_f64, s->stream));
    for (int t = 0; t < nsteps; ++t) {
        const bool last = (t + 1 == nsteps);
        if (grid) {
            // this evaluation's grid: all-pairs max of r2 (small N: one launch) and the threshold / factor tables
            // one launch for both up to N = 2048 (measured, INT4: N = 1024 22.5 -> 18.6 us per step; N = 3000 30.8 vs
            // 31.7: there the fused kernel's arrival counter and longer source chunks cost more than the launch)
            if (mode_levels(c) <= NB_LUT_MIN && c.n <= 2048 && !s->knobs.no_small_fuse) {
                HIPCHK(nb_launch_r2max_tables((const float *)s->pos, s->geom, c.dim, eps2f, s->tab, mode_levels(c),
                                              (float)c.G, 0.01f, s->knobs.no_grid_fast ? 0 : 1, s->stream));
            } else {
                HIPCHK(nb_launch_r2max((const float *)s->pos, s->geom, c.dim, eps2f, s->tab, s->stream));
                HIPCHK(nb_launch_grid_tables(s->tab, mode_levels(c), (float)c.G, eps2f, 0.01f, nullptr, s->stream,
                                             s->knobs.no_grid_fast ? 0 : 1));
            }
        }
        // INT8 / INT4: the forces are snapped to their grid (and the kicks applied) by the finish launch
        const int kick = fq ? 0 : (last ? 1 : 2);
        HIPCHK(nb_launch_small_step(s->pos, s->pos_alt, s->vel, s->acc, s->mass, c.n, c.dim, s->is_f64, hook, c.G,
                                    c.softening_sq, c.dt / 2, c.dt, kick, lanes, s->stream, grid ? s->tab : nullptr,
                                    fq ? s->small_part : nullptr));
        if (fq)      // one min / max pair per workgroup of the force launch
            HIPCHK(nb_launch_force_quant_finish((float *)s->acc, nd(s), mode_levels(c), s->small_part,
                                                (c.n + NB_BLOCK / lanes - 1) / (NB_BLOCK / lanes), s->scalars, s->fbins,
                                                (float *)s->vel, (float *)s->pos, c.dt / 2, c.dt, last ? 1 : 2, s->stream));
        else if (!last)
            std::swap(s->pos, s->pos_alt);
    }
    s->last_kernel = "small_step_kernel";
    s->last_generic = false;
    return NB_OK;
}

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; (void)hipSetDevice(dev); }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

// =============================================================================================
// C-ABI
// =============================================================================================
extern "C" {

int nb_abi_version(void) { return NB_ABI_VERSION; }
const char *nb_last_error(void) { return g_err.c_str(); }

int nb_device_count(int32_t *count)
{
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *count = 0; return fail(NB_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = c;
    return NB_OK;
}

int nb_create(nb_sim **out, const nb_config *cfg)
{
    if (!out || !cfg) return fail(NB_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->n < 1) return fail(NB_ERR_INVALID, "n must be >= 1 (got %d)", cfg->n);
    if (cfg->dim != 2 && cfg->dim != 3) return fail(NB_ERR_INVALID, "dim must be 2 or 3 (got %d)", cfg->dim);
    if (cfg->mode < NB_FLOAT64 || cfg->mode > NB_CUSTOM) return fail(NB_ERR_INVALID, "bad precision mode %d", cfg->mode);
    if (cfg->nranks < 1 || cfg->rank < 0 || cfg->rank >= cfg->nranks)
        return fail(NB_ERR_INVALID, "bad shard %d of %d", cfg->rank, cfg->nranks);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(NB_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "0 devices" : hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(NB_ERR_NO_DEVICE, "device %d out of range [0,%d)", cfg->device, ndev);
    nb_sim *s = new nb_sim();
    s->cfg = *cfg;
    s->knobs = nb_read_knobs();
    DeviceGuard guard(cfg->device);
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) {
        delete s;
        return fail(NB_ERR_HIP, "hipStreamCreate failed");
    }
    compute_geometry(s);
    *out = s;
    return NB_OK;
}

int nb_destroy(nb_sim *s)
{
    if (!s) return NB_OK;
    DeviceGuard guard(s->cfg.device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    {
        std::lock_guard<std::mutex> lock(g_p2p_mu);
        if (g_p2p_last_stream == s->stream) g_p2p_last_stream = nullptr;
    }
    s->comm = nullptr;              // borrowed from the process (nb_comm_shutdown destroys it)
    for (void *p : {s->pos, s->vel, s->mass, s->acc, (void *)s->partial, s->staging, (void *)s->tab,
                    (void *)s->scratch, (void *)s->scalars, (void *)s->fbins, (void *)s->sym.work,
                    (void *)s->sym.row_slot0, (void *)s->sym.row_nslots, (void *)s->sym.col_upto,
                    (void *)s->sym.packed, (void *)s->sym.packed_alt, (void *)s->sym.rowslab, (void *)s->sym.colslab,
                    (void *)s->prune_cand, (void *)s->prune_rho, (void *)s->prune_state, s->metrics_scratch, s->gen_scalars, s->pos_alt, (void *)s->small_part,
                    (void *)s->sums64})
        if (p) (void)hipFree(p);
    if (s->prof_init)
        for (int i = 0; i < PROF_RING; ++i) { (void)hipEventDestroy(s->ev_start[i]); (void)hipEventDestroy(s->ev_stop[i]); }
    if (s->chunk_graph_exec) (void)hipGraphExecDestroy(s->chunk_graph_exec);
    for (hipEvent_t e : {s->ev_ready, s->ev_done, s->ev_force[0], s->ev_force[1], s->ev_force[2], s->ev_force[3]})
        if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : {s->fstream2, s->cstream, s->stream})
        if (st) (void)hipStreamDestroy(st);
    delete s;
    return NB_OK;
}

int nb_set_params(nb_sim *s, double G, double softening_sq, double dt)
{
    if (!s) return fail(NB_ERR_INVALID, "null handle");
    s->cfg.G = G;
    s->cfg.softening_sq = softening_sq;
    s->cfg.dt = dt;
    return NB_OK;
}

int nb_set_state(nb_sim *s, const void *pos, const void *vel, const void *mass, int dtype, int on_device)
{
    if (!s) return fail(NB_ERR_INVALID, "null handle");
    if (dtype < NB_F16 || dtype > NB_F64) return fail(NB_ERR_INVALID, "bad dtype %d", dtype);
    DeviceGuard guard(s->cfg.device);
    const int mode = s->cfg.mode;
    const bool want_f64 = (mode == NB_FLOAT64) || dtype == NB_F64 || (s->cfg.flags & NB_FLAG_F64_STORAGE);
    if (int rc = ensure_storage(s, s->have_storage ? s->is_f64 : want_f64)) return rc;
    if (dtype == NB_F64 && !s->is_f64)
        return fail(NB_ERR_UNSUPPORTED, "cannot upload fp64 data into fp32 state storage: create the handle with "
                                        "NB_FLAG_F64_STORAGE when any of positions / velocities / masses is fp64");
    if (pos) { if (int rc = upload(s, pos, dtype, on_device, s->pos, nd(s))) return rc; s->logical[0] = dtype; s->have_pos = true; }
    if (vel) { if (int rc = upload(s, vel, dtype, on_device, s->vel, nd(s))) return rc; s->logical[1] = dtype; s->have_vel = true; }
    if (mass) {
        if (int rc = upload(s, mass, dtype, on_device, s->mass, s->cfg.n)) return rc;
        s->logical[2] = dtype;
        s->have_mass = true;
        // uniform-mass detection (one min/max pass on the device per upload, not per step)
        double mm[2];
        HIPCHK(nb_launch_minmax_generic(s->mass, s->is_f64, s->cfg.n, 0, 0.0, s->scalars + 4, s->scalars + 8, s->stream));
        HIPCHK(hipMemcpyAsync(mm, s->scalars + 4, sizeof mm, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        s->mass_uniform = (mm[0] == mm[1]) && std::isfinite(mm[0]) && !s->knobs.no_uniform;
        s->mass_value = mm[0];
    }
    return NB_OK;
}

int nb_set_accelerations(nb_sim *s, const void *acc, int dtype, int on_device)
{
    if (!s || !acc) return fail(NB_ERR_INVALID, "null argument");
    if (!s->have_storage) return fail(NB_ERR_INVALID, "set the state first");
    if (dtype != NB_F32 && dtype != NB_F64) return fail(NB_ERR_UNSUPPORTED, "acceleration dtype %d", dtype);
    if (dtype == NB_F64 && !s->is_f64) return fail(NB_ERR_UNSUPPORTED, "fp64 accelerations on fp32 state");
    DeviceGuard guard(s->cfg.device);
    if (int rc = upload(s, acc, dtype, on_device, s->acc, nd(s))) return rc;
    s->logical[3] = dtype;
    s->have_acc = true;
    return NB_OK;
}

int nb_state_dtypes(nb_sim *s, int32_t dtypes[4])
{
    if (!s || !dtypes) return fail(NB_ERR_INVALID, "null argument");
    for (int i = 0; i < 4; ++i) dtypes[i] = s->logical[i];
    return NB_OK;
}

int nb_get_state(nb_sim *s, void *pos, void *vel, void *acc, void *mass, int on_device)
{
    if (!s) return fail(NB_ERR_INVALID, "null handle");
    if (!s->have_storage) return fail(NB_ERR_INVALID, "no state uploaded yet");
    DeviceGuard guard(s->cfg.device);
    if (pos) if (int rc = download(s, s->pos, s->logical[0], pos, on_device, nd(s))) return rc;
    if (vel) if (int rc = download(s, s->vel, s->logical[1], vel, on_device, nd(s))) return rc;
    if (acc) if (int rc = download(s, s->acc, s->logical[3], acc, on_device, nd(s))) return rc;
    if (mass) if (int rc = download(s, s->mass, s->logical[2], mass, on_device, s->cfg.n)) return rc;
    HIPCHK(hipStreamSynchronize(s->stream));
    return NB_OK;
}

int nb_compute_accelerations(nb_sim *s)
{
    if (!s) return fail(NB_ERR_INVALID, "null handle");
    DeviceGuard guard(s->cfg.device);
    return force_eval(s, false);
}

int nb_kick_drift(nb_sim *s)
{
    if (!s) return fail(NB_ERR_INVALID, "null handle");
    if (!s->have_vel || !s->have_pos || !s->have_acc) return fail(NB_ERR_INVALID, "state incomplete");
    DeviceGuard guard(s->cfg.device);
    HIPCHK(nb_launch_kick_drift(s->pos, s->vel, s->acc, s->cfg.dt / 2, s->cfg.dt, nd(s), s->is_f64, s->stream));
    s->logical[1] = promote(s->logical[1], s->logical[3]);
    s->logical[0] = promote(s->logical[0], s->logical[1]);
    return NB_OK;
}

int nb_kick(nb_sim *s)
{
    if (!s) return fail(NB_ERR_INVALID, "null handle");
    if (!s->have_vel || !s->have_acc) return fail(NB_ERR_INVALID, "state incomplete");
    DeviceGuard guard(s->cfg.device);
    HIPCHK(nb_launch_axpy(s->vel, s->acc, s->cfg.dt / 2, nd(s), s->is_f64, s->stream));
    s->logical[1] = promote(s->logical[1], s->logical[3]);
    return NB_OK;
}

int nb_step(nb_sim *s, int32_t nsteps)
{
    if (!s) return fail(NB_ERR_INVALID, "null handle");
    if (!s->have_vel || !s->have_pos || !s->have_mass) return fail(NB_ERR_INVALID, "state incomplete");
    if (!s->have_acc) return fail(NB_ERR_INVALID, "no accelerations yet: call nb_compute_accelerations first");
    DeviceGuard guard(s->cfg.device);
    bool pending_close = false;     // closing kick of the previous step still to be applied
    bool opened = false;            // the previous step's reduction already did this step's opening kick + drift
    bool packed_by_prev = false;    // ... and repacked the positions for the symmetric kernel
    for (int t = 0; t < nsteps; ++t) {
        // multi-GPU, pair-symmetric, settled dtypes: the remaining steps run as pipelined chunks
        if (!opened && chunked_ok(s)) return step_chunked(s, nsteps - t, pending_close);
        // small systems with settled dtypes: one launch per step
        if (!pending_close && small_ok(s)) return step_small(s, nsteps - t, opened);
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
        if (int rc = force_eval(s, true, packed_ready, may_defer ? &pending_close : nullptr, &opened)) return rc;
        packed_by_prev = opened && s->sym.enabled;
        s->logical[1] = promote(s->logical[1], s->logical[3]);
    }
    return NB_OK;
}

int nb_energy(nb_sim *s, double *kinetic, double *potential)
{
    if (!s) return fail(NB_ERR_INVALID, "null handle");
    DeviceGuard guard(s->cfg.device);
    const nb_config &c = s->cfg;
    double host[2] = {0, 0};
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
            HIPCHK(nb_launch_pack(s->pos, s->vel, s->acc, s->mass, sp.packed, c.n, sp.np, c.dim, s->is_f64, 0, 0.0, 0.0,
                                  1.0, s->is_f64 && s->logical[0] != NB_F64, s->stream));
            HIPCHK(nb_launch_potential_sym(sp.packed, sp.work, sp.nwork, s->scratch, sp.np, c.dim, sp.r, s->is_f64,
                                           s->logical[0] != NB_F64, s->logical[2], c.softening_sq, s->stream));
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
    if (kinetic) {
        // ke = 0.5 * (masses * v_sq).sum() in the promoted dtype of (velocities, masses)
        const int t = promote(s->logical[1], s->logical[2]);
        *kinetic = round_dt(t, round_dt(t, 0.5) * round_dt(t, host[0]));
    }
    if (potential) {
        const int t = promote(s->logical[0], s->logical[2]);
        *potential = round_dt(t, round_dt(t, -c.G) * round_dt(t, host[1]));
        // the reference multiplies by the triu mask before dividing by dist (simulation.py:189): the masked
        // entries are 0 / dist = NaN where dist == 0, i.e. on the whole diagonal when the softening rounds to
        // zero in the positions' dtype (softening 0; 1e-4 with float16 positions).  dist > 0 otherwise.
        if (c.n > 0 && round_dt(s->logical[0], c.softening_sq) == 0.0) *potential = std::nan("");
    }
    return NB_OK;
}

}  // extern "C"

namespace {
// distance-bin indices of rows [i0, i1) with the tables of the last evaluation, copied to the host
int bins_rows(nb_sim *s, int i0, int i1, int16_t *host)
{
    const int n = s->cfg.n;
    const size_t bytes = (size_t)(i1 - i0) * n * sizeof(int16_t);
    int16_t *dev = nullptr;
    HIPCHK(hipMalloc((void **)&dev, bytes));
    hipError_t e = nb_launch_d2bins((const float *)s->pos, n, s->cfg.dim, (float)s->cfg.softening_sq, s->tab, dev, s->stream,
                                    i0, i1);
    if (e == hipSuccess) e = hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    (void)hipFree(dev);
    HIPCHK(e);
    return NB_OK;
}
}  // namespace

extern "C" {

int nb_quant_bins_rows(nb_sim *s, int32_t i0, int32_t i1, int16_t *d2bins)
{
    if (!s || !d2bins) return fail(NB_ERR_INVALID, "null argument");
    if (!s->have_storage || s->is_f64 || !grid_mode(s->cfg.mode))
        return fail(NB_ERR_INVALID, "quant debug is only defined for the grid modes");
    if (i0 < 0 || i1 > s->cfg.n || i0 >= i1) return fail(NB_ERR_INVALID, "bad row range [%d, %d)", i0, i1);
    DeviceGuard guard(s->cfg.device);
    return bins_rows(s, i0, i1, d2bins);
}

int nb_quant_debug(nb_sim *s, double info[8], int16_t *d2bins, int16_t *fbins)
{
    if (!s) return fail(NB_ERR_INVALID, "null handle");
    if (!s->have_storage || s->is_f64 || !grid_mode(s->cfg.mode))
        return fail(NB_ERR_INVALID, "quant debug is only defined for the grid modes");
    if (s->last_generic) return fail(NB_ERR_UNSUPPORTED, "quant debug: the last evaluation ran on the generic per-pair path (no tables)");
    DeviceGuard guard(s->cfg.device);
    GridTables h;
    double mnmx[2];
    HIPCHK(hipMemcpyAsync(&h, s->tab, sizeof h, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipMemcpyAsync(mnmx, s->scalars, sizeof mnmx, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (info) {
        info[0] = h.lmin; info[1] = h.lmax; info[2] = mnmx[0]; info[3] = mnmx[1]; info[4] = h.r2max;
        info[5] = h.fast_ok; info[6] = h.fast_maxdev; info[7] = h.fast_maxrel;
    }
    if (d2bins)
        if (int rc = bins_rows(s, 0, s->cfg.n, d2bins)) return rc;
    if (fbins) {
        if (!s->fbins) return fail(NB_ERR_INVALID, "this mode does not quantise forces");
        HIPCHK(hipMemcpyAsync(fbins, s->fbins, (size_t)nd(s) * sizeof(int16_t), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    return NB_OK;
}

}  // extern "C"

// ---- tensor-level hooks ----------------------------------------------------------------------
namespace {
// Per-device scratch of the handle-less entry points (tensor-level hooks, nb_metrics_tensors): allocated once and
// grown on demand -- no hipMalloc / hipFree per call.  One call at a time per device (mutex); work is queued on the
// CALLER's stream (nb_set_hook_stream, thread-local) so that it is ordered with the caller's own device work and
// needs no synchronisation for device-resident buffers; a use on a different stream than the previous one first
// waits for that one's event.  Without a hook stream: the NULL stream and a blocking wait, as a plain C caller expects.
constexpr int MAX_DEV = 64;
struct DevScratch {
    std::mutex mu;
    void *buf = nullptr;
    size_t cap = 0;
    hipEvent_t last = nullptr;
    hipStream_t last_stream = nullptr;
    bool used = false;
};
DevScratch g_scratch[MAX_DEV];
thread_local hipStream_t g_hook_stream[MAX_DEV];
thread_local bool g_hook_stream_set[MAX_DEV];

int check_device(int device)
{
    static std::atomic<int> cached{-1};           // the device count does not change while the process lives
    int ndev = cached.load();
    if (ndev < 0) {
        if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
        cached.store(ndev);
    }
    if (ndev == 0) return fail(NB_ERR_NO_DEVICE, "no HIP device available; this library has no CPU fallback");
    if (device < 0 || device >= ndev || device >= MAX_DEV) return fail(NB_ERR_NO_DEVICE, "device %d out of range", device);
    return NB_OK;
}

// body(stream, scratch) queues its work; host_out / host_bytes: copied back from scratch + out_off afterwards
template <typename F>
int with_scratch(int device, size_t bytes, bool must_wait, F &&body)
{
    if (int rc = check_device(device)) return rc;
    DeviceGuard guard(device);
    DevScratch &ds = g_scratch[device];
    std::lock_guard<std::mutex> lock(ds.mu);
    const bool caller_stream = g_hook_stream_set[device];
    hipStream_t st = caller_stream ? g_hook_stream[device] : nullptr;
    if (bytes > ds.cap) {
        if (ds.buf) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(ds.buf)); ds.buf = nullptr; ds.cap = 0; }
        const size_t want = std::max<size_t>(bytes + bytes / 4, (size_t)1 << 20);
        HIPCHK(hipMalloc(&ds.buf, want));
        ds.cap = want;
    }
    if (!ds.last) HIPCHK(hipEventCreateWithFlags(&ds.last, hipEventDisableTiming));
    if (ds.used && ds.last_stream != st) HIPCHK(hipStreamWaitEvent(st, ds.last, 0));
    if (int rc = body(st, (char *)ds.buf)) return rc;
    HIPCHK(hipEventRecord(ds.last, st));
    ds.last_stream = st;
    ds.used = true;
    if (must_wait || !caller_stream) HIPCHK(hipStreamSynchronize(st));
    return NB_OK;
}

// run `body(d_in, d_out, d_scal, stream)` with device views of the caller's buffers
template <typename F>
int with_device_buffers(int device, const void *in, void *out, size_t in_bytes, size_t out_bytes, int on_device, F &&body)
{
    // scalars + min/max partials, then room for one GridTables (tensor-level _grid_quantize_safe)
    const size_t sc_bytes = (((2 + 2 * NB_MINMAX_BLOCKS) * sizeof(double) + 255) & ~(size_t)255) + ((sizeof(GridTables) + 255) & ~(size_t)255);
    const size_t in_al = (in_bytes + 255) & ~(size_t)255, out_al = (out_bytes + 255) & ~(size_t)255;
    const size_t total = sc_bytes + (on_device ? 0 : in_al + out_al);
    return with_scratch(device, total, !on_device, [&](hipStream_t st, char *scr) {
        const void *din = in;
        void *dout = out;
        if (!on_device) {
            din = scr + sc_bytes;
            dout = scr + sc_bytes + in_al;
            HIPCHK(hipMemcpyAsync((void *)din, in, in_bytes, hipMemcpyHostToDevice, st));
        }
        if (int rc = body(din, dout, (double *)scr, st)) return rc;
        if (!on_device) HIPCHK(hipMemcpyAsync(out, dout, out_bytes, hipMemcpyDeviceToHost, st));
        return (int)NB_OK;
    });
}

// metrics.py:25-156 on device arrays of storage type S (see nb_metrics.hip); results to the host
int run_metrics(int device, hipStream_t st, char *scratch, const void *pos, const void *vel, const void *mass, int n, int dim,
                bool storage_f64, bool arith_f64, int num_bins, const float *edges_host, double max_radius, double percentile,
                double G, int radius_only, double *curve_mean, int64_t *curve_count, double *scalars)
{
    const size_t work = nb_metrics_scratch_bytes(n, num_bins);
    double *out_dev = (double *)(scratch + work);
    float *edges_dev = (float *)(scratch + work + (size_t)(8 + 2 * 256) * sizeof(double));
    if (edges_host && num_bins > 0)
        HIPCHK(hipMemcpyAsync(edges_dev, edges_host, (size_t)(num_bins + 1) * sizeof(float), hipMemcpyHostToDevice, st));
    NbMetricsArgs a{};
    a.pos = pos; a.vel = vel; a.mass = mass;
    a.n = n; a.dim = dim;
    a.storage_f64 = storage_f64; a.arith_f64 = arith_f64;
    a.num_bins = num_bins;
    a.edges = (edges_host && num_bins > 0) ? edges_dev : nullptr;
    a.max_radius = max_radius;
    const long long k = (long long)((double)n * percentile / 100.0);     // int(len(radii) * percentile / 100)
    a.kth = (int)std::min<long long>(std::max<long long>(k, 0), n - 1);
    a.G = G;
    a.radius_only = radius_only;
    a.scratch = scratch;
    a.out = out_dev;
    HIPCHK(nb_launch_metrics(a, st));
    std::vector<double> host(5 + 2 * (size_t)num_bins);
    const int nb_eff = radius_only ? 0 : num_bins;
    HIPCHK(hipMemcpyAsync(host.data(), out_dev, (5 + 2 * (size_t)nb_eff) * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (scalars) {
        scalars[0] = host[0];
        scalars[1] = host[1];
        // bound_mask.float().mean(): fp32 sum of ones (exact) divided by N in fp32
        scalars[2] = radius_only ? 0.0 : (double)((float)host[2] / (float)n);
        scalars[3] = host[3];
        scalars[4] = host[4 + 2 * nb_eff];
    }
    for (int b = 0; b < nb_eff; ++b) {
        if (curve_mean) curve_mean[b] = host[4 + b];
        if (curve_count) curve_count[b] = (int64_t)host[4 + nb_eff + b];
    }
    return NB_OK;
}
size_t metrics_total_bytes(int n, int num_bins)
{
    return nb_metrics_scratch_bytes(n, num_bins) + (size_t)(8 + 2 * 256) * sizeof(double) + 260 * sizeof(float);
}
}  // namespace

extern "C" {

int nb_grid_quantize(int device, const void *in, void *out, int64_t count, int dtype, int levels, int on_device)
{
    if (dtype != NB_F32 && dtype != NB_F64) return fail(NB_ERR_UNSUPPORTED, "dtype %d", dtype);
    if (count < 1 || levels < 2) return fail(NB_ERR_INVALID, "count >= 1 and levels >= 2 required");
    const size_t bytes = (size_t)count * dt_size(dtype);
    return with_device_buffers(device, in, out, bytes, bytes, on_device, [&](const void *din, void *dout, double *sc, hipStream_t st) {
        HIPCHK(nb_launch_minmax_generic(din, dtype == NB_F64, count, 0, 0.0, sc, sc + 2, st));
        HIPCHK(nb_launch_grid_quantize(din, dout, dtype == NB_F64, count, levels, sc, st));
        return (int)NB_OK;
    });
}

int nb_grid_quantize_safe(int device, const void *in, void *out, int64_t count, int dtype, int levels, double min_val,
                          int on_device)
{
    if (dtype != NB_F32 && dtype != NB_F64) return fail(NB_ERR_UNSUPPORTED, "dtype %d", dtype);
    if (count < 1 || levels < 2) return fail(NB_ERR_INVALID, "count >= 1 and levels >= 2 required");
    const size_t bytes = (size_t)count * dt_size(dtype);
    static const bool slow_hook = getenv("NB_HOOK_ELEMENTWISE") != nullptr;      // A/B: library log / exp per element
    return with_device_buffers(device, in, out, bytes, bytes, on_device, [&](const void *din, void *dout, double *sc, hipStream_t st) {
        // (five short launches: below ~2 M elements the three library calls per element are quicker -- measured
        // 21.8 vs 30.4 us at 1024 x 1024, 200 vs 55 us at 4096 x 4096)
        if (dtype == NB_F32 && levels <= NB_MAX_LUT && count >= (int64_t)1 << 21 && !slow_hook) {
            // plain min / max (log is monotone), tables for these bounds, one lookup pass (nb_force.hip)
            GridTables *tab = (GridTables *)((char *)sc + (((2 + 2 * NB_MINMAX_BLOCKS) * sizeof(double) + 255) & ~(size_t)255));
            HIPCHK(nb_launch_minmax_generic(din, 0, count, 0, 0.0, sc, sc + 2, st));
            HIPCHK(nb_launch_grid_quantize_safe_tab((const float *)din, (float *)dout, count, levels, (float)min_val, sc, tab, st));
            return (int)NB_OK;
        }
        HIPCHK(nb_launch_minmax_generic(din, dtype == NB_F64, count, 1, min_val, sc, sc + 2, st));
        HIPCHK(nb_launch_grid_quantize_safe(din, dout, dtype == NB_F64, count, levels, min_val, sc, st));
        return (int)NB_OK;
    });
}

int nb_quantize_distance_squared(int device, const void *in, void *out, int64_t count, int dtype, int mode, int levels,
                                 double min_dist_sq, int on_device, int32_t *out_dtype)
{
    if (dtype != NB_F32 && dtype != NB_F64) return fail(NB_ERR_UNSUPPORTED, "dtype %d", dtype);
    if (mode < NB_FLOAT64 || mode > NB_CUSTOM) return fail(NB_ERR_INVALID, "bad mode %d", mode);
    if (mode >= NB_INT8_SIM) {
        const int L = mode == NB_INT8_SIM ? 256 : (mode == NB_INT4_SIM ? 16 : (levels > 0 ? levels : 64));
        if (out_dtype) *out_dtype = dtype;
        return nb_grid_quantize_safe(device, in, out, count, dtype, L, min_dist_sq, on_device);
    }
    const int odt = (mode == NB_FLOAT64) ? NB_F64 : NB_F32;
    if (out_dtype) *out_dtype = odt;
    return with_device_buffers(device, in, out, (size_t)count * dt_size(dtype), (size_t)count * dt_size(odt), on_device,
                               [&](const void *din, void *dout, double *, hipStream_t st) {
                                   HIPCHK(nb_launch_cast_hook(din, dtype, dout, mode, count, st));
                                   return (int)NB_OK;
                               });
}

int nb_quantize_force(int device, const void *in, void *out, int64_t count, int dtype, int mode, int levels, int on_device,
                      int32_t *out_dtype)
{
    if (dtype != NB_F32 && dtype != NB_F64) return fail(NB_ERR_UNSUPPORTED, "dtype %d", dtype);
    if (mode < NB_FLOAT64 || mode > NB_CUSTOM) return fail(NB_ERR_INVALID, "bad mode %d", mode);
    if (mode >= NB_INT8_SIM) {
        const int L = mode == NB_INT8_SIM ? 256 : (mode == NB_INT4_SIM ? 16 : (levels > 0 ? levels : 64));
        if (out_dtype) *out_dtype = dtype;
        return nb_grid_quantize(device, in, out, count, dtype, L, on_device);
    }
    // FLOAT64 / FLOAT32: identity; BF16 / F16: round trip (quantization.py:139-146)
    const bool identity = (mode == NB_FLOAT64 || mode == NB_FLOAT32);
    const int odt = identity ? dtype : NB_F32;
    if (out_dtype) *out_dtype = odt;
    const int cast_mode = identity ? (dtype == NB_F64 ? NB_FLOAT64 : NB_FLOAT32) : mode;
    return with_device_buffers(device, in, out, (size_t)count * dt_size(dtype), (size_t)count * dt_size(odt), on_device,
                               [&](const void *din, void *dout, double *, hipStream_t st) {
                                   HIPCHK(nb_launch_cast_hook(din, dtype, dout, cast_mode, count, st));
                                   return (int)NB_OK;
                               });
}

int nb_set_hook_stream(int device, void *stream, int enable)
{
    if (device < 0 || device >= MAX_DEV) return fail(NB_ERR_INVALID, "device %d out of range", device);
    g_hook_stream[device] = (hipStream_t)stream;
    g_hook_stream_set[device] = enable != 0;
    return NB_OK;
}

// ---- diagnostics (metrics.py:25-156) ------------------------------------------------------------
int nb_metrics(nb_sim *s, int32_t num_bins, const float *edges, double max_radius, double percentile, int32_t radius_only,
               double *curve_mean, int64_t *curve_count, double scalars[5])
{
    if (!s) return fail(NB_ERR_INVALID, "null handle");
    if (!s->have_pos || !s->have_vel || !s->have_mass) return fail(NB_ERR_INVALID, "state incomplete");
    if (num_bins < 0 || num_bins > 255) return fail(NB_ERR_INVALID, "num_bins must be in [0, 255]");
    DeviceGuard guard(s->cfg.device);
    const size_t need = metrics_total_bytes(s->cfg.n, num_bins);
    if (need > s->metrics_cap) {
        HIPCHK(hipStreamSynchronize(s->stream));
        if (s->metrics_scratch) (void)hipFree(s->metrics_scratch);
        s->metrics_scratch = nullptr;
        s->metrics_cap = 0;
        HIPCHK(hipMalloc(&s->metrics_scratch, need));
        s->metrics_cap = need;
    }
    // per-particle arithmetic in the dtype the reference's tensors have at this moment (fp32-typed values may sit
    // in fp64 storage: FLOAT64 mode before the first step)
    const bool arith_f64 = s->logical[0] == NB_F64;
    return run_metrics(s->cfg.device, s->stream, (char *)s->metrics_scratch, s->pos, s->vel, s->mass, s->cfg.n, s->cfg.dim,
                       s->is_f64, arith_f64, num_bins, edges, max_radius, percentile, s->cfg.G, radius_only, curve_mean,
                       curve_count, scalars);
}

int nb_metrics_tensors(int device, const void *pos, const void *vel, const void *mass, int32_t n, int32_t dim, int dtype,
                       int on_device, double G, int32_t num_bins, const float *edges, double max_radius, double percentile,
                       int32_t radius_only, double *curve_mean, int64_t *curve_count, double scalars[5])
{
    if (!pos || !vel || !mass) return fail(NB_ERR_INVALID, "null argument");
    if (dtype != NB_F32 && dtype != NB_F64) return fail(NB_ERR_UNSUPPORTED, "dtype %d", dtype);
    if (n < 1 || (dim != 2 && dim != 3)) return fail(NB_ERR_INVALID, "bad shape (%d, %d)", n, dim);
    if (num_bins < 0 || num_bins > 255) return fail(NB_ERR_INVALID, "num_bins must be in [0, 255]");
    const size_t el = dt_size(dtype);
    const size_t pb = ((size_t)n * dim * el + 255) & ~(size_t)255, mb = ((size_t)n * el + 255) & ~(size_t)255;
    const size_t stage = on_device ? 0 : 2 * pb + mb;
    return with_scratch(device, stage + metrics_total_bytes(n, num_bins), true, [&](hipStream_t st, char *scr) {
        const void *dp = pos, *dv = vel, *dm = mass;
        if (!on_device) {
            HIPCHK(hipMemcpyAsync(scr, pos, (size_t)n * dim * el, hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(scr + pb, vel, (size_t)n * dim * el, hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(scr + 2 * pb, mass, (size_t)n * el, hipMemcpyHostToDevice, st));
            dp = scr; dv = scr + pb; dm = scr + 2 * pb;
        }
        return run_metrics(device, st, scr + stage, dp, dv, dm, n, dim, dtype == NB_F64, dtype == NB_F64, num_bins, edges,
                           max_radius, percentile, G, radius_only, curve_mean, curve_count, scalars);
    });
}

// ---- multi-GPU -------------------------------------------------------------------------------
int nb_comm_unique_id(void *id_out, int32_t *id_bytes)
{
    if (!id_out || !id_bytes) return fail(NB_ERR_INVALID, "null argument");
    if (*id_bytes < (int32_t)sizeof(ncclUniqueId)) return fail(NB_ERR_INVALID, "id buffer too small (need %zu)", sizeof(ncclUniqueId));
    if (int rc = load_rccl()) return rc;
    ncclUniqueId id;
    NCCLCHK(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    *id_bytes = (int32_t)sizeof id;
    return NB_OK;
}

int nb_comm_init(nb_sim *s, const void *id, int32_t id_bytes)
{
    if (!s) return fail(NB_ERR_INVALID, "null handle");
    // NB_FLAG_SHARD_TIMING: one process stands in for one of nranks shards on a 1-rank communicator
    const int want_n = (s->cfg.flags & NB_FLAG_SHARD_TIMING) ? 1 : s->cfg.nranks;
    const int want_r = (s->cfg.flags & NB_FLAG_SHARD_TIMING) ? 0 : s->cfg.rank;
    std::lock_guard<std::mutex> lock(g_pc_mu);
    if (!g_pc.comm && !id && nb_p2p_state() == 2 && nb_p2p_nranks() == want_n && nb_p2p_device() == s->cfg.device) {
        // no unique id but an enabled direct all-reduce between exactly these ranks: a direct-only communicator
        g_pc.comm = (ncclComm_t)&g_direct_sentinel;
        g_pc.direct_only = true;
        g_pc.nranks = want_n;
        g_pc.rank = want_r;
        g_pc.device = s->cfg.device;
    }
    if (!g_pc.comm) {
        if (!id) return fail(NB_ERR_COMM, "no process communicator yet: the first nb_comm_init needs a unique id");
        if (id_bytes != (int32_t)sizeof(ncclUniqueId)) return fail(NB_ERR_INVALID, "bad id size %d", id_bytes);
        if (int rc = load_rccl()) return rc;
        DeviceGuard guard(s->cfg.device);
        ncclUniqueId uid;
        memcpy(&uid, id, sizeof uid);
        ncclComm_t comm = nullptr;
        NCCLCHK(g_rccl.CommInitRank(&comm, want_n, uid, want_r));
        g_pc.comm = comm;
        g_pc.nranks = want_n;
        g_pc.rank = want_r;
        g_pc.device = s->cfg.device;
    }
    if (g_pc.nranks != want_n || g_pc.rank != want_r || g_pc.device != s->cfg.device)
        return fail(NB_ERR_COMM, "the process communicator is rank %d of %d on device %d; this handle wants rank %d of %d "
                                 "on device %d (one process drives one GPU)",
                    g_pc.rank, g_pc.nranks, g_pc.device, want_r, want_n, s->cfg.device);
    s->comm = g_pc.comm;
    return NB_OK;
}

// ---- direct xGMI all-reduce (nb_p2p.hip): setup is driven by the host language, which owns the transport ----
int nb_comm_p2p_export(int32_t device, int32_t rank, int32_t nranks, int64_t capacity_bytes, void *handle_out,
                       int32_t *handle_bytes)
{
    if (!handle_out || !handle_bytes) return fail(NB_ERR_INVALID, "null argument");
    if (*handle_bytes < (int32_t)nb_p2p_handle_bytes())
        return fail(NB_ERR_INVALID, "handle buffer too small (need %zu)", nb_p2p_handle_bytes());
    if (capacity_bytes < 8) return fail(NB_ERR_INVALID, "capacity must be positive");
    if (int rc = check_device(device)) return rc;
    DeviceGuard guard(device);
    std::lock_guard<std::mutex> lock(g_pc_mu);
    HIPCHK(nb_p2p_export(device, rank, nranks, (size_t)capacity_bytes, handle_out));
    *handle_bytes = (int32_t)nb_p2p_handle_bytes();
    return NB_OK;
}

int nb_comm_p2p_import(const void *handles, int32_t nranks)
{
    if (!handles) return fail(NB_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lock(g_pc_mu);
    if (nb_p2p_device() < 0) return fail(NB_ERR_COMM, "nb_comm_p2p_export has not run in this process");
    DeviceGuard guard(nb_p2p_device());
    HIPCHK(nb_p2p_import(handles));
    if (nb_p2p_nranks() != nranks) return fail(NB_ERR_INVALID, "%d handles for %d ranks", nranks, nb_p2p_nranks());
    return NB_OK;
}

// Collective.  Integer-valued patterns (exact sums in any order) of several lengths, both element types, against the
// closed form; short timeout.  Returns NB_OK only if every element of every round was right on THIS rank; the
// caller combines the ranks' verdicts over its own transport and calls nb_comm_p2p_enable with the result.
int nb_comm_p2p_selftest(int32_t rounds, double timeout_s)
{
    std::lock_guard<std::mutex> lock(g_pc_mu);
    if (nb_p2p_state() < 1) return fail(NB_ERR_COMM, "direct all-reduce not attached");
    DeviceGuard guard(nb_p2p_device());
    const size_t cap = nb_p2p_capacity();
    void *scratch = nullptr;
    int *bad = nullptr;
    HIPCHK(hipMalloc(&scratch, cap));
    if (hipMalloc((void **)&bad, sizeof(int)) != hipSuccess) { (void)hipFree(scratch); return fail(NB_ERR_HIP, "hipMalloc"); }
    int rc = NB_OK, host_bad = 0, status = 0;
    hipError_t e = hipMemset(bad, 0, sizeof(int));
    const int P = nb_p2p_nranks();
    const size_t lengths[5] = {2, 14, (size_t)(510 * P + 6), 131072, cap / 8};
    for (int r = 0; r < rounds && e == hipSuccess && !status; ++r)
        for (int f64 = 0; f64 < 2 && e == hipSuccess && !status; ++f64) {
            for (int k = 0; k < 5 && e == hipSuccess; ++k) {
                size_t count = lengths[k];
                if (count * 8 > cap) count = cap / 8;
                if (!f64) count *= 2;                       // same bytes
                e = nb_p2p_selftest_round(scratch, count, f64, r * 10 + k, timeout_s, bad, nullptr);
                // the very first launch alone: if the peers cannot be reached, find out after ONE bounded wait
                if (r == 0 && f64 == 0 && k == 0 && e == hipSuccess) {
                    e = hipDeviceSynchronize();
                    if (e == hipSuccess) e = nb_p2p_status(&status);
                    if (status) break;
                }
            }
            // at most five launches are queued behind a barrier that may time out
            if (e == hipSuccess) e = hipDeviceSynchronize();
            if (e == hipSuccess && !status) e = nb_p2p_status(&status);
        }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(&host_bad, bad, sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess && !status) e = nb_p2p_status(&status);
    (void)hipFree(scratch);
    (void)hipFree(bad);
    if (e != hipSuccess) rc = fail(NB_ERR_HIP, "direct all-reduce self-test: %s", hipGetErrorString(e));
    else if (status) rc = fail(NB_ERR_COMM, "direct all-reduce self-test: a peer did not arrive within %.1f s", timeout_s);
    else if (host_bad) rc = fail(NB_ERR_COMM, "direct all-reduce self-test: %d wrong elements", host_bad);
    return rc;
}

int nb_comm_p2p_enable(int32_t on)
{
    std::lock_guard<std::mutex> lock(g_pc_mu);
    nb_p2p_enable(on != 0);
    return NB_OK;
}

int nb_comm_p2p_state(void)
{
    std::lock_guard<std::mutex> lock(g_pc_mu);
    return nb_p2p_state();
}

// Collective, for tests: all-reduce `count` host elements (NB_F32 / NB_F64) through the direct path, result back
// in place.  Works without an RCCL communicator.
int nb_comm_p2p_allreduce(void *host_inout, int64_t count, int32_t dtype, double timeout_s)
{
    if (!host_inout || count < 1) return fail(NB_ERR_INVALID, "bad argument");
    if (dtype != NB_F32 && dtype != NB_F64) return fail(NB_ERR_INVALID, "dtype must be NB_F32 or NB_F64");
    std::lock_guard<std::mutex> lock(g_pc_mu);
    if (nb_p2p_state() < 1) return fail(NB_ERR_COMM, "direct all-reduce not attached");
    const size_t bytes = (size_t)count * (dtype == NB_F64 ? 8 : 4);
    if (bytes > nb_p2p_capacity() || (dtype == NB_F32 && (count & 1)))
        return fail(NB_ERR_INVALID, "count does not fit the direct all-reduce (capacity %zu bytes, fp32 counts even)",
                    nb_p2p_capacity());
    DeviceGuard guard(nb_p2p_device());
    void *dst = nullptr;
    HIPCHK(hipMalloc(&dst, bytes));
    hipError_t e = hipMemcpy(nb_p2p_data(), host_inout, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = nb_p2p_allreduce(dst, (size_t)count, dtype == NB_F64, timeout_s, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(host_inout, dst, bytes, hipMemcpyDeviceToHost);
    int status = 0;
    if (e == hipSuccess) e = nb_p2p_status(&status);
    (void)hipFree(dst);
    if (e != hipSuccess) return fail(NB_ERR_HIP, "direct all-reduce: %s", hipGetErrorString(e));
    if (status) return fail(NB_ERR_COMM, "direct all-reduce: a peer did not arrive within %.1f s", timeout_s);
    return NB_OK;
}

// Collective, measurement only: average time of `iters` back-to-back all-reduces of this handle's force-vector size
// (which = 0: RCCL, 1: the direct path) on zeroed scratch, HIP events on the handle's stream.
int nb_comm_allreduce_time(nb_sim *s, int32_t which, int32_t iters, double *us_per_call)
{
    if (!us_per_call || iters < 1) return fail(NB_ERR_INVALID, "bad argument");
    if (!s) {
        // no handle: 1 MiB of doubles (the benchmark's force vector) on the NULL stream, through the attached direct
        // path (which = 1; needs no RCCL communicator) or the process communicator (which = 0).  The host language
        // uses the pair to decide which carrier is faster on this node.
        std::lock_guard<std::mutex> lock(g_pc_mu);
        if (which == 1 && nb_p2p_state() < 1) return fail(NB_ERR_COMM, "the direct all-reduce is not attached");
        if (which != 1 && (!g_pc.comm || g_pc.direct_only)) return fail(NB_ERR_COMM, "no RCCL communicator");
        DeviceGuard guard(which == 1 ? nb_p2p_device() : g_pc.device);
        const size_t cnt = which == 1 ? std::min<size_t>(131072, nb_p2p_capacity() / 8) : 131072;
        void *buf = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        HIPCHK(hipMalloc(&buf, cnt * 8));
        hipError_t e = hipMemset(buf, 0, cnt * 8);
        if (e == hipSuccess && which == 1) e = hipMemset(nb_p2p_data(), 0, cnt * 8);
        if (e == hipSuccess) e = hipEventCreate(&e0);
        if (e == hipSuccess) e = hipEventCreate(&e1);
        ncclResult_t nr = ncclSuccess;
        for (int pass = 0; pass < 2 && e == hipSuccess && nr == ncclSuccess; ++pass) {
            e = hipEventRecord(e0, nullptr);
            for (int i = 0; i < (pass == 0 ? 10 : iters) && e == hipSuccess && nr == ncclSuccess; ++i) {
                if (which == 1) e = nb_p2p_allreduce(buf, cnt, 1, 30.0, nullptr);
                else nr = g_rccl.AllReduce(buf, buf, cnt, ncclDouble, ncclSum, g_pc.comm, nullptr);
            }
            if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
            if (e == hipSuccess) e = hipDeviceSynchronize();
        }
        float ms = 0.0f;
        if (e == hipSuccess && nr == ncclSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        (void)hipFree(buf);
        if (nr != ncclSuccess) return fail(NB_ERR_COMM, "ncclAllReduce failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(nr) : "?");
        if (e != hipSuccess) return fail(NB_ERR_HIP, "all-reduce timing: %s", hipGetErrorString(e));
        *us_per_call = 1e3 * ms / iters;
        return NB_OK;
    }
    if (!s->comm) return fail(NB_ERR_COMM, "the handle has no communicator");
    if (which != 1 && g_pc.direct_only) return fail(NB_ERR_COMM, "no RCCL communicator");
    DeviceGuard guard(s->cfg.device);
    const int64_t cnt = nd(s);
    const size_t bytes = (size_t)cnt * (s->is_f64 ? 8 : 4);
    if (which == 1) {
        if (nb_p2p_state() != 2 || bytes > nb_p2p_capacity() || (!s->is_f64 && (cnt & 1)))
            return fail(NB_ERR_COMM, "the direct all-reduce is not enabled for this vector");
    }
    void *buf = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIPCHK(hipMalloc(&buf, bytes));
    hipError_t e = hipMemsetAsync(buf, 0, bytes, s->stream);
    if (e == hipSuccess && which == 1) e = hipMemsetAsync(nb_p2p_data(), 0, bytes, s->stream);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    ncclResult_t nr = ncclSuccess;
    for (int pass = 0; pass < 2 && e == hipSuccess && nr == ncclSuccess; ++pass) {       // pass 0 warms up
        const int reps = pass == 0 ? 10 : iters;
        e = hipEventRecord(e0, s->stream);
        for (int i = 0; i < reps && e == hipSuccess && nr == ncclSuccess; ++i) {
            if (which == 1) e = nb_p2p_allreduce(buf, (size_t)cnt, s->is_f64, P2P_STEP_TIMEOUT_S, s->stream);
            else nr = g_rccl.AllReduce(buf, buf, (size_t)cnt, s->is_f64 ? ncclDouble : ncclFloat, ncclSum, s->comm, s->stream);
        }
        if (e == hipSuccess) e = hipEventRecord(e1, s->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    }
    float ms = 0.0f;
    if (e == hipSuccess && nr == ncclSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(buf);
    if (nr != ncclSuccess) return fail(NB_ERR_COMM, "ncclAllReduce failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(nr) : "?");
    if (e != hipSuccess) return fail(NB_ERR_HIP, "all-reduce timing: %s", hipGetErrorString(e));
    *us_per_call = 1e3 * ms / iters;
    return NB_OK;
}

// Tests: the direct all-reduce kernel between `nranks` VIRTUAL ranks of this one process (own regions, own streams),
// integer patterns against the closed form; see nb_p2p.hip.  *bad = wrong elements (+1e6 per timed-out rank).
int nb_comm_p2p_virtual_test(int32_t device, int32_t nranks, int64_t count, int32_t dtype, int32_t concurrent, int32_t iters,
                             double timeout_s, int32_t *bad, double *us_per_call)
{
    if (!bad || count < 1 || (dtype != NB_F32 && dtype != NB_F64)) return fail(NB_ERR_INVALID, "bad argument");
    if (int rc = check_device(device)) return rc;
    DeviceGuard guard(device);
    int b = 0;
    double us = 0.0;
    HIPCHK(nb_p2p_virtual(nranks, (size_t)count, dtype == NB_F64, concurrent, iters, timeout_s, &b, &us));
    *bad = b;
    if (us_per_call) *us_per_call = us;
    return NB_OK;
}

// First half of a shutdown: wait for this device's work.  The host language then runs a barrier of its own (no rank
// may free buffers a peer's kernel still reads) and calls nb_comm_shutdown.
int nb_comm_quiesce(void)
{
    std::lock_guard<std::mutex> lock(g_pc_mu);
    const int dev = g_pc.comm ? g_pc.device : nb_p2p_device();
    if (dev < 0) return NB_OK;
    DeviceGuard guard(dev);
    HIPCHK(hipDeviceSynchronize());
    return NB_OK;
}

int nb_comm_ready(void)
{
    std::lock_guard<std::mutex> lock(g_pc_mu);
    return g_pc.comm ? g_pc.nranks : 0;
}

int nb_comm_shutdown(void)
{
    std::lock_guard<std::mutex> lock(g_pc_mu);
    if (nb_p2p_device() >= 0) {
        DeviceGuard guard(nb_p2p_device());
        (void)hipDeviceSynchronize();
        nb_p2p_shutdown();
    }
    if (!g_pc.comm) return NB_OK;
    DeviceGuard guard(g_pc.device);
    (void)hipDeviceSynchronize();
    ncclComm_t comm = g_pc.comm;
    const bool direct_only = g_pc.direct_only;
    g_pc = ProcComm();
    if (!direct_only && g_rccl.CommDestroy) NCCLCHK(g_rccl.CommDestroy(comm));
    return NB_OK;
}

// ---- planning without a device ---------------------------------------------------------------
int nb_plan_debug(const nb_config *cfg, int32_t is_f64, int32_t multi, int32_t cus, int32_t info[16], int32_t *work,
                  int64_t work_capacity, int32_t *row_slot0, int32_t *row_nslots, int32_t *col_upto,
                  int32_t *chunk_work, int32_t *chunk_tile)
{
    if (!cfg || !info) return fail(NB_ERR_INVALID, "null argument");
    if (cfg->n < 1 || (cfg->dim != 2 && cfg->dim != 3) || cfg->nranks < 1 || cfg->rank < 0 || cfg->rank >= cfg->nranks)
        return fail(NB_ERR_INVALID, "bad configuration");
    PlanInput in;
    in.n = cfg->n; in.dim = cfg->dim; in.mode = cfg->mode; in.flags = cfg->flags; in.rank = cfg->rank; in.nranks = cfg->nranks;
    in.is_f64 = is_f64 != 0;
    in.multi = multi != 0;
    in.cus = cus > 0 ? cus : 256;
    SymPlanHost h;
    nb_plan_sym(in, nb_read_knobs(), h);
    const int nchunks = h.enabled ? (int)h.chunk_tile.size() - 1 : 0;
    const int32_t vals[16] = {h.enabled, h.r, h.tile_b, h.tiles, h.np, (int32_t)h.work.size(), h.nslots, h.ncol, h.cl, nchunks,
                              (int32_t)(h.col_bytes >> 20), (int32_t)(h.row_bytes >> 20), 0, 0, 0, 0};
    memcpy(info, vals, sizeof vals);
    if (!h.enabled) return NB_OK;
    if (work) {
        if (work_capacity < (int64_t)h.work.size()) return fail(NB_ERR_INVALID, "work buffer too small (%zu items)", h.work.size());
        static_assert(sizeof(SymWork) == 8 * sizeof(int32_t), "SymWork is eight ints");
        memcpy(work, h.work.data(), h.work.size() * sizeof(SymWork));
    }
    if (row_slot0) memcpy(row_slot0, h.row_slot0.data(), h.tiles * sizeof(int32_t));
    if (row_nslots) memcpy(row_nslots, h.row_nslots.data(), h.tiles * sizeof(int32_t));
    if (col_upto) memcpy(col_upto, h.col_upto.data(), h.tiles * sizeof(int32_t));
    if (chunk_work) memcpy(chunk_work, h.chunk_work.data(), (nchunks + 1) * sizeof(int32_t));
    if (chunk_tile) memcpy(chunk_tile, h.chunk_tile.data(), (nchunks + 1) * sizeof(int32_t));
    return NB_OK;
}

// ---- measurement -----------------------------------------------------------------------------
int nb_kernel_time(nb_sim *s, double *total_ms, int32_t *launches)
{
    if (!s) return fail(NB_ERR_INVALID, "null handle");
    DeviceGuard guard(s->cfg.device);
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

const char *nb_force_kernel_name(nb_sim *s) { return s ? s->last_kernel : "none"; }

int nb_synchronize(nb_sim *s)
{
    if (!s) return fail(NB_ERR_INVALID, "null handle");
    DeviceGuard guard(s->cfg.device);
    HIPCHK(hipStreamSynchronize(s->stream));
    if (s->used_p2p) {
        int st = 0;
        HIPCHK(nb_p2p_status(&st));
        if (st) return fail(NB_ERR_COMM, "direct xGMI all-reduce: a peer did not arrive within %.0f s (results invalid)",
                            P2P_STEP_TIMEOUT_S);
    }
    return NB_OK;
}

}  // extern "C"
