// nb_api.cpp -- C-ABI of the MI355X N-body engine (include/nbody_amd.h): the entry points of a simulation handle,
// their argument checks, and the movement of state between caller buffers and the handle's device storage.
//
// Host-side orchestration only: owns the per-handle device state and tracks the reference's dtype state machine
// (SURVEY.md section 8a); kernel selection and sequencing live in nb_step.cpp, the communicator in nb_comm.cpp, the
// handle-less tensor hooks in nb_hooks.cpp (shared declarations: nb_state.h).  No arithmetic of the hot path runs
// on the host; without a HIP device every entry point fails.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "nb_state.h"

namespace nbhost {

std::string &last_error_string()
{
    thread_local std::string err;
    return err;
}

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_string() = buf;
    return code;
}

namespace {
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
}  // namespace

double round_dt(int dt, double x)
{
    if (dt == NB_F64) return x;
    const double f = (double)(float)x;
    if (dt == NB_F16) return round_small(f, 10, -14, 15);
    if (dt == NB_BF16) return round_small(f, 7, -126, 127);
    return f;
}

}  // namespace nbhost

using namespace nbhost;

namespace {

// All device buffers a handle needs from its first upload on come out of ONE allocation (the reference's scripts build
// many short simulations: with ~20 hipMalloc / hipFree pairs, constructing and closing a handle cost 0.36 + 0.30 ms
// against the 1.0 ms of a whole 200-tick run at N = 1024; tools/create_timing.py).  Buffers that only some call
// sequences need (metrics, generic kernel, small-system ping-pong, bin read-out, multi-GPU sums) stay lazy allocations
// of their own.
struct ArenaRequest { void **target; size_t bytes; };

// Streams and small arenas of closed handles are kept for the next handle of the same device (process-level, bounded:
// arenas up to 64 MiB each, 256 MiB / 16 entries in total): hipFree and hipStreamDestroy wait for the whole device and
// cost ~0.1 ms apiece, hipMalloc / hipStreamCreate about as much.  nb_cache_trim() releases everything; NB_NO_CACHE=1
// turns the cache off.  A reused arena is cleared before use.
struct HandleCache {
    static constexpr int MAX_DEV = 64;
    static constexpr size_t MAX_ARENA = (size_t)64 << 20, MAX_TOTAL = (size_t)256 << 20, MAX_ENTRIES = 16;
    struct Arena { void *p; size_t bytes; };
    std::mutex mu;
    std::vector<hipStream_t> streams[MAX_DEV];
    std::vector<Arena> arenas[MAX_DEV];
    int cus[MAX_DEV] = {0};
    size_t total = 0, entries = 0;
    bool off = getenv("NB_NO_CACHE") != nullptr;
};
// never destroyed: handles may still be closed (Python finalisers) while the process runs its static destructors
HandleCache &g_cache = *new HandleCache;

int device_cus(int device)
{
    if (device < 0 || device >= HandleCache::MAX_DEV) return 256;
    std::lock_guard<std::mutex> lock(g_cache.mu);
    if (!g_cache.cus[device]) {
        hipDeviceProp_t prop;       // (slow call: once per device and process)
        g_cache.cus[device] = hipGetDeviceProperties(&prop, device) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    return g_cache.cus[device];
}

hipError_t arena_acquire(int device, size_t bytes, void **out, size_t *got, bool *reused)
{
    *reused = false;
    if (!g_cache.off && device >= 0 && device < HandleCache::MAX_DEV) {
        std::lock_guard<std::mutex> lock(g_cache.mu);
        auto &v = g_cache.arenas[device];
        int best = -1;
        for (int i = 0; i < (int)v.size(); ++i)
            if (v[i].bytes >= bytes && v[i].bytes <= 2 * bytes + ((size_t)1 << 20) && (best < 0 || v[i].bytes < v[best].bytes)) best = i;
        if (best >= 0) {
            *out = v[best].p; *got = v[best].bytes;
            g_cache.total -= v[best].bytes; g_cache.entries--;
            v.erase(v.begin() + best);
            *reused = true;
            return hipSuccess;
        }
    }
    *got = bytes;
    return hipMalloc(out, bytes);
}

void arena_release(int device, void *p, size_t bytes)
{
    if (!p) return;
    if (!g_cache.off && device >= 0 && device < HandleCache::MAX_DEV && bytes <= HandleCache::MAX_ARENA) {
        std::lock_guard<std::mutex> lock(g_cache.mu);
        if (g_cache.total + bytes <= HandleCache::MAX_TOTAL && g_cache.entries < HandleCache::MAX_ENTRIES) {
            g_cache.arenas[device].push_back({p, bytes});
            g_cache.total += bytes; g_cache.entries++;
            return;
        }
    }
    (void)hipFree(p);
}

hipError_t stream_acquire(int device, hipStream_t *out)
{
    if (!g_cache.off && device >= 0 && device < HandleCache::MAX_DEV) {
        std::lock_guard<std::mutex> lock(g_cache.mu);
        auto &v = g_cache.streams[device];
        if (!v.empty()) { *out = v.back(); v.pop_back(); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

void stream_release(int device, hipStream_t st)       // `st` is idle (the caller synchronised it)
{
    if (!st) return;
    if (!g_cache.off && device >= 0 && device < HandleCache::MAX_DEV) {
        std::lock_guard<std::mutex> lock(g_cache.mu);
        if (g_cache.streams[device].size() < 8) { g_cache.streams[device].push_back(st); return; }
    }
    (void)hipStreamDestroy(st);
}
constexpr size_t ARENA_ALIGN = 256;
inline size_t arena_round(size_t b) { return (b + ARENA_ALIGN - 1) / ARENA_ALIGN * ARENA_ALIGN; }

// Storage of a handle + the work list of the pair-symmetric kernel (planned on the host: nb_plan.cpp, device-free and
// testable on its own through nb_plan_debug; mirrored here into device buffers).
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
    const nb_config &c = s->cfg;
    const size_t el = f64 ? 8 : 4;
    const size_t cnt = (size_t)nd(s);

    // ---- the plan first: its sizes are part of the allocation ----------------------------------------------------
    auto &sp = s->sym;
    sp.enabled = false;
    PlanInput in;
    in.n = c.n; in.dim = c.dim; in.mode = c.mode; in.flags = c.flags; in.rank = c.rank; in.nranks = c.nranks;
    in.is_f64 = s->is_f64;
    in.multi = comm_active(s);
    in.cus = device_cus(c.device);
    SymPlanHost h;
    nb_plan_sym(in, s->knobs, h);
    if (h.enabled) {
        sp.r = h.r; sp.tile_b = h.tile_b; sp.tiles = h.tiles; sp.np = h.np;
        sp.nwork = (int)h.work.size();
        sp.nslots = h.nslots;
        sp.rowsplit = h.rowsplit > 0;
    }
    const size_t pe_blocks = (size_t)((c.n + NB_BLOCK - 1) / NB_BLOCK) * s->geom.nchunks;
    s->scratch_elems = std::max<size_t>(std::max<size_t>(pe_blocks, 1024), h.enabled ? h.work.size() : 0);
    const bool prune = !f64 && grid_mode(c.mode);

    // ---- one allocation ---------------------------------------------------------------------------------------------
    // zero-initialised block first (tab, scalars, acc: one memset), then the block initialised from the host (plan
    // tables, prune state: one copy), then everything else
    void *plan_blob = nullptr;
    const size_t plan_ints = h.enabled ? (size_t)sp.tiles * 3 : 0;
    const size_t plan_bytes = h.enabled ? arena_round(h.work.size() * sizeof(SymWork)) + arena_round(plan_ints * sizeof(int)) : 0;
    const size_t blob_bytes = plan_bytes + (prune ? arena_round(sizeof(PruneState)) : 0);
    const size_t scalars_bytes = (8 + 2 * NB_MINMAX_BLOCKS) * sizeof(double);   // [0..7] results, [8 ...] partials of the two-stage min/max
    std::vector<ArenaRequest> req = {
        {(void **)&s->tab, sizeof(GridTables)}, {(void **)&s->scalars, scalars_bytes}, {&s->acc, cnt * el},
        {&plan_blob, blob_bytes},
        {&s->pos, cnt * el}, {&s->vel, cnt * el}, {&s->mass, (size_t)c.n * el}, {&s->staging, cnt * 8},
        {(void **)&s->partial, (size_t)s->geom.nchunks * cnt * sizeof(double)},
        {(void **)&s->scratch, s->scratch_elems * sizeof(double)},
    };
    if (force_quant_mode(c)) req.push_back({(void **)&s->fbins, cnt * sizeof(int16_t)});
    if (prune) {
        req.push_back({(void **)&s->prune_cand, cnt * sizeof(float)});
        req.push_back({(void **)&s->prune_rho, (size_t)c.n * sizeof(float)});
        req.push_back({(void **)&s->prune_idx, (size_t)c.n * sizeof(int)});
    }
    if (h.enabled) {
        req.push_back({&sp.packed, h.packed_bytes});
        req.push_back({(void **)&sp.rowslab, h.row_bytes});
        req.push_back({&sp.colslab, h.col_bytes});
    }
    size_t total = 0;
    for (const ArenaRequest &r : req) total += arena_round(r.bytes);
    bool reused = false;
    HIPCHK(arena_acquire(c.device, std::max<size_t>(total, ARENA_ALIGN), &s->arena, &s->arena_bytes, &reused));
    size_t off = 0;
    for (const ArenaRequest &r : req) {
        *r.target = r.bytes ? (void *)((char *)s->arena + off) : nullptr;
        off += arena_round(r.bytes);
    }
    const size_t zero_bytes = arena_round(sizeof(GridTables)) + arena_round(scalars_bytes) + arena_round(cnt * el);
    // a reused arena (<= 64 MiB) is cleared as a whole: every buffer starts from zeros, as on freshly mapped memory --
    // except the host-initialised block, which the synchronous copy below fills (this memset is asynchronous on the
    // handle's non-blocking stream and must not touch what that copy writes)
    HIPCHK(hipMemsetAsync(s->arena, 0, zero_bytes, s->stream));
    if (reused) {
        const size_t rest = zero_bytes + arena_round(blob_bytes);
        if (s->arena_bytes > rest) HIPCHK(hipMemsetAsync((char *)s->arena + rest, 0, s->arena_bytes - rest, s->stream));
    }
    if (blob_bytes) {
        std::vector<char> host(blob_bytes, 0);
        size_t o = 0;
        if (h.enabled) {
            sp.work = (SymWork *)((char *)plan_blob + o);
            std::memcpy(host.data() + o, h.work.data(), h.work.size() * sizeof(SymWork));
            o += arena_round(h.work.size() * sizeof(SymWork));
            int *ints = (int *)((char *)plan_blob + o);
            sp.row_slot0 = ints; sp.row_nslots = ints + sp.tiles; sp.col_upto = ints + 2 * (size_t)sp.tiles;
            std::memcpy(host.data() + o, h.row_slot0.data(), sp.tiles * sizeof(int));
            std::memcpy(host.data() + o + sp.tiles * sizeof(int), h.row_nslots.data(), sp.tiles * sizeof(int));
            std::memcpy(host.data() + o + 2 * (size_t)sp.tiles * sizeof(int), h.col_upto.data(), sp.tiles * sizeof(int));
            o += arena_round(plan_ints * sizeof(int));
        }
        if (prune) {
            s->prune_state = (PruneState *)((char *)plan_blob + o);
            PruneState init{};
            for (int k = 0; k < 3; ++k) { init.box_min[k] = 0xffffffffu; init.box_max[k] = 0u; }
            std::memcpy(host.data() + o, &init, sizeof init);
        }
        HIPCHK(hipMemcpy(plan_blob, host.data(), blob_bytes, hipMemcpyHostToDevice));
    }
    sp.enabled = h.enabled;
    s->have_storage = true;
    return NB_OK;
}

// copy `count` elements of dtype `dt` from a caller buffer into state storage (with conversion), asynchronously on the
// handle's stream.  The caller owns `src` and may free or overwrite it as soon as the ENTRY POINT returns (a torch
// temporary goes back to the caching allocator): every entry point that uploads ends with ONE hipStreamSynchronize
// (three arrays used to cost three).  The staging buffer is reused by the next array in stream order.
int upload(nb_sim *s, const void *src, int dt, int on_device, void *dst, int64_t count)
{
    const int sdt = s->is_f64 ? NB_F64 : NB_F32;
    const size_t bytes = (size_t)count * dt_size(dt);
    if (dt == sdt) {
        HIPCHK(hipMemcpyAsync(dst, src, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                              s->stream));
        return NB_OK;
    }
    const void *dsrc = src;
    if (!on_device) {
        HIPCHK(hipMemcpyAsync(s->staging, src, bytes, hipMemcpyHostToDevice, s->stream));
        dsrc = s->staging;
    }
    HIPCHK(nb_launch_convert(dsrc, dt, dst, sdt, count, s->stream));
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

}  // namespace

// =============================================================================================
// C-ABI
// =============================================================================================
extern "C" {

int nb_abi_version(void) { return NB_ABI_VERSION; }
const char *nb_last_error(void) { return last_error_string().c_str(); }

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
    if (stream_acquire(cfg->device, &s->stream) != hipSuccess) {
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
                    (void *)s->sym.packed, (void *)s->sym.rowslab, (void *)s->sym.colslab,
                    (void *)s->prune_cand, (void *)s->prune_rho, (void *)s->prune_idx, (void *)s->prune_state, s->metrics_scratch, s->gen_scalars, s->pos_alt, (void *)s->small_part,
                    (void *)s->sums64, (void *)s->bin_out})
        if (p && !(s->arena && (char *)p >= (char *)s->arena && (char *)p < (char *)s->arena + s->arena_bytes)) (void)hipFree(p);
    arena_release(s->cfg.device, s->arena, s->arena_bytes);
    if (s->prof_init)
        for (int i = 0; i < PROF_RING; ++i) { (void)hipEventDestroy(s->ev_start[i]); (void)hipEventDestroy(s->ev_stop[i]); }
    stream_release(s->cfg.device, s->stream);
    delete s;
    return NB_OK;
}

int nb_cache_trim(int64_t *released_bytes)
{
    size_t freed = 0;
    std::lock_guard<std::mutex> lock(g_cache.mu);
    for (int d = 0; d < HandleCache::MAX_DEV; ++d) {
        if (g_cache.arenas[d].empty() && g_cache.streams[d].empty()) continue;
        DeviceGuard guard(d);
        for (auto &a : g_cache.arenas[d]) { (void)hipFree(a.p); freed += a.bytes; }
        for (hipStream_t st : g_cache.streams[d]) (void)hipStreamDestroy(st);
        g_cache.arenas[d].clear();
        g_cache.streams[d].clear();
    }
    g_cache.total = 0; g_cache.entries = 0;
    if (released_bytes) *released_bytes = (int64_t)freed;
    return NB_OK;
}

int nb_set_params(nb_sim *s, double G, double softening_sq, double dt)
{
    if (!s) return fail(NB_ERR_INVALID, "null handle");
    s->spec_open = false;          // state / parameters change: the speculative next positions are void
    s->cfg.G = G;
    s->cfg.softening_sq = softening_sq;
    s->cfg.dt = dt;
    return NB_OK;
}

int nb_set_state(nb_sim *s, const void *pos, const void *vel, const void *mass, int dtype, int on_device)
{
    if (!s) return fail(NB_ERR_INVALID, "null handle");
    s->spec_open = false;          // state / parameters change: the speculative next positions are void
    if (dtype < NB_F16 || dtype > NB_F64) return fail(NB_ERR_INVALID, "bad dtype %d", dtype);
    DeviceGuard guard(s->cfg.device);
    const int mode = s->cfg.mode;
    const bool want_f64 = (mode == NB_FLOAT64) || dtype == NB_F64 || (s->cfg.flags & NB_FLAG_F64_STORAGE);
    if (int rc = ensure_storage(s, s->have_storage ? s->is_f64 : want_f64)) return rc;
    if (dtype == NB_F64 && !s->is_f64)
        return fail(NB_ERR_UNSUPPORTED, "cannot upload fp64 data into fp32 state storage: create the handle with "
                                        "NB_FLAG_F64_STORAGE when any of positions / velocities / masses is fp64");
    if (pos) {
        if (int rc = upload(s, pos, dtype, on_device, s->pos, nd(s))) return rc;
        s->logical[0] = dtype;
        s->have_pos = true;
        s->prune_seeded = false;         // new positions: the next grid evaluation searches its farthest pair from scratch
    }
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
    } else {
        HIPCHK(hipStreamSynchronize(s->stream));      // the copies have consumed the caller's buffers
    }
    return NB_OK;
}

int nb_set_accelerations(nb_sim *s, const void *acc, int dtype, int on_device)
{
    if (!s || !acc) return fail(NB_ERR_INVALID, "null argument");
    s->spec_open = false;
    if (!s->have_storage) return fail(NB_ERR_INVALID, "set the state first");
    if (dtype != NB_F32 && dtype != NB_F64) return fail(NB_ERR_UNSUPPORTED, "acceleration dtype %d", dtype);
    if (dtype == NB_F64 && !s->is_f64) return fail(NB_ERR_UNSUPPORTED, "fp64 accelerations on fp32 state");
    DeviceGuard guard(s->cfg.device);
    if (int rc = upload(s, acc, dtype, on_device, s->acc, nd(s))) return rc;
    HIPCHK(hipStreamSynchronize(s->stream));
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
    return p2p_check(s);       // a timed-out barrier of the direct all-reduce makes the state garbage: say so
}

int nb_compute_accelerations(nb_sim *s)
{
    if (!s) return fail(NB_ERR_INVALID, "null handle");
    s->spec_open = false;          // state / parameters change: the speculative next positions are void
    DeviceGuard guard(s->cfg.device);
    return force_eval(s, false);
}

int nb_kick_drift(nb_sim *s)
{
    if (!s) return fail(NB_ERR_INVALID, "null handle");
    s->spec_open = false;          // state / parameters change: the speculative next positions are void
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
    s->spec_open = false;          // state / parameters change: the speculative next positions are void
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
    return step_run(s, nsteps);
}

int nb_energy(nb_sim *s, double *kinetic, double *potential)
{
    if (!s) return fail(NB_ERR_INVALID, "null handle");
    DeviceGuard guard(s->cfg.device);
    return energy_eval(s, kinetic, potential);
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

int nb_quant_bin_sums(nb_sim *s, int32_t which, int64_t *sum_k, int64_t *sum_kw, double info[8])
{
    if (!s || !sum_k || !sum_kw) return fail(NB_ERR_INVALID, "null argument");
    if (which < 0 || which > 2) return fail(NB_ERR_INVALID, "which must be 0 (as the last evaluation), 1 (tiled) or 2 (small-system kernel)");
    if (!s->have_storage || s->is_f64 || !grid_mode(s->cfg.mode))
        return fail(NB_ERR_INVALID, "quant debug is only defined for the grid modes");
    if (!s->have_pos || !s->have_mass) return fail(NB_ERR_INVALID, "positions and masses must be set first");
    DeviceGuard guard(s->cfg.device);
    return bin_sums_eval(s, which, sum_k, sum_kw, info);
}

}  // extern "C"

extern "C" {

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
    const int nchunks = h.enabled ? 1 : 0;      // (the pipelined multi-GPU step of round 2 is gone: always one chunk)
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
    if (chunk_work) { chunk_work[0] = 0; chunk_work[1] = (int32_t)h.work.size(); }
    if (chunk_tile) { chunk_tile[0] = 0; chunk_tile[1] = h.tiles; }
    return NB_OK;
}

// ---- measurement -----------------------------------------------------------------------------
int nb_kernel_time(nb_sim *s, double *total_ms, int32_t *launches)
{
    if (!s) return fail(NB_ERR_INVALID, "null handle");
    DeviceGuard guard(s->cfg.device);
    return prof_collect(s, total_ms, launches);
}

const char *nb_force_kernel_name(nb_sim *s) { return s ? s->last_kernel : "none"; }

int nb_synchronize(nb_sim *s)
{
    if (!s) return fail(NB_ERR_INVALID, "null handle");
    DeviceGuard guard(s->cfg.device);
    HIPCHK(hipStreamSynchronize(s->stream));
    return p2p_check(s);
}

}  // extern "C"
