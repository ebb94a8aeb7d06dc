// nb_comm.cpp -- the process communicator behind the C-ABI's nb_comm_* entry points (include/nbody_amd.h):
// RCCL (librccl resolved with dlopen, so single-GPU use never loads it) and the direct xGMI all-reduce of
// nb_p2p.hip.  One process drives one GPU; the communicator belongs to the process, simulation handles borrow it
// (nb_state.h).  The per-step exchange is SURVEY.md section 8(e)'s: ONE all-reduce (sum) of the (N, D) force vectors.
#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "nb_state.h"

namespace nbhost {

Rccl g_rccl;
ProcComm g_pc;
std::mutex g_pc_mu;
hipStream_t g_p2p_last_stream = nullptr;
std::mutex g_p2p_mu;

namespace {
char g_direct_sentinel;
unsigned g_generation = 0;       // communicators created by this process so far
}  // namespace

int load_rccl()
{
    if (g_rccl.lib) return NB_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(NB_ERR_COMM, "cannot load librccl: %s", dlerror());
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(h, "ncclCommCount");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy)
        return fail(NB_ERR_COMM, "librccl is missing expected symbols");
    g_rccl.lib = h;
    return NB_OK;
}

double p2p_step_timeout_s()
{
    static const double t = [] {
        const char *e = getenv("NB_P2P_TIMEOUT_S");
        const double v = e ? atof(e) : 0.0;
        return v > 0.0 ? v : 60.0;
    }();
    return t;
}

int comm_check(const nb_sim *s)
{
    if (s->comm && s->comm_generation != g_pc.generation)
        return fail(NB_ERR_COMM, "the process communicator this handle attached to was shut down (nb_comm_shutdown); "
                                 "call nb_comm_init again or create a new handle");
    return NB_OK;
}

// The direct xGMI all-reduce (nb_p2p.hip) serves this handle's force vector when every rank enabled it after the
// collective self-test, the process communicator is the one it was built for, and the vector fits its buffers.
// The decision depends only on values that are equal on all ranks.
bool p2p_use(const nb_sim *s, int64_t cnt)
{
    if (s->knobs.no_p2p || !s->comm || nb_p2p_state() != 2) return false;
    if (nb_p2p_nranks() != g_pc.nranks || nb_p2p_device() != s->cfg.device) return false;
    if (!s->is_f64 && (cnt & 1)) return false;                 // the kernel moves 8-byte units
    return (size_t)cnt * (s->is_f64 ? 8 : 4) <= nb_p2p_capacity();
}
bool p2p_use_x64(const nb_sim *s, int64_t cnt)
{
    return s->comm && !s->knobs.no_p2p && nb_p2p_state() == 2 && nb_p2p_nranks() == g_pc.nranks &&
           nb_p2p_device() == s->cfg.device && (size_t)cnt * 8 <= nb_p2p_capacity();
}
int p2p_claim_buffer(nb_sim *s)
{
    std::lock_guard<std::mutex> lock(g_p2p_mu);
    if (g_p2p_last_stream && g_p2p_last_stream != s->stream) HIPCHK(hipStreamSynchronize(g_p2p_last_stream));
    g_p2p_last_stream = s->stream;
    return NB_OK;
}

// Sum `count` elements of `buf` over the ranks, in place, on the handle's stream: RCCL, or -- on a direct-only
// communicator -- a copy into the shared input buffer and the direct all-reduce.
int comm_allreduce_sum(nb_sim *s, void *buf, size_t count, bool f64)
{
    if (int rc = comm_check(s)) return rc;
    if (!g_pc.direct_only) {
        NCCLCHK(g_rccl.AllReduce(buf, buf, count, f64 ? ncclDouble : ncclFloat, ncclSum, s->comm, s->stream));
        return NB_OK;
    }
    const size_t bytes = count * (f64 ? 8 : 4);
    if (nb_p2p_state() != 2 || bytes > nb_p2p_capacity() || (!f64 && (count & 1)))
        return fail(NB_ERR_COMM, "direct-only communicator: %zu %s elements do not fit the direct all-reduce (capacity %zu "
                                 "bytes, fp32 counts even); use an RCCL communicator", count, f64 ? "fp64" : "fp32",
                    nb_p2p_capacity());
    if (int rc = p2p_claim_buffer(s)) return rc;
    HIPCHK(hipMemcpyAsync(nb_p2p_data(), buf, bytes, hipMemcpyDeviceToDevice, s->stream));
    HIPCHK(nb_p2p_allreduce(buf, count, f64, p2p_step_timeout_s(), s->stream));
    s->used_p2p = true;
    return NB_OK;
}
int comm_allreduce_max_u32(nb_sim *s, unsigned int *buf)
{
    if (int rc = comm_check(s)) return rc;
    if (g_pc.direct_only) return fail(NB_ERR_COMM, "direct-only communicator: no max all-reduce (every rank scans all pairs)");
    NCCLCHK(g_rccl.AllReduce(buf, buf, 1, ncclUint32, ncclMax, s->comm, s->stream));
    return NB_OK;
}

int p2p_check(nb_sim *s)
{
    if (!s->used_p2p) return NB_OK;
    int st = 0;
    HIPCHK(nb_p2p_status(&st));
    if (st)
        return fail(NB_ERR_COMM, "direct xGMI all-reduce: a peer did not arrive within %.0f s (results invalid)",
                    p2p_step_timeout_s());
    return NB_OK;
}

}  // namespace nbhost

using namespace nbhost;

extern "C" {

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
        g_pc.generation = ++g_generation;
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
        g_pc.generation = ++g_generation;
    }
    if (g_pc.nranks != want_n || g_pc.rank != want_r || g_pc.device != s->cfg.device)
        return fail(NB_ERR_COMM, "the process communicator is rank %d of %d on device %d; this handle wants rank %d of %d "
                                 "on device %d (one process drives one GPU)",
                    g_pc.rank, g_pc.nranks, g_pc.device, want_r, want_n, s->cfg.device);
    s->comm = g_pc.comm;
    s->comm_generation = g_pc.generation;
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
        if (which == 1) {       // a timed-out barrier makes the figure meaningless: report it, do not time garbage
            int status = 0;
            HIPCHK(nb_p2p_status(&status));
            if (status) return fail(NB_ERR_COMM, "direct all-reduce timing: a peer did not arrive within 30 s");
        }
        *us_per_call = 1e3 * ms / iters;
        return NB_OK;
    }
    if (!s->comm) return fail(NB_ERR_COMM, "the handle has no communicator");
    if (int rc = comm_check(s)) return rc;
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
            if (which == 1) e = nb_p2p_allreduce(buf, (size_t)cnt, s->is_f64, p2p_step_timeout_s(), s->stream);
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
    if (which == 1) {
        s->used_p2p = true;
        if (int rc = p2p_check(s)) return rc;
    }
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

// What the process communicator is, for the host language's records (bench.py prints it for every multi-GPU run):
// info = {ranks, rank, device, direct_only, ncclCommCount of the RCCL communicator (-1: none), state of the direct
// all-reduce (0 none, 1 attached, 2 enabled), its rank count, communicator generation}.
int nb_comm_info(int32_t info[8])
{
    if (!info) return fail(NB_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lock(g_pc_mu);
    int count = -1;
    if (g_pc.comm && !g_pc.direct_only && g_rccl.CommCount) {
        int c = 0;
        if (g_rccl.CommCount(g_pc.comm, &c) == ncclSuccess) count = c;
    }
    const int32_t v[8] = {g_pc.comm ? g_pc.nranks : 0, g_pc.rank, g_pc.device, g_pc.direct_only ? 1 : 0, count,
                          nb_p2p_state(), nb_p2p_nranks(), (int32_t)g_pc.generation};
    memcpy(info, v, sizeof v);
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

}  // extern "C"
