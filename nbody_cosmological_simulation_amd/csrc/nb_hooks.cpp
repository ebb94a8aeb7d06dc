// nb_hooks.cpp -- handle-less entry points of the C-ABI on CALLER tensors (include/nbody_amd.h):
//   * the tensor-level precision hooks -- quantization.py module functions quantize_distance_squared / quantize_force /
//     _grid_quantize / _grid_quantize_safe (reference quantization.py:21-157), which the ~15 QuantSim-style subclass
//     overrides of the reference call once per step on an N x N tensor;
//   * the diagnostics of metrics.py:25-156 on caller tensors (nb_metrics_tensors) and on a handle's resident state
//     (nb_metrics).
// Kernels: nb_misc.hip / nb_force.hip (hooks), nb_metrics.hip + nb_sort.hip (diagnostics).
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "nb_state.h"

using namespace nbhost;

namespace {
constexpr int MAX_DEV = 64;
}

int nbhost::check_device(int device)
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

namespace {
// Per-device scratch of the handle-less entry points (tensor-level hooks, nb_metrics_tensors): allocated once and
// grown on demand -- no hipMalloc / hipFree per call.  One call at a time per device (mutex); work is queued on the
// CALLER's stream (nb_set_hook_stream, thread-local) so that it is ordered with the caller's own device work and
// needs no synchronisation for device-resident buffers; a use on a different stream than the previous one first
// waits for that one's event.  Without a hook stream: the NULL stream and a blocking wait, as a plain C caller expects.
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
    if (int rc = run_metrics(s->cfg.device, s->stream, (char *)s->metrics_scratch, s->pos, s->vel, s->mass, s->cfg.n, s->cfg.dim,
                             s->is_f64, arith_f64, num_bins, edges, max_radius, percentile, s->cfg.G, radius_only, curve_mean,
                             curve_count, scalars))
        return rc;
    return p2p_check(s);       // (run_metrics waited for the handle's stream)
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

}  // extern "C"
