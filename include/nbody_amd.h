/*
 * nbody_amd.h -- C-ABI of the MI355X-native direct-summation N-body engine.
 *
 * This is the drop-in boundary for the reference's hot path.  The reference
 * (nuclearbombmods/nbody-cosmological-simulation) has no FFI layer of its own: its operator
 * boundary is the Python class `GalaxySimulation` (simulation.py:12) plus the module-level
 * precision hooks of quantization.py.  Every entry point below names the reference
 * interface it replaces; the Python binding a maintainer would add is a ctypes stub
 * (INTEGRATION.md, and nbody_cosmological_simulation_amd/_native.py in this repo).
 *
 * Conventions
 *   - plain C types only; every function returns 0 (NB_OK) or a negative nb_status and
 *     leaves a thread-local message retrievable with nb_last_error();
 *   - the caller owns every buffer it passes in (host or device, flagged per call); the
 *     library owns the per-handle device state;
 *   - a handle is bound to one HIP device and one stream; distinct handles may be used
 *     concurrently from different threads, one handle must not be;
 *   - NaN/Inf propagate exactly as IEEE arithmetic dictates; nothing traps;
 *   - there is NO CPU fallback: without a HIP device nb_create() fails with NB_ERR_NO_DEVICE.
 */
#ifndef NBODY_AMD_H
#define NBODY_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NB_ABI_VERSION 3

typedef enum nb_status {
    NB_OK = 0,
    NB_ERR_INVALID = -1,       /* bad argument / shape / dtype                     */
    NB_ERR_NO_DEVICE = -2,     /* no HIP device, or device ordinal out of range    */
    NB_ERR_HIP = -3,           /* a HIP runtime call failed (message has details)  */
    NB_ERR_OOM = -4,           /* device allocation failed                         */
    NB_ERR_UNSUPPORTED = -5,   /* combination not implemented yet (message says)   */
    NB_ERR_COMM = -6           /* RCCL failure                                     */
} nb_status;

/* element types of caller buffers and of the Python-visible state tensors */
typedef enum nb_dtype { NB_F16 = 0, NB_BF16 = 1, NB_F32 = 2, NB_F64 = 3 } nb_dtype;

/* quantization.py:10-18 `PrecisionMode` (same order as the enum's declaration) */
typedef enum nb_mode {
    NB_FLOAT64 = 0, NB_FLOAT32 = 1, NB_BFLOAT16 = 2, NB_FLOAT16 = 3,
    NB_INT8_SIM = 4, NB_INT4_SIM = 5, NB_CUSTOM = 6
} nb_mode;

typedef struct nb_sim nb_sim;   /* opaque handle == one GalaxySimulation instance */

/* constructor arguments of GalaxySimulation.__init__ (simulation.py:31-40) */
typedef struct nb_config {
    int32_t n;             /* num_stars                                                     */
    int32_t dim;           /* 2 or 3 (positions are (n, dim) row-major)                     */
    int32_t mode;          /* nb_mode                                                       */
    int32_t levels;        /* CUSTOM grid levels (>= 2; above 4096 on the generic per-pair path); 0 -> 64 (quantization.py:66) */
    double  G;             /* gravitational constant                                        */
    double  softening_sq;  /* softening**2 evaluated in Python double (simulation.py:59)    */
    double  dt;            /* time step                                                     */
    int32_t device;        /* HIP device ordinal                                            */
    int32_t rank;          /* shard index of this process (0 for a single GPU)              */
    int32_t nranks;        /* number of shards = GPUs  (1 for a single GPU)                 */
    int32_t flags;         /* NB_FLAG_*                                                     */
} nb_config;

#define NB_FLAG_PROFILE       1   /* bracket every force launch with HIP events (nb_kernel_time) */
#define NB_FLAG_CUSTOM_FORCEQ 2   /* CUSTOM mode also quantises forces (never set by the stock class) */
#define NB_FLAG_NO_COMM       4   /* nranks > 1 without RCCL: nb_compute_accelerations leaves this rank's
                                     PARTIAL sums (no all-reduce, no force quantisation) for the caller
                                     to reduce; nb_step is refused.  Used by single-GPU shard tests.   */

#define NB_FLAG_F64_STORAGE   16   /* at least one of positions / velocities / masses will be uploaded as fp64 although the
                                     first upload may be fp32: keep the state in fp64 storage from the start (torch
                                     promotes such a simulation to fp64 through the mass product / the first kick) */

#define NB_FLAG_SHARD_TIMING   8   /* timing experiments on ONE GPU: compute only shard `rank` of `nranks`
                                     but run the collectives on a 1-rank communicator.  Results are
                                     partial sums -- never use for physics.                            */

/* ---- lifetime ------------------------------------------------------------------------ */

/* GalaxySimulation.__init__ (simulation.py:31-72) minus the state upload and first force. */
int nb_create(nb_sim **out, const nb_config *cfg);
int nb_destroy(nb_sim *s);

/* Handles keep their stream and (up to 64 MiB) device allocation in a bounded process-level cache when they are
 * destroyed, for the next handle on the same device: the reference's scripts build many short simulations
 * (simulation.py:199-250 run_comparison builds one per mode; main.py:140-176 one per requested mode), and hipFree / hipStreamDestroy wait for the whole device.  nb_cache_trim releases
 * everything the cache holds (bytes of device memory returned through released_bytes, may be NULL). */
int nb_cache_trim(int64_t *released_bytes);

/* attribute writes `sim.G = ..`, `sim.dt = ..`, `sim.softening_sq = ..` between steps
 * (crash_point_test.py, falsification_tests.py read/write them; simulation.py reads them
 * at every use :86,:101,:132-141). */
int nb_set_params(nb_sim *s, double G, double softening_sq, double dt);

/* ---- state --------------------------------------------------------------------------- */

/* simulation.py:63-65 (clone -> device).  pos/vel: n*dim elements, mass: n elements, all of
 * `dtype`; `on_device` != 0 means the pointers are device pointers on the handle's device.
 * Any of pos/vel/mass may be NULL to leave that array untouched (re-upload after an
 * in-place edit such as omega_point_test.py:738). */
int nb_set_state(nb_sim *s, const void *pos, const void *vel, const void *mass,
                 int dtype, int on_device);

/* Writes the Python-visible tensors: each non-NULL destination receives the array in its
 * CURRENT logical dtype (query with nb_state_dtypes).  Synchronises the handle's stream. */
int nb_get_state(nb_sim *s, void *pos, void *vel, void *acc, void *mass, int on_device);

/* dtypes[4] = {positions, velocities, masses, accelerations} as the reference would report
 * them at this moment (SURVEY.md section 8a "Facts": fp32 state is promoted to fp64 by the
 * first step() in FLOAT64 mode, accelerations are fp64 from __init__ on). */
int nb_state_dtypes(nb_sim *s, int32_t dtypes[4]);

/* Replace the stored accelerations (subclasses overriding _compute_accelerations,
 * sensitivity_test.py:61-76: the override's tensor becomes self.accelerations). */
int nb_set_accelerations(nb_sim *s, const void *acc, int dtype, int on_device);

/* ---- the hot path -------------------------------------------------------------------- */

/* GalaxySimulation._compute_accelerations (simulation.py:74-118) on the current positions,
 * including quantize_distance_squared (quantization.py:21-71) and, for INT8/INT4,
 * quantize_force (quantization.py:130-157).  With nranks > 1 the partial sums over this
 * rank's source block are all-reduced (RCCL) before force quantisation. */
int nb_compute_accelerations(nb_sim *s);

/* GalaxySimulation.step (simulation.py:120-143), `nsteps` times, entirely on the device. */
int nb_step(nb_sim *s, int32_t nsteps);

/* The two halves of step() around an externally supplied force (override path):
 * nb_kick_drift: v += a*(dt/2); x += v*dt   (simulation.py:132,135)
 * nb_kick:       v += a*(dt/2)              (simulation.py:141)                          */
int nb_kick_drift(nb_sim *s);
int nb_kick(nb_sim *s);

/* get_kinetic_energy / get_potential_energy (simulation.py:170-192); either may be NULL. */
int nb_energy(nb_sim *s, double *kinetic, double *potential);

/* ---- precision-hook introspection ---------------------------------------------------- */

/* Grid-mode internals of the LAST force evaluation: info[0..3] = lmin, lmax (log-grid of
 * quantization.py:109-113), fmin, fmax (linear force grid, quantization.py:78-79);
 * info[4] = max r^2 over all pairs; info[5] = 1 when the table-free pair path was enabled for that evaluation,
 * info[6] = measured deviation of its bin estimate at the bin edges (in bins), info[7] = measured relative
 * deviation of its force factors from the exact table entries (enabled only when <= 1e-6).  If d2bins != NULL (host, n*n int16, row i / column j)
 * the distance-bin index of every pair is recomputed on the device with the same tables the
 * force kernel used; fbins (host, n*dim int16) likewise for the force bins of INT8/INT4.
 * -1 marks "degenerate grid: value passed through" (quantization.py:115-116 / :81-82). */
int nb_quant_debug(nb_sim *s, double info[8], int16_t *d2bins, int16_t *fbins);
/* The same distance-bin indices for target rows [i0, i1) only (host, (i1-i0)*n int16): sizes where n*n is out of reach. */
int nb_quant_bins_rows(nb_sim *s, int32_t i0, int32_t i1, int16_t *d2bins);
/* Quant-bin assignments (the index round(normalized * (levels - 1)) of quantization.py:119-121) read out of the
 * PRODUCTION pair loop itself.  The two calls above walk the exact threshold tables in a kernel of their own; this one
 * runs the evaluation once more on the current positions with the very kernel templates the force path launches,
 * instantiated with an integer read-out at the point where each pair's bin is decided (table-free estimate, wave
 * ballot, threshold fallback, uniform / general-mass kernels, packed and scalar sweeps, the one-launch small-system
 * kernel), and returns per particle p (host arrays of n int64):
 *     sum_k[p]  = sum over all q of k(p, q)
 *     sum_kw[p] = sum over all q of k(p, q) * ((q mod 65521) + 1)
 * (q = p included: the reference's N x N bin matrix holds k = 0 on the diagonal).  Integer sums: exact, independent
 * of the summation order, comparable bit for bit with the same sums over a row of the reference's bin matrix.
 * which: 0 = the path the last evaluation / step took (nb_force_kernel_name), 1 = the tiled evaluation path of
 * nb_compute_accelerations, 2 = the one-launch small-system step kernel.
 * info (may be NULL): [0] path taken (1 pair-symmetric tiles, 2 one-sided tiles, 3 small-system kernel), [1] targets
 * per lane (path 1) / lanes per target (path 3), [2] 1 if the uniform-mass packed kernel did the work, [3] 1 if the
 * tables enabled the table-free pair path, [4] pair evaluations binned by the table-free estimate alone, [5] pair
 * evaluations binned through a threshold table, [6] grid levels.  Degenerate grids (values pass through, no bins),
 * the generic per-pair path and handles with a communicator return NB_ERR_UNSUPPORTED.  The forces are recomputed
 * (same values); velocities and positions are not touched. */
int nb_quant_bin_sums(nb_sim *s, int32_t which, int64_t *sum_k, int64_t *sum_kw, double info[8]);

/* ---- tensor-level hooks (quantization.py module functions used by override subclasses) -- */

/* quantize_distance_squared(dist_sq, mode, custom_levels, min_dist_sq) quantization.py:21.
 * in/out: `count` elements of `dtype` (NB_F32 or NB_F64); *out_dtype receives the result
 * dtype (FLOAT64 -> f64, FLOAT32/BF16/F16 -> f32, grid modes -> input dtype). */
int nb_quantize_distance_squared(int device, const void *in, void *out, int64_t count, int dtype,
                                 int mode, int levels, double min_dist_sq, int on_device,
                                 int32_t *out_dtype);
/* quantize_force(force, mode, custom_levels) quantization.py:130 */
int nb_quantize_force(int device, const void *in, void *out, int64_t count, int dtype,
                      int mode, int levels, int on_device, int32_t *out_dtype);
/* _grid_quantize(tensor, levels) quantization.py:74 */
int nb_grid_quantize(int device, const void *in, void *out, int64_t count, int dtype,
                     int levels, int on_device);
/* _grid_quantize_safe(tensor, levels, min_val) quantization.py:91 */
int nb_grid_quantize_safe(int device, const void *in, void *out, int64_t count, int dtype,
                          int levels, double min_val, int on_device);

/* Stream of the calling thread for the handle-less entry points on `device` (the four hooks above and
 * nb_metrics_tensors): they queue their kernels there and, for device-resident buffers, return without waiting --
 * ordered with the caller's own device work like any other stream operation.  enable = 0 restores the default
 * (NULL stream, blocking).  Thread-local; no device allocation per call (a per-device scratch is cached). */
int nb_set_hook_stream(int device, void *hip_stream, int enable);

/* ---- diagnostics --------------------------------------------------------------------- */

/* metrics.py:25-156 on the handle's CURRENT state, on the device (stable radix sort + fixed-order scan for the
 * order statistic and the enclosed masses):
 *   compute_rotation_curve  -> curve_mean[num_bins] (NaN for empty bins), curve_count[num_bins]
 *   compute_galaxy_radius   -> scalars[1] = sorted(r)[min(int(n * percentile / 100), n - 1)]
 *   compute_bound_fraction  -> scalars[2]        compute_velocity_dispersion -> scalars[3]
 *   scalars[0] = radii.max(), scalars[4] = the max_radius the bin edges were built from.
 * edges: host, num_bins + 1 float32 (what torch.linspace(0, max_radius, num_bins + 1) returns), or NULL to have
 * them built here from max_radius (< 0: radii.max()).  radius_only != 0: only scalars[0] (first phase of a caller
 * that builds the edges itself).  Any output pointer may be NULL. */
int nb_metrics(nb_sim *s, int32_t num_bins, const float *edges, double max_radius, double percentile,
               int32_t radius_only, double *curve_mean, int64_t *curve_count, double scalars[5]);
/* The same on caller tensors (n x dim positions / velocities, n masses; NB_F32 or NB_F64; host or device). */
int nb_metrics_tensors(int device, const void *pos, const void *vel, const void *mass, int32_t n, int32_t dim,
                       int dtype, int on_device, double G, int32_t num_bins, const float *edges, double max_radius,
                       double percentile, int32_t radius_only, double *curve_mean, int64_t *curve_count,
                       double scalars[5]);

/* ---- multi-GPU (one process per GPU, RCCL over xGMI) --------------------------------- */

/* Partition: every rank holds the full O(N) state.  The pair work is split by
 * TARGET super-rows of the pair-symmetric kernel (snake-dealt, equal pair counts) -- or, on the one-sided
 * kernels, by contiguous SOURCE blocks [rank*N/P, (rank+1)*N/P) -- each rank produces partial accelerations
 * for all particles, and one all-reduce (sum) of the (n, dim) force vectors per step -- RCCL, or the direct path
 * declared below -- completes them
 * (issued in 1..4 prefix slices that overlap the remaining pair work when a step is long enough).
 *
 * ONE communicator per process, shared by every handle of the process:
 *   rank 0 obtains an id (ncclGetUniqueId) and ships the bytes to the other ranks by any means
 *   (torch.distributed store / gloo broadcast); every rank then calls nb_comm_init on its first handle
 *   (collective).  Later handles attach with nb_comm_init(h, NULL, 0) -- no communication.
 *   nb_destroy() never touches the communicator; nb_comm_shutdown() destroys it and is collective:
 *   every rank calls it once, after its last step and before the transport that carried the id goes away. */
int nb_comm_unique_id(void *id_out, int32_t *id_bytes /* in: capacity, out: size */);
int nb_comm_init(nb_sim *s, const void *id /* NULL: attach the existing process communicator */, int32_t id_bytes);
int nb_comm_ready(void);      /* ranks of the process communicator, 0 when there is none */
/* What the process communicator is (for the caller's records): info = {ranks, rank, device, 1 if direct-only (no RCCL),
 * ncclCommCount() of the RCCL communicator (-1: none), state of the direct all-reduce (0 none, 1 attached, 2 enabled),
 * its rank count, communicator generation (handles attached to an older generation fail with NB_ERR_COMM)}. */
int nb_comm_info(int32_t info[8]);
int nb_comm_quiesce(void);    /* first half of a shutdown: wait for this process's device work (then: a barrier of the
                               * host transport, so that no rank frees buffers a peer still reads; then nb_comm_shutdown) */
int nb_comm_shutdown(void);

/* Direct xGMI all-reduce of small force vectors (csrc/nb_p2p.hip).  The per-step collective at
 * the benchmark size is 1 MiB between 8 GPUs -- pure latency -- so, next to the RCCL communicator, the ranks of ONE
 * node can attach a two-hop all-reduce that loads directly from the peers' memory over xGMI (one kernel, rank-ordered
 * sums: bit-identical on all ranks).  Setup, driven by the host language over its own transport:
 *   every rank: nb_comm_p2p_export (allocates the shared region; returns its HIP IPC handle)
 *   all-gather the handles in rank order; every rank: nb_comm_p2p_import(handles, nranks)
 *   barrier; every rank: nb_comm_p2p_selftest (collective; bounded waits); combine the verdicts (logical AND);
 *   every rank: nb_comm_p2p_enable(verdict).
 * A handle may also attach WITHOUT any RCCL communicator: when no process communicator exists yet, the direct path is
 * enabled between exactly the handle's ranks and nb_comm_init gets a NULL id, the process communicator becomes
 * "direct only" -- every sum (forces, potential energy) takes the direct path, vectors above capacity_bytes are an
 * error.  (Several ranks may then share one GPU, which RCCL refuses: how the multi-rank step is tested on one GPU.)
 * Once enabled, nb_step / nb_compute_accelerations use it for force vectors of at most capacity_bytes (fp32: even
 * element counts), and RCCL for everything else; NB_NO_P2P=1 (read at nb_create) keeps a handle on RCCL.  A peer that
 * does not arrive within 300 s raises an error at the next nb_synchronize.  nb_comm_shutdown releases the region. */
int nb_comm_p2p_export(int32_t device, int32_t rank, int32_t nranks, int64_t capacity_bytes, void *handle_out,
                       int32_t *handle_bytes /* in: capacity (>= 64), out: size */);
int nb_comm_p2p_import(const void *handles /* nranks handles, rank order */, int32_t nranks);
int nb_comm_p2p_selftest(int32_t rounds, double timeout_s);
int nb_comm_p2p_enable(int32_t on);
int nb_comm_p2p_state(void);  /* 0 none, 1 attached, 2 enabled */
/* measurement (collective): average microseconds of `iters` back-to-back all-reduces of this handle's force-vector
 * size on zeroed scratch -- which = 0: RCCL, 1: the direct path -- so a multi-GPU bench can state what its collective
 * costs on the node it ran on.  s = NULL: a 1 MiB vector of doubles on the process communicator / the attached direct
 * path (what the host language compares before it enables the direct path) */
int nb_comm_allreduce_time(nb_sim *s, int32_t which, int32_t iters, double *us_per_call);
/* tests: the direct all-reduce kernel between `nranks` VIRTUAL ranks inside this process (a one-GPU box can then run
 * the 8-rank geometry): concurrent = 0 runs the ranks' kernels one after the other with pre-satisfied flags, twice;
 * concurrent = 1 uses one stream per rank and the real barriers (as many hardware queues as ranks: up to 4),
 * concurrent = 2 runs all ranks in one dispatch (co-resident by construction, any rank count); both report the time
 * per all-reduce.  concurrent = 3: like 1, but the last rank never launches its all-reduce (a dead peer): the others
 * must leave their barriers after timeout_s, raise their status word and drain -- the call returns NB_OK with
 * *bad >= 1e6.  *bad = elements that differ from the closed form (+1e6 per rank whose barrier timed out). */
int nb_comm_p2p_virtual_test(int32_t device, int32_t nranks, int64_t count, int32_t dtype, int32_t concurrent,
                             int32_t iters, double timeout_s, int32_t *bad, double *us_per_call);
/* tests: all-reduce `count` host elements (NB_F32 / NB_F64) in place through the direct path (collective) */
int nb_comm_p2p_allreduce(void *host_inout, int64_t count, int32_t dtype, double timeout_s);

/* The work plan of the pair-symmetric kernels for one rank, computed WITHOUT a device (pure host code; what
 * nb_set_state uploads).  For tests of the partition: the union over ranks must cover every tile pair once.
 * info[0..11] = enabled, targets per lane R, tile size, padded tiles, padded particles, work items, row slots,
 * column-slab entries, source tiles per item, pipeline chunks, column-slab MiB, row-slab MiB.
 * work: items x 8 int32 {tile_i, jt_begin, jt_end, slot, slot_stride, col_ord, s_begin, s_count};
 * row_slot0 / row_nslots / col_upto: one int32 per padded tile; chunk_work / chunk_tile: chunks + 1 offsets
 * (work-item ranges / tile boundaries; chunk_tile is the same on every rank).  Any output may be NULL.
 * multi != 0: a communicator will sum the partial forces (enables pipeline chunks); cus: compute units (256). */
int nb_plan_debug(const nb_config *cfg, int32_t is_f64, int32_t multi, int32_t cus, int32_t info[16], int32_t *work,
                  int64_t work_capacity, int32_t *row_slot0, int32_t *row_nslots, int32_t *col_upto,
                  int32_t *chunk_work, int32_t *chunk_tile);

/* ---- measurement --------------------------------------------------------------------- */

/* With NB_FLAG_PROFILE: HIP-event time of the force kernel launches since the last call
 * (events recorded on the handle's own stream).  total_ms / launches = average duration. */
int nb_kernel_time(nb_sim *s, double *total_ms, int32_t *launches);
/* Name of the force kernel the last nb_compute_accelerations / nb_step launched (for matching
 * rocprofv3 rows by prefix): "force_sym_kernel<double", "force_sym_kernel<float", "force_f64_kernel",
 * "force_f32_kernel" or "none". */
const char *nb_force_kernel_name(nb_sim *s);
/* Block until all work queued on the handle's stream has finished. */
int nb_synchronize(nb_sim *s);

/* ---- misc ---------------------------------------------------------------------------- */
int nb_device_count(int32_t *count);
int nb_abi_version(void);
const char *nb_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_AMD_H */
