// nb_p2p.hip -- latency-bound all-reduce of the force vectors by direct loads over xGMI.
//
// At the benchmark size the per-step collective is 1 MiB (N = 65 536, D = 2, fp64) between 8 GPUs: far below the
// size where link bandwidth matters, so its cost is protocol latency.  xGMI is a full point-to-point mesh and every
// GPU can load from every peer's HBM, which allows the shortest possible schedule -- one kernel, two hops:
//
//   barrier A   every workgroup b tells workgroup b of every peer "my input is complete" (one 8-byte flag per peer)
//   stage 1     rank r owns slice r of the vector: it loads that slice from all P ranks' input buffers and adds them
//               in rank order 0..P-1 (every element is summed by exactly one rank, in one fixed order: all ranks end
//               with bit-identical results, run after run), keeps the result and publishes it in its `out` buffer
//   barrier B   workgroup b of every peer has published portion b of its slice
//   stage 2     every rank loads the P-1 other slices from their owners
//
// Barriers are per workgroup (block b only ever touches portion b of every slice), so there is no grid-wide sync, and
// flags carry a monotonically increasing epoch, so nothing is ever reset.  Every buffer a peer touches is one
// allocation per process, shared through HIP IPC; peer data is read and published with system-scope (cache-bypassing)
// loads and write-through stores.  A workgroup that waits longer than the timeout raises a status word and leaves (results are
// then invalid and the caller reports an error) -- the grid always drains.
//
// This is an accelerator for small vectors only: the RCCL communicator stays, serves every other collective (scalars,
// large vectors) and is the fallback when the self-test of this path (nb_api.cpp) does not pass on every rank.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "nb_internal.h"

namespace {

// Ordering between a rank's data stores / loads and the barrier flags.  Default: the relaxed form -- system-scope
// write-through stores and cache-bypassing loads, ordered by s_waitcnt vmcnt(0) and the workgroup barrier only.  It is
// OUTSIDE the HIP memory model and has never run across real xGMI links, which is why the whole path is opt-in
// (NB_P2P=auto | force) and guarded by the collective self-test and bench.py's direct-vs-RCCL comparison.  The model's
// own recipe -- system-scope release before signalling, acquire after waiting -- is the build knob
// -DP2P_FENCE=1 -DP2P_ACQ=1; measured on one GPU between two processes (round 3): 52 us instead of 6.7 us per 1 MiB
// all-reduce (whole-L2 write-back / invalidate per barrier), i.e. slower than RCCL, so it cannot be the default of a path
// whose only purpose is to beat RCCL's latency.
#ifndef P2P_FENCE
#define P2P_FENCE 0
#endif
#ifndef P2P_ACQ
#define P2P_ACQ 0
#endif
constexpr int P2P_MAX_RANKS = 8;
constexpr int P2P_MAX_BLOCKS = 256;       // workgroups per all-reduce: NB_P2P_BLOCKS (same on every rank), default 256 --
                                          // every rank touches all N*D elements, and the work is pure latency
constexpr int P2P_THREADS = 256;
constexpr int P2P_BATCH = 2;              // units a thread has in flight per peer
constexpr size_t P2P_SIG_BYTES = 40960;   // 2 phases x 256 blocks x 8 ranks x 8 B = 32 KiB of flags, then the status word
constexpr size_t P2P_STATUS_OFF = 32768;
typedef unsigned long long u64;

struct P2PArgs {
    char *base[P2P_MAX_RANKS];
    int rank, nranks;
    u64 epoch;
    long long units;            // 8-byte units (one double / two floats)
    long long timeout_ticks;    // of wall_clock64()
    size_t data_off, out_off;
    // leapfrog work fused behind the sum (what reduce_sym_kernel fuses on one GPU): 0 none, 1 closing half kick,
    // 2 closing kick + the next step's opening kick + drift (+ repack of the pair-symmetric kernel's positions)
    int kick, dim, np;
    void *vel, *pos, *packed;
    double half_dt, dt;
    int f64_to_f32;             // the units are fp64 sums; dst / vel / pos / packed are fp32: result = (float)(sum * scale)
    double scale;
};

struct P2PState {
    bool attached = false, enabled = false;
    int device = -1, rank = 0, nranks = 0;
    size_t cap = 0;             // bytes of the data (and of the out) buffer
    char *local = nullptr;
    char *base[P2P_MAX_RANKS] = {};
    u64 epoch = 0;
    double ticks_per_s = 1e8;
    int blocks = P2P_MAX_BLOCKS;
};
P2PState g;

int p2p_env_blocks()
{
    const char *v = getenv("NB_P2P_BLOCKS");
    const int b = v ? atoi(v) : P2P_MAX_BLOCKS;
    return b < 1 ? 1 : (b > P2P_MAX_BLOCKS ? P2P_MAX_BLOCKS : b);
}

// The shared region is ORDINARY device memory (hipMalloc).  Measured on MI355X / ROCm 7.2 with the virtual-node test
// below: regions from hipExtMallocWithFlags(hipDeviceMallocUncached) returned stale data to the very next kernel
// (thousands of wrong elements per all-reduce at 4 MiB, with or without a device synchronisation in between, with
// system-scope or plain stores by the producer), ordinary allocations never did.  Every access a peer can observe is
// a system-scope (sc0 sc1) load or write-through store, so no cache of either side holds these lines across the
// barriers; the producer kernel's plain stores are written back when that kernel ends.
hipError_t p2p_region_alloc(void **p, size_t bytes)
{
#ifdef P2P_UNCACHED
    return hipExtMallocWithFlags(p, bytes, hipDeviceMallocUncached);
#else
    return hipMalloc(p, bytes);
#endif
}

__device__ __forceinline__ u64 load_sys(const void *p)
{
#ifdef P2P_PLAIN_LOAD
    return *(const volatile u64 *)p;
#else
    return __hip_atomic_load((const u64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}
__device__ __forceinline__ void store_sys(void *p, u64 v)
{
    __hip_atomic_store((u64 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Shared buffers are only touched with system-scope (sc0 sc1) accesses inside this kernel, so flags need no acquire /
// release cache maintenance (a system-scope release writes back the whole L2, an acquire invalidates it: ~10 us per
// all-reduce when measured); ordering comes from program order within a wave, vmcnt waits and the workgroup barrier.
__device__ __forceinline__ void p2p_barrier(const P2PArgs &a, int phase)
{
    __syncthreads();
    const int t = threadIdx.x;
    if (t < a.nranks) {
        const size_t slot = ((size_t)phase * P2P_MAX_BLOCKS + blockIdx.x) * P2P_MAX_RANKS;
        u64 *mine_at_peer = (u64 *)a.base[t] + slot + a.rank;
        __hip_atomic_store(mine_at_peer, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const u64 *peer_at_mine = (const u64 *)a.base[a.rank] + slot + t;
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(peer_at_mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < a.epoch) {
            if (wall_clock64() - t0 > a.timeout_ticks) {
                atomicExch((int *)(a.base[a.rank] + P2P_STATUS_OFF), 1);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
}

template <bool F64> __device__ __forceinline__ u64 add_units(u64 x, u64 y)
{
    if (F64) return (u64)__double_as_longlong(__longlong_as_double((long long)x) + __longlong_as_double((long long)y));
    const float lo = __uint_as_float((unsigned)x) + __uint_as_float((unsigned)y);
    const float hi = __uint_as_float((unsigned)(x >> 32)) + __uint_as_float((unsigned)(y >> 32));
    return (u64)__float_as_uint(lo) | ((u64)__float_as_uint(hi) << 32);
}

// One summed element (index e of the (N, D) force vector, value f): store it and apply the kicks / drift in the
// storage type with torch's separate multiply and add roundings (simulation.py:132-141), exactly as the single-GPU
// reduction does.
template <typename T>
__device__ __forceinline__ void p2p_leapfrog(const P2PArgs &a, long long e, T f)
{
    T *vel = (T *)a.vel;
    T v = vel[e];
    const T h = (T)a.half_dt;
    if (sizeof(T) == 8) v = (T)__dadd_rn((double)v, __dmul_rn((double)f, (double)h));
    else v = (T)__fadd_rn((float)v, __fmul_rn((float)f, (float)h));
    if (a.kick == 2) {
        T *pos = (T *)a.pos;
        T x;
        if (sizeof(T) == 8) {
            v = (T)__dadd_rn((double)v, __dmul_rn((double)f, (double)h));
            x = (T)__dadd_rn((double)pos[e], __dmul_rn((double)v, a.dt));
        } else {
            v = (T)__fadd_rn((float)v, __fmul_rn((float)f, (float)h));
            x = (T)__fadd_rn((float)pos[e], __fmul_rn((float)v, (float)a.dt));
        }
        pos[e] = x;
        if (a.packed) ((T *)a.packed)[(size_t)(e % a.dim) * a.np + e / a.dim] = x;
    }
    vel[e] = v;
}

template <bool F64>
__device__ __forceinline__ void p2p_finish(const P2PArgs &a, u64 *__restrict__ dst, long long u, u64 s)
{
    if (F64 && a.f64_to_f32) {
        const float f = (float)(__longlong_as_double((long long)s) * a.scale);
        ((float *)dst)[u] = f;
        if (a.kick) p2p_leapfrog<float>(a, u, f);
        return;
    }
    dst[u] = s;
    if (a.kick) {
        if (F64) {
            p2p_leapfrog<double>(a, u, __longlong_as_double((long long)s));
        } else {
            p2p_leapfrog<float>(a, 2 * u, __uint_as_float((unsigned)s));
            p2p_leapfrog<float>(a, 2 * u + 1, __uint_as_float((unsigned)(s >> 32)));
        }
    }
}

template <bool F64>
__device__ __forceinline__ void p2p_body(const P2PArgs &a, u64 *__restrict__ dst)
{
    const int P = a.nranks, t = threadIdx.x;
    const long long S = (a.units + P - 1) / P;                          // slice of a rank
    const long long per = (S + gridDim.x - 1) / gridDim.x;              // portion of a workgroup
    p2p_barrier(a, 0);
    {
        const long long lo = (long long)a.rank * S + (long long)blockIdx.x * per;
        long long hi = lo + per;
        if (hi > (long long)(a.rank + 1) * S) hi = (long long)(a.rank + 1) * S;
        if (hi > a.units) hi = a.units;
        // P2P_BATCH units per thread at a time: all their loads are in flight together (one memory round trip per
        // batch, not per unit -- the loads are system-scope, the compiler keeps them in order but does not wait)
        for (long long u0 = lo + t; u0 < hi; u0 += (long long)P2P_THREADS * P2P_BATCH) {
            u64 v[P2P_BATCH][P2P_MAX_RANKS];
#pragma unroll
            for (int k = 0; k < P2P_BATCH; ++k) {
                const long long u = u0 + (long long)k * P2P_THREADS;
#pragma unroll
                for (int q = 0; q < P2P_MAX_RANKS; ++q)
                    if (q < P && u < hi) v[k][q] = load_sys(a.base[q] + a.data_off + (size_t)u * 8);
            }
#pragma unroll
            for (int k = 0; k < P2P_BATCH; ++k) {
                const long long u = u0 + (long long)k * P2P_THREADS;
                if (u >= hi) continue;
                u64 s = v[k][0];
#pragma unroll
                for (int q = 1; q < P2P_MAX_RANKS; ++q)
                    if (q < P) s = add_units<F64>(s, v[k][q]);
                store_sys(a.base[a.rank] + a.out_off + (size_t)u * 8, s);
                p2p_finish<F64>(a, dst, u, s);
            }
        }
    }
    // every store above is a write-through store to the shared region (or to `dst`, which no peer reads): once they are
    // acknowledged (vmcnt = 0) the data is where the peers load it from -- no cache write-back is needed, and the
    // flags can be relaxed (every wave waits here, then the barrier's __syncthreads, then the signalling threads' stores)
#if P2P_FENCE == 1
    __threadfence_system();
#elif P2P_FENCE == 2
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    p2p_barrier(a, 1);
#if P2P_ACQ == 1
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
#elif P2P_ACQ == 2
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    for (long long k0 = t; k0 < per; k0 += (long long)P2P_THREADS * P2P_BATCH) {
        u64 v[P2P_BATCH][P2P_MAX_RANKS];
        long long at[P2P_BATCH][P2P_MAX_RANKS];
#pragma unroll
        for (int k = 0; k < P2P_BATCH; ++k) {
            const long long kk = k0 + (long long)k * P2P_THREADS;
#pragma unroll
            for (int q = 0; q < P2P_MAX_RANKS; ++q) {
                at[k][q] = -1;
                if (q < P && q != a.rank && kk < per) {
                    const long long u = (long long)q * S + (long long)blockIdx.x * per + kk;
                    if (u < (long long)(q + 1) * S && u < a.units) {
                        at[k][q] = u;
                        v[k][q] = load_sys(a.base[q] + a.out_off + (size_t)u * 8);
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < P2P_BATCH; ++k)
#pragma unroll
            for (int q = 0; q < P2P_MAX_RANKS; ++q)
                if (at[k][q] >= 0) p2p_finish<F64>(a, dst, at[k][q], v[k][q]);
    }
}

template <bool F64>
__global__ void __launch_bounds__(P2P_THREADS)
p2p_allreduce_kernel(P2PArgs a, u64 *__restrict__ dst)
{
    p2p_body<F64>(a, dst);
}

// tests: all ranks of a virtual node in ONE dispatch (blockIdx.y = rank), so that they are co-resident by construction
struct P2PDsts { u64 *p[P2P_MAX_RANKS]; };
template <bool F64>
__global__ void __launch_bounds__(P2P_THREADS)
p2p_allreduce_node_kernel(P2PArgs a, P2PDsts d)
{
    a.rank = blockIdx.y;
    p2p_body<F64>(a, d.p[blockIdx.y]);
}

// self-test helpers: integer-valued patterns (sums are exact in any order)
#ifdef P2P_FILL_SYS
#define P2P_FILL_STORE(p, v) store_sys((p), (v))
#else
#define P2P_FILL_STORE(p, v) (*(p) = (v))
#endif
template <bool F64>
__global__ void p2p_fill_kernel(u64 *data, long long units, int rank, int round)
{
    const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= units) return;
    if (F64) {
        P2P_FILL_STORE(data + u, (u64)__double_as_longlong((double)((rank + 1) * (int)((u + round) % 1021))));
    } else {
        const float lo = (float)((rank + 1) * (int)((2 * u + round) % 1021));
        const float hi = (float)((rank + 1) * (int)((2 * u + 1 + round) % 1021));
        P2P_FILL_STORE(data + u, (u64)__float_as_uint(lo) | ((u64)__float_as_uint(hi) << 32));
    }
}
template <bool F64>
__global__ void p2p_check_kernel(const u64 *res, long long units, int nranks, int round, int *bad, int *dbg = nullptr,
                                 unsigned long long *dbg_val = nullptr)
{
    const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= units) return;
    const int tri = nranks * (nranks + 1) / 2;
    bool ok;
    if (F64) {
        ok = __longlong_as_double((long long)res[u]) == (double)(tri * (int)((u + round) % 1021));
    } else {
        ok = __uint_as_float((unsigned)res[u]) == (float)(tri * (int)((2 * u + round) % 1021)) &&
             __uint_as_float((unsigned)(res[u] >> 32)) == (float)(tri * (int)((2 * u + 1 + round) % 1021));
    }
    if (!ok) {
        atomicAdd(bad, 1);
        if (dbg) {
            atomicAdd(dbg, 1); atomicMin(dbg + 1, (int)u); atomicMax(dbg + 2, (int)u);
            // the largest bad index of the rank also leaves its value (integer-valued patterns): packed as u * 2^20 + value
            if (F64) atomicMax((unsigned long long *)(dbg_val), ((unsigned long long)u << 24) | (unsigned long long)(long long)__longlong_as_double((long long)res[u]));
        }
    }
}

}  // namespace

size_t nb_p2p_handle_bytes() { return sizeof(hipIpcMemHandle_t); }

static void p2p_set_kick(P2PArgs &a, const NbP2PKick *k)
{
    a.kick = 0; a.dim = 1; a.np = 0; a.vel = a.pos = a.packed = nullptr; a.half_dt = a.dt = 0.0;
    a.f64_to_f32 = 0; a.scale = 1.0;
    if (k) {
        a.f64_to_f32 = k->f64_to_f32; a.scale = k->scale;
        if (k->mode) {
            a.kick = k->mode; a.dim = k->dim; a.np = k->np; a.vel = k->vel; a.pos = k->pos; a.packed = k->packed;
            a.half_dt = k->half_dt; a.dt = k->dt;
        }
    }
}

// Allocate this process's shared region and describe it for the peers.
hipError_t nb_p2p_export(int device, int rank, int nranks, size_t cap_bytes, void *handle_out)
{
    if (g.local) return hipErrorAlreadyMapped;
    if (nranks < 1 || nranks > P2P_MAX_RANKS || rank < 0 || rank >= nranks) return hipErrorInvalidValue;
    cap_bytes = (cap_bytes + 255) & ~(size_t)255;
    void *p = nullptr;
    hipError_t e = p2p_region_alloc(&p, P2P_SIG_BYTES + 2 * cap_bytes);
    if (e != hipSuccess) return e;
    e = hipMemset(p, 0, P2P_SIG_BYTES);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) { (void)hipFree(p); return e; }
    memcpy(handle_out, &h, sizeof h);
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0)
        g.ticks_per_s = 1e3 * khz;
    g.local = (char *)p;
    g.device = device; g.rank = rank; g.nranks = nranks; g.cap = cap_bytes;
    g.epoch = 0;
    g.blocks = p2p_env_blocks();
    return hipSuccess;
}

// Map every peer's region (handles: nranks consecutive hipIpcMemHandle_t in rank order).
hipError_t nb_p2p_import(const void *handles)
{
    if (!g.local || g.attached) return hipErrorInvalidValue;
    for (int q = 0; q < g.nranks; ++q) {
        if (q == g.rank) { g.base[q] = g.local; continue; }
        hipIpcMemHandle_t h;
        memcpy(&h, (const char *)handles + (size_t)q * sizeof h, sizeof h);
        void *p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            for (int k = 0; k < q; ++k)
                if (k != g.rank && g.base[k]) { (void)hipIpcCloseMemHandle(g.base[k]); g.base[k] = nullptr; }
            return e;
        }
        g.base[q] = (char *)p;
    }
    g.attached = true;
    return hipSuccess;
}

int nb_p2p_state() { return g.enabled ? 2 : (g.attached ? 1 : 0); }
void nb_p2p_enable(bool on) { g.enabled = on && g.attached; }
size_t nb_p2p_capacity() { return g.attached ? g.cap : 0; }
void *nb_p2p_data() { return g.local ? g.local + P2P_SIG_BYTES : nullptr; }
int nb_p2p_nranks() { return g.attached ? g.nranks : 0; }
int nb_p2p_device() { return g.device; }

// dst <- sum over ranks of their data buffers (count elements of double / float).  Collective: every rank issues the
// same sequence of calls.  `dst` is ordinary device memory of this rank.
hipError_t nb_p2p_allreduce(void *dst, size_t count, int is_f64, double timeout_s, hipStream_t st, const NbP2PKick *kick)
{
    if (!g.attached) return hipErrorNotInitialized;
    const size_t bytes = count * (is_f64 ? 8 : 4);
    if (bytes > g.cap || count == 0) return hipErrorInvalidValue;
    P2PArgs a;
    for (int q = 0; q < P2P_MAX_RANKS; ++q) a.base[q] = q < g.nranks ? g.base[q] : nullptr;
    a.rank = g.rank; a.nranks = g.nranks;
    a.epoch = ++g.epoch;
    a.units = (long long)((bytes + 7) / 8);
    a.timeout_ticks = (long long)(timeout_s * g.ticks_per_s);
    a.data_off = P2P_SIG_BYTES; a.out_off = P2P_SIG_BYTES + g.cap;
    p2p_set_kick(a, kick);
    if (is_f64) hipLaunchKernelGGL(p2p_allreduce_kernel<true>, dim3(g.blocks), dim3(P2P_THREADS), 0, st, a, (u64 *)dst);
    else hipLaunchKernelGGL(p2p_allreduce_kernel<false>, dim3(g.blocks), dim3(P2P_THREADS), 0, st, a, (u64 *)dst);
    return hipGetLastError();
}

// status word of the last kernels (the caller has synchronised the stream): 0 fine, 1 a barrier timed out
hipError_t nb_p2p_status(int *status)
{
    *status = 0;
    if (!g.local) return hipSuccess;
    return hipMemcpy(status, g.local + P2P_STATUS_OFF, sizeof(int), hipMemcpyDeviceToHost);
}

// One round of the self-test: fill the data buffer with a rank-dependent integer pattern, all-reduce `count` elements
// into `scratch`, count the mismatches against the closed form.  Collective.
hipError_t nb_p2p_selftest_round(void *scratch, size_t count, int is_f64, int round, double timeout_s, int *bad_dev,
                                 hipStream_t st)
{
    const long long units = (long long)((count * (is_f64 ? 8 : 4) + 7) / 8);
    const int blocks = (int)((units + 255) / 256);
    if (is_f64) hipLaunchKernelGGL(p2p_fill_kernel<true>, dim3(blocks), dim3(256), 0, st, (u64 *)nb_p2p_data(), units, g.rank, round);
    else hipLaunchKernelGGL(p2p_fill_kernel<false>, dim3(blocks), dim3(256), 0, st, (u64 *)nb_p2p_data(), units, g.rank, round);
    hipError_t e = nb_p2p_allreduce(scratch, count, is_f64, timeout_s, st);
    if (e != hipSuccess) return e;
    // an odd float count leaves half a unit of padding: check whole units only when the count is even
    const long long check_units = is_f64 ? units : (long long)(count / 2);
    if (check_units > 0) {
        const int cb = (int)((check_units + 255) / 256);
        if (is_f64) hipLaunchKernelGGL(p2p_check_kernel<true>, dim3(cb), dim3(256), 0, st, (const u64 *)scratch, check_units, g.nranks, round, bad_dev);
        else hipLaunchKernelGGL(p2p_check_kernel<false>, dim3(cb), dim3(256), 0, st, (const u64 *)scratch, check_units, g.nranks, round, bad_dev);
    }
    return hipGetLastError();
}

// Virtual ranks: `nranks` regions, streams and kernels inside ONE process, so a one-GPU box can execute the kernel
// with the geometry of a full node (8 slices, 8 flags per barrier) -- the multi-process test is limited to the box's
// process quota.  concurrent = 0: flags pre-satisfied, the ranks' kernels run one after the other, twice (the first
// pass publishes every slice, the second gathers them): checks the index arithmetic without relying on co-residency.
// concurrent = 1: one stream per rank, real barriers; needs as many hardware queues as ranks (4 by default).
// concurrent = 2: all ranks in ONE dispatch (blockIdx.y = rank): co-resident by construction, real barriers, any P.
hipError_t nb_p2p_virtual(int nranks, size_t count, int is_f64, int concurrent, int iters, double timeout_s, int *bad_total,
                          double *us_per_call)
{
    if (nranks < 1 || nranks > P2P_MAX_RANKS || count == 0 || iters < 1) return hipErrorInvalidValue;
    const size_t bytes = count * (is_f64 ? 8 : 4), cap = (bytes + 255) & ~(size_t)255;
    if (!is_f64 && (count & 1)) return hipErrorInvalidValue;
    const long long units = (long long)((bytes + 7) / 8);
    char *region[P2P_MAX_RANKS] = {};
    void *dst[P2P_MAX_RANKS] = {};
    hipStream_t st[P2P_MAX_RANKS] = {};
    int *bad = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc((void **)&bad, sizeof(int) * (1 + 4 * P2P_MAX_RANKS));
    if (e == hipSuccess) e = hipMemset(bad, 0, sizeof(int) * (1 + 4 * P2P_MAX_RANKS));
    const bool debug = getenv("NB_P2P_DEBUG") != nullptr;
    unsigned long long *dbg_val = nullptr;
    if (e == hipSuccess) e = hipMalloc((void **)&dbg_val, 8 * P2P_MAX_RANKS);
    if (e == hipSuccess) e = hipMemset(dbg_val, 0, 8 * P2P_MAX_RANKS);
    for (int q = 0; q < nranks && e == hipSuccess; ++q) {
        e = p2p_region_alloc((void **)&region[q], P2P_SIG_BYTES + 2 * cap);
        if (e == hipSuccess) e = hipMemset(region[q], concurrent ? 0 : 0x7f, P2P_SIG_BYTES);
        if (e == hipSuccess) e = hipMemset(region[q] + P2P_STATUS_OFF, 0, sizeof(int));
        if (e == hipSuccess) e = hipMalloc(&dst[q], bytes);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&st[q], hipStreamNonBlocking);
    }
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    double ticks = 1e8;
    {
        int dev = 0, khz = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0)
            ticks = 1e3 * khz;
    }
    u64 epoch = 0;
    // workgroups per rank: the production count, but a whole virtual node in one dispatch must be co-resident
    // (the kernel needs <= 128 VGPRs: 4 workgroups per CU)
    int vblocks = p2p_env_blocks();
    if (concurrent == 2) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            vblocks = std::min(vblocks, std::max(1, 2 * cus / nranks));
    }
    // concurrent = 2: the whole virtual node in one dispatch
    auto launch_node = [&](hipStream_t s) {
        P2PArgs a;
        P2PDsts d;
        for (int k = 0; k < P2P_MAX_RANKS; ++k) { a.base[k] = region[k]; d.p[k] = (u64 *)dst[k]; }
        a.rank = 0; a.nranks = nranks; a.epoch = epoch; a.units = units;
        a.timeout_ticks = (long long)(timeout_s * ticks);
        a.data_off = P2P_SIG_BYTES; a.out_off = P2P_SIG_BYTES + cap;
        p2p_set_kick(a, nullptr);
        if (is_f64) hipLaunchKernelGGL(p2p_allreduce_node_kernel<true>, dim3(vblocks, nranks), dim3(P2P_THREADS), 0, s, a, d);
        else hipLaunchKernelGGL(p2p_allreduce_node_kernel<false>, dim3(vblocks, nranks), dim3(P2P_THREADS), 0, s, a, d);
    };
    auto launch = [&](int q, hipStream_t s) {
        P2PArgs a;
        for (int k = 0; k < P2P_MAX_RANKS; ++k) a.base[k] = region[k];
        a.rank = q; a.nranks = nranks; a.epoch = concurrent ? epoch : 0; a.units = units;      // epoch 0: no barrier ever waits
        a.timeout_ticks = (long long)(timeout_s * ticks);
        a.data_off = P2P_SIG_BYTES; a.out_off = P2P_SIG_BYTES + cap;
        p2p_set_kick(a, nullptr);
        if (is_f64) hipLaunchKernelGGL(p2p_allreduce_kernel<true>, dim3(vblocks), dim3(P2P_THREADS), 0, s, a, (u64 *)dst[q]);
        else hipLaunchKernelGGL(p2p_allreduce_kernel<false>, dim3(vblocks), dim3(P2P_THREADS), 0, s, a, (u64 *)dst[q]);
    };
    const int fb = (int)((units + 255) / 256);
    for (int it = 0; it < iters && e == hipSuccess; ++it) {
        for (int q = 0; q < nranks; ++q) {
            hipStream_t s = (concurrent == 1 || concurrent == 3) ? st[q] : st[0];
            if (is_f64) hipLaunchKernelGGL(p2p_fill_kernel<true>, dim3(fb), dim3(256), 0, s, (u64 *)(region[q] + P2P_SIG_BYTES), units, q, it);
            else hipLaunchKernelGGL(p2p_fill_kernel<false>, dim3(fb), dim3(256), 0, s, (u64 *)(region[q] + P2P_SIG_BYTES), units, q, it);
        }
        ++epoch;
        if (getenv("NB_P2P_SYNC")) (void)hipDeviceSynchronize();
        if (concurrent == 2) {
            launch_node(st[0]);
        } else {
            // concurrent = 3: like 1, but the LAST rank never launches its all-reduce -- a dead peer.  The others must
            // leave their barrier after timeout_s, raise their status word and drain (error-path test).
            const int launched = concurrent == 3 ? nranks - 1 : nranks;
            for (int pass = 0; pass < (concurrent ? 1 : 2); ++pass)
                for (int q = 0; q < launched; ++q) launch(q, concurrent ? st[q] : st[0]);
        }
        for (int q = 0; q < nranks; ++q) {
            hipStream_t s = (concurrent == 1 || concurrent == 3) ? st[q] : st[0];
            int *dbg = debug ? bad + 1 + 4 * q : nullptr;
            if (is_f64) hipLaunchKernelGGL(p2p_check_kernel<true>, dim3(fb), dim3(256), 0, s, (const u64 *)dst[q], units, nranks, it, bad, dbg, dbg_val + q);
            else hipLaunchKernelGGL(p2p_check_kernel<false>, dim3(fb), dim3(256), 0, s, (const u64 *)dst[q], units, nranks, it, bad, dbg, dbg_val + q);
        }
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();       // next round's fill must not overtake a peer's stage 2
        if (debug && e == hipSuccess) {
            int h[1 + 4 * P2P_MAX_RANKS];
            e = hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost);
            unsigned long long hv[P2P_MAX_RANKS];
            if (e == hipSuccess) e = hipMemcpy(hv, dbg_val, sizeof hv, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemset(dbg_val, 0, 8 * P2P_MAX_RANKS);
            for (int q = 0; q < nranks; ++q)
                if (h[1 + 4 * q]) {
                    const long long ub = (long long)(hv[q] >> 24), val = (long long)(hv[q] & 0xffffff);
                    const int tri = nranks * (nranks + 1) / 2;
                    fprintf(stderr, "  p2p debug: P=%d units=%lld iter %d rank %d: %d bad, u in [%d, %d]; at u=%lld got %lld, want %lld "
                                    "(previous round's: %lld; pattern now %lld, before %lld)\n", nranks, units, it, q, h[1 + 4 * q],
                            h[2 + 4 * q], h[3 + 4 * q], ub, val, (long long)tri * ((ub + it) % 1021),
                            (long long)tri * ((ub + it - 1) % 1021), (ub + it) % 1021, (ub + it - 1) % 1021);
                }
            int init[4 * P2P_MAX_RANKS];
            for (int q = 0; q < P2P_MAX_RANKS; ++q) { init[4 * q] = 0; init[4 * q + 1] = 0x7fffffff; init[4 * q + 2] = -1; init[4 * q + 3] = 0; }
            if (e == hipSuccess) e = hipMemcpy(bad + 1, init, sizeof init, hipMemcpyHostToDevice);
        }
    }
    // a barrier that timed out means the kernels were not co-resident: report it, never time it
    bool timed_out = false;
    for (int q = 0; q < nranks && e == hipSuccess; ++q) {
        int status = 0;
        e = hipMemcpy(&status, region[q] + P2P_STATUS_OFF, sizeof(int), hipMemcpyDeviceToHost);
        timed_out = timed_out || status != 0;
    }
    if (e == hipSuccess && concurrent && us_per_call && !timed_out) {
        // timing: all-reduce kernels only, back to back on every stream; stream 0's events
        for (int pass = 0; pass < 2 && e == hipSuccess; ++pass) {
            const int reps = pass == 0 ? 10 : 200;
            e = hipEventRecord(e0, st[0]);
            for (int i = 0; i < reps; ++i) {
                ++epoch;
                if (concurrent == 2) launch_node(st[0]);
                else for (int q = 0; q < nranks; ++q) launch(q, st[q]);
            }
            if (e == hipSuccess) e = hipEventRecord(e1, st[0]);
            if (e == hipSuccess) e = hipDeviceSynchronize();
            int status = 0;
            if (e == hipSuccess) e = hipMemcpy(&status, region[0] + P2P_STATUS_OFF, sizeof(int), hipMemcpyDeviceToHost);
            if (status) break;
        }
        float ms = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        *us_per_call = 1e3 * ms / 200;
    } else if (us_per_call) {
        *us_per_call = 0.0;
    }
    int host_bad = 0;
    if (e == hipSuccess) e = hipMemcpy(&host_bad, bad, sizeof(int), hipMemcpyDeviceToHost);
    for (int q = 0; q < nranks && e == hipSuccess; ++q) {
        int status = 0;
        e = hipMemcpy(&status, region[q] + P2P_STATUS_OFF, sizeof(int), hipMemcpyDeviceToHost);
        if (status) host_bad += 1000000;                          // a barrier timed out
    }
    if (bad_total) *bad_total = host_bad;
    for (int q = 0; q < nranks; ++q) {
        if (st[q]) (void)hipStreamDestroy(st[q]);
        if (dst[q]) (void)hipFree(dst[q]);
        if (region[q]) (void)hipFree(region[q]);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (bad) (void)hipFree(bad);
    if (dbg_val) (void)hipFree(dbg_val);
    return e;
}

void nb_p2p_shutdown()
{
    if (!g.local) return;
    for (int q = 0; q < g.nranks; ++q)
        if (q != g.rank && g.base[q]) (void)hipIpcCloseMemHandle(g.base[q]);
    (void)hipFree(g.local);
    g = P2PState();
}
