// nb_metrics.hip -- galaxy diagnostics on the device (reference metrics.py:25-156).
//
//   rotation curve  mean |x v_y - y v_x| / max(r, 0.1) in radial bins [edge_b, edge_b+1)      metrics.py:25-78
//   r_percentile    sorted(r)[min(int(N p / 100), N - 1)]                                      metrics.py:81-95
//   bound fraction  |v| < sqrt(2 G M_enc / max(r_com, 0.1)), M_enc = mass inside r_com (own mass included)
//                                                                                              metrics.py:98-145
//   dispersion      unbiased standard deviation of |v|                                         metrics.py:148-156
//
// Design: the two rankings (the percentile radius; the enclosed mass inside every star's r_com) follow the reference's
// own route -- sort, then a prefix sum -- in O(N log N): a stable radix sort of the radii's bit patterns (nb_sort.hip),
// a fixed-order three-pass fp64 scan of the masses in r_com order, and one pass of per-star decisions.  (Round 2's
// first version ranked by an all-pairs sweep: 8.2 ms at N = 65 536, 40 ms at N = 262 144)  Ties
// keep index order.  Bin sums are added in a fixed order (one workgroup per bin), in fp64.  Per-particle arithmetic
// follows torch op by op in the state's LOGICAL dtype A (fp32 state tensors: fp32 products, sums, sqrt, divide -- no
// fma), so that bin membership and the escape test take the decisions the reference takes.
#include "nb_internal.h"

namespace {

constexpr int MB = 256;              // threads per block
constexpr int M_PART_BLOCKS = 256;   // stage-1 blocks of the O(N) reductions

template <typename A> __device__ __forceinline__ A sqrt_rn(A x);
// (__fsqrt_rn compiles to the bare v_sqrt_f32 -- 1 ulp -- on this toolchain; sqrtf carries the fix-up to correct rounding)
template <> __device__ __forceinline__ float sqrt_rn<float>(float x) { return __builtin_sqrtf(x); }
template <> __device__ __forceinline__ double sqrt_rn<double>(double x) { return __dsqrt_rn(x); }
template <typename A> __device__ __forceinline__ A div_rn(A a, A b);
template <> __device__ __forceinline__ float div_rn<float>(float a, float b) { return __fdiv_rn(a, b); }
template <> __device__ __forceinline__ double div_rn<double>(double a, double b) { return __ddiv_rn(a, b); }

// fixed-order block sum / max of doubles
__device__ __forceinline__ double block_sum(double v, double *s)
{
    s[threadIdx.x] = v;
    __syncthreads();
    for (int st = MB / 2; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
        __syncthreads();
    }
    const double r = s[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ double block_max(double v, double *s)
{
    s[threadIdx.x] = v;
    __syncthreads();
    for (int st = MB / 2; st >= 1; st >>= 1) {
        // NaN-propagating like torch.max(): a NaN radius poisons the maximum
        if ((int)threadIdx.x < st) {
            const double a = s[threadIdx.x], b = s[threadIdx.x + st];
            s[threadIdx.x] = (a != a || b != b) ? __builtin_nan("") : (a > b ? a : b);
        }
        __syncthreads();
    }
    const double r = s[0];
    __syncthreads();
    return r;
}

// K1: per-particle radii / tangential speeds / speeds, and stage 1 of the O(N) sums.
// part[blk][0..6] = max r, sum m, sum x0 m, sum x1 m, sum x2 m, sum |v|, (unused)
template <typename S, typename A, int D>
__global__ void __launch_bounds__(MB)
metrics_prep_kernel(const S *__restrict__ pos, const S *__restrict__ vel, const S *__restrict__ mass, int n,
                    A *__restrict__ r_out, A *__restrict__ vt_out, A *__restrict__ vm_out, double *__restrict__ part)
{
    __shared__ double s_red[MB];
    double mx = -1.0, sm = 0.0, sx[3] = {0.0, 0.0, 0.0}, sv = 0.0;
    for (int i = blockIdx.x * MB + threadIdx.x; i < n; i += gridDim.x * MB) {
        A x[D], v[D];
#pragma unroll
        for (int k = 0; k < D; ++k) { x[k] = (A)pos[(size_t)i * D + k]; v[k] = (A)vel[(size_t)i * D + k]; }
        A r2 = x[0] * x[0], v2 = v[0] * v[0];
#pragma unroll
        for (int k = 1; k < D; ++k) { r2 = r2 + x[k] * x[k]; v2 = v2 + v[k] * v[k]; }     // -ffp-contract=off: no fma
        const A r = sqrt_rn<A>(r2), vm = sqrt_rn<A>(v2);
        A cr = x[0] * v[1] - x[1] * v[0];
        cr = cr < (A)0 ? -cr : cr;
        const A rc = (r < (A)0.1) ? (A)0.1 : r;                  // clamp(min=0.1); NaN stays NaN
        r_out[i] = r;
        vt_out[i] = div_rn<A>(cr, rc);
        vm_out[i] = vm;
        const A m = (A)mass[i];
        const double rd = (double)r;
        mx = (rd != rd || mx != mx) ? __builtin_nan("") : (rd > mx ? rd : mx);
        sm += (double)m;
#pragma unroll
        for (int k = 0; k < D; ++k) sx[k] += (double)(A)(x[k] * m);
        sv += (double)vm;
    }
    const double bmx = block_max(mx, s_red);
    const double bsm = block_sum(sm, s_red);
    const double b0 = block_sum(sx[0], s_red), b1 = block_sum(sx[1], s_red), b2 = block_sum(sx[2], s_red);
    const double bsv = block_sum(sv, s_red);
    if (threadIdx.x == 0) {
        double *p = part + (size_t)blockIdx.x * 8;
        p[0] = bmx; p[1] = bsm; p[2] = b0; p[3] = b1; p[4] = b2; p[5] = bsv;
    }
}

// K2 (one block): finish the O(N) sums; centre of mass in A; bin edges.
// glob[0] = max r, [1..3] = com, [4] = mean |v|, [5] = max_radius used; edges[0..nb] float.
template <typename A>
__global__ void __launch_bounds__(MB)
metrics_finish_sums_kernel(const double *__restrict__ part, int nblocks, int n, double max_radius_in,
                           const float *__restrict__ edges_in, int num_bins, double *__restrict__ glob,
                           float *__restrict__ edges)
{
    __shared__ double s_red[MB];
    double mx = -1.0, sm = 0.0, s0 = 0.0, s1 = 0.0, s2 = 0.0, sv = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += MB) {
        const double *p = part + (size_t)b * 8;
        mx = (p[0] != p[0] || mx != mx) ? __builtin_nan("") : (p[0] > mx ? p[0] : mx);
        sm += p[1]; s0 += p[2]; s1 += p[3]; s2 += p[4]; sv += p[5];
    }
    mx = block_max(mx, s_red);
    sm = block_sum(sm, s_red);
    s0 = block_sum(s0, s_red);
    s1 = block_sum(s1, s_red);
    s2 = block_sum(s2, s_red);
    sv = block_sum(sv, s_red);
    if (threadIdx.x == 0) {
        const A tm = (A)sm;
        glob[0] = mx;
        glob[1] = (double)div_rn<A>((A)s0, tm);
        glob[2] = (double)div_rn<A>((A)s1, tm);
        glob[3] = (double)div_rn<A>((A)s2, tm);
        glob[4] = sv / (double)n;
        glob[5] = max_radius_in >= 0.0 ? max_radius_in : mx;
        glob[6] = __builtin_nan("");          // r_kth: written by the one particle of that rank
    }
    if (num_bins > 0 && (int)threadIdx.x <= num_bins) {
        if (edges_in) {
            edges[threadIdx.x] = edges_in[threadIdx.x];
        } else {
            // torch.linspace(0, max_radius, num_bins + 1) in float32: symmetric formulation of ATen (start + i*step for
            // the lower half, end - (steps-1-i)*step for the upper half); callers that need the bins of the reference
            // to the last bit pass the edges they built with torch itself
            const float end = (float)(max_radius_in >= 0.0 ? max_radius_in : mx);
            const int steps = num_bins + 1, i = threadIdx.x;
            const float step = __fdiv_rn(end, (float)(steps - 1));
            edges[i] = (i < steps / 2) ? __fmul_rn(step, (float)i) : __fsub_rn(end, __fmul_rn(step, (float)(steps - i - 1)));
        }
    }
}

// Sort keys: radii are >= +0 or NaN, so their bit patterns order like the values; NaN goes last (torch.sort's rule).
template <typename A> struct KeyOf;
template <> struct KeyOf<float> {
    typedef unsigned type;
    static __device__ __forceinline__ unsigned make(float x) { return x != x ? 0xffffffffu : __float_as_uint(x); }
    static __device__ __forceinline__ float back(unsigned k) { return __uint_as_float(k); }   // all-ones is a NaN
};
template <> struct KeyOf<double> {
    typedef unsigned long long type;
    static __device__ __forceinline__ unsigned long long make(double x)
    { return x != x ? ~0ull : (unsigned long long)__double_as_longlong(x); }
    static __device__ __forceinline__ double back(unsigned long long k) { return __longlong_as_double((long long)k); }
};

// K3: distance from the centre of mass; sort keys of r and r_com; identity permutation
template <typename S, typename A, int D>
__global__ void __launch_bounds__(MB)
metrics_rcom_kernel(const S *__restrict__ pos, const A *__restrict__ r, int n, const double *__restrict__ glob,
                    A *__restrict__ rcom, typename KeyOf<A>::type *__restrict__ key_r,
                    typename KeyOf<A>::type *__restrict__ key_rc, int *__restrict__ idx)
{
    const int i = blockIdx.x * MB + threadIdx.x;
    if (i >= n) return;
    A s = (A)0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const A d = (A)pos[(size_t)i * D + k] - (A)glob[1 + k];
        s = (k == 0) ? d * d : s + d * d;
    }
    const A rc = sqrt_rn<A>(s);
    rcom[i] = rc;
    key_r[i] = KeyOf<A>::make(r[i]);
    key_rc[i] = KeyOf<A>::make(rc);
    idx[i] = i;
}

// Enclosed mass = inclusive prefix sum of the masses in r_com order (metrics.py:128-134: argsort, cumsum -- whose CPU
// kernel accumulates in double and rounds every element to the tensor's dtype).  Three fixed-order passes over
// blocks of SCAN_EB sorted positions: block totals, their exclusive scan, the in-block scan + per-star decisions.
constexpr int SCAN_IT = 4, SCAN_EB = MB * SCAN_IT;

template <typename S, typename A>
__global__ void __launch_bounds__(MB)
metrics_scan_totals_kernel(const S *__restrict__ mass, const int *__restrict__ order, int n, double *__restrict__ bsum)
{
    __shared__ double s_red[MB];
    const int k0 = blockIdx.x * SCAN_EB + threadIdx.x * SCAN_IT;
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < SCAN_IT; ++q)
        if (k0 + q < n) t += (double)(A)mass[order[k0 + q]];
    t = block_sum(t, s_red);
    if (threadIdx.x == 0) bsum[blockIdx.x] = t;
}

__global__ void metrics_scan_offsets_kernel(double *__restrict__ bsum, int nblocks)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double run = 0.0;
    for (int b = 0; b < nblocks; ++b) { const double t = bsum[b]; bsum[b] = run; run += t; }
}

// K4: star i = order[k] of sorted position k: enclosed mass (own mass included, ties in index order -- the sort is
// stable), bound flag, radial bin; thread 0 publishes the kth smallest radius.
template <typename S, typename A>
__global__ void __launch_bounds__(MB)
metrics_decide_kernel(const A *__restrict__ r, const A *__restrict__ rcom, const S *__restrict__ mass,
                      const A *__restrict__ vm, const int *__restrict__ order, const double *__restrict__ boff,
                      const typename KeyOf<A>::type *__restrict__ key_r_sorted, int n, int kth, double G,
                      const float *__restrict__ edges, int num_bins, int *__restrict__ bin_out,
                      unsigned char *__restrict__ bound_out, double *__restrict__ glob)
{
    __shared__ double s_scan[MB];
    const int k0 = blockIdx.x * SCAN_EB + threadIdx.x * SCAN_IT;
    int who[SCAN_IT];
    double pre[SCAN_IT], t = 0.0;
#pragma unroll
    for (int q = 0; q < SCAN_IT; ++q) {
        who[q] = k0 + q < n ? order[k0 + q] : -1;
        if (who[q] >= 0) t += (double)(A)mass[who[q]];
        pre[q] = t;
    }
    // exclusive scan of the thread totals (Hillis-Steele, fixed order)
    s_scan[threadIdx.x] = t;
    __syncthreads();
    for (int st = 1; st < MB; st <<= 1) {
        const double add = (int)threadIdx.x >= st ? s_scan[threadIdx.x - st] : 0.0;
        __syncthreads();
        s_scan[threadIdx.x] += add;
        __syncthreads();
    }
    const double base = boff[blockIdx.x] + (s_scan[threadIdx.x] - t);
    if (blockIdx.x == 0 && threadIdx.x == 0) glob[6] = (double)KeyOf<A>::back(key_r_sorted[kth]);
    const A g2 = (A)(2.0 * G);            // metrics.py:133: the Python scalar 2*G enters as A
#pragma unroll
    for (int q = 0; q < SCAN_IT; ++q) {
        const int i = who[q];
        if (i < 0) continue;
        const A enc = (A)(base + pre[q]);
        const A rci = rcom[i], ri = r[i];
        const A rcc = (rci < (A)0.1) ? (A)0.1 : rci;
        const A vesc = sqrt_rn<A>(div_rn<A>(g2 * enc, rcc));
        bound_out[i] = (vm[i] < vesc) ? 1 : 0;
        // bin b holds edge_b <= r < edge_b+1 (metrics.py:65); r is compared in A against the float32 edges
        int b = -1;
        if (num_bins > 0) {
            int c = 0;
            for (int e = 0; e <= num_bins; ++e) c += ((A)edges[e] <= ri) ? 1 : 0;
            b = (c >= 1 && c <= num_bins) ? c - 1 : -1;
        }
        bin_out[i] = b;
    }
}

// K5: block b < num_bins: mean tangential speed and count of bin b; block num_bins: bound count and dispersion.
// out[0] = max r, [1] = r_kth, [2] = bound count, [3] = dispersion, [4 + b] = mean of bin b, [4 + nb + b] = count
template <typename A>
__global__ void __launch_bounds__(MB)
metrics_bins_kernel(const A *__restrict__ vt, const A *__restrict__ vm, const int *__restrict__ bin,
                    const unsigned char *__restrict__ bound, int n, int num_bins, const double *__restrict__ glob,
                    double *__restrict__ out)
{
    __shared__ double s_red[MB];
    const int b = blockIdx.x;
    if (b < num_bins) {
        double s = 0.0, c = 0.0;
        for (int i = threadIdx.x; i < n; i += MB)
            if (bin[i] == b) { s += (double)vt[i]; c += 1.0; }
        s = block_sum(s, s_red);
        c = block_sum(c, s_red);
        if (threadIdx.x == 0) {
            out[4 + b] = c > 0.0 ? (double)(A)(s / c) : __builtin_nan("");
            out[4 + num_bins + b] = c;
        }
        return;
    }
    double nb = 0.0, ss = 0.0;
    const double mean = glob[4];
    for (int i = threadIdx.x; i < n; i += MB) {
        nb += bound[i] ? 1.0 : 0.0;
        const double d = (double)vm[i] - mean;
        ss += d * d;
    }
    nb = block_sum(nb, s_red);
    ss = block_sum(ss, s_red);
    if (threadIdx.x == 0) {
        out[0] = glob[0];
        out[1] = glob[6];
        out[2] = nb;
        out[3] = n > 1 ? (double)(A)sqrt(ss / (double)(n - 1)) : __builtin_nan("");
        out[5 + 2 * num_bins - 1] = glob[5];        // out[4 + 2 nb] = max_radius used for the edges
    }
}

template <typename S, typename A, int D>
hipError_t run(const NbMetricsArgs &a, hipStream_t st)
{
    const int n = a.n, nb = a.num_bins;
    char *p = (char *)a.scratch;
    auto take = [&](size_t bytes) { char *q = p; p += (bytes + 255) & ~(size_t)255; return q; };
    A *r = (A *)take(sizeof(A) * n), *vt = (A *)take(sizeof(A) * n), *vm = (A *)take(sizeof(A) * n);
    A *rcom = (A *)take(sizeof(A) * n);
    typedef typename KeyOf<A>::type K;
    K *key_r = (K *)take(sizeof(K) * n), *key_rc = (K *)take(sizeof(K) * n);
    K *key_r_s = (K *)take(sizeof(K) * n), *key_rc_s = (K *)take(sizeof(K) * n);
    int *idx = (int *)take(sizeof(int) * n), *order = (int *)take(sizeof(int) * n);
    int *bin = (int *)take(sizeof(int) * n);
    unsigned char *bound = (unsigned char *)take(n);
    const int sblocks = (n + SCAN_EB - 1) / SCAN_EB;
    double *bsum = (double *)take(sizeof(double) * (sblocks + 1));
    double *part = (double *)take(sizeof(double) * 8 * M_PART_BLOCKS);
    double *glob = (double *)take(sizeof(double) * 8);
    float *edges = (float *)take(sizeof(float) * (nb + 2));
    const int key64 = sizeof(K) == 8;
    const size_t sort_bytes = nb_sort_temp_bytes(n, key64);
    void *sort_tmp = take(sort_bytes);
    int blocks = (n + MB - 1) / MB;
    const int pblocks = blocks < M_PART_BLOCKS ? blocks : M_PART_BLOCKS;
    hipLaunchKernelGGL((metrics_prep_kernel<S, A, D>), dim3(pblocks), dim3(MB), 0, st, (const S *)a.pos, (const S *)a.vel,
                       (const S *)a.mass, n, r, vt, vm, part);
    hipLaunchKernelGGL((metrics_finish_sums_kernel<A>), dim3(1), dim3(MB), 0, st, part, pblocks, n, a.max_radius,
                       a.edges, nb, glob, edges);
    if (!a.radius_only) {
        hipLaunchKernelGGL((metrics_rcom_kernel<S, A, D>), dim3(blocks), dim3(MB), 0, st, (const S *)a.pos, r, n, glob, rcom,
                           key_r, key_rc, idx);
        hipError_t e = nb_sort_keys(sort_tmp, sort_bytes, key_r, key_r_s, n, key64, st);
        if (e != hipSuccess) return e;
        e = nb_sort_pairs(sort_tmp, sort_bytes, key_rc, key_rc_s, idx, order, n, key64, st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((metrics_scan_totals_kernel<S, A>), dim3(sblocks), dim3(MB), 0, st, (const S *)a.mass, order, n,
                           bsum);
        hipLaunchKernelGGL(metrics_scan_offsets_kernel, dim3(1), dim3(64), 0, st, bsum, sblocks);
        hipLaunchKernelGGL((metrics_decide_kernel<S, A>), dim3(sblocks), dim3(MB), 0, st, r, rcom, (const S *)a.mass, vm,
                           order, bsum, key_r_s, n, a.kth, a.G, edges, nb, bin, bound, glob);
    }
    // radius_only: no bins, no particles -> the last block just publishes max r / the max_radius in use
    const int nb_eff = a.radius_only ? 0 : nb, n_eff = a.radius_only ? 0 : n;
    hipLaunchKernelGGL((metrics_bins_kernel<A>), dim3(nb_eff + 1), dim3(MB), 0, st, vt, vm, bin, bound, n_eff, nb_eff, glob,
                       a.out);
    return hipGetLastError();
}

}  // namespace

size_t nb_metrics_scratch_bytes(int n, int num_bins)
{
    const size_t al = 256;
    auto up = [&](size_t bytes) { return (bytes + al - 1) & ~(al - 1); };
    size_t b = 0;
    b += 4 * up((size_t)n * 8);                     // r, vt, vm, rcom
    b += 4 * up((size_t)n * 8);                     // keys and sorted keys of r, r_com
    b += 3 * up((size_t)n * 4);                     // idx, order, bin
    b += up((size_t)n);                             // bound
    b += up(8 * ((size_t)(n + SCAN_EB - 1) / SCAN_EB + 1));
    b += up(8 * 8 * M_PART_BLOCKS) + up(64) + up((size_t)(num_bins + 2) * 4);
    const size_t s32 = nb_sort_temp_bytes(n, 0), s64 = nb_sort_temp_bytes(n, 1);
    b += up(s32 > s64 ? s32 : s64);
    return b + 1024;
}

hipError_t nb_launch_metrics(const NbMetricsArgs &a, hipStream_t st)
{
    if (a.dim != 2 && a.dim != 3) return hipErrorInvalidValue;
    if (a.num_bins < 0 || a.num_bins > MB - 1) return hipErrorInvalidValue;
#define NB_MET(SS, AA) (a.dim == 2 ? run<SS, AA, 2>(a, st) : run<SS, AA, 3>(a, st))
    if (a.storage_f64) return a.arith_f64 ? NB_MET(double, double) : NB_MET(double, float);
    return a.arith_f64 ? NB_MET(float, double) : NB_MET(float, float);
#undef NB_MET
}
