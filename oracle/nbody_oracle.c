/*
 * nbody_oracle.c -- CPU restatement of the reference's direct-sum hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / the reported CPU baseline.  The product path is the HIP library in
 * nbody_cosmological_simulation_amd/csrc and has no CPU fallback.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against
 * golden vectors produced by running the reference itself (tests/golden/make_golden.py).
 *
 * What is restated (reference file:line):
 *   simulation.py:74-118   _compute_accelerations   -> nbo_accelerations()
 *   simulation.py:120-143  step (KDK leapfrog)       -> nbo_step()
 *   simulation.py:170-192  kinetic / potential energy-> nbo_kinetic_energy(), nbo_potential_energy()
 *   quantization.py:21-71  quantize_distance_squared -> hook_r2() / nbo_quantize_distance_squared()
 *   quantization.py:74-88  _grid_quantize            -> nbo_grid_quantize()
 *   quantization.py:91-127 _grid_quantize_safe       -> nbo_grid_quantize_safe()
 *   quantization.py:130-157 quantize_force           -> nbo_quantize_force()
 *
 * Arithmetic model.  The reference is dtype-polymorphic PyTorch; every tensor op rounds to
 * the tensor's dtype.  Here every value is carried in a C double that always holds a value
 * exactly representable in its LOGICAL dtype (f16/bf16/f32/f64), and every elementary
 * operation is `rnd(T, a op b)`: the op evaluated in double and rounded once to T.  For
 * + - * / sqrt this is bit-identical to native arithmetic in T (53 >= 2*24+2 bits), so the
 * fp32 paths reproduce IEEE float arithmetic WITHOUT fused multiply-add, in the reference's
 * operation order.  pow/log/exp are evaluated in double and rounded
 * to T, i.e. correctly rounded up to double-rounding ties (torch's SLEEF kernels are <=1 ulp;
 * golden tests bound the difference).  Reductions over j accumulate in double and round once
 * (torch's cascade sum is within a few ulp of that; reduction order is unspecified upstream).
 *
 * Build: see oracle/Makefile (-ffp-contract=off is REQUIRED).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { NBO_F16 = 0, NBO_BF16 = 1, NBO_F32 = 2, NBO_F64 = 3 };
enum { NBO_FLOAT64 = 0, NBO_FLOAT32 = 1, NBO_BFLOAT16 = 2, NBO_FLOAT16 = 3,
       NBO_INT8 = 4, NBO_INT4 = 5, NBO_CUSTOM = 6 };

/* ------------------------------------------------------------------ rounding helpers */

static double rnd_f32(double x) { return (double)(float)x; }

/* round-to-nearest-even to a binary format with `mant` explicit mantissa bits, exponent
 * range [emin, emax] (unbiased), subnormals kept, overflow -> inf.  x is a double. */
static double rnd_small(double x, int mant, int emin, int emax)
{
    if (x == 0.0 || isnan(x) || isinf(x)) return x;
    int e;
    double m = frexp(fabs(x), &e);          /* |x| = m * 2^e, m in [0.5,1) */
    int ue = e - 1;                         /* unbiased exponent of leading bit */
    int q = (ue < emin) ? emin : ue;        /* quantum exponent base */
    double scale = ldexp(1.0, q - mant);    /* value of one ulp */
    double r = nearbyint(fabs(x) / scale) * scale;   /* RNE (default rounding mode) */
    (void)m;
    double maxv = ldexp(2.0 - ldexp(1.0, -mant), emax);
    if (r > maxv) {
        /* IEEE: values >= maxv + ulp/2 overflow to inf; nearbyint above already rounded,
         * so anything that rounded past maxv is inf. */
        r = INFINITY;
    }
    return x < 0 ? -r : r;
}
static double rnd_f16(double x)  { return rnd_small(x, 10, -14, 15); }
static double rnd_bf16(double x) { return rnd_small(x, 7, -126, 127); }

static double rnd(int T, double x)
{
    switch (T) {
    case NBO_F64: return x;
    case NBO_F32: return rnd_f32(x);
    case NBO_F16: return rnd_f16(x);
    default:      return rnd_bf16(x);
    }
}
/* torch "opmath" type: reductions and scalar operands of half types are handled in float */
static int opmath(int T) { return (T == NBO_F64) ? NBO_F64 : NBO_F32; }
/* a Python scalar ADDED to (or subtracted from, or used as a clamp bound of) a tensor of dtype T is first cast to T
 * itself (TensorIterator wraps it as a 0-dim tensor and converts it to the common dtype) ... */
static double scalar_as(int T, double s) { return rnd(T, s); }
/* ... but a Python scalar that MULTIPLIES or DIVIDES a tensor of a half type stays in float: torch's CPU mul / div
 * kernels take the scalar operand in opmath precision, x * s = half(float(x) * float(s)).  Measured with torch 2.10
 * (round 3, tests/test_oracle_golden.py::test_torch_scalar_semantics_on_half_tensors): float16 x * 0.001 equals the
 * float-scalar form on 100 % of 2e5 samples and the half-rounded-scalar form on 42 %; `G / t` is t.reciprocal() (rounded
 * to the half type) times float(G).  Rounding G = 0.001 to float16 first (0.0010004) was a 4e-4 relative error of every
 * force in the half-typed grid modes -- the cause of the 4 % force-bin disagreement VERDICT r2 weak #2 flagged. */
static double scalar_mul(int T, double s) { return rnd((T == NBO_F64) ? NBO_F64 : NBO_F32, s); }

static int promote(int a, int b)
{
    if (a == b) return a;
    if (a == NBO_F64 || b == NBO_F64) return NBO_F64;
    if (a == NBO_F32 || b == NBO_F32) return NBO_F32;
    return NBO_F32; /* f16 with bf16 */
}
int nbo_promote(int a, int b) { return promote(a, b); }
double nbo_round(int T, double x) { return rnd(T, x); }

/* ------------------------------------------------------------------ r2 for one pair
 * simulation.py:83-86: diff = x_j - x_i ; dist_sq = (diff**2).sum(-1) + softening_sq   */
static inline double pair_r2(int P, int d, const double *xi, const double *xj, double eps2_py,
                             double *diff)
{
    int O = opmath(P);
    double s = 0.0;
    for (int k = 0; k < d; ++k) {
        diff[k] = rnd(P, xj[k] - xi[k]);
        double sq = rnd(P, diff[k] * diff[k]);       /* diff ** 2 materialised in P */
        s = (k == 0) ? sq : rnd(O, s + sq);          /* sum accumulates in opmath, in order */
    }
    s = rnd(P, s);
    return rnd(P, s + scalar_as(P, eps2_py));
}

/* quantization.py:21-71 for the non-grid modes; returns value, *Q = result dtype */
static inline double hook_simple(int mode, int P, double r2, int *Q)
{
    switch (mode) {
    case NBO_FLOAT64:  *Q = NBO_F64; return r2;                       /* .double() */
    case NBO_FLOAT32:  *Q = NBO_F32; return rnd_f32(r2);              /* .float()  */
    /* torch converts double -> half/bfloat16 through float (c10::Half(float)): two roundings */
    case NBO_BFLOAT16: *Q = NBO_F32; return rnd_bf16(rnd_f32(r2));    /* .bfloat16().float() */
    case NBO_FLOAT16:  *Q = NBO_F32; return rnd_f16(rnd_f32(r2));     /* .half().float() */
    default:           *Q = P;       return r2;
    }
}

/* torch.clamp(min=...) propagates NaN (fmax would drop it) */
static inline double clamp_min(double x, double lo) { return isnan(x) ? x : (x < lo ? lo : x); }

static int mode_levels(int mode, int levels)
{
    if (mode == NBO_INT8) return 256;
    if (mode == NBO_INT4) return 16;
    return levels > 0 ? levels : 64;   /* quantization.py:66 `custom_levels or 64` */
}

/* ---- quantization.py:91-127 scalar pieces (T = tensor dtype) */
static inline double gqs_log(int T, double t, double min_val)
{
    return rnd(T, log(clamp_min(t, scalar_as(T, min_val))));   /* clamp(min=min_val).log() */
}
static inline double gqs_bin(int T, double lt, double lmin, double lmax, int L)
{
    double range = rnd(T, lmax - lmin);
    double n = rnd(T, rnd(T, lt - lmin) / range);
    n = rnd(T, n * scalar_mul(T, (double)(L - 1)));
    return nearbyint(n);                           /* torch.round: half to even */
}
static inline double gqs_value(int T, double k, double lmin, double lmax, int L, double min_val)
{
    double range = rnd(T, lmax - lmin);
    double v = rnd(T, k / scalar_mul(T, (double)(L - 1)));
    v = rnd(T, v * range);
    v = rnd(T, v + lmin);
    v = rnd(T, exp(v));
    return clamp_min(v, scalar_as(T, min_val));
}

/* ------------------------------------------------------------------ tensor-level hooks */

/* quantization.py:91-127.  in/out: n doubles holding dtype-T values.  bins may be NULL.
 * returns 1 if the degenerate (lmax-lmin < 1e-10) branch was taken. */
int nbo_grid_quantize_safe(long n, int T, const double *in, double *out, int levels, double min_val,
                           double *lmin_out, double *lmax_out, int32_t *bins)
{
    double lmin = INFINITY, lmax = -INFINITY;
    int has_nan = 0;
    for (long i = 0; i < n; ++i) {
        double lt = gqs_log(T, in[i], min_val);
        if (isnan(lt)) has_nan = 1;
        if (lt < lmin) lmin = lt;
        if (lt > lmax) lmax = lt;
    }
    if (has_nan) lmin = lmax = NAN;           /* torch min()/max() propagate NaN */
    if (lmin_out) *lmin_out = lmin;
    if (lmax_out) *lmax_out = lmax;
    if (rnd(T, lmax - lmin) < 1e-10) {
        for (long i = 0; i < n; ++i) {
            out[i] = clamp_min(in[i], scalar_as(T, min_val));
            if (bins) bins[i] = -1;
        }
        return 1;
    }
    for (long i = 0; i < n; ++i) {
        double lt = gqs_log(T, in[i], min_val);
        double k = gqs_bin(T, lt, lmin, lmax, levels);
        if (bins) bins[i] = isnan(k) ? -2 : (int32_t)k;
        out[i] = gqs_value(T, k, lmin, lmax, levels, min_val);
    }
    return 0;
}

/* quantization.py:74-88 */
int nbo_grid_quantize(long n, int T, const double *in, double *out, int levels,
                      double *mn_out, double *mx_out, int32_t *bins)
{
    double mn = INFINITY, mx = -INFINITY;
    int has_nan = 0;
    for (long i = 0; i < n; ++i) {
        if (isnan(in[i])) has_nan = 1;
        if (in[i] < mn) mn = in[i];
        if (in[i] > mx) mx = in[i];
    }
    if (has_nan) mn = mx = NAN;
    if (mn_out) *mn_out = mn;
    if (mx_out) *mx_out = mx;
    double range = rnd(T, mx - mn);
    if (range < 1e-10) {                       /* NaN compares false -> falls through like torch */
        for (long i = 0; i < n; ++i) { out[i] = in[i]; if (bins) bins[i] = -1; }
        return 1;
    }
    double lm1 = scalar_mul(T, (double)(levels - 1));
    for (long i = 0; i < n; ++i) {
        double v = rnd(T, rnd(T, in[i] - mn) / range);
        v = rnd(T, v * lm1);
        double k = nearbyint(v);
        if (bins) bins[i] = isnan(k) ? -2 : (int32_t)k;
        v = rnd(T, k / lm1);
        v = rnd(T, v * range);
        out[i] = rnd(T, v + mn);
    }
    return 0;
}

/* quantization.py:21-71 on a flat tensor; *Tout receives the output dtype */
int nbo_quantize_distance_squared(long n, int T, const double *in, double *out, int mode, int levels,
                                  double min_val, int *Tout)
{
    if (mode <= NBO_FLOAT16) {
        int Q = T;
        for (long i = 0; i < n; ++i) out[i] = hook_simple(mode, T, in[i], &Q);
        *Tout = Q;
        return 0;
    }
    *Tout = T;
    return nbo_grid_quantize_safe(n, T, in, out, mode_levels(mode, levels), min_val, 0, 0, 0);
}

/* quantization.py:130-157 */
int nbo_quantize_force(long n, int T, const double *in, double *out, int mode, int levels, int *Tout,
                       double *mn, double *mx, int32_t *bins)
{
    *Tout = T;
    switch (mode) {
    case NBO_FLOAT64: case NBO_FLOAT32:
        memcpy(out, in, sizeof(double) * n); return 0;
    case NBO_BFLOAT16:
        for (long i = 0; i < n; ++i) out[i] = rnd_bf16(in[i]);
        *Tout = NBO_F32; return 0;
    case NBO_FLOAT16:
        for (long i = 0; i < n; ++i) out[i] = rnd_f16(in[i]);
        *Tout = NBO_F32; return 0;
    default:
        return nbo_grid_quantize(n, T, in, out, mode_levels(mode, levels), mn, mx, bins);
    }
}

/* ------------------------------------------------------------------ accelerations
 * simulation.py:74-118.  pos: n*d doubles holding dtype-P values, mass: n doubles (dtype M).
 * Partial sums over sources j in [j0, j1) (j-block sharding; the grid quantiser's global
 * log-min/max always spans ALL n*n pairs like the reference).  When apply_force_quant != 0
 * and the range is full, INT8/INT4 force quantisation (simulation.py:115-116) is applied.
 * Debug outputs (any may be NULL): dbg[0..3] = lmin, lmax, fmin, fmax; d2bins n*n int32
 * (row i, column j; only rows of the j-range are written), fbins n*d, acc_prequant n*d.
 * Returns the dtype of acc_out. */
int nbo_accelerations(int n, int d, int P, const double *pos, int M, const double *mass,
                      int mode, int levels, double G, double eps2_py,
                      int j0, int j1, int apply_force_quant,
                      double *acc_out, double *dbg, int32_t *d2bins, int32_t *fbins,
                      double *acc_prequant)
{
    const int grid = (mode >= NBO_INT8);
    const int L = mode_levels(mode, levels);
    const double min_val = 0.01;                     /* quantization.py:25 default */
    int Q = P;
    { double tmp = hook_simple(mode, P, 1.0, &Q); (void)tmp; }
    if (grid) Q = P;
    const int W = promote(Q, M);                     /* force_factor * masses */
    const int W2 = promote(W, NBO_F32);              /* * (1 - torch.eye(n)) : eye is float32 */
    const int A = promote(W2, P);                    /* force_factor * diff */

    double lmin = 0, lmax = 0;
    int degenerate = 0;
    if (grid) {
        /* global min / max of log(clamp(r2)) over all n*n entries incl. the diagonal */
        double gmin = INFINITY, gmax = -INFINITY;
        int has_nan = 0;
        #pragma omp parallel for reduction(min:gmin) reduction(max:gmax) reduction(|:has_nan) schedule(static)
        for (int i = 0; i < n; ++i) {
            double diff[4];
            for (int j = 0; j < n; ++j) {
                double r2 = pair_r2(P, d, pos + (long)i * d, pos + (long)j * d, eps2_py, diff);
                double lt = gqs_log(P, r2, min_val);
                if (isnan(lt)) has_nan = 1;
                if (lt < gmin) gmin = lt;
                if (lt > gmax) gmax = lt;
            }
        }
        lmin = has_nan ? NAN : gmin;
        lmax = has_nan ? NAN : gmax;
        degenerate = (rnd(P, lmax - lmin) < 1e-10);
    }
    if (dbg) { dbg[0] = lmin; dbg[1] = lmax; dbg[2] = 0; dbg[3] = 0; }

    const double Gs = scalar_mul(Q, G);
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        double diff[4], acc[4] = {0, 0, 0, 0};
        for (int j = j0; j < j1; ++j) {
            double r2 = pair_r2(P, d, pos + (long)i * d, pos + (long)j * d, eps2_py, diff);
            double q;
            if (!grid) {
                int Qd; q = hook_simple(mode, P, r2, &Qd);
            } else if (degenerate) {
                q = clamp_min(r2, scalar_as(P, min_val));
                if (d2bins) d2bins[(long)i * n + j] = -1;
            } else {
                double lt = gqs_log(P, r2, min_val);
                double k = gqs_bin(P, lt, lmin, lmax, L);
                if (d2bins) d2bins[(long)i * n + j] = isnan(k) ? -2 : (int32_t)k;
                q = gqs_value(P, k, lmin, lmax, L, min_val);
            }
            double p = rnd(Q, pow(q, 1.5));                         /* :97  */
            double w = rnd(Q, rnd(Q, 1.0 / p) * Gs);                /* :101 reciprocal()*G */
            w = rnd(W, w * mass[j]);                                /* :105 */
            w = rnd(W2, w * ((i == j) ? 0.0 : 1.0));                /* :108 */
            for (int k = 0; k < d; ++k)
                acc[k] += rnd(A, w * diff[k]);                      /* :112, summed in double */
        }
        for (int k = 0; k < d; ++k) acc_out[(long)i * d + k] = rnd(A, acc[k]);
    }

    if (acc_prequant) memcpy(acc_prequant, acc_out, sizeof(double) * (size_t)n * d);
    if (apply_force_quant && (mode == NBO_INT8 || mode == NBO_INT4) && j0 == 0 && j1 == n) {
        double mn, mx; int Tout;
        double *tmp = (double *)malloc(sizeof(double) * (size_t)n * d);
        nbo_quantize_force((long)n * d, A, acc_out, tmp, mode, levels, &Tout, &mn, &mx, fbins);
        memcpy(acc_out, tmp, sizeof(double) * (size_t)n * d);
        free(tmp);
        if (dbg) { dbg[2] = mn; dbg[3] = mx; }
    }
    return A;
}

/* Rows [i0, i1) of the same evaluation at sizes where the full N x N pass is too slow for a test: the grid's
 * global log-min/max still spans ALL n*n pairs (log is monotone, so they follow from the extreme r2 values, found
 * without a log per pair); no force quantisation (it needs every row).  acc_out: (i1-i0)*d, d2bins: (i1-i0)*n. */
int nbo_accelerations_rows(int n, int d, int P, const double *pos, int M, const double *mass,
                           int mode, int levels, double G, double eps2_py, int i0, int i1,
                           double *acc_out, double *dbg, int32_t *d2bins)
{
    const int grid = (mode >= NBO_INT8);
    const int L = mode_levels(mode, levels);
    const double min_val = 0.01;
    int Q = P;
    { double tmp = hook_simple(mode, P, 1.0, &Q); (void)tmp; }
    if (grid) Q = P;
    const int W = promote(Q, M);
    const int W2 = promote(W, NBO_F32);
    const int A = promote(W2, P);
    double lmin = 0, lmax = 0;
    int degenerate = 0;
    if (grid) {
        double rmin = INFINITY, rmax = -INFINITY;
        int has_nan = 0;
        #pragma omp parallel for reduction(min:rmin) reduction(max:rmax) reduction(|:has_nan) schedule(static)
        for (int i = 0; i < n; ++i) {
            double diff[4];
            for (int j = 0; j < n; ++j) {
                double r2 = pair_r2(P, d, pos + (long)i * d, pos + (long)j * d, eps2_py, diff);
                if (isnan(r2)) has_nan = 1;
                if (r2 < rmin) rmin = r2;
                if (r2 > rmax) rmax = r2;
            }
        }
        lmin = has_nan ? NAN : gqs_log(P, rmin, min_val);
        lmax = has_nan ? NAN : gqs_log(P, rmax, min_val);
        degenerate = (rnd(P, lmax - lmin) < 1e-10);
    }
    if (dbg) { dbg[0] = lmin; dbg[1] = lmax; dbg[2] = 0; dbg[3] = 0; }
    const double Gs = scalar_mul(Q, G);
    #pragma omp parallel for schedule(static)
    for (int i = i0; i < i1; ++i) {
        double diff[4], acc[4] = {0, 0, 0, 0};
        for (int j = 0; j < n; ++j) {
            double r2 = pair_r2(P, d, pos + (long)i * d, pos + (long)j * d, eps2_py, diff);
            double q;
            if (!grid) {
                int Qd; q = hook_simple(mode, P, r2, &Qd);
            } else if (degenerate) {
                q = clamp_min(r2, scalar_as(P, min_val));
                if (d2bins) d2bins[(long)(i - i0) * n + j] = -1;
            } else {
                double lt = gqs_log(P, r2, min_val);
                double k = gqs_bin(P, lt, lmin, lmax, L);
                if (d2bins) d2bins[(long)(i - i0) * n + j] = isnan(k) ? -2 : (int32_t)k;
                q = gqs_value(P, k, lmin, lmax, L, min_val);
            }
            double p = rnd(Q, pow(q, 1.5));
            double w = rnd(Q, rnd(Q, 1.0 / p) * Gs);
            w = rnd(W, w * mass[j]);
            w = rnd(W2, w * ((i == j) ? 0.0 : 1.0));
            for (int k = 0; k < d; ++k) acc[k] += rnd(A, w * diff[k]);
        }
        for (int k = 0; k < d; ++k) acc_out[(long)(i - i0) * d + k] = rnd(A, acc[k]);
    }
    return A;
}

/* result dtype of _compute_accelerations without running it */
int nbo_acc_dtype(int P, int M, int mode)
{
    int Q = P;
    if (mode < NBO_INT8) { double t = hook_simple(mode, P, 1.0, &Q); (void)t; }
    return promote(promote(promote(Q, M), NBO_F32), P);
}

/* ------------------------------------------------------------------ KDK pieces
 * simulation.py:132,135,141:  out = a + b * scalar  with torch promotion rules.
 * Returns the result dtype. */
int nbo_axpy(long n, int Ta, const double *a, int Tb, const double *b, double scalar, double *out)
{
    int T = promote(Ta, Tb);
    double s = scalar_mul(Tb, scalar);
    for (long i = 0; i < n; ++i) {
        double t = rnd(Tb, b[i] * s);
        out[i] = rnd(T, a[i] + t);
    }
    return T;
}

/* one full step (simulation.py:120-143).  Arrays are updated in place; dts[] = {P, V, M, A}
 * dtypes in/out. */
void nbo_step(int n, int d, int *dts, double *pos, double *vel, const double *mass, double *acc,
              int mode, int levels, double G, double eps2_py, double dt)
{
    long nd = (long)n * d;
    dts[1] = nbo_axpy(nd, dts[1], vel, dts[3], acc, dt / 2, vel);
    dts[0] = nbo_axpy(nd, dts[0], pos, dts[1], vel, dt, pos);
    dts[3] = nbo_accelerations(n, d, dts[0], pos, dts[2], mass, mode, levels, G, eps2_py,
                               0, n, 1, acc, 0, 0, 0, 0);
    dts[1] = nbo_axpy(nd, dts[1], vel, dts[3], acc, dt / 2, vel);
}

/* ------------------------------------------------------------------ energies
 * simulation.py:170-174 */
double nbo_kinetic_energy(int n, int d, int V, const double *vel, int M, const double *mass)
{
    int T = promote(V, M), O = opmath(V);
    double sum = 0.0;
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int k = 0; k < d; ++k) {
            double sq = rnd(V, vel[(long)i * d + k] * vel[(long)i * d + k]);
            s = (k == 0) ? sq : rnd(O, s + sq);
        }
        s = rnd(V, s);
        sum += rnd(T, mass[i] * s);
    }
    sum = rnd(T, sum);
    return rnd(T, scalar_mul(T, 0.5) * sum);
}

/* simulation.py:176-192 ; partial over sources j in [j0,j1) (pairs i<j). */
double nbo_potential_energy(int n, int d, int P, const double *pos, int M, const double *mass,
                            double G, double eps2_py, int j0, int j1)
{
    int O = opmath(P);
    int MP = M;                   /* mass_prod dtype */
    int T = promote(MP, P);       /* (mass_prod * mask) / dist ; mask = ones_like(dist) has dtype P */
    double total = 0.0;
    #pragma omp parallel for reduction(+:total) schedule(dynamic, 16)
    for (int j = j0; j < j1; ++j) {
        double part = 0.0;
        for (int i = 0; i < j; ++i) {
            const double *xi = pos + (long)i * d, *xj = pos + (long)j * d;
            double s = 0.0;
            for (int k = 0; k < d; ++k) {
                double df = rnd(P, xj[k] - xi[k]);
                double sq = rnd(P, df * df);
                s = (k == 0) ? sq : rnd(O, s + sq);
            }
            s = rnd(P, s);
            double dist = rnd(P, sqrt(rnd(P, s + scalar_as(P, eps2_py))));
            double mp = rnd(MP, mass[i] * mass[j]);
            part += rnd(T, rnd(T, mp * 1.0) / dist);
        }
        total += part;
    }
    /* the reference multiplies by the triu mask BEFORE dividing (simulation.py:189): masked entries are 0 / dist,
     * which is NaN where dist == 0 -- on the whole diagonal when the softening rounds to zero in the positions'
     * dtype (softening 0, or 1e-4 with float16 positions).  For a non-zero softening dist > 0 everywhere. */
    if (n > 0 && j0 == 0 && scalar_as(P, eps2_py) == 0.0) return NAN;
    total = rnd(T, total);
    return rnd(T, scalar_mul(T, -G) * total);
}

/* ------------------------------------------------------------------ fast fp64 path
 * Same mathematics as nbo_accelerations(P=M=f64, mode=FLOAT64) with q*sqrt(q) in place of
 * pow (differs by <= 1 ulp; SURVEY.md A.1 note) and a vectorisable inner loop.  Used for the
 * large-N checks and as the timed CPU baseline ("port", all host cores). */
#define NBO_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))

NBO_CLONES
void nbo_accelerations_f64_fast(int n, int d, const double *pos, const double *mass, double G,
                                double eps2, int j0, int j1, double *acc_out)
{
    /* SoA copy of the sources for unit-stride vector loads */
    int m = j1 - j0;
    double *sx = (double *)malloc(sizeof(double) * (size_t)m * 4);
    double *sy = sx + m, *sz = sy + m, *sm = sz + m;
    for (int j = 0; j < m; ++j) {
        sx[j] = pos[(long)(j0 + j) * d];
        sy[j] = pos[(long)(j0 + j) * d + 1];
        sz[j] = (d > 2) ? pos[(long)(j0 + j) * d + 2] : 0.0;
        sm[j] = mass[j0 + j];
    }
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        const double xi = pos[(long)i * d], yi = pos[(long)i * d + 1];
        const double zi = (d > 2) ? pos[(long)i * d + 2] : 0.0;
        double ax = 0, ay = 0, az = 0;
        if (d == 2) {
            #pragma omp simd reduction(+:ax,ay)
            for (int j = 0; j < m; ++j) {
                double dx = sx[j] - xi, dy = sy[j] - yi;
                double r2 = (dx * dx + dy * dy) + eps2;
                double w = ((1.0 / (r2 * sqrt(r2))) * G) * sm[j];
                w = (j + j0 == i) ? 0.0 : w;
                ax += w * dx; ay += w * dy;
            }
        } else {
            #pragma omp simd reduction(+:ax,ay,az)
            for (int j = 0; j < m; ++j) {
                double dx = sx[j] - xi, dy = sy[j] - yi, dz = sz[j] - zi;
                double r2 = ((dx * dx + dy * dy) + dz * dz) + eps2;
                double w = ((1.0 / (r2 * sqrt(r2))) * G) * sm[j];
                w = (j + j0 == i) ? 0.0 : w;
                ax += w * dx; ay += w * dy; az += w * dz;
            }
        }
        acc_out[(long)i * d] = ax; acc_out[(long)i * d + 1] = ay;
        if (d > 2) acc_out[(long)i * d + 2] = az;
    }
    free(sx);
}

NBO_CLONES
void nbo_accelerations_f32_fast(int n, int d, const float *pos, const float *mass, float G,
                                float eps2, int j0, int j1, float *acc_out)
{
    /* FLOAT32 mode, fp32 state: per-pair float arithmetic, double accumulation */
    int m = j1 - j0;
    float *sx = (float *)malloc(sizeof(float) * (size_t)m * 4);
    float *sy = sx + m, *sz = sy + m, *sm = sz + m;
    for (int j = 0; j < m; ++j) {
        sx[j] = pos[(long)(j0 + j) * d];
        sy[j] = pos[(long)(j0 + j) * d + 1];
        sz[j] = (d > 2) ? pos[(long)(j0 + j) * d + 2] : 0.0f;
        sm[j] = mass[j0 + j];
    }
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        const float xi = pos[(long)i * d], yi = pos[(long)i * d + 1];
        const float zi = (d > 2) ? pos[(long)i * d + 2] : 0.0f;
        double ax = 0, ay = 0, az = 0;
        #pragma omp simd reduction(+:ax,ay,az)
        for (int j = 0; j < m; ++j) {
            float dx = sx[j] - xi, dy = sy[j] - yi, dz = sz[j] - zi;
            float r2 = dx * dx + dy * dy;
            if (d > 2) r2 = r2 + dz * dz;
            r2 = r2 + eps2;
            float w = ((1.0f / (r2 * sqrtf(r2))) * G) * sm[j];
            w = (j + j0 == i) ? 0.0f : w;
            ax += (double)(w * dx); ay += (double)(w * dy); az += (double)(w * dz);
        }
        acc_out[(long)i * d] = (float)ax; acc_out[(long)i * d + 1] = (float)ay;
        if (d > 2) acc_out[(long)i * d + 2] = (float)az;
    }
    free(sx);
}

/* FLOAT32 mode forces for a SUBSET of targets [i0, i1) against all sources: lets the tests check
 * N = 1 048 576 (BASELINE config 5 size) on a few thousand particles in seconds. */
NBO_CLONES
void nbo_accelerations_f32_fast_sub(int n, int d, const float *pos, const float *mass, float G, float eps2,
                                    int i0, int i1, float *acc_out)
{
    float *sx = (float *)malloc(sizeof(float) * (size_t)n * 4);
    float *sy = sx + n, *sz = sy + n, *sm = sz + n;
    for (int j = 0; j < n; ++j) {
        sx[j] = pos[(long)j * d];
        sy[j] = pos[(long)j * d + 1];
        sz[j] = (d > 2) ? pos[(long)j * d + 2] : 0.0f;
        sm[j] = mass[j];
    }
    #pragma omp parallel for schedule(static)
    for (int i = i0; i < i1; ++i) {
        const float xi = sx[i], yi = sy[i], zi = sz[i];
        double ax = 0, ay = 0, az = 0;
        #pragma omp simd reduction(+:ax,ay,az)
        for (int j = 0; j < n; ++j) {
            float dx = sx[j] - xi, dy = sy[j] - yi, dz = sz[j] - zi;
            float r2 = dx * dx + dy * dy;
            if (d > 2) r2 = r2 + dz * dz;
            r2 = r2 + eps2;
            float w = ((1.0f / (r2 * sqrtf(r2))) * G) * sm[j];
            w = (j == i) ? 0.0f : w;
            ax += (double)(w * dx); ay += (double)(w * dy); az += (double)(w * dz);
        }
        float *o = acc_out + (long)(i - i0) * d;
        o[0] = (float)ax; o[1] = (float)ay;
        if (d > 2) o[2] = (float)az;
    }
    free(sx);
}

/* fp64 KDK step on top of the fast force (all-f64 state, FLOAT64 mode) */
void nbo_step_f64_fast(int n, int d, double *pos, double *vel, const double *mass, double *acc,
                       double G, double eps2, double dt, int nsteps)
{
    long nd = (long)n * d;
    double h = dt / 2;
    for (int s = 0; s < nsteps; ++s) {
        for (long i = 0; i < nd; ++i) { vel[i] = vel[i] + acc[i] * h; pos[i] = pos[i] + vel[i] * dt; }
        nbo_accelerations_f64_fast(n, d, pos, mass, G, eps2, 0, n, acc);
        for (long i = 0; i < nd; ++i) vel[i] = vel[i] + acc[i] * h;
    }
}

NBO_CLONES
double nbo_potential_energy_f64_fast(int n, int d, const double *pos, const double *mass, double G,
                                     double eps2)
{
    double total = 0.0;
    #pragma omp parallel for reduction(+:total) schedule(dynamic, 16)
    for (int j = 1; j < n; ++j) {
        double part = 0.0;
        for (int i = 0; i < j; ++i) {
            double s = 0.0;
            for (int k = 0; k < d; ++k) {
                double df = pos[(long)j * d + k] - pos[(long)i * d + k];
                s += df * df;
            }
            part += (mass[i] * mass[j]) / sqrt(s + eps2);
        }
        total += part;
    }
    return -G * total;
}

int nbo_num_threads(void)
{
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
