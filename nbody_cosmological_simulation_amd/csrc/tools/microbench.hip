// microbench.hip -- instruction-rate and accuracy probes that size the force kernel's budget
// on gfx950 (DESIGN.md "instruction budget").  Not part of the library; built by
// csrc/Makefile target `tools`, run on the GPU box:  ./microbench.bin > gpurun_out/microbench.txt
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e = (x);                                                             \
        if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } \
    } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 8;

// Each op kernel keeps UNROLL independent chains per lane so issue rate, not latency, is measured.
enum { OP_FMA64, OP_MUL64, OP_ADD64, OP_RSQ64, OP_RCP64, OP_SQRT64, OP_CVT64_32, OP_CVT32_64, OP_FMA32, OP_RSQ32,
       OP_MFMA64_4, OP_MFMA64_4_FMA6, OP_FMA64_X6, OP_PKFMA32, OP_PKFMA32_RSQ, OP_MIX_PAIR, OP_NOPS };
const char *OP_NAMES[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rsq_f64", "v_rcp_f64", "v_sqrt_f64",
                          "v_cvt_f32_f64", "v_cvt_f64_f32", "v_fma_f32", "v_rsq_f32",
                          "mfma_f64_4x4x4", "mfma4x4x4+6fma", "6 x v_fma_f64", "v_pk_fma_f32", "pk_fma+2rsq_f32",
                          "pair_body_f64"};

template <int OP>
__global__ void __launch_bounds__(256) op_kernel(double *out, double seed)
{
    double a[UNROLL], g[UNROLL][6];
    float f[UNROLL], h[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        a[u] = seed + threadIdx.x * 1e-3 + u;
        f[u] = (float)a[u];
        h[u] = f[u] + 0.5f;
#pragma unroll
        for (int k = 0; k < 6; ++k) g[u][k] = a[u] + k;
    }
    const double b = 1.0000001, c = 1e-9;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (OP == OP_FMA64) a[u] = __builtin_fma(a[u], b, c);
            if (OP == OP_MUL64) a[u] = a[u] * b;
            if (OP == OP_ADD64) a[u] = a[u] + c;
            if (OP == OP_RSQ64) a[u] = __builtin_amdgcn_rsq(a[u]);
            if (OP == OP_RCP64) a[u] = __builtin_amdgcn_rcp(a[u]);
            if (OP == OP_SQRT64) a[u] = __builtin_amdgcn_sqrt(a[u]);
            if (OP == OP_CVT64_32) { f[u] = (float)a[u]; asm volatile("" : "+v"(f[u])); a[u] += 0; }
            if (OP == OP_CVT32_64) { a[u] = (double)f[u]; asm volatile("" : "+v"(a[u])); }
            if (OP == OP_FMA32) f[u] = __builtin_fmaf(f[u], 1.0000001f, 1e-9f);
            if (OP == OP_RSQ32) f[u] = __builtin_amdgcn_rsqf(f[u]);
            if (OP == OP_PKFMA32 || OP == OP_PKFMA32_RSQ) {
                typedef float f2v __attribute__((ext_vector_type(2)));
                f2v t = {f[u], h[u]};
                t = __builtin_elementwise_fma(t, f2v{1.0000001f, 0.9999999f}, f2v{1e-9f, 2e-9f});
                if (OP == OP_PKFMA32_RSQ) { t.x = __builtin_amdgcn_rsqf(t.x); t.y = __builtin_amdgcn_rsqf(t.y); }
                f[u] = t.x; h[u] = t.y;
            }
            // matrix pipe: does v_mfma_f64_4x4x4_4b overlap with VALU FMAs of the same and of other waves?
            if (OP == OP_MFMA64_4 || OP == OP_MFMA64_4_FMA6) a[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(b, c, a[u], 0, 0, 0);
            if (OP == OP_MFMA64_4_FMA6 || OP == OP_FMA64_X6) {
#pragma unroll
                for (int k = 0; k < 6; ++k) g[u][k] = __builtin_fma(g[u][k], b, c);
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        s += a[u] + f[u] + h[u];
        if (OP == OP_MFMA64_4_FMA6 || OP == OP_FMA64_X6)
#pragma unroll
            for (int k = 0; k < 6; ++k) s += g[u][k];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the fp64 pair body of nb_force.hip (13 VALU + 1 rsq per pair), operands in registers
__global__ void __launch_bounds__(256) pair_kernel(double *out, double seed)
{
    double xi = seed + threadIdx.x * 1e-3, yi = seed * 0.5 + threadIdx.x * 2e-3;
    double ax = 0, ay = 0;
    double xj = 0.25, yj = 0.75;
    const double eps2 = 0.01, gm = 1e-3;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const double dx = xj - xi, dy = yj - yi;
            double q = __builtin_fma(dy, dy, eps2);
            q = __builtin_fma(dx, dx, q);
            const double y0 = __builtin_amdgcn_rsq(q);
            const double y02 = y0 * y0;
            const double e = __builtin_fma(-q, y02, 1.0);
            const double uu = y0 * gm;
            const double v = uu * y02;
            const double cc = __builtin_fma(e, 1.875, 1.5);
            const double ce = cc * e;
            const double w = __builtin_fma(v, ce, v);
            ax = __builtin_fma(w, dx, ax);
            ay = __builtin_fma(w, dy, ay);
            xj += 1e-3; yj -= 1e-3;       // 2 extra adds per pair: subtracted in the report
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = ax + ay;
}

__global__ void rsq_accuracy_kernel(const double *x, double *y_rsq, double *y_r3, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double q = x[i];
    y_rsq[i] = __builtin_amdgcn_rsq(q);
    const double y0 = y_rsq[i];
    const double y02 = y0 * y0;
    const double e = __builtin_fma(-q, y02, 1.0);
    const double v = y0 * y02;
    const double cc = __builtin_fma(e, 1.875, 1.5);
    y_r3[i] = __builtin_fma(v, cc * e, v);
}

template <typename K>
double time_kernel(K launch, int reps)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clk_ghz = prop.clockRate * 1e-6;
    printf("device %s  CUs %d  clock %.3f GHz\n", prop.name, cus, clk_ghz);
    double *out;
    const int blocks_per_cu[] = {1, 2, 4};
    CHECK(hipMalloc(&out, sizeof(double) * 256 * cus * 8));

    // ~150 ms of fp64 FMAs first: the chip needs tens of ms to reach its sustained clock
    for (int r = 0; r < 300; ++r) hipLaunchKernelGGL(op_kernel<OP_FMA64>, dim3(cus * 4), dim3(256), 0, 0, out, 1.5);
    CHECK(hipDeviceSynchronize());

    printf("\n# issue rate: cycles per wave64 instruction per SIMD (at nominal clock), by waves/SIMD\n");
    printf("%-16s %10s %10s %10s\n", "op", "1 wave", "2 waves", "4 waves");
    for (int op = 0; op <= OP_MIX_PAIR; ++op) {
        printf("%-16s", OP_NAMES[op]);
        for (int bpc : blocks_per_cu) {
            const int grid = cus * bpc;   // 256 threads = 4 waves = 1 wave per SIMD per block
            auto launch = [&]() {
                switch (op) {
#define CASE(O) case O: hipLaunchKernelGGL(op_kernel<O>, dim3(grid), dim3(256), 0, 0, out, 1.5); break;
                    CASE(OP_FMA64) CASE(OP_MUL64) CASE(OP_ADD64) CASE(OP_RSQ64) CASE(OP_RCP64) CASE(OP_SQRT64)
                    CASE(OP_CVT64_32) CASE(OP_CVT32_64) CASE(OP_FMA32) CASE(OP_RSQ32)
                    CASE(OP_MFMA64_4) CASE(OP_MFMA64_4_FMA6) CASE(OP_FMA64_X6) CASE(OP_PKFMA32) CASE(OP_PKFMA32_RSQ)
#undef CASE
                case OP_MIX_PAIR: hipLaunchKernelGGL(pair_kernel, dim3(grid), dim3(256), 0, 0, out, 1.5); break;
                }
            };
            const double ms = time_kernel(launch, 5);
            const double instr_per_simd = (double)ITERS * UNROLL * bpc;   // wave-instructions per SIMD
            const double cyc = ms * 1e-3 * clk_ghz * 1e9 / instr_per_simd;
            printf(" %10.2f", cyc);
        }
        if (op == OP_MIX_PAIR) printf("   <- cycles per PAIR-instruction group (13 VALU + rsq + 2 adds)");
        printf("\n");
    }

    // accuracy of v_rsq_f64 and of the corrected q^-1.5
    const int n = 1 << 20;
    std::vector<double> hx(n), hr(n), h3(n);
    srand(1);
    for (int i = 0; i < n; ++i) hx[i] = exp((rand() / (double)RAND_MAX) * 40.0 - 20.0);
    double *dx, *dr, *d3;
    CHECK(hipMalloc(&dx, n * 8)); CHECK(hipMalloc(&dr, n * 8)); CHECK(hipMalloc(&d3, n * 8));
    CHECK(hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rsq_accuracy_kernel, dim3(n / 256), dim3(256), 0, 0, dx, dr, d3, n);
    CHECK(hipMemcpy(hr.data(), dr, n * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h3.data(), d3, n * 8, hipMemcpyDeviceToHost));
    long double e1 = 0, e3 = 0;
    for (int i = 0; i < n; ++i) {
        const long double t1 = 1.0L / sqrtl((long double)hx[i]);
        const long double t3 = t1 * t1 * t1;
        e1 = fmaxl(e1, fabsl((hr[i] - t1) / t1));
        e3 = fmaxl(e3, fabsl((h3[i] - t3) / t3));
    }
    printf("\n# accuracy over %d log-uniform samples in [e^-20, e^20]\n", n);
    printf("v_rsq_f64 max rel err      = %.3Le  (2^%.1Lf)\n", e1, log2l(e1));
    printf("corrected q^-1.5 max rel err = %.3Le  (%.2Lf ulp of fp64)\n", e3, e3 / 1.1102230246251565e-16L);
    return 0;
}
