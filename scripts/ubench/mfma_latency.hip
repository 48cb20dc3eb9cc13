// Microbenchmark: issue interval vs dependent latency of the matrix instructions the kernels use (one wave).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k_f64(double *out, unsigned long long *cyc, int iters)
{
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5;
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// dependent through the B operand (result feeds next instruction's input, as in W -> H)
__global__ void k_f64_chainB(double *out, unsigned long long *cyc, int iters)
{
    double a = 1e-3;
    d4 acc = {1, 1, 1, 1};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[0], (d4){0, 0, 0, 0}, 0, 0, 0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc[0];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC>
__global__ void k_f32_16(float *out, unsigned long long *cyc, int iters)
{
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f4){0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC>
__global__ void k_f32_32(float *out, unsigned long long *cyc, int iters)
{
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
    f16v acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// plain f64 FMA chains for comparison
template <int NACC>
__global__ void k_fma64(double *out, unsigned long long *cyc, int iters)
{
    double a = 1.0 + threadIdx.x * 1e-9, acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, 1e-9);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_rcp64(double *out, unsigned long long *cyc, int iters)
{
    double x = 1.5 + threadIdx.x * 1e-9;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) x = 1.0 / x + 0.25;
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_rcpfast64(double *out, unsigned long long *cyc, int iters)
{
    double x = 1.5 + threadIdx.x * 1e-9;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        double r = __builtin_amdgcn_rcp(x);
        r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
        r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
        x = r + 0.25;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// accuracy of the v_rcp_f64 seed and of 1 / 2 Newton steps against the IEEE quotient
__global__ void k_rcp_acc(double *out)
{
    double worst0 = 0, worst1 = 0, worst2 = 0;
    unsigned long long st = 0x9E3779B97F4A7C15ull * (threadIdx.x + 1);
    for (int it = 0; it < 20000; ++it) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        double x = 1e-3 + (double)(st >> 11) * (1.0 / 9007199254740992.0) * 1e6;
        double ref = 1.0 / x;
        double r0 = __builtin_amdgcn_rcp(x);
        double r1 = __builtin_fma(__builtin_fma(-x, r0, 1.0), r0, r0);
        double r2 = __builtin_fma(__builtin_fma(-x, r1, 1.0), r1, r1);
        worst0 = fmax(worst0, fabs(r0 - ref) / ref);
        worst1 = fmax(worst1, fabs(r1 - ref) / ref);
        worst2 = fmax(worst2, fabs(r2 - ref) / ref);
    }
    out[threadIdx.x * 3] = worst0; out[threadIdx.x * 3 + 1] = worst1; out[threadIdx.x * 3 + 2] = worst2;
}
#define RUN(name, launch, n)                                                          \
    do {                                                                              \
        launch; hipDeviceSynchronize(); launch; hipDeviceSynchronize();               \
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);           \
        printf("%-34s %8.1f cycles per instruction\n", name, (double)c / (iters * (n))); \
    } while (0)
int main()
{
    double *o; float *of; unsigned long long *cyc; const int iters = 2000;
    hipMalloc(&o, 64 * 8); hipMalloc(&of, 64 * 4); hipMalloc(&cyc, 8);
    RUN("mfma_f64_16x16x4 1 acc (dependent)", (k_f64<1><<<1, 64>>>(o, cyc, iters)), 1);
    RUN("mfma_f64_16x16x4 2 acc", (k_f64<2><<<1, 64>>>(o, cyc, iters)), 2);
    RUN("mfma_f64_16x16x4 4 acc", (k_f64<4><<<1, 64>>>(o, cyc, iters)), 4);
    RUN("mfma_f64_16x16x4 chain via B", (k_f64_chainB<<<1, 64>>>(o, cyc, iters)), 1);
    RUN("mfma_f32_16x16x4 1 acc (dependent)", (k_f32_16<1><<<1, 64>>>(of, cyc, iters)), 1);
    RUN("mfma_f32_16x16x4 4 acc", (k_f32_16<4><<<1, 64>>>(of, cyc, iters)), 4);
    RUN("mfma_f32_32x32x2 1 acc (dependent)", (k_f32_32<1><<<1, 64>>>(of, cyc, iters)), 1);
    RUN("mfma_f32_32x32x2 4 acc", (k_f32_32<4><<<1, 64>>>(of, cyc, iters)), 4);
    RUN("v_fma_f64 1 chain (dependent)", (k_fma64<1><<<1, 64>>>(o, cyc, iters)), 1);
    RUN("v_fma_f64 8 chains", (k_fma64<8><<<1, 64>>>(o, cyc, iters)), 8);
    RUN("f64 divide (IEEE) dependent", (k_rcp64<<<1, 64>>>(o, cyc, iters)), 1);
    RUN("f64 rcp + 2 Newton dependent", (k_rcpfast64<<<1, 64>>>(o, cyc, iters)), 1);
    double *acc; hipMalloc(&acc, 64 * 3 * 8);
    k_rcp_acc<<<1, 64>>>(acc);
    double h[192]; hipMemcpy(h, acc, sizeof(h), hipMemcpyDeviceToHost);
    double w0 = 0, w1 = 0, w2 = 0;
    for (int i = 0; i < 64; ++i) { w0 = fmax(w0, h[3 * i]); w1 = fmax(w1, h[3 * i + 1]); w2 = fmax(w2, h[3 * i + 2]); }
    printf("v_rcp_f64 max rel err: seed %.3g, +1 Newton %.3g, +2 Newton %.3g\n", w0, w1, w2);
    return 0;
}
