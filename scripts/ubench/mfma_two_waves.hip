// Microbenchmark: two waves on the SAME SIMD (waves 0 and 4 of a 512-thread workgroup): one runs a dependent MFMA
// chain, the other independent VALU work.  Do they overlap, or does the matrix instruction hold the SIMD's VALU?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// MK: 0 = v_mfma_f64_16x16x4_f64, 1 = v_mfma_f32_32x32x16_f16; role A (wave 0) = MFMA chain, role B (wave 4) = VK work
// VK: 0 = v_fma_f64 (8 independent chains), 1 = v_fma_f32, 2 = f64 MFMA chain too
template <int MK, int VK>
__global__ __launch_bounds__(512) void k(double *out, unsigned long long *cyc, int iters, int runA, int runB, double seed)
{
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (wave != 0 && wave != 4) return;
    unsigned long long t0 = 0, t1 = 0;
    double res = 0.0;
    if (wave == 0 && runA) {
        double a = 1e-3 * (l + 1), b = 2e-3 * (l + 2);
        d4 acc = {seed, seed, seed, seed};
        f16v accf; for (int i = 0; i < 16; ++i) accf[i] = (float)seed;
        h8 ha, hb; for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.01f * (l + i)); hb[i] = (_Float16)(0.02f * (l - i)); }
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (MK == 0) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                else accf = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, accf, 0, 0, 0);
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        res = acc[0] + acc[1] + acc[2] + acc[3] + accf[0] + accf[15];
    }
    if (wave == 4 && runB) {
        double x[8]; float xf[8];
        for (int i = 0; i < 8; ++i) { x[i] = seed + i; xf[i] = (float)seed + i; }
        double a = 1e-3 * (l + 1), b = 2e-3 * (l + 2);
        d4 acc = {seed, seed, seed, seed};
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int v = 0; v < 64; ++v) {
                if (VK == 0) x[v & 7] = __builtin_fma(x[v & 7], 0.999, 1e-3);
                if (VK == 1) xf[v & 7] = __builtin_fmaf(xf[v & 7], 0.999f, 1e-3f);
            }
            if (VK == 2) {
#pragma unroll
                for (int m = 0; m < 8; ++m) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 8; ++i) res += x[i] + xf[i];
        res += acc[0];
    }
    out[threadIdx.x] = res;
    if (l == 0) cyc[wave == 0 ? 0 : 1] = t1 - t0;
}
template <int MK, int VK>
void run(const char *name, double *o, unsigned long long *cyc, int iters)
{
    unsigned long long c[2], solo[2];
    for (int mode = 0; mode < 3; ++mode) {
        const int ra = mode != 1, rb = mode != 0;
        for (int rep = 0; rep < 2; ++rep) { k<MK, VK><<<1, 512>>>(o, cyc, iters, ra, rb, 1.0); (void)hipDeviceSynchronize(); }
        (void)hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
        if (mode == 0) solo[0] = c[0];
        if (mode == 1) solo[1] = c[1];
    }
    printf("%-44s A alone %7.1f  B alone %7.1f | together: A %7.1f  B %7.1f   (cycles per iteration: A = 8 MFMAs, B = 64 VALU ops or 8 MFMAs)\n",
           name, (double)solo[0] / iters, (double)solo[1] / iters, (double)c[0] / iters, (double)c[1] / iters);
}
int main()
{
    double *o; unsigned long long *cyc; const int iters = 2000;
    (void)hipMalloc(&o, 512 * 8); (void)hipMalloc(&cyc, 16);
    run<0, 0>("A: f64 MFMA chain   B: v_fma_f64", o, cyc, iters);
    run<0, 1>("A: f64 MFMA chain   B: v_fma_f32", o, cyc, iters);
    run<0, 2>("A: f64 MFMA chain   B: f64 MFMA chain", o, cyc, iters);
    run<1, 0>("A: f16 MFMA chain   B: v_fma_f64", o, cyc, iters);
    run<1, 1>("A: f16 MFMA chain   B: v_fma_f32", o, cyc, iters);
    run<1, 2>("A: f16 MFMA chain   B: f64 MFMA chain", o, cyc, iters);
    return 0;
}
