// Microbenchmark: do a wave's own f32 VALU ops overlap with its in-flight v_mfma_f32_32x32x16_f16 (MLP tile) and
// v_mfma_f32_32x32x2_f32?  One wave (optionally WAVES waves in one workgroup, one per SIMD); a dependent MFMA chain with NV
// independent VALU ops issued behind every MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// KIND 0: v_fma_f32 (8 independent chains), 1: v_cvt_pk_f16_f32 + back, 2: v_med3_f32 ; MF 0: 32x32x16 f16, 1: 32x32x2 f32
template <int NV, int KIND, int MF, int NACC>
__global__ void k(float *out, unsigned long long *cyc, int iters, float seed)
{
    const int l = threadIdx.x;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(1e-3f * (l + i)); b[i] = (_Float16)(2e-3f * (l + 2 * i)); }
    f16v acc[NACC];
    for (int n = 0; n < NACC; ++n)
        for (int i = 0; i < 16; ++i) acc[n][i] = seed;
    float xf[8];
    for (int i = 0; i < 8; ++i) xf[i] = seed + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (MF == 0) acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m % NACC], 0, 0, 0);
            else acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a[0], (float)b[0], acc[m % NACC], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (KIND == 0) xf[v & 7] = __builtin_fmaf(xf[v & 7], 0.999f, 1e-3f);
                if (KIND == 1) { _Float16 hh = (_Float16)xf[v & 7]; xf[v & 7] = xf[v & 7] - (float)hh + 1.0f; }
                if (KIND == 2) xf[v & 7] = __builtin_amdgcn_fmed3f(xf[v & 7] + 1.0f, 0.0f, 65000.0f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
    for (int i = 0; i < 8; ++i) s += xf[i];
    out[blockIdx.x * blockDim.x + l] = s;
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
#define RUN(name, NV, KIND, MF, NACC, BLK)                                                        \
    do {                                                                                          \
        k<NV, KIND, MF, NACC><<<1, BLK>>>(o, cyc, iters, 1.0f); (void)hipDeviceSynchronize();           \
        k<NV, KIND, MF, NACC><<<1, BLK>>>(o, cyc, iters, 1.0f); (void)hipDeviceSynchronize();           \
        unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);                 \
        printf("%-78s %8.1f cycles per MFMA\n", name, (double)c / iters / 4);                     \
    } while (0)
int main()
{
    float *o; unsigned long long *cyc; const int iters = 2000;
    (void)hipMalloc(&o, 1024 * 4); (void)hipMalloc(&cyc, 8);
    RUN("f16 32x32x16 dependent chain (1 acc), nothing else", 0, 0, 0, 1, 64);
    RUN("f16 32x32x16, 4 accumulators, nothing else", 0, 0, 0, 4, 64);
    RUN("f16 32x32x16 1 acc + 2 v_fma_f32 behind each", 2, 0, 0, 1, 64);
    RUN("f16 32x32x16 1 acc + 4 v_fma_f32 behind each", 4, 0, 0, 1, 64);
    RUN("f16 32x32x16 1 acc + 6 v_fma_f32 behind each", 6, 0, 0, 1, 64);
    RUN("f16 32x32x16 1 acc + 8 v_fma_f32 behind each", 8, 0, 0, 1, 64);
    RUN("f16 32x32x16 1 acc + 16 v_fma_f32 behind each", 16, 0, 0, 1, 64);
    RUN("f16 32x32x16 4 acc + 4 v_fma_f32 behind each", 4, 0, 0, 4, 64);
    RUN("f16 32x32x16 4 acc + 8 v_fma_f32 behind each", 8, 0, 0, 4, 64);
    RUN("f16 32x32x16 1 acc + 4 (cvt f16 + cvt back + sub + add) behind each", 4, 1, 0, 1, 64);
    RUN("f16 32x32x16 1 acc + 4 (add + med3) behind each", 4, 2, 0, 1, 64);
    RUN("f16 32x32x16 1 acc + 4 v_fma_f32, 4 waves in the workgroup (one per SIMD)", 4, 0, 0, 1, 256);
    RUN("f32 32x32x2 dependent chain, nothing else", 0, 0, 1, 1, 64);
    RUN("f32 32x32x2 1 acc + 4 v_fma_f32 behind each", 4, 0, 1, 1, 64);
    RUN("f32 32x32x2 1 acc + 8 v_fma_f32 behind each", 8, 0, 1, 1, 64);
    RUN("f32 32x32x2 1 acc + 16 v_fma_f32 behind each", 16, 0, 1, 1, 64);
    return 0;
}
