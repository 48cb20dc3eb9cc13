// Microbenchmark: do f64 VALU ops overlap with an in-flight v_mfma_f64_16x16x4_f64 of the same wave?
// One wave; a dependent MFMA chain with NV independent VALU ops issued behind every MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)

// KIND 0: v_fma_f64, 1: v_fma_f32, 2: v_add_u32, 3: ds_read_b64 (LDS)
template <int NV, int KIND>
__global__ void k(double *out, unsigned long long *cyc, int iters, double seed)
{
    __shared__ double sm[1024];
    const int l = threadIdx.x;
    for (int i = l; i < 1024; i += 64) sm[i] = 1e-3 * i;
    __syncthreads();
    double a = 1e-3 * (l + 1), b = 2e-3 * (l + 2);
    d4 acc = {seed, seed, seed, seed};
    double x[8]; float xf[8]; int xi[8];
    for (int i = 0; i < 8; ++i) { x[i] = seed + i; xf[i] = (float)seed + i; xi[i] = l + i; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            acc = MF(a, b, acc);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (KIND == 0) x[v & 7] = __builtin_fma(x[v & 7], 0.999, 1e-3);
                if (KIND == 1) xf[v & 7] = __builtin_fmaf(xf[v & 7], 0.999f, 1e-3f);
                if (KIND == 2) xi[v & 7] = xi[v & 7] * 3 + 1;
                if (KIND == 3) x[v & 7] += sm[(xi[v & 7] + 67 * v + it) & 1023];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = acc[0] + acc[1] + acc[2] + acc[3];
    for (int i = 0; i < 8; ++i) s += x[i] + xf[i] + xi[i];
    out[l] = s;
    if (l == 0) cyc[0] = t1 - t0;
}
#define RUN(name, NV, KIND)                                                                       \
    do {                                                                                          \
        k<NV, KIND><<<1, 64>>>(o, cyc, iters, 1.0); hipDeviceSynchronize();                       \
        k<NV, KIND><<<1, 64>>>(o, cyc, iters, 1.0); hipDeviceSynchronize();                       \
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);                       \
        printf("%-60s %8.1f cycles per MFMA\n", name, (double)c / iters / 4);                     \
    } while (0)
int main()
{
    double *o; unsigned long long *cyc; const int iters = 2000;
    (void)hipMalloc(&o, 64 * 8); (void)hipMalloc(&cyc, 8);
    RUN("dependent f64 MFMA chain, nothing else", 0, 0);
    RUN("+ 4 independent v_fma_f64 behind each MFMA", 4, 0);
    RUN("+ 8 independent v_fma_f64 behind each MFMA", 8, 0);
    RUN("+ 16 independent v_fma_f64 behind each MFMA", 16, 0);
    RUN("+ 8 independent v_fma_f32 behind each MFMA", 8, 1);
    RUN("+ 16 independent v_fma_f32 behind each MFMA", 16, 1);
    RUN("+ 8 independent int mad behind each MFMA", 8, 2);
    RUN("+ 16 independent int mad behind each MFMA", 16, 2);
    RUN("+ 4 ds_read_b64 behind each MFMA", 4, 3);
    RUN("+ 8 ds_read_b64 behind each MFMA", 8, 3);
    return 0;
}
