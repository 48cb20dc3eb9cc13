// Microbenchmark: the MFMA dependency pattern of one backward Riccati stage (one wave), cycles per stage.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)

// variant 0: Wf(3, A-dep on H) -> Hb(3, B-dep on Wf) -> H = Hb                       (matrix chain only)
// variant 1: ... + G = mfma(l, Wf3) -> gs = G0*k -> Hn = mfma(-Wf3, gs, Hb)          (full MFMA pattern)
// variant 2: variant 1 + a dependent f64 VALU chain of 12 ops between H and G's A operand (the inverse)
template <int VAR>
__global__ void k_stage(double *out, unsigned long long *cyc, int iters, double seed)
{
    const int l = threadIdx.x;
    double m0 = 1e-3 * (l + 1), m1 = 2e-3 * (l + 2), m2 = 3e-3 * (l + 3), lin = 1e-2 * (l & 3);
    d4 H = {seed, seed * 0.5, seed * 0.25, seed * 0.125};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        d4 Wf = {0, 0, 0, 0};
        Wf = MF(H[0], m0, Wf); Wf = MF(H[1], m1, Wf); Wf = MF(H[2], m2, Wf);
        double a = lin;
        if (VAR == 2) {
            double x = H[3];
#pragma unroll
            for (int i = 0; i < 12; ++i) x = __builtin_fma(x, 0.999, 1e-3);
            a = x * 1e-3;
        }
        d4 Hb = {1e-3, 1e-3, 1e-3, 1e-3};
        Hb = MF(m0, Wf[0], Hb); Hb = MF(m1, Wf[1], Hb); Hb = MF(m2, Wf[2], Hb);
        if (VAR >= 1) {
            d4 z = {0, 0, 0, 0};
            d4 G = MF(a, Wf[3], z);
            double gs = G[0] * 1e-3;
            Hb = MF(-Wf[3], gs, Hb);
        }
        H = Hb * 1e-3;       // keep magnitudes bounded (one extra VALU op per register)
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[l] = H[0] + H[1] + H[2] + H[3];
    if (l == 0) cyc[0] = t1 - t0;
}
// forward-stage pattern: Y(3, B-dep on z) -> xn = mfma(mu, Y3, Y) -> z = xn
__global__ void k_fwd(double *out, unsigned long long *cyc, int iters, double seed)
{
    const int l = threadIdx.x;
    double f0 = 1e-3 * (l + 1), f1 = 2e-3 * (l + 2), f2 = 3e-3 * (l + 3), mu = 1e-3;
    double z0 = seed, z1 = seed * 0.5, z2 = seed * 0.25;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        d4 Y = {0, 0, 0, 0};
        Y = MF(f0, z0, Y); Y = MF(f1, z1, Y); Y = MF(f2, z2, Y);
        d4 xn = MF(mu, Y[3], Y);
        z0 = xn[0]; z1 = xn[1]; z2 = xn[2];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[l] = z0 + z1 + z2;
    if (l == 0) cyc[0] = t1 - t0;
}
#define RUN(name, launch)                                                             \
    do {                                                                              \
        launch; hipDeviceSynchronize(); launch; hipDeviceSynchronize();               \
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);           \
        printf("%-58s %8.1f cycles per stage\n", name, (double)c / iters);            \
    } while (0)
int main()
{
    double *o; unsigned long long *cyc; const int iters = 2000;
    (void)hipMalloc(&o, 64 * 8); (void)hipMalloc(&cyc, 8);
    RUN("backward: Wf(3) -> Hb(3)                      [6 MFMA]", (k_stage<0><<<1, 64>>>(o, cyc, iters, 1.0)));
    RUN("backward: + G -> gs -> rank-4 correction      [8 MFMA]", (k_stage<1><<<1, 64>>>(o, cyc, iters, 1.0)));
    RUN("backward: + 12-op f64 VALU chain feeding G    [8 MFMA]", (k_stage<2><<<1, 64>>>(o, cyc, iters, 1.0)));
    RUN("forward : Y(3) -> xn(1) -> z                  [4 MFMA]", (k_fwd<<<1, 64>>>(o, cyc, iters, 1.0)));
    return 0;
}
