// Microbenchmark: register layout and timing of v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 products per instruction)
// and of the row broadcast (DPP row_newbcast) that chains its result into the next instruction's B operand.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_f64_4x4x4.hip -o /tmp/mfma444 && /tmp/mfma444
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

// layout: A one-hot in lane la, B = 1 + lane; D[ld] != 0 names the (la, lb = D - 1, ld) triples of the product
__global__ void k_layout(double *out)
{
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la) {
        double a = lane == la ? 1.0 : 0.0, b = 1.0 + lane;
        double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
        out[la * 64 + lane] = d;
    }
}

template <int ROW>
__device__ inline double row_bcast(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + ROW, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + ROW, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__global__ void k_bcast(double *out)
{
    const int lane = threadIdx.x;
    double v = 100.0 + lane;
    out[lane] = row_bcast<0>(v);
    out[64 + lane] = row_bcast<4>(v);
    out[128 + lane] = row_bcast<8>(v);
    out[192 + lane] = row_bcast<12>(v);
}

// timing: NACC independent accumulators, dependent through C
template <int NACC>
__global__ void k_time_c(double *out, unsigned long long *cyc, int iters)
{
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5, acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// dependent through the B operand
__global__ void k_time_b(double *out, unsigned long long *cyc, int iters)
{
    double a = 1e-3, acc = 1.0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, acc, 0.0, 0, 0, 0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// the forward-sweep chain: 3 accumulating products, broadcast of quad 3, a 4th product, three broadcasts -> next stage
__global__ void k_time_stage(double *out, unsigned long long *cyc, int iters)
{
    double a0 = 1e-3 + threadIdx.x * 1e-6, a1 = 2e-3, a2 = 3e-3, a3 = 1e-3;
    double z0 = 1.0, z1 = 0.5, z2 = 0.25;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        double y = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, z0, 0.0, 0, 0, 0);
        y = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, z1, y, 0, 0, 0);
        y = __builtin_amdgcn_mfma_f64_4x4x4f64(a2, z2, y, 0, 0, 0);
        double du = row_bcast<12>(y);
        y = __builtin_amdgcn_mfma_f64_4x4x4f64(a3, du, y, 0, 0, 0);
        z0 = row_bcast<0>(y); z1 = row_bcast<4>(y); z2 = row_bcast<8>(y);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = z0 + z1 + z2;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// the same stage on the 16x16x4 instruction (what the sweeps do today): 3 + 1 products, results chain as they lie
__global__ void k_time_stage16(double *out, unsigned long long *cyc, int iters)
{
    double a0 = 1e-3 + threadIdx.x * 1e-6, a1 = 2e-3, a2 = 3e-3, a3 = 1e-3;
    double z0 = 1.0, z1 = 0.5, z2 = 0.25;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        d4 y = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, z0, (d4){0, 0, 0, 0}, 0, 0, 0);
        y = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, z1, y, 0, 0, 0);
        y = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, z2, y, 0, 0, 0);
        y = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, y[3], y, 0, 0, 0);
        z0 = y[0]; z1 = y[1]; z2 = y[2];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = z0 + z1 + z2;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    double *out; unsigned long long *cyc;
    hipMalloc(&out, 64 * 64 * 8); hipMalloc(&cyc, 8);
    std::vector<double> h(64 * 64);
    k_layout<<<1, 64>>>(out);
    hipMemcpy(h.data(), out, 64 * 64 * 8, hipMemcpyDeviceToHost);
    // hypothesis: A[i][k] of block b in lane i + 4b + 16k, B[k][j] in lane j + 4b + 16k, D[i][j] in lane j + 4b + 16i
    int bad = 0, nz = 0;
    for (int la = 0; la < 64; ++la)
        for (int ld = 0; ld < 64; ++ld) {
            double d = h[la * 64 + ld];
            int i = la & 3, b = (la >> 2) & 3, k = la >> 4;
            bool expect = ((ld >> 2) & 3) == b && (ld >> 4) == i;
            double want = expect ? 1.0 + ((ld & 3) + 4 * b + 16 * k) : 0.0;
            if (d != 0) nz++;
            if (d != want) { if (bad < 12) printf("  la %2d ld %2d: got %g want %g\n", la, ld, d, want); bad++; }
        }
    printf("layout A: i+4b+16k, B: j+4b+16k, D: j+4b+16i : %s (%d non-zeros, %d mismatches)\n", bad ? "NO" : "yes", nz, bad);
    if (bad) {
        printf("raw triples (la -> ld:lb):\n");
        for (int la = 0; la < 64; ++la) {
            printf("la %2d:", la);
            for (int ld = 0; ld < 64; ++ld) if (h[la * 64 + ld] != 0) printf(" %d:%d", ld, (int)h[la * 64 + ld] - 1);
            printf("\n");
        }
    }
    k_bcast<<<1, 64>>>(out);
    hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost);
    int bb = 0;
    for (int q = 0; q < 4; ++q)
        for (int l = 0; l < 64; ++l) if (h[q * 64 + l] != 100.0 + (l & 48) + 4 * q) bb++;
    printf("row_newbcast:4q gives lane (l & 48) + 4q: %s\n", bb ? "NO" : "yes");
    const int iters = 2000;
    unsigned long long c;
#define RUN(name, kern, per) kern<<<1, 64>>>(out, cyc, iters); hipDeviceSynchronize(); kern<<<1, 64>>>(out, cyc, iters); \
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-58s %7.1f cycles\n", name, (double)c / iters / (per));
    RUN("4x4x4 dependent through C, per instruction", k_time_c<1>, 1)
    RUN("4x4x4 2 accumulators, per instruction", k_time_c<2>, 2)
    RUN("4x4x4 4 accumulators, per instruction", k_time_c<4>, 4)
    RUN("4x4x4 dependent through B, per instruction", k_time_b, 1)
    RUN("forward stage on 4x4x4 (4 products + 4 row broadcasts)", k_time_stage, 1)
    RUN("forward stage on 16x16x4 (4 products)", k_time_stage16, 1)
    return 0;
}
