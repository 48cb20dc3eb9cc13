// Microbenchmark: back-to-back launch interval of a kernel that does nothing, with the control step's launch shape
// (256 workgroups x 256 threads, 152 KB dynamic LDS each): plain stream launches and the same chain replayed as a hipGraph.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_nop(double *p) { extern __shared__ double sm[]; if (p == nullptr && threadIdx.x == 9999) sm[0] = 1.0; }
int main()
{
    hipStream_t s; (void)hipStreamCreate(&s);
    (void)hipFuncSetAttribute((const void *)k_nop, hipFuncAttributeMaxDynamicSharedMemorySize, 152576);
    for (int shm : {0, 152576}) {
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_nop, dim3(256), dim3(256), shm, s, (double *)nullptr);
        (void)hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        const int n = 2000;
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_nop, dim3(256), dim3(256), shm, s, (double *)nullptr);
        (void)hipStreamSynchronize(s);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
        printf("empty kernel, 256 x 256 threads, %6d B LDS per workgroup: %.2f us per back-to-back launch\n", shm, us);
        // the same chain as a graph of 200 kernel nodes, replayed 10 times
        hipGraph_t g; hipGraphExec_t ge;
        (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_nop, dim3(256), dim3(256), shm, s, (double *)nullptr);
        (void)hipStreamEndCapture(s, &g);
        (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 10; ++r) (void)hipGraphLaunch(ge, s);
        (void)hipStreamSynchronize(s);
        us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 2000;
        printf("   as a hipGraph of 200 dependent kernel nodes:                     %.2f us per node\n", us);
    }
    return 0;
}
