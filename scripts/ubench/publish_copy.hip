// publish_copy.hip -- what a "copy n bytes, then make them visible system-wide and flag it" launch costs, three ways.
//   v1: system-scope (write-through) stores, one system fence per block, last block flags
//   v2: plain stores, one agent-scope release per block (on a multi-XCD part that writes the XCD's L2 back), last block flags
//   v3: plain copy kernel (visibility = the kernel boundary) + a second one-thread launch that flags
// hipcc --offload-arch=gfx950 -O3 -o publish_copy.bin publish_copy.hip && ./publish_copy.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned long long u64;
template <int V>
__global__ __launch_bounds__(256) void pub(const double2 *src, double2 *dst, size_t n2, unsigned *done, u64 *flag, u64 t)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
        const double2 v = src[i];
        if (V == 1) {
            u64 *dw = reinterpret_cast<u64 *>(dst);
            __hip_atomic_store(dw + 2 * i, (u64)__double_as_longlong(v.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(dw + 2 * i + 1, (u64)__double_as_longlong(v.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            dst[i] = v;
        }
    }
    if (V == 3) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (V == 1) __threadfence_system();
        const unsigned prev = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1) {
            __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(flag, t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ void flag_only(u64 *flag, u64 t) { __hip_atomic_store(flag, t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

int main()
{
    const size_t sizes[] = {1024ull * 210 * 8, 12288ull * 210 * 8};
    for (size_t bytes : sizes) {
        double2 *src, *dst; unsigned *done; u64 *flag;
        hipMalloc(&src, bytes); hipMalloc(&dst, bytes); hipMalloc(&done, 256); hipMalloc(&flag, 256);
        hipMemset(src, 1, bytes); hipMemset(done, 0, 256); hipMemset(flag, 0, 256);
        const size_t n2 = bytes / 16;
        for (int blocks_cap : {256, 1024, 4096}) {
            size_t blocks = (n2 + 255) / 256; if (blocks > (size_t)blocks_cap) blocks = blocks_cap;
            for (int v = 1; v <= 4; ++v) {
                hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
                const int reps = 50;
                for (int r = 0; r < reps + 5; ++r) {
                    if (r == 5) hipEventRecord(a, 0);
                    if (v == 1) hipLaunchKernelGGL(pub<1>, dim3(blocks), dim3(256), 0, 0, src, dst, n2, done, flag, (u64)r);
                    if (v == 2) hipLaunchKernelGGL(pub<2>, dim3(blocks), dim3(256), 0, 0, src, dst, n2, done, flag, (u64)r);
                    if (v == 3) { hipLaunchKernelGGL(pub<3>, dim3(blocks), dim3(256), 0, 0, src, dst, n2, done, flag, (u64)r);
                                  hipLaunchKernelGGL(flag_only, dim3(1), dim3(1), 0, 0, flag, (u64)r); }
                    if (v == 4) { hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0);
                                  hipLaunchKernelGGL(flag_only, dim3(1), dim3(1), 0, 0, flag, (u64)r); }
                }
                hipEventRecord(b, 0); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                printf("%8.2f MB blocks<=%4d v%d: %7.2f us per publish (%6.1f GB/s)\n", bytes / 1e6, blocks_cap, v, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e9);
            }
        }
        hipFree(src); hipFree(dst); hipFree(done); hipFree(flag);
    }
    return 0;
}
