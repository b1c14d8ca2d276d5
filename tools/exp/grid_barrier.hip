#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__global__ __launch_bounds__(256) void k_barrier(unsigned* count, unsigned target, const float* in, float* out, int n, int use_barrier) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float v = i < n ? in[i] : 0.f;
    if (use_barrier) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
    if (i < n) out[i] = v * 2.f;
}
int main() {
    unsigned* count; float *in, *out;
    const int maxn = 2048 * 256;
    hipMalloc(&count, 4); hipMemset(count, 0, 4);
    hipMalloc(&in, maxn * 4); hipMalloc(&out, maxn * 4); hipMemset(in, 0, maxn * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512, 800, 1024}) {
        for (int ub = 0; ub < 2; ++ub) {
            unsigned base = 0;
            hipMemset(count, 0, 4);
            const int reps = 200;
            // warm
            for (int r = 0; r < 5; ++r) { base += blocks; hipLaunchKernelGGL(k_barrier, dim3(blocks), dim3(256), 0, 0, count, base, in, out, blocks * 256, ub); }
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < reps; ++r) { base += blocks; hipLaunchKernelGGL(k_barrier, dim3(blocks), dim3(256), 0, 0, count, ub ? base : 0u, in, out, blocks * 256, ub); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("blocks %4d barrier %d: %.2f us per kernel\n", blocks, ub, ms / reps * 1e3);
        }
    }
    return 0;
}
