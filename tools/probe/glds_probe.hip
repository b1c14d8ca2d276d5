// Probe: semantics of buffer_load_dwordx4 ... lds (LDS-DMA) on gfx950 -- lane-linear LDS placement, zero fill for out-of-range
// lanes, M0 base handling by the builtin.  hipcc --offload-arch=gfx950 -O3 glds_probe.hip -o glds_probe && ./glds_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const uint32_t* __restrict__ src, uint32_t src_bytes, uint32_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[2 * 64 * 4 + 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * 64 * 4 + 64; i += blockDim.x) lds[i] = 0xDEADBEEFu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, src_bytes, 0x00020000);
    // lane l fetches 16 bytes at a PERMUTED source offset (reverse order); odd lanes of wave 1 go out of range
    uint32_t off = (uint32_t)(63 - lane) * 16u + (uint32_t)wave * 1024u;
    if (wave == 1 && (lane & 1)) off = 0xFFFFFFF0u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + wave * 256), 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * 64 * 4 + 64; i += blockDim.x) out[i] = lds[i];
}

int main() {
    const int n = 2 * 64 * 4;
    std::vector<uint32_t> h(n);
    for (int i = 0; i < n; ++i) h[i] = 1000 + i;
    uint32_t *d, *o;
    hipMalloc(&d, n * 4);
    hipMalloc(&o, (n + 64) * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(128), 0, 0, d, (uint32_t)(n * 4), o);
    std::vector<uint32_t> r(n + 64);
    hipMemcpy(r.data(), o, (n + 64) * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 2; ++w)
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 4; ++e) {
                const uint32_t got = r[w * 256 + l * 4 + e];
                uint32_t want = 1000 + w * 256 + (63 - l) * 4 + e;
                if (w == 1 && (l & 1)) want = 0;
                if (got != want) { if (bad < 8) printf("wave %d lane %d e %d: got %u (0x%x) want %u\n", w, l, e, got, got, want); ++bad; }
            }
    for (int i = n; i < n + 64; ++i) if (r[i] != 0xDEADBEEFu) { printf("guard word %d overwritten: 0x%x\n", i, r[i]); ++bad; }
    printf(bad ? "PROBE FAIL (%d mismatches)\n" : "PROBE OK: lane-linear placement, zero fill for out-of-range lanes\n", bad);
    return bad != 0;
}
