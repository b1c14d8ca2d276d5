// Does `s_waitcnt vmcnt(K)` still cover an OLDER in-range buffer load when the K younger operations are ALL out of range?
// (Round 4: counted waits in front of register-prefetched operands gave wrong sums on blocks whose younger operations were all
// out-of-range ones.)  Each wave: one cold in-range buffer_load_dwordx4, then K all-out-of-range operations of a given kind, then
// s_waitcnt vmcnt(K), then the loaded value is stored.  A value that differs from memory = the wait let go early.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#ifndef DSN_PROBE_AUX
#define DSN_PROBE_AUX 0     // cache policy of the "hot" younger LDS-DMAs (gfx950: 1 = sc0, 2 = nt, 16 = sc1): does an L1 bypass keep them in order?
#endif
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int K, int KIND>   // KIND 0: younger = OOB register loads through a ZERO-record resource; 1: OOB offsets on the real resource;
                             //      2: OOB LDS-DMA; 3: in-range register loads (control)
__global__ void probe(const u32x4* __restrict__ src, unsigned bytes, u32x4* __restrict__ out, unsigned stride) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[8192];
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0, 0x00020000);
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned off = (tid * stride) % (bytes / 16) * 16;      // scattered: cache-cold lines
    u32x4 a, b[K > 0 ? K : 1];
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(a) : "v"(off), "s"(r) : "memory");
#pragma unroll
    for (int i = 0; i < K; ++i) {
        if (KIND == 0) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(b[i]) : "v"(off), "s"(rz) : "memory");
        else if (KIND == 1) { const unsigned o = 0xFFFFFFF0u; asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(b[i]) : "v"(o), "s"(r) : "memory"); }
        else if (KIND == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + (threadIdx.x >> 6) * 1024), 16, 0xFFFFFFF0u, 0, 0, 0);
        else if (KIND == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + (threadIdx.x >> 6) * 1024), 16, (off + 16 * (i + 1)) % bytes, 0, 0, 0);
        else { const unsigned o = (off + 16 * (i + 1)) % bytes; asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(b[i]) : "v"(o), "s"(r) : "memory"); }
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory");
    u32x4 v;
    asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                 : "=v"(v.x), "=v"(v.y), "=v"(v.z), "=v"(v.w) : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < K; ++i) acc += (KIND == 2 || KIND == 4) ? 0u : b[i].x * 0u;
    v.w += acc;
    out[tid] = v;
}

// the OLDER operation is an in-range LDS-DMA (16 bytes per lane into the wave's 1 KB of LDS); younger: KIND as above
__device__ __attribute__((aligned(256))) unsigned char zero_page[1024];
static const void* g_hot = nullptr;       // a small hipMalloc'd buffer (ordinary device memory, L2-hot): passed to probe_dma

template <int K, int KIND, int PAT = 0>
__global__ void probe_dma(const u32x4* __restrict__ src, unsigned bytes, u32x4* __restrict__ out, unsigned stride, const void* hot_buf) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[16384];
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0, 0x00020000);
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned gw = tid >> 6;                                  // global wave
    const unsigned off = PAT == 0 ? (tid * stride) % (bytes / 16) * 16
                                  : (unsigned)(((unsigned long long)gw * 40961u * 1024u + (lane >> 2) * 1024u + (lane & 3) * 16u) % (bytes - 65536u));   // 16 rows x 64 B, rows 1 KB apart, cold
    // poison, then the older DMA
    *reinterpret_cast<u32x4*>(lds + wave * 1024 + lane * 16) = u32x4{0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + wave * 1024), 16, off, 0, 0, 0);
    u32x4 b[K > 0 ? K : 1];
#pragma unroll
    for (int i = 0; i < K; ++i) {
        if (KIND == 0) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(b[i]) : "v"(off), "s"(rz) : "memory");
        else if (KIND == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 8192 + wave * 1024), 16, 0xFFFFFFF0u, 0, 0, 0);
        else if (KIND == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 8192 + wave * 1024), 16, (off + 16 * (i + 1)) % bytes, 0, 0, 0);
        else if (KIND == 5) {       // in-range DMA from a ZERO PAGE through its own resource (what replaces an all-out-of-range DMA)
            const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc((void*)zero_page, 0, 1024, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(zr, (__attribute__((address_space(3))) void*)(lds + 8192 + wave * 1024), 16, lane * 16, 0, 0, DSN_PROBE_AUX);
        } else if (KIND == 10) {    // in-range DMA from a small hipMalloc'd buffer (ordinary device memory, L2-hot), own resource
            const __amdgpu_buffer_rsrc_t hr = __builtin_amdgcn_make_buffer_rsrc((void*)hot_buf, 0, 1024, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(hr, (__attribute__((address_space(3))) void*)(lds + 8192 + wave * 1024), 16, lane * 16, 0, 0, DSN_PROBE_AUX);
        } else if (KIND == 11) {    // in-range DMA from the SAME resource as the older one, but a small L2-hot part of it (its first KB)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 8192 + wave * 1024), 16, lane * 16, 0, 0, DSN_PROBE_AUX);
        } else if (KIND == 6) {     // a DMA with HALF of its lanes out of range
            const unsigned o = (lane & 1) ? 0xFFFFFFF0u : (off + 16 * (i + 1)) % bytes;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 8192 + wave * 1024), 16, o, 0, 0, 0);
        } else if (KIND == 8) {     // buffer STORES (16 bytes per lane) to a scratch area
            const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7FFFFFFF, 0x00020000);
            const u32x4 val = {tid, (unsigned)i, 0u, 0u};
            const unsigned so = (gridDim.x * blockDim.x + tid * 16 + (unsigned)i) * 16u;
            asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" :: "v"(val), "v"(so), "s"(orr) : "memory");
        } else if (KIND == 7) {     // a DMA with ONE lane in range
            const unsigned o = lane ? 0xFFFFFFF0u : (off + 16 * (i + 1)) % bytes;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 8192 + wave * 1024), 16, o, 0, 0, 0);
        }
        else { const unsigned o = (off + 16 * (i + 1)) % bytes; asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(b[i]) : "v"(o), "s"(r) : "memory"); }
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory");
    u32x4 v;
    const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(lds + wave * 1024 + lane * 16);
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(la) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < K; ++i) acc += (KIND == 2 || KIND >= 4) ? 0u : b[i].x * 0u;
    v.w += acc;
    out[tid] = v;
}

template <int K, int KIND, int PAT = 0> int run_dma(const u32x4* d, unsigned bytes, u32x4* o, const std::vector<unsigned>& h, int blocks) {
    const unsigned stride = 9973;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((probe_dma<K, KIND, PAT>), dim3(blocks), dim3(256), 0, 0, d, bytes, o, stride, g_hot);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned> r((size_t)blocks * 256 * 4);
    CHECK(hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (unsigned t = 0; t < (unsigned)blocks * 256; ++t) {
        const unsigned lane = t & 63, gw = t >> 6;
        const unsigned off = PAT == 0 ? (t * stride) % (bytes / 16) * 4
                                      : (unsigned)((((unsigned long long)gw * 40961u * 1024u + (lane >> 2) * 1024u + (lane & 3) * 16u) % (bytes - 65536u)) / 4);
        for (int e = 0; e < 4; ++e) bad += r[(size_t)t * 4 + e] != h[off + e];
    }
    return bad;
}

template <int K, int KIND> int run(const u32x4* d, unsigned bytes, u32x4* o, const std::vector<unsigned>& h, int blocks) {
    const unsigned stride = 9973;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((probe<K, KIND>), dim3(blocks), dim3(256), 0, 0, d, bytes, o, stride);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned> r((size_t)blocks * 256 * 4);
    CHECK(hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (unsigned t = 0; t < (unsigned)blocks * 256; ++t) {
        const unsigned off = (t * stride) % (bytes / 16) * 4;
        for (int e = 0; e < 4; ++e) bad += r[(size_t)t * 4 + e] != h[off + e];
    }
    return bad;
}

int main() {
    const unsigned bytes = 1u << 30;
    std::vector<unsigned> h(bytes / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u + 12345u);
    u32x4 *d, *o;
    const int blocks = 4096;
    CHECK(hipMalloc(&d, bytes));
    CHECK(hipMalloc(&o, (size_t)blocks * 256 * 16 * 18));      // results + scratch for the store case
    CHECK(hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice));
    void* hot; CHECK(hipMalloc(&hot, 4096)); CHECK(hipMemset(hot, 0, 4096));
    g_hot = hot;
    const char* names[4] = {"OOB register loads, zero-record resource", "OOB register loads, out-of-range offset", "OOB LDS-DMA", "in-range register loads (control)"};
    printf("wrong values among %d loaded dwords (one cold in-range load, then K younger operations, s_waitcnt vmcnt(K)):\n", blocks * 256 * 4);
    printf("K = 4:  %-46s %d\n", names[0], run<4, 0>(d, bytes, o, h, blocks));
    printf("K = 4:  %-46s %d\n", names[1], run<4, 1>(d, bytes, o, h, blocks));
    printf("K = 4:  %-46s %d\n", names[2], run<4, 2>(d, bytes, o, h, blocks));
    printf("K = 4:  %-46s %d\n", names[3], run<4, 3>(d, bytes, o, h, blocks));
    printf("K = 12: %-46s %d\n", names[0], run<12, 0>(d, bytes, o, h, blocks));
    printf("K = 12: %-46s %d\n", names[1], run<12, 1>(d, bytes, o, h, blocks));
    printf("K = 12: %-46s %d\n", names[2], run<12, 2>(d, bytes, o, h, blocks));
    printf("K = 12: %-46s %d\n", names[3], run<12, 3>(d, bytes, o, h, blocks));
    printf("K = 4:  %-46s %d\n", "in-range LDS-DMA", run<4, 4>(d, bytes, o, h, blocks));
    printf("K = 0:  %-46s %d\n", "(no younger operation: vmcnt(0))", run<0, 0>(d, bytes, o, h, blocks));
    printf("OLDER operation = in-range LDS-DMA (read back from LDS after the wait):\n");
    printf("K = 4:  %-46s %d\n", names[0], run_dma<4, 0>(d, bytes, o, h, blocks));
    printf("K = 4:  %-46s %d\n", names[2], run_dma<4, 2>(d, bytes, o, h, blocks));
    printf("K = 4:  %-46s %d\n", "in-range LDS-DMA", run_dma<4, 4>(d, bytes, o, h, blocks));
    printf("K = 4:  %-46s %d\n", names[3], run_dma<4, 3>(d, bytes, o, h, blocks));
    printf("K = 4:  %-46s %d\n", "in-range LDS-DMA from a zero page (own resource)", run_dma<4, 5>(d, bytes, o, h, blocks));
    printf("K = 4:  %-46s %d\n", "LDS-DMA, every second lane out of range", run_dma<4, 6>(d, bytes, o, h, blocks));
    printf("K = 4:  %-46s %d\n", "LDS-DMA, one lane in range", run_dma<4, 7>(d, bytes, o, h, blocks));
    printf("K = 4:  %-46s %d\n", "buffer stores", run_dma<4, 8>(d, bytes, o, h, blocks));
    printf("K = 12: %-46s %d\n", "in-range LDS-DMA from a zero page (own resource)", run_dma<12, 5>(d, bytes, o, h, blocks));
    printf("K = 12: %-46s %d\n", names[2], run_dma<12, 2>(d, bytes, o, h, blocks));
    printf("K = 12: %-46s %d\n", "in-range LDS-DMA", run_dma<12, 4>(d, bytes, o, h, blocks));
    printf("K = 0:  %-46s %d\n", "(no younger operation: vmcnt(0))", run_dma<0, 0>(d, bytes, o, h, blocks));
    printf("OLDER operation = in-range LDS-DMA of 16 rows x 64 B, cold (the kernels' access pattern):\n");
    printf("K = 4:  %-46s %d\n", "in-range LDS-DMA from a zero page (own resource)", run_dma<4, 5, 1>(d, bytes, o, h, blocks));
    printf("K = 2:  %-46s %d\n", "in-range LDS-DMA from a zero page (own resource)", run_dma<2, 5, 1>(d, bytes, o, h, blocks));
    printf("K = 4:  %-46s %d\n", "in-range LDS-DMA, hot hipMalloc'd buffer (own rsrc)", run_dma<4, 10, 1>(d, bytes, o, h, blocks));
    printf("K = 4:  %-46s %d\n", "in-range LDS-DMA, hot first KB of the same buffer", run_dma<4, 11, 1>(d, bytes, o, h, blocks));
    printf("K = 2:  %-46s %d\n", "in-range LDS-DMA, hot first KB of the same buffer", run_dma<2, 11, 1>(d, bytes, o, h, blocks));
    printf("K = 4:  %-46s %d\n", names[2], run_dma<4, 2, 1>(d, bytes, o, h, blocks));
    printf("K = 4:  %-46s %d\n", "in-range LDS-DMA (cold)", run_dma<4, 4, 1>(d, bytes, o, h, blocks));
    printf("K = 4:  %-46s %d\n", names[3], run_dma<4, 3, 1>(d, bytes, o, h, blocks));
    return 0;
}
