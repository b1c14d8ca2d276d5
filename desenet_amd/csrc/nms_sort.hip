// Large-case sort of the NMS candidate keys (validation settings: conf 0.001 + multi_label leave up to n*nc = 151200 keys per
// image): ONE device-wide radix sort over all images (rocPRIM through hipCUB -- a library sort, like a library GEMM, is not a
// hot kernel to hand-write) instead of the one-workgroup-per-image bitonic network of detect_nms.hip (5.4 ms at 16 x 151200
// keys), which is kept for the common case (<= 32768 keys per image: 0.2 ms).  The large-case key carries the image in its
// top bits -- (image << 48) | (~conf bits, 30 significant) << 18 | row*nc + cls -- so one flat sort groups the images in order;
// unused slots hold ~0 and sort to the very end; image b's keys then start at sum(counts[:b]).  Keys are unique, so any correct
// sort yields the same order: the selection stays bit-exact.
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace {
__global__ void nms_starts_kernel(const int32_t* __restrict__ counts, int bs, int32_t* __restrict__ starts) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int acc = 0;
        for (int b = 0; b < bs; ++b) { starts[b] = acc; acc += counts[b]; }
    }
}
}  // namespace

// temp layout: [starts (bs ints) | pad to 256 | hipcub temp storage]
int64_t dsn_nms_radix_temp_bytes(int32_t bs, int64_t cap) {
    size_t tb = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, tb, (const uint64_t*)nullptr, (uint64_t*)nullptr, (int)(bs * cap), 0, 56,
                                            (hipStream_t)0);
    return (int64_t)(((size_t)bs * 4 + 255) / 256 * 256 + tb + 256);
}

// returns the device array of per-image start offsets into keys_out through *starts_out
int dsn_nms_radix_sort(const uint64_t* keys_in, uint64_t* keys_out, const int32_t* counts, int32_t bs, int64_t cap, void* temp,
                       int64_t temp_bytes, const int32_t** starts_out, hipStream_t st) {
    int32_t* starts = (int32_t*)temp;
    char* cub = (char*)temp + ((size_t)bs * 4 + 255) / 256 * 256;
    size_t tb = (size_t)temp_bytes - (size_t)(cub - (char*)temp);
    hipLaunchKernelGGL(nms_starts_kernel, dim3(1), dim3(64), 0, st, counts, bs, starts);
    hipError_t e = hipcub::DeviceRadixSort::SortKeys(cub, tb, keys_in, keys_out, (int)(bs * cap), 0, 56, st);
    *starts_out = starts;
    return e == hipSuccess ? 0 : (int)e;
}
