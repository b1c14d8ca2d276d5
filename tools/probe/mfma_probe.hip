// Probe: lane->element maps of v_mfma_f32_16x16x4_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* D) {   // A[16][4], B[4][16] row-major, D[16][16]
    int l = threadIdx.x;
    float a = A[(l & 15) * 4 + (l >> 4)];
    float b = B[(l >> 4) * 16 + (l & 15)];
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}
int main() {
    float hA[64], hB[64], hD[256], ref[256];
    for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 4; ++kk) hA[i * 4 + kk] = (float)(i * 7 + kk * 3 + 1);
    for (int kk = 0; kk < 4; ++kk) for (int j = 0; j < 16; ++j) hB[kk * 16 + j] = (float)(kk * 11 + j * 2 + 5);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int kk = 0; kk < 4; ++kk) s += hA[i*4+kk] * hB[kk*16+j]; ref[i*16+j] = s; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) if (hD[i] != ref[i]) ++bad;
    printf("16x16x4f32 probe: %d mismatches; D[0][0..3]=%g %g %g %g ref %g %g %g %g; D[1][0]=%g ref %g\n", bad, hD[0], hD[1], hD[2], hD[3], ref[0], ref[1], ref[2], ref[3], hD[16], ref[16]);
    return 0;
}
