// Register layout of v_mfma_f32_32x32x2_f32 on gfx950, measured: D = A (32 x 2) * B (2 x 32) with A[i][k] = 100 i + k + 1 coded
// so that every product is unique; prints, per accumulator register r and lane l, which (i, j) the value belongs to.
// build: hipcc --offload-arch=gfx950 -O2 -o scratch/ubench/mfma_layout scratch/ubench/mfma_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(const float* A, const float* B, float* D) {
    int l = threadIdx.x;
    f16v acc = {0};
    // assumed operand layout: lane l holds A[i = l % 32][k = l / 32] and B[k = l / 32][j = l % 32]
    float a = A[(l % 32) * 2 + l / 32], b = B[(l / 32) * 32 + l % 32];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[r * 64 + l] = acc[r];
}
int main() {
    float hA[64], hB[64], hD[1024];
    for (int i = 0; i < 32; ++i) for (int kk = 0; kk < 2; ++kk) hA[i * 2 + kk] = (float)(i + 1) * (kk == 0 ? 1.f : 64.f);
    for (int kk = 0; kk < 2; ++kk) for (int j = 0; j < 32; ++j) hB[kk * 32 + j] = kk == 0 ? (float)(j + 1) : 0.f;
    // D[i][j] = (i + 1)(j + 1): decode i and j from the value with k = 0 only, then check k = 1 separately
    float *dA, *dB, *dD;
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int r = 0; r < 16; ++r) for (int l = 0; l < 64; ++l) {
        int i = 8 * (r / 4) + 4 * (l / 32) + r % 4, j = l % 32;  // the layout the policy kernel would assume
        if (hD[r * 64 + l] != (float)((i + 1) * (j + 1))) { if (bad < 8) printf("r %d lane %d: got %g, assumed (i %d, j %d) -> %d\n", r, l, hD[r * 64 + l], i, j, (i + 1) * (j + 1)); ++bad; }
    }
    printf("k = 0 pass: %d mismatches of 1024\n", bad);
    for (int kk = 0; kk < 2; ++kk) for (int j = 0; j < 32; ++j) hB[kk * 32 + j] = kk == 1 ? (float)(j + 1) : 0.f;
    hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    bad = 0;
    for (int r = 0; r < 16; ++r) for (int l = 0; l < 64; ++l) {
        int i = 8 * (r / 4) + 4 * (l / 32) + r % 4, j = l % 32;
        if (hD[r * 64 + l] != 64.f * (float)((i + 1) * (j + 1))) ++bad;
    }
    printf("k = 1 pass: %d mismatches of 1024\n", bad);
    return 0;
}
