// Do the matrix pipe and the vector ALU of a SIMD overlap across two waves?  A 512-thread workgroup: waves 0..3 (role A) and
// 4..7 (role B) land pairwise on the same SIMD (wave_place.hip).  Roles: M = a stream of v_mfma_f32_32x32x2_f32 on two
// accumulators, V = a stream of v_fma_f32 (two chains), T = v_exp_f32 / v_rcp_f32 (transcendental rate), idle = return.
// Timed: M | idle, V | idle, M | M, V | V, M | V, M | T  -- if the pipes overlap, M | V costs max(M | idle, V | idle).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void role_m(float* out, int iters) {
    f16v c0 = {0}, c1 = {0};
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        }
    }
    float r = 0;
    for (int j = 0; j < 16; ++j) r += c0[j] + c1[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__device__ __forceinline__ void role_v(float* out, int iters) {
    float x = threadIdx.x * 1e-3f, y = x + 1.f;
    const float m = 1.0001f, c = 1e-6f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                         "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                         : "+v"(x), "+v"(y) : "v"(m), "v"(c));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y;
}
__device__ __forceinline__ void role_t(float* out, int iters) {
    float x = threadIdx.x * 1e-3f + 0.5f, y = x + 1.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile("v_exp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_exp_f32 %0, %0\n v_rcp_f32 %1, %1\n"
                         "v_exp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_exp_f32 %0, %0\n v_rcp_f32 %1, %1\n"
                         : "+v"(x), "+v"(y));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y;
}
// RA / RB: 0 idle, 1 M, 2 V, 3 T; iteration counts per role so that every busy role runs about as long alone
template <int RA, int RB>
__global__ __launch_bounds__(512) void k(float* out, int im, int iv, int it) {
    const int role = threadIdx.x < 256 ? RA : RB;
    if (role == 1) role_m(out, im);
    else if (role == 2) role_v(out, iv);
    else if (role == 3) role_t(out, it);
}
template <int RA, int RB>
float run(const char* name, float* out, int im, int iv, int it) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<RA, RB><<<256, 512>>>(out, im / 50 + 1, iv / 50 + 1, it / 50 + 1);
    (void)hipEventRecord(e0);
    k<RA, RB><<<256, 512>>>(out, im, iv, it);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-12s %8.3f ms\n", name, ms);
    return ms;
}
int main() {
    float* out; (void)hipMalloc(&out, 4 << 20);
    const int im = 4000, iv = 16000, it = 8000;  // 64 000 MFMAs (64 cycles each) | 1 024 000 FMAs | 512 000 transcendentals per wave
    float m = run<1, 0>("M | idle", out, im, iv, it);
    float v = run<2, 0>("V | idle", out, im, iv, it);
    float t = run<3, 0>("T | idle", out, im, iv, it);
    run<1, 1>("M | M", out, im, iv, it);
    run<2, 2>("V | V", out, im, iv, it);
    run<3, 3>("T | T", out, im, iv, it);
    run<1, 2>("M | V", out, im, iv, it);
    run<1, 3>("M | T", out, im, iv, it);
    run<2, 3>("V | T", out, im, iv, it);
    printf("per MFMA alone: %.1f ns; per FMA alone: %.2f ns; per transcendental alone: %.2f ns\n", m * 1e6 / (im * 16.0), v * 1e6 / (iv * 64.0), t * 1e6 / (it * 64.0));
    return 0;
}
