// issue rate of ONE wave per SIMD vs several: long straight-line blocks (loop overhead amortised over 64 instructions),
// independent chains (ILP 8) vs one dependent chain
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(float* out, int iters) {
    float a[8];
    for (int j = 0; j < 8; ++j) a[j] = threadIdx.x * 1e-3f + j;
    const float m = 1.0001f, c = 1e-6f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {  // 8 independent chains
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m), "v"(c));
            } else if (MODE == 1) {  // one dependent chain
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             : "+v"(a[0]) : "v"(m), "v"(c));
            } else {  // two interleaved chains
                asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                             "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                             : "+v"(a[0]), "+v"(a[1]) : "v"(m), "v"(c));
            }
        }
    }
    float r = 0;
    for (int j = 0; j < 8; ++j) r += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE>
void run(const char* name, int blocks, int threads, float* out) {
    int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(out, 100);
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double ns_per = ms * 1e6 / (iters * 64.0);
    printf("%-22s %d waves/SIMD: %.3f ns per wave-instruction = %.2f cycles at 2.4 GHz (per SIMD: %.2f)\n", name, threads / 256, ns_per, ns_per * 2.4, ns_per * 2.4 / (threads / 256));
}
int main() {
    float* out; (void)hipMalloc(&out, 4 << 20);
    for (int thr : {256, 512, 1024}) {
        run<0>("8 independent chains", 256, thr, out);
        run<2>("2 interleaved chains", 256, thr, out);
        run<1>("1 dependent chain", 256, thr, out);
    }
    return 0;
}
