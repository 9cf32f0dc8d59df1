// where do the 8 waves of a 512-thread workgroup land?  HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh[12] se[15:13]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(unsigned* out) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2] = id; out[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2 + 1] = xcc; }
    // keep the wave alive a little so that all workgroups are resident together
    for (int i = 0; i < 2000; ++i) asm volatile("s_nop 15");
}
int main(int argc, char** argv) {
    unsigned* d; int blocks = 256, thr = argc > 1 ? atoi(argv[1]) : 512, waves = blocks * thr / 64;
    hipMalloc(&d, waves * 8);
    k<<<blocks, thr>>>(d);
    unsigned* h = new unsigned[waves * 2];
    hipMemcpy(h, d, waves * 8, hipMemcpyDeviceToHost);
    int pairs_ok = 0; const int nw = thr / 64;
    for (int b = 0; b < blocks; ++b) {
        int simd[16];
        for (int w = 0; w < nw; ++w) simd[w] = (h[(b * nw + w) * 2] >> 4) & 3;
        bool ok = true;
        for (int w = 4; w < nw; ++w) ok &= simd[w] == simd[w - 4];
        pairs_ok += ok;
        if (b < 6) { printf("wg %d: simd", b); for (int w = 0; w < nw; ++w) printf(" %d", simd[w]); printf("\n"); }
    }
    printf("%d-thread workgroups whose wave w and w+4 (and w+8 ...) share a SIMD: %d of %d\n", thr, pairs_ok, blocks);
    return 0;
}
