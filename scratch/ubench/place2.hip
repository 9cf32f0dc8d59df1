// place2.hip -- where do the waves of SMALL workgroups land when several workgroups share a compute unit?
//   ./place2 <blocks> <threads> [vgprs: 0 | 1 = pad the kernel to 512 registers (one wave per SIMD)]
// HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh[12] se[15:13]; XCC_ID[3:0]
// Prints, per role (= wave index inside its workgroup), how the roles mix on the SIMDs: for k_rollout_ws in 64-env workgroups
// (128 threads: wave 0 = physics, wave 1 = reward) the question is whether a SIMD gets one wave of each role or two of the same.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
template <bool PAD>
__global__ void k(unsigned* out) {
    if (PAD) asm volatile("" ::: "a255");  // 256 AGPRs + VGPRs: one wave per SIMD
    unsigned id, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    if ((threadIdx.x & 63) == 0) { out[w * 2] = id; out[w * 2 + 1] = xcc; }
    for (int i = 0; i < 4000; ++i) asm volatile("s_nop 15");  // stay resident until the whole grid is placed
}
int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 1024, thr = argc > 2 ? atoi(argv[2]) : 128, pad = argc > 3 ? atoi(argv[3]) : 0;
    const int nw = thr / 64, waves = blocks * nw;
    unsigned* d; hipMalloc(&d, waves * 8); hipMemset(d, 0, waves * 8);
    if (pad) k<true><<<blocks, thr>>>(d); else k<false><<<blocks, thr>>>(d);
    std::vector<unsigned> h(waves * 2);
    hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost);
    // key: (xcc, se, sh, cu) -> per SIMD: count of waves of each role
    std::map<unsigned, std::vector<int>> cus;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < nw; ++w) {
            const unsigned id = h[(b * nw + w) * 2], xcc = h[(b * nw + w) * 2 + 1] & 15u;
            const unsigned key = (xcc << 16) | (id & 0xFF00u);
            auto& v = cus[key];
            if (v.empty()) v.assign(4 * 16, 0);
            v[((id >> 4) & 3) * 16 + w] += 1;
        }
    std::map<std::string, int> pattern;
    int cu_n = 0;
    for (auto& kv : cus) {
        char buf[256]; int o = 0;
        for (int s = 0; s < 4; ++s) {
            o += snprintf(buf + o, sizeof(buf) - o, "[");
            for (int w = 0; w < nw; ++w) o += snprintf(buf + o, sizeof(buf) - o, "%d", kv.second[s * 16 + w]);
            o += snprintf(buf + o, sizeof(buf) - o, "]");
        }
        pattern[buf] += 1; ++cu_n;
    }
    printf("%d blocks x %d threads%s on %d compute units; per CU, per SIMD [waves of role 0, role 1, ..]:\n", blocks, thr, pad ? " (512 regs)" : "", cu_n);
    for (auto& p : pattern) printf("  %4d CUs  %s\n", p.second, p.first.c_str());
    return 0;
}
