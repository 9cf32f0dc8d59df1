// vecsim_mixed.hip -- BASELINE config 5: several env families stepped by ONE launch (k_rollout_mixed / k_step_mixed).
#define VS_TU_MIXED 1
#include "vecsim_kernels.h"

namespace vs {

void launch_rollout_mixed(const Segs* dev_segs, int total_blocks, hipStream_t st, bool ar, int rec, int k_steps, uint64_t seed, bool drk) {
    dim3 g((unsigned)total_blocks), b(BLOCK);
#define LM(AR, REC, DK) hipLaunchKernelGGL((k_rollout_mixed<AR, REC, DK>), g, b, 0, st, dev_segs, k_steps, seed)
    if (ar && drk) { if (rec == 0) LM(true, 0, true); else if (rec == 1) LM(true, 1, true); else LM(true, 2, true); }
    else if (ar) { if (rec == 0) LM(true, 0, false); else if (rec == 1) LM(true, 1, false); else LM(true, 2, false); }
    else { if (rec == 0) LM(false, 0, false); else if (rec == 1) LM(false, 1, false); else LM(false, 2, false); }
#undef LM
}

void launch_step_mixed(const Segs* dev_segs, int total_blocks, hipStream_t st, bool ar, bool drk) {
    dim3 g((unsigned)total_blocks), b(BLOCK);
    if (ar && drk) hipLaunchKernelGGL((k_step_mixed<true, true>), g, b, 0, st, dev_segs);
    else if (ar) hipLaunchKernelGGL((k_step_mixed<true, false>), g, b, 0, st, dev_segs);
    else hipLaunchKernelGGL((k_step_mixed<false, false>), g, b, 0, st, dev_segs);
}

}  // namespace vs
