// one_kernel.hip -- compile ONE instantiation of a libvecsim kernel (register / scratch experiments):
//   hipcc <build.py FLAGS> -DK_E='QcpT<0>' -DK_ARGS='false,true,1,4,64,false,2' -c scratch/r3/one_kernel.hip
#include "../../simurlacra_amd/csrc/vecsim_kernels.h"
namespace vs {
#ifndef K_NAME
#define K_NAME k_rollout_ws
#endif
template __global__ void K_NAME<K_E, K_ARGS>(Task, Dev, int, uint64_t, uint64_t, uint64_t);
}
