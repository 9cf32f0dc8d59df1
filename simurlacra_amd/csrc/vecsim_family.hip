// vecsim_family.hip -- the kernels of ONE env family and their launchers; compiled once per family with -DVS_FAMILY=<n>
// (the value of enum vs_env_type) so that the nine families build in parallel (build.py).
#define VS_TU_FAMILY 1
#include "vecsim_kernels.h"

#ifndef VS_FAMILY
#error "compile with -DVS_FAMILY=<vs_env_type>"
#endif

namespace vs {
#if VS_FAMILY == 0
template struct Launch<Omo>;
#elif VS_FAMILY == 1
template struct Launch<Bob>;
#elif VS_FAMILY == 2
template struct Launch<QQ>;
#elif VS_FAMILY == 3
template struct Launch<Qcp>;
#elif VS_FAMILY == 4
template struct Launch<Qbb>;
#elif VS_FAMILY == 5
template struct Launch<QQSt>;
#elif VS_FAMILY == 6
template struct Launch<QcpSt>;
#elif VS_FAMILY == 7
template struct Launch<Pend>;
#elif VS_FAMILY == 8
template struct Launch<BobD>;
#else
#error "unknown VS_FAMILY"
#endif
}  // namespace vs
