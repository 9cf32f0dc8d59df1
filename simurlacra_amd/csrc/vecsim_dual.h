// vecsim_dual.h -- forward-mode dual numbers for the step Jacobians.
//
// The env math in vecsim_envs.h is templated on its scalar type R.  R = float is the production path (identical code to
// the untemplated version); R = Dual<N> carries N tangents (one per input: S state dims + A action dims) through the very
// same expressions, so d(s', r, obs)/d(s, a) comes out of one extra evaluation per lane with no hand-derived formulas.
// Conventions at kinks follow the true one-sided derivative of the branch that was taken: a clipped action or an action
// inside the dead zone has zero gradient, fmod has slope 1, the +-pi fold slope -1 on the folded branch, sign() slope 0.
#pragma once
#include <hip/hip_runtime.h>

namespace vs {

template <int N>
struct Dual {
    float v;
    float d[N];
    __device__ __forceinline__ Dual() {}
    __device__ __forceinline__ Dual(float x) : v(x) {
#pragma unroll
        for (int k = 0; k < N; ++k) d[k] = 0.f;
    }
};

#define VS_DUAL_BIN(OP, VAL, TAN)                                                                        \
    template <int N>                                                                                     \
    __device__ __forceinline__ Dual<N> operator OP(const Dual<N>& a, const Dual<N>& b) {                 \
        Dual<N> r;                                                                                       \
        r.v = VAL;                                                                                       \
        _Pragma("unroll") for (int k = 0; k < N; ++k) r.d[k] = TAN;                                      \
        return r;                                                                                        \
    }                                                                                                    \
    template <int N>                                                                                     \
    __device__ __forceinline__ Dual<N> operator OP(const Dual<N>& a, float b) { return a OP Dual<N>(b); } \
    template <int N>                                                                                     \
    __device__ __forceinline__ Dual<N> operator OP(float a, const Dual<N>& b) { return Dual<N>(a) OP b; }

VS_DUAL_BIN(+, a.v + b.v, a.d[k] + b.d[k])
VS_DUAL_BIN(-, a.v - b.v, a.d[k] - b.d[k])
VS_DUAL_BIN(*, a.v * b.v, a.d[k] * b.v + a.v * b.d[k])
VS_DUAL_BIN(/, a.v / b.v, (a.d[k] - (a.v / b.v) * b.d[k]) / b.v)
#undef VS_DUAL_BIN

template <int N>
__device__ __forceinline__ Dual<N> operator-(const Dual<N>& a) {
    Dual<N> r;
    r.v = -a.v;
#pragma unroll
    for (int k = 0; k < N; ++k) r.d[k] = -a.d[k];
    return r;
}
template <int N, class B>
__device__ __forceinline__ Dual<N>& operator+=(Dual<N>& a, const B& b) { a = a + b; return a; }
template <int N, class B>
__device__ __forceinline__ Dual<N>& operator-=(Dual<N>& a, const B& b) { a = a - b; return a; }
template <int N, class B>
__device__ __forceinline__ Dual<N>& operator*=(Dual<N>& a, const B& b) { a = a * b; return a; }

#define VS_DUAL_CMP(OP)                                                                                   \
    template <int N>                                                                                      \
    __device__ __forceinline__ bool operator OP(const Dual<N>& a, const Dual<N>& b) { return a.v OP b.v; } \
    template <int N>                                                                                      \
    __device__ __forceinline__ bool operator OP(const Dual<N>& a, float b) { return a.v OP b; }           \
    template <int N>                                                                                      \
    __device__ __forceinline__ bool operator OP(float a, const Dual<N>& b) { return a OP b.v; }
VS_DUAL_CMP(<)
VS_DUAL_CMP(>)
VS_DUAL_CMP(<=)
VS_DUAL_CMP(>=)
#undef VS_DUAL_CMP

// value access / generic helpers with float overloads that keep the production code unchanged
__device__ __forceinline__ float val(float x) { return x; }
template <int N>
__device__ __forceinline__ float val(const Dual<N>& x) { return x.v; }

__device__ __forceinline__ float vsel(bool c, float a, float b) { return c ? a : b; }
template <int N>
__device__ __forceinline__ Dual<N> vsel(bool c, const Dual<N>& a, const Dual<N>& b) { return c ? a : b; }

__device__ __forceinline__ float vmin(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ float vmax(float a, float b) { return fmaxf(a, b); }
template <int N>
__device__ __forceinline__ Dual<N> vmin(const Dual<N>& a, const Dual<N>& b) { return fminf(a.v, b.v) == a.v ? a : b; }
template <int N>
__device__ __forceinline__ Dual<N> vmax(const Dual<N>& a, const Dual<N>& b) { return fmaxf(a.v, b.v) == a.v ? a : b; }
template <int N>
__device__ __forceinline__ Dual<N> vmin(const Dual<N>& a, float b) { return vmin(a, Dual<N>(b)); }
template <int N>
__device__ __forceinline__ Dual<N> vmax(const Dual<N>& a, float b) { return vmax(a, Dual<N>(b)); }

__device__ __forceinline__ float vabs(float a) { return fabsf(a); }
template <int N>
__device__ __forceinline__ Dual<N> vabs(const Dual<N>& a) { return a.v < 0.f ? -a : a; }

__device__ __forceinline__ bool visnan(float a) { return isnan(a); }
template <int N>
__device__ __forceinline__ bool visnan(const Dual<N>& a) { return isnan(a.v); }

}  // namespace vs
