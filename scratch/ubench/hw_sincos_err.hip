// hw_sincos_err.hip: absolute error of v_sin_f32 / v_cos_f32 (argument in revolutions) behind an exact two-term reduction to
// [-pi, pi], against double-precision sin / cos, and of the library's polynomial sincos_fast, over the angles the envs see
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__device__ inline void sincos_hw(float x, float* sn, float* cs) {
    const float TWOPI_HI = 6.28318548202514648f, TWOPI_LO = -1.74845553146951715e-07f;
    float q = rintf(x * 0.159154943091895336f);
    float r = fmaf(-q, TWOPI_HI, x);
    r = fmaf(-q, TWOPI_LO, r);
    float rev = r * 0.159154943091895336f;
    *sn = __builtin_amdgcn_sinf(rev);
    *cs = __builtin_amdgcn_cosf(rev);
}
__device__ inline void sincos_poly(float x, float* sn, float* cs) {
    const float PIO2_HI = 1.57079637050628662109375f, PIO2_MID = -4.37113882867379127e-08f, PIO2_LO = -1.71512451008199912e-15f;
    float q = rintf(x * 0.636619772367581343f);
    float r = fmaf(-q, PIO2_HI, x);
    r = fmaf(-q, PIO2_MID, r);
    r = fmaf(-q, PIO2_LO, r);
    int n = (int)q;
    float r2 = r * r;
    float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f), r2 * r, r);
    float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f), r2 * r2, fmaf(-0.5f, r2, 1.0f));
    float s0 = (n & 1) ? pc : ps, c0 = (n & 1) ? ps : pc;
    *sn = (n & 2) ? -s0 : s0;
    *cs = ((n + 1) & 2) ? -c0 : c0;
}
__global__ void k(const float* x, float* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    sincos_hw(x[i], &o[4 * i], &o[4 * i + 1]);
    sincos_poly(x[i], &o[4 * i + 2], &o[4 * i + 3]);
}
int main() {
    const int n = 1 << 22;
    for (double span : {0.01, 0.8, 3.2, 13.0, 70.0}) {
        std::vector<float> x(n), o(4 * n);
        for (int i = 0; i < n; ++i) x[i] = (float)(-span + 2.0 * span * (i + 0.37) / n);
        float *dx, *dout;
        hipMalloc(&dx, n * 4); hipMalloc(&dout, 16 * (size_t)n);
        hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
        k<<<n / 256, 256>>>(dx, dout, n);
        hipMemcpy(o.data(), dout, 16 * (size_t)n, hipMemcpyDeviceToHost);
        double e[4] = {0, 0, 0, 0};
        for (int i = 0; i < n; ++i) {
            double s = sin((double)x[i]), c = cos((double)x[i]);
            e[0] = fmax(e[0], fabs(o[4 * i] - s)); e[1] = fmax(e[1], fabs(o[4 * i + 1] - c));
            e[2] = fmax(e[2], fabs(o[4 * i + 2] - s)); e[3] = fmax(e[3], fabs(o[4 * i + 3] - c));
        }
        printf("|x| <= %-5g  hw: sin %.3e cos %.3e   poly: sin %.3e cos %.3e\n", span, e[0], e[1], e[2], e[3]);
        hipFree(dx); hipFree(dout);
    }
    return 0;
}
