// What does the record stream of the fused kernels cost as a write PATTERN?  Every wave writes, per env step, two 16-byte
// values per lane (1 KB per wave and store) after `pad` dependent FMAs of "physics".
//   layout 0: the library's planes  [t][plane][ld]  (a row of all envs per step: waves of one step are neighbours in memory)
//   layout 1: wave-contiguous       [wave][t][plane][64]  (every wave streams through its own contiguous region)
// build: hipcc -O3 --offload-arch=gfx950 rec_stream.hip -o rec_stream ; run: ./rec_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int LAYOUT, bool NT>
__global__ __launch_bounds__(256) void k(v4f* __restrict__ out, int n, int T, int pad, int jitter) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int wave = i >> 6, lane = i & 63;
    float x = (float)i * 1e-9f;
    const int mypad = pad + (jitter ? (wave * 2654435761u >> 16) % (unsigned)jitter : 0);
    for (int t = 0; t < T; ++t) {
        for (int q = 0; q < mypad; ++q) x = __builtin_fmaf(x, 0.999f, 1e-3f);
        v4f a = {x, x + 1.f, x + 2.f, x + 3.f}, b = {x + 4.f, x + 5.f, x + 6.f, x + 7.f};
        v4f *p0, *p1;
        if (LAYOUT == 0) {
            p0 = out + ((size_t)t * 2 + 0) * n + i;
            p1 = out + ((size_t)t * 2 + 1) * n + i;
        } else {
            p0 = out + (((size_t)wave * T + t) * 2 + 0) * 64 + lane;
            p1 = p0 + 64;
        }
        if (NT) {
            __builtin_nontemporal_store(a, p0);
            __builtin_nontemporal_store(b, p1);
        } else {
            *p0 = a;
            *p1 = b;
        }
    }
}
int main() {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const size_t cap = (size_t)6 << 30;
    v4f* buf;
    if (hipMalloc(&buf, cap) != hipSuccess) return 1;
    hipMemset(buf, 0, cap);
    const int ns[] = {65536, 131072, 262144, 1048576};
    const int pads[] = {0, 100, 200, 400};
    for (int n : ns)
        for (int wg : {64, 256})
            for (int pad : pads)
                for (int jit : {0, 64})
                    for (int layout = 0; layout < 2; ++layout)
                        for (int nt = 0; nt < 2; ++nt) {
                            size_t per_t = (size_t)n * 32;
                            int T = (int)((cap / 2) / per_t);
                            if (T > 400) T = 400;
                            float best = 1e30f;
                            for (int rep = 0; rep < 4; ++rep) {
                                v4f* o = buf + (rep & 1) * (cap / 2 / 16);
                                hipEventRecord(e0);
                                if (layout == 0 && nt == 0) k<0, false><<<n / wg, wg>>>(o, n, T, pad, jit);
                                if (layout == 0 && nt == 1) k<0, true><<<n / wg, wg>>>(o, n, T, pad, jit);
                                if (layout == 1 && nt == 0) k<1, false><<<n / wg, wg>>>(o, n, T, pad, jit);
                                if (layout == 1 && nt == 1) k<1, true><<<n / wg, wg>>>(o, n, T, pad, jit);
                                hipEventRecord(e1);
                                hipEventSynchronize(e1);
                                float ms;
                                hipEventElapsedTime(&ms, e0, e1);
                                if (rep && ms < best) best = ms;
                            }
                            printf("n=%8d wg=%3d pad=%3d jitter=%2d layout=%d nt=%d T=%3d : %7.1f us  %6.0f GB/s  %6.1f ns/step\n", n, wg, pad,
                                   jit, layout, nt, T, best * 1e3, per_t * T / best * 1e-6, best * 1e6 / T);
                            fflush(stdout);
                        }
    return 0;
}
