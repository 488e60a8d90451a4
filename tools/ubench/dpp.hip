// DPP wave shift semantics + ripple-step cost on gfx950 (dev tool)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double shr1(double v) {  // lane i <- lane i-1 (lane 0 keeps its own)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shl1(double v) {  // lane i <- lane i+1 (lane 63 keeps its own)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__global__ void sem(double *o) { double v = threadIdx.x; o[threadIdx.x] = shr1(v); o[64 + threadIdx.x] = shl1(v); }
__global__ void ripple(double *out, long long *cyc, int n) {
    double p[9], b[3], y0 = threadIdx.x == 0 ? 1.0 : 0.0, y1 = 0, y2 = 0;
    for (int i = 0; i < 9; ++i) p[i] = 0.1 * (i + 1) / 9.0 + threadIdx.x * 1e-3;
    for (int i = 0; i < 3; ++i) b[i] = 0.01 * i;
    long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < n; ++s) {
        const double o0 = fma(p[0], y0, fma(p[1], y1, fma(p[2], y2, b[0])));
        const double o1 = fma(p[3], y0, fma(p[4], y1, fma(p[5], y2, b[1])));
        const double o2 = fma(p[6], y0, fma(p[7], y1, fma(p[8], y2, b[2])));
        const double n0 = shr1(o0), n1 = shr1(o1), n2 = shr1(o2);
        if (threadIdx.x != 0) { y0 = n0; y1 = n1; y2 = n2; }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = y0 + y1 + y2; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double *d; long long *c; hipMalloc(&d, 128 * 8); hipMalloc(&c, 8);
    double h[128]; long long cy;
    sem<<<1, 64>>>(d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("shr1: lane0=%g lane1=%g lane15=%g lane16=%g lane17=%g lane32=%g lane63=%g\n", h[0], h[1], h[15], h[16], h[17], h[32], h[63]);
    printf("shl1: lane0=%g lane15=%g lane16=%g lane31=%g lane47=%g lane62=%g lane63=%g\n", h[64], h[79], h[80], h[95], h[111], h[126], h[127]);
    for (int r = 0; r < 2; ++r) { ripple<<<1, 64>>>(d, c, 10000); hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost); printf("ripple step (9 fma + 3 dpp shr): %.1f cyc\n", cy / 10000.0); }
    return 0;
}
