// cost of one boundary-recursion step under different cross-lane mechanisms (dev tool)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ __forceinline__ double dppmov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bcast(double v, int k) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, k); hi = __builtin_amdgcn_readlane(hi, k);
    return __hiloint2double(hi, lo);
}
template <int MODE> __global__ void ripple(double *out, long long *cyc, int n) {
    double p[9], b[3], y0 = threadIdx.x == 0 ? 1.0 : 0.0, y1 = 0, y2 = 0, f0 = 0, f1 = 0, f2 = 0;
    for (int i = 0; i < 9; ++i) p[i] = 0.1 * (i + 1) / 9.0 + threadIdx.x * 1e-3;
    for (int i = 0; i < 3; ++i) b[i] = 0.01 * i;
    long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < n; ++s) {
        const double o0 = fma(p[0], y0, fma(p[1], y1, fma(p[2], y2, b[0])));
        const double o1 = fma(p[3], y0, fma(p[4], y1, fma(p[5], y2, b[1])));
        const double o2 = fma(p[6], y0, fma(p[7], y1, fma(p[8], y2, b[2])));
        if (MODE == 0) { const double n0 = dppmov<0x138>(o0), n1 = dppmov<0x138>(o1), n2 = dppmov<0x138>(o2); if (threadIdx.x != 0) { y0 = n0; y1 = n1; y2 = n2; } }
        if (MODE == 1) { const double n0 = dppmov<0x111>(o0), n1 = dppmov<0x111>(o1), n2 = dppmov<0x111>(o2); if (threadIdx.x & 15) { y0 = n0; y1 = n1; y2 = n2; } }  // row_shr:1
        if (MODE == 2) { const int k = s & 63; const bool me = (int)threadIdx.x == k; f0 = me ? o0 : f0; f1 = me ? o1 : f1; f2 = me ? o2 : f2; y0 = bcast(f0, k); y1 = bcast(f1, k); y2 = bcast(f2, k); }
        if (MODE == 3) { y0 = o0; y1 = o1; y2 = o2; }  // no cross-lane at all: pure dependent FMA chains
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = y0 + y1 + y2 + f0; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double *d; long long *c; (void)hipMalloc(&d, 128 * 8); (void)hipMalloc(&c, 8);
    long long cy; const int n = 10000;
    ripple<0><<<1, 64>>>(d, c, n); (void)hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost); printf("wave_shr:1      : %.1f cyc/step\n", cy / (double)n);
    ripple<1><<<1, 64>>>(d, c, n); (void)hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost); printf("row_shr:1       : %.1f cyc/step\n", cy / (double)n);
    ripple<2><<<1, 64>>>(d, c, n); (void)hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost); printf("latch+readlane  : %.1f cyc/step\n", cy / (double)n);
    ripple<3><<<1, 64>>>(d, c, n); (void)hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost); printf("no cross-lane   : %.1f cyc/step\n", cy / (double)n);
    return 0;
}
