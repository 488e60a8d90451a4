// Dev microbenchmark (gfx950): latency of the column-to-column chain of a lane-parallel banded substitution.  One wave, clock64 around N repetitions.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double bcast(double v, int k) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), k), hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
    return __hiloint2double(hi, lo);
}
template <int MODE> __global__ void k(double *out, long long *cyc, int n, double c0) {
    const int l = threadIdx.x;
    double w = 1.0 + l * 1e-3, w2 = 2.0, c = c0 + l * 1e-9, e1 = 0.5, e2 = 0.25, e3 = 0.125;
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int uu = 0; uu < 8; ++uu) {
            if constexpr (MODE == 0) { w = __builtin_fma(-c, w, w); }                              // dependent FMA chain
            if constexpr (MODE == 1) { const double y = bcast(w, uu); w = __builtin_fma(-c, y, w); }  // readlane -> fma
            if constexpr (MODE == 2) { const double y = bcast(w, uu); w = __builtin_fma(-c, y, w); w2 = __builtin_fma(-c, y, w2); }  // + 1 independent fma
            if constexpr (MODE == 3) {  // + masks as in lane_fma2 (2 cndmask + add + 2 fma)
                const double y = bcast(w, uu); double ca = l > uu ? c : 0.0; asm volatile("" : "+v"(ca)); const double cb = c - ca;
                w = __builtin_fma(-ca, y, w); w2 = __builtin_fma(-cb, y, w2);
            }
            if constexpr (MODE == 4) {  // readlane -> fma, plus 6 independent VALU
                const double y = bcast(w, uu); w = __builtin_fma(-c, y, w);
                e1 = __builtin_fma(e1, c, e2); e2 = __builtin_fma(e2, c, e3); e3 = __builtin_fma(e3, c, e1);
                asm volatile("" : "+v"(e1), "+v"(e2), "+v"(e3));
            }
            if constexpr (MODE == 5) {  // 9 independent accumulations from SGPR operands (matvec row): 2 readlane + 1 fma per term, accumulators independent of the readlane source
                const double y = bcast(w2, uu); w = __builtin_fma(-c, y, w);
            }
            if constexpr (MODE == 6) {  // v_readlane only (lo/hi), result folded by integer ops
                const double y = bcast(w, uu); w = __hiloint2double(__double2hiint(y), __double2loint(y) + 1);
            }
        }
    }
    const long long t1 = clock64();
    out[l] = w + w2 + e1 + e2 + e3;
    if (l == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char *name, double *out, long long *cyc) {
    const int n = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, n, 1e-3);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, n, 1e-3);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-60s %.1f cycles per step\n", name, (double)c / (n * 8.0));
}
int main() {
    double *out; long long *cyc; hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    run<0>("dependent v_fma_f64", out, cyc);
    run<1>("readlane x2 -> fma (chain)", out, cyc);
    run<2>("readlane x2 -> fma, + 1 independent fma", out, cyc);
    run<3>("readlane x2, 2 cndmask, add, 2 fma (lane_fma2)", out, cyc);
    run<4>("readlane x2 -> fma, + 3 independent fma", out, cyc);
    run<5>("readlane x2 (off chain) + fma accumulate", out, cyc);
    run<6>("readlane x2 -> int add (chain)", out, cyc);
    return 0;
}
