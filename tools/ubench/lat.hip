// latency microbenchmarks for the chain design (dev tool): dependent v_fma_f64, fma+readlane round trip, LDS round trip
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double bcast(double v, int k) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, k); hi = __builtin_amdgcn_readlane(hi, k);
    return __hiloint2double(hi, lo);
}
__global__ void k_fma(double *out, long long *cyc, int n, double a, double b) {
    double x = out[threadIdx.x];
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) { x = fma(x, a, b); x = fma(x, a, b); x = fma(x, a, b); x = fma(x, a, b); }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_fma_rl(double *out, long long *cyc, int n, double a, double b) {
    double x = out[threadIdx.x];
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        x = fma(x, a, b); x = bcast(x, 0) + threadIdx.x; x = fma(x, a, b); x = bcast(x, 1) + threadIdx.x;
        x = fma(x, a, b); x = bcast(x, 2) + threadIdx.x; x = fma(x, a, b); x = bcast(x, 0) + threadIdx.x;
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_step(double *out, long long *cyc, int n, double a, double b) {  // 3 fma + 3 bcast like a chain step
    double x = out[threadIdx.x], y0 = a, y1 = b, y2 = a;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            double r = x; r = fma(-a, y0, r); r = fma(-b, y1, r); r = fma(-a, y2, r);
            y0 = bcast(r, 0); y1 = bcast(r, 1); y2 = bcast(r, 2); x = r;
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x + y0 + y1 + y2; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_lds(double *out, long long *cyc, int n) {
    __shared__ double s[256];
    s[threadIdx.x] = threadIdx.x; __syncthreads();
    int idx = threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    double acc = 0;
    for (int i = 0; i < n; ++i) { double v = s[idx]; idx = ((int)v + 1) & 63; acc += v; v = s[idx]; idx = ((int)v + 1) & 63; acc += v; v = s[idx]; idx = ((int)v+1)&63; acc += v; v = s[idx]; idx = ((int)v+1)&63; acc+=v; }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = acc; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double *d; long long *c; hipMalloc(&d, 64 * 8); hipMalloc(&c, 8); hipMemset(d, 0, 512);
    long long h; const int n = 10000;
    for (int rep = 0; rep < 2; ++rep) {
    k_fma<<<1, 64>>>(d, c, n, 0.999, 0.001); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); printf("dep fma f64 (64 lanes): %.1f cyc\n", (double)h / (4.0 * n));
    k_fma<<<1, 4>>>(d, c, n, 0.999, 0.001); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); printf("dep fma f64 (4 lanes): %.1f cyc\n", (double)h / (4.0 * n));
    k_fma_rl<<<1, 64>>>(d, c, n, 0.999, 0.001); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); printf("fma+bcast+add: %.1f cyc\n", (double)h / (4.0 * n));
    k_step<<<1, 4>>>(d, c, n, 0.3, 0.2); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); printf("chain step (3 fma + 3 bcast): %.1f cyc\n", (double)h / (4.0 * n));
    k_lds<<<1, 64>>>(d, c, n); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); printf("dependent LDS read (+cvt,add,and): %.1f cyc\n", (double)h / (4.0 * n));
    }
    return 0;
}
