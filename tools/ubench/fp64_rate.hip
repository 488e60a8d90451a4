// fp64 VALU throughput vs waves per SIMD (dev tool): independent v_fma_f64 streams, whole chip, hipEvent timing.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP> __global__ void k(double *out, int n, double a, double b) {
    double x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = out[threadIdx.x] + i;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) x[i] = fma(x[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP> void run(int waves_per_simd, double *d) {
    const int n = 20000, threads = 64 * 4 * waves_per_simd > 1024 ? 1024 : 64 * 4 * waves_per_simd;
    const int blocks_per_cu = (64 * 4 * waves_per_simd) / threads;
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<ILP><<<blocks, threads>>>(d, 10, 0.999, 0.001);
    hipEventRecord(e0);
    k<ILP><<<blocks, threads>>>(d, n, 0.999, 0.001);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * ILP * (double)n * blocks * threads;
    printf("ILP %d waves/SIMD %d: %.2f ms  %.1f TFLOP/s  (%.2f cycles/wave-instr at 2.4 GHz)\n", ILP, waves_per_simd, ms, fl / ms / 1e9,
           ms * 1e-3 * 2.4e9 / ((double)n * ILP * waves_per_simd));
}
int main() {
    double *d; hipMalloc(&d, 8 * 1024 * 1024 * 8); hipMemset(d, 0, 8 * 1024 * 1024 * 8);
    for (int w : {1, 2, 4}) { run<1>(w, d); run<4>(w, d); run<8>(w, d); }
    return 0;
}
