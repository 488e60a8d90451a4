// Issue cost of instruction MIXES for a lone wave per SIMD on gfx950 (dev tool, round 6: what do the 34 % SQ_WAIT_ANY of newton_kernel consist of?).
// Every variant is a loop of hand-written (asm volatile: not reordered) instructions on independent registers; per wave the s_memtime ticks per loop body are
// reported, so that e.g. "16 v_fma_f64" against "16 v_fma_f64 interleaved with 16 v_add_u32" shows whether a non-fp64 instruction issues in the shadow of an fp64 one.
// Run under `rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS SQ_ACTIVE_INST_VALU` to see which counter holds the bubbles.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_mix tools/ubench/issue_mix.hip && /tmp/issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define F64(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b));
#define F64D(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[0]) : "v"(a), "v"(b));
#define M64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[i]) : "v"(a));
#define A64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[i]) : "v"(b));
#define F32(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(fa), "v"(fb));
#define I32(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(one));
#define MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(one));
#define CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(one));
#define ACW(i) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ac[i]) : "v"(one));
#define ACR(i) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(u[i]) : "a"(ac[i]));
#define SAL(i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc[i & 3]) : : "scc");  // (SCC clobber declared: the loop test lives in SCC)
#define DPP(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : "v"(one));
#define LDR(i) asm volatile("ds_read_b64 %0, %1" : "=v"(x[i]) : "v"(ldsaddr));
#define LDW() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define NOP(i) asm volatile("s_nop 0");

#define R16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#define P16(A, B) A(0) B(0) A(1) B(1) A(2) B(2) A(3) B(3) A(4) B(4) A(5) B(5) A(6) B(6) A(7) B(7) A(8) B(8) A(9) B(9) A(10) B(10) A(11) B(11) A(12) B(12) A(13) B(13) A(14) B(14) A(15) B(15)
#define T16(A, B, C) A(0) B(0) C(0) A(1) B(1) C(1) A(2) B(2) C(2) A(3) B(3) C(3) A(4) B(4) C(4) A(5) B(5) C(5) A(6) B(6) C(6) A(7) B(7) C(7) A(8) B(8) C(8) A(9) B(9) C(9) A(10) B(10) C(10) A(11) B(11) C(11) A(12) B(12) C(12) A(13) B(13) C(13) A(14) B(14) C(14) A(15) B(15) C(15)

#define KERNEL(name, BODY, NINST)                                                                                                  \
    __global__ __launch_bounds__(256) void name(double *out, long long *ticks, int n, double a, double b) {                         \
        __shared__ double lds[512];                                                                                                \
        double x[16]; float f[16]; unsigned u[16], ac[16]; unsigned sc[4] = {0, 0, 0, 0};                                           \
        const float fa = (float)a, fb = (float)b; const unsigned one = threadIdx.x | 1u;                                            \
        lds[threadIdx.x] = a; lds[threadIdx.x + 256] = b;                                                                           \
        const unsigned ldsaddr = (threadIdx.x & 63) * 8;                                                                           \
        for (int i = 0; i < 16; ++i) { x[i] = out[threadIdx.x] + i; f[i] = (float)x[i]; u[i] = threadIdx.x + i; ac[i] = u[i]; }       \
        __syncthreads();                                                                                                           \
        const long long t0 = __builtin_readcyclecounter();                                                                        \
        for (int it = 0; it < n; ++it) { BODY }                                                                                     \
        const long long t1 = __builtin_readcyclecounter();                                                                        \
        double s = 0; for (int i = 0; i < 16; ++i) s += x[i] + f[i] + u[i] + ac[i]; s += sc[0] + sc[1] + sc[2] + sc[3];               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                            \
        if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;                                            \
    }                                                                                                                              \
    static const int name##_ninst = NINST;

KERNEL(k_f64, R16(F64), 16)
KERNEL(k_mul64, R16(M64), 16)
KERNEL(k_add64, R16(A64), 16)
KERNEL(k_f32, R16(F32), 16)
KERNEL(k_i32, R16(I32), 16)
KERNEL(k_f64_dep, R16(F64D), 16)
KERNEL(k_f64_i32_alt, P16(F64, I32), 32)
KERNEL(k_f64_then_i32, R16(F64) R16(I32), 32)
KERNEL(k_f64_2i32, T16(F64, I32, MOV), 48)
KERNEL(k_f64_accw_alt, P16(F64, ACW), 32)
KERNEL(k_f64_accr_alt, P16(F64, ACR), 32)
KERNEL(k_accw, R16(ACW), 16)
KERNEL(k_f64_salu_alt, P16(F64, SAL), 32)
KERNEL(k_f64_cnd_alt, P16(F64, CND), 32)
KERNEL(k_f64_dpp_alt, P16(F64, DPP), 32)
KERNEL(k_f64_f32_alt, P16(F64, F32), 32)
KERNEL(k_f64_nop_alt, P16(F64, NOP), 32)
KERNEL(k_lds16_wait, R16(LDR) LDW(), 17)
KERNEL(k_lds16_f64, P16(LDR, F64) LDW(), 33)
KERNEL(k_i32_salu_alt, P16(I32, SAL), 32)

template <class K> void run(const char *name, K kern, int ninst, int blocks, double *d, long long *dt) {
    const int n = 4000;
    kern<<<blocks, 256>>>(d, dt, 10, 0.999, 0.001);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    kern<<<blocks, 256>>>(d, dt, n, 0.999, 0.001);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> t(blocks * 4);
    hipMemcpy(t.data(), dt, sizeof(long long) * t.size(), hipMemcpyDeviceToHost);
    double mean = 0; for (long long v : t) mean += (double)v; mean /= t.size();
    printf("%-18s blocks %4d: %7.2f ticks per loop body of %2d instructions = %5.2f ticks / instruction   (launch %.3f ms = %.2f cycles/instr at 2.4 GHz)\n", name, blocks, mean / n, ninst,
           mean / n / ninst, ms, ms * 1e-3 * 2.4e9 / ((double)n * ninst));
}
#define RUN(name) run(#name, name, name##_ninst, blocks, d, dt)
int main(int argc, char **argv) {
    double *d; long long *dt;
    hipMalloc(&d, 8 * 1024 * 256 * 8); hipMemset(d, 0, 8 * 1024 * 256 * 8); hipMalloc(&dt, 8 * 1024 * 4);
    for (int blocks : {1, 256, 512}) {  // one CU alone; one wave per SIMD chip-wide; two waves per SIMD
        RUN(k_f64); RUN(k_mul64); RUN(k_add64); RUN(k_f32); RUN(k_i32); RUN(k_f64_dep); RUN(k_f64_i32_alt); RUN(k_f64_then_i32); RUN(k_f64_2i32); RUN(k_f64_accw_alt);
        RUN(k_f64_accr_alt); RUN(k_accw); RUN(k_f64_salu_alt); RUN(k_f64_cnd_alt); RUN(k_f64_dpp_alt); RUN(k_f64_f32_alt); RUN(k_f64_nop_alt); RUN(k_lds16_wait); RUN(k_lds16_f64); RUN(k_i32_salu_alt);
    }
    return 0;
}
