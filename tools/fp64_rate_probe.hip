// fp64_rate_probe.hip -- issue cost of the VALU instructions the pixel kernel's component loop is made of, on gfx950.
// build: hipcc --offload-arch=gfx950 -O2 -o fp64_rate_probe tools/fp64_rate_probe.hip ; run on the GPU box.
// Every kernel runs ITER trips of 32 copies of one instruction on 8 independent register chains per lane (no dependency
// stalls), 2 waves per SIMD; reported: ns per wave-instruction per SIMD and the ratio to v_fma_f64.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define ITER 2000

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

#define KERNEL(name, ASM)                                                                    \
__global__ void __launch_bounds__(256) name(double *out, double seed) {                       \
    double a[8]; int ii[8];                                                                   \
    for (int k = 0; k < 8; ++k) { a[k] = seed + threadIdx.x * 1e-9 + k; ii[k] = threadIdx.x + k; } \
    double b = seed * 0.999, c = 1e-30;                                                      \
    for (int it = 0; it < ITER; ++it) {                                                       \
        REP32(ASM)                                                                           \
    }                                                                                        \
    double s = 0; for (int k = 0; k < 8; ++k) s += a[k] + ii[k];                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                          \
}

#define A_FMA(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_MUL(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define A_ADD(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c));
#define A_MAX(k) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c));
#define A_RNDNE(k) asm volatile("v_rndne_f64 %0, %0" : "+v"(a[k]));
#define A_CVT(k) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(ii[k]) : "v"(a[k]));
#define A_LDEXP(k) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a[k]) : "v"(ii[k]));
#define A_MOV64(k) asm volatile("v_mov_b64 %0, %1" : "=v"(a[k]) : "v"(b));
#define A_MOV32(k) asm volatile("v_mov_b32 %0, %1" : "=v"(ii[k]) : "v"(ii[(k + 1) & 7]));
#define A_AND(k) asm volatile("v_and_b32 %0, 63, %0" : "+v"(ii[k]));
#define A_LSHLADD(k) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(ii[k]) : "v"(ii[(k + 1) & 7]));
#define A_ASHR(k) asm volatile("v_ashrrev_i32 %0, 6, %0" : "+v"(ii[k]));
#define A_FMA32(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(ii[k]) : "v"(ii[(k + 1) & 7]), "v"(ii[(k + 2) & 7]));
#define A_PKFMA32(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_EXP32(k) asm volatile("v_exp_f32 %0, %0" : "+v"(ii[k]));
#define A_RCP64(k) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[k]));
#define A_READLANE(k) { int s_; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s_) : "v"(ii[k])); asm volatile("" :: "s"(s_)); }
// pairs: does an integer / move instruction next to FP64 work cost a full FP64 slot?
#define A_FMA_AND(k) asm volatile("v_fma_f64 %0, %0, %2, %3\n\tv_and_b32 %1, 63, %1" : "+v"(a[k]), "+v"(ii[k]) : "v"(b), "v"(c));
#define A_FMA_MUL(k) asm volatile("v_fma_f64 %0, %0, %1, %2\n\tv_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_FMAC(k) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_FMA_S(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "s"(seed), "v"(c));

KERNEL(k_fma, A_FMA) KERNEL(k_mul, A_MUL) KERNEL(k_add, A_ADD) KERNEL(k_max, A_MAX) KERNEL(k_rndne, A_RNDNE)
KERNEL(k_cvt, A_CVT) KERNEL(k_ldexp, A_LDEXP) KERNEL(k_mov64, A_MOV64) KERNEL(k_mov32, A_MOV32) KERNEL(k_and, A_AND)
KERNEL(k_lshladd, A_LSHLADD) KERNEL(k_ashr, A_ASHR) KERNEL(k_fma32, A_FMA32) KERNEL(k_pkfma32, A_PKFMA32)
KERNEL(k_exp32, A_EXP32) KERNEL(k_rcp64, A_RCP64) KERNEL(k_readlane, A_READLANE) KERNEL(k_fma_and, A_FMA_AND)
KERNEL(k_fma_mul, A_FMA_MUL) KERNEL(k_fmac, A_FMAC) KERNEL(k_fma_s, A_FMA_S)

struct Case { const char *name; void (*fn)(double *, double); int per; };

int main(int argc, char **argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 2;  // waves per SIMD (default 2; 1: what a lone wavefront on its SIMD gets)
    const int blocks = 256 * wps, threads = 256;   // wps workgroups of 4 waves per CU
    double *out;
    hipMalloc(&out, (size_t)blocks * threads * sizeof(double));
    Case cases[] = {{"v_fma_f64", k_fma, 1}, {"v_fmac_f64", k_fmac, 1}, {"v_fma_f64 (sgpr operand)", k_fma_s, 1}, {"v_mul_f64", k_mul, 1},
                    {"v_add_f64", k_add, 1}, {"v_max_f64", k_max, 1},
                    {"v_rndne_f64", k_rndne, 1}, {"v_cvt_i32_f64", k_cvt, 1}, {"v_ldexp_f64", k_ldexp, 1},
                    {"v_rcp_f64", k_rcp64, 1}, {"v_mov_b64", k_mov64, 1}, {"v_mov_b32", k_mov32, 1}, {"v_and_b32", k_and, 1},
                    {"v_lshl_add_u32", k_lshladd, 1}, {"v_ashrrev_i32", k_ashr, 1}, {"v_fma_f32", k_fma32, 1},
                    {"v_pk_fma_f32", k_pkfma32, 1}, {"v_exp_f32", k_exp32, 1}, {"v_readlane_b32", k_readlane, 1},
                    {"v_fma_f64 + v_and_b32 (pair)", k_fma_and, 2}, {"v_fma_f64 + v_mul_f64 (pair, dependent)", k_fma_mul, 2}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double base = 0;
    for (auto &c : cases) {
        hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(threads), 0, 0, out, 1.0000001);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(threads), 0, 0, out, 1.0000001);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        // wave-instructions per SIMD: 2 waves x ITER x 32 x per
        const double n = (double)wps * ITER * 32 * c.per;
        const double ns = best * 1e6 / n;
        if (base == 0) base = ns;
        printf("%-42s %8.3f ms  %6.3f ns per wave-instruction per SIMD  = %5.2f x v_fma_f64 (%.2f cycles at 2.4 GHz)\n", c.name, best, ns,
               ns / base, ns * 2.4);
    }
    return 0;
}
