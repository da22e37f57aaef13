// tred_probe.hip -- what one tridiagonalisation of the trust-region sub-problem costs, outside the optimiser kernels.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/variants/tred_probe tools/tred_probe.hip ; run on the GPU box.
// The reduction as optim_step_kernel runs it (one wavefront, the reflections of g applied as they are formed) against the
// arrangement of optim_fused_kernel (wavefront 0 reduces, wavefront 1 applies the reflections afterwards) on the same random
// symmetric 41 x 41 matrices: shader clocks per reduction with 1 workgroup on the chip (pure latency) and with 2 per CU,
// first call and repeated calls; and that the two agree bit for bit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "../celeste.jl_amd/csrc/elbo_kernels.h"
#include "../celeste.jl_amd/csrc/optim_kernels.h"

__device__ __noinline__ TredOut tred_w1(double *A, double g, int ln) {
    double v = g, td, ev, hvv;
    tred_reg<true>(A, v, td, ev, hvv, ln);
    return TredOut{td, ev, hvv, v};
}

template <int NT>
__global__ void __launch_bounds__(NT) probe(const double *__restrict__ H, double *__restrict__ out, long long *__restrict__ cyc, int reps) {
    __shared__ double sA[LDA * NF];
    __shared__ double shv[64];
    const int tid = threadIdx.x, ln = tid & 63, part = tid >> 6;
    const double *Hb = H + (size_t)(blockIdx.x % 64) * NF * NF;
    long long first = 0, total = 0;
    TredOut T = {0, 0, 0, 0};
    const double g = 1.0 + 0.01 * ln;
    for (int rep = 0; rep < reps; ++rep) {
        for (int k = tid; k < NF * NF; k += NT) { const int j = k / NF; sA[(k - j * NF) + LDA * j] = Hb[k]; }
        __syncthreads();
        const long long t0 = clock64();
        if constexpr (NT == 64) T = tred_w1(sA, g, ln);
        else if (part == 0) T = tred_only(sA, ln);
        const long long t1 = clock64();
        if constexpr (NT != 64) {
            if (part == 0) shv[ln] = T.hv;
            __syncthreads();
            if (part == 1) shv[ln] = tred_qtv(sA, shv[ln], g, ln);
            __syncthreads();
            T.gt = shv[ln];
        }
        __syncthreads();
        if (rep == 0) first = t1 - t0; else total += t1 - t0;
    }
    if (part == 0 && ln < NF) {
        double *o = out + (size_t)blockIdx.x * 4 * NF;
        o[ln] = T.td; o[NF + ln] = T.te; o[2 * NF + ln] = T.hv; o[3 * NF + ln] = T.gt;
    }
    if (tid == 0) { cyc[2 * blockIdx.x] = first; cyc[2 * blockIdx.x + 1] = reps > 1 ? total / (reps - 1) : 0; }
}

int main() {
    const int NM = 64;
    std::vector<double> H((size_t)NM * NF * NF);
    srand(7);
    for (int m = 0; m < NM; ++m) {
        double *h = H.data() + (size_t)m * NF * NF;
        for (int i = 0; i < NF; ++i)
            for (int j = 0; j <= i; ++j) {
                const double r = (rand() / (double)RAND_MAX - 0.5) * (i == j ? 20.0 : 2.0);
                h[i + NF * j] = r; h[j + NF * i] = r;
            }
    }
    double *dH, *dout1, *dout4;
    long long *dc;
    const int maxb = 512;
    hipMalloc(&dH, H.size() * sizeof(double));
    hipMalloc(&dout1, (size_t)maxb * 4 * NF * sizeof(double));
    hipMalloc(&dout4, (size_t)maxb * 4 * NF * sizeof(double));
    hipMalloc(&dc, (size_t)maxb * 2 * sizeof(long long));
    hipMemcpy(dH, H.data(), H.size() * sizeof(double), hipMemcpyHostToDevice);
    std::vector<long long> c(maxb * 2);
    std::vector<double> o1((size_t)maxb * 4 * NF), o4((size_t)maxb * 4 * NF);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nt : {64, 256}) {   // calibration: wall time of 200 back-to-back reductions by one workgroup
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(e0);
            if (nt == 64) hipLaunchKernelGGL(probe<64>, dim3(1), dim3(64), 0, 0, dH, dout1, dc, 200);
            else hipLaunchKernelGGL(probe<256>, dim3(1), dim3(256), 0, 0, dH, dout4, dc, 200);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        hipMemcpy(c.data(), dc, 2 * sizeof(long long), hipMemcpyDeviceToHost);
        printf("%s: %.2f us per reduction by the host's clock (with the copy of H into LDS), %lld clock64 ticks\n", nt == 64 ? "with the reflections of g" : "reduction alone      ",
               best * 1000.0 / 200, c[1]);
    }
    for (int blocks : {1, 64, 512}) {
        for (int reps : {1, 8}) {
            hipMemset(dc, 0, (size_t)maxb * 2 * sizeof(long long));
            hipMemset(dout1, 0, (size_t)maxb * 4 * NF * sizeof(double)); hipMemset(dout4, 0, (size_t)maxb * 4 * NF * sizeof(double));
            hipLaunchKernelGGL(probe<64>, dim3(blocks), dim3(64), 0, 0, dH, dout1, dc, reps);
            { hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) printf("probe<64>: %s\n", hipGetErrorString(e)); }
            hipMemcpy(c.data(), dc, (size_t)blocks * 2 * sizeof(long long), hipMemcpyDeviceToHost);
            double f1 = 0, r1 = 0;
            for (int b = 0; b < blocks; ++b) { f1 += c[2 * b]; r1 += c[2 * b + 1]; }
            hipLaunchKernelGGL(probe<256>, dim3(blocks), dim3(256), 0, 0, dH, dout4, dc, reps);
            hipMemcpy(c.data(), dc, (size_t)blocks * 2 * sizeof(long long), hipMemcpyDeviceToHost);
            double f4 = 0, r4 = 0;
            for (int b = 0; b < blocks; ++b) { f4 += c[2 * b]; r4 += c[2 * b + 1]; }
            printf("%3d workgroups, %d calls each: with reflections first %7.0f repeat %7.0f | reduction alone first %7.0f repeat %7.0f  (clock64 ticks per reduction)\n",
                   blocks, reps, f1 / blocks, r1 / blocks, f4 / blocks, r4 / blocks);
        }
        hipMemcpy(o1.data(), dout1, (size_t)blocks * 4 * NF * sizeof(double), hipMemcpyDeviceToHost);
        hipMemcpy(o4.data(), dout4, (size_t)blocks * 4 * NF * sizeof(double), hipMemcpyDeviceToHost);
        {   // independent check: the reduction is a similarity transformation -- trace and Frobenius norm of H survive it
            int nbad = 0, ndiff = 0;
            double worst = 0;
            for (int b = 0; b < blocks; ++b) {
                const double *h = H.data() + (size_t)(b % NM) * NF * NF, *o = o4.data() + (size_t)b * 4 * NF;
                double trH = 0, frH = 0, trT = 0, frT = 0;
                for (int i = 0; i < NF; ++i) { trH += h[i + NF * i]; for (int j = 0; j < NF; ++j) frH += h[i + NF * j] * h[i + NF * j]; }
                for (int i = 0; i < NF; ++i) { trT += o[i]; frT += o[i] * o[i] + 2 * o[NF + i] * o[NF + i]; }
                const double err = fmax(fabs(trT - trH) / (fabs(trH) + 1.0), fabs(frT - frH) / frH);
                if (!(err < 1e-12)) ++nbad;
                if (err > worst) worst = err;
                if (memcmp(o, o1.data() + (size_t)b * 4 * NF, 4 * NF * sizeof(double)) != 0) ++ndiff;
            }
            printf("    trace and Frobenius norm of T against H: %d of %d workgroups off by more than 1e-12 (worst %.2e); %d differ between the arrangements\n", nbad, blocks, worst, ndiff);
        }
        printf("    td / te / hv / Q'g of the two arrangements agree bit for bit: %s\n", memcmp(o1.data(), o4.data(), (size_t)blocks * 4 * NF * sizeof(double)) == 0 ? "yes" : "NO");
    }
    return 0;
}
