// Cost of accumulating per-lane f64 values into LDS with ds_add_f64 (16 slots per entry: lanes l, l+16, l+32, l+48
// collide) next to FP64 VALU work; and whether the result is run-to-run deterministic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

template <int NF, int NA, int WAVES>
__global__ void __launch_bounds__(64, WAVES) mix(double *out, int iters) {
    __shared__ double acc[68 * 16];
    for (int i = threadIdx.x; i < 68 * 16; i += 64) acc[i] = 0.0;
    __syncthreads();
    double f[8];
    const double x = 1.0 - 1e-7 * (threadIdx.x + 1);
    for (int i = 0; i < 8; ++i) f[i] = 1.0 + 0.01 * i;
    double *slot = acc + (threadIdx.x & 15);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NF; ++k) f[k & 7] = __builtin_fma(f[k & 7], x, 1e-3);
#pragma unroll
        for (int k = 0; k < NA; ++k)
            __hip_atomic_fetch_add(slot + 16 * k, f[k & 7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += f[i];
    for (int k = threadIdx.x; k < 68 * 16; k += 64) s += acc[k] * 1e-3;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int NF, int NA, int WAVES>
static float run_mix(double *d_out, int blocks, int iters) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    mix<NF, NA, WAVES><<<blocks, 64>>>(d_out, 10);
    (void)hipEventRecord(a);
    mix<NF, NA, WAVES><<<blocks, 64>>>(d_out, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    double *d_out; (void)hipMalloc(&d_out, 256 * 12 * 64 * sizeof(double));
    const int iters = 5000;
    {
        const int blocks = 256 * 8;
        const float t_f = run_mix<256, 0, 2>(d_out, blocks, iters), t_a = run_mix<0, 68, 2>(d_out, blocks, iters),
                    t_fa = run_mix<256, 68, 2>(d_out, blocks, iters), t_fa2 = run_mix<256, 34, 2>(d_out, blocks, iters);
        printf("2 waves/SIMD, per iteration: 256 FMA %.0f ns | 68 ds_add_f64 %.0f ns | both %.0f ns | 256 FMA + 34 ds_add %.0f ns\n",
               t_f * 1e6 / iters, t_a * 1e6 / iters, t_fa * 1e6 / iters, t_fa2 * 1e6 / iters);
    }
    {
        const int blocks = 256 * 12;
        const float t_f = run_mix<256, 0, 3>(d_out, blocks, iters), t_a = run_mix<0, 68, 3>(d_out, blocks, iters),
                    t_fa = run_mix<256, 68, 3>(d_out, blocks, iters);
        printf("3 waves/SIMD, per iteration: 256 FMA %.0f ns | 68 ds_add_f64 %.0f ns | both %.0f ns\n",
               t_f * 1e6 / iters, t_a * 1e6 / iters, t_fa * 1e6 / iters);
    }
    // determinism: same launch 5 times, compare bits
    std::vector<double> ref(256 * 8 * 64), cur(256 * 8 * 64);
    bool same = true;
    for (int r = 0; r < 5; ++r) {
        mix<16, 68, 2><<<256 * 8, 64>>>(d_out, 300);
        (void)hipMemcpy(cur.data(), d_out, cur.size() * sizeof(double), hipMemcpyDeviceToHost);
        if (r == 0) ref = cur; else same = same && memcmp(ref.data(), cur.data(), cur.size() * sizeof(double)) == 0;
    }
    printf("bitwise reproducible over 5 runs: %s\n", same ? "yes" : "NO");
    return 0;
}
