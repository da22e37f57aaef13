// Device-to-host copy bandwidth on stream b while a chip-filling kernel runs on stream a (gpurun; hipcc tools/d2h_under_load_probe.hip).
// Question behind it: why does one context in four sweep 2000 sources through the host-pointer entry in 2.0 - 2.4 ms instead of 1.35?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void spin(long long ticks, double *out) {
    const long long t0 = wall_clock64();
    double x = threadIdx.x;
    while (wall_clock64() - t0 < ticks) { for (int k = 0; k < 64; ++k) x = __builtin_fma(x, 1.0000001, 1e-9); }
    if (x == 12345.0) out[0] = x;
}
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r)); return 1; } } while (0)
int main(int argc, char **argv) {
    const int waves_per_wg = argc > 1 ? atoi(argv[1]) : 1;     // 1: 64-thread workgroups like pixel_kernel
    const size_t N = 32u << 20;
    void *d, *h; double *dout;
    CK(hipMalloc(&d, N)); CK(hipHostMalloc(&h, N, hipHostMallocDefault)); CK(hipMalloc((void **)&dout, 8));
    const int NS = 10;
    std::vector<hipStream_t> s(NS);
    for (int i = 0; i < NS; ++i) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    hipEvent_t e0, e1, k0, k1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&k0)); CK(hipEventCreate(&k1));
    // first use of every stream, in order (the hardware queue is created on first use)
    for (int i = 0; i < NS; ++i) { CK(hipMemcpyAsync(h, d, 4096, hipMemcpyDeviceToHost, s[i])); CK(hipStreamSynchronize(s[i])); }
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < NS; ++b) {
            if (a == b) continue;
            float best = 1e9f, kms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(k0, s[a]));
                hipLaunchKernelGGL(spin, dim3(20000), dim3(64 * waves_per_wg), 0, s[a], (long long)(10000), dout);   // 20000 workgroups of 100 us: a chip-filling ~1 ms launch
                CK(hipEventRecord(k1, s[a]));
                CK(hipEventRecord(e0, s[b]));
                CK(hipMemcpyAsync(h, d, N, hipMemcpyDeviceToHost, s[b]));
                CK(hipEventRecord(e1, s[b]));
                CK(hipStreamSynchronize(s[a])); CK(hipStreamSynchronize(s[b]));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
                CK(hipEventElapsedTime(&kms, k0, k1));
            }
            printf("compute on stream %d (%.2f ms), copy on stream %d: %.1f GB/s (%.3f ms)\n", a, kms, b, N / best / 1e6, best);
        }
    return 0;
}
