// Does a short kernel on a second stream get wave slots while a long kernel with a deep workgroup backlog runs on
// the first?  With and without stream priorities.  (The optimiser wants to run the trust-region step of one half of
// a batch under the pixel kernel of the other half.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>

__global__ void __launch_bounds__(64, 2) busy(double *out, int iters) {   // ~VALU-bound, 2 waves / SIMD like pixel_kernel
    double f[8];
    for (int i = 0; i < 8; ++i) f[i] = 1.0 + i;
    const double x = 1.0 - 1e-9 * threadIdx.x;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int k = 0; k < 64; ++k) f[k & 7] = __builtin_fma(f[k & 7], x, 1e-3);
    double s = 0; for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(64) latency(double *out, int iters) {   // latency-bound chain, 1 wave per workgroup, 20 KB LDS
    __shared__ double buf[2560];
    buf[threadIdx.x] = threadIdx.x;
    __syncthreads();
    double s = 0;
    for (int it = 0; it < iters; ++it) { s += buf[(threadIdx.x * 7 + it) & 63]; __syncthreads(); buf[threadIdx.x] = s * 1e-3; __syncthreads(); }
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

static double run(bool use_priority) {
    hipStream_t a, b;
    int lo, hi; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (use_priority) { (void)hipStreamCreateWithPriority(&a, hipStreamNonBlocking, lo); (void)hipStreamCreateWithPriority(&b, hipStreamNonBlocking, hi); }
    else { (void)hipStreamCreateWithFlags(&a, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&b, hipStreamNonBlocking); }
    double *o1, *o2; (void)hipMalloc(&o1, 40000 * 64 * 8); (void)hipMalloc(&o2, 2000 * 64 * 8);
    hipEvent_t e0, e1, e2, e3; for (hipEvent_t *e : {&e0, &e1, &e2, &e3}) (void)hipEventCreate(e);
    // calibrate alone
    busy<<<28000, 64, 0, a>>>(o1, 400); latency<<<1000, 64, 0, b>>>(o2, 3000); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, a); busy<<<28000, 64, 0, a>>>(o1, 400); (void)hipEventRecord(e1, a); (void)hipStreamSynchronize(a);
    (void)hipEventRecord(e2, b); latency<<<1000, 64, 0, b>>>(o2, 3000); (void)hipEventRecord(e3, b); (void)hipStreamSynchronize(b);
    float t_busy, t_lat; (void)hipEventElapsedTime(&t_busy, e0, e1); (void)hipEventElapsedTime(&t_lat, e2, e3);
    // together: start busy, 100 us later launch the latency kernel on the other stream
    auto t0 = std::chrono::steady_clock::now();
    (void)hipEventRecord(e0, a); busy<<<28000, 64, 0, a>>>(o1, 400); (void)hipEventRecord(e1, a);
    std::this_thread::sleep_for(std::chrono::microseconds(100));
    (void)hipEventRecord(e2, b); latency<<<1000, 64, 0, b>>>(o2, 3000); (void)hipEventRecord(e3, b);
    (void)hipDeviceSynchronize();
    auto t1 = std::chrono::steady_clock::now();
    float both_busy, both_lat; (void)hipEventElapsedTime(&both_busy, e0, e1); (void)hipEventElapsedTime(&both_lat, e2, e3);
    float lat_end; (void)hipEventElapsedTime(&lat_end, e0, e3);
    printf("%s: alone busy %.3f ms, latency kernel %.3f ms | together: busy %.3f ms, latency kernel %.3f ms (ends %.3f ms after busy started), wall %.3f ms\n",
           use_priority ? "priorities " : "no priority", t_busy, t_lat, both_busy, both_lat, lat_end,
           std::chrono::duration<double, std::milli>(t1 - t0).count());
    return 0;
}
int main() { run(false); run(true); run(false); run(true); return 0; }
