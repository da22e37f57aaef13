// Probe of the lane -> element maps of v_mfma_f64_4x4x4f64 (4 blocks) and v_mfma_f64_16x16x4f64 on gfx950, and of
// their issue cost next to FP64 VALU work.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/mfma_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void probe44(int *cb, int *ca) {   // cb[s * 64 + lane] = does B of lane s reach D of `lane` (A = 1)?
    const int lane = threadIdx.x;
    for (int s = 0; s < 64; ++s) {
        double d = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, lane == s ? 1.0 : 0.0, 0.0, 0, 0, 0);
        cb[s * 64 + lane] = (int)d;
        d = __builtin_amdgcn_mfma_f64_4x4x4f64(lane == s ? 1.0 : 0.0, 1.0, 0.0, 0, 0, 0);
        ca[s * 64 + lane] = (int)d;
    }
}
__global__ void probe16(int *cb, int *ca) {   // 4 results per lane: cb[(s * 64 + lane) * 4 + r]
    const int lane = threadIdx.x;
    for (int s = 0; s < 64; ++s) {
        d4 z = {0, 0, 0, 0};
        d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0, lane == s ? 1.0 : 0.0, z, 0, 0, 0);
        for (int r = 0; r < 4; ++r) cb[(s * 64 + lane) * 4 + r] = (int)d[r];
        d = __builtin_amdgcn_mfma_f64_16x16x4f64(lane == s ? 1.0 : 0.0, 1.0, z, 0, 0, 0);
        for (int r = 0; r < 4; ++r) ca[(s * 64 + lane) * 4 + r] = (int)d[r];
    }
}

// throughput: per loop iteration NF dependent-chain FMAs on 8 chains + NM 4x4x4 MFMAs on 8 accumulators
template <int NF, int NM>
__global__ void __launch_bounds__(64, 2) mix(double *out, int iters) {
    double f[8], m[8];
    const double x = 1.0 + 1e-9 * threadIdx.x, mask = (threadIdx.x & 3) == 0 ? 1.0 : 0.0;
    for (int i = 0; i < 8; ++i) { f[i] = i; m[i] = 0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NF; ++k) f[k & 7] = __builtin_fma(f[k & 7], x, 1e-3);
#pragma unroll
        for (int k = 0; k < NM; ++k) m[k & 7] = __builtin_amdgcn_mfma_f64_4x4x4f64(mask, f[k & 7], m[k & 7], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += f[i] + m[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int NF, int NM>
static float run_mix(double *d_out, int blocks, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    mix<NF, NM><<<blocks, 64>>>(d_out, 10);
    hipEventRecord(a);
    mix<NF, NM><<<blocks, 64>>>(d_out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    int *cb, *ca;
    hipMalloc(&cb, 64 * 64 * 4 * sizeof(int)); hipMalloc(&ca, 64 * 64 * 4 * sizeof(int));
    std::vector<int> hb(64 * 64 * 4), ha(64 * 64 * 4);
    probe44<<<1, 64>>>(cb, ca);
    hipMemcpy(hb.data(), cb, 64 * 64 * sizeof(int), hipMemcpyDeviceToHost);
    hipMemcpy(ha.data(), ca, 64 * 64 * sizeof(int), hipMemcpyDeviceToHost);
    printf("v_mfma_f64_4x4x4f64: D[lane] = sum over (a, b) lane pairs listed\n");
    for (int lane = 0; lane < 64; ++lane) {
        printf("D lane %2d: B lanes {", lane);
        for (int s = 0; s < 64; ++s) if (hb[s * 64 + lane]) printf(" %d", s);
        printf(" }  A lanes {");
        for (int s = 0; s < 64; ++s) if (ha[s * 64 + lane]) printf(" %d", s);
        printf(" }\n");
    }
    probe16<<<1, 64>>>(cb, ca);
    hipMemcpy(hb.data(), cb, 64 * 64 * 4 * sizeof(int), hipMemcpyDeviceToHost);
    hipMemcpy(ha.data(), ca, 64 * 64 * 4 * sizeof(int), hipMemcpyDeviceToHost);
    printf("v_mfma_f64_16x16x4f64 (lanes 0, 1, 17, 63 only)\n");
    for (int lane : {0, 1, 17, 63}) for (int r = 0; r < 4; ++r) {
        printf("D lane %2d reg %d: B lanes {", lane, r);
        for (int s = 0; s < 64; ++s) if (hb[(s * 64 + lane) * 4 + r]) printf(" %d", s);
        printf(" }  A lanes {");
        for (int s = 0; s < 64; ++s) if (ha[(s * 64 + lane) * 4 + r]) printf(" %d", s);
        printf(" }\n");
    }
    double *d_out; hipMalloc(&d_out, 256 * 8 * 64 * sizeof(double));
    const int blocks = 256 * 8, iters = 20000;   // 2 waves per SIMD
    const float t_f = run_mix<64, 0>(d_out, blocks, iters);
    const float t_m = run_mix<0, 16>(d_out, blocks, iters);
    const float t_fm = run_mix<64, 16>(d_out, blocks, iters);
    const float t_fm8 = run_mix<64, 8>(d_out, blocks, iters);
    printf("per iteration and wave (2 waves / SIMD): 64 FMA %.1f ns | 16 MFMA 4x4x4 %.1f ns | 64 FMA + 16 MFMA %.1f ns | 64 FMA + 8 MFMA %.1f ns\n",
           t_f * 1e6 / iters, t_m * 1e6 / iters, t_fm * 1e6 / iters, t_fm8 * 1e6 / iters);
    return 0;
}
