#!/bin/bash
# Achievable v_mfma_f32_16x16x4_f32 rate on this GPU: independent accumulators, no memory traffic (run on the GPU box).
cat > /tmp/mfma_peak.hip <<'SRC'
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
    for (int i = 1; i < NACC; ++i) s += acc[i];
    if (s[0] == 12345.f) out[threadIdx.x] = s[1];
}
template <int NACC>
void run(int wgs_per_cu, int threads) {
    float *out; hipMalloc(&out, 4096);
    const int iters = 20000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(threads), 0, 0, out, 10, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(threads), 0, 0, out, iters, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)grid * (threads / 64) * iters * 4.0 * NACC;
    printf("acc/wave %d  waves/SIMD %.1f : %.1f TFLOP/s  (%.2f ms)\n", NACC, grid * (threads / 64) / 1024.0, mf * 2048.0 / (ms * 1e-3) / 1e12, ms);
}
int main() {
    run<1>(1, 256); run<2>(1, 256); run<4>(1, 256); run<8>(1, 256);
    run<4>(2, 256); run<4>(4, 256); run<1>(4, 256); run<1>(8, 256); run<2>(8, 256);
    return 0;
}
SRC
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 /tmp/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
