#!/bin/bash
# Does VALU / transcendental work hide under v_mfma_f32_16x16x4_f32 on gfx950?  (run on the GPU box)
# One MFMA followed by K plain VALU (v_fma_f32) and T transcendentals (v_exp_f32), all independent, order pinned with asm volatile;
# 1 or 2 waves per SIMD.  Prints ns and cycles (at the measured clock) per MFMA slot.
cat > /tmp/ov.hip <<'SRC'
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int K, int T, int M>
__global__ __launch_bounds__(256) void k(float *out, int reps) {
    f32x4 acc[8];
    float x[8], t[8];
    const float a = threadIdx.x * 1e-3f, b = 1.0f + a;
    for (int m = 0; m < 8; ++m) { acc[m] = f32x4{a, b, a, b}; x[m] = a + m; t[m] = b * 0.01f * m; }
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (M) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < K; ++j) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[(m + j) & 7]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < T; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(t[(m + j) & 7]));
        }
    }
    float s = 0.f;
    for (int m = 0; m < 8; ++m) s += acc[m][0] + acc[m][3] + x[m] + t[m];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int K, int T, int M>
void run(int wgs_per_cu) {
    float *out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    const int reps = 20000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<K, T, M>), dim3(grid), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<K, T, M>), dim3(grid), dim3(256), 0, 0, out, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mfma %d valu %d trans %d  waves/SIMD %d : %.2f ns per slot per wave = %.2f ns per slot per SIMD\n", M, K, T, wgs_per_cu, ms * 1e6 / (reps * 8.0), ms * 1e6 / (reps * 8.0) / wgs_per_cu);
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4, 6}) {
        run<0, 0, 1>(w); run<2, 0, 1>(w); run<4, 0, 1>(w); run<6, 0, 1>(w); run<8, 0, 1>(w);
        run<0, 1, 1>(w); run<2, 1, 1>(w); run<0, 2, 1>(w);
        run<4, 0, 0>(w); run<8, 0, 0>(w); run<0, 1, 0>(w); run<0, 2, 0>(w); run<2, 1, 0>(w);
    }
    return 0;
}
SRC
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ov /tmp/ov.hip && /tmp/ov
