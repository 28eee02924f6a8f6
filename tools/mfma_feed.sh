#!/bin/bash
# What feeding the f32 MFMA costs (run on the GPU box): one ds_read_b32 per MFMA and/or one 16-B global load per
# 4*MT MFMAs, the shape of the gemm helpers, at several occupancies.
cat > /tmp/mfma_feed.hip <<'SRC'
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// MODE bit0: activations from LDS (one ds_read_b32 per MFMA); bit1: weights from global (one float4 per 4*MT MFMAs);
// bit2 (with bit0): k4-interleaved activations, one ds_read_b128 per 4 MFMAs
template <int MT, int MODE>
__global__ __launch_bounds__(512) void k(const float *__restrict__ W, float *out, int kb, int reps) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, i = lane & 15;
    for (int e = threadIdx.x; e < 256 * 68; e += blockDim.x) lds[e] = 1e-3f * (e & 127);
    __syncthreads();
    f32x4 acc[MT];
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *wl = W + (size_t)wave * kb * 256 + lane * 4;
    const float *ap = lds + 4 * q * 68 + i;
    for (int r = 0; r < reps; ++r) {
        f32x4 wc = (MODE & 2) ? *reinterpret_cast<const f32x4 *>(wl) : f32x4{1.f, 2.f, 3.f, 4.f};
        for (int S = 0; S < kb; ++S) {
            const int Sn = S + 1 < kb ? S + 1 : S;
            f32x4 wn = (MODE & 2) ? *reinterpret_cast<const f32x4 *>(wl + 256 * Sn) : wc;
            const float *aps = ap + 16 * S * 68;
            f32x4 a4[MT];
            if (MODE & 4) {
#pragma unroll
                for (int m = 0; m < MT; ++m) a4[m] = *reinterpret_cast<const f32x4 *>(lds + (((4 * S + q) * 64 + 16 * (m & 3) + i) * 4));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float av[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) av[m] = (MODE & 4) ? a4[m][j] : ((MODE & 1) ? aps[j * 68 + 16 * (m & 3)] : 1.0f + m);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], wc[j], acc[m], 0, 0, 0);
            }
            wc = wn;
        }
    }
    f32x4 s = acc[0];
    for (int m = 1; m < MT; ++m) s += acc[m];
    if (s[0] == 12345.f) out[threadIdx.x] = s[1];
}
template <int MT, int MODE>
void run(int wgs_per_cu) {
    const int kb = 16, reps = 400, grid = 256 * wgs_per_cu, threads = 512;
    float *W, *out; hipMalloc(&W, (size_t)8 * kb * 256 * 4); hipMemset(W, 0, (size_t)8 * kb * 256 * 4); hipMalloc(&out, 4096);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MT, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 68 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MT, MODE>), dim3(grid), dim3(threads), 256 * 68 * 4, 0, W, out, kb, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MT, MODE>), dim3(grid), dim3(threads), 256 * 68 * 4, 0, W, out, kb, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)grid * 8 * reps * kb * 4.0 * MT;
    printf("MT %d  lds %d  gweights %d  WGs/CU %d (waves/SIMD %d): %6.1f TFLOP/s\n", MT, MODE & 1, (MODE >> 1) & 1, wgs_per_cu, 2 * wgs_per_cu,
           mf * 2048.0 / (ms * 1e-3) / 1e12);
}
int main() {
    run<4, 0>(1); run<4, 1>(1); run<4, 2>(1); run<4, 3>(1); run<4, 3>(2);
    run<2, 1>(1); run<2, 3>(1); run<2, 3>(2); run<2, 3>(3);
    run<1, 1>(1); run<1, 3>(1); run<1, 3>(2); run<1, 3>(3);
    printf("-- k4-interleaved activations (ds_read_b128)\n");
    run<4, 5>(1); run<4, 7>(1); run<4, 7>(2); run<2, 7>(1); run<2, 7>(2); run<2, 7>(3); run<1, 7>(1); run<1, 7>(2); run<1, 7>(3);
    return 0;
}
SRC
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w /tmp/mfma_feed.hip -o /tmp/mfma_feed && /tmp/mfma_feed
