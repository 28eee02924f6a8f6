#!/bin/bash
# Sustained shader clock under matrix load (run on the GPU box).  The chip clocks to its power budget: a kernel that keeps every matrix
# pipe busy does not run at the 2.4 GHz the peak figures assume.  Every CU runs 8 waves (2 per SIMD) of one instruction stream for a few
# milliseconds on random operands; wave 0 of every workgroup reads s_memtime (shader cycles) and the constant 100 MHz counter around it.
#   idle     s_sleep only                       f32      v_mfma_f32_16x16x4_f32 back to back (8 independent accumulators)
#   b16      v_mfma_f32_16x16x32_bf16            b32      v_mfma_f32_32x32x16_bf16
#   b16z     b16 on all-zero operands            b16l     b16 with one ds_read_b128 per two MFMAs (operands streamed from LDS)
#   b16h     b16, one wave per SIMD issuing at half rate (an s_sleep between groups)
cat > /tmp/dv.hip <<'SRC'
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512, 2) void k(int mode, int iters, const float *src, float *sink, unsigned long long *clk) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int t = threadIdx.x;
    for (int e = t; e < 8192; e += 512) lds[e] = src[e];
    __syncthreads();
    const long long c0 = clock64(), w0 = wall_clock64();
    f32x4 acc[8];
    for (int a = 0; a < 8; ++a) acc[a] = f32x4{0, 0, 0, 0};
    f32x16 big[2];
    for (int a = 0; a < 2; ++a) for (int e = 0; e < 16; ++e) big[a][e] = 0.f;
    bf16x8 A = *reinterpret_cast<const bf16x8 *>(lds + 4 * (t & 63)), B = *reinterpret_cast<const bf16x8 *>(lds + 1024 + 4 * (t & 63));
    if (mode == 4) { for (int e = 0; e < 8; ++e) { A[e] = 0; B[e] = 0; } }
    const float fa = lds[t & 63], fb = lds[64 + (t & 63)];
    if (mode == 0) {
        for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_sleep(64);
    } else if (mode == 1) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[a], 0, 0, 0);
    } else if (mode == 2 || mode == 4) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc[a], 0, 0, 0);
    } else if (mode == 3) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int a = 0; a < 4; ++a) big[a & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, big[a & 1], 0, 0, 0);
    } else if (mode == 5) {
        for (int it = 0; it < iters; ++it) {
            bf16x8 b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8 *>(lds + ((it * 4 + j) & 7) * 1024 + 4 * (t & 63) + (t >> 6) * 256 % 768);
#pragma unroll
            for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, b[a & 3], acc[a], 0, 0, 0);
        }
    } else if (mode == 6) {
        if ((t >> 6) < 4)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc[a], 0, 0, 0);
                __builtin_amdgcn_s_sleep(2);        // 128 cycles idle per 128 cycles of matrix issue
            }
    }
    f32x4 s = acc[0];
    for (int a = 1; a < 8; ++a) s += acc[a];
    for (int e = 0; e < 16; ++e) s[e & 3] += big[0][e] + big[1][e];
    if (s[0] == 123.456f) sink[t] = s[1] + s[2] + s[3];
    if (t == 0) { atomicAdd(&clk[0], (unsigned long long)(clock64() - c0)); atomicAdd(&clk[1], (unsigned long long)(wall_clock64() - w0)); }
}
int main() {
    std::vector<float> h(8192);
    srand(7);
    for (auto &v : h) v = (float)(rand() % 20001) / 20000.0f - 0.5f;
    float *src, *sink; unsigned long long *clk;
    hipMalloc(&src, 8192 * 4); hipMalloc(&sink, 4096); hipMalloc(&clk, 16);
    hipMemcpy(src, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    const char *names[] = {"idle", "f32", "b16", "b32", "b16z", "b16l", "b16h"};
    const int iters[] = {3000, 30000, 60000, 60000, 60000, 60000, 30000};
    const double mf[] = {0, 8, 8, 4, 8, 8, 8};       // MFMAs per iteration per wave
    const double flop[] = {0, 2048, 16384, 32768, 16384, 16384, 16384};
    for (int rep = 0; rep < 2; ++rep)
        for (int m = 0; m < 7; ++m) {
            hipMemset(clk, 0, 16);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            k<<<256, 512>>>(m, iters[m], src, sink, clk);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
            const double waves = (m == 6 ? 4 : 8) * 256.0, n = waves * iters[m] * mf[m];
            printf("%-5s %8.3f ms  shader clock %.2f GHz  %7.1f G MFMA/s  %7.1f TFLOP/s  SIMD cycles per MFMA %.1f\n", names[m], ms,
                   (double)c[0] / c[1] / 10.0, n / ms / 1e6, n * flop[m] / ms / 1e9, n ? (double)c[0] / c[1] * 1e8 * ms * 1e-3 * 1024.0 / n : 0.0);
        }
    return 0;
}
SRC
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/dv /tmp/dv.hip && /tmp/dv
