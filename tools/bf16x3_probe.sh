#!/bin/bash
# bf16 x 3 exact-split products on the fast matrix pipe (run on the GPU box): accuracy and rate.
#   An f32 value splits EXACTLY into three bf16 terms by truncation (8 + 8 + 8 significand bits, f32 exponent range).  A product of two
#   such values is the sum of nine bf16 x bf16 products (each exact in f32); the six with weight >= 2^-16 relative (i + j <= 2, terms
#   numbered from 0) carry everything down to 2^-24.  This probe measures, against a float64 evaluation of the SAME float32 operands:
#     f32      v_mfma_f32_16x16x4_f32 chain (what the product path uses today)
#     s6       six v_mfma_f32_16x16x32_bf16 products, one accumulator
#     s6h      six products, the five small ones in their own accumulator (added to the hi x hi accumulator at the end)
#     s3       three products (hi x hi, hi x mid, mid x hi): 2^-16 class, for scale
#   on (a) normal random operands, (b) a windowed DFT table row block (unreduced f32 angles, as the reference builds it) times
#   int16-range PCM after pre-emphasis.  Then the issue rate of the six-product group against eight f32 MFMAs (same K = 32), bare and
#   with VALU fillers beside it.
cat > /tmp/p.hip <<'SRC'
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __host__ inline unsigned fbits(float x) { union { float f; unsigned u; } v; v.f = x; return v.u; }
__device__ __host__ inline float bitsf(unsigned u) { union { float f; unsigned u; } v; v.u = u; return v.f; }
// exact three-way split by truncation: x = t0 + t1 + t2, each with <= 8 significand bits
__device__ __host__ inline void split3(float x, unsigned short &h0, unsigned short &h1, unsigned short &h2) {
    const float t0 = bitsf(fbits(x) & 0xffff0000u);
    const float r1 = x - t0;
    const float t1 = bitsf(fbits(r1) & 0xffff0000u);
    const float r2 = r1 - t1;
    h0 = fbits(t0) >> 16; h1 = fbits(t1) >> 16; h2 = fbits(r2) >> 16;
}

// A [16][K] row-major, B [K][16] (column j of B = one frame of K samples).  One wave.  mode: 0 f32, 1 s6, 2 s6h, 3 s3
__global__ void prod(const float *A, const float *B, int K, int mode, float *D) {
    const int l = threadIdx.x, r = l & 15, q = l >> 4;
    f32x4 acc = {0, 0, 0, 0}, lo = {0, 0, 0, 0};
    if (mode == 0) {
        for (int k = 0; k < K; k += 4)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + k + q], B[(k + q) * 16 + r], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 32) {
            bf16x8 a[3], b[3];
            for (int e = 0; e < 8; ++e) {
                unsigned short h0, h1, h2;
                split3(A[r * K + k + 8 * q + e], h0, h1, h2);
                a[0][e] = h0; a[1][e] = h1; a[2][e] = h2;
                split3(B[(k + 8 * q + e) * 16 + r], h0, h1, h2);
                b[0][e] = h0; b[1][e] = h1; b[2][e] = h2;
            }
            if (mode == 1) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
            } else if (mode == 2) {
                lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], lo, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
            }
        }
        for (int j = 0; j < 4; ++j) acc[j] += lo[j];
    }
    for (int j = 0; j < 4; ++j) D[(4 * q + j) * 16 + r] = acc[j];
}

static void accuracy(const char *name, const std::vector<float> &A, const std::vector<float> &B, int K, int trials) {
    float *dA, *dB, *dD;
    hipMalloc(&dA, 16 * K * 4); hipMalloc(&dB, K * 16 * 4); hipMalloc(&dD, 256 * 4);
    const char *mn[4] = {"f32 ", "s6  ", "s6h ", "s3  "};
    double worst[4] = {0, 0, 0, 0}, rms[4] = {0, 0, 0, 0};
    long n = 0;
    for (int t = 0; t < trials; ++t) {
        const float *a = A.data() + (size_t)t * 16 * K, *b = B.data() + (size_t)t * K * 16;
        hipMemcpy(dA, a, 16 * K * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, b, K * 16 * 4, hipMemcpyHostToDevice);
        for (int m = 0; m < 4; ++m) {
            float D[256];
            hipLaunchKernelGGL(prod, dim3(1), dim3(64), 0, 0, dA, dB, K, m, dD);
            hipMemcpy(D, dD, sizeof(D), hipMemcpyDeviceToHost);
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    double ref = 0, mag = 0;
                    for (int k = 0; k < K; ++k) { const double p = (double)a[i * K + k] * (double)b[k * 16 + j]; ref += p; mag += fabs(p); }
                    const double e = fabs((double)D[i * 16 + j] - ref) / (mag > 0 ? mag : 1);
                    if (e > worst[m]) worst[m] = e;
                    rms[m] += e * e;
                }
        }
        n += 256;
    }
    printf("%s K=%d (%d tiles): error / sum|a b| against float64\n", name, K, trials);
    for (int m = 0; m < 4; ++m) printf("   %s max %.3e  rms %.3e\n", mn[m], worst[m], sqrt(rms[m] / n));
    hipFree(dA); hipFree(dB); hipFree(dD);
}

// ---- rate: groups of six bf16 MFMAs (one K = 32 step of the split product) or eight f32 MFMAs, V plain VALU per MFMA beside them
template <int MODE, int V>
__global__ __launch_bounds__(256) void rate(float *out, int reps) {
    f32x4 acc[4];
    float x[8];
    const float fa = threadIdx.x * 1e-3f, fb = 1.0f + fa;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (short)(0x3f80 + threadIdx.x + e); b[e] = (short)(0x3f00 + e); }
    for (int m = 0; m < 4; ++m) acc[m] = f32x4{fa, fb, fa, fb};
    for (int m = 0; m < 8; ++m) x[m] = fa + m;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int m = 0; m < (MODE ? 6 : 8); ++m) {
            if (MODE == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m & 1]) : "v"(a), "v"(b));
            if (MODE == 2) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
            if (MODE == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m & 1]) : "v"(fa), "v"(fb));
#pragma unroll
            for (int j = 0; j < V; ++j) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[(m + j) & 7]) : "v"(fa), "v"(fb));
        }
    }
    float s = 0.f;
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][3];
    for (int m = 0; m < 8; ++m) s += x[m];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int V>
void run_rate(int wgs_per_cu) {
    float *out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    const int reps = 20000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((rate<MODE, V>), dim3(grid), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate<MODE, V>), dim3(grid), dim3(256), 0, 0, out, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const char *mn[3] = {"8 x f32 16x16x4 (2 acc)", "6 x bf16 16x16x32 (2 acc)", "6 x bf16 16x16x32 (4 acc)"};
    printf("%-28s + %d VALU per MFMA, %d waves/SIMD : %7.2f ns per K=32 step per SIMD\n", mn[MODE], V, wgs_per_cu, ms * 1e6 / reps / wgs_per_cu);
    hipFree(out);
}

int main() {
    srand(7);
    auto rnd = []() { return (float)((rand() + 0.5) / (RAND_MAX + 1.0)); };
    auto gauss = [&]() { return sqrtf(-2.f * logf(rnd())) * cosf(6.2831853f * rnd()); };
    for (int K : {64, 256, 512}) {
        const int T = 24;
        std::vector<float> A((size_t)T * 16 * K), B((size_t)T * K * 16);
        for (auto &v : A) v = gauss();
        for (auto &v : B) v = gauss();
        accuracy("normal x normal", A, B, K, T);
    }
    {   // a windowed DFT table as the reference builds it (float32 angles 2 pi f t / n_fft, unreduced; hann window 400 in 512) times
        // pre-emphasised int16-range samples: 16 bins per tile, cos rows
        const int K = 416, T = 16;      // 400 taps padded to a multiple of 32 with zeros
        std::vector<float> A((size_t)T * 16 * K, 0.f), B((size_t)T * K * 16, 0.f);
        for (int t = 0; t < T; ++t)
            for (int i = 0; i < 16; ++i)
                for (int k = 0; k < 400; ++k) {
                    const float w = 0.5f - 0.5f * cosf(6.2831853f * k / 399.f);
                    const float ang = 6.2831853f * (float)(t * 16 + i) * (float)(k + 56) / 512.f;
                    A[((size_t)t * 16 + i) * K + k] = (t & 1 ? -sinf(ang) : cosf(ang)) * w;
                }
        for (int t = 0; t < T; ++t)
            for (int j = 0; j < 16; ++j) {
                float prev = 0.f;
                const float amp = (j & 1) ? 30.f : 3000.f;
                for (int k = 0; k < 400; ++k) {
                    const float s = roundf(gauss() * amp + 2000.f * sinf(0.05f * (j + 1) * k));
                    B[(size_t)t * K * 16 + (size_t)k * 16 + j] = (s - 0.97f * prev) * (1.f / 32768.f);
                    prev = s;
                }
            }
        accuracy("DFT table x pre-emphasised PCM", A, B, K, T);
    }
    for (int w : {1, 2, 4}) {
        run_rate<0, 0>(w); run_rate<1, 0>(w); run_rate<2, 0>(w);
        run_rate<1, 1>(w); run_rate<1, 2>(w); run_rate<1, 3>(w); run_rate<1, 4>(w); run_rate<1, 6>(w);
        run_rate<2, 2>(w); run_rate<2, 4>(w);
        run_rate<0, 2>(w);
    }
    return 0;
}
SRC
/opt/rocm/bin/hipcc -w --offload-arch=gfx950 -O3 -o /tmp/p /tmp/p.hip && /tmp/p
