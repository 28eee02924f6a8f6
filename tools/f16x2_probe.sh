#!/bin/bash
# fp16 x 2 split products on the fast matrix pipe (run on the GPU box): accuracy and rate, beside the bf16 x 3 figures of
# tools/bf16x3_probe.sh.
#   A float32 value x is approximated by TWO fp16 terms taken with round-to-nearest:  h0 = RN16(x),  h1 = RN16((x - h0) * 2^11).
#   RN leaves a SIGNED residual, so the two 11-bit significands cover 11 + 1 + 11 = 23 bits: |x - h0 - h1 2^-11| <= 2^-23 |x| (one
#   float32 ulp at worst, 2^-24.8 rms).  The residual term is stored scaled by 2^11, so it keeps its 11 bits wherever h0 is a normal
#   number (no fp16 subnormals for |x| >= 2^-14), and the products h0 h1' + h1' h0 go into an accumulator of their own that joins the
#   h0 h0 accumulator with one multiply by 2^-11 at the end.  THREE v_mfma_f32_16x16x32_f16 per K = 32 step (the dropped h1 h1 term is
#   <= 2^-22, 2^-25.6 rms, of the product), against SIX bf16 MFMAs for the exact three-way bf16 split and EIGHT f32 MFMAs.
#   Measured against a float64 evaluation of the SAME float32 operands:
#     f32      v_mfma_f32_16x16x4_f32 chain
#     s6h      bf16 x 3, six products, hi / lo accumulators (what round 4 ships)
#     h3       fp16 x 2, three products, hi / mid accumulators
#     h4       fp16 x 2, four products (h1 h1 into a third accumulator)
#     h3t      h3 with TRUNCATED (round-toward-zero) terms, for scale: what v_cvt_pkrtz_f16_f32 would give
cat > /tmp/p.hip <<'SRC'
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __host__ inline unsigned fbits(float x) { union { float f; unsigned u; } v; v.f = x; return v.u; }
__device__ __host__ inline float bitsf(unsigned u) { union { float f; unsigned u; } v; v.u = u; return v.f; }
__device__ inline void split3(float x, unsigned short &h0, unsigned short &h1, unsigned short &h2) {
    const float t0 = bitsf(fbits(x) & 0xffff0000u);
    const float r1 = x - t0;
    const float t1 = bitsf(fbits(r1) & 0xffff0000u);
    const float r2 = r1 - t1;
    h0 = fbits(t0) >> 16; h1 = fbits(t1) >> 16; h2 = fbits(r2) >> 16;
}
__device__ inline void split2(float x, _Float16 &h0, _Float16 &h1, bool trunc) {
    if (!trunc) {
        h0 = (_Float16)x;
        h1 = (_Float16)((x - (float)h0) * 2048.f);
    } else {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 t = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(x, 0.f));
        h0 = t[0];
        h2 u = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz((x - (float)h0) * 2048.f, 0.f));
        h1 = u[0];
    }
}

// A [16][K] row-major, B [K][16].  One wave.  mode: 0 f32, 1 s6h, 2 h3, 3 h4, 4 h3t
__global__ void prod(const float *A, const float *B, int K, int mode, float *D) {
    const int l = threadIdx.x, r = l & 15, q = l >> 4;
    f32x4 acc = {0, 0, 0, 0}, lo = {0, 0, 0, 0}, ll = {0, 0, 0, 0};
    if (mode == 0) {
        for (int k = 0; k < K; k += 4)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + k + q], B[(k + q) * 16 + r], acc, 0, 0, 0);
    } else if (mode == 1) {
        for (int k = 0; k < K; k += 32) {
            bf16x8 a[3], b[3];
            for (int e = 0; e < 8; ++e) {
                unsigned short h0, h1, h2;
                split3(A[r * K + k + 8 * q + e], h0, h1, h2);
                a[0][e] = h0; a[1][e] = h1; a[2][e] = h2;
                split3(B[(k + 8 * q + e) * 16 + r], h0, h1, h2);
                b[0][e] = h0; b[1][e] = h1; b[2][e] = h2;
            }
            lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], lo, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
        }
        for (int j = 0; j < 4; ++j) acc[j] += lo[j];
    } else {
        for (int k = 0; k < K; k += 32) {
            f16x8 a[2], b[2];
            for (int e = 0; e < 8; ++e) {
                _Float16 h0, h1;
                split2(A[r * K + k + 8 * q + e], h0, h1, mode == 4);
                a[0][e] = h0; a[1][e] = h1;
                split2(B[(k + 8 * q + e) * 16 + r], h0, h1, mode == 4);
                b[0][e] = h0; b[1][e] = h1;
            }
            if (mode == 3) ll = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[1], ll, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[0], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[1], lo, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], acc, 0, 0, 0);
        }
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(fmaf(ll[j], 1.f / 2048.f, lo[j]), 1.f / 2048.f, acc[j]);
    }
    for (int j = 0; j < 4; ++j) D[(4 * q + j) * 16 + r] = acc[j];
}

static void accuracy(const char *name, const std::vector<float> &A, const std::vector<float> &B, int K, int trials) {
    float *dA, *dB, *dD;
    hipMalloc(&dA, 16 * K * 4); hipMalloc(&dB, K * 16 * 4); hipMalloc(&dD, 256 * 4);
    const char *mn[5] = {"f32 ", "s6h ", "h3  ", "h4  ", "h3t "};
    double worst[5] = {0}, rms[5] = {0}, bias[5] = {0};
    long n = 0;
    for (int t = 0; t < trials; ++t) {
        const float *a = A.data() + (size_t)t * 16 * K, *b = B.data() + (size_t)t * K * 16;
        hipMemcpy(dA, a, 16 * K * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, b, K * 16 * 4, hipMemcpyHostToDevice);
        for (int m = 0; m < 5; ++m) {
            float D[256];
            hipLaunchKernelGGL(prod, dim3(1), dim3(64), 0, 0, dA, dB, K, m, dD);
            hipMemcpy(D, dD, sizeof(D), hipMemcpyDeviceToHost);
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    double ref = 0, mag = 0;
                    for (int k = 0; k < K; ++k) { const double p = (double)a[i * K + k] * (double)b[k * 16 + j]; ref += p; mag += fabs(p); }
                    const double e = ((double)D[i * 16 + j] - ref) / (mag > 0 ? mag : 1);
                    if (fabs(e) > worst[m]) worst[m] = fabs(e);
                    rms[m] += e * e;
                    bias[m] += e;
                }
        }
        n += 256;
    }
    printf("%s K=%d (%d tiles): error / sum|a b| against float64\n", name, K, trials);
    for (int m = 0; m < 5; ++m) printf("   %s max %.3e  rms %.3e  mean %+.3e\n", mn[m], worst[m], sqrt(rms[m] / n), bias[m] / n);
    hipFree(dA); hipFree(dB); hipFree(dD);
}

// ---- rate: N MFMAs per K = 32 step (MODE 0: 8 x f32, 1: 6 x bf16, 2: 3 x f16), V plain VALU per MFMA beside them
template <int MODE, int V>
__global__ __launch_bounds__(256) void rate(float *out, int reps, unsigned long long *clk) {
    f32x4 acc[4];
    float x[8];
    const float fa = threadIdx.x * 1e-3f, fb = 1.0f + fa;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (short)(0x3c00 + 37 * threadIdx.x + e); b[e] = (short)(0x3800 + 11 * e + threadIdx.x); }
    for (int m = 0; m < 4; ++m) acc[m] = f32x4{fa, fb, fa, fb};
    for (int m = 0; m < 8; ++m) x[m] = fa + m;
    const long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int m = 0; m < (MODE == 0 ? 8 : MODE == 1 ? 6 : 3); ++m) {
            if (MODE == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m & 1]) : "v"(a), "v"(b));
            if (MODE == 2) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m & 1]) : "v"(a), "v"(b));
            if (MODE == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m & 1]) : "v"(fa), "v"(fb));
#pragma unroll
            for (int j = 0; j < V; ++j) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[(m + j) & 7]) : "v"(fa), "v"(fb));
        }
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = wall_clock64() - w0; }
    float s = 0.f;
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][3];
    for (int m = 0; m < 8; ++m) s += x[m];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int V>
void run_rate(int wgs_per_cu) {
    float *out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    unsigned long long *clk; hipMalloc(&clk, 16);
    const int reps = 20000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((rate<MODE, V>), dim3(grid), dim3(256), 0, 0, out, 100, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate<MODE, V>), dim3(grid), dim3(256), 0, 0, out, reps, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const char *mn[3] = {"8 x f32 16x16x4", "6 x bf16 16x16x32", "3 x f16 16x16x32"};
    printf("%-20s + %d VALU per MFMA, %d waves/SIMD : %7.2f ns per K=32 step per SIMD   shader clock %.2f GHz\n", mn[MODE], V, wgs_per_cu,
           ms * 1e6 / reps / wgs_per_cu, (double)h[0] / ((double)h[1] * 10.0));
    hipFree(out); hipFree(clk);
}

int main() {
    srand(7);
    auto rnd = []() { return (float)((rand() + 0.5) / (RAND_MAX + 1.0)); };
    auto gauss = [&]() { return sqrtf(-2.f * logf(rnd())) * cosf(6.2831853f * rnd()); };
    for (int K : {64, 128, 512}) {
        const int T = 24;
        std::vector<float> A((size_t)T * 16 * K), B((size_t)T * K * 16);
        for (auto &v : A) v = gauss();
        for (auto &v : B) v = gauss();
        accuracy("normal x normal", A, B, K, T);
    }
    {   // small weights (uniform +-0.05, a default-initialised conv row) x post-ReLU activations with a wide dynamic range
        const int K = 384, T = 24;
        std::vector<float> A((size_t)T * 16 * K), B((size_t)T * K * 16);
        for (auto &v : A) v = (rnd() - 0.5f) * 0.1f;
        for (auto &v : B) { const float g = gauss() * expf(3.f * gauss()); v = g > 0.f ? g : 0.f; }
        accuracy("weights +-0.05 x relu(lognormal)", A, B, K, T);
    }
    {   // tiny operands: everything below the fp16 normal range (|x| < 6e-5)
        const int K = 128, T = 8;
        std::vector<float> A((size_t)T * 16 * K), B((size_t)T * K * 16);
        for (auto &v : A) v = gauss() * 1e-5f;
        for (auto &v : B) v = gauss() * 3e-6f;
        accuracy("tiny x tiny (1e-5 x 3e-6)", A, B, K, T);
    }
    {   // a windowed DFT table as the reference builds it times pre-emphasised int16-range samples (scaled 1/32768)
        const int K = 416, T = 16;
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
        run_rate<2, 1>(w); run_rate<2, 2>(w); run_rate<2, 3>(w); run_rate<2, 4>(w); run_rate<2, 6>(w);
        run_rate<1, 2>(w);
    }
    return 0;
}
SRC
/opt/rocm/bin/hipcc -w --offload-arch=gfx950 -O3 -o /tmp/p /tmp/p.hip && /tmp/p
