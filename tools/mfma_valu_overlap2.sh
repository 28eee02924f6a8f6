#!/bin/bash
# Follow-up to mfma_valu_overlap.sh: which instruction classes add to v_mfma_f32_16x16x4_f32's 32 cycles (one wave per SIMD)?
cat > /tmp/ov2.hip <<'SRC'
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// KIND: 0 v_fma_f32, 1 v_mov_b32, 2 v_add_u32, 3 v_pk_fma_f32, 4 ds_read_b32, 5 s_add_u32, 6 v_lshl_add_u32, 7 ds_read_b128, 8 acc in AGPRs + v_fma
template <int KIND, int K, int M>
__global__ __launch_bounds__(256) void k(float *out, int reps) {
    __shared__ float lds[4096];
    f32x4 acc[8];
    float x[8];
    f32x2 p2[8];
    f32x4 l4[8];
    unsigned u[8];
    unsigned sacc = reps;
    const float a = threadIdx.x * 1e-3f, b = 1.0f + a;
    lds[threadIdx.x] = a; lds[threadIdx.x + 256] = b;
    __syncthreads();
    const unsigned laddr = (threadIdx.x & 63) * 4;
    for (int m = 0; m < 8; ++m) { acc[m] = f32x4{a, b, a, b}; x[m] = a + m; u[m] = threadIdx.x + m; p2[m] = f32x2{a, b}; l4[m] = f32x4{a, a, a, a}; }
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (M) {
                if (KIND == 8) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
            }
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int s = (m + j) & 7;
                if (KIND == 0 || KIND == 8) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[s]) : "v"(a), "v"(b));
                if (KIND == 1) asm volatile("v_mov_b32 %0, %1" : "=v"(x[s]) : "v"(a));
                if (KIND == 2) asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[s]) : "v"(laddr));
                if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p2[s]) : "v"(p2[(s + 4) & 7]));
                if (KIND == 4) asm volatile("ds_read_b32 %0, %1" : "=v"(x[s]) : "v"(laddr));
                if (KIND == 5) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sacc));
                if (KIND == 6) asm volatile("v_lshl_add_u32 %0, %1, 2, %0" : "+v"(u[s]) : "v"(laddr));
                if (KIND == 7) asm volatile("ds_read_b128 %0, %1" : "=v"(l4[s]) : "v"(laddr));
            }
        }
        if (KIND == 4 || KIND == 7) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    float s = (float)sacc;
    for (int m = 0; m < 8; ++m) s += acc[m][0] + acc[m][3] + x[m] + u[m] + p2[m][0] + l4[m][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND, int K, int M>
void run() {
    float *out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    const int reps = 20000, grid = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, K, M>), dim3(grid), dim3(256), 0, 0, out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, K, M>), dim3(grid), dim3(256), 0, 0, out, reps);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    static const char *names[] = {"v_fma_f32", "v_mov_b32", "v_add_u32", "v_pk_fma_f32", "ds_read_b32", "s_add_u32", "v_lshl_add_u32", "ds_read_b128", "v_fma_f32 (acc in AGPR)"};
    printf("mfma %d + %d x %-24s: %.2f ns per slot\n", M, K, names[KIND], ms * 1e6 / (reps * 8.0));
    (void)hipFree(out);
}
int main() {
    run<0, 0, 1>();
    run<0, 2, 1>(); run<0, 4, 1>(); run<8, 2, 1>(); run<8, 4, 1>();
    run<1, 2, 1>(); run<1, 4, 1>(); run<2, 2, 1>(); run<2, 4, 1>(); run<3, 2, 1>(); run<3, 4, 1>();
    run<4, 1, 1>(); run<4, 2, 1>(); run<4, 4, 1>(); run<5, 2, 1>(); run<5, 4, 1>(); run<6, 2, 1>(); run<6, 4, 1>(); run<7, 1, 1>(); run<7, 2, 1>();
    run<1, 4, 0>(); run<2, 4, 0>(); run<3, 4, 0>(); run<4, 4, 0>(); run<5, 4, 0>(); run<7, 2, 0>();
    return 0;
}
SRC
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ov2 /tmp/ov2.hip 2>&1 | grep -E "error" ; /tmp/ov2
