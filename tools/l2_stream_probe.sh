#!/bin/bash
# How fast can every CU stream the SAME weight blob out of L2 into registers? (run on the GPU box)
# The tile kernels stream their weights as 1 KiB wave-wide fragment loads (16 B per lane) from a blob every workgroup re-reads; with
# bf16 x 3 split products the matrix time per weight byte falls 3.8 x, so this rate becomes the wall.  Grid = many tiles, each wave walks
# its 1/8 of the blob, D loads in flight per lane, the loaded values folded into one register (v_or) so nothing is dropped.
cat > /tmp/l2.hip <<'SRC'
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int D>
__global__ __launch_bounds__(512) void stream(const u32x4 *__restrict__ blob, int frags_per_wave, unsigned *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4 *p = blob + (size_t)wave * frags_per_wave * 64 + lane;
    u32x4 acc = {0, 0, 0, 0};
    for (int f = 0; f < frags_per_wave; f += D) {
        u32x4 v[D];
#pragma unroll
        for (int d = 0; d < D; ++d) v[d] = p[(size_t)(f + d) * 64];
#pragma unroll
        for (int d = 0; d < D; ++d) acc |= v[d];
    }
    if (acc[0] == 0x12345678u) out[blockIdx.x] = acc[1] + acc[2] + acc[3];
}
template <int D>
void run(size_t blob_bytes, int tiles, size_t lds) {
    u32x4 *blob; unsigned *out;
    hipMalloc(&blob, blob_bytes); hipMemset(blob, 0, blob_bytes); hipMalloc(&out, tiles * 4);
    const int fpw = (int)(blob_bytes / 1024 / 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((stream<D>), dim3(tiles), dim3(512), lds, 0, blob, fpw, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream<D>), dim3(tiles), dim3(512), lds, 0, blob, fpw, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("blob %5zu KB, %6d tiles, %d loads in flight, LDS %3zu KB/WG: %7.3f ms  %6.1f TB/s L2 -> registers\n", blob_bytes >> 10, tiles, D, lds >> 10, ms,
           (double)blob_bytes * tiles / (ms * 1e-3) / 1e12);
    hipFree(blob); hipFree(out);
}
int main() {
    for (size_t kb : {256, 1024}) {
        for (size_t lds : {(size_t)20 << 10, (size_t)52 << 10, (size_t)100 << 10}) {   // 8 (capped by waves: 4), 3, 1 workgroups per CU
            run<4>(kb << 10, 20000, lds);
            run<8>(kb << 10, 20000, lds);
            run<16>(kb << 10, 20000, lds);
        }
    }
    return 0;
}
SRC
/opt/rocm/bin/hipcc -w --offload-arch=gfx950 -O3 -o /tmp/l2 /tmp/l2.hip && /tmp/l2
