#!/bin/bash
# Which SIMD does wave w of a workgroup land on?  (run on the GPU box)  Reads HW_REG_HW_ID.SIMD_ID per wave.
cat > /tmp/simd_probe.hip <<'SRC'
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int *out) {
    const int wave = threadIdx.x >> 6;
    const unsigned simd = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);      // HW_REG_HW_ID bits [5:4]
    const unsigned cu = __builtin_amdgcn_s_getreg((3 << 11) | (8 << 6) | 4);        // bits [11:8]
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wave] = (int)(simd | (cu << 8));
    // keep the workgroup resident for a while so several share a CU
    for (volatile int i = 0; i < 20000; ++i) {}
}
int main() {
    for (int threads : {128, 256, 384, 512}) {
        int *d; hipMalloc(&d, 4096 * 16 * 4); hipMemset(d, 0xff, 4096 * 16 * 4);
        hipLaunchKernelGGL(k, dim3(1024), dim3(threads), 0, 0, d);
        hipDeviceSynchronize();
        static int h[4096 * 16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("threads %d:", threads);
        for (int b : {0, 1, 2, 300, 301, 777}) { printf("  wg%d[", b); for (int w = 0; w < threads / 64; ++w) printf("%d", h[b * 16 + w] & 3); printf("]cu%d", (h[b * 16] >> 8) & 15); }
        printf("\n");
        hipFree(d);
    }
}
SRC
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 /tmp/simd_probe.hip -o /tmp/simd_probe && /tmp/simd_probe
