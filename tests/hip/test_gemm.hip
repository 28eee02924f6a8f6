// Test-only build (tests/hip/libvadx_testhooks.so, built by vadx.build.build_test_hooks): exposes the LDS/MFMA tile helper
// of csrc/common.h in isolation so tests/test_gpu_silero.py::test_mfma_tile_helper can check it against a float64 matmul.
// NOT part of the product ABI (include/vadx.h) and not linked into libvadx.so.
#include "../../voice-activity-detection-vad-onnx_amd/csrc/common.h"

namespace vadx {
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    fputc('\n', stderr);
    va_end(ap);
}
}  // namespace vadx
using namespace vadx;

// ---- test hook: C = A * W^T through gemm_pass (W fragment-major, see vadx_frag_major_host) ---------------------------------------------------
__global__ void test_gemm_kernel(const float *A, const float *W, float *C, int M, int N, int K, int swap) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lda = M + 4;                        // M in {16,32,48,64}: (M+4) % 8 == 4
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    for (int e = tid; e < M * K; e += blockDim.x) {
        const int m = e / K, k = e % K;
        lds[k * lda + m] = A[e];
    }
    __syncthreads();
    const int q = lane >> 4, i = lane & 15;
    for (int nt = wave; nt < N / 16; nt += nw) {
        for (int mt = 0; mt < M / 16; ++mt) {
            f32x4 acc[1][1] = {{f32x4{0.f, 0.f, 0.f, 0.f}}};
            const float *const wrow[1] = {frag_ptr(W, K, nt, 0, lane)};
            const int moff[1] = {mt * 16};
            for (int kb = 0; kb < K / 16; ++kb) {
                const float *const wr[1] = {wrow[0] + kb * FRAG};
                if (swap) gemm_pass<1, 1, 1, true>(acc, lds + kb * 16 * lda, lda, moff, wr, lane);
                else gemm_pass<1, 1, 1, false>(acc, lds + kb * 16 * lda, lda, moff, wr, lane);
            }
            for (int r = 0; r < 4; ++r) {
                if (swap) C[(size_t)(mt * 16 + i) * N + nt * 16 + 4 * q + r] = acc[0][0][r];
                else C[(size_t)(mt * 16 + 4 * q + r) * N + nt * 16 + i] = acc[0][0][r];
            }
        }
    }
}

/* C[M][N] = A[M][K] * W[N][K]^T through the same LDS/MFMA tile helper the nets use.
 * M multiple of 16 (<=64), N multiple of 16, K multiple of 16; w fragment-major (vadx_frag_major_host). */
extern "C" int vadx_test_gemm(const float *a, const float *w, float *c, int m, int n, int k, int swap, void *stream) {
    VADX_REQUIRE(a && w && c, "vadx_test_gemm: NULL pointer");
    VADX_REQUIRE(m > 0 && m <= 64 && m % 16 == 0 && n > 0 && n % 16 == 0 && k > 0 && k % 16 == 0 && (size_t)(m + 4) * k * 4 <= 160 * 1024,
                 "vadx_test_gemm: unsupported shape %dx%dx%d", m, n, k);
    const size_t lds = (size_t)(m + 4) * k * sizeof(float);
    VADX_DYN_LDS(test_gemm_kernel, 160 * 1024);
    hipLaunchKernelGGL(test_gemm_kernel, dim3(1), dim3(256), lds, static_cast<hipStream_t>(stream), a, w, c, m, n, k, swap);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}
