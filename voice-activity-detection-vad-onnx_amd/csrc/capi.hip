// capi.hip -- error plumbing and the weight-layout helper shared by every libvadx entry point.
#include "common.h"
#include "split2.h"

#include <stdlib.h>
#include <string.h>

namespace vadx {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace vadx

extern "C" int vadx_abi_version(void) { return VADX_ABI_VERSION; }      // include/vadx.h is the one place the number lives
extern "C" const char *vadx_last_error(void) { return vadx::g_err; }

extern "C" size_t vadx_frag_major_floats(int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    return (size_t)((rows + 15) & ~15) * (size_t)((cols + 15) & ~15);
}

extern "C" int vadx_frag_major_host(const float *src, int rows, int cols, float *dst) {
    VADX_REQUIRE(src && dst && rows > 0 && cols > 0, "vadx_frag_major_host: bad argument");
    const int ldw = (cols + 15) & ~15;
    const size_t n = vadx_frag_major_floats(rows, cols);
    for (size_t e = 0; e < n; ++e) dst[e] = 0.f;
    for (int r = 0; r < rows; ++r)
        for (int k = 0; k < cols; ++k) dst[vadx::frag_index(ldw, r, k)] = src[(size_t)r * cols + k];
    return VADX_OK;
}

// fp16 x 2 fragments of a [rows][cols] weight matrix for the entry points that take bare weight pointers (include/vadx.h):
// [rows/16 tiles][cols/32 chunks][2 planes][64 lanes][8 fp16]; lane 16 q + i, slot e holds W[16 tile + i][32 chunk + k(q, e)] with
//   VADX_H2_K_PLAIN    k = 8 q + e                       (operand planes [k / 8][column][8]: layers_split.h)
//   VADX_H2_K_QUARTER  k = 16 (e >> 2) + 4 q + (e & 3)   (the order in which marblenet.hip's kgemm_h reads k-major float32 rows without
//                                                         bank conflicts)
extern "C" size_t vadx_frag_h2_floats(int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    return (size_t)((rows + 15) / 16) * (size_t)((cols + 31) / 32) * 2 * vadx::HFRAG;
}

extern "C" int vadx_frag_h2_host(const float *src, int rows, int cols, int k_order, float *dst, float *wmax_out) {
    VADX_REQUIRE(src && dst && rows > 0 && cols > 0, "vadx_frag_h2_host: bad argument");
    VADX_REQUIRE(k_order == VADX_H2_K_PLAIN || k_order == VADX_H2_K_QUARTER, "vadx_frag_h2_host: k_order must be VADX_H2_K_PLAIN or VADX_H2_K_QUARTER");
    const size_t n = vadx_frag_h2_floats(rows, cols);
    const int nch = (cols + 31) / 32;
    memset(dst, 0, n * sizeof(float));
    float wmax = 0.f;
    for (int r = 0; r < rows; ++r)
        for (int k = 0; k < cols; ++k) {
            const int kk = k & 31, q = (kk >> 2) & 3, e = ((kk >> 4) << 2) | (kk & 3);
            const float a = vadx::hfrag_put(dst + (size_t)(((r / 16) * nch + k / 32) * 2) * vadx::HFRAG, r % 16,
                                            k_order == VADX_H2_K_QUARTER ? 8 * q + e : kk, src[(size_t)r * cols + k]);
            if (!(a <= wmax)) wmax = a;
        }
    if (wmax_out) *wmax_out = wmax;
    VADX_REQUIRE(wmax <= vadx::H_MAX, "vadx_frag_h2_host: a weight (|w| = %g) is outside the fp16 range: keep this matrix on VADX_ARITH_F32", (double)wmax);
    return VADX_OK;
}
