// capi.hip -- error plumbing and the weight-layout helper shared by every libvadx entry point.
#include "common.h"

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
