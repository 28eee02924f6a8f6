// capi.hip -- error plumbing shared by every libvadx entry point.
#include "common.h"

namespace vadx {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace vadx

extern "C" int vadx_abi_version(void) { return 1; }
extern "C" const char *vadx_last_error(void) { return vadx::g_err; }
