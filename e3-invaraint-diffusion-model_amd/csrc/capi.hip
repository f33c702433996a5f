// ABI version + thread-local error string shared by every entry point.
#include <stdarg.h>
#include <string.h>

#include "e3d_common.h"

static thread_local char g_err[512] = "";

void e3d_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int e3d_abi_version(void) { return E3D_ABI_VERSION; }
extern "C" const char* e3d_last_error(void) { return g_err; }
