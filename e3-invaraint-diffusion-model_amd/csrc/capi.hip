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

// device-side dropout epoch, one registration per device ordinal (see E3dDrop in e3d_common.h)
static const uint64_t* g_drop_epoch[64] = {nullptr};
const uint64_t* e3d_dropout_epoch_ptr() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    return g_drop_epoch[dev];
}
extern "C" int e3d_dropout_set_epoch_ptr(const uint64_t* device_word) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    E3D_REQUIRE(dev >= 0 && dev < 64, "dropout_set_epoch_ptr: device ordinal %d", dev);
    g_drop_epoch[dev] = device_word;
    return 0;
}
