// Error reporting for the vlr C-ABI: every entry point returns 0 on success; on failure the message is kept in a
// thread-local buffer and returned by vlr_last_error().
#include <stdarg.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[1024] = "";

void vlr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int vlr_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        vlr_set_error("%s: kernel launch failed: %s", what, hipGetErrorString(e));
        return VLR_ERR_HIP;
    }
    return VLR_OK;
}

extern "C" const char* vlr_last_error(void) { return g_err; }
extern "C" int vlr_abi_version(void) { return 1; }
