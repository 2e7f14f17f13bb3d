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
extern "C" int vlr_abi_version(void) { return 3; }

// ---- in-library kernel timing (bench.py roofline leg): HIP events on the launch stream around selected kernels -------
#include <vector>
struct ProfRec { hipEvent_t a, b; int kernel; double work; };
static std::vector<ProfRec> g_prof;
static int g_prof_on = 0;

extern "C" int vlr_prof_enable(int on) {
    g_prof_on = on;
    if (on) {
        for (auto& r : g_prof) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
        g_prof.clear();
    }
    return VLR_OK;
}
int vlr_prof_begin(int kernel, double work, hipStream_t st) {
    if (!g_prof_on) return -1;
    ProfRec r;
    r.kernel = kernel; r.work = work;
    hipEventCreate(&r.a); hipEventCreate(&r.b);
    hipEventRecord(r.a, st);
    g_prof.push_back(r);
    return (int)g_prof.size() - 1;
}
void vlr_prof_end(int idx, hipStream_t st) {
    if (idx >= 0) hipEventRecord(g_prof[idx].b, st);
}
// out[kernel*3 + {0,1,2}] = {launches, total ms, total work}; kernels 0..nk-1.  Synchronises on the recorded events.
extern "C" int vlr_prof_collect(double* out, int nk) {
    for (int i = 0; i < nk * 3; ++i) out[i] = 0.0;
    for (auto& r : g_prof) {
        hipEventSynchronize(r.b);
        float ms = 0.f;
        hipEventElapsedTime(&ms, r.a, r.b);
        if (r.kernel < nk) { out[r.kernel * 3] += 1.0; out[r.kernel * 3 + 1] += ms; out[r.kernel * 3 + 2] += r.work; }
    }
    return VLR_OK;
}
