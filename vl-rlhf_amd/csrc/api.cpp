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
extern "C" int vlr_abi_version(void) { return 9; }

// ---- in-library kernel timing (bench.py roofline leg): HIP events on the launch stream around selected kernels -------
#include <vector>
struct ProfRec { hipEvent_t a, b; int kernel; double work; };
static std::vector<ProfRec> g_prof;
static int g_prof_on = 0;                 // 0 off; N >= 1: bracket one launch in N (pseudo-randomly, per kernel id)
#define VLR_PROF_KERNELS 16
static unsigned long g_prof_seen[VLR_PROF_KERNELS];
static double g_prof_work[VLR_PROF_KERNELS];

// on = 0 stops; on = N >= 1 starts and brackets one launch in N.  An event pair costs two queue packets around the kernel
// (measured: see DESIGN.md section 6), so the bench samples; launch counts and work are exact, the time is the sampled mean
// scaled to the launch count.
extern "C" int vlr_prof_enable(int on) {
    g_prof_on = on < 0 ? 0 : on;
    if (on) {
        for (auto& r : g_prof) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
        g_prof.clear();
        for (int i = 0; i < VLR_PROF_KERNELS; ++i) { g_prof_seen[i] = 0; g_prof_work[i] = 0.0; }
    }
    return VLR_OK;
}
int vlr_prof_begin(int kernel, double work, hipStream_t st) {
    if (!g_prof_on || kernel < 0 || kernel >= VLR_PROF_KERNELS) return -1;
    const unsigned long n = g_prof_seen[kernel]++;
    g_prof_work[kernel] += work;
    // the launches of one kernel id repeat with the layer structure: pick by a hash of the running count, not every N-th
    if (g_prof_on > 1 && (unsigned)((n * 2654435761ul) >> 16) % (unsigned)g_prof_on != 0) return -1;
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
    double ms_sum[VLR_PROF_KERNELS] = {0}, n_sampled[VLR_PROF_KERNELS] = {0};
    for (int i = 0; i < nk * 3; ++i) out[i] = 0.0;
    for (auto& r : g_prof) {
        hipEventSynchronize(r.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;     // a bracket whose launch was declined (end never recorded)
        ms_sum[r.kernel] += ms; n_sampled[r.kernel] += 1.0;
    }
    for (int k = 0; k < nk && k < VLR_PROF_KERNELS; ++k) {
        out[k * 3] = (double)g_prof_seen[k];
        out[k * 3 + 1] = n_sampled[k] > 0 ? ms_sum[k] * (double)g_prof_seen[k] / n_sampled[k] : 0.0;
        out[k * 3 + 2] = g_prof_work[k];
    }
    return VLR_OK;
}

// ---- CUs left to the compute kernels.  The persistent GEMM / attention launches size their grids to this many CUs (whole XCD
// octets): under data parallelism the RCCL ring kernels of the gradient exchange run beside the backward, and a persistent launch
// that fills every CU is displaced by them workgroup by workgroup (every displaced workgroup is a tail on a 256-workgroup launch -
// DESIGN.md section 5).  vlr_set_comm_cus(k) / VLR_COMM_CUS=k leaves k CUs (rounded up to a multiple of 8: one per XCD) free.
#include <stdlib.h>
static int g_comm_cus = -1;
extern "C" int vlr_set_comm_cus(int k) {
    VLR_REQUIRE(k >= -1 && k <= 128, "vlr_set_comm_cus: 0 <= k <= 128 (or -1: back to VLR_COMM_CUS / 0), got %d", k);
    g_comm_cus = k;
    return VLR_OK;
}
extern "C" int vlr_compute_cus(void) {
    static int dev_cus = 0;
    if (!dev_cus) {
        int dev = 0, cus = 0;
        dev_cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus >= 8) ? cus : 256;
    }
    if (g_comm_cus < 0) { const char* e = getenv("VLR_COMM_CUS"); g_comm_cus = e ? atoi(e) : 0; if (g_comm_cus < 0 || g_comm_cus > 128) g_comm_cus = 0; }
    int n = (dev_cus & ~7) - ((g_comm_cus + 7) & ~7);
    return n < 8 ? 8 : n;
}
