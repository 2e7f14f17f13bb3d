// Data-parallel gradient exchange of the DPO step on RCCL over xGMI - the native replacement of what the reference
// reaches through accelerate's MULTI_GPU mode / torch DistributedDataParallel (/root/reference accelerate_config/ddp.yaml:1-14).
//
// RCCL is resolved at run time with dlopen (the library that is already mapped into the process - PyTorch ships one -
// else librccl.so.1 from the ROCm install), so libvlr_hip.so itself has no link-time dependency on it and single-GPU
// boxes never touch it.  One communicator per process (one process per GPU); the 128-byte unique id is created on
// rank 0 by vlr_comm_unique_id and carried to the other ranks by the host launcher (torch.distributed's store / a
// broadcast), exactly like ncclGetUniqueId / ncclCommInitRank are meant to be used.
#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>
#include <string.h>

#include "../../include/vlr.h"
#include "common.h"

namespace {
struct Api {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitRankConfig)(ncclComm_t*, int, ncclUniqueId, int, void*) = nullptr;      // optional (NCCL >= 2.17)
    ncclResult_t (*GetVersion)(int*) = nullptr;                                                    // optional
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char path[512] = "";
};
Api g_api;

int find_loaded(struct dl_phdr_info* info, size_t, void* out) {
    if (info->dlpi_name && strstr(info->dlpi_name, "librccl")) {
        strncpy((char*)out, info->dlpi_name, 511);
        return 1;
    }
    return 0;
}

int load_api() {
    if (g_api.handle) return VLR_OK;
    char loaded[512] = "";
    const char* env = getenv("VLR_RCCL_LIB");
    const char* cand[4] = {env, nullptr, "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    dl_iterate_phdr(find_loaded, loaded);
    cand[1] = loaded[0] ? loaded : nullptr;
    void* h = nullptr;
    for (int i = 0; i < 4 && !h; ++i) {
        if (!cand[i]) continue;
        h = dlopen(cand[i], RTLD_NOW | RTLD_LOCAL);
        if (h) strncpy(g_api.path, cand[i], sizeof(g_api.path) - 1);
    }
    if (!h) {
        vlr_set_error("vlr_comm: cannot load RCCL (tried VLR_RCCL_LIB, the loaded librccl, librccl.so.1, /opt/rocm/lib/librccl.so): %s", dlerror());
        return VLR_ERR_HIP;
    }
#define SYM(field, name)                                                                  \
    g_api.field = (decltype(g_api.field))dlsym(h, name);                                  \
    if (!g_api.field) {                                                                   \
        vlr_set_error("vlr_comm: %s has no symbol %s", g_api.path, name);                 \
        dlclose(h);                                                                       \
        return VLR_ERR_HIP;                                                               \
    }
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_api.CommInitRankConfig = (decltype(g_api.CommInitRankConfig))dlsym(h, "ncclCommInitRankConfig");
    g_api.GetVersion = (decltype(g_api.GetVersion))dlsym(h, "ncclGetVersion");
    g_api.handle = h;
    return VLR_OK;
}

int nccl_fail(const char* what, ncclResult_t r) {
    vlr_set_error("%s: RCCL error %d (%s)", what, (int)r, g_api.GetErrorString ? g_api.GetErrorString(r) : "?");
    return VLR_ERR_HIP;
}
}  // namespace

extern "C" int vlr_comm_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

extern "C" const char* vlr_comm_library(void) { return load_api() == VLR_OK ? g_api.path : ""; }

extern "C" int vlr_comm_unique_id(void* id_host) {
    VLR_REQUIRE(id_host, "vlr_comm_unique_id: null argument");
    int rc = load_api();
    if (rc != VLR_OK) return rc;
    ncclUniqueId id;
    ncclResult_t r = g_api.GetUniqueId(&id);
    if (r != ncclSuccess) return nccl_fail("vlr_comm_unique_id", r);
    memcpy(id_host, &id, sizeof(id));
    return VLR_OK;
}

extern "C" int vlr_comm_init(const void* id_host, int rank, int world, void** comm_out) {
    VLR_REQUIRE(id_host && comm_out, "vlr_comm_init: null argument");
    VLR_REQUIRE(world >= 1 && rank >= 0 && rank < world, "vlr_comm_init: rank %d outside world %d", rank, world);
    int rc = load_api();
    if (rc != VLR_OK) return rc;
    ncclUniqueId id;
    memcpy(&id, id_host, sizeof(id));
    ncclComm_t c = nullptr;
    ncclResult_t r = g_api.CommInitRank(&c, world, id, rank);
    if (r != ncclSuccess) return nccl_fail("vlr_comm_init", r);
    *comm_out = (void*)c;
    return VLR_OK;
}

// The same with a PER-COMMUNICATOR bound on the channels (= workgroups of the ring kernel; ncclConfig_t::minCTAs / maxCTAs, NCCL >= 2.17):
// the DPO step leaves `comm_cus` CUs to the ring kernels (vlr_set_comm_cus) and the communicator is told to use exactly that many - without
// the process-wide NCCL_MAX_NCHANNELS / NCCL_MIN_NCHANNELS environment, so that two communicators of one process (bench.py's bucket probe:
// bounded and unbounded side by side) can differ.  max_ctas <= 0: no bound (= vlr_comm_init).  The configuration is passed in the layout
// of NCCL 2.18 (size / magic / version header + blocking, cgaClusterSize, minCTAs, maxCTAs, netName, splitShare), which every later
// library accepts by its version field - the RCCL that is loaded at run time (PyTorch's) need not be the one whose header was compiled
// against.  VLR_ERR_HIP with "no ncclCommInitRankConfig" when the loaded library has no such entry: the caller falls back to the environment.
extern "C" int vlr_comm_init_cfg(const void* id_host, int rank, int world, int min_ctas, int max_ctas, void** comm_out) {
    if (max_ctas <= 0) return vlr_comm_init(id_host, rank, world, comm_out);
    VLR_REQUIRE(id_host && comm_out, "vlr_comm_init_cfg: null argument");
    VLR_REQUIRE(world >= 1 && rank >= 0 && rank < world, "vlr_comm_init_cfg: rank %d outside world %d", rank, world);
    VLR_REQUIRE(min_ctas >= 0 && min_ctas <= max_ctas, "vlr_comm_init_cfg: min_ctas %d max_ctas %d", min_ctas, max_ctas);
    int rc = load_api();
    if (rc != VLR_OK) return rc;
    if (!g_api.CommInitRankConfig) {
        vlr_set_error("vlr_comm_init_cfg: %s has no ncclCommInitRankConfig", g_api.path);
        return VLR_ERR_HIP;
    }
    struct Cfg218 { size_t size; unsigned int magic; unsigned int version; int blocking; int cgaClusterSize; int minCTAs; int maxCTAs; const char* netName; int splitShare; };
    const int UNDEF = (int)0x80000000;       // NCCL_CONFIG_UNDEF_INT (INT_MIN)
    Cfg218 cfg = {sizeof(Cfg218), 0xcafebeefu, 21800u, UNDEF, UNDEF, min_ctas > 0 ? min_ctas : UNDEF, max_ctas, nullptr, UNDEF};
    ncclUniqueId id;
    memcpy(&id, id_host, sizeof(id));
    ncclComm_t c = nullptr;
    ncclResult_t r = g_api.CommInitRankConfig(&c, world, id, rank, &cfg);
    if (r != ncclSuccess) return nccl_fail("vlr_comm_init_cfg", r);
    *comm_out = (void*)c;
    return VLR_OK;
}
// version code of the RCCL library in use (ncclGetVersion: major * 10000 + minor * 100 + patch), 0 when it cannot be asked
// 1 when the loaded library exports ncclCommInitRankConfig (the ranks agree on this BEFORE any of them calls vlr_comm_init_cfg: a rank
// that fails there while the others are already inside the bootstrap would leave them waiting for it)
extern "C" int vlr_comm_has_config(void) { return load_api() == VLR_OK && g_api.CommInitRankConfig ? 1 : 0; }
extern "C" int vlr_comm_rccl_version(void) {
    int v = 0;
    if (load_api() != VLR_OK || !g_api.GetVersion || g_api.GetVersion(&v) != ncclSuccess) return 0;
    return v;
}

extern "C" int vlr_comm_destroy(void* comm) {
    if (!comm || !g_api.handle) return VLR_OK;
    ncclResult_t r = g_api.CommDestroy((ncclComm_t)comm);
    return r == ncclSuccess ? VLR_OK : nccl_fail("vlr_comm_destroy", r);
}

// in-place SUM all-reduce of one contiguous bucket of the flat gradient buffer (dtype 0 = bf16, 1 = fp32)
extern "C" int vlr_allreduce_bucket(void* comm, void* buf, long n, int dtype, vlr_stream_t stream) {
    VLR_REQUIRE(comm && g_api.handle, "vlr_allreduce_bucket: no communicator (vlr_comm_init first)");
    VLR_REQUIRE(buf && n >= 0, "vlr_allreduce_bucket: bad buffer");
    VLR_REQUIRE(dtype == 0 || dtype == 1, "vlr_allreduce_bucket: dtype must be 0 (bf16) or 1 (fp32), got %d", dtype);
    if (n == 0) return VLR_OK;
    ncclResult_t r = g_api.AllReduce(buf, buf, (size_t)n, dtype == 0 ? ncclBfloat16 : ncclFloat32, ncclSum, (ncclComm_t)comm,
                                     (hipStream_t)stream);
    return r == ncclSuccess ? VLR_OK : nccl_fail("vlr_allreduce_bucket", r);
}
