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
