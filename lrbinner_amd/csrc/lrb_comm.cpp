// lrb_comm.cpp -- the path's ONE collective behind the C ABI: in-place sum of (the canonical half of)
// the 15-mer table over the GPUs of a node, RCCL over xGMI, on the context's stream.
//
// RCCL is bound at run time (dlopen): a process that already carries an RCCL -- torch ships its own --
// keeps using THAT copy, so a communicator made by either side is valid for both, and a single-GPU
// user of liblrb_hip.so needs no RCCL at all.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "lrb_device.h"

namespace {

struct rccl_uid {
    char internal[128]; // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128)
};
static_assert(sizeof(rccl_uid) == LRB_RCCL_ID_BYTES, "id size");

// rccl.h: ncclResult_t = int (0 = ncclSuccess); ncclDataType_t ncclUint32 = 3; ncclRedOp_t ncclSum = 0
typedef int (*fn_get_unique_id)(rccl_uid *);
typedef int (*fn_comm_init_rank)(void **, int, rccl_uid, int);
typedef int (*fn_comm_destroy)(void *);
typedef int (*fn_all_reduce)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef const char *(*fn_error_string)(int);

struct rccl_api {
    void *handle;
    fn_get_unique_id get_unique_id;
    fn_comm_init_rank comm_init_rank;
    fn_comm_destroy comm_destroy;
    fn_all_reduce all_reduce;
    fn_error_string error_string;
};

rccl_api g_api = {};

int load_rccl()
{
    if (g_api.handle) return LRB_OK;
    static const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *nm : names) // a copy that is already in the process first
        if ((h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD)) != nullptr) break;
    if (!h)
        for (const char *nm : names)
            if ((h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
    if (!h) {
        lrb_set_error("RCCL is not available: %s%s", dlerror(), "");
        return LRB_ERR_NODEVICE;
    }
    g_api.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
    g_api.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
    g_api.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
    g_api.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
    g_api.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
    if (!g_api.get_unique_id || !g_api.comm_init_rank || !g_api.comm_destroy || !g_api.all_reduce) {
        lrb_set_error("RCCL symbols missing%s%s", "", "");
        return LRB_ERR_NODEVICE;
    }
    g_api.handle = h;
    return LRB_OK;
}

int rccl_fail(const char *what, int rc)
{
    lrb_set_error("%s failed: %s", what, g_api.error_string ? g_api.error_string(rc) : "RCCL error");
    return LRB_ERR_HIP;
}

} // namespace

extern "C" int lrb_rccl_unique_id(uint8_t *id)
{
    ARG_TRY(id != nullptr);
    int rc = load_rccl();
    if (rc != LRB_OK) return rc;
    rccl_uid u;
    const int r = g_api.get_unique_id(&u);
    if (r != 0) return rccl_fail("ncclGetUniqueId", r);
    memcpy(id, u.internal, sizeof(u.internal));
    return LRB_OK;
}

extern "C" int lrb_rccl_comm_create(lrb_ctx *c, int n_ranks, int rank, const uint8_t *id, void **rccl_comm)
{
    ARG_TRY(c != nullptr && id != nullptr && rccl_comm != nullptr);
    ARG_TRY(n_ranks >= 1 && rank >= 0 && rank < n_ranks);
    int rc = load_rccl();
    if (rc != LRB_OK) return rc;
    HIP_TRY(hipSetDevice(c->device));
    rccl_uid u;
    memcpy(u.internal, id, sizeof(u.internal));
    void *comm = nullptr;
    const int r = g_api.comm_init_rank(&comm, n_ranks, u, rank);
    if (r != 0) return rccl_fail("ncclCommInitRank", r);
    *rccl_comm = comm;
    return LRB_OK;
}

extern "C" int lrb_rccl_comm_destroy(void *rccl_comm)
{
    if (!rccl_comm) return LRB_OK;
    int rc = load_rccl();
    if (rc != LRB_OK) return rc;
    const int r = g_api.comm_destroy(rccl_comm);
    if (r != 0) return rccl_fail("ncclCommDestroy", r);
    return LRB_OK;
}

extern "C" int lrb_k15_allreduce(lrb_ctx *c, void *rccl_comm, uint32_t *d_buf, uint64_t count)
{
    ARG_TRY(c != nullptr && rccl_comm != nullptr && (d_buf != nullptr || count == 0));
    if (count == 0) return LRB_OK;
    int rc = load_rccl();
    if (rc != LRB_OK) return rc;
    HIP_TRY(hipSetDevice(c->device));
    const int r = g_api.all_reduce(d_buf, d_buf, (size_t)count, /*ncclUint32*/ 3, /*ncclSum*/ 0, rccl_comm, c->stream);
    if (r != 0) return rccl_fail("ncclAllReduce", r);
    return LRB_OK;
}
