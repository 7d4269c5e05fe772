// Collectives of the row-sharded path in the C ABI: a C-ABI consumer gets the same exchange the Python
// host does through torch.distributed -- RCCL all-reduce (sum) of the small replicated terms over xGMI:
//   the r x r Gram and the r x n cross term of the V update (one buffer), the per-sweep stopping sums of the U-side solve,
//   the cost scalar.  One process per GPU; the 128-byte RCCL unique id travels by whatever channel the host application has.
// RCCL is bound at run time (dlopen): libnnfac_hip.so itself has no link-time dependency on it, and inside a PyTorch
// process the library PyTorch already loaded is the one that answers.
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "nnf_internal.h"

typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess_ = 0 };
enum { ncclFloat32_ = 7, ncclFloat64_ = 8, ncclSum_ = 0 };   // rccl.h: ncclDataType_t / ncclRedOp_t

struct nnf_comm {
    ncclComm_t comm;
    int nranks, rank, device;
};

namespace {
struct rccl_api {
    void* handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    bool ok = false;
};
rccl_api& api() {
    static rccl_api a;
    static std::once_flag once;
    std::call_once(once, [] {
        // the copy that is ALREADY in the process answers (PyTorch's bundled library has the SONAME librccl.so.1; a plain
        // dlopen("librccl.so") would not match it and could map a second RCCL from /opt/rocm next to it): RTLD_NOLOAD over
        // the known names first, a fresh load only when nothing is resident (a host that never imported torch)
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* name : names) {
            a.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
            if (a.handle) break;
        }
        // resident under another name (a statically named copy inside the host): the global scope answers (RTLD_DEFAULT == 0)
        bool found = a.handle != nullptr || dlsym(RTLD_DEFAULT, "ncclAllReduce") != nullptr;
        for (const char* name : names) {
            if (found) break;
            a.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            found = a.handle != nullptr;
        }
        if (found) {
            a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.handle, "ncclGetUniqueId");
            a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.handle, "ncclCommInitRank");
            a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.handle, "ncclCommDestroy");
            a.AllReduce = (decltype(a.AllReduce))dlsym(a.handle, "ncclAllReduce");
            a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce;
        }
    });
    return a;
}
}   // namespace

extern "C" int nnf_comm_unique_id(void* id_out_128_bytes) {
    if (!id_out_128_bytes) return NNF_ERR_ARG;
    rccl_api& a = api();
    if (!a.ok) return NNF_ERR_DEVICE;
    ncclUniqueId id;
    if (a.GetUniqueId(&id) != ncclSuccess_) return NNF_ERR_DEVICE;
    memcpy(id_out_128_bytes, &id, sizeof(id));
    return NNF_OK;
}

extern "C" int nnf_comm_create(nnf_comm** out, nnf_ctx* ctx, int nranks, int rank, const void* id_128_bytes) {
    if (!out || !ctx || !id_128_bytes || nranks < 1 || rank < 0 || rank >= nranks) return NNF_ERR_ARG;
    *out = nullptr;
    rccl_api& a = api();
    if (!a.ok) return NNF_ERR_DEVICE;
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(ctx->device) != hipSuccess) return NNF_ERR_DEVICE;
    ncclUniqueId id;
    memcpy(&id, id_128_bytes, sizeof(id));
    ncclComm_t c = nullptr;
    const int rc = a.CommInitRank(&c, nranks, id, rank);
    (void)hipSetDevice(prev);
    if (rc != ncclSuccess_ || !c) return NNF_ERR_DEVICE;
    nnf_comm* h = new (std::nothrow) nnf_comm{c, nranks, rank, ctx->device};
    if (!h) {
        a.CommDestroy(c);
        return NNF_ERR_DEVICE;
    }
    *out = h;
    return NNF_OK;
}

extern "C" int nnf_comm_destroy(nnf_comm* comm) {
    if (!comm) return NNF_ERR_ARG;
    rccl_api& a = api();
    if (a.ok && comm->comm) a.CommDestroy(comm->comm);
    delete comm;
    return NNF_OK;
}

extern "C" int nnf_comm_size(const nnf_comm* comm) { return comm ? comm->nranks : 0; }
extern "C" int nnf_comm_rank(const nnf_comm* comm) { return comm ? comm->rank : -1; }

static int allreduce(nnf_comm* comm, void* buf, int64_t count, int dtype, void* stream) {
    if (!comm || !buf || count < 1) return NNF_ERR_ARG;
    rccl_api& a = api();
    if (!a.ok) return NNF_ERR_DEVICE;
    return a.AllReduce(buf, buf, (size_t)count, dtype, ncclSum_, comm->comm, (hipStream_t)stream) == ncclSuccess_ ? NNF_OK
                                                                                                               : NNF_ERR_LAUNCH;
}
extern "C" int nnf_allreduce_f32(nnf_comm* comm, float* buf, int64_t count, void* stream) {
    return allreduce(comm, buf, count, ncclFloat32_, stream);
}
extern "C" int nnf_allreduce_f64(nnf_comm* comm, double* buf, int64_t count, void* stream) {
    return allreduce(comm, buf, count, ncclFloat64_, stream);
}
