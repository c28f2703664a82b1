// Gradient all-reduce of the data-parallel rollout step over RCCL (xGMI), on the caller's stream.
// The reference trains single-process (src/nsbench/scripts/train.py:36,66; no DistributedDataParallel anywhere, SURVEY.md
// §2.3): data parallelism is a build addition whose one collective is a sum all-reduce of the flat fp32 gradient buffer
// per optimizer step.  The Python host normally goes through torch.distributed's "nccl" backend (= RCCL);
// these entry points give a host WITHOUT torch the same exchange: one communicator per process / GPU, created from a
// 128-byte unique id that rank 0 generates and the host distributes (any side channel), collectives enqueued on the
// caller's hipStream (capture-safe as far as RCCL is).
//
// librccl is bound lazily (dlopen at the first comm call): libdlwpmi keeps loading on machines without RCCL and single-GPU
// runs never pay its load time; a missing library is reported loudly by the first call, there is no fallback.
#include "common.hip.h"
#include "dlwpmi_internal.h"
#include <dlfcn.h>
#include <cstring>
#include <mutex>

namespace {

typedef struct { char internal[128]; } rccl_unique_id;      // ncclUniqueId (rccl.h:40-43)
typedef void* rccl_comm_t;
typedef int rccl_result_t;                                   // ncclSuccess = 0
constexpr int RCCL_FLOAT32 = 7, RCCL_SUM = 0;                // ncclFloat32 / ncclSum (rccl.h:448-466)

struct RcclApi {
    void* handle = nullptr;
    rccl_result_t (*GetUniqueId)(rccl_unique_id*) = nullptr;
    rccl_result_t (*CommInitRank)(rccl_comm_t*, int, rccl_unique_id, int) = nullptr;
    rccl_result_t (*CommDestroy)(rccl_comm_t) = nullptr;
    rccl_result_t (*AllReduce)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
    rccl_result_t (*Broadcast)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(rccl_result_t) = nullptr;
    char why[256] = "not attempted";             // the loader's message, captured once where the failure happened
};

RcclApi g_rccl;

RcclApi* rccl() {
    static std::once_flag once;
    static bool ok = false;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.handle) break;
            const char* e = dlerror();                // reading clears the loader's state: read once, keep the last one
            snprintf(g_rccl.why, sizeof(g_rccl.why), "%s", e ? e : "dlopen failed");
        }
        if (!g_rccl.handle) return;
        g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(g_rccl.handle, "ncclGetUniqueId"));
        g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(g_rccl.handle, "ncclCommInitRank"));
        g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(g_rccl.handle, "ncclCommDestroy"));
        g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(g_rccl.handle, "ncclAllReduce"));
        g_rccl.Broadcast = reinterpret_cast<decltype(g_rccl.Broadcast)>(dlsym(g_rccl.handle, "ncclBroadcast"));
        g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(g_rccl.handle, "ncclGetErrorString"));
        ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.Broadcast && g_rccl.GetErrorString;
        if (!ok) snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl was loaded but lacks one of the nccl* entry points");
    });
    return ok ? &g_rccl : nullptr;
}


#define DLWP_RCCL(api, call)                                                                   \
    do {                                                                                       \
        const rccl_result_t r__ = (call);                                                      \
        if (r__ != 0) {                                                                        \
            dlwp_set_error("%s -> RCCL error %d: %s", #call, r__, (api)->GetErrorString(r__)); \
            return DLWP_E_HIP;                                                                 \
        }                                                                                      \
    } while (0)

}  // namespace

struct dlwp_comm {
    rccl_comm_t comm;
    int rank, world;
};

extern "C" int dlwp_comm_unique_id(void* out128) {
    DLWP_REQUIRE(out128, DLWP_E_INVALID, "comm_unique_id: NULL argument");
    RcclApi* api = rccl();
    DLWP_REQUIRE(api, DLWP_E_UNSUPPORTED, "comm_unique_id: librccl.so could not be loaded (%s)", g_rccl.why);
    DLWP_RCCL(api, api->GetUniqueId(static_cast<rccl_unique_id*>(out128)));
    return DLWP_OK;
}

extern "C" int dlwp_comm_create(const void* unique_id128, int rank, int world, dlwp_comm** out) {
    DLWP_REQUIRE(unique_id128 && out && world >= 1 && rank >= 0 && rank < world, DLWP_E_INVALID, "comm_create: bad argument");
    RcclApi* api = rccl();
    DLWP_REQUIRE(api, DLWP_E_UNSUPPORTED, "comm_create: librccl.so could not be loaded (%s)", g_rccl.why);
    rccl_unique_id id;
    memcpy(&id, unique_id128, sizeof(id));
    rccl_comm_t c = nullptr;
    DLWP_RCCL(api, api->CommInitRank(&c, world, id, rank));       // blocks until every rank has joined
    *out = new dlwp_comm{c, rank, world};
    return DLWP_OK;
}

extern "C" void dlwp_comm_destroy(dlwp_comm* c) {
    if (!c) return;
    if (RcclApi* api = rccl()) (void)api->CommDestroy(c->comm);
    delete c;
}

extern "C" int dlwp_comm_allreduce(dlwp_comm* c, float* buf, long long n, void* stream) {
    DLWP_REQUIRE(c && buf && n >= 0, DLWP_E_INVALID, "comm_allreduce: bad argument");
    if (n == 0) return DLWP_OK;
    RcclApi* api = rccl();
    DLWP_REQUIRE(api, DLWP_E_UNSUPPORTED, "comm_allreduce: librccl.so could not be loaded (%s)", g_rccl.why);
    DLWP_RCCL(api, api->AllReduce(buf, buf, (size_t)n, RCCL_FLOAT32, RCCL_SUM, c->comm, static_cast<hipStream_t>(stream)));
    return DLWP_OK;
}

extern "C" int dlwp_comm_broadcast(dlwp_comm* c, float* buf, long long n, int root, void* stream) {
    DLWP_REQUIRE(c && buf && n >= 0 && root >= 0 && root < c->world, DLWP_E_INVALID, "comm_broadcast: bad argument");
    if (n == 0) return DLWP_OK;
    RcclApi* api = rccl();
    DLWP_REQUIRE(api, DLWP_E_UNSUPPORTED, "comm_broadcast: librccl.so could not be loaded (%s)", g_rccl.why);
    DLWP_RCCL(api, api->Broadcast(buf, buf, (size_t)n, RCCL_FLOAT32, root, c->comm, static_cast<hipStream_t>(stream)));
    return DLWP_OK;
}
