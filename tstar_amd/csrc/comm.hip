// The ONE collective of the path: an all-gather of the final keyframe indices of every
// rank's (video, question) items over RCCL / xGMI.  The reference has no multi-GPU code; this replaces the
// sequential dataset loop's result list (LVHaystackBench/run_TStar_onDataset.py:195-205) for a host in ANY language:
// plain C ABI, device buffers, the caller's stream.
//
// RCCL is bound at run time (dlopen) rather than at link time: a host that already carries an RCCL (PyTorch-ROCm
// bundles one) must end up with ONE copy in the process -- dlopen("librccl.so") returns the already-loaded library --
// and a single-GPU user of this library never loads RCCL at all.
#include "../../include/tstar_hip.h"
#include "common.h"
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>
#include <string.h>

namespace tstar {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static Rccl g_rccl;
static std::mutex g_rccl_mu;

static int load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return TSTAR_OK;
    const char* env = getenv("TSTAR_RCCL_LIB");
    const char* names[] = {env, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    std::string tried;
    for (const char* n : names) {
        if (!n || !*n) continue;
        lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (lib) break;
        tried += std::string(n) + ": " + dlerror() + "; ";
    }
    if (!lib) { set_error("RCCL is not loadable (" + tried + ")"); return TSTAR_ERR_STATE; }
    Rccl r;
    r.lib = lib;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(lib, "ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString) {
        set_error("the RCCL library lacks an expected nccl* symbol");
        dlclose(lib);
        return TSTAR_ERR_STATE;
    }
    g_rccl = r;
    return TSTAR_OK;
}

#define TSTAR_NCCL_CHECK(expr)                                                                     \
    do {                                                                                           \
        ncclResult_t _r = (expr);                                                                  \
        if (_r != ncclSuccess) {                                                                   \
            ::tstar::set_error(std::string(#expr) + ": " + g_rccl.GetErrorString(_r));             \
            return TSTAR_ERR_HIP;                                                                  \
        }                                                                                          \
    } while (0)

}  // namespace tstar

using namespace tstar;

struct tstar_comm {
    ncclComm_t comm = nullptr;
    int world = 0, rank = 0;
};

extern "C" {

int tstar_comm_available(void) { return load_rccl(); }

int tstar_comm_unique_id(void* h_id) {
    TSTAR_REQUIRE(h_id, "tstar_comm_unique_id: null argument");
    static_assert(sizeof(ncclUniqueId) == TSTAR_COMM_ID_BYTES, "TSTAR_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
    if (int rc = load_rccl()) return rc;
    ncclUniqueId id;
    TSTAR_NCCL_CHECK(g_rccl.GetUniqueId(&id));
    memcpy(h_id, &id, sizeof(id));
    return TSTAR_OK;
}

int tstar_comm_create(tstar_comm** out, const void* h_id, int world, int rank) {
    TSTAR_REQUIRE(out && h_id, "tstar_comm_create: null argument");
    TSTAR_REQUIRE(world >= 1 && rank >= 0 && rank < world, "tstar_comm_create: rank must be in 0..world-1");
    if (int rc = load_rccl()) return rc;
    ncclUniqueId id;
    memcpy(&id, h_id, sizeof(id));
    tstar_comm* c = new tstar_comm();
    c->world = world; c->rank = rank;
    const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);      // collective over all ranks; uses the current device
    if (r != ncclSuccess) {
        set_error(std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r));
        delete c;
        return TSTAR_ERR_HIP;
    }
    *out = c;
    return TSTAR_OK;
}

int tstar_comm_destroy(tstar_comm* c) {
    if (!c) return TSTAR_OK;
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return TSTAR_OK;
}

int tstar_allgather_i32(tstar_comm* c, const int32_t* d_send, int32_t* d_recv, int count, void* stream) {
    TSTAR_REQUIRE(c && c->comm && d_send && d_recv, "tstar_allgather_i32: null argument");
    TSTAR_REQUIRE(count >= 1, "tstar_allgather_i32: count must be positive");
    TSTAR_NCCL_CHECK(g_rccl.AllGather(d_send, d_recv, (size_t)count, ncclInt32, c->comm, (hipStream_t)stream));
    return TSTAR_OK;
}

}  // extern "C"
