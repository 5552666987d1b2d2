// collective.hip -- the RCCL form of the elite all-gather (icem_allgather_elites): the ranks' K candidate
// records {cost, gidx, actions[h*d]} gathered in place with ncclAllGather on the launch stream, between an iteration's
// record pack and its merge.  It is the fallback of exchange.hip's hand-rolled peer-to-peer exchange: a rank whose
// peers' blocks do not IPC-map (or whose exchange self-test fails) still runs an MPC step as ONE C call with zero
// host-side collectives -- icem_plan_step_sharded picks this path when a communicator is connected and the exchange
// is not.  Reference analogue: the pipe gather of icem/models/gt_par_model.py:77-94.
//
// RCCL is bound at run time (dlopen), never at link time: single-GPU users and the CPU-side ABI tests must not need
// librccl.so, and a process that already carries torch's copy must use THAT copy (two RCCL instances in one process
// fight over the same devices).  Search order: a library with soname librccl.so.1 already loaded in the process, the path
// given to icem_rccl_load / $ICEM_RCCL_LIB, then the loader's default search for librccl.so.1.
#include <dlfcn.h>

#include <mutex>

#include "host_common.h"

namespace icem {

namespace {

// the slice of the NCCL API this file uses (rccl.h, /opt/rocm/include/rccl): declared here so that building the library
// needs no RCCL headers either
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;  // 0 = ncclSuccess
constexpr int kNcclChar = 0;  // ncclInt8 / ncclChar: the records travel as bytes (any handle dtype)
struct NcclUniqueId {
    char internal[ICEM_RCCL_ID_BYTES];
};

struct Rccl {
    void* so = nullptr;
    std::string from;
    ncclResult_t (*GetUniqueId)(NcclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, NcclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;
std::mutex g_rccl_mu;

int rccl_bind(const char* path_or_null) {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.so) return ICEM_OK;
    std::vector<std::pair<std::string, int>> tries;
    tries.push_back({"librccl.so.1", RTLD_NOW | RTLD_NOLOAD});  // the copy this process already carries (torch's)
    tries.push_back({"librccl.so", RTLD_NOW | RTLD_NOLOAD});
    if (path_or_null && *path_or_null) tries.push_back({path_or_null, RTLD_NOW | RTLD_GLOBAL});
    tries.push_back({"librccl.so.1", RTLD_NOW | RTLD_GLOBAL});
    tries.push_back({"/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL});
    std::string errs;
    for (const auto& t : tries) {
        void* so = dlopen(t.first.c_str(), t.second);
        if (!so) {
            const char* de = dlerror();
            if (!(t.second & RTLD_NOLOAD)) errs += std::string(" [") + t.first + ": " + (de ? de : "?") + "]";
            continue;
        }
        Rccl r;
        r.so = so;
        r.from = t.first + ((t.second & RTLD_NOLOAD) ? " (already loaded)" : "");
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(so, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(so, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(so, "ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))dlsym(so, "ncclAllGather");
        r.CommCount = (decltype(r.CommCount))dlsym(so, "ncclCommCount");
        r.CommUserRank = (decltype(r.CommUserRank))dlsym(so, "ncclCommUserRank");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(so, "ncclGetErrorString");
        if (r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.CommCount && r.CommUserRank && r.GetErrorString) {
            g_rccl = r;
            return ICEM_OK;
        }
        errs += " [" + t.first + ": NCCL symbols missing]";
        dlclose(so);
    }
    return fail(ICEM_E_UNSUPPORTED, "RCCL not available:" + errs);
}

#define ICEM_NCCL_TRY(expr)                                                                                       \
    do {                                                                                                          \
        ncclResult_t r_ = (expr);                                                                                 \
        if (r_ != 0) return ::icem::fail(ICEM_E_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_));      \
    } while (0)

}  // namespace

bool rccl_connected(const icem_handle* h) { return h->rccl_comm != nullptr; }

void rccl_release(icem_handle* h) {
    if (h->rccl_comm && h->rccl_owned && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)h->rccl_comm);
    h->rccl_comm = nullptr;
    h->rccl_owned = false;
}

// this rank's K records (slot `rank` of `records`) -> every rank's `records`, in place, on `st`
int rccl_allgather_records(icem_handle* h, void* records, hipStream_t st) {
    if (!h->rccl_comm) return fail(ICEM_E_STATE, "icem_rccl_connect / icem_rccl_adopt has not been called");
    const size_t bytes = (size_t)h->cfg.num_elites * (h->hd + 2) * h->tsize;
    unsigned char* base = (unsigned char*)records;
    ICEM_NCCL_TRY(g_rccl.AllGather(base + (size_t)h->cfg.rank * bytes, base, bytes, kNcclChar, (ncclComm_t)h->rccl_comm, st));
    return ICEM_OK;
}

}  // namespace icem

using namespace icem;

extern "C" {

int icem_rccl_load(const char* path_or_null) { return rccl_bind(path_or_null); }

const char* icem_rccl_library(void) { return g_rccl.so ? g_rccl.from.c_str() : ""; }

int icem_rccl_unique_id(void* id_out_host) {
    if (!id_out_host) return fail(ICEM_E_INVALID, "null output");
    static_assert(sizeof(NcclUniqueId) == ICEM_RCCL_ID_BYTES, "ncclUniqueId size");
    int rc = rccl_bind(nullptr);
    if (rc) return rc;
    NcclUniqueId id;
    ICEM_NCCL_TRY(g_rccl.GetUniqueId(&id));
    std::memcpy(id_out_host, &id, sizeof(id));
    return ICEM_OK;
}

int icem_rccl_connect(icem_handle* h, const void* id_host) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (!id_host) return fail(ICEM_E_INVALID, "null id");
    if (h->pm_pending) return fail(ICEM_E_STATE, "a deferred merge is pending: finish the MPC step first");
    int rc = rccl_bind(nullptr);
    if (rc) return rc;
    rccl_release(h);
    NcclUniqueId id;
    std::memcpy(&id, id_host, sizeof(id));
    ncclComm_t comm = nullptr;
    ICEM_NCCL_TRY(g_rccl.CommInitRank(&comm, h->cfg.world, id, h->cfg.rank));
    h->rccl_comm = comm;
    h->rccl_owned = true;
    return ICEM_OK;
}

int icem_rccl_adopt(icem_handle* h, void* nccl_comm) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (!nccl_comm) return fail(ICEM_E_INVALID, "null communicator");
    int rc = rccl_bind(nullptr);
    if (rc) return rc;
    int n = 0, r = -1;
    ICEM_NCCL_TRY(g_rccl.CommCount((ncclComm_t)nccl_comm, &n));
    ICEM_NCCL_TRY(g_rccl.CommUserRank((ncclComm_t)nccl_comm, &r));
    if (n != h->cfg.world || r != h->cfg.rank)
        return fail(ICEM_E_INVALID, "the communicator's size / rank differ from the handle's world / rank");
    rccl_release(h);
    h->rccl_comm = nccl_comm;
    h->rccl_owned = false;
    return ICEM_OK;
}

int icem_rccl_disconnect(icem_handle* h) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (h->pm_pending) return fail(ICEM_E_STATE, "a deferred merge is pending: finish the MPC step first");
    rccl_release(h);
    return ICEM_OK;
}

int icem_allgather_elites(icem_handle* h, void* records, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (!records) return fail(ICEM_E_INVALID, "null records");
    return rccl_allgather_records(h, records, (hipStream_t)stream);
}

}  // extern "C"
