// tmac_comm.cpp — the exchange step of the row-sharded multi-GPU path behind the C-ABI (include/tmac_hip.h,
// tmac_hip_comm_*): RCCL all-gather over xGMI of what every rank needs whole before its next LUT build -- the activation
// block a row shard produced (or the int8 QLUT built from a K slice).  The reference is single-process; the axis this
// parallelises is the one its callers already split over threads (include/t-mac/tmac_gemm_wrapper.h:197-199: "split the
// blocks ... and pass the right ptr for scales, A and C"; python/t_mac/ops/qgemm.py:268-273).  Integer sums need no
// reduction: K is never split.
// RCCL is resolved at run time (dlopen): a single-GPU user of libtmac_hip.so needs no RCCL, and a process that already
// carries one (PyTorch ships its own librccl.so) keeps exactly one instance.
// No RCCL header is needed to BUILD the library either: the five entry points used are declared below with the signatures
// NCCL has kept stable since 2.0 (ncclUniqueId = 128 opaque bytes, passed by value; ncclInt8 = 0; ncclSuccess = 0).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>

#include "../../include/tmac_hip.h"

namespace {

// the slice of the NCCL / RCCL ABI this file binds at run time
struct ncclUniqueId { char internal[TMAC_HIP_COMM_ID_BYTES]; };
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
constexpr ncclResult_t ncclSuccess = 0;
constexpr int ncclInt8 = 0;
extern "C" {
typedef ncclResult_t (*ncclGetUniqueId_fn)(ncclUniqueId*);
typedef ncclResult_t (*ncclCommInitRank_fn)(ncclComm_t*, int, ncclUniqueId, int);
typedef ncclResult_t (*ncclAllGather_fn)(const void*, void*, size_t, int /* ncclDataType_t */, ncclComm_t, hipStream_t);
typedef ncclResult_t (*ncclCommDestroy_fn)(ncclComm_t);
typedef const char* (*ncclGetErrorString_fn)(ncclResult_t);
}

struct Rccl {
    void* h = nullptr;
    ncclGetUniqueId_fn GetUniqueId = nullptr;
    ncclCommInitRank_fn CommInitRank = nullptr;
    ncclAllGather_fn AllGather = nullptr;
    ncclCommDestroy_fn CommDestroy = nullptr;
    ncclGetErrorString_fn GetErrorString = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;
thread_local char g_comm_err[384] = "";

bool load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.h) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) { snprintf(g_comm_err, sizeof(g_comm_err), "RCCL not found (dlopen librccl.so.1: %s)", dlerror()); return false; }
    Rccl r;
    r.h = h;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy || !r.GetErrorString) {
        snprintf(g_comm_err, sizeof(g_comm_err), "librccl lacks an expected symbol");
        return false;
    }
    g_rccl = r;
    return true;
}

int32_t comm_fail(int32_t code, const char* what, ncclResult_t r) {
    snprintf(g_comm_err, sizeof(g_comm_err), "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error");
    return code;
}

}  // namespace

struct tmac_hip_comm {
    ncclComm_t c = nullptr;
    int rank = 0, world = 1;
};

extern "C" const char* tmac_hip_comm_last_error(void) { return g_comm_err; }

extern "C" int32_t tmac_hip_comm_unique_id(void* id_out) {
    if (!id_out) { snprintf(g_comm_err, sizeof(g_comm_err), "null argument"); return TMAC_HIP_E_ARG; }
    if (!load_rccl()) return TMAC_HIP_E_RUNTIME;
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return comm_fail(TMAC_HIP_E_RUNTIME, "ncclGetUniqueId", r);
    memcpy(id_out, &id, sizeof(id));
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_comm_init(tmac_hip_comm** out, const void* id, int rank, int world) {
    if (!out || !id || world < 1 || rank < 0 || rank >= world) { snprintf(g_comm_err, sizeof(g_comm_err), "bad arguments"); return TMAC_HIP_E_ARG; }
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { snprintf(g_comm_err, sizeof(g_comm_err), "no HIP device"); return TMAC_HIP_E_NODEVICE; }
    if (!load_rccl()) return TMAC_HIP_E_RUNTIME;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    auto* c = new tmac_hip_comm();
    c->rank = rank; c->world = world;
    const ncclResult_t r = g_rccl.CommInitRank(&c->c, world, uid, rank);      // one process per GPU: the caller has selected its device
    if (r != ncclSuccess) { delete c; return comm_fail(TMAC_HIP_E_RUNTIME, "ncclCommInitRank", r); }
    *out = c;
    return TMAC_HIP_OK;
}

namespace tmac_host { bool chain_record_gather_if_recording(const void*, void*, size_t, int, int, int32_t*); }

extern "C" int32_t tmac_hip_comm_allgather(tmac_hip_comm* c, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream) {
    if (!c || !send_dev || !recv_dev || !bytes_per_rank) { snprintf(g_comm_err, sizeof(g_comm_err), "bad arguments"); return TMAC_HIP_E_ARG; }
    int32_t rrc = 0;      // between tmac_hip_chain_begin and _end the exchange step becomes part of the recorded chain
    if (tmac_host::chain_record_gather_if_recording(send_dev, recv_dev, bytes_per_rank, c->rank, c->world, &rrc)) {
        if (rrc) snprintf(g_comm_err, sizeof(g_comm_err), "%s", tmac_hip_last_error());
        return rrc;
    }
    const ncclResult_t r = g_rccl.AllGather(send_dev, recv_dev, bytes_per_rank, ncclInt8, c->c, (hipStream_t)stream);
    if (r != ncclSuccess) return comm_fail(TMAC_HIP_E_RUNTIME, "ncclAllGather", r);
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_comm_destroy(tmac_hip_comm* c) {
    if (!c) return TMAC_HIP_OK;
    if (c->c) (void)g_rccl.CommDestroy(c->c);
    delete c;
    return TMAC_HIP_OK;
}
