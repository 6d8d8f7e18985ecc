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
    // transport "ipc": every rank owns a window (two halves + two flag words, fine-grained device memory) that all peers map
    bool ipc = false, connected = false;
    unsigned char* win = nullptr;          // [2][win_half] + flags behind them
    size_t win_half = 0;
    void* peer[8] = {nullptr};             // peers' windows as mapped here (own slot: win)
    unsigned* err = nullptr;
    unsigned gen = 0;
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
#include "tmac_kernels.h"

// ---- transport "ipc": the same all-gather without RCCL -- every rank's window is mapped by all peers (hipIpc*), a small kernel
// publishes this rank's part and copies the peers' parts out of their windows.  Serves nodes (and tests) where RCCL cannot be used:
// e.g. several ranks on ONE device, which RCCL refuses.  Protocol: tmac_hip_comm_init_ipc on every rank, tmac_hip_comm_export,
// all-gather the blobs over any transport (rank order), tmac_hip_comm_connect; from then on tmac_hip_comm_allgather as usual.
struct CommBlob {
    hipIpcMemHandle_t handle;
    unsigned long long win_half;
    int rank, world;
};
static_assert(sizeof(CommBlob) <= TMAC_HIP_COMM_BLOB_BYTES, "blob size");

extern "C" int32_t tmac_hip_comm_init_ipc(tmac_hip_comm** out, size_t max_bytes_per_rank, int rank, int world) {
    if (!out || !max_bytes_per_rank || world < 1 || world > 8 || rank < 0 || rank >= world) { snprintf(g_comm_err, sizeof(g_comm_err), "bad arguments (1..8 ranks)"); return TMAC_HIP_E_ARG; }
    *out = nullptr;
    auto* c = new tmac_hip_comm();
    c->rank = rank; c->world = world; c->ipc = true;
    c->win_half = (max_bytes_per_rank + 255) & ~(size_t)255;
    const size_t total = 2 * c->win_half + 256;
    if (hipExtMallocWithFlags((void**)&c->win, total, hipDeviceMallocFinegrained) != hipSuccess || hipMemset(c->win, 0, total) != hipSuccess ||
        hipMalloc((void**)&c->err, 256) != hipSuccess || hipMemset(c->err, 0, 256) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        snprintf(g_comm_err, sizeof(g_comm_err), "window allocation failed (%zu bytes)", total);
        if (c->win) (void)hipFree(c->win);
        if (c->err) (void)hipFree(c->err);
        delete c;
        return TMAC_HIP_E_RUNTIME;
    }
    c->peer[rank] = c->win;
    c->connected = world == 1;
    *out = c;
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_comm_export(const tmac_hip_comm* c, void* blob_out) {
    if (!c || !blob_out || !c->ipc) { snprintf(g_comm_err, sizeof(g_comm_err), "not an IPC communicator"); return TMAC_HIP_E_ARG; }
    CommBlob b;
    memset(&b, 0, sizeof(b));
    if (hipIpcGetMemHandle(&b.handle, c->win) != hipSuccess) { snprintf(g_comm_err, sizeof(g_comm_err), "hipIpcGetMemHandle failed (HSA_ENABLE_IPC_MODE_LEGACY=0?)"); return TMAC_HIP_E_RUNTIME; }
    b.win_half = c->win_half; b.rank = c->rank; b.world = c->world;
    memset(blob_out, 0, TMAC_HIP_COMM_BLOB_BYTES);
    memcpy(blob_out, &b, sizeof(b));
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_comm_connect(tmac_hip_comm* c, const void* blobs, int world) {
    if (!c || !blobs || !c->ipc || world != c->world) { snprintf(g_comm_err, sizeof(g_comm_err), "bad arguments"); return TMAC_HIP_E_ARG; }
    if (c->connected && c->world > 1) { snprintf(g_comm_err, sizeof(g_comm_err), "already connected"); return TMAC_HIP_E_ARG; }
    for (int r = 0; r < world; ++r) {
        CommBlob b;
        memcpy(&b, (const char*)blobs + (size_t)r * TMAC_HIP_COMM_BLOB_BYTES, sizeof(b));
        if (b.rank != r || b.world != world || b.win_half != c->win_half) { snprintf(g_comm_err, sizeof(g_comm_err), "rank %d's blob does not match (rank, world or window size)", r); return TMAC_HIP_E_ARG; }
        if (r == c->rank) continue;
        void* p = nullptr;
        if (hipIpcOpenMemHandle(&p, b.handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { snprintf(g_comm_err, sizeof(g_comm_err), "hipIpcOpenMemHandle of rank %d's window failed", r); return TMAC_HIP_E_RUNTIME; }
        c->peer[r] = p;
    }
    c->connected = true;
    return TMAC_HIP_OK;
}

// 0 if every part of the all-gathers so far arrived; else a bit per rank whose part timed out (cleared by the call)
extern "C" int32_t tmac_hip_comm_status(tmac_hip_comm* c, uint32_t* error_word) {
    if (!c || !error_word) { snprintf(g_comm_err, sizeof(g_comm_err), "null argument"); return TMAC_HIP_E_ARG; }
    *error_word = 0;
    if (!c->ipc) return TMAC_HIP_OK;
    unsigned w = 0;
    if (hipMemcpy(&w, c->err, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) return TMAC_HIP_E_RUNTIME;
    if (w) (void)hipMemset(c->err, 0, sizeof(w));
    *error_word = w;
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_comm_allgather(tmac_hip_comm* c, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream) {
    if (!c || !send_dev || !recv_dev || !bytes_per_rank) { snprintf(g_comm_err, sizeof(g_comm_err), "bad arguments"); return TMAC_HIP_E_ARG; }
    int32_t rrc = 0;      // between tmac_hip_chain_begin and _end the exchange step becomes part of the recorded chain
    if (tmac_host::chain_record_gather_if_recording(send_dev, recv_dev, bytes_per_rank, c->rank, c->world, &rrc)) {
        if (rrc) snprintf(g_comm_err, sizeof(g_comm_err), "%s", tmac_hip_last_error());
        return rrc;
    }
    if (c->ipc) {
        if (!c->connected) { snprintf(g_comm_err, sizeof(g_comm_err), "the IPC communicator is not connected to its peers"); return TMAC_HIP_E_ARG; }
        if (bytes_per_rank > c->win_half) { snprintf(g_comm_err, sizeof(g_comm_err), "%zu bytes per rank exceed the window (%zu)", bytes_per_rank, c->win_half); return TMAC_HIP_E_ARG; }
        tmac::IpcGatherArgs a;
        memset(&a, 0, sizeof(a));
        a.send = (const unsigned char*)send_dev; a.recv = (unsigned char*)recv_dev; a.bytes = bytes_per_rank;
        for (int r = 0; r < c->world; ++r) {
            a.win[r] = (unsigned char*)c->peer[r];
            a.flag[r] = (unsigned*)((unsigned char*)c->peer[r] + 2 * c->win_half);
        }
        a.win_half = c->win_half; a.rank = c->rank; a.world = c->world;
        a.gen = ++c->gen;                               // (every rank calls the same sequence of all-gathers)
        a.spin_limit = 1u << 22;                        // ~ seconds
        a.err = c->err;
        const hipError_t e = tmac::launch_ipc_allgather(a, (hipStream_t)stream);
        if (e != hipSuccess) { snprintf(g_comm_err, sizeof(g_comm_err), "IPC all-gather launch: %s", hipGetErrorString(e)); return TMAC_HIP_E_RUNTIME; }
        return TMAC_HIP_OK;
    }
    const ncclResult_t r = g_rccl.AllGather(send_dev, recv_dev, bytes_per_rank, ncclInt8, c->c, (hipStream_t)stream);
    if (r != ncclSuccess) return comm_fail(TMAC_HIP_E_RUNTIME, "ncclAllGather", r);
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_comm_destroy(tmac_hip_comm* c) {
    if (!c) return TMAC_HIP_OK;
    if (c->c) (void)g_rccl.CommDestroy(c->c);
    if (c->ipc) {
        for (int r = 0; r < c->world; ++r) if (r != c->rank && c->peer[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
        if (c->win) (void)hipFree(c->win);
        if (c->err) (void)hipFree(c->err);
    }
    delete c;
    return TMAC_HIP_OK;
}
