// tmac_host.h — internals shared by the host-side translation units of libtmac_hip.so (nothing here is exported; the
// export list is include/tmac_hip.h).
//
//   tmac_runtime.cpp     error text, device binding, the knob block, lifecycle, self-tests, tmac_hip_reset_state
//   tmac_kcfg.cpp        the kcfg.ini table and its lookups
//   tmac_weights.cpp     weight registration (reference layout -> device layout)
//   tmac_workspace.cpp   LUT workspace + the preprocessor entry point
//   tmac_dispatch.cpp    qgemm_lut dispatch, the fused entry point, parity taps
//   tmac_tuner.cpp       launch-configuration tuner of the decode kernel
//   tmac_chain_host.cpp  recording / building / launching the persistent decode chain
//   tmac_hostptr.cpp     the reference-named host-pointer entry points (struct HostRoute)
//   tmac_comm.cpp        multi-GPU exchange step (RCCL)
#pragma once
#include <hip/hip_runtime.h>

#include <array>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/tmac_hip.h"
#include "tmac_chain.h"
#include "tmac_kernels.h"

// ---- the C-ABI's opaque objects ---------------------------------------------------------------
struct tmac_hip_weights {
    tmac::Shape s;
    void* W = nullptr;      // device layout weights
    void* SC = nullptr;     // device layout scales
    tmac::Dtype sc_dtype = tmac::F32;
    void* A_ref = nullptr;  // reference blobs kept on the device only when the generic kernel needs them
    void* S_ref = nullptr;
    tmac::Dtype ref_dtype = tmac::F32;
    bool lo_ok = false;
    int fa = 0;             // fast-aggregation mode these weights were registered under (0 = exact)
    size_t w_bytes = 0, sc_bytes = 0;
};

struct tmac_hip_workspace {
    int maxK = 0, maxN = 0;
    int8_t* qlut_ref = nullptr;  // int8 [maxN][maxK/4][16]
    void* qlut_dev = nullptr;    // uint4 [maxN][qdev_u4(maxK)]   (ts = 16 layout)
    void* qlut_lds = nullptr;    // uint4 [maxN][qlut_lds_u4(K)]  (LDS image for the fused-layout kernel)
    float* lut_scales = nullptr; // fp32 [maxN][maxK/32]
    float* lut_biases = nullptr;
    int32_t* dump = nullptr;     // lazily sized parity tap
    size_t dump_elems = 0;
    int K = 0, N = 0, ags = 0;   // what the LUT currently holds
    size_t qdev_u4_per_row = 0;
    // the LUT as k_gemm_planes streams it (tmac_gemm2.hip): built next to the layouts above when N > 1 and ags = 64
    void* gimg = nullptr;        // 2 maxK gNpad bytes: the LUT image of the plane-combined GEMM (layouts: Gemm2Args, tmac_kernels.h)
    float* gcol = nullptr;       // 4 (maxK / 64) gNpad floats: its column values
    int gNpad = 0;
    bool gimg_valid = false;
    int gimg_kind = 0;           // 1: row-wise image (one act group per row, k_gemm_planes_us) | 2: chunk-major image (act groups of 64, k_gemm_planes)
};

namespace tmac_host {
using namespace tmac;

// ---- errors -----------------------------------------------------------------------------------
int32_t fail(int32_t code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));   // sets the thread's message, returns code

// device scratch of the test taps and self-tests: freed on every return path
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return ::tmac_host::fail(TMAC_HIP_E_RUNTIME, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                     __FILE__, __LINE__);                                          \
    } while (0)

// ---- process-wide state -----------------------------------------------------------------------
// Every tmac_hip_set_* / tmac_hip_debug_* setting lives in this one block; tmac_hip_reset_state() assigns a fresh one.
struct Knobs {
    int variant = V_AUTO;          // tmac_hip_set_variant
    int gemm_min_n = 32;           // tmac_hip_set_gemm_min_n: measured crossover on MI355X (llama-2-7B W2 shapes): GEMV loop ~1.7 us per row, GEMM 40-60 us up to 64 rows
    int gemm_kernel = 0;           // N > 1 kernel: 0 = k_gemm_planes where it covers the configuration, 1 = k_gemm_onehot, 2 / 3 = k_gemm_planes forced to eight- / four-wave workgroups (tmac_hip_debug_gemm_kernel)
    int pairs_min_n = 2;           // tmac_hip_preprocessor_dev: rows from which the pair-wise LUT build is used (tmac_hip_debug_pairs_min_n)
    int fa_mode = 0;               // fast aggregation for weights registered from now on (tmac_hip_set_fast_aggregation)
    int force_ft = 0, force_wpq = 0;   // A/B knobs of the quad kernel (0 = heuristic)
    int host_runs = 1;             // host-pointer layer: group contiguous tiles into runs (tmac_hip_debug_host_runs)
    int ws_fill_sync = 1;          // test knob (tmac_hip_debug_ws_fill_sync): 0 re-opens the round-2 race between the workspace fills and its first user
    int chain_wpq = 0;             // waves per row quad for every op of chains built from now on (0 = per-op choice)
    unsigned chain_spin_limit = 1u << 18;   // polls of one hand-off before a wave gives up (~0.4 s)
    int chain_grid = 0;            // workgroups of chains built from now on (0 = one per CU; tests run two chains side by side on one device)
    unsigned long long* stamps = nullptr;        // phase stamps of the next fused launches (tmac_hip_debug_stamps)
    int32_t* stamp_dump = nullptr;               // scratch the stamp instantiation stores its tap into (allocated once, survives resets)
    unsigned long long* gemm_stamps = nullptr;   // k_gemm_planes step stamps (tmac_hip_debug_gemm_stamps)
};
extern Knobs g_knobs;
extern std::mutex g_mu;       // kcfg table, the fused entry point's workspace map, the slow path of the host-pointer layer
extern int g_device;

// tmac_hip_init selects the device for the calling thread only (hipSetDevice is per thread); entry points reached from other
// threads -- llama.cpp calls qgemm_lut_int8 from every worker -- bind to the same device on their first call.
void bind_thread_device();
int32_t ensure_device();

constexpr int PLANES_MIN_N = 12;   // default GEMM threshold where k_gemm_planes covers the configuration (tmac_dispatch.cpp)
inline size_t qdev_u4_for_K(int K) { return (size_t)((K / (4 * TS) + KL - 1) / KL) * 8 * KL; }

// ---- kcfg (tmac_kcfg.cpp; callers hold g_mu) -----------------------------------------------------
// 1 = found, 0 = no section matches, -1 = several sections match and disagree on what the bytes mean
int find_cfg(int k, int n, int b, int bm_filter, int m_filter, tmac_kcfg* out, bool act_only = false);
void kcfg_clear_locked();

// ---- weights (tmac_weights.cpp) -------------------------------------------------------------------
int32_t make_shape(Shape& s, int Mw, int K, int bits, const tmac_kcfg* cfg);
size_t ref_weight_bytes(const Shape& s);
size_t ref_scale_elems(const Shape& s);
size_t dt_size(Dtype d);
int32_t register_impl(tmac_hip_weights** out, const void* A_ref, const void* scales_ref, bool src_on_device, int Mw, int K, int bits,
                      const tmac_kcfg* cfg, tmac_dtype_t host_float, tmac_dtype_t dev_float, void* stream);

// ---- workspace (tmac_workspace.cpp) ---------------------------------------------------------------
int32_t check_lut_shape(tmac_hip_workspace* ws, int K, int N, int ags);

// ---- dispatch (tmac_dispatch.cpp) -----------------------------------------------------------------
int32_t qgemm_impl(const tmac_hip_weights* w, const tmac_hip_workspace* ws, void* C_dev, tmac_dtype_t out_dtype, int N, int32_t* dump,
                   hipStream_t st);
int32_t fused_impl(const tmac_hip_weights* const* wl, int nmat, const void* B_dev, tmac_dtype_t act_dtype, void* const* C_list,
                   tmac_dtype_t out_dtype, int N, int32_t* dump, float* lut_tap, hipStream_t st);
void release_fused_workspaces();   // caller holds g_mu

// ---- tuner (tmac_tuner.cpp) -----------------------------------------------------------------------
void tuned_config(const FusedArgs& fa, int total_q, int& ft, int& wpq);   // leaves ft / wpq alone when nothing is recorded

// ---- decode chain (tmac_chain_host.cpp) -----------------------------------------------------------
bool chain_recording();            // is the calling thread between tmac_hip_chain_begin and tmac_hip_chain_end?
void chain_clear_xform();          // drops a transform declared for the next recorded call (the call was rejected before it could be recorded)
int32_t chain_record(const tmac_hip_weights* const* wl, int nmat, const void* B_dev, tmac_dtype_t act_dtype, void* const* C_list,
                     tmac_dtype_t out_dtype, int N);
// true (and *rc set) when the calling thread is recording: the exchange step was noted, not executed
// deferred launches (tmac_hip_defer): true (and *rc set) when the call was queued instead of launched
bool defer_if_on(const tmac_hip_weights* const* wl, int nmat, const void* B_dev, tmac_dtype_t act_dtype, void* const* C_list, tmac_dtype_t out_dtype,
                 int N, hipStream_t st, int32_t* rc);
void defer_release_thread();       // frees the calling thread's cached recordings (tmac_hip_cache_clear)
void defer_forget_all();           // weights were freed / the library was reset: cached recordings of every thread are stale from now on
bool chain_record_gather_if_recording(const void* send_dev, void* recv_dev, size_t bytes_per_rank, int rank, int world, int32_t* rc);

// ---- host-pointer layer (tmac_hostptr.cpp) --------------------------------------------------------
void host_route_release();         // frees the layer's workspace, staging buffers and LUT memo (tiles / runs: tmac_hip_cache_clear)

}  // namespace tmac_host
