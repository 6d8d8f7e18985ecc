// tmac_runtime.cpp — process-wide pieces of libtmac_hip.so: error text, device binding, the knob block, lifecycle entry
// points, ISA self-tests and tmac_hip_reset_state.
#include "tmac_host.h"

using namespace tmac_host;

static thread_local char g_err[512] = "";

int32_t tmac_host::fail(int32_t code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

std::mutex tmac_host::g_mu;
int tmac_host::g_device = -1;
Knobs tmac_host::g_knobs;

// tmac_hip_init selects the device for the calling thread only (hipSetDevice is per thread); entry points reached from other
// threads -- llama.cpp calls qgemm_lut_int8 from every worker -- bind to the same device on their first call.
void tmac_host::bind_thread_device() {
    static thread_local int bound = -2;
    const int d = g_device;
    if (d >= 0 && bound != d) {
        (void)hipSetDevice(d);
        bound = d;
    }
}

int32_t tmac_host::ensure_device() {
    bind_thread_device();
    if (g_device >= 0) return TMAC_HIP_OK;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(TMAC_HIP_E_NODEVICE, "no HIP device available (%s); libtmac_hip has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    int dev = 0;
    (void)hipGetDevice(&dev);
    g_device = dev;
    return TMAC_HIP_OK;
}
// ---------------------------------------------------------------------------------------------
// lifecycle
// ---------------------------------------------------------------------------------------------
extern "C" int32_t tmac_hip_init(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(TMAC_HIP_E_NODEVICE, "no HIP device available; libtmac_hip has no CPU path");
    if (device < 0 || device >= n) return fail(TMAC_HIP_E_ARG, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    g_device = device;
    return TMAC_HIP_OK;
}
extern "C" const char* tmac_hip_last_error(void) { return g_err; }
extern "C" const char* tmac_hip_version(void) { return "tmac_hip 0.1 (gfx950)"; }
extern "C" int32_t tmac_hip_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
extern "C" int32_t tmac_hip_pointer_on_device(const void* p) {
    if (!p) return 0;
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return 0; }   // (plain malloc memory: not registered with HIP)
    return at.type == hipMemoryTypeDevice ? 1 : 0;
}
extern "C" int32_t tmac_hip_set_fast_aggregation(int mode) {
    if (mode < 0 || mode > 2) return fail(TMAC_HIP_E_ARG, "unknown fast-aggregation mode %d", mode);
    g_knobs.fa_mode = mode;
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_set_variant(int variant) {
    if (variant < 0 || variant > 7) return fail(TMAC_HIP_E_ARG, "unknown variant %d", variant);
    g_knobs.variant = variant;
    return TMAC_HIP_OK;
}

// N at and above which qgemm runs the one-hot MFMA GEMM instead of looping the GEMV kernel (0 = never)
extern "C" int32_t tmac_hip_set_gemm_min_n(int n) {
    if (n < 0) return fail(TMAC_HIP_E_ARG, "gemm_min_n must be >= 0");
    g_knobs.gemm_min_n = n;
    return TMAC_HIP_OK;
}
extern "C" int32_t tmac_hip_selftest(const uint32_t* in_host, uint32_t* out_host, int n) {
    if (!in_host || !out_host || n <= 0) return fail(TMAC_HIP_E_ARG, "bad selftest arguments");
    int32_t rc = ensure_device();
    if (rc) return rc;
    DevBuf din, dout;
    const size_t bytes = sizeof(uint32_t) * 4 * (size_t)n;
    HIP_TRY(din.alloc(bytes));
    HIP_TRY(dout.alloc(bytes));
    HIP_TRY(hipMemcpy(din.p, in_host, bytes, hipMemcpyHostToDevice));
    hipError_t e = launch_selftest(din.as<uint32_t>(), dout.as<uint32_t>(), n, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out_host, dout.p, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(TMAC_HIP_E_RUNTIME, "selftest: %s", hipGetErrorString(e));
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_selftest_permlane(const uint32_t* in_host, uint32_t* out_host) {
    int32_t rc = ensure_device();
    if (rc) return rc;
    DevBuf din, dout;
    HIP_TRY(din.alloc(128 * 4));
    HIP_TRY(dout.alloc(256 * 4));
    HIP_TRY(hipMemcpy(din.p, in_host, 128 * 4, hipMemcpyHostToDevice));
    hipError_t e = launch_selftest_permlane(din.as<uint32_t>(), dout.as<uint32_t>(), nullptr);
    if (e == hipSuccess) e = hipMemcpy(out_host, dout.p, 256 * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(TMAC_HIP_E_RUNTIME, "selftest_permlane: %s", hipGetErrorString(e));
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_selftest_mfma(const uint32_t* in_host, int32_t* out_host) {
    if (!in_host || !out_host) return fail(TMAC_HIP_E_ARG, "bad selftest arguments");
    int32_t rc = ensure_device();
    if (rc) return rc;
    DevBuf din, dout;
    HIP_TRY(din.alloc(64 * 8 * 4));
    HIP_TRY(dout.alloc(64 * 4 * 4));
    HIP_TRY(hipMemcpy(din.p, in_host, 64 * 8 * 4, hipMemcpyHostToDevice));
    hipError_t e = launch_selftest_mfma(din.as<uint32_t>(), dout.as<int32_t>(), nullptr);
    if (e == hipSuccess) e = hipMemcpy(out_host, dout.p, 64 * 4 * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(TMAC_HIP_E_RUNTIME, "selftest_mfma: %s", hipGetErrorString(e));
    return TMAC_HIP_OK;
}
extern "C" int32_t tmac_hip_debug_pairs_min_n(int n) {
    g_knobs.pairs_min_n = n < 1 ? 1 : n;
    return TMAC_HIP_OK;
}

// measurement aid: a launch that only reads `bytes` from dev_src (sink: >= 4 KB of device scratch, practically never written)
extern "C" int32_t tmac_hip_debug_stream_read(const void* dev_src, size_t bytes, void* dev_sink, void* stream) {
    if (!dev_src || !dev_sink || bytes < 16) return fail(TMAC_HIP_E_ARG, "bad stream_read arguments");
    hipError_t e = launch_stream_read(dev_src, bytes, dev_sink, (hipStream_t)stream);
    if (e != hipSuccess) return fail(TMAC_HIP_E_RUNTIME, "stream_read launch: %s", hipGetErrorString(e));
    return TMAC_HIP_OK;
}

// A/B knobs of the quad kernel: threads per workgroup (512/1024) and waves per quad (1/2); 0 = heuristic
extern "C" int32_t tmac_hip_debug_quad_config(int force_ft, int force_wpq) {
    g_knobs.force_ft = force_ft; g_knobs.force_wpq = force_wpq;
    return TMAC_HIP_OK;
}

// debug/profiling: s_memtime phase stamps [nblocks][8] of the fused launches issued while enabled
extern "C" int32_t tmac_hip_debug_stamps(unsigned long long* dev_buffer) {
    if (dev_buffer && !g_knobs.stamp_dump) HIP_TRY(hipMalloc((void**)&g_knobs.stamp_dump, (size_t)256 << 20));   // scratch for the tap's stores
    g_knobs.stamps = dev_buffer;
    return TMAC_HIP_OK;
}

// ---------------------------------------------------------------------------------------------
// process-global state: one call puts all of it back to the state of a freshly loaded library
// ---------------------------------------------------------------------------------------------
extern "C" int32_t tmac_hip_debug_ws_fill_sync(int on) {
    g_knobs.ws_fill_sync = on ? 1 : 0;
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_reset_state(void) {
    (void)tmac_hip_defer(0);                  // (launches what the calling thread still has queued, then leaves deferred mode)
    defer_forget_all();
    int32_t rc = tmac_hip_cache_clear();      // host-pointer tiles / runs, the fused entry point's per-stream workspaces
    host_route_release();                     // ... and the host-pointer layer's workspace, staging buffers and LUT memo
    {
        std::lock_guard<std::mutex> lk(g_mu);
        kcfg_clear_locked();
        int32_t* keep = g_knobs.stamp_dump;   // a scratch allocation, not a setting
        g_knobs = Knobs();
        g_knobs.stamp_dump = keep;
    }
    (void)tmac_hip_tune_clear();
    return rc;
}
