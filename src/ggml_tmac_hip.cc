// ggml_tmac_hip.cc — device-resident ggml op-hook glue on top of libtmac_hip.so's C-ABI (include/ggml-tmac-hip.h).
// Plain C++ (no HIP headers): it binds tmac_hip.h exactly as a llama.cpp fork would.  The only device memory it owns is a
// staging pair (activations in, outputs out) for HOST tensors; tensors that already live in device memory are passed on as
// they are; weights live in tmac_hip_weights handles.
#include "../include/ggml-tmac-hip.h"

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "../include/tmac_hip.h"

namespace {

// the HIP runtime calls the glue needs (device + pinned buffers, copies, a stream), resolved from the runtime that
// libtmac_hip.so already brought into the process: the fork's build needs no hipcc and no HIP headers
struct Hip {
    int (*Malloc)(void**, size_t) = nullptr;
    int (*Free)(void*) = nullptr;
    int (*HostMalloc)(void**, size_t, unsigned) = nullptr;
    int (*HostFree)(void*) = nullptr;
    int (*MemcpyAsync)(void*, const void*, size_t, int, void*) = nullptr;
    int (*StreamCreateWithFlags)(void**, unsigned) = nullptr;
    int (*StreamSynchronize)(void*) = nullptr;
} hip;
bool g_ready = false;
void* g_stream = nullptr;       // the stream every launch and copy goes to: the glue's own, or the backend's (ggml_tmac_hip_set_stream)
void* g_own_stream = nullptr;
void *g_dx = nullptr, *g_dy = nullptr, *g_px = nullptr, *g_py = nullptr;
size_t g_nx = 0, g_ny = 0;
std::mutex g_mu;
thread_local char g_err[256] = "";

struct Handle {
    tmac_hip_weights* w;
    int M, K, bits;
};

int fail(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg); return -3; }

bool load_hip() {
    void* h = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libamdhip64.so.7", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libamdhip64.so.6", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return false;
#define SYM(f, n) (*(void**)(&hip.f) = dlsym(h, n))
    SYM(Malloc, "hipMalloc"); SYM(Free, "hipFree"); SYM(HostMalloc, "hipHostMalloc"); SYM(HostFree, "hipHostFree");
    SYM(MemcpyAsync, "hipMemcpyAsync"); SYM(StreamCreateWithFlags, "hipStreamCreateWithFlags"); SYM(StreamSynchronize, "hipStreamSynchronize");
#undef SYM
    return hip.Malloc && hip.Free && hip.HostMalloc && hip.HostFree && hip.MemcpyAsync && hip.StreamCreateWithFlags && hip.StreamSynchronize;
}

int grow(void** dev, void** pin, size_t* have, size_t need) {
    if (*have >= need) return 0;
    if (g_stream) hip.StreamSynchronize(g_stream);
    if (*dev) hip.Free(*dev);
    if (*pin) hip.HostFree(*pin);
    *dev = *pin = nullptr; *have = 0;
    if (hip.Malloc(dev, need) != 0 || hip.HostMalloc(pin, need, 0) != 0) return fail("staging allocation failed");
    *have = need;
    return 0;
}

}  // namespace

extern "C" const char* ggml_tmac_hip_last_error(void) { return g_err[0] ? g_err : tmac_hip_last_error(); }

extern "C" int ggml_tmac_hip_init(const char* kcfg_file, int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_err[0] = 0;
    int rc = tmac_hip_init(device);
    if (rc) return rc;
    if ((rc = tmac_hip_load_kcfg(kcfg_file))) return rc;
    if (!g_ready) {
        if (!load_hip()) return fail("HIP runtime not found");
        if (hip.StreamCreateWithFlags(&g_own_stream, 1 /* hipStreamNonBlocking */) != 0) return fail("stream creation failed");
        g_stream = g_own_stream;
        g_ready = true;
    }
    return 0;
}

extern "C" int ggml_tmac_hip_set_stream(void* hip_stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_ready) return fail("ggml_tmac_hip_init has not been called");
    if (hip.StreamSynchronize(g_stream) != 0) return fail("stream synchronisation failed");   // staging buffers and recorded segments may be in flight there
    g_stream = hip_stream ? hip_stream : g_own_stream;
    return 0;
}

extern "C" int ggml_tmac_hip_can_mul_mat(const struct tmac_ggml_tensor* w, int bits) {
    if (!w) return 0;
    tmac_kcfg c;
    return tmac_hip_get_kcfg((int)w->ne[1], (int)w->ne[0], 1, bits, &c) == 0;
}

extern "C" int ggml_tmac_hip_upload(struct tmac_ggml_tensor* w, int bits) {
    if (!w || !w->data) return fail("null tensor");
    g_err[0] = 0;
    const int M = (int)w->ne[1], K = (int)w->ne[0];
    tmac_kcfg c;
    int rc = tmac_hip_get_kcfg(M, K, 1, bits, &c);
    if (rc) return rc;
    // blob = [M * K * bits / 8 bytes of weight tiles][scales_size fp32] (python/t_mac/model_utils.py:243-271)
    const char* blob = (const char*)w->data;
    const void* scales = blob + (size_t)M * K * bits / 8;
    tmac_hip_weights* h = nullptr;
    rc = tmac_hip_register_weights(&h, blob, scales, M, K, bits, &c, TMAC_F32, TMAC_F32, nullptr);
    if (rc) return rc;
    w->extra = new Handle{h, M, K, bits};
    return 0;
}

extern "C" int ggml_tmac_hip_mul_mat(const struct tmac_ggml_tensor* w, const struct tmac_ggml_tensor* x, struct tmac_ggml_tensor* dst) {
    std::lock_guard<std::mutex> lk(g_mu);      // before ANY shared state (g_ready, g_err, the staging buffers) is touched
    if (!g_ready) return fail("ggml_tmac_hip_init has not been called");
    if (!w || !w->extra || !x || !x->data || !dst || !dst->data) return fail("null tensor");
    g_err[0] = 0;
    const Handle* h = (const Handle*)w->extra;
    const int N = (int)x->ne[1];
    if (x->ne[0] != h->K || dst->ne[0] != h->M || dst->ne[1] != N) return fail("shape mismatch");
    const size_t bx = sizeof(float) * (size_t)N * h->K, by = sizeof(float) * (size_t)N * h->M;
    // Tensors of a device backend are passed on as they are; host tensors (ggml's CPU buffers, what the reference's fork hands over) are
    // staged through pinned memory.  Either way the call returns when dst holds the result.
    const bool x_dev = tmac_hip_pointer_on_device(x->data) == 1, y_dev = tmac_hip_pointer_on_device(dst->data) == 1;
    int rc;
    const void* xin = x->data;
    void* yout = dst->data;
    if (!x_dev) {
        if ((rc = grow(&g_dx, &g_px, &g_nx, bx))) return rc;
        memcpy(g_px, x->data, bx);
        if (hip.MemcpyAsync(g_dx, g_px, bx, 1 /* H2D */, g_stream) != 0) return fail("H2D copy failed");
        xin = g_dx;
    }
    if (!y_dev) {
        if ((rc = grow(&g_dy, &g_py, &g_ny, by))) return rc;
        yout = g_dy;
    }
    const tmac_hip_weights* wl[1] = {h->w};
    void* cl[1] = {yout};
    if ((rc = tmac_hip_qgemm_fused_dev(wl, 1, xin, TMAC_F32, cl, TMAC_F32, N, g_stream))) return rc;   // LUT build + mpGEMM, one launch at N = 1
    if ((rc = tmac_hip_flush(g_stream))) return rc;          // (deferred mode: an N = 1 call was queued, not launched)
    if (!y_dev && hip.MemcpyAsync(g_py, g_dy, by, 2 /* D2H */, g_stream) != 0) return fail("D2H copy failed");
    if (hip.StreamSynchronize(g_stream) != 0) return fail("stream synchronisation failed");
    if (!y_dev) memcpy(dst->data, g_py, by);
    return 0;
}

// ---- device-resident mat-muls without recording (see the header) ----
extern "C" int ggml_tmac_hip_mul_mat_dev(const struct tmac_ggml_tensor* const* w, int nw, const void* x_dev, int x_is_f32, void* const* dst_dev, int dst_is_f32) {
    if (!g_ready) return fail("ggml_tmac_hip_init has not been called");
    if (!w || nw < 1 || nw > 4 || !x_dev || !dst_dev) return fail("bad mul_mat_dev");
    g_err[0] = 0;
    const tmac_hip_weights* wl[4];
    void* cl[4];
    for (int i = 0; i < nw; ++i) {
        if (!w[i] || !w[i]->extra || !dst_dev[i]) return fail("null tensor");
        wl[i] = ((const Handle*)w[i]->extra)->w;
        cl[i] = dst_dev[i];
    }
    return tmac_hip_qgemm_fused_dev(wl, nw, x_dev, x_is_f32 ? TMAC_F32 : TMAC_F16, cl, dst_is_f32 ? TMAC_F32 : TMAC_F16, 1, g_stream);
}
extern "C" int ggml_tmac_hip_set_deferred(int on) { return tmac_hip_defer(on); }
extern "C" int ggml_tmac_hip_flush(void) { return tmac_hip_flush(g_stream); }
extern "C" int ggml_tmac_hip_synchronize(void) {
    if (!g_ready) return fail("ggml_tmac_hip_init has not been called");
    const int rc = tmac_hip_flush(g_stream);
    if (rc) return rc;
    return hip.StreamSynchronize(g_stream) == 0 ? 0 : fail("stream synchronisation failed");
}

extern "C" void ggml_tmac_hip_free(struct tmac_ggml_tensor* w) {
    if (!w || !w->extra) return;
    Handle* h = (Handle*)w->extra;
    if (g_stream) hip.StreamSynchronize(g_stream);
    tmac_hip_free_weights(h->w);
    delete h;
    w->extra = nullptr;
}

// ---- decoder segments: thin glue over tmac_hip_chain_* / tmac_hip_chain_xform (see the header) ----
struct ggml_tmac_hip_segment { tmac_hip_chain* chain; };

extern "C" void* ggml_tmac_hip_stream(void) { return g_stream; }

extern "C" int ggml_tmac_hip_segment_begin(void) {
    if (!g_ready) return fail("ggml_tmac_hip_init has not been called");
    g_err[0] = 0;
    return tmac_hip_chain_begin();
}

extern "C" int ggml_tmac_hip_segment_norm(const float* residual, int residual_is_kept, const float* norm_weight, float eps, float* residual_out, int keep) {
    tmac_hip_xform xf;
    memset(&xf, 0, sizeof(xf));
    xf.kind = TMAC_XF_NORM;
    xf.residual = residual_is_kept ? TMAC_XF_CARRY : residual;
    xf.gamma = norm_weight; xf.eps = eps; xf.residual_out = residual_out; xf.keep = keep;
    return tmac_hip_chain_xform(&xf);
}

extern "C" int ggml_tmac_hip_segment_glu(const void* in2_f16) {
    tmac_hip_xform xf;
    memset(&xf, 0, sizeof(xf));
    xf.kind = TMAC_XF_GLU;
    xf.in2 = in2_f16;
    return tmac_hip_chain_xform(&xf);
}

static int segment_mul_mat(const struct tmac_ggml_tensor* const* w, int nw, const void* x, tmac_dtype_t x_dtype, void* const* dst_f16) {
    // a segment is all or nothing (include/ggml-tmac-hip.h): EVERY non-zero return ends the recording, so that a caller who falls back
    // to the graph's own nodes does not leave the thread recording (later fused calls would be noted instead of launched)
    if (!w || nw < 1 || nw > 4 || !x || !dst_f16) { (void)tmac_hip_chain_abort(); return fail("bad segment mul_mat"); }
    const tmac_hip_weights* wl[4];
    void* cl[4];
    for (int i = 0; i < nw; ++i) {
        if (!w[i] || !w[i]->extra || !dst_f16[i]) { (void)tmac_hip_chain_abort(); return fail("null tensor"); }
        wl[i] = ((const Handle*)w[i]->extra)->w;
        cl[i] = dst_f16[i];
    }
    const int rc = tmac_hip_qgemm_fused_dev(wl, nw, x, x_dtype, cl, TMAC_F16, 1, g_stream);   // recorded, not launched
    if (rc) (void)tmac_hip_chain_abort();     // a segment is all or nothing: the recording ends here, the caller runs the graph's own nodes
    return rc;
}

extern "C" int ggml_tmac_hip_segment_mul_mat(const struct tmac_ggml_tensor* const* w, int nw, const void* x_f16, void* const* dst_f16) {
    return segment_mul_mat(w, nw, x_f16, TMAC_F16, dst_f16);
}

extern "C" int ggml_tmac_hip_segment_mul_mat_f32(const struct tmac_ggml_tensor* const* w, int nw, const float* x_f32, void* const* dst_f16) {
    return segment_mul_mat(w, nw, x_f32, TMAC_F32, dst_f16);
}

extern "C" int ggml_tmac_hip_segment_abort(void) { return tmac_hip_chain_abort(); }

extern "C" int ggml_tmac_hip_segment_end(ggml_tmac_hip_segment** seg) {
    if (!seg) { (void)tmac_hip_chain_abort(); return fail("null argument"); }
    tmac_hip_chain* c = nullptr;
    int rc = tmac_hip_chain_end(&c);      // (ends the recording whether or not a chain comes out of it)
    if (rc) return rc;
    *seg = new ggml_tmac_hip_segment{c};
    return 0;
}

extern "C" int ggml_tmac_hip_segment_compute(ggml_tmac_hip_segment* seg) {
    if (!seg || !seg->chain) return fail("null segment");
    return tmac_hip_chain_launch(seg->chain, g_stream);
}

extern "C" int ggml_tmac_hip_segment_wait(ggml_tmac_hip_segment* seg) {
    if (!seg || !seg->chain) return fail("null segment");
    if (hip.StreamSynchronize(g_stream) != 0) return fail("stream synchronisation failed");
    uint32_t word = 0;
    int rc = tmac_hip_chain_status(seg->chain, &word);
    if (rc) return rc;
    return word ? fail("a hand-off inside the segment timed out") : 0;
}

extern "C" void ggml_tmac_hip_segment_free(ggml_tmac_hip_segment* seg) {
    if (!seg) return;
    if (g_stream) hip.StreamSynchronize(g_stream);
    if (seg->chain) tmac_hip_chain_free(seg->chain);
    delete seg;
}
