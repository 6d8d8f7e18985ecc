// tmac_weights.cpp — weight registration: reference-layout blobs (python/t_mac/weights.py:57-87) -> device layout.
#include "tmac_host.h"

using namespace tmac_host;

int32_t tmac_host::make_shape(Shape& s, int Mw, int K, int bits, const tmac_kcfg* cfg) {
    if (!cfg) return fail(TMAC_HIP_E_ARG, "null kcfg");
    memset(&s, 0, sizeof(s));
    s.Mw = Mw; s.K = K; s.bits = bits; s.bm = cfg->bm; s.kfactor = cfg->kfactor;
    s.m_groups = cfg->m_groups >= 1 ? cfg->m_groups : -1;
    s.gs = s.m_groups >= 1 ? 0 : cfg->group_size;
    s.ags = cfg->act_group_size > 0 ? cfg->act_group_size : 64;
    s.zero_point = (s.m_groups >= 1) ? 0 : (cfg->zero_point ? 1 : 0);
    // the reference's own shape constraints (python/t_mac/ops/qgemm.py:118-129, weights.py:57-73)
    if (bits < 1 || bits > 4 || Mw <= 0 || K <= 0) return fail(TMAC_HIP_E_NOMATCH, "bad shape");
    if (s.bm <= 0 || s.bm % 32 || s.bm % bits || (s.bm / bits) % 8 || (Mw * bits) % s.bm)
        return fail(TMAC_HIP_E_NOMATCH, "M=%d*%d not tileable by bm=%d", Mw, bits, s.bm);
    if (s.kfactor <= 0 || (K / 4) % s.kfactor || K % 4) return fail(TMAC_HIP_E_NOMATCH, "K=%d not tileable by kfactor=%d", K, s.kfactor);
    if (s.ags % 32 || K % s.ags) return fail(TMAC_HIP_E_NOMATCH, "K=%d not divisible by act_group_size=%d", K, s.ags);
    if (s.m_groups < 0 && (s.gs <= 0 || K % s.gs)) return fail(TMAC_HIP_E_NOMATCH, "K=%d not divisible by group_size=%d", K, s.gs);
    if (!(s.m_groups >= 1 && s.ags == K) && (4 * s.kfactor) % s.ags)
        return fail(TMAC_HIP_E_NOMATCH, "act_group_size=%d must divide 4*kfactor=%d (qgemm.py:113-115)", s.ags, 4 * s.kfactor);
    if (s.m_groups < 0 && s.gs % (4 * s.kfactor)) return fail(TMAC_HIP_E_NOMATCH, "group_size %% (4*kfactor) != 0");
    // device layout: 8-table units for the fused kernel, 16-table segments for the two-kernel path
    s.ts = 8;
    s.lay = 0;
    const bool fused_ok = gemv_fused_supported(s), quad_ok = gemv_quad_supported(s);
    if (g_knobs.fa_mode) {
        // the reference has no fast aggregation on the int32 / unified-scale path (tbl.cc:534) and the halving tree
        // needs a power-of-two number of tables per act group
        if (s.m_groups >= 1) return fail(TMAC_HIP_E_NOMATCH, "fast aggregation is defined for per-group scales only");
        if (s.ags != 32 && s.ags != 64) return fail(TMAC_HIP_E_NOMATCH, "fast aggregation needs act_group_size 32 or 64");
        s.ts = 16;   // one act group's 16 (or 2 x 8) tables per lane: the tree stays inside a thread (k_gemv_lo)
        return TMAC_HIP_OK;
    }
    if ((g_knobs.variant == V_AUTO || g_knobs.variant == V_QUAD || g_knobs.variant == V_QUAD_MQSAD) && quad_ok) s.lay = 2;
    else if (g_knobs.variant == V_LO_MQSAD || g_knobs.variant == V_LO_SDWA || !fused_ok) s.ts = 16;
    return TMAC_HIP_OK;
}

size_t tmac_host::ref_weight_bytes(const Shape& s) { return (size_t)s.M() * (s.K / 4) / 2; }
size_t tmac_host::ref_scale_elems(const Shape& s) {
    return s.m_groups >= 1 ? (size_t)s.m_groups : (size_t)s.Mw * (s.K / s.gs) * (s.zero_point ? 2 : 1);
}
size_t tmac_host::dt_size(Dtype d) { return d == F16 ? 2 : 4; }

int32_t tmac_host::register_impl(tmac_hip_weights** out, const void* A_ref, const void* scales_ref, bool src_on_device,
                             int Mw, int K, int bits, const tmac_kcfg* cfg, tmac_dtype_t host_float,
                             tmac_dtype_t dev_float, void* stream) {
    if (!out || !A_ref || !scales_ref) return fail(TMAC_HIP_E_ARG, "null argument");
    int32_t rc = ensure_device();
    if (rc) return rc;
    Shape s;
    rc = make_shape(s, Mw, K, bits, cfg);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    auto* w = new tmac_hip_weights();
    w->s = s;
    w->sc_dtype = (Dtype)dev_float;
    w->ref_dtype = (Dtype)host_float;
    w->fa = g_knobs.fa_mode;
    w->lo_ok = (s.lay == 2) ? gemv_quad_supported(s) : (s.ts == 8) ? gemv_fused_supported(s) : gemv_lo_supported(s);
    const size_t ab = ref_weight_bytes(s), se = ref_scale_elems(s), sb = se * dt_size((Dtype)host_float);
    const bool keep_ref = !w->lo_ok || g_knobs.variant == V_REF_LAYOUT;
    void *dA = nullptr, *dS = nullptr;
    auto cleanup = [&](int32_t code) {  // error path: drop everything this call allocated
        if (!(src_on_device && !keep_ref)) { if (dA) (void)hipFree(dA); if (dS) (void)hipFree(dS); }
        tmac_hip_free_weights(w);
        return code;
    };
#define REG_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return cleanup(fail(TMAC_HIP_E_RUNTIME, "%s failed: %s", #expr, hipGetErrorString(e_))); \
    } while (0)
    if (src_on_device && !keep_ref) {
        dA = const_cast<void*>(A_ref);
        dS = const_cast<void*>(scales_ref);
    } else {
        REG_TRY(hipMalloc(&dA, ab));
        REG_TRY(hipMalloc(&dS, sb));
        const hipMemcpyKind kind = src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        REG_TRY(hipMemcpyAsync(dA, A_ref, ab, kind, st));
        REG_TRY(hipMemcpyAsync(dS, scales_ref, sb, kind, st));
    }
    if (w->lo_ok) {
        w->w_bytes = s.weight_u4() * 16;
        REG_TRY(hipMalloc(&w->W, w->w_bytes));
        hipError_t e = launch_retile_weights((const uint8_t*)dA, w->W, s, st);
        if (e != hipSuccess) return cleanup(fail(TMAC_HIP_E_RUNTIME, "retile_weights: %s", hipGetErrorString(e)));
        const size_t de = s.m_groups >= 1 ? (size_t)s.m_groups : s.scale_elems();
        w->sc_bytes = de * dt_size(w->sc_dtype);
        REG_TRY(hipMalloc(&w->SC, w->sc_bytes));
        e = launch_retile_scales(dS, (Dtype)host_float, w->SC, w->sc_dtype, s, st);
        if (e != hipSuccess) return cleanup(fail(TMAC_HIP_E_RUNTIME, "retile_scales: %s", hipGetErrorString(e)));
    } else {
        w->w_bytes = ab;
        w->sc_bytes = sb;
    }
    REG_TRY(hipStreamSynchronize(st));
    if (keep_ref) {
        w->A_ref = dA;
        w->S_ref = dS;
    } else if (!src_on_device) {
        (void)hipFree(dA);
        (void)hipFree(dS);
    }
    *out = w;
    return TMAC_HIP_OK;
#undef REG_TRY
}

extern "C" int32_t tmac_hip_register_weights(tmac_hip_weights** out, const void* A_ref, const void* scales_ref, int Mw,
                                             int K, int bits, const tmac_kcfg* cfg, tmac_dtype_t host_float,
                                             tmac_dtype_t dev_float, void* stream) {
    return register_impl(out, A_ref, scales_ref, false, Mw, K, bits, cfg, host_float, dev_float, stream);
}
extern "C" int32_t tmac_hip_register_weights_dev(tmac_hip_weights** out, const void* A_ref_dev, const void* scales_ref_dev,
                                                 int Mw, int K, int bits, const tmac_kcfg* cfg, tmac_dtype_t host_float,
                                                 tmac_dtype_t dev_float, void* stream) {
    return register_impl(out, A_ref_dev, scales_ref_dev, true, Mw, K, bits, cfg, host_float, dev_float, stream);
}

extern "C" int32_t tmac_hip_free_weights(tmac_hip_weights* w) {
    if (!w) return TMAC_HIP_OK;
    defer_forget_all();       // recordings cached by the deferred-launch queue may name this matrix
    if (w->W) (void)hipFree(w->W);
    if (w->SC) (void)hipFree(w->SC);
    if (w->A_ref) (void)hipFree(w->A_ref);
    if (w->S_ref) (void)hipFree(w->S_ref);
    delete w;
    return TMAC_HIP_OK;
}

extern "C" size_t tmac_hip_weights_bytes(const tmac_hip_weights* w) {
    if (!w) return 0;
    // algorithmic bytes: Mw*K*bits/8 of weight planes + the scale(+zero) values
    const Shape& s = w->s;
    const size_t se = s.m_groups >= 1 ? (size_t)s.m_groups : (size_t)s.Mw * (s.K / s.gs) * (s.zero_point ? 2 : 1);
    return (size_t)s.Mw * s.K * s.bits / 8 + se * dt_size(w->sc_dtype);
}
