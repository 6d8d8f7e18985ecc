// tmac_dispatch.cpp — qgemm_lut dispatch: which kernel serves a call (decode GEMV variants, the N > 1 GEMMs, the fused
// LUT-build + GEMV entry point and its prefill route) and the parity taps around them.
#include "tmac_host.h"

using namespace tmac_host;

// GEMM or row loop for N activation rows on matrices with total_Mw output rows?  An explicitly set threshold
// (tmac_hip_set_gemm_min_n) is taken literally.  Otherwise: the measured crossover per launch where k_gemm_planes covers the
// configuration (planes_pays), and k_gemm_onehot only from 32 rows on with a grid that fills the chip (onehot_pays: with fewer
// than 128 workgroups of 128 bit-plane rows the row loop is faster up to 64 rows: 4096 x 11008 at N = 32: 88 us against 152 us).
static bool planes_covers(const Shape& s) { return g_knobs.gemm_kernel != 1 && s.lay == 2 && s.ts == 8 && gemm_planes_supported(s); }
static bool onehot_pays(const Shape& s, long total_Mw, int N) {
    if (g_knobs.gemm_min_n <= 0) return false;
    if (g_knobs.gemm_min_n != 32) return N >= g_knobs.gemm_min_n;
    return N >= 32 && (N >= 64 || (total_Mw * s.bits + 127) / 128 >= 128);
}
static bool planes_pays(const Shape& s, long total_Mw, int N) {
    if (g_knobs.gemm_min_n <= 0) return false;
    if (g_knobs.gemm_min_n != 32) return N >= g_knobs.gemm_min_n;
    // From how many activation rows on k_gemm_planes beats the GEMV kernel looped over the rows: measured on MI355X, llama-2-7B shapes,
    // 1- to 4-bit weights (tools/bench_small_n.py, profiles/r03_small_n.txt).  The row loop costs ~3 us + N x (0.7 us + 0.155 us per MB
    // of weights + 0.2 us per 1000 of K beyond 4096); the GEMM (6 + 14 K / 4096) us per wave of 64 x 64 tiles (x 1.18 for 3- / 4-bit
    // operand rows) whatever N <= 64 is.  Crossovers (W2): o 14, q/k/v 7, gate/up 7, down 9 rows; the fixed 12 rows of round 2 -- which
    // the fused entry point applied on top of a fixed 32 -- left up to 2 x on the table for 7-31 rows.  5 % margin for the row loop.
    const double mb = (double)total_Mw * s.K * s.bits / 8e6, tiles = (double)((total_Mw + 63) / 64) * ((N + 63) / 64);
    const double waves = tiles <= 256.0 ? 1.0 : s.bits == 4 ? (double)(((long)tiles + 255) / 256) : (s.bits == 3 ? 0.45 : 0.25) + tiles / 256.0;
    const double tp = (6.0 + 14.0 * s.K / 4096.0) * waves * (s.bits >= 3 ? 1.18 : 1.0);
    const double c1 = 0.7 + 0.155 * mb + 0.2 * (s.K > 4096 ? (s.K - 4096) / 1000.0 : 0.0);
    int nmin = (int)((1.05 * tp - 3.0) / c1 + 0.999);
    nmin = nmin < 4 ? 4 : nmin > 16 ? 16 : nmin;
    return N >= nmin;
}
// the fused entry point builds whatever LUT form the chosen kernel wants, so one question decides
static bool gemm_pays(const Shape& s, long total_Mw, int N) { return planes_covers(s) ? planes_pays(s, total_Mw, N) : onehot_pays(s, total_Mw, N); }
// one-hot MFMA GEMM over 1..4 matrices that share K, the quantisation config (checked by the callers) and the LUT in ws
static int32_t gemm_multi(const tmac_hip_weights* const* wl, int nmat, const tmac_hip_workspace* ws, void* const* C_list,
                          tmac_dtype_t out_dtype, int N, int32_t* dump, hipStream_t st) {
    GemmArgs ga;
    memset(&ga, 0, sizeof(ga));
    const tmac_hip_weights* w0 = wl[0];
    ga.s = w0->s; ga.nmat = nmat;
    for (int i = 0; i < nmat; ++i) { ga.m[i].W = wl[i]->W; ga.m[i].SC = wl[i]->SC; ga.m[i].C = C_list[i]; ga.m[i].Mw = wl[i]->s.Mw; }
    ga.sc_f16 = w0->sc_dtype == F16; ga.out_f16 = out_dtype == TMAC_F16;
    ga.qlut_lds = ws->qlut_lds; ga.tstride = (((w0->s.K / 32) + 15) & ~15) + 1; ga.lut_scales = ws->lut_scales; ga.lut_biases = ws->lut_biases;
    ga.dump = dump; ga.N = N;
    hipError_t e = launch_gemm_onehot(ga, st);
    if (e != hipSuccess) return fail(TMAC_HIP_E_RUNTIME, "one-hot gemm launch: %s", hipGetErrorString(e));
    return TMAC_HIP_OK;
}


static bool planes_ok(const tmac_hip_weights* w) {
    return g_knobs.gemm_kernel != 1 && w->s.lay == 2 && w->lo_ok && w->s.ts == 8 && !w->fa && gemm_planes_supported(w->s) &&
           w->w_bytes < ((size_t)1 << 31);
}
static bool planes_image_fits(const tmac_hip_workspace* ws, int K) { return ws->gimg && (size_t)2 * K * ws->gNpad < ((size_t)1 << 31); }

// k_gemm_planes over up to 4 matrices that share K and the quantisation config; the workspace holds the LUT image
static int32_t planes_multi(const tmac_hip_weights* const* wl, int nmat, const tmac_hip_workspace* ws, void* const* C_list,
                            tmac_dtype_t out_dtype, int N, int32_t* comb_dump, hipStream_t st) {
    Gemm2Args ga;
    memset(&ga, 0, sizeof(ga));
    const tmac_hip_weights* w0 = wl[0];
    ga.s = w0->s; ga.nmat = nmat;
    for (int i = 0; i < nmat; ++i) { ga.m[i].W = wl[i]->W; ga.m[i].SC = wl[i]->SC; ga.m[i].C = C_list[i]; ga.m[i].Mw = wl[i]->s.Mw; }
    ga.sc_f16 = w0->sc_dtype == F16; ga.out_f16 = out_dtype == TMAC_F16;
    ga.bimg = (const uint4*)ws->gimg; ga.colv = ws->gcol; ga.Npad = ws->gNpad; ga.N = N; ga.dump = comb_dump;
    ga.stamps = g_knobs.gemm_stamps;
    ga.form = g_knobs.gemm_kernel == 2 ? 1 : g_knobs.gemm_kernel == 3 ? 2 : 0;
    hipError_t e = launch_gemm_planes(ga, st);
    if (e == hipErrorInvalidValue) return fail(TMAC_HIP_E_NOMATCH, "plane-combined gemm: configuration or sizes not covered (LUT image and matrices must stay below 2 GB)");
    if (e != hipSuccess) return fail(TMAC_HIP_E_RUNTIME, "plane-combined gemm launch: %s", hipGetErrorString(e));
    return TMAC_HIP_OK;
}

int32_t tmac_host::qgemm_impl(const tmac_hip_weights* w, const tmac_hip_workspace* ws, void* C_dev, tmac_dtype_t out_dtype,
                          int N, int32_t* dump, hipStream_t st) {
    bind_thread_device();
    if (!w || !ws || !C_dev) return fail(TMAC_HIP_E_ARG, "null argument");
    if (ws->K != w->s.K || ws->ags != w->s.ags)
        return fail(TMAC_HIP_E_ARG, "workspace LUT (K=%d, ags=%d) does not match the weights (K=%d, ags=%d)", ws->K, ws->ags, w->s.K, w->s.ags);
    if (N <= 0 || N > ws->N) return fail(TMAC_HIP_E_ARG, "N=%d but the workspace LUT holds %d rows", N, ws->N);
    Variant v = (Variant)g_knobs.variant;
    if (v != V_REF_LAYOUT) {   // the weights' device layout decides which tiled kernel can run
        if (!w->lo_ok) v = V_REF_LAYOUT;
        else if (w->s.ts == 8) v = V_FUSED;
        else if (v != V_LO_SDWA) v = V_LO_MQSAD;
    }
    if (w->fa && v != V_REF_LAYOUT && v != V_LO_MQSAD && v != V_LO_SDWA)
        return fail(TMAC_HIP_E_NOMATCH, "fast-aggregation weights run on the two-kernel path only");
    // split entry points: tmac_hip_preprocessor_dev builds k_gemm_planes' LUT image from PLANES_MIN_N rows on (it does not know the
    // matrix); without the image only k_gemm_onehot's own, later crossover counts -- below it the row loop is the faster kernel
    if (v == V_FUSED && !dump && ws->gimg_valid && ws->gimg_kind == (w->s.m_groups >= 1 ? 1 : 2) && planes_ok(w) && planes_image_fits(ws, w->s.K) && planes_pays(w->s, w->s.Mw, N)) {
        void* cl[1] = {C_dev};
        return planes_multi(&w, 1, ws, cl, out_dtype, N, nullptr, st);
    }
    if (v == V_FUSED && onehot_pays(w->s, w->s.Mw, N) && gemm_onehot_supported(w->s)) {
        void* cl[1] = {C_dev};
        return gemm_multi(&w, 1, ws, cl, out_dtype, N, dump, st);
    }
    if (v == V_FUSED) {
        FusedArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.nmat = 1; fa.s = w->s;
        fa.m[0].W = (const uint4*)w->W; fa.m[0].SC = w->SC; fa.m[0].C = C_dev; fa.m[0].Mw = w->s.Mw; fa.m[0].nb_end = w->s.nb();
        fa.qlut_lds = ws->qlut_lds; fa.lut_scales = ws->lut_scales; fa.lut_biases = ws->lut_biases;
        fa.sc_f16 = w->sc_dtype == F16; fa.out_f16 = out_dtype == TMAC_F16; fa.dump = dump;
        fa.acc_mfma = (w->s.lay == 2) ? (g_knobs.variant != V_QUAD_MQSAD) : (g_knobs.variant == V_FUSED_MFMA);
        if (w->s.lay == 2) fa.m[0].nb_end = w->s.nquads();
        hipError_t e = (w->s.lay == 2) ? launch_gemv_quad(fa, N, false, g_knobs.force_ft, g_knobs.force_wpq, st) : launch_gemv_fused(fa, N, false, st);
        if (e == hipErrorInvalidValue) return fail(TMAC_HIP_E_NOMATCH, "no fused GEMV kernel for this configuration");
        if (e != hipSuccess) return fail(TMAC_HIP_E_RUNTIME, "fused gemv launch: %s", hipGetErrorString(e));
        return TMAC_HIP_OK;
    }
    GemvArgs a;
    a.s = w->s; a.N = N; a.qlut_dev = ws->qlut_dev; a.qlut_ref = ws->qlut_ref;
    a.lut_scales = ws->lut_scales; a.lut_biases = ws->lut_biases; a.C = C_dev; a.out_dtype = (Dtype)out_dtype;
    a.ps_dump = dump;
    a.fa_mode = w->fa;
    if (v == V_REF_LAYOUT) {
        if (!w->A_ref) return fail(TMAC_HIP_E_NOMATCH, "reference-layout blobs were not kept for these weights (register them with variant 3 selected)");
        a.W = w->A_ref; a.SC = w->S_ref; a.sc_dtype = w->ref_dtype;
    } else {
        a.W = w->W; a.SC = w->SC; a.sc_dtype = w->sc_dtype;
    }
    hipError_t e = launch_gemv(a, v, st);
    if (e == hipErrorInvalidValue) return fail(TMAC_HIP_E_NOMATCH, "no GEMV kernel for this configuration");
    if (e != hipSuccess) return fail(TMAC_HIP_E_RUNTIME, "gemv launch: %s", hipGetErrorString(e));
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_qgemm_dev(const tmac_hip_weights* w, const tmac_hip_workspace* ws, void* C_dev,
                                      tmac_dtype_t out_dtype, int N, void* stream) {
    return qgemm_impl(w, ws, C_dev, out_dtype, N, nullptr, (hipStream_t)stream);
}

extern "C" int32_t tmac_hip_qgemm_partial_sums(const tmac_hip_weights* w, const tmac_hip_workspace* ws_c, int32_t* PS_host,
                                               int N, void* stream) {
    if (!w || !ws_c || !PS_host) return fail(TMAC_HIP_E_ARG, "null argument");
    auto* ws = const_cast<tmac_hip_workspace*>(ws_c);
    hipStream_t st = (hipStream_t)stream;
    const size_t G = (w->s.m_groups >= 1 && w->s.ags == w->s.K) ? 1 : (size_t)w->s.ngroups();
    const size_t elems = (size_t)N * w->s.M() * G;
    if (ws->dump_elems < elems) {
        if (ws->dump) (void)hipFree(ws->dump);
        HIP_TRY(hipMalloc((void**)&ws->dump, elems * sizeof(int32_t)));
        ws->dump_elems = elems;
    }
    HIP_TRY(hipMemsetAsync(ws->dump, 0x7f, elems * sizeof(int32_t), st));
    DevBuf Ctmp;
    HIP_TRY(Ctmp.alloc(sizeof(float) * (size_t)N * w->s.Mw));
    int32_t rc = qgemm_impl(w, ws, Ctmp.p, TMAC_F32, N, ws->dump, st);
    if (rc == TMAC_HIP_OK) {
        hipError_t e = hipMemcpyAsync(PS_host, ws->dump, elems * sizeof(int32_t), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = fail(TMAC_HIP_E_RUNTIME, "partial-sum readback: %s", hipGetErrorString(e));
    } else {
        (void)hipStreamSynchronize(st);      // nothing of this call may still use the scratch that is freed on return
    }
    return rc;
}

extern "C" int32_t tmac_hip_debug_gemm_stamps(unsigned long long* dev_buffer) {
    g_knobs.gemm_stamps = dev_buffer;
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_debug_gemm_kernel(int which) {
    if (which < 0 || which > 3)
        return fail(TMAC_HIP_E_ARG, "gemm kernel selector must be 0 (auto), 1 (k_gemm_onehot), 2 / 3 (k_gemm_planes with eight- / four-wave workgroups)");
    g_knobs.gemm_kernel = which;
    return TMAC_HIP_OK;
}

// Parity tap of k_gemm_planes: the combined integer sums comb[n][o][kk] = sum_p 2^p PS_p it feeds into the fp32 chain.
extern "C" int32_t tmac_hip_debug_gemm_comb_sums(const tmac_hip_weights* w, const tmac_hip_workspace* ws_c, int32_t* comb_host,
                                                 int N, void* stream) {
    if (!w || !ws_c || !comb_host) return fail(TMAC_HIP_E_ARG, "null argument");
    auto* ws = const_cast<tmac_hip_workspace*>(ws_c);
    hipStream_t st = (hipStream_t)stream;
    if (!ws->gimg_valid || ws->gimg_kind != (w->s.m_groups >= 1 ? 1 : 2) || ws->K != w->s.K || N <= 0 || N > ws->N) return fail(TMAC_HIP_E_ARG, "the workspace holds no LUT image for K=%d, N=%d", w->s.K, N);
    if (!planes_ok(w)) return fail(TMAC_HIP_E_NOMATCH, "k_gemm_planes does not cover this configuration");
    const size_t elems = (size_t)N * w->s.Mw * (w->s.m_groups >= 1 ? 1 : w->s.K / 64);
    if (ws->dump_elems < elems) {
        if (ws->dump) (void)hipFree(ws->dump);
        ws->dump = nullptr; ws->dump_elems = 0;
        HIP_TRY(hipMalloc((void**)&ws->dump, elems * sizeof(int32_t)));
        ws->dump_elems = elems;
    }
    HIP_TRY(hipMemsetAsync(ws->dump, 0x7f, elems * sizeof(int32_t), st));
    DevBuf Ctmp;
    HIP_TRY(Ctmp.alloc(sizeof(float) * (size_t)N * w->s.Mw));
    void* cl[1] = {Ctmp.p};
    int32_t rc = planes_multi(&w, 1, ws, cl, TMAC_F32, N, ws->dump, st);
    if (rc == TMAC_HIP_OK) {
        hipError_t e = hipMemcpyAsync(comb_host, ws->dump, elems * sizeof(int32_t), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = fail(TMAC_HIP_E_RUNTIME, "comb-sum readback: %s", hipGetErrorString(e));
    } else {
        (void)hipStreamSynchronize(st);
    }
    return rc;
}

// The LUT image of the workspace in plain layouts: half tables int8 [N][K/4][8], then lut_scales, lut_biases and the
// per-act-group entry sums, fp32 [N][K/64] each.
extern "C" int32_t tmac_hip_debug_gemm_image_read(const tmac_hip_workspace* ws, int8_t* half_tables_host, float* lut_scales_host,
                                                  float* lut_biases_host, float* entry_sums_host, int N, void* stream) {
    if (!ws || !half_tables_host || !lut_scales_host || !lut_biases_host || !entry_sums_host) return fail(TMAC_HIP_E_ARG, "null argument");
    if (!ws->gimg_valid || N <= 0 || N > ws->N) return fail(TMAC_HIP_E_ARG, "the workspace holds no LUT image for N=%d", N);
    hipStream_t st = (hipStream_t)stream;
    const int K = ws->K, Np = ws->gNpad;
    const bool rowwise = ws->gimg_kind == 1;           // one act group per row: k_preprocess_pairs_row's layout
    const int G = rowwise ? 1 : K / 64;
    std::vector<uint8_t> img((size_t)2 * K * Np);
    std::vector<float> col((size_t)(rowwise ? 3 : 4) * G * Np);
    HIP_TRY(hipMemcpyAsync(img.data(), ws->gimg, img.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(col.data(), ws->gcol, col.size() * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int n = 0; n < N; ++n) {
        for (int t2 = 0; t2 < K / 8; ++t2) {           // pair t2 = tables 2 t2, 2 t2 + 1: unit t2 / 4, pair t2 % 4; act group t2 / 8, part t2 % 8
            const size_t u4 = rowwise ? ((size_t)(t2 >> 2) * 4 + (t2 & 3)) * Np + n
                                      : (((size_t)(t2 >> 3) * (Np >> 6) + (n >> 6)) * 8 + (t2 & 7)) * 64 + (n & 63);
            memcpy(half_tables_host + ((size_t)n * (K / 4) + 2 * t2) * 8, img.data() + u4 * 16, 16);
        }
        for (int kk = 0; kk < G; ++kk) {
            if (rowwise) {
                lut_scales_host[n] = col[n];
                lut_biases_host[n] = col[(size_t)Np + n];
                int32_t esi;                               // (int32 bits, see k_preprocess_pairs_row)
                memcpy(&esi, &col[(size_t)2 * Np + n], sizeof(esi));
                entry_sums_host[n] = (float)esi;
            } else {
                const float* c4 = &col[((size_t)kk * Np + n) * 4];        // lut_scales / 2 | lut_biases / 2 | entry sum | lut_biases
                lut_scales_host[(size_t)n * G + kk] = c4[0] * 2.0f;
                lut_biases_host[(size_t)n * G + kk] = c4[3];
                entry_sums_host[(size_t)n * G + kk] = c4[2];
            }
        }
    }
    return TMAC_HIP_OK;
}


// Prefill through the fused entry point: one LUT build (k_preprocess) into a workspace owned by the library, one
// one-hot MFMA GEMM per matrix.  The workspace is per stream (launches on one stream are ordered; two streams must not
// share LUT buffers) and grows on demand; tmac_hip_cache_clear() releases them.
static std::map<std::pair<int, hipStream_t>, tmac_hip_workspace*> g_fused_ws;   // per (device, stream): the null stream exists on every device

static int32_t fused_prefill(const tmac_hip_weights* const* wl, int nmat, const void* B_dev, tmac_dtype_t act_dtype,
                             void* const* C_list, tmac_dtype_t out_dtype, int N, hipStream_t st) {
    const Shape& s0 = wl[0]->s;
    tmac_hip_workspace* ws = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        tmac_hip_workspace*& slot = g_fused_ws[std::make_pair(g_device, st)];
        int needK = s0.K, needN = N;
        if (slot && (slot->maxK < s0.K || slot->maxN < N)) {
            // grow to the maximum seen in BOTH dimensions (mixed shapes -- K = 4096 / 11008, growing N -- would otherwise
            // free and reallocate on every other call); the old buffers may still be read by launches in flight
            needK = slot->maxK > s0.K ? slot->maxK : s0.K;
            needN = slot->maxN > N ? slot->maxN : N;
            hipError_t e = hipStreamSynchronize(st);
            if (e != hipSuccess) return fail(TMAC_HIP_E_RUNTIME, "stream sync: %s", hipGetErrorString(e));
            tmac_hip_workspace_free(slot);
            slot = nullptr;
        }
        if (!slot) {
            int32_t rc = tmac_hip_workspace_create(&slot, needK, needN);
            if (rc) { slot = nullptr; return rc; }
        }
        ws = slot;
    }
    int32_t rc;
    bool planes = g_knobs.variant != V_REF_LAYOUT && planes_image_fits(ws, s0.K);
    for (int i = 0; i < nmat && planes; ++i) {
        const Shape &x = wl[i]->s, &y = s0;
        planes = planes_ok(wl[i]) && x.bits == y.bits && x.gs == y.gs && x.zero_point == y.zero_point && x.ags == y.ags &&
                 x.m_groups == y.m_groups && wl[i]->sc_dtype == wl[0]->sc_dtype;
    }
    if (planes && s0.m_groups >= 1 && s0.K > 12288) planes = false;   // (the row-wise LUT build's limit)
    if (planes) {
        // the plane-combined GEMM reads its own LUT image only: one build, one launch for all matrices
        rc = check_lut_shape(ws, s0.K, N, s0.ags);
        if (rc) return rc;
        ws->K = 0; ws->N = 0; ws->gimg_valid = false;      // the other layouts of this workspace are not built
        hipError_t e = s0.m_groups >= 1
            ? launch_preprocess_pairs_row(B_dev, act_dtype == TMAC_F16, ws->qlut_lds, ws->lut_scales, ws->lut_biases, s0.K, N, nullptr, nullptr, 0,
                                          ws->gimg, ws->gcol, ws->gNpad, st)
            : launch_lut_image(B_dev, act_dtype == TMAC_F16, ws->gimg, ws->gcol, s0.K, N, ws->gNpad, st);
        if (e != hipSuccess) return fail(TMAC_HIP_E_RUNTIME, "LUT image launch: %s", hipGetErrorString(e));
        return planes_multi(wl, nmat, ws, C_list, out_dtype, N, nullptr, st);
    }
    bool onehot_all = true;       // (1- / 3-bit weights reach this point only when k_gemm_planes cannot take them: they go to the row loop)
    for (int i = 0; i < nmat; ++i) onehot_all = onehot_all && gemm_onehot_supported(wl[i]->s);
    if (onehot_all && (s0.ags == 64 || (s0.ags == s0.K && s0.K <= 12288)) && g_knobs.variant != V_REF_LAYOUT) {
        // only the one-hot GEMM reads this workspace: build the half-table image alone, two tables per lane
        rc = check_lut_shape(ws, s0.K, N, s0.ags);
        if (rc) return rc;
        ws->K = s0.K; ws->N = N; ws->ags = s0.ags; ws->qdev_u4_per_row = qdev_u4_for_K(s0.K);
        hipError_t e = s0.ags == 64
            ? launch_preprocess_pairs(B_dev, act_dtype == TMAC_F16, ws->qlut_lds, ws->lut_scales, ws->lut_biases, s0.K, N, nullptr, nullptr, 0, st)
            : launch_preprocess_pairs_row(B_dev, act_dtype == TMAC_F16, ws->qlut_lds, ws->lut_scales, ws->lut_biases, s0.K, N, nullptr, nullptr, 0,
                                          nullptr, nullptr, 0, st);
        if (e != hipSuccess) return fail(TMAC_HIP_E_RUNTIME, "preprocess launch: %s", hipGetErrorString(e));
    } else {
        rc = tmac_hip_preprocessor_dev(ws, B_dev, act_dtype, s0.K, N, s0.ags, st);
    }
    if (rc) return rc;
    bool same = onehot_all && g_knobs.variant != V_REF_LAYOUT;      // (the caller has established that the GEMM pays for these matrices)
    for (int i = 0; i < nmat && same; ++i) {
        const Shape &x = wl[i]->s, &y = s0;
        same = x.lay == 2 && wl[i]->lo_ok && x.ts == 8 && x.bits == y.bits && x.gs == y.gs && x.zero_point == y.zero_point &&
               x.m_groups == y.m_groups && wl[i]->sc_dtype == wl[0]->sc_dtype && !wl[i]->fa;
    }
    if (same) return gemm_multi(wl, nmat, ws, C_list, out_dtype, N, nullptr, st);   // q/k/v or gate/up: one launch fills the chip
    for (int i = 0; i < nmat && rc == TMAC_HIP_OK; ++i) rc = tmac_hip_qgemm_dev(wl[i], ws, C_list[i], out_dtype, N, st);
    return rc;
}
int32_t tmac_host::fused_impl(const tmac_hip_weights* const* wl, int nmat, const void* B_dev, tmac_dtype_t act_dtype,
                          void* const* C_list, tmac_dtype_t out_dtype, int N, int32_t* dump, float* lut_tap, hipStream_t st) {
    bind_thread_device();
    if (!wl || !C_list || !B_dev || nmat < 1 || nmat > 4 || N < 1) {
        if (chain_recording()) chain_clear_xform();      // a rejected call must not leave its transform pending for the next recorded call
        return fail(TMAC_HIP_E_ARG, "bad fused arguments (1..4 matrices)");
    }
    if (chain_recording() && !dump && !lut_tap) return chain_record(wl, nmat, B_dev, act_dtype, C_list, out_dtype, N);
    if (!dump && !lut_tap) {           // deferred launches: queued until tmac_hip_flush (or a call that depends on a queued one)
        int32_t drc;
        if (defer_if_on(wl, nmat, B_dev, act_dtype, C_list, out_dtype, N, st, &drc)) return drc;
    }
    if (g_knobs.gemm_min_n > 0 && N >= 2 && !dump && !lut_tap) {       // (gemm_pays applies the threshold: a set one, or the measured crossover)
        bool ok = true;
        long rows = 0;
        for (int i = 0; i < nmat; ++i) {
            ok = ok && wl[i] && C_list[i] && (gemm_onehot_supported(wl[i]->s) || planes_ok(wl[i])) && wl[i]->s.K == wl[0]->s.K &&
                 wl[i]->s.ags == wl[0]->s.ags;
            if (ok) rows += wl[i]->s.Mw;
        }
        if (ok && gemm_pays(wl[0]->s, rows, N)) return fused_prefill(wl, nmat, B_dev, act_dtype, C_list, out_dtype, N, st);
    }
    FusedArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.nmat = nmat;
    int nb = 0;
    for (int i = 0; i < nmat; ++i) {
        const tmac_hip_weights* w = wl[i];
        if (!w || !C_list[i]) return fail(TMAC_HIP_E_ARG, "null matrix or output");
        if (w->s.ts != 8 || !w->lo_ok || w->s.lay != wl[0]->s.lay) return fail(TMAC_HIP_E_NOMATCH, "matrix %d is not registered in the fused layout", i);
        const Shape &a = w->s, &b = wl[0]->s;
        if (a.K != b.K || a.bits != b.bits || a.gs != b.gs || a.ags != b.ags || a.zero_point != b.zero_point ||
            a.m_groups != b.m_groups || w->sc_dtype != wl[0]->sc_dtype)
            return fail(TMAC_HIP_E_ARG, "matrices fused in one launch must share K, bits and quantisation config");
        nb += (a.lay == 2) ? a.nquads() : a.nb();
        fa.m[i].W = (const uint4*)w->W; fa.m[i].SC = w->SC; fa.m[i].C = C_list[i]; fa.m[i].Mw = a.Mw; fa.m[i].nb_end = nb;
    }
    fa.s = wl[0]->s;
    fa.B = B_dev; fa.act_f16 = act_dtype == TMAC_F16;
    fa.sc_f16 = wl[0]->sc_dtype == F16; fa.out_f16 = out_dtype == TMAC_F16; fa.dump = dump;
    fa.stamps = g_knobs.stamps;
    if (g_knobs.stamps && !fa.dump) fa.dump = g_knobs.stamp_dump;   // the stamps live in the tap (DUMP) instantiation of the kernel
    fa.lut_tap = lut_tap;
    fa.acc_mfma = (fa.s.lay == 2) ? (g_knobs.variant != V_QUAD_MQSAD) : (g_knobs.variant == V_FUSED_MFMA);
    int ft = g_knobs.force_ft, wpq = g_knobs.force_wpq;
    if (fa.s.lay == 2 && !ft && !wpq && !fa.dump && N == 1) tuned_config(fa, nb, ft, wpq);
    hipError_t e = (fa.s.lay == 2) ? launch_gemv_quad(fa, N, true, ft, wpq, st) : launch_gemv_fused(fa, N, true, st);
    if (e == hipErrorInvalidValue) return fail(TMAC_HIP_E_NOMATCH, "no fused GEMV kernel for this configuration");
    if (e != hipSuccess) return fail(TMAC_HIP_E_RUNTIME, "fused gemv launch: %s", hipGetErrorString(e));
    return TMAC_HIP_OK;
}
extern "C" int32_t tmac_hip_qgemm_fused_dev(const tmac_hip_weights* const* weights, int nmat, const void* B_dev,
                                            tmac_dtype_t act_dtype, void* const* C_dev, tmac_dtype_t out_dtype, int N,
                                            void* stream) {
    return fused_impl(weights, nmat, B_dev, act_dtype, C_dev, out_dtype, N, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int32_t tmac_hip_qgemm_fused_partial_sums(const tmac_hip_weights* w, const void* B_dev, tmac_dtype_t act_dtype,
                                                     int32_t* PS_host, float* C_host, float* lut_host, int N, void* stream) {
    if (!w || !PS_host) return fail(TMAC_HIP_E_ARG, "null argument");
    hipStream_t st = (hipStream_t)stream;
    const size_t G = (w->s.m_groups >= 1 && w->s.ags == w->s.K) ? 1 : (size_t)w->s.ngroups();
    const size_t elems = (size_t)N * w->s.M() * G;
    DevBuf dump, Ctmp, ltap;
    const size_t lt = (size_t)N * 2 * w->s.ngroups();
    HIP_TRY(ltap.alloc(lt * sizeof(float)));
    HIP_TRY(dump.alloc(elems * sizeof(int32_t)));
    HIP_TRY(Ctmp.alloc(sizeof(float) * (size_t)N * w->s.Mw));
    HIP_TRY(hipMemsetAsync(dump.p, 0x7f, elems * sizeof(int32_t), st));
    void* cl[1] = {Ctmp.p};
    int32_t rc = fused_impl(&w, 1, B_dev, act_dtype, cl, TMAC_F32, N, dump.as<int32_t>(), ltap.as<float>(), st);
    if (rc == TMAC_HIP_OK) {
        hipError_t e = hipMemcpyAsync(PS_host, dump.p, elems * sizeof(int32_t), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && C_host) e = hipMemcpyAsync(C_host, Ctmp.p, sizeof(float) * (size_t)N * w->s.Mw, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && lut_host) e = hipMemcpyAsync(lut_host, ltap.p, lt * sizeof(float), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = fail(TMAC_HIP_E_RUNTIME, "fused tap readback: %s", hipGetErrorString(e));
    } else {
        (void)hipStreamSynchronize(st);
    }
    return rc;
}

// the per-stream workspaces of the fused entry point's prefill route (tmac_hip_cache_clear; caller holds g_mu)
void tmac_host::release_fused_workspaces() {
    for (auto& kv : g_fused_ws) {
        (void)hipStreamSynchronize(kv.first.second);      // launches in flight may still read the LUT workspace
        tmac_hip_workspace_free(kv.second);
    }
    g_fused_ws.clear();
}
