// tmac_workspace.cpp — the LUT workspace (TMACGeMMWrapper::set_workspace, tmac_gemm_wrapper.h:257-270, on the device) and
// the preprocessor entry point that fills it (lut_ctor.cc:38-266 + generated glue).
#include "tmac_host.h"

using namespace tmac_host;

extern "C" int32_t tmac_hip_workspace_create(tmac_hip_workspace** out, int maxK, int maxN) {
    if (!out || maxK <= 0 || maxN <= 0 || maxK % 64) return fail(TMAC_HIP_E_ARG, "bad workspace size (maxK must be a multiple of 64)");
    int32_t rc = ensure_device();
    if (rc) return rc;
    auto* ws = new tmac_hip_workspace();
    ws->maxK = maxK; ws->maxN = maxN;
    const size_t nq = (size_t)maxN * (maxK / 4) * 16, nd = (size_t)maxN * qdev_u4_for_K(maxK) * 16, nl = (size_t)maxN * qlut_lds_u4(maxK) * 16;
    const size_t ns = sizeof(float) * (size_t)maxN * (maxK / 32);
    hipError_t e = hipMalloc((void**)&ws->qlut_ref, nq);
    if (e == hipSuccess) e = hipMalloc(&ws->qlut_dev, nd);
    if (e == hipSuccess) e = hipMemset(ws->qlut_dev, 0x80, nd);
    if (e == hipSuccess) e = hipMalloc(&ws->qlut_lds, nl);
    if (e == hipSuccess) e = hipMemset(ws->qlut_lds, 0x80, nl);
    if (e == hipSuccess) e = hipMalloc((void**)&ws->lut_scales, ns);
    if (e == hipSuccess) e = hipMalloc((void**)&ws->lut_biases, ns);
    if (maxN > 1) {
        ws->gNpad = (maxN + 63) & ~63;
        if (e == hipSuccess) e = hipMalloc(&ws->gimg, (size_t)2 * maxK * ws->gNpad);
        if (e == hipSuccess) e = hipMalloc((void**)&ws->gcol, sizeof(float) * 4 * (size_t)(maxK / 64) * ws->gNpad);
    }
    // The fills above are null-stream work and the workspace's users launch on streams of their own (the host-pointer layer
    // and the ggml glue on NON-BLOCKING streams, which the null stream does not order): a fill that lands after the first
    // LUT build leaves all-zero half tables behind (round 2: qgemm_lut_int8 returned the bias terms only).  Complete them here.
    if (e == hipSuccess && g_knobs.ws_fill_sync) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {   // nothing of a half-built workspace is left behind
        tmac_hip_workspace_free(ws);
        return fail(TMAC_HIP_E_RUNTIME, "workspace allocation (K=%d, N=%d): %s", maxK, maxN, hipGetErrorString(e));
    }
    *out = ws;
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_workspace_free(tmac_hip_workspace* ws) {
    if (!ws) return TMAC_HIP_OK;
    if (ws->qlut_ref) (void)hipFree(ws->qlut_ref);
    if (ws->qlut_dev) (void)hipFree(ws->qlut_dev);
    if (ws->qlut_lds) (void)hipFree(ws->qlut_lds);
    if (ws->lut_scales) (void)hipFree(ws->lut_scales);
    if (ws->lut_biases) (void)hipFree(ws->lut_biases);
    if (ws->gimg) (void)hipFree(ws->gimg);
    if (ws->gcol) (void)hipFree(ws->gcol);
    if (ws->dump) (void)hipFree(ws->dump);
    delete ws;
    return TMAC_HIP_OK;
}

int32_t tmac_host::check_lut_shape(tmac_hip_workspace* ws, int K, int N, int ags) {
    if (!ws) return fail(TMAC_HIP_E_ARG, "null workspace");
    if (K <= 0 || K > ws->maxK || N <= 0 || N > ws->maxN) return fail(TMAC_HIP_E_ARG, "K=%d N=%d exceed the workspace (%d, %d)", K, N, ws->maxK, ws->maxN);
    if (ags <= 0 || ags % 32 || K % ags) return fail(TMAC_HIP_E_NOMATCH, "act_group_size=%d must be a multiple of 32 dividing K=%d (qgemm.py:402-404)", ags, K);
    if (K % 64) return fail(TMAC_HIP_E_NOMATCH, "K=%d must be a multiple of 64", K);
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_preprocessor_dev(tmac_hip_workspace* ws, const void* B_dev, tmac_dtype_t act_dtype, int K,
                                             int N, int act_group_size, void* stream) {
    bind_thread_device();
    int32_t rc = check_lut_shape(ws, K, N, act_group_size);
    if (rc) return rc;
    if (!B_dev) return fail(TMAC_HIP_E_ARG, "null activations");
    ws->K = K; ws->N = N; ws->ags = act_group_size; ws->qdev_u4_per_row = qdev_u4_for_K(K);
    const int gmin = g_knobs.gemm_min_n <= 0 ? 0x7fffffff : (g_knobs.gemm_min_n != 32 ? (g_knobs.gemm_min_n > 2 ? g_knobs.gemm_min_n : 2) : PLANES_MIN_N);
    // one act group per row: the row-wise pair build also writes the LUT image of the plane-combined GEMM when that may be chosen
    const bool row_img = act_group_size == K && K <= 12288 && N >= g_knobs.pairs_min_n && N >= gmin && ws->gimg && g_knobs.gemm_kernel != 1;
    // several activation rows with 64-activation groups: the pair-wise build (two tables per lane, all three layouts);
    // otherwise one workgroup per act group (any act_group_size, and cheaper than it looks for a single row)
    hipError_t e = (act_group_size == 64 && N >= g_knobs.pairs_min_n)
        ? launch_preprocess_pairs(B_dev, act_dtype == TMAC_F16, ws->qlut_lds, ws->lut_scales, ws->lut_biases, K, N, ws->qlut_ref,
                                  ws->qlut_dev, ws->qdev_u4_per_row, (hipStream_t)stream)
        : (act_group_size == K && K <= 12288 && N >= g_knobs.pairs_min_n)
        ? launch_preprocess_pairs_row(B_dev, act_dtype == TMAC_F16, ws->qlut_lds, ws->lut_scales, ws->lut_biases, K, N, ws->qlut_ref,
                                      ws->qlut_dev, ws->qdev_u4_per_row, row_img ? ws->gimg : nullptr, row_img ? ws->gcol : nullptr, ws->gNpad,
                                      (hipStream_t)stream)
        : launch_preprocess(B_dev, (Dtype)act_dtype, ws->qlut_ref, ws->qlut_dev, ws->qlut_lds, ws->lut_scales, ws->lut_biases,
                            K, N, act_group_size, ws->qdev_u4_per_row, (hipStream_t)stream);
    if (e != hipSuccess) return fail(TMAC_HIP_E_RUNTIME, "preprocess launch: %s", hipGetErrorString(e));
    ws->gimg_valid = row_img; ws->gimg_kind = row_img ? 1 : 0;
    if (act_group_size == 64 && N >= gmin && ws->gimg && g_knobs.gemm_kernel != 1) {   // what k_gemm_planes streams (tmac_hip_qgemm_dev may pick it)
        e = launch_lut_image(B_dev, act_dtype == TMAC_F16, ws->gimg, ws->gcol, K, N, ws->gNpad, (hipStream_t)stream);
        if (e != hipSuccess) return fail(TMAC_HIP_E_RUNTIME, "LUT image launch: %s", hipGetErrorString(e));
        ws->gimg_valid = true; ws->gimg_kind = 2;      // (K == 64 with one act group per row: this image wins, the unified-scale GEMM is not offered it)
    }
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_workspace_ptrs(tmac_hip_workspace* ws, void** qlut_dev, size_t* nbytes_qlut_per_row,
                                           void** lut_scales, void** lut_biases) {
    if (!ws) return fail(TMAC_HIP_E_ARG, "null workspace");
    if (qlut_dev) *qlut_dev = ws->qlut_dev;
    if (nbytes_qlut_per_row) *nbytes_qlut_per_row = ws->qdev_u4_per_row * 16;
    if (lut_scales) *lut_scales = ws->lut_scales;
    if (lut_biases) *lut_biases = ws->lut_biases;
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_workspace_read(tmac_hip_workspace* ws, int8_t* qlut_host, float* lut_scales_host,
                                           float* lut_biases_host, int K, int N, int act_group_size, void* stream) {
    int32_t rc = check_lut_shape(ws, K, N, act_group_size);
    if (rc) return rc;
    if (ws->K != K || ws->N != N || ws->ags != act_group_size) return fail(TMAC_HIP_E_ARG, "workspace holds a different LUT");
    hipStream_t st = (hipStream_t)stream;
    const size_t G = (size_t)K / act_group_size;
    if (qlut_host) HIP_TRY(hipMemcpyAsync(qlut_host, ws->qlut_ref, (size_t)N * (K / 4) * 16, hipMemcpyDeviceToHost, st));
    if (lut_scales_host) HIP_TRY(hipMemcpyAsync(lut_scales_host, ws->lut_scales, sizeof(float) * N * G, hipMemcpyDeviceToHost, st));
    if (lut_biases_host) HIP_TRY(hipMemcpyAsync(lut_biases_host, ws->lut_biases, sizeof(float) * N * G, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_workspace_write(tmac_hip_workspace* ws, const int8_t* qlut_host, const float* lut_scales_host,
                                            const float* lut_biases_host, int K, int N, int act_group_size, void* stream) {
    int32_t rc = check_lut_shape(ws, K, N, act_group_size);
    if (rc) return rc;
    if (!qlut_host || !lut_scales_host || !lut_biases_host) return fail(TMAC_HIP_E_ARG, "null LUT pointer");
    hipStream_t st = (hipStream_t)stream;
    const size_t G = (size_t)K / act_group_size;
    ws->K = K; ws->N = N; ws->ags = act_group_size; ws->qdev_u4_per_row = qdev_u4_for_K(K);
    ws->gimg_valid = false;      // k_gemm_planes' LUT image (built from activations by tmac_hip_preprocessor_dev) no longer matches this LUT
    HIP_TRY(hipMemcpyAsync(ws->qlut_ref, qlut_host, (size_t)N * (K / 4) * 16, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(ws->lut_scales, lut_scales_host, sizeof(float) * N * G, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(ws->lut_biases, lut_biases_host, sizeof(float) * N * G, hipMemcpyHostToDevice, st));
    hipError_t e = launch_qlut_ref_to_dev(ws->qlut_ref, ws->qlut_dev, ws->qlut_lds, K, N, ws->qdev_u4_per_row, st);
    if (e != hipSuccess) return fail(TMAC_HIP_E_RUNTIME, "qlut_ref_to_dev launch: %s", hipGetErrorString(e));
    return TMAC_HIP_OK;
}
