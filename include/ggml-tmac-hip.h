/*
 * ggml-tmac-hip.h — the llama.cpp / ggml side of the T-MAC op hook on MI355X.
 *
 * The reference ships no source for its hook (3rdparty/llama.cpp is an empty submodule); what is visible is its contract:
 * the fork is built with -DGGML_TMAC=ON, includes "t-mac/tmac_gemm_wrapper.h" and "t-mac/kernels.h", converts weights with
 * --enable-t-mac into GGUF tensors whose data is the blob of python/t_mac/model_utils.py:243-271 ([weight tiles][fp32 scales],
 * laid out by the kcfg.ini the model was converted with), and for every mul_mat with such a weight calls
 * TMACGeMMWrapper::llama_cpp_init (preprocessor) on the main thread and llama_cpp_compute tile by tile on its worker threads
 * (include/t-mac/tmac_gemm_wrapper.h:170-228, tools/run_pipeline.py:181-188).
 *
 * Two ways to put that on the GPU:
 *   (1) unchanged call sites: include/t-mac/tmac_gemm_wrapper.h of this repository is source-compatible; host pointers in,
 *       host pointers out (PCIe in the loop);
 *   (2) this file: the three calls a ggml backend hook needs for a DEVICE-RESIDENT weight -- upload once at model load,
 *       mul_mat per token, free -- written against the handful of ggml_tensor fields it touches, so that the glue in the
 *       fork is three one-line call sites.
 * `struct tmac_ggml_tensor` mirrors those fields (ggml.h: ne[], data, extra); a fork passes its own ggml_tensor members.
 */
#ifndef GGML_TMAC_HIP_H_
#define GGML_TMAC_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct tmac_ggml_tensor {
    int64_t ne[4];   /* ggml: ne[0] = K (inner dimension), ne[1] = rows (M for weights, N for activations) */
    void* data;      /* host memory: weights = the T-MAC blob; activations / outputs = fp32 row-major */
    void* extra;     /* backend-private: ggml_tmac_hip_upload stores its handle here (ggml_tensor::extra) */
};

/* once per process: loads kcfg.ini (path, or $TMAC_KCFG_FILE when NULL) and selects the device */
int ggml_tmac_hip_init(const char* kcfg_file, int device);
/* does the hook take this mul_mat?  (a kcfg entry exists for the weight's shape and bit width) */
int ggml_tmac_hip_can_mul_mat(const struct tmac_ggml_tensor* w, int bits);
/* model load: w->data = blob of a [M = ne[1]][K = ne[0]] weight with `bits` bits; registers it on the GPU, w->extra = handle */
int ggml_tmac_hip_upload(struct tmac_ggml_tensor* w, int bits);
/* dst[N][M] (fp32, host) = x[N][K] (fp32, host) x W^T: activations up through pinned staging, LUT build + mpGEMM on the
 * device, outputs back down; safe to call from one thread per process (ggml calls a backend's mul_mat from its main thread) */
int ggml_tmac_hip_mul_mat(const struct tmac_ggml_tensor* w, const struct tmac_ggml_tensor* x, struct tmac_ggml_tensor* dst);
void ggml_tmac_hip_free(struct tmac_ggml_tensor* w);
const char* ggml_tmac_hip_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
