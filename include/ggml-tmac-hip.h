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
/* Tensors whose data already lives in device memory are passed on unstaged.  The glue launches on a stream of its own
 * (ggml_tmac_hip_stream(), non-blocking): work that PRODUCED such an x on another stream is not ordered in front of the launch.
 * A device backend therefore hands the glue ITS stream once -- every launch and copy of the glue is then ordered with the backend's
 * own kernels -- or synchronises its stream before it calls.  NULL returns to the glue's own stream.  Pending work is drained first. */
int ggml_tmac_hip_set_stream(void* hip_stream);
void ggml_tmac_hip_free(struct tmac_ggml_tensor* w);
const char* ggml_tmac_hip_last_error(void);

/* ---- device-resident mat-muls without recording ----------------------------------------------------------------------------
 * A device backend whose hook is called mat-mul by mat-mul (no view of the graph) and whose tensors live in device memory:
 * ggml_tmac_hip_mul_mat_dev enqueues ONE N = 1 call -- 1..4 weights that share x (fp32 or fp16 [K]), outputs fp32 or fp16 [M] each -- on
 * ggml_tmac_hip_stream() and returns at once.  With ggml_tmac_hip_set_deferred(1) such calls are QUEUED and ggml_tmac_hip_flush() launches
 * the queue as one stream-mode launch (tmac_hip.h: tmac_hip_defer / tmac_hip_flush): the calls between two synchronisation points that do
 * not depend on each other -- q, k, v issued one by one; the projections of several sequences -- run at 0.6-0.75 of the HBM peak instead
 * of 0.25.  A call that depends on a queued one flushes the queue by itself; ggml_tmac_hip_synchronize() flushes, then waits. */
int ggml_tmac_hip_mul_mat_dev(const struct tmac_ggml_tensor* const* w, int nw, const void* x_dev, int x_is_f32, void* const* dst_dev, int dst_is_f32);
int ggml_tmac_hip_set_deferred(int on);
int ggml_tmac_hip_flush(void);
int ggml_tmac_hip_synchronize(void);

/* ---- decoder segments ------------------------------------------------------------------------------------------------------
 * A decoded token issues the same mat-muls in the same order every time; between two of them sit element-wise operators (residual
 * add + RMSNorm in front of q/k/v and gate/up, silu(gate) * up in front of the down projection) and, once per layer, an operator
 * that stays outside the hook (attention).  A backend that sees the graph (ggml_backend_graph_compute) records, ONCE, the mat-muls
 * between two outside operators as a segment -- typically o -> gate/up -> down -> next layer's q/k/v -- with the element-wise
 * operators declared in front of the mat-mul that consumes their result; per token it computes the segment with ONE launch
 * (include/tmac_hip.h: tmac_hip_chain_*, tmac_hip_chain_xform).  All tensors of a segment live in DEVICE memory: activations and
 * outputs fp16 ([K] / [M], N = 1), the residual stream and the norm weights fp32.
 *     ggml_tmac_hip_segment_begin();
 *     ggml_tmac_hip_segment_mul_mat(&wo, 1, attn_out, &o);                                  // o projection
 *     ggml_tmac_hip_segment_norm(h, 0, ffn_norm_w, eps, NULL, 1);                           // t = o + h, x = rmsnorm(t) * w; t kept
 *     ggml_tmac_hip_segment_mul_mat(w_gate_up, 2, o, gate_up);
 *     ggml_tmac_hip_segment_glu(up);                                                        // x = silu(gate) * up
 *     ggml_tmac_hip_segment_mul_mat(&wdown, 1, gate, &down);
 *     ggml_tmac_hip_segment_norm(NULL, 1, next_attn_norm_w, eps, h_next, 0);                // t = down + kept t; h_next = t
 *     ggml_tmac_hip_segment_mul_mat(w_qkv_next, 3, down, qkv_next);
 *     ggml_tmac_hip_segment_end(&seg);
 *     per token:  <attention of layer l on ggml_tmac_hip_stream()>;  ggml_tmac_hip_segment_compute(seg_l);  ...
 * The transform declared by _norm / _glu applies to the NEXT _mul_mat, whose x is the transform's `in`. */
typedef struct ggml_tmac_hip_segment ggml_tmac_hip_segment;
int ggml_tmac_hip_segment_begin(void);
int ggml_tmac_hip_segment_norm(const float* residual, int residual_is_kept, const float* norm_weight, float eps, float* residual_out, int keep);
int ggml_tmac_hip_segment_glu(const void* in2_f16);
int ggml_tmac_hip_segment_mul_mat(const struct tmac_ggml_tensor* const* w, int nw, const void* x_f16, void* const* dst_f16);
/* the same with x as an fp32 vector in device memory that no earlier mat-mul of the segment wrote (ggml's graphs are fp32: the output of
 * the attention operator in front of the o projection, the token embedding in front of the first q/k/v): the tables are built from the
 * fp32 values, as tmac_hip_qgemm_fused_dev does on its own */
int ggml_tmac_hip_segment_mul_mat_f32(const struct tmac_ggml_tensor* const* w, int nw, const float* x_f32, void* const* dst_f16);
int ggml_tmac_hip_segment_end(ggml_tmac_hip_segment** seg);
/* A segment is all or nothing: when ggml_tmac_hip_segment_mul_mat or _end returns non-zero the recording is over (no chain exists, nothing
 * is pending) and the caller evaluates the graph's own nodes -- the element-wise operators of a segment have no stand-alone counterpart in
 * this library.  ggml_tmac_hip_segment_abort ends a recording explicitly (after a failed _norm / _glu, or a change of mind). */
int ggml_tmac_hip_segment_abort(void);
int ggml_tmac_hip_segment_compute(ggml_tmac_hip_segment* seg);   /* one launch on ggml_tmac_hip_stream(); does not wait */
int ggml_tmac_hip_segment_wait(ggml_tmac_hip_segment* seg);      /* synchronises the stream; 0 if every hand-off of the segment's launches completed */
void ggml_tmac_hip_segment_free(ggml_tmac_hip_segment* seg);
void* ggml_tmac_hip_stream(void);                                /* the hipStream_t the glue launches on: outside operators go there too */

#ifdef __cplusplus
}
#endif
#endif
