/*
 * tmac_hip.h — C-ABI of libtmac_hip.so: T-MAC's LUT mpGEMM hot path on MI355X (gfx950).
 *
 * Drop-in boundary.  Plain pointers and sizes only; every function returns
 * 0 on success, -1 when no kernel matches the requested shape/configuration (the reference's
 * dispatcher contract, deploy/tuned/<set>/kernels.h `return -1`), and < -1 for runtime
 * failures (tmac_hip_last_error() explains).  No CPU fallback exists: without a HIP device
 * every compute entry point fails with TMAC_HIP_E_NODEVICE.
 *
 * Three layers:
 *   (1) reference-named entry points taking HOST pointers in the reference's own layouts —
 *       what the llama.cpp fork binds today (generated kernels.h):
 *         preprocessor_int8 / qgemm_lut_int8 and the shape-named kernels.
 *   (2) device-resident extensions — weights registered (uploaded + re-tiled) once, LUT
 *       workspace on the GPU, launches on a caller-provided hipStream_t.  This is the path
 *       that is benchmarked; a per-call H2D copy of 11 MB would erase the point.
 *   (3) parity taps used by tests: read back QLUT / integer partial sums.
 *
 * float_type: the reference uses fp16 on ARM and fp32 on x86 (python/t_mac/intrins/tbl.cc:10-15).
 * Layer (1) follows the HOST's float_type (fp32 on the x86 hosts of MI355X boxes); layer (2)
 * takes an explicit dtype per tensor.
 */
#ifndef TMAC_HIP_H_
#define TMAC_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libtmac_hip.so is built with -fvisibility=hidden: exactly what this header declares is exported */
#pragma GCC visibility push(default)

#define TMAC_HIP_OK 0
#define TMAC_HIP_E_NOMATCH (-1)   /* no kernel for this shape/config (reference: dispatcher returns -1) */
#define TMAC_HIP_E_NODEVICE (-2)  /* no usable HIP device */
#define TMAC_HIP_E_RUNTIME (-3)   /* a HIP call failed */
#define TMAC_HIP_E_ARG (-4)       /* invalid argument */

typedef enum { TMAC_F32 = 0, TMAC_F16 = 1 } tmac_dtype_t;

#define TMAC_HIP_ABI_VERSION 1   /* bumped when an exported signature changes */

/* Mirrors TMAC::TMACGeMMConfig (include/t-mac/tmac_gemm_wrapper.h:26-35) plus the three facts the
 * reference bakes into the compiled kernel instead of kcfg.ini (zero_point, act_group_size,
 * m_groups: python/t_mac/ops/qgemm.py:16-96). */
typedef struct tmac_kcfg {
    int bm;              /* M-tile in bit-plane rows                       (kcfg.ini: bm)        */
    int simd_n_in;       /* 16                                              (kcfg.ini)            */
    int simd_n_out;      /* 8                                               (kcfg.ini)            */
    int kfactor;         /* LUT groups per tbl call                         (kcfg.ini)            */
    int group_size;      /* weight quantisation group along K               (kcfg.ini)            */
    int lut_scales_size; /* N * K / act_group_size                          (kcfg.ini)            */
    int scales_size;     /* number of scale(+zero) values                   (kcfg.ini)            */
    int n_tile_num;      /* M / bm                                          (kcfg.ini)            */
    int act_group_size;  /* activations sharing one LUT scale (64; == K for BitNet on x86)       */
    int zero_point;      /* scales interleaved with zero points                                   */
    int m_groups;        /* -1: per-(row, group) scales; >=1: unified scale(s) (BitNet)           */
} tmac_kcfg;

typedef struct tmac_hip_weights tmac_hip_weights;      /* one registered weight matrix (device) */
typedef struct tmac_hip_workspace tmac_hip_workspace;  /* QLUT + LUT scales/biases (device)     */

/* ---- lifecycle ------------------------------------------------------------------------- */
int32_t tmac_hip_init(int device);            /* selects the device; idempotent               */
const char* tmac_hip_last_error(void);        /* thread-local message of the last failure     */
const char* tmac_hip_version(void);
int32_t tmac_hip_device_count(void);
/* 1 when p points into device memory (hipMalloc and friends), 0 for host memory of any kind (pageable, pinned) and for NULL.  For glue
 * code that receives tensors from a host framework and must decide between staging and passing the pointer on (src/ggml_tmac_hip.cc). */
int32_t tmac_hip_pointer_on_device(const void* p);

/* kcfg.ini handling: same file format and section names as the reference
 * (deploy/compile.py:153-165; lookup tmac_gemm_wrapper.h:230-255).  `path` NULL -> $TMAC_KCFG_FILE. */
int32_t tmac_hip_load_kcfg(const char* path);
/* replace != 0: the file becomes the whole table (the reference's runtime holds exactly one kcfg.ini); 0 merges as
 * tmac_hip_load_kcfg does.  tmac_hip_clear_kcfg empties the table.  The per-tile host-pointer entry points return -1 when
 * several loaded sections match their (bm, k, n, b) key and disagree on the quantisation layout. */
int32_t tmac_hip_load_kcfg_ex(const char* path, int replace);
int32_t tmac_hip_clear_kcfg(void);
/* Puts every piece of process-global state back to that of a freshly loaded library: kcfg table, tuning table, all
 * tmac_hip_set_* / tmac_hip_debug_* knobs, the host-pointer layer's caches, LUT workspace and staging buffers, the fused
 * entry point's per-stream workspaces.  Registered weights, workspaces and chains the caller holds stay valid.
 * Synchronises the library's own streams.  (tests/conftest.py calls it before every GPU test.) */
int32_t tmac_hip_reset_state(void);
/* M here is the number of WEIGHT rows (as in TMACGeMMWrapper::get_kcfg); fills zero_point /
 * act_group_size / m_groups from scales_size / lut_scales_size as the reference's shapes imply. */
int32_t tmac_hip_get_kcfg(int M, int K, int N, int bits, tmac_kcfg* out);
/* programmatic alternative to a file (used by tests and bench) */
int32_t tmac_hip_set_kcfg(int M, int K, int N, int bits, const tmac_kcfg* cfg);

/* ---- (2) device-resident path ---------------------------------------------------------- */

/* Upload + re-tile one weight matrix given in the REFERENCE layout (python/t_mac/weights.py:57-87):
 *   A_ref      uint8 [M/bm][K/4][bm/2]           (M = Mw*bits bit-plane rows)
 *   scales_ref float_type: zero_point [M/bm][K/gs][bm/bits/8][2][8]; else [M/bm][K/gs][bm/bits/8][8];
 *              m_groups>=1: [m_groups]
 * host_float : dtype of scales_ref (TMAC_F32 on x86 hosts, TMAC_F16 for ARM-produced blobs)
 * dev_float  : dtype the scales are STORED in on the GPU (TMAC_F16 halves their HBM traffic and is
 *              exact when the values are fp16-representable; TMAC_F32 keeps x86 bit parity)
 * The re-tiling is a pure permutation of nibbles (+ a bijective recoding of each nibble), so every
 * integer partial sum is identical to the reference's.  Row shards for multi-GPU: pass the tile
 * pointers of the shard and its Mw (tiles are contiguous in the reference layout). */
int32_t tmac_hip_register_weights(tmac_hip_weights** out, const void* A_ref, const void* scales_ref,
                                  int Mw, int K, int bits, const tmac_kcfg* cfg,
                                  tmac_dtype_t host_float, tmac_dtype_t dev_float, void* stream);
/* Same, but A_ref / scales_ref already live in device memory (bench: avoids a 1.6 GB PCIe upload). */
int32_t tmac_hip_register_weights_dev(tmac_hip_weights** out, const void* A_ref_dev, const void* scales_ref_dev,
                                      int Mw, int K, int bits, const tmac_kcfg* cfg,
                                      tmac_dtype_t host_float, tmac_dtype_t dev_float, void* stream);
int32_t tmac_hip_free_weights(tmac_hip_weights* w);
/* bytes one GEMV must read from HBM for this matrix (weights + scales), i.e. SURVEY.md §8d's
 * algorithmic bytes minus the activation-side terms */
size_t tmac_hip_weights_bytes(const tmac_hip_weights* w);

/* LUT workspace = TMACGeMMWrapper::set_workspace (tmac_gemm_wrapper.h:257-270) on the device. */
int32_t tmac_hip_workspace_create(tmac_hip_workspace** out, int maxK, int maxN);
int32_t tmac_hip_workspace_free(tmac_hip_workspace* ws);

/* preprocessor (lut_ctor.cc:38-266 + generated glue): activations B_dev [N][K] (act_dtype) ->
 * ws {QLUT int8, lut_scales, lut_biases}.  Bit-exact with the reference's fp32 arithmetic. */
int32_t tmac_hip_preprocessor_dev(tmac_hip_workspace* ws, const void* B_dev, tmac_dtype_t act_dtype,
                                  int K, int N, int act_group_size, void* stream);

/* qgemm_lut (tbl.cc + glue): C_dev [N][Mw] (out_dtype) = W x LUT(ws).  ws must hold the LUT of the
 * same K and act_group_size.  Whole matrix in one launch. */
int32_t tmac_hip_qgemm_dev(const tmac_hip_weights* w, const tmac_hip_workspace* ws, void* C_dev,
                           tmac_dtype_t out_dtype, int N, void* stream);

/* Fused form of llama_cpp_init + llama_cpp_compute (tmac_gemm_wrapper.h:170-228) for up to 4 weight
 * matrices that consume the SAME activation rows (q/k/v, gate/up): the LUT is built inside the GEMV
 * kernel (bit-exact with tmac_hip_preprocessor_dev) and every matrix is covered by one launch.
 *   weights[i] : registered matrices sharing K, bits and quantisation config
 *   B_dev      : activations [N][K] (act_dtype);   C_dev[i] : [N][Mw_i] (out_dtype)
 * With N at or above the GEMM threshold (tmac_hip_set_gemm_min_n) the call runs the preprocessor once into a
 * library-owned, per-stream workspace and the one-hot MFMA GEMM per matrix; the workspace is allocated on first use
 * (call once outside any stream capture) and released by tmac_hip_cache_clear(). */
int32_t tmac_hip_qgemm_fused_dev(const tmac_hip_weights* const* weights, int nmat, const void* B_dev,
                                 tmac_dtype_t act_dtype, void* const* C_dev, tmac_dtype_t out_dtype, int N,
                                 void* stream);

/* ---- deferred launches ---------------------------------------------------------------------
 * For callers that do not record (a backend hook called mat-mul by mat-mul).  tmac_hip_defer(1): from now on the calling thread's N = 1
 * tmac_hip_qgemm_fused_dev calls are QUEUED instead of launched; tmac_hip_flush launches what is queued as ONE stream-mode launch
 * (k_lut_images + k_gemv_stream: the independent-call path, 0.6-0.75 of the HBM peak instead of 0.25 for stand-alone launches).  The
 * queue never holds a dependence: a call that reads, or overwrites, anything a queued call writes (or overwrites what one reads), a call on
 * another stream, an N > 1 call and tmac_hip_defer(0) flush it first -- results are those of launching the calls in order.  The recording
 * built from a batch is cached by the batch's signature (matrices, pointers, dtypes; invalidated when weights are freed): a decode loop
 * pays for it once.  A batch that mixes configurations (bits, zero points, per-group / unified scales, scale or output dtype) becomes one
 * stream launch per configuration -- its calls are independent of each other; calls the persistent kernels do not cover, and configurations
 * with fewer than three calls (a stream launch costs ~10 us before its first byte), are launched one by one at the flush.  The caller must flush before it
 * synchronises the stream or reads an output.  tmac_hip_defer_stats: flushes, cache hits, stream-mode launches, calls launched singly.
 * The cached recordings of a thread are released by tmac_hip_cache_clear() / tmac_hip_reset_state() called on that thread. */
int32_t tmac_hip_defer(int on);
int32_t tmac_hip_flush(void* stream);
int32_t tmac_hip_defer_stats(uint64_t* flushes, uint64_t* cache_hits, uint64_t* stream_launches, uint64_t* single_calls);

/* ---- persistent decode chain ---------------------------------------------------------------
 * The fused calls of one decoded token (llama.cpp issues the same tmac_hip_qgemm_fused_dev sequence for every token:
 * llama_cpp_init + llama_cpp_compute per projection, tmac_gemm_wrapper.h:170-228) recorded ONCE and executed by ONE
 * persistent kernel launch (k_decode_chain): one workgroup per CU walks the whole list, weights of the next call stream in
 * while the current one computes, and an output that a later call takes as its activations is handed over inside the
 * launch (self-tagged fp16 granules, no kernel boundary).  Recording works like stream capture:
 *     tmac_hip_chain_begin();
 *     ... the thread's usual tmac_hip_qgemm_fused_dev(..., N = 1, ...) calls: noted, not launched ...
 *     tmac_hip_chain_end(&chain);
 *     tmac_hip_chain_launch(chain, stream);      // per token; outputs land in the C_dev buffers given while recording
 * Data flow is inferred from pointer identity: a call whose B_dev equals an earlier call's C_dev[i] consumes that output
 * inside the launch; any other B_dev must hold its activations when the launch starts.  Calls execute in recorded order.
 * An output buffer may be written by several calls (a decoder reuses its buffers layer after layer); the last one wins.
 * Scope: 1- to 4-bit QUAD-layout weights; per-group scales (group size >= 128, a power of two) with act groups of 64, or
 * unified scales (m_groups >= 1, BitNet) with one act group per row; fp16 activations, or fp32 for vectors that are in memory before the launch; chained outputs fp16; one weight
 * width, scale flavour, scale dtype and zero-point setting per chain.  Anything else: -1 from tmac_hip_chain_end and
 * the caller keeps launching the calls one by one.  Results are bit-identical to tmac_hip_qgemm_fused_dev with
 * the same threads per workgroup and waves per row quad (tmac_hip_debug_quad_config(tmac_hip_chain_threads(), wpq)).
 * A chain must not be launched concurrently with itself; the GPU must be able to hold one workgroup per CU (true unless
 * other work occupies CUs for the whole duration: every wait inside the kernel is bounded and reports through
 * tmac_hip_chain_status instead of hanging). */
typedef struct tmac_hip_chain tmac_hip_chain;
int32_t tmac_hip_chain_begin(void);
int32_t tmac_hip_chain_end(tmac_hip_chain** out);
/* ends the recording without building a chain (after an error while recording: the thread launches its calls again) */
int32_t tmac_hip_chain_abort(void);
int32_t tmac_hip_chain_launch(tmac_hip_chain* chain, void* stream);
/* after synchronising the stream: *error_word == 0 means every hand-off completed; otherwise (bit 31 | op << 8 | wave of
 * the first wave that gave up) the outputs of that launch are invalid.  Clears the word and re-arms the chain. */
int32_t tmac_hip_chain_status(tmac_hip_chain* chain, uint32_t* error_word);
/* number of recorded calls, waves per row quad chosen for call `op`, workgroups, weight + scale bytes one launch streams
 * (any pointer may be NULL) */
int32_t tmac_hip_chain_info(const tmac_hip_chain* chain, int op, int32_t* nops, int32_t* wpq, int32_t* grid, size_t* bytes);
int32_t tmac_hip_chain_free(tmac_hip_chain* chain);
/* Vector transforms inside the chain.  Between two mpGEMMs of a decoder layer sit element-wise operators the hot path does not own
 * (residual add + RMSNorm in front of q/k/v and gate/up, silu(gate) * up in front of the down projection); as kernels of their own they
 * would end the persistent launch after every call.  Every workgroup of the chain holds a call's whole activation vector when it builds
 * the LUT, so these operators are applied THERE: tmac_hip_chain_xform, while recording, describes a transform of the activations of the
 * NEXT recorded tmac_hip_qgemm_fused_dev call (B_dev = `in`).  fp32 arithmetic; the LUT is built from the fp32 result.
 *   TMAC_XF_NORM  t = in + residual;  x = gamma ? t * (1 / sqrt(mean(t^2) + eps)) * gamma : t
 *                 residual: fp32 [K] in device memory, NULL (none), or TMAC_XF_CARRY = the t that the latest NORM with keep != 0 kept
 *                 (inside the launch, no memory round trip; K <= 8192); residual_out: t also goes to memory (fp32 [K], e.g. the residual
 *                 stream for the next launch).  tmac_hip_chain_end checks it like every other write of the launch: it may overlap a
 *                 vector an EARLIER call of the chain reads from memory only when a hand-off path from a call in which every workgroup
 *                 owns rows leads to this one (the in-place residual stream of a decoder segment); never the vectors this or a later
 *                 call reads (a later NORM takes TMAC_XF_CARRY), never an output
 *   TMAC_XF_GLU   x = silu(in) * in2;  in2: fp16 [K], an earlier output of the chain (handed over like `in`) or external memory -- of the
 *                 same kind as `in`; K <= 12288
 *                 (when `in` and `in2` are outputs 0 and 1 of one earlier two-matrix call of the chain and nothing else reads output 0 through a
 *                 hand-off, that call publishes silu(in) * in2 itself, once per row, rounded to fp16 like every handed-over vector)
 * A decoder then runs one launch per segment between two operators that stay outside (attention): o -> gate/up -> down -> next q/k/v.
 * These are extensions without a reference counterpart (T-MAC has no norm operator): tests compare them with the same formulas in
 * numpy fed through the oracle (tolerance, not bits: the mean square is summed in another order).
 * NO PER-CALL FALLBACK: "-1 from tmac_hip_chain_end, keep launching the calls one by one" holds for recordings WITHOUT transforms only --
 * a transform has no stand-alone counterpart in this library.  A caller that records transforms must be able to run the operators itself
 * (ggml does: the segment glue is an optimisation of a graph that already has norm / glu nodes) when tmac_hip_chain_end refuses the
 * recording (LDS beyond 160 KB, K beyond the limits above).  A transform declared in front of a call that is rejected is dropped with it. */
#define TMAC_XF_NONE 0
#define TMAC_XF_NORM 1
#define TMAC_XF_GLU 2
#define TMAC_XF_CARRY ((const float*)1)
typedef struct {
    int32_t kind;
    const void* in2;
    const float* residual;
    const float* gamma;
    float eps;
    float* residual_out;
    int32_t keep;
} tmac_hip_xform;
int32_t tmac_hip_chain_xform(const tmac_hip_xform* xf);
/* Row-sharded chains (one process per GPU; weight ROWS split over the ranks).  While recording, the exchange step between
 * a call and the calls that need its output whole is recorded too -- tmac_hip_comm_allgather(comm, send, recv, ...) notes itself, or
 * tmac_hip_chain_record_gather where no communicator exists -- and inside the launch it becomes part of the hand-off: every rank's
 * producers store their granules into the hand-off arena of EVERY rank (IPC-mapped peer memory, system-scope stores over xGMI), the
 * consumers' polls are unchanged.  `recv` itself is NOT written by the launch.  Protocol, same recorded sequence on every rank:
 *     tmac_hip_chain_end(&chain);  tmac_hip_chain_export(chain, blob);          // TMAC_HIP_CHAIN_BLOB_BYTES
 *     ... all-gather the blobs over any transport (rank order) ...
 *     tmac_hip_chain_connect(chain, blobs_of_all_ranks, world);                 // hipIpcOpenMemHandle of the peers' arenas
 *     per token, on every rank: tmac_hip_chain_launch(chain, stream)            // the ranks' launches must overlap in time
 * K is never split, integer sums are those of one GPU.  HSA_ENABLE_IPC_MODE_LEGACY=0 must be set where the driver supports dmabuf IPC only. */
#define TMAC_HIP_CHAIN_BLOB_BYTES 128
int32_t tmac_hip_chain_record_gather(const void* send_dev, void* recv_dev, size_t bytes_per_rank, int rank, int world);
int32_t tmac_hip_chain_export(const tmac_hip_chain* chain, void* blob_out);
int32_t tmac_hip_chain_connect(tmac_hip_chain* chain, const void* blobs_of_all_ranks, int world);
int32_t tmac_hip_chain_threads(void);   /* threads per workgroup of k_decode_chain (the launch configuration its results are bit-identical with) */
/* Stream mode.  A recording in which NO call consumes another call's output (and none carries a transform) is a list of independent
 * GEMVs -- SURVEY 8(d)'s back-to-back measurement, or a caller that evaluates many vectors against many matrices.  tmac_hip_chain_end
 * then builds a different launch (1 = yes): k_lut_images builds the tables of every call ONCE -- the reference's own call structure,
 * llama_cpp_init per activation vector, then lookups only (tmac_gemm_wrapper.h:170-228) -- and k_gemv_stream walks the calls with the
 * tables prebuilt, a loader wave per workgroup staging the next call's tables while the lookup waves stream this call's weights; the
 * weight prefetch runs across call boundaries.  No hand-offs, no spins: residency is not a correctness condition there.  Since round 6
 * the calls are dealt to CLASSES of workgroups (a call that is small for 256 CUs is served by a fraction of them with more rows each,
 * other classes work on other calls meanwhile: TMAC_STREAM_NCLS=1 in the environment = every workgroup visits every call; further A/B
 * knobs read by tmac_hip_chain_end: TMAC_STREAM_VISIT_ITEMS (items a visit should give a workgroup, 160), TMAC_STREAM_LPT=0 (calls dealt in
 * recorded order instead of largest first), TMAC_STREAM_SPLIT=1 (one workgroup per CU), TMAC_STREAM_QW=0 | 1 (item form, below)).  Same integers
 * as every N = 1 path (tmac_hip_chain_set_tap); float outputs: see the return value 2 below.  Per-group scales, or unified scales
 * (BitNet: the row's scale and the sequential bias chain by k_lut_images_us, scale-final on exact int32 totals);
 * TMAC_CHAIN_STREAM=0 in the environment keeps the ordinary chain (A/B). */
/* Returns 0 (k_decode_chain), 1 (stream mode) or 2: stream mode in the QUARTER-WALK form, chosen when a K with a ragged last 64-unit step
 * (11008, 3200, 8640 ...) would otherwise spend whole lookup items on zero tables: rows dealt in groups of 16, K walked in quarters of a
 * step.  Same integers (tmac_hip_chain_set_tap shows them); a row's fp32 partial sums are added in another order than the stand-alone
 * launch adds them, so per-group-scale outputs are specified to the path's tolerance (<= 1e-3 of the reference, measured <= 2e-5) instead
 * of bit-identical to that launch; unified-scale outputs stay bit-identical.  TMAC_STREAM_QW=0 in the environment keeps form 1 (A/B). */
int32_t tmac_hip_chain_is_stream(const tmac_hip_chain* chain);
/* profiling / A-B knobs: s_memrealtime stamps (100 MHz) [calls][workgroups][8] of wave 0 (0 call entry, 1 activations complete, 2 LUT
 * built, 3 weights of the call landed, 5 last row quad published, 6 all loads landed, 7 number of polls) into a device buffer (NULL = off); waves per row quad forced for chains built from now on (0 = per-call
 * choice) and the poll limit of a hand-off (0 = keep).  The stamps exist in profiling builds of the library only (-DTMAC_CHAIN_STAMPS=1 for
 * k_decode_chain, -DTMAC_STREAM_STAMPS for k_gemv_stream: the idle hooks cost the dependent token 3 %); otherwise a non-NULL buffer returns -1 */
int32_t tmac_hip_chain_set_stamps(tmac_hip_chain* chain, unsigned long long* dev_buffer);
/* Parity tap of the persistent kernels (k_decode_chain, k_gemv_stream): the INTEGERS of every recorded call as they enter the float part of
 * the path, written by the launch itself into a caller's device buffer (NULL = off; launches with a tap run the kernels' tap instantiation).
 * Call `op` owns ints [offset, offset + count) (tmac_hip_chain_tap_layout; op == number of calls: offset = size of the whole buffer):
 * per-group scales: int32 [rows of the call's matrices, concatenated][K / 64] = sum_p 2^p PS_p per (weight row, act group), PS_p the
 * per-plane partial sums of tbl.cc:445-462 (what tmac_hip_qgemm_partial_sums returns per plane); unified scales: int32 [rows][bits], the
 * exact per-plane totals of tbl.cc:586-628.  Not available for calls whose rows are dealt in gate / up pairs (GLU in the producer). */
int32_t tmac_hip_chain_set_tap(tmac_hip_chain* chain, int32_t* dev_buffer);
int32_t tmac_hip_chain_tap_layout(const tmac_hip_chain* chain, int op, size_t* offset_ints, size_t* count_ints);
int32_t tmac_hip_debug_chain_config(int force_waves_per_quad, unsigned spin_limit);
/* workgroups of chains built from now on (0 = one per CU): lets two chains run side by side on one device (tests/test_gpu_chain_ipc.py) */
int32_t tmac_hip_debug_chain_grid(int workgroups);
/* The stream-mode schedule as a pure function (no device is touched): n independent calls of items[i] lookup items each, `grid` row
 * ranges in `ncls` classes (a power of two <= 16, <= grid).  Call i is dealt to the aligned block of out_w[i] classes that starts at
 * class out_lo[i]; out_load[c] (may be NULL) = the items one range of class c walks over the launch.  `target` = the fewest items a
 * visit should give a range (TMAC_STREAM_VISIT_ITEMS), lpt != 0 = largest calls first (TMAC_STREAM_LPT). */
int32_t tmac_hip_debug_stream_schedule(const double* items, int n, int grid, int ncls, int target, int lpt, int32_t* out_lo, int32_t* out_w, double* out_load);

/* ---- multi-GPU exchange step ------------------------------------------------------------------
 * One process per GPU; weight ROWS are sharded over the ranks (tile-aligned; register a rank's tiles with
 * tmac_hip_register_weights: tiles are contiguous in the reference layout).  An M-tile needs only its own A / Scales
 * slice plus the WHOLE LUT (include/t-mac/tmac_gemm_wrapper.h:197-199), so the only exchange is an all-gather of what the
 * next LUT build needs whole: the activation block the shards produced (N x rows x 2 bytes), or int8 QLUT slices built from
 * a K slice (tmac_hip_workspace_ptrs gives the device pointers).  K is never split: integer sums stay bit-identical to
 * one GPU.  RCCL over xGMI, resolved with dlopen at the first call (single-GPU users need no RCCL).
 *   rank 0:     tmac_hip_comm_unique_id(id)  -> hand the TMAC_HIP_COMM_ID_BYTES to every rank (MPI, a file, a socket)
 *   every rank: hipSetDevice / tmac_hip_init(its GPU); tmac_hip_comm_init(&comm, id, rank, world)
 *   per step:   tmac_hip_comm_allgather(comm, my_part_dev, whole_dev, bytes_per_rank, stream)   (whole = world x bytes_per_rank)
 * Errors: tmac_hip_comm_last_error(). */
#define TMAC_HIP_COMM_ID_BYTES 128
typedef struct tmac_hip_comm tmac_hip_comm;
int32_t tmac_hip_comm_unique_id(void* id_out);
int32_t tmac_hip_comm_init(tmac_hip_comm** out, const void* id, int rank, int world);
int32_t tmac_hip_comm_allgather(tmac_hip_comm* comm, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream);
int32_t tmac_hip_comm_destroy(tmac_hip_comm* comm);
/* The same exchange step without RCCL: every rank owns a window in fine-grained device memory that all peers map through hipIpc*; a
 * small kernel publishes this rank's part and copies the peers' parts out of their windows (flags in the windows, no host in the
 * loop).  For nodes or tests where RCCL cannot serve -- several ranks on ONE device, for instance, which RCCL refuses.
 *   every rank: tmac_hip_comm_init_ipc(&comm, max_bytes_per_rank, rank, world);  tmac_hip_comm_export(comm, blob)
 *   all-gather the TMAC_HIP_COMM_BLOB_BYTES blobs over any transport (rank order);  tmac_hip_comm_connect(comm, blobs, world)
 *   per step:   tmac_hip_comm_allgather(comm, ...) as above (every rank issues the same sequence);  tmac_hip_comm_status after a
 *   synchronisation: 0, or a bit per rank whose part did not arrive in time.  HSA_ENABLE_IPC_MODE_LEGACY=0 where the driver has dmabuf IPC only. */
#define TMAC_HIP_COMM_BLOB_BYTES 128
int32_t tmac_hip_comm_init_ipc(tmac_hip_comm** out, size_t max_bytes_per_rank, int rank, int world);
int32_t tmac_hip_comm_export(const tmac_hip_comm* comm, void* blob_out);
int32_t tmac_hip_comm_connect(tmac_hip_comm* comm, const void* blobs_of_all_ranks, int world);
int32_t tmac_hip_comm_status(tmac_hip_comm* comm, uint32_t* error_word);
const char* tmac_hip_comm_last_error(void);

/* Raw device pointers of the workspace, for collectives (RCCL all-gather of the LUT over xGMI):
 *   qlut_dev  : kernel-layout half tables, nbytes_qlut per activation row
 *   lut_scales/lut_biases : fp32 [N][K/act_group_size] */
int32_t tmac_hip_workspace_ptrs(tmac_hip_workspace* ws, void** qlut_dev, size_t* nbytes_qlut_per_row,
                                void** lut_scales, void** lut_biases);

/* ---- (3) parity taps -------------------------------------------------------------------- */
/* QLUT in the reference layout int8 [N][K/4][16] + fp32 lut_scales/lut_biases [N][K/ags] -> host */
int32_t tmac_hip_workspace_read(tmac_hip_workspace* ws, int8_t* qlut_host, float* lut_scales_host,
                                float* lut_biases_host, int K, int N, int act_group_size, void* stream);
/* load a host-side reference-layout LUT into the workspace (what qgemm_lut_int8 does internally) */
int32_t tmac_hip_workspace_write(tmac_hip_workspace* ws, const int8_t* qlut_host, const float* lut_scales_host,
                                 const float* lut_biases_host, int K, int N, int act_group_size, void* stream);
/* Same kernel as tmac_hip_qgemm_dev with the integer tap enabled: PS_host int32 [N][M][K/ags] in the
 * reference's M-space (bit-plane) row order; for the unified-scale path [N][M] (K/ags == 1). */
int32_t tmac_hip_qgemm_partial_sums(const tmac_hip_weights* w, const tmac_hip_workspace* ws,
                                    int32_t* PS_host, int N, void* stream);
/* integer tap of the fused kernel (one matrix): PS_host as above, optional fp32 C_host [N][Mw], optional
 * lut_host [N][2][K/ags] = the LUT scales then biases the kernel built in LDS */
int32_t tmac_hip_qgemm_fused_partial_sums(const tmac_hip_weights* w, const void* B_dev, tmac_dtype_t act_dtype,
                                          int32_t* PS_host, float* C_host, float* lut_host, int N, void* stream);
/* Runs v_perm_b32 / v_mqsad_pk_u16_u8 / lookup4 on n quadruples of host words (in[4n] -> out[4n]); the
 * test-suite compares the result with the host models of tmac_amd/csrc/tmac_core.h. */
int32_t tmac_hip_selftest(const uint32_t* in_host, uint32_t* out_host, int n);
/* one wave of v_mfma_i32_16x16x64_i8: in[64][8] (A regs 0-3, B regs 4-7 per lane) -> out[64][4] (D regs) */
int32_t tmac_hip_selftest_mfma(const uint32_t* in_host, int32_t* out_host);
/* Select the GEMV kernel variant: 0 auto (fused layout where supported), 1/2 two-kernel tiled path
 * (mqsad / byte-add accumulate), 3 generic reference-layout kernel, 4 fused (v_mqsad accumulate),
 * 5 fused with the MFMA accumulate, 6 wave-per-row-quad kernel (QUAD layout, MFMA accumulate; what auto picks
 * for 1- to 4-bit weights), 7 the same with the v_mqsad accumulate.  Affects weights
 * registered AFTER the call (the variant fixes their device layout).  For A/B benchmarking and tests. */
int32_t tmac_hip_set_variant(int variant);
/* From n activation rows on the N > 1 entry points run an MFMA GEMM (k_gemm_planes, else k_gemm_onehot: the table gather written as
 * an int8 contraction, same bit-exact integer sums) instead of looping the GEMV kernel over the rows (qgemm.py:183-190); n = 0
 * disables it, any value other than the default 32 is taken literally.  At the default the library decides per launch: where
 * k_gemm_planes covers the configuration (1- to 4-bit weights with per-group scales and act groups of 64, 2-bit with a unified scale)
 * from the measured crossover on -- 7 to 16 rows depending on the matrices, tmac_dispatch.cpp gemm_pays; 12 rows through the split
 * entry points, whose LUT build does not know the matrix -- else from 32 rows (64 when the matrices of the call have fewer than
 * 128 x 64 output rows x bits / 2: an under-filled grid). */
int32_t tmac_hip_set_gemm_min_n(int n);
/* Fast aggregation (the reference's `-fa` build option, deploy/compile.py:167-174,223): the
 * looked-up bytes of an act group are folded by a tree of rounding-halving adds instead of being summed exactly
 * (SignedHalvingAdder, tbl.cc:86-141,201-256), the result rescaled by the group's table count and corrected by the
 * analytic bias (tbl.cc:301-318,474-477).  LOSSY; off by default as in every shipped reference configuration.
 *   0  exact sums (default)
 *   1  signed halving adds (vrhaddq_s8) -- the reference's ARM build, the one `-fa` is meant for
 *   2  the reference's AVX2 build (_mm256_avg_epu8 applied to the signed bytes): kept so that the implementation can
 *      be pinned bit for bit against reference code compiled on an x86 host; numerically meaningless
 * A build-time property in the reference, a registration-time property here: applies to weights registered AFTER the
 * call (they take the 16-table-segment layout and run the two-kernel path: preprocessor + qgemm).  Per-group-scale
 * weights with act_group_size 32 or 64 only, as in the reference (no fast aggregation on its int32 path). */
int32_t tmac_hip_set_fast_aggregation(int mode);
/* Measurement aid for bench.py: one launch that only reads `bytes` of device memory (non-temporal 16-byte loads) and
 * keeps nothing -- the cost of streaming a matrix's bytes with no LUT build and no lookups, under the same launch
 * mechanism as the GEMV it is compared with.  dev_sink: >= 4 KB of device scratch. */
int32_t tmac_hip_debug_stream_read(const void* dev_src, size_t bytes, void* dev_sink, void* stream);
/* A/B and profiling knobs (tools/, tests/): force the decode kernel's launch configuration (0, 0 = heuristic / tuned
 * table); s_memtime phase stamps [workgroups][2][8] of the next fused launches into a device buffer (NULL = off);
 * activation rows from which tmac_hip_preprocessor_dev builds the LUT pair-wise (k_preprocess_pairs; default 2). */
int32_t tmac_hip_debug_quad_config(int force_threads, int force_waves_per_quad);
int32_t tmac_hip_debug_stamps(unsigned long long* dev_buffer);
int32_t tmac_hip_debug_pairs_min_n(int n);
/* N > 1 kernel selection and parity taps.  0 (default): k_gemm_planes (bit-planes combined inside the matrix-core operand,
 * tmac_gemm2.hip) wherever it covers the configuration (1- to 4-bit weights with per-group scales and act_group_size 64; 2-bit
 * weights with a unified scale);
 * 1: always k_gemm_onehot (one matrix-core row per bit-plane row).  2 / 3: k_gemm_planes with its eight-wave (one workgroup per
 * CU) / four-wave (two per CU) workgroup form forced; 0 picks by the number of tiles of the launch (tmac_gemm2.hip, PForm).
 * tmac_hip_debug_gemm_comb_sums: int32 [N][Mw][K/64], the integers sum_p 2^p PS_p (PS_p as tmac_hip_qgemm_partial_sums
 * returns them per bit-plane) that k_gemm_planes feeds into the fp32 chain; the workspace must hold the LUT of a
 * tmac_hip_preprocessor_dev call with N >= 2 rows.  tmac_hip_debug_gemm_image_read: the LUT image that kernel streams,
 * in plain layouts: signed half tables int8 [N][K/4][8] (entry j of table t; entry 15 - j is its negation,
 * lut_ctor.cc:152-155), lut_scales, lut_biases and the sum of each act group's 128 half-table entries, fp32 [N][K/64]. */
int32_t tmac_hip_debug_gemm_kernel(int which);
/* profiling: workgroup 0 of the following k_gemm_planes launches writes s_memrealtime stamps [wave 8][step 64][8] into this
 * device buffer (NULL = off); tools/gemm2_stamps.py prints them.  The kernel honours it in profiling builds of the library only
 * (-DTMAC_G2_STAMPS=1; compiled out by default: the hook cost the 4-bit prefill line 1 %) -- otherwise the buffer stays untouched */
int32_t tmac_hip_debug_gemm_stamps(unsigned long long* dev_buffer);
int32_t tmac_hip_debug_gemm_comb_sums(const tmac_hip_weights* w, const tmac_hip_workspace* ws, int32_t* comb_host, int N, void* stream);
int32_t tmac_hip_debug_gemm_image_read(const tmac_hip_workspace* ws, int8_t* half_tables_host, float* lut_scales_host,
                                       float* lut_biases_host, float* entry_sums_host, int N, void* stream);
/* host-pointer entry points: 1 (default) = tiles with contiguous weight / scale pointers are grouped into runs once they
 * have been seen, and a run's output is computed in one launch per LUT and handed out tile by tile; 0 = every tile call is
 * served on its own */
int32_t tmac_hip_debug_host_runs(int on);
/* test knob: 0 skips the synchronisation that orders tmac_hip_workspace_create's null-stream fills before the workspace's
 * first user on another stream -- the round-2 defect (all-zero LUT image), kept reproducible for tests/test_gpu_hostptr.py */
int32_t tmac_hip_debug_ws_fill_sync(int on);
/* Launch-configuration tuner of the fused decode kernel (the role autotvm's grid search over
 * (bm, kfactor, bn) plays for the reference's CPU kernels, python/t_mac/ops/base.py:84-127, qgemm.py:98-116).
 * tmac_hip_autotune_fused times every (threads per workgroup, waves per row quad) configuration of k_gemv_quad on the
 * given 1..4 registered matrices (the set that tmac_hip_qgemm_fused_dev will be called with, N = 1), on HBM-cold
 * rotating copies of the weights inside a replayed hipGraph, and records the fastest when it beats the built-in
 * heuristic by more than 4 %; later fused calls with the same (bits, K, rows, matrices, dtypes) use it.  Results do not
 * change (same kernel, same arithmetic).  best_ft = 0 on return means "the heuristic stands".  Allocates and frees its
 * own buffers (a few hundred MB); synchronous; not for use inside a stream capture.
 * tmac_hip_tune_save / tmac_hip_tune_load persist the table as text (return the number of entries, or < 0);
 * $TMAC_HIP_TUNE_FILE is loaded on the first fused call. */
int32_t tmac_hip_autotune_fused(const tmac_hip_weights* const* weights, int nmat, tmac_dtype_t act_dtype,
                                tmac_dtype_t out_dtype, int* best_ft, int* best_wpq, float* best_us, float* heuristic_us);
int32_t tmac_hip_tune_save(const char* path);
int32_t tmac_hip_tune_load(const char* path);
int32_t tmac_hip_tune_clear(void);

/* ---- (1) reference-named host-pointer entry points ---------------------------------------
 * Signatures identical to the generated deploy/tuned/<set>/kernels.h.  `m` is bm for qgemm_lut
 * (one M-tile; A/Scales/C are the tile pointers, tmac_gemm_wrapper.h:197-228) and Mw*bits for the
 * preprocessor.  Tile weights are uploaded on first use and cached by host pointer (llama.cpp
 * keeps weights mmap'd for the life of the model); tmac_hip_cache_clear() drops the cache.
 * Configuration (group_size, zero_point, act_group_size) comes from the loaded kcfg
 * (tmac_hip_load_kcfg / $TMAC_KCFG_FILE), exactly as the reference couples kernels to kcfg.ini. */
int32_t qgemm_lut_int8(int m, int k, int n, int b, void* A, void* LUT, void* Scales, void* LUT_Scales,
                       void* LUT_Biases, void* C);
int32_t preprocessor_int8(int m, int k, int n, int b, void* B, void* LUT_Scales, void* LUT_Biases, void* QLUT);
int32_t tmac_hip_cache_clear(void);

/* shape-named kernels of the four checked-in sets (deploy/tuned/aarch64-<set>/kernels.h) */
#define TMAC_DECL_Q(bm, k, n, b) \
    int32_t qgemm_lut_t1_int8_m##bm##_k##k##_n##n##_b##b(void* A, void* LUT, void* Scales, void* LUT_Scales, void* LUT_Biases, void* C);
#define TMAC_DECL_P(m, k, n, b) \
    int32_t preprocessor_t1_int8_m##m##_k##k##_n##n##_b##b(void* B, void* LUT_Scales, void* LUT_Biases, void* QLUT);
/* llama-2-7b-2bit */
TMAC_DECL_Q(128, 4096, 1, 2) TMAC_DECL_Q(128, 11008, 1, 2)
TMAC_DECL_P(8192, 4096, 1, 2) TMAC_DECL_P(22016, 4096, 1, 2) TMAC_DECL_P(8192, 11008, 1, 2)
/* llama-2-7b-4bit */
TMAC_DECL_Q(1024, 4096, 1, 4) TMAC_DECL_Q(256, 4096, 1, 4) TMAC_DECL_Q(256, 11008, 1, 4)
TMAC_DECL_P(16384, 4096, 1, 4) TMAC_DECL_P(44032, 4096, 1, 4) TMAC_DECL_P(16384, 11008, 1, 4)
/* llama-3-8b-2bit */
TMAC_DECL_Q(256, 4096, 1, 2) TMAC_DECL_Q(512, 4096, 1, 2) TMAC_DECL_Q(128, 14336, 1, 2)
TMAC_DECL_P(28672, 4096, 1, 2) TMAC_DECL_P(8192, 14336, 1, 2) TMAC_DECL_P(2048, 4096, 1, 2)
/* hf-bitnet-3b */
TMAC_DECL_Q(128, 8640, 1, 2) TMAC_DECL_Q(128, 3200, 1, 2) TMAC_DECL_Q(320, 3200, 1, 2)
TMAC_DECL_P(6400, 8640, 1, 2) TMAC_DECL_P(17280, 3200, 1, 2) TMAC_DECL_P(6400, 3200, 1, 2)
#undef TMAC_DECL_Q
#undef TMAC_DECL_P

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* TMAC_HIP_H_ */
