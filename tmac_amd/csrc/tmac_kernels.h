// tmac_kernels.h — host-callable launchers of the gfx950 kernels (implemented in tmac_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tmac_layout.h"

namespace tmac {

enum Dtype { F32 = 0, F16 = 1 };

// GEMV kernel variants (A/B-able at run time through tmac_hip_set_variant)
enum Variant {
    V_AUTO = 0,      // k_gemv_quad (V_QUAD) where it supports the configuration, else the row-block fused kernel, else 1
    V_LO_MQSAD = 1,  // two-kernel path: k_preprocess + tiled k_gemv_lo, v_mqsad_pk_u16_u8 accumulate
    V_LO_SDWA = 2,   // same, byte-select adds
    V_REF_LAYOUT = 3,// generic kernel on the reference blobs (reference float order)
    V_FUSED = 4,     // row-block fused kernel k_gemv_fused (ts = 8 layout): LUT in LDS, v_mqsad accumulate
    V_FUSED_MFMA = 5,// same kernel with the matrix-pipe (v_mfma_i32_16x16x64_i8) accumulate
    V_QUAD = 6,      // wave-owns-row-quad fused kernel k_gemv_quad (QUAD layout), MFMA accumulate
    V_QUAD_MQSAD = 7 // same with the v_mqsad_pk_u16_u8 accumulate (512-thread workgroups)
};

struct FusedMat {
    const uint4* W;   // ts = 8 device layout
    const void* SC;   // device layout scales (or [m_groups] unified scales)
    void* C;          // [N][Mw]
    int Mw;
    int nb_end;       // cumulative number of 16-row blocks up to and including this matrix
};

struct FusedArgs {
    FusedMat m[4];
    int nmat;
    Shape s;                 // K, bits, gs, ags, zero_point, m_groups, ts = 8 (Mw unused)
    const void* B;           // activations [N][K]                       (LUT built in-kernel)
    int act_f16;
    const void* qlut_lds;    // uint4 [N][qlut_lds_u4(K)]                (LUT prebuilt by k_preprocess)
    const float* lut_scales; // fp32 [N][K/ags]
    const float* lut_biases;
    int sc_f16, out_f16;
    int32_t* dump;           // optional integer tap (nmat == 1)
    unsigned long long* stamps; // optional s_memtime phase stamps [blocks][8] (debug/profiling)
    float* lut_tap;          // optional: block 0 writes [N][2][G] LUT scales | biases it built (parity tap)
    int acc_mfma;            // 1: v_mfma_i32_16x16x64_i8 accumulate, 0: v_mqsad_pk_u16_u8 (default: measured faster, profiles/)
    int nu, nsb, tstride, G, nsg, gs_shift;   // filled by launch_gemv_fused (host-side divides)
};

// one-hot MFMA GEMM for N > 1 activation rows (tmac_gemm.hip); weights in the QUAD layout; up to 4 matrices that share
// K, the quantisation config and the LUT in one launch
struct GemmMat {
    const void* W;            // QUAD layout weights
    const void* SC;           // QUAD layout scales
    void* C;                  // [N][Mw]
    int Mw;
    int wg_end;               // cumulative workgroup count along x (filled by the launcher)
};
struct GemmArgs {
    Shape s;                  // K, bits, gs, ags, zero_point (Mw unused)
    GemmMat m[4];
    int nmat;
    int sc_f16, out_f16;
    const void* qlut_lds;     // uint4 [N][4][tstride]: half tables, unit-major image written by k_preprocess
    int tstride;
    const float* lut_scales;  // fp32 [N][K/ags]
    const float* lut_biases;
    int32_t* dump;            // optional integer tap [N][M][K/ags] (nmat == 1)
    int N;
};

// plane-combined one-hot GEMM (tmac_gemm2.hip): the LUT comes as the image k_lut_image writes
struct Gemm2Args {
    Shape s;                  // K, bits, gs, ags = 64, zero_point (Mw unused)
    GemmMat m[4];
    int nmat;
    int sc_f16, out_f16;
    const uint4* bimg;        // per-group scales (k_lut_image): uint4 [K/64][Npad/64][8][64], the chunk of (act group, 64 rows) in one piece | unified
                              // scales (k_preprocess_pairs_row): uint4 [K/32 units][4 pairs][Npad]: signed half tables of tables 2P, 2P+1 of the unit
    const float* colv;        // per-group: float4 [K/64][Npad] = lut_scales / 2 | lut_biases / 2 | entry sum | lut_biases;  unified: fp32 [3][Npad]
    int Npad, N;
    int32_t* dump;            // optional tap [N][Mw][K/64]: sum_p 2^p PS_p (nmat == 1)
    int gx, gy;               // filled by the launcher: row blocks (64 rows, all matrices) and token blocks (64 rows of activations)
    int apg_shift;            // filled by the launcher: log2(act groups per weight group)
    unsigned long long* stamps; // optional s_memrealtime stamps of workgroup 0: [wave 8][step 64][8] (profiling; tools/gemm2_stamps.py)
    int form;                 // 0: by the number of tiles | 1: eight-wave workgroups, one per CU | 2: four-wave workgroups, two per CU
};

struct GemvArgs {
    Shape s;
    const void* W;        // device layout weights (uint4)        | reference blob for V_REF_LAYOUT
    const void* SC;       // device layout scales (sc_dtype)      | reference blob (float_type = sc_dtype)
    Dtype sc_dtype;
    const void* qlut_dev; // uint4 [N][qlut_dev_u4]
    const int8_t* qlut_ref; // int8 [N][K/4][16] (V_REF_LAYOUT only)
    const float* lut_scales;
    const float* lut_biases;
    void* C;              // [N][Mw]
    Dtype out_dtype;
    int32_t* ps_dump;     // optional int32 tap (device)
    int N;
    int fa_mode;          // 0 exact sums | 1, 2: fast aggregation (a9; see SegAcc<BITS, 2> in tmac_core.h)
};

hipError_t launch_selftest(const uint32_t* in, uint32_t* out, int n, hipStream_t st);
hipError_t launch_selftest_mfma(const uint32_t* in, int32_t* out, hipStream_t st);
hipError_t launch_selftest_permlane(const uint32_t* in, uint32_t* out, hipStream_t st);
hipError_t launch_retile_weights(const uint8_t* A_ref, void* Wd, const Shape& s, hipStream_t st);
hipError_t launch_retile_scales(const void* S_ref, Dtype in_dt, void* Sd, Dtype out_dt, const Shape& s, hipStream_t st);
hipError_t launch_preprocess(const void* B, Dtype act_dt, int8_t* qlut_ref, void* qlut_dev, void* qlut_lds, float* lut_scales,
                             float* lut_biases, int K, int N, int ags, size_t qdev_u4_per_row, hipStream_t st);
// all-gather over IPC-mapped windows (tmac_comm.cpp, transport "ipc"): workgroup p of `world` handles rank p's part
struct IpcGatherArgs {
    const unsigned char* send;     // this rank's part
    unsigned char* recv;           // world x bytes
    size_t bytes;                  // per rank
    unsigned char* win[8];         // every rank's window (own + the peers' as mapped into this process): two halves of win_half bytes
    unsigned* flag[8];             // every rank's two flag words (generation of the all-gather whose part the half holds)
    size_t win_half;
    int rank, world;
    unsigned gen;                  // 1, 2, ...: half = gen & 1
    unsigned spin_limit;
    unsigned* err;                 // set when a peer's part did not arrive
};
hipError_t launch_ipc_allgather(const IpcGatherArgs& a, hipStream_t st);
hipError_t launch_qlut_ref_to_dev(const int8_t* qlut_ref, void* qlut_dev, void* qlut_lds, int K, int N, size_t qdev_u4_per_row, hipStream_t st);
// pair-wise LUT build (tmac_quad.hip; ags = 64): the half-table image + LUT scales/biases, and -- when qlut_ref / qlut_dev are
// given (both or neither) -- the other two layouts of the workspace as well
hipError_t launch_preprocess_pairs(const void* B, int act_f16, void* qlut_lds, float* lut_scales, float* lut_biases, int K, int N,
                                   int8_t* qlut_ref, void* qlut_dev, size_t qdev_u4_per_row, hipStream_t st);
// the same for act_group_size = K (one act group per activation row; K <= 12288)
// bimg / colv (both or neither; Npad = row stride of the image): also the LUT image k_gemm_planes_us streams (tmac_gemm2.hip)
hipError_t launch_preprocess_pairs_row(const void* B, int act_f16, void* qlut_lds, float* lut_scales, float* lut_biases, int K, int N,
                                       int8_t* qlut_ref, void* qlut_dev, size_t qdev_u4_per_row, void* bimg, float* colv, int Npad, hipStream_t st);
hipError_t launch_stream_read(const void* src, size_t bytes, void* sink, hipStream_t st);
// host-pointer route (tmac_kernels.hip): results into pinned host memory by the GPU's own stores, then a flag the host spins on
hipError_t launch_host_copy3_flag(const void* q, size_t nq, const void* ls, const void* lb, size_t ns, void* dst_pinned, uint32_t* flag,
                                  uint32_t val, hipStream_t st);
hipError_t launch_host_flag(uint32_t* flag, uint32_t val, hipStream_t st);   // measurement aid, see tmac_kernels.hip
bool gemm_onehot_supported(const Shape& s);
hipError_t launch_gemm_onehot(const GemmArgs& a, hipStream_t st);
bool gemm_planes_supported(const Shape& s);
hipError_t launch_lut_image(const void* B, int act_f16, void* bimg, float* colv, int K, int N, int Npad, hipStream_t st);
hipError_t launch_gemm_planes(const Gemm2Args& a, hipStream_t st);
// fused kernel (tmac_fused.hip)
bool gemv_fused_supported(const Shape& s);
size_t qlut_lds_u4(int K);   // uint4 per activation row of the LDS-image LUT
hipError_t launch_gemv_fused(const FusedArgs& a, int N, bool build_lut, hipStream_t st);
// quad kernel (tmac_quad.hip): a.m[i].nb_end = cumulative ROW QUAD counts; force_ft/force_wpq 0 = heuristic
bool gemv_quad_supported(const Shape& s);
hipError_t launch_gemv_quad(const FusedArgs& a, int N, bool build_lut, int force_ft, int force_wpq, hipStream_t st);
// returns hipErrorInvalidValue when the variant does not cover the configuration
hipError_t launch_gemv(const GemvArgs& a, Variant v, hipStream_t st);
bool gemv_lo_supported(const Shape& s);

}  // namespace tmac
