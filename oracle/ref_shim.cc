// ref_shim.cc — builds the REFERENCE's own intrinsics into oracle/_ref/libtmac_ref_intrins.so.
//
// TEST INFRASTRUCTURE.  This file contains no T-MAC arithmetic of its own: it #includes
// python/t_mac/intrins/{tbl,lut_ctor}.cc straight from /root/reference (via -I on the compiler
// command line, see oracle/Makefile — the sources are never copied into this repo),
// instantiates the reference's macros for the parameter combinations our tests use, and wraps
// them in the same loop nest the TVM-generated glue uses
// (deploy/tuned/aarch64-llama-2-7b-2bit/kernels.cc:1059-1064 and :1223-1231) so that shapes
// with no checked-in prebuilt kernel (bits 1/3/4, no zero-point, act_group 32, int32
// aggregation) can still be run through the reference arithmetic.
#include <cstdio>
#include <cstdint>
#include <cstring>

#include "tbl.cc"       // -I/root/reference/python/t_mac/intrins
#include "lut_ctor.cc"

extern "C" {
// float path: kfactor 16, ActK 16 (act_group 64), zero-point on/off, bits 1..4
tbl_g4_int8_float_update(true, 16, 1, 16, false, true, false)
tbl_g4_int8_float_update(true, 16, 2, 16, false, true, false)
tbl_g4_int8_float_update(true, 16, 3, 16, false, true, false)
tbl_g4_int8_float_update(true, 16, 4, 16, false, true, false)
tbl_g4_int8_float_update(true, 16, 1, 16, false, false, false)
tbl_g4_int8_float_update(true, 16, 2, 16, false, false, false)
tbl_g4_int8_float_update(true, 16, 3, 16, false, false, false)
tbl_g4_int8_float_update(true, 16, 4, 16, false, false, false)
// act_group 32: ActK 8 with kfactor 8 and kfactor 16
tbl_g4_int8_float_update(true, 8, 2, 8, false, true, false)
tbl_g4_int8_float_update(true, 8, 2, 8, false, false, false)
tbl_g4_int8_float_update(true, 8, 4, 8, false, true, false)
tbl_g4_int8_float_update(true, 16, 2, 8, false, true, false)
tbl_g4_int8_float_update(true, 16, 2, 8, false, false, false)
// fast aggregation (a9): the AVX2 flavour of SignedHalvingAdder (tbl.cc:201-256)
tbl_g4_int8_float_update(true, 16, 1, 16, true, true, false)
tbl_g4_int8_float_update(true, 16, 2, 16, true, true, false)
tbl_g4_int8_float_update(true, 16, 3, 16, true, true, false)
tbl_g4_int8_float_update(true, 16, 4, 16, true, true, false)
tbl_g4_int8_float_update(true, 16, 2, 16, true, false, false)
tbl_g4_int8_float_update(true, 16, 3, 16, true, false, false)
tbl_g4_int8_float_update(true, 16, 4, 16, true, false, false)
tbl_g4_int8_float_update(true, 8, 2, 8, true, true, false)
tbl_g4_int8_float_update(true, 16, 2, 8, true, true, false)
// int32 aggregation (scale-final / BitNet on x86)
tbl_g4_int8_int32_update(true, 16, 1, 16, false, false, false)
tbl_g4_int8_int32_update(true, 16, 2, 16, false, false, false)
tbl_g4_int8_int32_update(true, 16, 3, 16, false, false, false)
tbl_g4_int8_int32_update(true, 16, 4, 16, false, false, false)
tbl_g4_int8_int32_update(true, 8, 2, 8, false, false, false)
lut_ctor(0, 1)
lut_ctor(0, 2)
lut_ctor(0, 3)
lut_ctor(0, 4)
}

typedef int32_t (*tbl_float_fn)(int32_t, void*, int8_t*, uint8_t*, void*, void*, void*);
typedef int32_t (*tbl_i32_fn)(int32_t, int32_t*, int8_t*, uint8_t*);

static tbl_float_fn pick_float(int kfactor, int bits, int actk, int zp) {
#define PICK(k, b, ak, z, fn) if (kfactor == k && bits == b && actk == ak && zp == z) return fn;
    PICK(16, 1, 16, 1, tbl_g4_int8_float_update_strue_k16_b1_ak16_fafalse_ztrue_osfalse)
    PICK(16, 2, 16, 1, tbl_g4_int8_float_update_strue_k16_b2_ak16_fafalse_ztrue_osfalse)
    PICK(16, 3, 16, 1, tbl_g4_int8_float_update_strue_k16_b3_ak16_fafalse_ztrue_osfalse)
    PICK(16, 4, 16, 1, tbl_g4_int8_float_update_strue_k16_b4_ak16_fafalse_ztrue_osfalse)
    PICK(16, 1, 16, 0, tbl_g4_int8_float_update_strue_k16_b1_ak16_fafalse_zfalse_osfalse)
    PICK(16, 2, 16, 0, tbl_g4_int8_float_update_strue_k16_b2_ak16_fafalse_zfalse_osfalse)
    PICK(16, 3, 16, 0, tbl_g4_int8_float_update_strue_k16_b3_ak16_fafalse_zfalse_osfalse)
    PICK(16, 4, 16, 0, tbl_g4_int8_float_update_strue_k16_b4_ak16_fafalse_zfalse_osfalse)
    PICK(8, 2, 8, 1, tbl_g4_int8_float_update_strue_k8_b2_ak8_fafalse_ztrue_osfalse)
    PICK(8, 2, 8, 0, tbl_g4_int8_float_update_strue_k8_b2_ak8_fafalse_zfalse_osfalse)
    PICK(8, 4, 8, 1, tbl_g4_int8_float_update_strue_k8_b4_ak8_fafalse_ztrue_osfalse)
    PICK(16, 2, 8, 1, tbl_g4_int8_float_update_strue_k16_b2_ak8_fafalse_ztrue_osfalse)
    PICK(16, 2, 8, 0, tbl_g4_int8_float_update_strue_k16_b2_ak8_fafalse_zfalse_osfalse)
#undef PICK
    return nullptr;
}

static tbl_float_fn pick_float_fa(int kfactor, int bits, int actk, int zp) {
#define PICK(k, b, ak, z, fn) if (kfactor == k && bits == b && actk == ak && zp == z) return fn;
    PICK(16, 1, 16, 1, tbl_g4_int8_float_update_strue_k16_b1_ak16_fatrue_ztrue_osfalse)
    PICK(16, 2, 16, 1, tbl_g4_int8_float_update_strue_k16_b2_ak16_fatrue_ztrue_osfalse)
    PICK(16, 3, 16, 1, tbl_g4_int8_float_update_strue_k16_b3_ak16_fatrue_ztrue_osfalse)
    PICK(16, 4, 16, 1, tbl_g4_int8_float_update_strue_k16_b4_ak16_fatrue_ztrue_osfalse)
    PICK(16, 2, 16, 0, tbl_g4_int8_float_update_strue_k16_b2_ak16_fatrue_zfalse_osfalse)
    PICK(16, 3, 16, 0, tbl_g4_int8_float_update_strue_k16_b3_ak16_fatrue_zfalse_osfalse)
    PICK(16, 4, 16, 0, tbl_g4_int8_float_update_strue_k16_b4_ak16_fatrue_zfalse_osfalse)
    PICK(8, 2, 8, 1, tbl_g4_int8_float_update_strue_k8_b2_ak8_fatrue_ztrue_osfalse)
    PICK(16, 2, 8, 1, tbl_g4_int8_float_update_strue_k16_b2_ak8_fatrue_ztrue_osfalse)
#undef PICK
    return nullptr;
}

static tbl_i32_fn pick_i32(int kfactor, int bits) {
    if (kfactor == 16 && bits == 1) return tbl_g4_int8_int32_update_strue_k16_b1_ak16_fafalse_zfalse_osfalse;
    if (kfactor == 16 && bits == 2) return tbl_g4_int8_int32_update_strue_k16_b2_ak16_fafalse_zfalse_osfalse;
    if (kfactor == 16 && bits == 3) return tbl_g4_int8_int32_update_strue_k16_b3_ak16_fafalse_zfalse_osfalse;
    if (kfactor == 16 && bits == 4) return tbl_g4_int8_int32_update_strue_k16_b4_ak16_fafalse_zfalse_osfalse;
    if (kfactor == 8 && bits == 2) return tbl_g4_int8_int32_update_strue_k8_b2_ak8_fafalse_zfalse_osfalse;
    return nullptr;
}

extern "C" {

// One M-tile of the float path: CBits[bm] (fp32) after the k_outer loop.  Pointers are the
// per-tile pointers the llama.cpp caller would pass (tmac_gemm_wrapper.h:197-199).
static int32_t tile_cbits_float(int fa, int bits, int kfactor, int ags, int zp, int bm, int K, int gs,
                                void* A_tile, void* LUT, void* Scales_tile, void* LUT_Scales,
                                void* LUT_Biases, float* CBits) {
    const int actk = (ags / 4 < kfactor) ? ags / 4 : kfactor;
    tbl_float_fn fn = fa ? pick_float_fa(kfactor, bits, actk, zp) : pick_float(kfactor, bits, actk, zp);
    if (!fn) return -1;
    const int sstride = bm / bits * (zp ? 2 : 1);
    tbl_float_reset(bm, CBits);
    for (int k_outer = 0; k_outer < K / 4 / kfactor; ++k_outer) {
        fn(bm, CBits, (int8_t*)LUT + (size_t)k_outer * kfactor * 16,
           (uint8_t*)A_tile + (size_t)k_outer * (bm / 2) * kfactor,
           (float*)Scales_tile + (size_t)(k_outer * 4 * kfactor / gs) * sstride,
           (float*)LUT_Scales + k_outer * 4 * kfactor / ags,
           (float*)LUT_Biases + k_outer * 4 * kfactor / ags);
    }
    return 0;
}

int32_t ref_tile_cbits_float(int bits, int kfactor, int ags, int zp, int bm, int K, int gs,
                             void* A_tile, void* LUT, void* Scales_tile, void* LUT_Scales,
                             void* LUT_Biases, float* CBits) {
    return tile_cbits_float(0, bits, kfactor, ags, zp, bm, K, gs, A_tile, LUT, Scales_tile, LUT_Scales, LUT_Biases, CBits);
}
// same with the fast-aggregation instantiation of the intrinsic (FastAggregation = true)
int32_t ref_tile_cbits_float_fa(int bits, int kfactor, int ags, int zp, int bm, int K, int gs,
                                void* A_tile, void* LUT, void* Scales_tile, void* LUT_Scales,
                                void* LUT_Biases, float* CBits) {
    return tile_cbits_float(1, bits, kfactor, ags, zp, bm, K, gs, A_tile, LUT, Scales_tile, LUT_Scales, LUT_Biases, CBits);
}

// One M-tile of the int32 path: CBits32[bm] after the k_outer loop.
int32_t ref_tile_cbits_int32(int bits, int kfactor, int bm, int K, void* A_tile, void* LUT,
                             int32_t* CBits) {
    tbl_i32_fn fn = pick_i32(kfactor, bits);
    if (!fn) return -1;
    tbl_int32_reset(bm, CBits);
    for (int k_outer = 0; k_outer < K / 4 / kfactor; ++k_outer)
        fn(bm, CBits, (int8_t*)LUT + (size_t)k_outer * kfactor * 16,
           (uint8_t*)A_tile + (size_t)k_outer * (bm / 2) * kfactor);
    return 0;
}

// Per-(row, act group) integer partial sums of one M-tile, obtained from the reference's
// int32 intrinsic by resetting the accumulator at every act group.  PS is [bm][K/ags].
int32_t ref_tile_partial_sums(int bits, int kfactor, int ags, int bm, int K, void* A_tile,
                              void* LUT, int32_t* PS) {
    tbl_i32_fn fn = pick_i32(kfactor, bits);
    if (!fn || ags % (4 * kfactor)) return -1;
    const int G = K / ags, calls_per_group = ags / 4 / kfactor;
    int32_t* tmp = new int32_t[bm];
    for (int kk = 0; kk < G; ++kk) {
        tbl_int32_reset(bm, tmp);
        for (int c = 0; c < calls_per_group; ++c) {
            int k_outer = kk * calls_per_group + c;
            fn(bm, tmp, (int8_t*)LUT + (size_t)k_outer * kfactor * 16,
               (uint8_t*)A_tile + (size_t)k_outer * (bm / 2) * kfactor);
        }
        for (int r = 0; r < bm; ++r) PS[(size_t)r * G + kk] = tmp[r];
    }
    delete[] tmp;
    return 0;
}

// Preprocessor for one activation row, same call sequence as the generated glue.
int32_t ref_preprocessor(int K, int ags, void* B, void* LUT_Scales, void* LUT_Biases, void* QLUT) {
    float* b = (float*)B;
    float* ls = (float*)LUT_Scales;
    float* lb = (float*)LUT_Biases;
    for (int kk = 0; kk < K / ags; ++kk) {
        partial_max_reset(ls + kk);
        for (int k = 0; k < ags / 32; ++k) partial_max_g4_int8_k8(ls + kk, b + kk * ags + k * 32);
    }
    for (int kk = 0; kk < K / ags; ++kk)
        lut_ctor_g4_int8_k0_b2(ags, (int8_t*)QLUT + (size_t)kk * ags / 4 * 16, b + kk * ags, ls + kk, lb + kk);
    return 0;
}

}  // extern "C"
