// tmac_quad_core.h — device functions shared by the QUAD-layout decode kernels: k_gemv_quad (tmac_quad.hip, one launch
// per fused GEMV group) and k_decode_chain (tmac_chain.hip, one persistent launch per recorded call sequence).
// The LUT build primitives follow lut_ctor.cc:120-215 bit for bit; the lookup follows tbl.cc:445-462 (see tmac_core.h).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "tmac_core.h"
#include "tmac_fastdiv.h"

namespace tmac {

typedef uint32_t u32x4q __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float qdpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ uint32_t qdpp_u(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
// max over the 16 lanes of a DPP row, result in every lane.  v_max_f32 with a DPP source operand: fmaxf() through
// update_dpp costs a v_mov_dpp plus canonicalising v_max pairs (5 instructions per step instead of 1).  The s_nop
// covers the VALU-write -> DPP-read hazard, which the compiler does not track through inline asm.
__device__ __forceinline__ float q_row_allmax(float v) {
    asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
        : "+v"(v));
    return v;
}
__device__ __forceinline__ float q_alpha(int p) { return p == 0 ? 0.5f : (p == 1 ? 1.0f : (p == 2 ? 2.0f : 4.0f)); }
__device__ __forceinline__ float q_ld_scale(const void* p, int f16, size_t i) {
    return f16 ? __half2float(reinterpret_cast<const __half*>(p)[i]) : reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void q_st_out(void* C, int f16, size_t i, float v) {
    if (f16) reinterpret_cast<__half*>(C)[i] = __float2half_rn(v);
    else reinterpret_cast<float*>(C)[i] = v;
}

// biased half-table byte U = sat8(rne(x)) + 128 packed into byte `pos` of `acc`:
// v_mul, v_rndne, v_med3, v_add, v_cvt_pk_u8_f32 (the conversion of an integer-valued float is exact)
__device__ __forceinline__ uint32_t q_quant_pack(float e, float t_scales, int pos, uint32_t acc) {
    float y = rintf(__fmul_rn(e, t_scales));
    y = fminf(fmaxf(y, -127.0f), 127.0f);   // finite inputs never exceed +-127; keeps U in [1,255]
    return __builtin_amdgcn_cvt_pk_u8_f32(__fadd_rn(y, 128.0f), pos, acc);
}

template <int BITS>
struct QFrag {
    uint32_t wd[8 * BITS / 2];
    uint32_t sraw[4];
};

template <bool ZP>
__device__ __forceinline__ float qfrag_scale(const uint32_t (&sraw)[4], int f16, int i, int which) {
    const int e = i * (ZP ? 2 : 1) + which;
    if (f16) {
        const uint32_t wv = sraw[e >> 1];
        return __half2float(__ushort_as_half((unsigned short)((e & 1) ? (wv >> 16) : (wv & 0xffff))));
    }
    return __uint_as_float(sraw[e]);
}

typedef int qv4i_t __attribute__((ext_vector_type(4)));
typedef float qv2f __attribute__((ext_vector_type(2)));

// (x & m) | k in one VALU instruction (hipcc emits v_and_b32 + v_or_b32 for two literal operands: VOP3 takes no
// literals on gfx9, so the constants are kept in an SGPR and a VGPR)
__device__ __forceinline__ uint32_t q_and_or(uint32_t x, uint32_t m_sgpr, uint32_t k_vgpr) {
    uint32_t d;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "s"(m_sgpr), "v"(k_vgpr));
    return d;
}

// Signed lookup for the MFMA accumulate: `all` = the four looked-up half-table entries, `neg` = those whose nibble has
// the negate bit, zero elsewhere.  The selector matrix weighs them +1 and -2: sum(all) - 2 sum(neg) = sum(pos) - sum(neg),
// exact in int32, and one v_perm_b32 less per four lookups than routing every entry to a plus or a minus word.
template <int H>
__device__ __forceinline__ void q_lookup4_pm(uint32_t w, uint32_t tab_lo, uint32_t tab_hi, uint32_t k3, uint32_t& all, uint32_t& neg) {
    const uint32_t x = H ? (w >> 4) : w;
    all = __builtin_amdgcn_perm(tab_hi, tab_lo, x & 0x07070707u);
    const uint32_t sel3 = q_and_or(x >> 1, 0x04040404u, k3);       // byte i: i (-> 0) or 4 + i (-> entry i) by the negate bit
    neg = __builtin_amdgcn_perm(all, 0u, sel3);
}

// One LUT table from its 4 activations (lut_ctor.cc:120-215): the 8 distinct magnitudes ((x0 +- x1) +- x2) +- x3 in the
// reference's association order, two per v_pk_add_f32; q = rne(L * t_scales) through the 1.5*2^23 magic add (|L * t_scales|
// <= 127 for finite input, so the sum's ulp is 1 and its low byte is q in two's complement; + 128 in the magic gives the
// biased byte).  Half table j = 0..7 holds {-L15, L1, -L13, L3, -L11, L5, -L9, L7}: negated entries as magic - product.
// Returns the dwords [j0 j1 j2 j3], [j4 j5 j6 j7] and L15 (its negation is the table's LUT[0], summed into lut_biases).
template <bool SIGNED>
__device__ __forceinline__ void q_table8(float x0, float x1, float x2, float x3, float t_scales, uint32_t& lo, uint32_t& hi, float& L15) {
    const qv2f x01 = {x0, x1}, x23 = {x2, x3};
    qv2f apm, l2m, l2p, L31, L119, L75, L1513;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(apm) : "v"(x01));                      // {x0+x1, x0-x1}
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(l2m) : "v"(apm), "v"(x23)); // {a_p-x2, a_m-x2}
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(l2p) : "v"(apm), "v"(x23));                          // {a_p+x2, a_m+x2}
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(L31) : "v"(l2m), "v"(x23)); // {L3, L1}
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(L119) : "v"(l2m), "v"(x23));                         // {L11, L9}
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(L75) : "v"(l2p), "v"(x23)); // {L7, L5}
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(L1513) : "v"(l2p), "v"(x23));                        // {L15, L13}
    L15 = L1513.x;
    const qv2f tt = {t_scales, t_scales};
    const qv2f mg = {SIGNED ? 12582912.0f : 12583040.0f, 0.0f};
    qv2f za, zb, zc, zd;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(za) : "v"(L31), "v"(tt));
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(zb) : "v"(L75), "v"(tt));
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(zc) : "v"(L119), "v"(tt));
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(zd) : "v"(L1513), "v"(tt));
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(za) : "v"(za), "v"(mg));                                  // j = 3 | 1
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(zb) : "v"(zb), "v"(mg));                                  // j = 7 | 5
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]" : "=v"(zc) : "v"(zc), "v"(mg));        // j = 4 | 6
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]" : "=v"(zd) : "v"(zd), "v"(mg));        // j = 0 | 2
    lo = __builtin_amdgcn_perm(__float_as_uint(za.y), __float_as_uint(zd.x), 0x0c0c0400u) |
         __builtin_amdgcn_perm(__float_as_uint(za.x), __float_as_uint(zd.y), 0x04000c0cu);
    hi = __builtin_amdgcn_perm(__float_as_uint(zb.y), __float_as_uint(zc.x), 0x0c0c0400u) |
         __builtin_amdgcn_perm(__float_as_uint(zb.x), __float_as_uint(zc.y), 0x04000c0cu);
}

// max over the 8 lanes of half a DPP row (the two quads of one act group when a lane holds two tables)
__device__ __forceinline__ float q_half_allmax(float v) {
    asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf"
        : "+v"(v));
    return v;
}

// value of lane ^ 16 / lane ^ 32 combined with the lane's own, on the VALU: gfx950's v_permlane16_swap / v_permlane32_swap exchange rows of 16
// (halves of 32) lanes between two registers; fed the same value twice they return {own | partner} and {partner | own} halves, whose sum
// (max) is what `x op __shfl_xor(x, 16 | 32)` computes -- bit for bit (the operations are commutative) -- without the ds_bpermute round trip
// through the LDS crossbar (two in a row in front of every reduction barrier of k_gemv_quad and k_decode_chain).  A/B knob: -DTMAC_SWAP_REDUCE=0.
#ifndef TMAC_SWAP_REDUCE
#define TMAC_SWAP_REDUCE 1
#endif
template <int W> __device__ __forceinline__ void q_swap_pair(uint32_t x, uint32_t& a0, uint32_t& a1) {
    if constexpr (W == 16) { const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false); a0 = r[0]; a1 = r[1]; }
    else { const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false); a0 = r[0]; a1 = r[1]; }
}
__device__ __forceinline__ float q_xor_add_f(float x) {          // x + x(lane ^ 16), then + (lane ^ 32)
#if TMAC_SWAP_REDUCE
    uint32_t a0, a1;
    q_swap_pair<16>(__float_as_uint(x), a0, a1); x = __fadd_rn(__uint_as_float(a0), __uint_as_float(a1));
    q_swap_pair<32>(__float_as_uint(x), a0, a1); x = __fadd_rn(__uint_as_float(a0), __uint_as_float(a1));
    return x;
#else
    x = __fadd_rn(x, __shfl_xor(x, 16, 64));
    return __fadd_rn(x, __shfl_xor(x, 32, 64));
#endif
}
__device__ __forceinline__ uint32_t q_xor_add_u(uint32_t x) {
#if TMAC_SWAP_REDUCE
    uint32_t a0, a1;
    q_swap_pair<16>(x, a0, a1); x = a0 + a1;
    q_swap_pair<32>(x, a0, a1); x = a0 + a1;
    return x;
#else
    x += (uint32_t)__shfl_xor((int)x, 16, 64);
    return x + (uint32_t)__shfl_xor((int)x, 32, 64);
#endif
}
__device__ __forceinline__ float q_xor_max_f(float x) {
#if TMAC_SWAP_REDUCE
    uint32_t a0, a1;
    q_swap_pair<16>(__float_as_uint(x), a0, a1); x = fmaxf(__uint_as_float(a0), __uint_as_float(a1));
    q_swap_pair<32>(__float_as_uint(x), a0, a1); x = fmaxf(__uint_as_float(a0), __uint_as_float(a1));
    return x;
#else
    x = fmaxf(x, __shfl_xor(x, 16, 64));
    return fmaxf(x, __shfl_xor(x, 32, 64));
#endif
}


}  // namespace tmac
