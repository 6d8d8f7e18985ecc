// tmac_core.h — the per-thread arithmetic of the GEMV kernels, written so the same source
// runs inside the HIP kernels (v_perm_b32 / v_mqsad_pk_u16_u8) and, for the CPU emulation test
// (tests/test_emulation.py -> csrc/emu_test.cpp), as plain C++ with the two instructions
// modelled bit-for-bit.  No floating point here: this file is the integer (bit-exact) contract.
//
// Replaces python/t_mac/intrins/tbl.cc:445-462 (AVX2: _mm256_shuffle_epi8 on nibble indices +
// SignedWideningAdder).  How the lookup maps to gfx950:
//   * a 16-entry int8 table does not fit one v_perm_b32 (8 source bytes), but the table is exactly
//     antisymmetric (QLUT[15-j] == -QLUT[j], lut_ctor.cc:152-155), so 8 biased entries
//     U[i] = QLUT[i] + 128 are enough:      rP = perm({U[4..7], U[0..3]}, idx)        4 lookups
//     negated candidates, bytewise 256-U:   rN = 0x01010100 - rP   (no cross-byte borrow since
//                                                                    U in [1,255])
//     per-byte choice by the sign bit:      r  = perm({rN, rP}, beta + 4*sign)
//     r's bytes are V = value + 128 for the 4 rows of the quad.
//   * accumulate: v_mqsad_pk_u16_u8 with reference 0x000000ff adds (255 - V_beta) into four
//     packed u16 accumulators in ONE instruction (masked SAD: only byte 0 of each sliding window
//     is unmasked).  After T tables: sum(value) = 127*T - acc16.
#pragma once
#include <stdint.h>
#include "tmac_layout.h"

namespace tmac {

// v_perm_b32 / v_mqsad_pk_u16_u8: the instruction on the device, a bit-exact model on the host
// (CDNA3/4 ISA; the models are verified against the hardware by tests/test_gpu_parity.py::test_isa_models).
TMAC_HD uint32_t perm_b32(uint32_t s0, uint32_t s1, uint32_t sel) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(s0, s1, sel);
#else
    const uint64_t src = ((uint64_t)s0 << 32) | s1;
    uint32_t out = 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t c = (sel >> (8 * i)) & 0xff;
        uint32_t byte;
        if (c <= 7) byte = (uint32_t)(src >> (8 * c)) & 0xff;
        else if (c == 8) byte = ((s1 >> 15) & 1) ? 0xff : 0;
        else if (c == 9) byte = ((s1 >> 31) & 1) ? 0xff : 0;
        else if (c == 10) byte = ((s0 >> 15) & 1) ? 0xff : 0;
        else if (c == 11) byte = ((s0 >> 31) & 1) ? 0xff : 0;
        else if (c == 12) byte = 0;
        else byte = 0xff;
        out |= byte << (8 * i);
    }
    return out;
#endif
}
TMAC_HD uint64_t mqsad_acc(uint32_t r, uint64_t acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    // only byte 0 of each sliding window is unmasked, so the high dword of src0 is never read: leave it
    // undefined instead of zero-extending (saves one v_mov_b32 per accumulate)
    uint32_t hi;
    asm volatile("" : "=v"(hi));   // "defines" hi without emitting an instruction
    return __builtin_amdgcn_mqsad_pk_u16_u8(((uint64_t)hi << 32) | r, 0xffu, acc);
#else
    uint64_t out = 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t v = (r >> (8 * i)) & 0xff;   // window i, byte 0 (reference byte 0xff)
        const uint32_t a = (uint32_t)(acc >> (16 * i)) & 0xffff;
        uint32_t s = a + (255u - v);
        if (s > 0xffff) s = 0xffff;                 // the instruction saturates
        out |= (uint64_t)s << (16 * i);
    }
    return out;
#endif
}

// v_lerp_u8 with rounding bits set: per byte (a + b + 1) >> 1, the unsigned rounding average
TMAC_HD uint32_t avg_u8x4(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_lerp(a, b, 0x01010101u);
#else
    uint32_t out = 0;
    for (int i = 0; i < 4; ++i) out |= ((((a >> (8 * i)) & 0xff) + ((b >> (8 * i)) & 0xff) + 1u) >> 1) << (8 * i);
    return out;
#endif
}

// 4 signed lookups (biased by +128) for the low (h=0) or high (h=1) nibbles of weight dword w.
// tab = {U[0..3], U[4..7]} of the table these nibbles index.
template <int H>
TMAC_HD uint32_t lookup4(uint32_t w, uint32_t tab_lo, uint32_t tab_hi) {
    const uint32_t x = H ? (w >> 4) : w;
    const uint32_t sel = x & 0x07070707u;
    const uint32_t rP = perm_b32(tab_hi, tab_lo, sel);
    const uint32_t rN = 0x01010100u - rP;
    const uint32_t sel3 = ((x >> 1) & 0x04040404u) | 0x03020100u;
    return perm_b32(rN, rP, sel3);
}

// MFMA-accumulate flavour (signed tables, no bias): the sign bit routes each looked-up byte either to
// the "plus" word or to the "minus" word (the other gets 0), and v_mfma_i32_16x16x64_i8 against a
// constant +1/-1 selector matrix adds plus - minus of 2 tables x 4 rows per lane in one instruction on
// the matrix pipe (see k_gemv_fused, ACC = 1).  tab = {Q[0..3], Q[4..7]} as signed bytes.
template <int H>
TMAC_HD void lookup4_pm(uint32_t w, uint32_t tab_lo, uint32_t tab_hi, uint32_t& plus, uint32_t& minus) {
    const uint32_t x = H ? (w >> 4) : w;
    const uint32_t sel = x & 0x07070707u;
    const uint32_t rP = perm_b32(tab_hi, tab_lo, sel);
    const uint32_t sel3 = ((x >> 1) & 0x04040404u) | 0x03020100u;   // byte beta: beta (keep) or 4+beta (negate)
    plus = perm_b32(0u, rP, sel3);    // selectors 0-3 -> rP, 4-7 -> 0
    minus = perm_b32(rP, 0u, sel3);   // selectors 0-3 -> 0,  4-7 -> rP
}

// Accumulators for one (row quad, act group): per bit-plane, four row sums.
//   MODE 0: one packed-u16 quad per plane, fed by v_mqsad_pk_u16_u8 (1 instruction / 4 lookups)
//   MODE 1: four int32 per plane, fed by byte-select adds (v_add_u32_sdwa, 4 instructions)
template <int BITS, int MODE>
struct SegAcc;

template <int BITS>
struct SegAcc<BITS, 0> {
    uint64_t a[BITS];
    TMAC_HD void reset() {
#pragma unroll
        for (int p = 0; p < BITS; ++p) a[p] = 0;
    }
    TMAC_HD void add(int p, uint32_t r, int = 0) { a[p] = mqsad_acc(r, a[p]); }
    // integer partial sum of (plane p, row beta) after `tables` lookups
    TMAC_HD int32_t ps(int p, int beta, int tables) const {
        return 127 * tables - (int32_t)((a[p] >> (16 * beta)) & 0xffff);
    }
};

template <int BITS>
struct SegAcc<BITS, 1> {
    uint32_t a[BITS][4];
    TMAC_HD void reset() {
#pragma unroll
        for (int p = 0; p < BITS; ++p)
#pragma unroll
            for (int b = 0; b < 4; ++b) a[p][b] = 0;
    }
    TMAC_HD void add(int p, uint32_t r, int = 0) {
        a[p][0] += r & 0xff;
        a[p][1] += (r >> 8) & 0xff;
        a[p][2] += (r >> 16) & 0xff;
        a[p][3] += r >> 24;
    }
    TMAC_HD int32_t ps(int p, int beta, int tables) const { return (int32_t)a[p][beta] - 128 * tables; }
};

// MODE 2: fast aggregation (a9; tbl.cc:86-141 NEON / :201-256 AVX2).  The act group's lookups are folded by a
// balanced tree of rounding-halving adds in table order instead of being summed: push k merges, binary-counter
// fashion, every completed pair of equal-sized subtrees.  The bytes arrive biased (V = value + 128), and
//   signed rounding-halving add (vrhaddq_s8):  ((a + b + 1) >> 1) + 128 == (Va + Vb + 1) >> 1      -> xr = 0
//   the AVX2 flavour (_mm256_avg_epu8 on the raw signed bytes): average V ^ 0x80 instead           -> xr = 0x80808080
// both on v_lerp_u8.  ps() is the tree result as a signed byte (it stands for sum / tables).
template <int BITS>
struct SegAcc<BITS, 2> {
    uint32_t lvl[BITS][4];
    uint32_t res[BITS];
    uint32_t xr;
    TMAC_HD void reset() {}
    TMAC_HD void add(int p, uint32_t r, int k) {
        uint32_t cur = r ^ xr;
        int l = 0;
#pragma unroll
        for (; l < 4; ++l) {
            if (!((k >> l) & 1)) break;
            cur = avg_u8x4(lvl[p][l], cur);
        }
        if (l < 4) lvl[p][l] = cur;
        res[p] = cur;
    }
    TMAC_HD int32_t ps(int p, int beta, int) const {
        return (int32_t)(((res[p] ^ xr) >> (8 * beta)) & 0xff) - 128;
    }
};

TMAC_HD int fa_bias_factor(int tables, int bits) {   // mylog2<ActK>::value / 4 * get_bias_scale(Bits), tbl.cc:287-318
    int l = -1;
    for (int k = tables; k; k /= 2) ++l;
    return l / 4 * ((1 << bits) - 1);
}

// Consume tables [TL0, TL0+NT) of a segment.  wd: the thread's TS*BITS/2 weight dwords of this
// segment; tb: its TS half tables as (lo,hi) dword pairs.  Everything is compile-time indexed.
template <int BITS, int TL0, int NT, typename Acc, typename WD, typename TB>
TMAC_HD void accumulate_tables(const WD& wd, const TB& tb, Acc& acc) {
#pragma unroll
    for (int q = TL0 * BITS; q < (TL0 + NT) * BITS; ++q) {
        const int tl = q / BITS, p = q % BITS, d = q >> 1;
        const uint32_t r = (q & 1) ? lookup4<1>(wd[d], tb[2 * tl], tb[2 * tl + 1])
                                   : lookup4<0>(wd[d], tb[2 * tl], tb[2 * tl + 1]);
        acc.add(p, r, tl - TL0);
    }
}

}  // namespace tmac
