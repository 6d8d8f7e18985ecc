// tmac_layout.h — index math shared by the HIP kernels, the host dispatcher and the CPU
// emulation test.  Plain C++ (no HIP types) so g++ can compile it too.
//
// Reference layout (what the C-ABI receives; python/t_mac/weights.py:57-87):
//   A  uint8 [M/bm][K/4/kfactor][bm/32][kfactor][16]; byte lane = rr%16, low nibble rows 0-15
//      of a 32-row block, high nibble rows 16-31; M-space row r = (o/8)*8*bits + p*8 + o%8.
//   nibble j of (row, table t): bit ig <-> weight bit-plane p of K index 4t+ig.
//
// Device ("LO" = lane-owns-segment) layout, designed for 16-byte coalesced streaming on gfx950:
//   * K is cut into SEGMENTS of TS=16 tables (64 activations; == one act group when ags == 64).
//   * output rows are cut into ROW QUADS (4 consecutive output rows); RL=4 quads = 16 rows form
//     the row block one workgroup owns.
//   * a thread owns (row quad, segment): NJ = TS*bits/8 uint4 of weights.  Inside them the unit is
//     the NIBBLE QUAD q = tl*bits + p (table tl of the segment, bit-plane p): 4 nibbles, one per
//     row of the quad, stored in byte beta's low (q even) or high (q odd) nibble of dword q/2.
//   * uint4 index = (((b*NSB + sb)*NJ + j)*RL + rl)*KL + kl,  rq = b*RL+rl, segment = sb*KL+kl,
//     so wave-instruction j of a (row block, segment block) reads 1 KiB contiguous.
//   * each nibble is recoded  j -> c = (j<8 ? j : 8|(15-j))  : low 3 bits index the 8-entry
//     half table, bit 3 says "negate" (QLUT[15-j] == -QLUT[j], lut_ctor.cc:152-155).
//   Both steps are bijections on nibbles, so integer partial sums are unchanged.
//
// Device LUT layout: per activation row, half tables biased to unsigned: U[t][i] = QLUT[t][i]+128,
//   i<8, 8 bytes per table; uint4 index = ((sb*8 + j8)*KL + kl) holds tables 2*j8, 2*j8+1 of
//   segment sb*KL+kl.
// Device scale layout: [b][sg][rl][beta][2|1] (scale, zero) in the registered dtype.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define TMAC_HD __host__ __device__ __forceinline__
#else
#define TMAC_HD inline
#endif

namespace tmac {

constexpr int TS = 16;  // tables per segment in the two-kernel ("LO") layout; the fused kernel uses ts = 8
constexpr int KL = 16;  // segment lanes per wave
constexpr int RL = 4;   // row-quad lanes per wave
constexpr int NW = 4;   // waves per workgroup (each takes a different segment block)
constexpr int ROWS_PER_BLOCK = 4 * RL;

struct Shape {
    int Mw, K, bits;
    int bm, kfactor;  // reference tiling
    int gs;           // weight group size (0 when unified scale)
    int ags;          // act group size
    int zero_point;
    int m_groups;     // -1 or >= 1
    int ts;           // tables per layout unit ("segment"): 16 (two-kernel path) or 8 (fused paths)
    int lay;          // 0: row-block layouts above (ts = 16 "LO" / ts = 8 fused);  2: QUAD layout (ts = 8):
                      //    uint4 index = ((quad*nst64 + st)*NJ + j)*64 + lane, unit = st*64 + lane — the 64 lanes
                      //    of a wave hold 64 consecutive 8-table units of ONE row quad (k_gemv_quad)
    // derived
    TMAC_HD int M() const { return Mw * bits; }
    TMAC_HD int nseg() const { return K / (4 * ts); }
    TMAC_HD int nsb() const { return (nseg() + KL - 1) / KL; }
    TMAC_HD int nb() const { return (Mw + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK; }
    TMAC_HD int nj() const { return ts * bits / 8; }  // uint4 per (row quad, segment)
    TMAC_HD int ngroups() const { return K / ags; }
    TMAC_HD int nsg() const { return gs > 0 ? K / gs : 1; }
    TMAC_HD int nquads() const { return (Mw + 3) / 4; }
    TMAC_HD int nst64() const { return (K / 32 + 63) / 64; }
    TMAC_HD size_t weight_u4() const {
        return lay == 2 ? (size_t)nquads() * nst64() * nj() * 64 : (size_t)nb() * nsb() * nj() * RL * KL;
    }
    TMAC_HD size_t scale_elems() const {
        return (lay == 2 ? (size_t)nquads() : (size_t)nb() * RL) * nsg() * 4 * (zero_point ? 2 : 1);
    }
    TMAC_HD size_t qlut_dev_u4() const { return (size_t)((K / (4 * TS) + KL - 1) / KL) * 8 * KL; }  // per activation row (TS=16 layout)
};

// M-space row of (output row o, plane p)  — weights.py:65
TMAC_HD int mrow(int o, int p, int bits) { return (o / 8) * 8 * bits + p * 8 + (o % 8); }

// nibble of M-space row r at table t in the reference blob — SURVEY.md A.3
TMAC_HD int ref_nibble(const uint8_t* A, int K, int bm, int kfactor, int r, int t) {
    const int tile = r / bm, rr = r % bm;
    const size_t byte = (size_t)tile * ((size_t)(bm / 2) * (K / 4)) +
                        ((size_t)(t / kfactor) * (bm / 32) + rr / 32) * kfactor * 16 +
                        (size_t)(t % kfactor) * 16 + (rr % 16);
    return (A[byte] >> (4 * ((rr % 32) / 16))) & 15;
}

TMAC_HD uint32_t recode_nibble(uint32_t j) { return j < 8 ? j : (8u | (15u - j)); }

// One dword of the device weight layout: position (u4 index, e = dword within the uint4).
TMAC_HD uint32_t retile_dword(const uint8_t* A_ref, const Shape& s, size_t u4, int e) {
    int rq, seg, j;
    if (s.lay == 2) {
        const int lane = (int)(u4 % 64); size_t x = u4 / 64;
        j = (int)(x % s.nj()); x /= s.nj();
        const int st = (int)(x % s.nst64());
        rq = (int)(x / s.nst64());
        seg = st * 64 + lane;
    } else {
        const int kl = (int)(u4 % KL); size_t x = u4 / KL;
        const int rl = (int)(x % RL); x /= RL;
        j = (int)(x % s.nj()); x /= s.nj();
        const int sb = (int)(x % s.nsb());
        const int b = (int)(x / s.nsb());
        rq = b * RL + rl; seg = sb * KL + kl;
    }
    const int d = 4 * j + e;
    if (seg >= s.nseg()) return 0;
    uint32_t out = 0;
    for (int h = 0; h < 2; ++h) {
        const int q = 2 * d + h, tl = q / s.bits, p = q % s.bits;
        const int t = seg * s.ts + tl;
        for (int beta = 0; beta < 4; ++beta) {
            const int o = 4 * rq + beta;
            if (o >= s.Mw) continue;
            const uint32_t c = recode_nibble((uint32_t)ref_nibble(A_ref, s.K, s.bm, s.kfactor, mrow(o, p, s.bits), t));
            out |= c << (8 * beta + 4 * h);
        }
    }
    return out;
}

// Reference scale blob element for (output row o, scale group sg, which: 0 scale / 1 zero)
TMAC_HD size_t ref_scale_index(const Shape& s, int o, int sg, int which) {
    const int rpt = s.bm / s.bits, tile = o / rpt, m = o % rpt;
    const int per = s.zero_point ? 2 : 1;
    return ((size_t)tile * s.nsg() + sg) * rpt * per + (size_t)(m / 8) * 8 * per + which * 8 + (m % 8);
}

// device scale element index for (row block b, scale group sg, rl, beta, which)
TMAC_HD size_t dev_scale_index(const Shape& s, int b, int sg, int rl, int beta, int which) {
    const int per = s.zero_point ? 2 : 1;
    return ((((size_t)b * s.nsg() + sg) * RL + rl) * 4 + beta) * per + which;
}

// QUAD layout: scale element index for (row quad, scale group sg, beta, which); weights uint4 index
TMAC_HD size_t quad_scale_index(const Shape& s, int quad, int sg, int beta, int which) {
    const int per = s.zero_point ? 2 : 1;
    return (((size_t)quad * s.nsg() + sg) * 4 + beta) * per + which;
}
TMAC_HD size_t quad_weight_u4_index(const Shape& s, int quad, int st, int j, int lane) {
    return (((size_t)quad * s.nst64() + st) * s.nj() + j) * 64 + lane;
}

// uint4 index of weights for (row block b, segment block sb, j, rl, kl)
TMAC_HD size_t weight_u4_index(const Shape& s, int b, int sb, int j, int rl, int kl) {
    return ((((size_t)b * s.nsb() + sb) * s.nj() + j) * RL + rl) * KL + kl;
}

// uint4 index (within one activation row's LUT) of tables 2*j8, 2*j8+1 of segment seg
TMAC_HD size_t qlut_dev_u4_index(int seg, int j8) { return ((size_t)(seg / KL) * 8 + j8) * KL + (seg % KL); }

}  // namespace tmac
