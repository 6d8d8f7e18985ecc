// tmac_gemm.hip — k_gemm_onehot: qgemm_lut for N > 1 activation rows (prefill) on the matrix cores.
//
// The reference handles N > 1 by looping its GEMV micro-kernel over the activation rows
// (python/t_mac/ops/qgemm.py:183-190,228-231): per (row, act group) the integer partial sum
//     PS[n][r][kk] = sum_t QLUT[n][t][nibble(r, t)]
// is a gather.  It becomes a dense int8 contraction — and therefore MFMA work — by writing the gather as
//     PS = onehot(nibble(r, t)) [16 bit-plane rows x (4 tables x 16 entries)]  x  QLUT[n][t][:] [(4 x 16) x 16 rows n]
// i.e. one v_mfma_i32_16x16x64_i8 adds 4 tables for a 16 x 16 (row, n) tile, exactly (int32 accumulate):
//   A operand, lane (g, i): the 16 one-hot bytes of nibble(row i, table 4*tb + g)        (built on the VALU)
//   B operand, lane (g, j): QLUT[n_j][4*tb + g][0..15] — one 16-byte row of the reference-layout QLUT
// Four MFMAs complete an act group (64 activations); the int32 tile is then scaled in fp32 exactly like the
// GEMV epilogue (tbl.cc:464-526 per act group) and accumulated; bit-planes are combined in-lane at the end
// (the 16 rows of a tile are ordered [output row][plane], so a lane's 4 accumulator rows are planes of the
// same output rows).  Same integer contract as the GEMV kernels: PS is bit-exact.
//
// Tiling (v1): wave = 64 bit-plane rows x 32 activation rows (4 x 2 MFMA tiles), workgroup = 4 waves along n.
// Weights are read straight from the QUAD layout (one dword gather per lane and table step, L1/L2 resident:
// 2-4 bit weights are tiny next to the 16x inflated one-hot operand), B rows from the workspace QLUT (L2).
// Roofline: int8 MFMA.  ops = 2 * (Mw*bits) * (K/4*16) * N.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "tmac_core.h"
#include "tmac_kernels.h"

namespace tmac {

typedef int gv4i_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float g_alpha(int p) { return p == 0 ? 0.5f : (p == 1 ? 1.0f : (p == 2 ? 2.0f : 4.0f)); }
__device__ __forceinline__ float g_ld(const void* p, int f16, size_t i) {
    return f16 ? __half2float(reinterpret_cast<const __half*>(p)[i]) : reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void g_st(void* C, int f16, size_t i, float v) {
    if (f16) reinterpret_cast<__half*>(C)[i] = __float2half_rn(v);
    else reinterpret_cast<float*>(C)[i] = v;
}

constexpr int GRT = 4;   // MFMA row tiles per wave (16 bit-plane rows each)
constexpr int GNT = 2;   // MFMA n tiles per wave (16 activation rows each)

template <int BITS, bool ZP>
__global__ __launch_bounds__(256) void k_gemm_onehot(GemmArgs a) {
    const Shape& s = a.s;
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int g = lane >> 4, i16 = lane & 15;
    const int T = s.K / 4, G = s.K / s.ags, nst = (s.K / 32 + 63) >> 6;
    constexpr int NJ = 8 * BITS / 8;
    constexpr int ORPT = 16 / BITS;                        // output rows per MFMA row tile
    const int orow_blk = blockIdx.x * (GRT * ORPT);        // first output row of this workgroup's 64 bit-plane rows
    const int n0 = (blockIdx.y * 4 + w) * (GNT * 16);      // first activation row of this wave

    // ---- per-lane constants of the A gather: lane (g, i16) looks up row (o, p) of tile rt at table 4*tb + g ----
    const int p_a = i16 % BITS;
    int quad_a[GRT], beta_a[GRT];
#pragma unroll
    for (int rt = 0; rt < GRT; ++rt) {
        const int o = orow_blk + rt * ORPT + i16 / BITS;
        quad_a[rt] = o >> 2;
        beta_a[rt] = o & 3;
    }
    const uint32_t* W32 = reinterpret_cast<const uint32_t*>(a.W);

    gv4i_t c[GRT][GNT];
    float facc[GRT][GNT][4];
#pragma unroll
    for (int rt = 0; rt < GRT; ++rt)
#pragma unroll
        for (int nt = 0; nt < GNT; ++nt) {
            c[rt][nt] = (gv4i_t){0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < 4; ++r) facc[rt][nt][r] = 0.f;
        }

    const int tpg = s.ags / 4;                              // tables per act group (16)
    for (int tb = 0; tb < T / 4; ++tb) {
        const int t = 4 * tb + g;
        // B rows: QLUT[n][t][0..15]
        gv4i_t b[GNT];
#pragma unroll
        for (int nt = 0; nt < GNT; ++nt) {
            const int n = n0 + nt * 16 + i16;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (n < a.N) v = *reinterpret_cast<const uint4*>(a.qlut_ref + ((size_t)n * T + t) * 16);
            b[nt] = (gv4i_t){(int)v.x, (int)v.y, (int)v.z, (int)v.w};
        }
        // weight nibble position in the QUAD layout (tmac_layout.h): unit u = t/8, table tl = t%8 of the unit,
        // nibble quad q = tl*BITS + p -> dword d = q/2, half h = q%2
        const int u = t >> 3, tl = t & 7, q = tl * BITS + p_a, d = q >> 1, hsh = 4 * (q & 1);
        const uint32_t dw_off = ((uint32_t)((u >> 6) * NJ + (d >> 2)) * 64 + (u & 63)) * 4 + (d & 3);
#pragma unroll
        for (int rt = 0; rt < GRT; ++rt) {
            uint32_t code = 0;
            if (4 * quad_a[rt] < s.Mw) {
                const uint32_t dw = W32[(size_t)quad_a[rt] * nst * NJ * 256 + dw_off];
                code = (dw >> (8 * beta_a[rt] + hsh)) & 15u;
            }
            const uint32_t j = code ^ ((code & 8u) ? 7u : 0u);          // undo the device recode: c -> reference nibble
            const uint32_t one = 1u << (8 * (j & 3));
            const uint32_t jd = j >> 2;
            const gv4i_t av = {(int)(jd == 0 ? one : 0u), (int)(jd == 1 ? one : 0u), (int)(jd == 2 ? one : 0u), (int)(jd == 3 ? one : 0u)};
#pragma unroll
            for (int nt = 0; nt < GNT; ++nt) c[rt][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, b[nt], c[rt][nt], 0, 0, 0);
        }
        // ---- act group complete: fp32 scale-apply of the int32 tiles, then reset them ----------------------
        if ((4 * tb + 4) % tpg == 0) {
            const int kk = (4 * tb) / tpg;
            const int sg = (kk * s.ags) / s.gs;
#pragma unroll
            for (int nt = 0; nt < GNT; ++nt) {
                const int n = n0 + nt * 16 + i16;                           // C layout: col = lane & 15
                float ls = 0.f, lb = 0.f;
                if (n < a.N) { ls = a.lut_scales[(size_t)n * G + kk]; lb = a.lut_biases[(size_t)n * G + kk]; }
#pragma unroll
                for (int rt = 0; rt < GRT; ++rt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 4 * g + r;                              // C layout: row = 4*(lane>>4) + reg
                        const int pl = i % BITS, o = orow_blk + rt * ORPT + i / BITS;
                        const int32_t ps = c[rt][nt][r];
                        if (a.dump && n < a.N && o < s.Mw) a.dump[((size_t)n * s.M() + mrow(o, pl, BITS)) * G + kk] = ps;
                        float sc = 0.f, zr = 0.f;
                        if (o < s.Mw) {
                            const size_t si = quad_scale_index(s, o >> 2, sg, o & 3, 0);
                            sc = g_ld(a.SC, a.sc_f16, si);
                            if (ZP) zr = g_ld(a.SC, a.sc_f16, si + 1);
                        }
                        const float v = (pl == 0) ? __fmaf_rn((float)ps, ls, lb) : __fmul_rn((float)ps, ls);
                        float acc = __fmaf_rn(v, sc, facc[rt][nt][r]);
                        if (ZP && pl == 0) acc = __fmaf_rn(zr, __fmul_rn(2.0f, lb), acc);
                        facc[rt][nt][r] = acc;
                    }
                    c[rt][nt] = (gv4i_t){0, 0, 0, 0};
                }
            }
        }
    }

    // ---- bit-plane combine (in-lane: a lane's 4 rows are consecutive [output row][plane] rows) and store ----
#pragma unroll
    for (int nt = 0; nt < GNT; ++nt) {
        const int n = n0 + nt * 16 + i16;
        if (n >= a.N) continue;
#pragma unroll
        for (int rt = 0; rt < GRT; ++rt) {
#pragma unroll
            for (int oo = 0; oo < 4 / BITS; ++oo) {
                const int o = orow_blk + rt * ORPT + (4 * g) / BITS + oo;
                float acc = __fmul_rn(facc[rt][nt][oo * BITS], 0.5f);
#pragma unroll
                for (int pl = 1; pl < BITS; ++pl) acc = __fadd_rn(acc, __fmul_rn(facc[rt][nt][oo * BITS + pl], g_alpha(pl)));
                if (o < s.Mw) g_st(a.C, a.out_f16, (size_t)n * s.Mw + o, acc);
            }
        }
    }
}

bool gemm_onehot_supported(const Shape& s) {
    return s.lay == 2 && (s.bits == 2 || s.bits == 4) && s.m_groups < 0 && s.ags == 64 && s.gs % 64 == 0 &&
           s.K % 64 == 0;
}

hipError_t launch_gemm_onehot(const GemmArgs& a, hipStream_t st) {
    if (!gemm_onehot_supported(a.s)) return hipErrorInvalidValue;
    const int bits = a.s.bits;
    const int rows_per_wg = GRT * 16 / bits;
    dim3 g((a.s.Mw + rows_per_wg - 1) / rows_per_wg, (a.N + 4 * GNT * 16 - 1) / (4 * GNT * 16)), b(256);
#define GL(B, Z) hipLaunchKernelGGL((k_gemm_onehot<B, Z>), g, b, 0, st, a)
    if (bits == 2) { if (a.s.zero_point) GL(2, true); else GL(2, false); }
    else { if (a.s.zero_point) GL(4, true); else GL(4, false); }
#undef GL
    return hipGetLastError();
}

}  // namespace tmac
