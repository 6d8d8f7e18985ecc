// tmac_gemm.hip — k_gemm_onehot: qgemm_lut for N > 1 activation rows (prefill) on the matrix cores.
//
// The reference handles N > 1 by looping its GEMV micro-kernel over the activation rows
// (python/t_mac/ops/qgemm.py:183-190,228-231): per (activation row n, bit-plane row r, act group kk) the integer
// partial sum
//     PS[n][r][kk] = sum_t QLUT[n][t][nibble(r, t)]
// is a gather.  It becomes a dense int8 contraction — MFMA work — by writing the gather as a product with a signed
// one-hot matrix.  With the antisymmetric tables (QLUT[15-j] = -QLUT[j]) only the 8-entry half table is needed:
//     PS = S [rows x (tables x 8)]  x  H [(tables x 8) x n],   S[r][(t, e)] = +-1 if the recoded nibble of (r, t)
// selects entry e (sign = its bit 3), 0 otherwise;  H = the half tables k_preprocess already writes for the GEMV
// kernels (workspace qlut_lds image: one uint4 = the half tables of two consecutive tables).
// One v_mfma_i32_16x16x64_i8 covers 8 tables (one 32-activation "unit" of the weight layout) for a 16 x 16 tile:
//   A operand, lane (g, i): 16 bytes = one-hot rows of tables 2g, 2g+1 of the unit for bit-plane row i
//   B operand, lane (g, j): 16 bytes = half tables 2g, 2g+1 of the unit for activation row j   (one uint4 load)
// Two MFMAs complete an act group (64 activations); the int32 tile is then scaled in fp32 exactly like the GEMV
// epilogue (tbl.cc:464-526 per act group) and accumulated; bit-planes are combined in-lane at the end (the 16 rows
// of a tile are ordered [output row][plane], so the 4 accumulator rows of a lane are planes of its own output
// rows).  Same integer contract as the GEMV kernels: PS is bit-exact with the reference.
//
// Tiling: workgroup = 4 waves = 128 bit-plane rows x 64 activation rows; wave = 32 rows x 64 columns (2 x 4 MFMA
// tiles, 8 MFMAs per unit).  Everything the inner loop consumes is staged through LDS in chunks of 4 units (two act
// groups), double buffered and fetched one chunk ahead with ~22 registers per thread:
//   B: every thread fetches the 64 contiguous bytes [n][j4][u0..u0+3] of the half-table image and stores them
//      unit-major, so a wave's operand read is one conflict-free 1 KB ds_read_b128 (4 KB per unit and workgroup);
//   weights: the workgroup's 16 (W2) / 8 (W4) row quads x 4 units of the QUAD layout, 2 KB, one uint4 per thread;
//   epilogue operands: LUT scale/bias of the 64 columns and weight scale/zero of the rows for the two act groups.
// The signed one-hot bytes are formed on the VALU from the nibble (8 instructions per table).
// Roofline: int8 MFMA.  ops = 2 * (Mw*bits) * (K/4*8) * N.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "tmac_core.h"
#include "tmac_kernels.h"

namespace tmac {

typedef int gv4i_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float g_alpha(int p) { return p == 0 ? 0.5f : (p == 1 ? 1.0f : (p == 2 ? 2.0f : 4.0f)); }
__device__ __forceinline__ void g_st(void* C, int f16, size_t i, float v) {
    if (f16) reinterpret_cast<__half*>(C)[i] = __float2half_rn(v);
    else reinterpret_cast<float*>(C)[i] = v;
}

constexpr int GRT = 2;   // MFMA row tiles per wave (16 bit-plane rows each)
constexpr int GWV = 4;   // waves per workgroup (consecutive row blocks)
constexpr bool W4COMB = true;   // W4: integer plane combine + one scale chain per output (false: the per-plane packed chains, A/B)

typedef float gv2f_t __attribute__((ext_vector_type(2)));

// GNT: MFMA n tiles per workgroup and wave (16 activation rows each): 4, or 2 when that is needed to fill the chip.
// GCH: units (8 tables = 32 activations each) per LDS chunk: 4, or 8 with the narrow tile (longer chunks hide the
// fetch latency when a CU holds few waves; 48 KB of LDS).
// US: unified-scale flavour (BitNet: m_groups >= 1, act group = K; tbl_g4_int8_int32_update + qgemm.py:170-174): the int32
// tiles accumulate over the whole K, nothing is scaled per act group, and the epilogue is
//   C = ((sum_p float(cb_p) alpha_p) * lut_scale[n] + lut_bias[n] / 2) * scale[o / (Mw / m_groups)]      bit-exact
template <int BITS, bool ZP, bool DUMP, int GNT, int GCH, bool US>
__global__ __launch_bounds__(64 * GWV) void k_gemm_onehot(GemmArgs a) {
    constexpr int NJ = BITS;                               // uint4 per unit and quad in the QUAD layout
    constexpr int ORPT = 16 / BITS;                        // output rows per MFMA row tile
    constexpr int ORW = GWV * GRT * ORPT;                  // output rows per workgroup (64 / 32)
    constexpr int QW = ORW / 4;                            // row quads per workgroup
    __shared__ uint4 bt[2][GCH][GNT * 16][4];              // [buffer][unit][n][j4]            2 x 16 KB
    __shared__ uint4 wt[2][GCH][QW][NJ];                   // [buffer][unit][quad][j]          2 x 2 KB
    constexpr int NAG = GCH / 2;                           // act groups (64 activations) per chunk
    __shared__ float ep[2][NAG][4][64];                    // [buffer][act group][ls, lb, sc, zr][n | row]  (n < GNT*16)
    __shared__ uint2 pat[16];                              // signed one-hot row of a recoded nibble c: byte (c & 7) = +1, or -1 if c & 8
    const Shape& s = a.s;
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int g = lane >> 4, i16 = lane & 15;
    // several matrices that share K, the quantisation config and the LUT (q/k/v, gate/up) in one launch: the workgroup
    // index selects the matrix (uniform), the rest of the kernel sees one matrix
    int mi = 0, bx = blockIdx.x;
    while (mi + 1 < a.nmat && bx >= a.m[mi].wg_end) ++mi;
    if (mi > 0) bx -= a.m[mi - 1].wg_end;
    const GemmMat M = a.m[mi];
    const int Mw = M.Mw;
    const int G = s.K / s.ags, nu = s.K / 32, nst = (nu + 63) >> 6, nq = (Mw + 3) >> 2;
    const int orow_wg = bx * ORW;                          // first output row of this workgroup
    const int n0 = blockIdx.y * (GNT * 16);                // first activation row of this workgroup
    const int nchunk = (nu + GCH - 1) / GCH;

    if (tid < 16) {
        const uint32_t v = ((tid & 8) ? 0xffu : 0x01u) << (8 * (tid & 3));
        pat[tid] = (tid & 4) ? make_uint2(0u, v) : make_uint2(v, 0u);
    }
    // ---- staging roles of this thread -----------------------------------------------------------------------
    // B: activation row n0 + tid/4, j4 = tid%4, the chunk's GCH consecutive units
    const uint4* bsrc = reinterpret_cast<const uint4*>(a.qlut_lds) + ((size_t)min(n0 + (tid >> 2), a.N - 1) * 4 + (tid & 3)) * a.tstride;
    // weights (tid < GCH*QW*NJ = 128 | 256): unit tid / (QW*NJ), quad (tid / NJ) % QW, uint4 j = tid % NJ
    const int w_ul = tid / (QW * NJ), w_ql = (tid / NJ) % QW, w_j = tid % NJ;
    const uint4* wsrc = reinterpret_cast<const uint4*>(M.W) + ((size_t)min(orow_wg / 4 + w_ql, nq - 1) * nst * NJ + w_j) * 64;
    // epilogue operands: e1 = ls / lb of column tid%64, e2 = scale / zero of row (tid/2)%64, act groups tid/128 + 2k
    const int e_ag = tid >> 7, e1_which = (tid >> 6) & 1, e1_n = min(n0 + (tid & (GNT * 16 - 1)), a.N - 1);
    const int e2_o = min(orow_wg + ((tid >> 1) & 63), Mw - 1), e2_which = tid & 1;
    const float* e1_src = (e1_which ? a.lut_biases : a.lut_scales) + (size_t)e1_n * G;

    uint4 bst[GCH], wst;
    float e1[NAG / 2], e2[NAG / 2];
    auto fetch_chunk = [&](int c) {
#pragma unroll
        for (int ul = 0; ul < GCH; ++ul)
            if (tid < GNT * 64) bst[ul] = bsrc[min(c * GCH + ul, nu - 1)];   // tail units: clamped here, skipped below
        if (tid < GCH * QW * NJ) {
            const int u = min(c * GCH + w_ul, nu - 1);
            wst = wsrc[(size_t)(u >> 6) * NJ * 64 + (u & 63)];
        }
#pragma unroll
        for (int k = 0; k < NAG / 2; ++k) {
            if (US) continue;                               // no per-act-group operands
            const int kk = min(c * NAG + e_ag + 2 * k, G - 1);
            e1[k] = e1_src[kk];
            e2[k] = 0.f;
            if (ZP || !e2_which) {
                const size_t si = quad_scale_index(s, e2_o >> 2, (kk * s.ags) / s.gs, e2_o & 3, e2_which);
                e2[k] = a.sc_f16 ? __half2float(reinterpret_cast<const __half*>(M.SC)[si]) : reinterpret_cast<const float*>(M.SC)[si];
            }
        }
    };
    auto stage_chunk = [&](int buf) {       // biased -> signed table bytes on the way into LDS
#pragma unroll
        for (int ul = 0; ul < GCH; ++ul)
            if (tid < GNT * 64)
                bt[buf][ul][tid >> 2][tid & 3] = make_uint4(bst[ul].x ^ 0x80808080u, bst[ul].y ^ 0x80808080u,
                                                            bst[ul].z ^ 0x80808080u, bst[ul].w ^ 0x80808080u);
        if (tid < GCH * QW * NJ) wt[buf][w_ul][w_ql][w_j] = wst;
#pragma unroll
        for (int k = 0; k < NAG / 2; ++k) {
            if (US) continue;
            ep[buf][e_ag + 2 * k][e1_which][tid & 63] = e1[k];
            ep[buf][e_ag + 2 * k][2 + e2_which][(tid >> 1) & 63] = e2[k];
        }
    };

    // ---- compute roles ----------------------------------------------------------------------------------------
    // A operand, lane (g, i16): bit-plane row (o, p) of row tile rt; tables 2g, 2g+1 of the unit are nibble quads
    // q = 2g*BITS + p and q + BITS -> dwords q >> 1 (and + BITS/2) of the unit's BITS uint4 (tmac_layout.h)
    // Row order inside a 16-row tile.  W4: [output][plane], the 4 accumulator rows of a lane are the planes of one output.
    // W2: tile row 4q + r is plane r >> 1 of output 2q + (r & 1), so a lane's accumulator rows are (o0 p0, o1 p0, o0 p1,
    // o1 p1): each packed-fp32 register pair holds ONE plane of the lane's two outputs and the planes combine pairwise.
    const int p_a = (BITS == 2) ? ((i16 >> 1) & 1) : (i16 % BITS);
    const int ol_t = (BITS == 2) ? (2 * (i16 >> 2) + (i16 & 1)) : (i16 / BITS);   // output row within the tile
    const int d0 = (2 * g * BITS + p_a) >> 1;              // BITS 2: 2g, 2g+1;  BITS 4: 4g + (p>>1), + 2
    int a_ql[GRT], a_sh[GRT];
#pragma unroll
    for (int rt = 0; rt < GRT; ++rt) {
        const int ol = (w * GRT + rt) * ORPT + ol_t;         // output row within the workgroup
        a_ql[rt] = ol >> 2;
        a_sh[rt] = (8 * (ol & 3) + 4 * (p_a & 1) + 29) & 31; // rotate-right count that puts the nibble (byte beta, half q & 1) at bits 3..6
    }

    gv4i_t c[GRT][GNT];
    // W4: per-plane accumulators, rows (4g, 4g+1), (4g+2, 4g+3) of the tile as v_pk_*_f32 operands; combined at the end.
    // W2: ONE pair per tile = the lane's two outputs, planes already combined (index [1] unused).
    gv2f_t facc[GRT][GNT][2];
#pragma unroll
    for (int rt = 0; rt < GRT; ++rt)
#pragma unroll
        for (int nt = 0; nt < GNT; ++nt) {
            c[rt][nt] = (gv4i_t){0, 0, 0, 0};
            facc[rt][nt][0] = (gv2f_t){0.f, 0.f};
            facc[rt][nt][1] = (gv2f_t){0.f, 0.f};
        }

    // The int32 accumulation starts from 0x4B400000 = bits of 1.5 * 2^23: with |PS| <= 16 * 127 the accumulator's bit
    // pattern IS the float 12582912 + PS, so the int -> float conversion is one exact (packed) subtraction.
    const gv4i_t cinit = {0x4B400000, 0x4B400000, 0x4B400000, 0x4B400000};
    fetch_chunk(0);
    stage_chunk(0);
    for (int ck = 0; ck < nchunk; ++ck) {
        __syncthreads();                                    // chunk ck is in LDS buffer ck & 1; buffer (ck+1) & 1 is free
        if (ck + 1 < nchunk) fetch_chunk(ck + 1);
        const int buf = ck & 1;
        // operands of a unit: B tiles straight from LDS; A = signed one-hot rows, two dependent LDS reads (weight dwords,
        // then the pattern table).  They are formed one unit ahead of the MFMAs that consume them.
        auto operands = [&](int ul, gv4i_t (&av)[GRT], gv4i_t (&bv)[GNT]) {
#pragma unroll
            for (int nt = 0; nt < GNT; ++nt) {
                const uint4 v = bt[buf][ul][nt * 16 + i16][g];
                bv[nt] = (gv4i_t){(int)v.x, (int)v.y, (int)v.z, (int)v.w};
            }
#pragma unroll
            for (int rt = 0; rt < GRT; ++rt) {
                const uint32_t* wq = reinterpret_cast<const uint32_t*>(&wt[buf][ul][a_ql[rt]][0]);
                const uint32_t w0 = wq[d0], w1 = wq[d0 + BITS / 2];
                const char* pb = reinterpret_cast<const char*>(pat);     // byte offset = nibble * 8
                const uint2 p0 = *reinterpret_cast<const uint2*>(pb + (__builtin_amdgcn_alignbit(w0, w0, a_sh[rt]) & 0x78u));
                const uint2 p1 = *reinterpret_cast<const uint2*>(pb + (__builtin_amdgcn_alignbit(w1, w1, a_sh[rt]) & 0x78u));
                av[rt] = (gv4i_t){(int)p0.x, (int)p0.y, (int)p1.x, (int)p1.y};
            }
        };
        // operands one unit ahead of the MFMAs that consume them, where the registers allow: the 32-column tile, and the
        // 64-column tile with W2 (one fp32 accumulator pair per tile since the planes are combined first: 150 VGPRs)
        constexpr bool AHEAD = (GNT == 2) || (BITS == 2) || (BITS == 4 && W4COMB);
        gv4i_t avc[GRT], bvc[GNT], avn[GRT], bvn[GNT];
        if (AHEAD) operands(0, avc, bvc);
#pragma unroll
        for (int ul = 0; ul < GCH; ++ul) {
            if (AHEAD) { if (ul + 1 < GCH) operands(ul + 1, avn, bvn); }
            else operands(ul, avc, bvc);
            if (ck * GCH + ul < nu) {
#pragma unroll
                for (int rt = 0; rt < GRT; ++rt)
#pragma unroll
                    for (int nt = 0; nt < GNT; ++nt)         // first unit of an act group starts from the constant accumulator operand
                        c[rt][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(avc[rt], bvc[nt], (US || (ul & 1)) ? c[rt][nt] : cinit, 0, 0, 0);
                // ---- act group complete: fp32 scale-apply of the int32 tiles, then reset them --------------
                if (!US && (ul & 1)) {                       // ags = 64: units 2kk, 2kk+1
                    const int ag = ul >> 1, kk = (ck * GCH + ul) >> 1;
                    float sc[GRT][4 / BITS], zr[GRT][4 / BITS];
#pragma unroll
                    for (int rt = 0; rt < GRT; ++rt)
#pragma unroll
                        for (int oo = 0; oo < 4 / BITS; ++oo) {
                            const int ol = (w * GRT + rt) * ORPT + (4 * g) / BITS + oo;
                            sc[rt][oo] = ep[buf][ag][2][ol];
                            zr[rt][oo] = ep[buf][ag][3][ol];
                        }
#pragma unroll
                    for (int nt = 0; nt < GNT; ++nt) {
                        const float ls = ep[buf][ag][0][nt * 16 + i16], lb = ep[buf][ag][1][nt * 16 + i16];
                        const float lb2 = __fmul_rn(2.0f, lb), hlb = __fmul_rn(0.5f, lb), hls = __fmul_rn(0.5f, ls);
#pragma unroll
                        for (int rt = 0; rt < GRT; ++rt) {
                            if (DUMP) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int o = orow_wg + (w * GRT + rt) * ORPT + (BITS == 2 ? 2 * g + (r & 1) : (4 * g + r) / BITS), nn = n0 + nt * 16 + i16;
                                    const int pl = (BITS == 2) ? (r >> 1) : (r % BITS);
                                    if (nn < a.N && o < Mw) a.dump[((size_t)nn * Mw * BITS + mrow(o, pl, BITS)) * G + kk] = c[rt][nt][r] - 0x4B400000;
                                }
                            }
                            // rows 4g + 2pr, 4g + 2pr + 1 = planes (2pr) % BITS and the next one (never plane 0) of output
                            // row oo = 2pr / BITS.  tbl.cc: plane 0 takes v = fma(ps, ls, lb), the others v = ps * ls
                            // (= fma(ps, ls, +0) up to the sign of a zero), then acc = fma(v, scale, acc) and, plane 0 with
                            // zero points, acc = fma(zero, 2 lb, acc) (adding fma(0, 2 lb, acc) = acc to the odd lane).
                            if constexpr (BITS == 2) {
                                // planes first (exact: small integers and halves), then ONE scale chain per output:
                                //   C += ((ps0/2 + ps1) * ls + lb/2) * scale + zero * lb
                                // = alpha0 * [(ps0 ls + lb) scale + zero 2 lb] + alpha1 * [ps1 ls scale] of tbl.cc:479-526 +
                                // kernels.cc:1068 with the roundings of the per-plane chains merged (fp32, <= 1e-3 contract)
                                const gv2f_t m2 = {12582912.0f, 12582912.0f};
                                const gv2f_t ps0 = (gv2f_t){__int_as_float(c[rt][nt][0]), __int_as_float(c[rt][nt][1])} - m2;
                                const gv2f_t ps1 = (gv2f_t){__int_as_float(c[rt][nt][2]), __int_as_float(c[rt][nt][3])} - m2;
                                const gv2f_t psc = __builtin_elementwise_fma(ps0, (gv2f_t){0.5f, 0.5f}, ps1);
                                const gv2f_t v = __builtin_elementwise_fma(psc, (gv2f_t){ls, ls}, (gv2f_t){hlb, hlb});
                                gv2f_t acc = __builtin_elementwise_fma(v, (gv2f_t){sc[rt][0], sc[rt][1]}, facc[rt][nt][0]);
                                if (ZP) acc = __builtin_elementwise_fma((gv2f_t){zr[rt][0], zr[rt][1]}, (gv2f_t){lb, lb}, acc);
                                facc[rt][nt][0] = acc;
                            } else if constexpr (BITS == 4 && W4COMB) {
                                // the lane's four accumulator rows are the planes of ONE output: combine them as integers
                                // (Horner on the accumulator bits, each 0x4B400000 + ps; the 15 offsets leave in one add),
                                // one conversion and one scale chain: sum_p 2^p ps_p * (ls / 2) + lb / 2, as in k_gemv_quad
                                uint32_t h = (uint32_t)c[rt][nt][3];
                                h = (h << 1) + (uint32_t)c[rt][nt][2];
                                h = (h << 1) + (uint32_t)c[rt][nt][1];
                                h = (h << 1) + (uint32_t)c[rt][nt][0];
                                const int32_t comb = (int32_t)(h - 15u * 0x4B400000u);
                                const float v = __fmaf_rn((float)comb, hls, hlb);
                                float acc = __fmaf_rn(v, sc[rt][0], facc[rt][nt][0][0]);
                                if (ZP) acc = __fmaf_rn(zr[rt][0], lb, acc);
                                facc[rt][nt][0][0] = acc;
                            } else
#pragma unroll
                            for (int pr = 0; pr < 2; ++pr) {
                                const int oo = (2 * pr) / BITS;
                                const bool p0 = (2 * pr) % BITS == 0;
                                const gv2f_t ps = (gv2f_t){__int_as_float(c[rt][nt][2 * pr]), __int_as_float(c[rt][nt][2 * pr + 1])} -
                                                  (gv2f_t){12582912.0f, 12582912.0f};
                                const gv2f_t v = __builtin_elementwise_fma(ps, (gv2f_t){ls, ls}, (gv2f_t){p0 ? lb : 0.f, 0.f});
                                gv2f_t acc = __builtin_elementwise_fma(v, (gv2f_t){sc[rt][oo], sc[rt][oo]}, facc[rt][nt][pr]);
                                if (ZP && p0) acc = __builtin_elementwise_fma((gv2f_t){zr[rt][oo], 0.f}, (gv2f_t){lb2, lb2}, acc);
                                facc[rt][nt][pr] = acc;
                            }
                        }
                    }
                }
            }
            if (AHEAD) {
#pragma unroll
                for (int rt = 0; rt < GRT; ++rt) avc[rt] = avn[rt];
#pragma unroll
                for (int nt = 0; nt < GNT; ++nt) bvc[nt] = bvn[nt];
            }
        }
        if (ck + 1 < nchunk) stage_chunk((ck + 1) & 1);     // buffer (ck+1)&1 was last read in iteration ck-1, before this iteration's barrier
    }

    // ---- bit-plane combine (in-lane) and store --------------------------------------------------------------
#pragma unroll
    for (int nt = 0; nt < GNT; ++nt) {
        const int n = n0 + nt * 16 + i16;
        if (n >= a.N) continue;
#pragma unroll
        for (int rt = 0; rt < GRT; ++rt) {
#pragma unroll
            for (int oo = 0; oo < 4 / BITS; ++oo) {
                const int o = orow_wg + (w * GRT + rt) * ORPT + (4 * g) / BITS + oo;
                float acc;
                if constexpr (US) {
                    // accumulator rows of the lane: W2 (o0 p0, o1 p0, o0 p1, o1 p1), W4 planes 0..3 of one output
                    float t = 0.f;
#pragma unroll
                    for (int pl = 0; pl < BITS; ++pl) {
                        const int r = (BITS == 2) ? (2 * pl + oo) : pl;
                        const int32_t cb = c[rt][nt][r];
                        if (DUMP && o < Mw) a.dump[(size_t)n * Mw * BITS + mrow(o, pl, BITS)] = cb;
                        const float tp = __fmul_rn((float)cb, g_alpha(pl));
                        t = (pl == 0) ? tp : __fadd_rn(t, tp);
                    }
                    const float v = __fadd_rn(__fmul_rn(t, a.lut_scales[n]), __fmul_rn(a.lut_biases[n], 0.5f));
                    const int sgi = o < Mw ? o / (Mw / s.m_groups) : 0;
                    const float scv = a.sc_f16 ? __half2float(reinterpret_cast<const __half*>(M.SC)[sgi]) : reinterpret_cast<const float*>(M.SC)[sgi];
                    acc = __fmul_rn(v, scv);
                } else if constexpr (BITS == 2) {
                    acc = facc[rt][nt][0][oo];
                } else if constexpr (BITS == 4 && W4COMB) {
                    acc = facc[rt][nt][0][0];
                } else {
                    acc = __fmul_rn(facc[rt][nt][(oo * BITS) >> 1][0], 0.5f);
#pragma unroll
                    for (int pl = 1; pl < BITS; ++pl) acc = __fadd_rn(acc, __fmul_rn(facc[rt][nt][(oo * BITS + pl) >> 1][(oo * BITS + pl) & 1], g_alpha(pl)));
                }
                if (o < Mw) g_st(M.C, a.out_f16, (size_t)n * Mw + o, acc);
            }
        }
    }
}

bool gemm_onehot_supported(const Shape& s) {
    if (s.lay != 2 || (s.bits != 2 && s.bits != 4) || s.K % 64 != 0) return false;
    if (s.m_groups >= 1) return s.ags == s.K && s.Mw % s.m_groups == 0;      // unified scale (BitNet): int32 over the whole K
    return s.ags == 64 && s.gs % 64 == 0;
}

hipError_t launch_gemm_onehot(const GemmArgs& a_in, hipStream_t st) {
    if (!gemm_onehot_supported(a_in.s) || a_in.nmat < 1 || a_in.nmat > 4 || (a_in.dump && a_in.nmat != 1)) return hipErrorInvalidValue;
    GemmArgs a = a_in;
    const int bits = a.s.bits;
    const int rows_per_wg = GWV * GRT * 16 / bits;
    int gx = 0;
    for (int i = 0; i < a.nmat; ++i) {
        gx += (a.m[i].Mw + rows_per_wg - 1) / rows_per_wg;
        a.m[i].wg_end = gx;
    }
    // 64-column tiles unless that leaves fewer than two workgroups (two waves per SIMD) per CU
    const bool narrow = (long)gx * ((a.N + 63) / 64) < 2 * 256 && a.N > 32;
    const int ncols = narrow ? 32 : 64;
    dim3 g(gx, (a.N + ncols - 1) / ncols), b(64 * GWV);
#define GL2(B, Z, D, U) do { if (narrow) hipLaunchKernelGGL((k_gemm_onehot<B, Z, D, 2, 8, U>), g, b, 0, st, a); \
                             else hipLaunchKernelGGL((k_gemm_onehot<B, Z, D, 4, 4, U>), g, b, 0, st, a); } while (0)
#define GL(B, Z) do { if (a.dump) GL2(B, Z, true, false); else GL2(B, Z, false, false); } while (0)
    if (a.s.m_groups >= 1) {
        if (bits == 2) { if (a.dump) GL2(2, false, true, true); else GL2(2, false, false, true); }
        else { if (a.dump) GL2(4, false, true, true); else GL2(4, false, false, true); }
    } else if (bits == 2) { if (a.s.zero_point) GL(2, true); else GL(2, false); }
    else { if (a.s.zero_point) GL(4, true); else GL(4, false); }
#undef GL
#undef GL2
    return hipGetLastError();
}

}  // namespace tmac
