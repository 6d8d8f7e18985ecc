// tmac_gemm2.hip — k_gemm_planes: qgemm_lut for N > 1 activation rows (prefill), bit-planes combined inside the
// matrix-core operand; k_lut_image: the LUT build that feeds it.
//
// The reference loops its GEMV micro-kernel over the activation rows (python/t_mac/ops/qgemm.py:183-190,228-231); per
// (activation row n, output row o, act group kk) it needs the integer
//     comb[n][o][kk] = sum_p 2^p * PS_p,   PS_p = sum_t QLUT[n][t][nibble_p(o, t)]      (tbl.cc:445-462, planes p)
// and then the fp32 chain  C += ((comb / 2) * lut_scale + lut_bias / 2) * scale + zero * lut_bias   (tbl.cc:464-526).
// k_gemm_onehot (tmac_gemm.hip) forms PS_p per bit-plane row as a product with a signed one-hot matrix.  Here the
// planes are merged BEFORE the matrix core: the A operand byte of (o, table t, half-table entry e) is
//     S[o][(t, e)] = sum_p 2^p * (+-1 if the recoded nibble of plane p selects entry e)        in {-3..3} (W2)
// so one MFMA row is an OUTPUT row (not a bit-plane row): half the MFMA work for W2, a quarter for W4, and one fp32
// chain per output instead of one conversion per plane.  comb is bit-identical to the per-plane sums combined as
// integers (integer arithmetic, no rounding); a debug tap exposes it for the parity tests.
//
// Operand construction: in the QUAD layout byte beta of dword 2P (W2) holds the plane-0 and plane-1 nibbles of table 2P
// for row 4q + beta: that byte is a JOINT index into a 256-entry table of 8-byte operand rows, kept in LDS in 32 copies
// (copy = lane & 31, entry stride 256 B) so that a wave's ds_read_b64 gather never has a bank conflict.  W4 has two joint
// bytes per table (planes 0/1 and 2/3): the entries carry a +3 bias per byte so that  row01 + (row23 << 2)  needs no
// byte-wise carry handling; the bias (15 per operand byte) leaves with the sum of the table entries, which the LUT build
// provides per (n, kk) and the epilogue folds into the int -> float conversion constant.
//
// 1- and 3-bit weights (round 3): the layout packs nibble quad q = table * bits + plane, so a byte of an odd width mixes TABLES,
// not just planes.  Its table entry is 16 bytes: the operand row of the low nibble | the operand row of the high nibble, each a
// single plane's +-1 (W3: + 1 per byte, the W4 trick) -- 16 copies, ds_read_b128 (whose lane groups are 16 wide).  W1: one gather
// IS the A operand of a 32-deep step (two tables).  W3: three gathers per table pair, bytes (t0p0|t0p1), (t0p2|t1p0), (t1p1|t1p2):
// row(t0) = lo(G0) + 2 hi(G0) + 4 lo(G1), row(t1) = hi(G1) + 2 lo(G2) + 4 hi(G2); the +7 per operand byte leaves like W4's +15.
// Here a lane holds ALL tables of one unit (its k half picks the unit of the act group), not half the tables of both: whole uint4.
//
// v_mfma_i32_32x32x32_i8: one instruction = 32 output rows x 32 activation rows x 4 tables.  Wave tile 64 x 64 (2 x 2
// MFMA tiles); a workgroup = 8 (or 4: PForm below) waves = ONE 64 x 64 output tile, the waves split K by weight groups and reduce through
// LDS at the end (at N = 256 a llama-2-7B projection has only 256 such tiles: one per CU; a larger workgroup tile would
// idle most of the chip).  Each wave is on its own between the kernel's two barriers.  Round 6: everything a step loads goes global ->
// REGISTERS one act group ahead through buffer instructions whose only varying part is a scalar offset -- the B operands (half tables of
// 64 activation rows, 8 KB per step) each 16-byte piece straight to the lane whose MFMA operand it is, into the registers of the
// operand it replaces (rounds 2-5 moved them global -> LDS by DMA and LDS -> registers), weights, column values, weight scales.
// LDS holds the operand-row table (32 copies) and the rows' weight scales only.  All waits are counted by the compiler (the prologue
// issues its loads in the loop's order; no conditional loads).
// Roofline: int8 MFMA; ops = 2 * Mw * (K / 4 * 8) * N.  Measured (profiles/r06_prefill_step_diet.txt): bound by what a step
// moves and issues -- the B operands' 64 distinct cache lines through the CU's vector-memory path 34 %, the per-act-group fp32 chain
// 18 %, the matrix core 14 %, the operand-row gathers 13 % -- not by latency.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdlib.h>

#include "tmac_quad_core.h"
#include "tmac_kernels.h"

namespace tmac {

typedef int p4i_t __attribute__((ext_vector_type(4)));
typedef int p16i_t __attribute__((ext_vector_type(16)));
typedef float p2f_t __attribute__((ext_vector_type(2)));
typedef float p4f_t __attribute__((ext_vector_type(4)));
typedef float p16f_t __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// LUT build for the GEMM: the pair-wise build of k_preprocess_pairs (same arithmetic, lut_ctor.cc:120-215 bit for bit)
// with 8 activation rows x 8 pairs per 64 lanes, written in the layout the GEMM streams (round 6: chunk-major, so that everything a
// step of k_gemm_planes fetches is ONE scalar offset plus immediates):
//   bimg [act group kk][n tile = n / 64][part = 4 (unit & 1) + pair][n & 63] uint4   signed half tables of tables 2 pair, 2 pair + 1 of
//                                                   unit 2 kk + (part >> 2) (32 activations) for row n: 8 KB per (act group, 64 rows)
//   colv [kk][n] float4            lut_scales / 2 | lut_biases / 2 | sum of the act group's 128 half-table entries | lut_biases
// (the halves are exact: what the reference multiplies by 0.5 in tbl.cc:464-526).  n runs over Npad rows (rows >= N repeat row N-1:
// finite values, never stored).
// ---------------------------------------------------------------------------------------------
template <bool F16>
__global__ __launch_bounds__(256) void k_lut_image(const void* __restrict__ B, uint4* __restrict__ bimg, float* __restrict__ colv,
                                                   int K, int N, int Npad) {
    const int kk = blockIdx.y;
    const int n = blockIdx.x * 32 + (threadIdx.x >> 3), p = threadIdx.x & 7;
    const int nn = min(n, N - 1);
    float x[8];
    if (F16) {
        const uint4 v = reinterpret_cast<const uint4*>(reinterpret_cast<const __half*>(B) + (size_t)nn * K + kk * 64)[p];
        const uint32_t r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const __half2 hh = *reinterpret_cast<const __half2*>(&r[i]);
            x[2 * i] = __low2float(hh); x[2 * i + 1] = __high2float(hh);
        }
    } else {
        const float4* src = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(B) + (size_t)nn * K + kk * 64) + 2 * p;
        const float4 a0 = src[0], a1 = src[1];
        x[0] = a0.x; x[1] = a0.y; x[2] = a0.z; x[3] = a0.w; x[4] = a1.x; x[5] = a1.y; x[6] = a1.z; x[7] = a1.w;
    }
    const float s0 = __fadd_rn(__fadd_rn(fabsf(x[0]), fabsf(x[1])), __fadd_rn(fabsf(x[2]), fabsf(x[3])));
    const float s1 = __fadd_rn(__fadd_rn(fabsf(x[4]), fabsf(x[5])), __fadd_rn(fabsf(x[6]), fabsf(x[7])));
    const float mx = q_half_allmax(fmaxf(s0, s1));
    const float scales = div127(mx);
    const float t_scales = (scales != 0.0f) ? rcp_exact(scales) : 0.0f;
    uint32_t lo0, hi0, lo1, hi1;
    float La, Lb;
    q_table8<true>(x[0], x[1], x[2], x[3], t_scales, lo0, hi0, La);
    q_table8<true>(x[4], x[5], x[6], x[7], t_scales, lo1, hi1, Lb);
    bimg[(((size_t)kk * (Npad >> 6) + (n >> 6)) * 8 + p) * 64 + (n & 63)] = make_uint4(lo0, hi0, lo1, hi1);
    // sum of the 16 signed entries of this lane's two half tables, then over the 8 lanes of the act group
    int h = 0;
    const uint32_t d[4] = {lo0, hi0, lo1, hi1};
#pragma unroll
    for (int i = 0; i < 4; ++i)
        h += (int)(int8_t)(d[i] & 0xff) + (int)(int8_t)((d[i] >> 8) & 0xff) + (int)(int8_t)((d[i] >> 16) & 0xff) + ((int)d[i] >> 24);
    h += (int)qdpp_u<0xB1>((uint32_t)h);
    h += (int)qdpp_u<0x4E>((uint32_t)h);
    h += (int)qdpp_u<0x104>((uint32_t)h);
    float va = -La, vb = -Lb;               // lut_biases, lut_ctor.cc:25-31 (summation order as in k_preprocess_pairs)
    va = __fadd_rn(va, qdpp_f<0x4E>(va));
    vb = __fadd_rn(vb, qdpp_f<0x4E>(vb));
    va = __fadd_rn(va, qdpp_f<0xB1>(va));
    vb = __fadd_rn(vb, qdpp_f<0xB1>(vb));
    const float v = __fadd_rn(va, vb);
    const float c1 = qdpp_f<0x104>(v);
    if (p == 0) {
        const float lb = __fadd_rn(__fadd_rn(0.0f, v), c1);
        reinterpret_cast<float4*>(colv)[(size_t)kk * Npad + n] = make_float4(__fmul_rn(0.5f, scales), __fmul_rn(0.5f, lb), (float)h, lb);
    }
}

hipError_t launch_lut_image(const void* B, int act_f16, void* bimg, float* colv, int K, int N, int Npad, hipStream_t st) {
    if (K % 64 != 0 || N < 1 || Npad < N || Npad % 64 != 0) return hipErrorInvalidValue;
    dim3 g(Npad / 32, K / 64), b(256);
    if (act_f16) hipLaunchKernelGGL((k_lut_image<true>), g, b, 0, st, B, (uint4*)bimg, colv, K, N, Npad);
    else hipLaunchKernelGGL((k_lut_image<false>), g, b, 0, st, B, (uint4*)bimg, colv, K, N, Npad);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Two forms of the workgroup (same tile, same arithmetic, same order of the fp32 sums within a K range):
//   NWV = 8: eight K ranges, 128 KB of LDS (the K-range reduction) -- one workgroup per CU.  For launches with at most one tile per CU
//            (o / down projection at N = 256): all of a CU's waves work on its one tile.
//   NWV = 4: four K ranges, 68 KB -- TWO workgroups per CU, each with one wave per SIMD.  A tile's fixed phases (operand rows,
//            K-range reduction and store 1.3 us, the wait for its slowest wave) then overlap the other workgroup's main loop.
// LDS: the joint-index operand rows in 32 copies (copy = lane & 31, entry stride 256 B: a wave's gather never has a bank conflict),
// then the weight scales / zeros of every wave's 64 rows [wave][buffer 2][sc 64 | zr 64] float; the reduction reuses the front.
// Rounds 2-5 also kept a half-table chunk buffer per wave there, filled by LDS-DMA (8 x 1 KB pieces per step) and read back into the B
// operands; round 6 loads the B operands global -> registers (below).  What that and the other steps bought, the knock-outs and the
// experiments that lost (fp32 / MFMA interleave, another tile order, operand-row prefetch, scales kept in registers, cache-policy
// bits, the magic conversion for W4) are in profiles/r06_prefill_step_diet.txt; the code of each is in the history of this file.
template <int NWV>
struct PForm {
    static constexpr int PAT_BYTES = 256 * 32 * 8;
    static constexpr int SC_OFF = PAT_BYTES;
    static constexpr int SC_WAVE = 2 * 512;
    static constexpr int LDS_BYTES = (SC_OFF + NWV * SC_WAVE) > NWV * 16384 ? (SC_OFF + NWV * SC_WAVE) : NWV * 16384;
};
template <int NWV>
struct PFormU {                  // the unified-scale kernel: operand rows only (the reduction reuses them)
    static constexpr int LDS_BYTES = 65536 > NWV * 16384 ? 65536 : NWV * 16384;
};

#ifndef TMAC_G2_KO
#define TMAC_G2_KO 0            // timing experiments only (results wrong): 1 = half of the B operand loads, 2 = none of them, 4 = no fp32 chain, 8 = no weight
                                // loads after the first, 16 = no operand-row gathers, 64 = no MFMA, 128 = no weight-scale rows from LDS
#endif

// Everything a step loads goes global -> registers one act group ahead through buffer instructions: per-lane byte offsets are computed
// once, the part that changes from act group to act group is a scalar offset (no vector address arithmetic in the loop).  The B
// operands -- the half tables of 64 activation rows, 8 KB per step -- go each 16-byte piece straight to the lane whose MFMA operand it
// is (the chunk-major image of k_lut_image is laid out for exactly that), INTO THE REGISTERS OF THE OPERAND THEY REPLACE, issued right
// behind the last chain that reads it.  The loads of the next act group are unconditional (the last step fetches its own act group
// again, harmlessly): the step is one basic block and every wait is counted by the compiler -- for that the prologue issues its loads
// in the order a step does (the wait counts of both ways into the loop are merged: another order, or a conditional load, turned waits
// of every step into vmcnt(0), a drain of the B loads issued last).
template <int BITS, bool ZP, bool DUMP, bool SCF16, int NWV>
__global__ __launch_bounds__(64 * NWV, 8 / NWV) void k_gemm_planes(Gemm2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char plds[];
    using PF = PForm<NWV>;
    constexpr int P_NWV = NWV, P_SC_OFF = PF::SC_OFF, P_SC_WAVE = PF::SC_WAVE;
    constexpr int NJ = BITS;                   // uint4 per unit and row quad in the QUAD layout
    constexpr bool ODD = (BITS & 1) != 0;      // 1- / 3-bit weights: 16-byte table entries, a lane holds one whole unit per tile row
    constexpr int WPU = ODD ? BITS : BITS / 2; // uint4 of weights per lane, tile row and [even widths: unit | odd: act group]
    constexpr int WUN = ODD ? 1 : 2;
    constexpr int BIASB = BITS == 4 ? 15 : BITS == 3 ? 7 : 0;   // what the biased operand bytes add per half-table entry
    const Shape& s = a.s;
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int kb = lane >> 5, j = lane & 31;

    // blockIdx -> (row block, token block): consecutive workgroup ids go round the 8 XCDs, so XCD x takes row blocks
    // x, x + 8, ... with all their token blocks one after the other (they share the weight rows in that XCD's L2)
    const int xcd = blockIdx.x & 7, qid = blockIdx.x >> 3;
    int bx = xcd + 8 * (qid / a.gy);
    const int by = qid % a.gy;
    if (bx >= a.gx) return;
    int mi = 0;
    while (mi + 1 < a.nmat && bx >= a.m[mi].wg_end) ++mi;
    if (mi > 0) bx -= a.m[mi - 1].wg_end;
    const GemmMat M = a.m[mi];
    const int Mw = M.Mw, G = s.K / 64, nu = s.K / 32, nst = (nu + 63) >> 6, nq = (Mw + 3) >> 2;
    const int row0 = bx * 64, n0 = by * 64;
    const int apg_sh = a.apg_shift, apg_m = (1 << apg_sh) - 1, nsg = G >> apg_sh;      // act groups per weight group: 1 << apg_sh
    const int g_lo = (w * nsg) / P_NWV, g_hi = ((w + 1) * nsg) / P_NWV;
    const int k_lo = g_lo << apg_sh, k_end = g_hi << apg_sh;

    // ---- per-lane constants and the first loads (they do not need the operand rows built below) ------------------------
    // v_perm selector that builds an operand-row address from a weight dword: byte 0 = copy offset, byte 1 = byte beta of the dword
    const uint32_t psel = 0x0c0c0000u | ((4u + (lane & 3)) << 8);
    const uint32_t copyoff = ODD ? (uint32_t)(lane & 15) * 16u : (uint32_t)j * 8u;     // even widths: 32 copies of 8 bytes; odd: 16 copies of 16 bytes
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(M.W), (short)0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.bimg), (short)0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.colv), (short)0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(M.SC), (short)0, 0x7fffffff, 0x00020000);
    int wvoff[2];                              // byte offset of (row quad of this lane in tile row rt, uint4 kb [W2] / 2 kb [W4]) in the weights
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int quad = min((row0 >> 2) + rt * 8 + (j >> 2), nq - 1);
        wvoff[rt] = ODD ? (quad * nst * NJ * 64 + kb) * 16                      // the unit of the act group this lane's k half stands for
                        : ((quad * nst * NJ + (BITS == 2 ? kb : 2 * kb)) * 64) * 16;
    }
    // chunk-major LUT image (k_lut_image): the chunk of (act group, n tile) is 8 KB in one piece, the column values one float4 per row
    const int cvoff = (n0 + j) * 16, chunk_stride = (a.Npad >> 6) * 8192, col_stride = a.Npad * 16;
    constexpr int SC_ESZ = SCF16 ? 2 : 4, SC_PER = ZP ? 2 : 1;
    const int scvoff = (min((row0 >> 2) + (lane >> 2), nq - 1) * nsg * 4 + (lane & 3)) * SC_PER * SC_ESZ;   // quad_scale_index, group 0
    const uint32_t sc_wave = P_SC_OFF + w * P_SC_WAVE;

    uint4 wv[WUN][2][WPU];                     // weights of the act group: [unit][tile row][..]
    // raw scale / zero of row (row0 + lane) of the NEXT weight group, staged.  One slot, not two by parity: a register array indexed by a
    // run-time parity made every load land through a select, i.e. behind an s_waitcnt vmcnt(0) right after its issue (ISA reading, round 3)
    uint32_t st_sc = 0u, st_zr = 0u, st0_sc = 0u, st0_zr = 0u;
    float lbs[2] = {0.f, 0.f};                 // lut_biases summed over the act groups of the current weight group, per n tile
    float za[2] = {0.f, 0.f}, zb[2] = {0.f, 0.f};   // pending zero-point update: A (zero points of the lane's row) and B (summed lut_biases of its column), see zero_stage

    p4i_t bv[2][4];                            // B operands: [n tile][32-deep step], loaded a step ahead, in place
    // lane (kb, j), step ks, n tile nt: the 16 bytes at uint4 (pslot 64 + nt 32 + j) of the chunk, pslot = pair 2 kb + (ks & 1) of unit ks >> 1
    // (odd widths: pair ks of unit kb)
    const int bdvoff = ((ODD ? kb * 4 : 2 * kb) * 64 + j) * 16;
    auto load_b = [&](int kk, int nt) {
        if (TMAC_G2_KO & 2) return;
        const int so = kk * chunk_stride + by * 8192;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if ((TMAC_G2_KO & 1) && (ks & 1)) continue;
            const int c = ODD ? (ks * 64 + nt * 32) * 16 : ((ks & 1) * 64 + nt * 32) * 16;
            const u32x4q v = __builtin_amdgcn_raw_buffer_load_b128(rs_b, bdvoff + c, ODD ? so : so + (ks >> 1) * 4096, 0);
            bv[nt][ks] = (p4i_t){(int)v[0], (int)v[1], (int)v[2], (int)v[3]};
        }
    };
    auto load_weights = [&](int kk, int rt) {  // tile row rt of the act group's two units
        if ((TMAC_G2_KO & 8) && kk != k_lo) return;
        const int u0 = 2 * kk, so = ((u0 >> 6) * NJ * 64 + (u0 & 63)) * 16;       // (the act group's second unit: + 16, an immediate)
#pragma unroll
        for (int ul = 0; ul < WUN; ++ul) {
#pragma unroll
            for (int q = 0; q < WPU; ++q) {
                const u32x4q v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wvoff[rt] + ul * 16 + q * 1024, so, 0);
                wv[ul][rt][q] = make_uint4(v[0], v[1], v[2], v[3]);
            }
        }
    };
    // (the dynamic LDS block starts at LDS address 0 -- the kernel has no static LDS -- so the perm's result IS the address: going through
    // `plds + offset` costs a v_add_u32 with the relocated base, zero, per gather)
    typedef unsigned int p2u_t __attribute__((ext_vector_type(2)));
    typedef unsigned int p4u_t __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const p2u_t* lds_u2_ptr;
    typedef __attribute__((address_space(3))) const p4u_t* lds_u4_ptr;
    auto pat_row = [&](uint32_t d) -> uint2 {
        const p2u_t v = *(lds_u2_ptr)(uintptr_t)__builtin_amdgcn_perm(d, copyoff, psel);
        return make_uint2(v.x, v.y);
    };
    auto pat_row2 = [&](uint32_t d) -> uint4 {  // odd widths: the operand rows of the byte's two nibbles
        const p4u_t v = *(lds_u4_ptr)(uintptr_t)__builtin_amdgcn_perm(d, copyoff, psel);
        return make_uint4(v.x, v.y, v.z, v.w);
    };
    auto load_staged = [&](int g) {            // scale / zero of row (row0 + lane), weight group g -> registers, raw (converted when written to LDS)
        const int so = g * 4 * SC_PER * SC_ESZ;   // buffer loads: the lane's part of the address is a constant, the group a scalar offset
        if constexpr (SCF16) {
            // (scale, zero) is one aligned dword: kept whole, split when it is written to LDS -- splitting here is arithmetic on the
            // load's result, i.e. a wait for it
            if (ZP) st_sc = __builtin_amdgcn_raw_buffer_load_b32(rs_s, scvoff, so, 0);
            else st_sc = __builtin_amdgcn_raw_buffer_load_b16(rs_s, scvoff, so, 0);
        } else {
            st_sc = __builtin_amdgcn_raw_buffer_load_b32(rs_s, scvoff, so, 0);
            if (ZP) st_zr = __builtin_amdgcn_raw_buffer_load_b32(rs_s, scvoff + 4, so, 0);
        }
    };
    auto st_val = [&](uint32_t raw) -> float {
        if constexpr (SCF16) return __half2float(__ushort_as_half((unsigned short)raw));
        else return __uint_as_float(raw);
    };
    auto write_staged = [&](int g) {           // the staged values are group g's; LDS buffer (g - g_lo) & 1
        const int sl = (g - g_lo) & 1;
        float* p = reinterpret_cast<float*>(plds + sc_wave + sl * 512);
        if constexpr (SCF16 && ZP) {
            p[lane] = st_val(st_sc & 0xffffu);
            p[64 + lane] = st_val(st_sc >> 16);
        } else {
            p[lane] = st_val(st_sc);
            if (ZP) p[64 + lane] = st_val(st_zr);
        }
    };
    // accumulator rows of this lane in tile row rt: 8 q4 + 4 kb + (0..3), q4 = 0..3
    auto read_rows = [&](int buf, int which, int rt, p2f_t (&dst)[8]) {
        if constexpr ((TMAC_G2_KO & 128) != 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) dst[q] = (p2f_t){(float)(buf + q), (float)(which + rt)};
            return;
        }
        const unsigned char* p = plds + sc_wave + buf * 512 + which * 256;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const p4f_t v = *reinterpret_cast<const p4f_t*>(p + (32 * rt + 8 * q4 + 4 * kb) * 4);
            dst[2 * q4] = (p2f_t){v.x, v.y}; dst[2 * q4 + 1] = (p2f_t){v.z, v.w};
        }
    };

    // profiling: s_memrealtime (100 MHz) at fixed points of every act-group step of workgroup 0
    unsigned long long* stp = (a.stamps && blockIdx.x == 0) ? a.stamps + (size_t)w * 64 * 8 : nullptr;
#ifndef TMAC_G2_STAMPS
#define TMAC_G2_STAMPS 0          // profiling builds only (tools/build_variant_obj.sh x tmac_gemm2 "-DTMAC_G2_STAMPS=1"): the hook -- a branch and a
#endif                           // scalar-memory read per step even with no buffer set -- cost the W4 line 1 % (7.80 -> 7.72 ms), W2 0.2 %
#define PSTAMP(step, i) do { if (TMAC_G2_STAMPS && stp && (step) < 64 && lane == 0) stp[(step) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    // stamps INSIDE a step (tools/gemm2_stamps.py prints the phases): a diagnostic build only (-DTMAC_G2_STEP_STAMPS=1) -- their mere
    // presence (a branch and an exec-mask change at four places of the step) cost 1.7 % of the prefill line in rounds 2-5.  With the
    // round-6 loads into registers they DISTORT the step: a pending store makes the compiler's wait counting give up, every stamp is
    // followed by vmcnt(0) -- a drain of the B loads in flight -- and a step takes twice as long.  Knock-outs and counters instead.
#ifndef TMAC_G2_STEP_STAMPS
#define TMAC_G2_STEP_STAMPS 0
#endif
#define PSTAMP_IN(step, i) do { if (TMAC_G2_STEP_STAMPS) PSTAMP(step, i); } while (0)
    PSTAMP(0, 5);
    u32x4q cn[2];                              // column values (lut_scales / 2, lut_biases / 2, entry sum, lut_biases) of the NEXT act group, per n tile
    auto load_cols = [&](int kk) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) cn[nt] = __builtin_amdgcn_raw_buffer_load_b128(rs_c, cvoff + nt * 512, kk * col_stride, 0);
    };
    const bool work = k_lo < k_end;
    if (work) {                                // everything the first step needs is in flight while the operand rows are built,
        load_staged(g_lo);                     // in the order a step issues the loads of its successor
        st0_sc = st_sc; st0_zr = st_zr;        // (the first group's values: written to LDS behind the table's barrier)
        load_staged(g_lo + 1 < g_hi ? g_lo + 1 : g_lo);     // ... and the second group's, where every later step has its staged load: in front
        load_cols(k_lo);
        load_weights(k_lo, 0);
        load_weights(k_lo, 1);
        load_b(k_lo, 0); load_b(k_lo, 1);
    }

    // ---- joint-index operand rows: entry b = (i1 << 4) | i0, byte e = s(i0) [e == i0 & 7] + 2 s(i1) [e == i1 & 7] (+ 3 for W4)
    {
        const int b = tid & 255, i0 = b & 15, i1 = b >> 4;
        uint32_t lo = 0, hi = 0, lo1 = 0, hi1 = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (ODD) {                         // one plane per nibble: +-1 (W3: + 1), low nibble's row | high nibble's row
                const int v0 = (BITS == 3 ? 1 : 0) + (e == (i0 & 7) ? ((i0 & 8) ? -1 : 1) : 0);
                const int v1 = (BITS == 3 ? 1 : 0) + (e == (i1 & 7) ? ((i1 & 8) ? -1 : 1) : 0);
                const uint32_t b0 = (uint32_t)(v0 & 0xff) << (8 * (e & 3)), b1 = (uint32_t)(v1 & 0xff) << (8 * (e & 3));
                if (e < 4) { lo |= b0; lo1 |= b1; } else { hi |= b0; hi1 |= b1; }
            } else {
                int v = (BITS == 4) ? 3 : 0;
                if (e == (i0 & 7)) v += (i0 & 8) ? -1 : 1;
                if (e == (i1 & 7)) v += (i1 & 8) ? -2 : 2;
                const uint32_t by8 = (uint32_t)(v & 0xff) << (8 * (e & 3));
                if (e < 4) lo |= by8; else hi |= by8;
            }
        }
        if (!ODD) { lo1 = lo; hi1 = hi; }
        // an entry = 256 B = 32 copies of 8 bytes / 16 of 16.  NWV = 8: two threads per entry, NWV = 4: one; consecutive entries (lanes)
        // start at different 16-byte slots of their rows
        if constexpr (NWV == 4) {
            uint4* pt = reinterpret_cast<uint4*>(plds) + b * 16;
#pragma unroll
            for (int c = 0; c < 16; ++c) pt[(c + b) & 15] = make_uint4(lo, hi, lo1, hi1);
        } else {
            uint4* pt = reinterpret_cast<uint4*>(plds) + b * 16 + (tid >> 8) * 8;
#pragma unroll
            for (int c = 0; c < 8; ++c) pt[(c + b) & 7] = make_uint4(lo, hi, lo1, hi1);
        }
    }
    __syncthreads();

    p2f_t facc[2][2][8];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 8; ++r) facc[rt][nt][r] = (p2f_t){0.f, 0.f};

    PSTAMP(0, 6);
    if (work) {
        const uint32_t k_sc = st_sc, k_zr = st_zr;
        st_sc = st0_sc; st_zr = st0_zr;
        write_staged(g_lo);
        st_sc = k_sc; st_zr = k_zr;
    }
    // accumulators start from the bits of 3.0f, the middle of the binade [2, 4): as a float the int32 result is 3 + comb * 2^-22
    // exactly for |comb| < 2^21 (it is < 2^19 here), so int -> float is one packed subtraction per pair.  The 16 registers are
    // made opaque to the compiler, which otherwise rebuilds the constant vector with 15 moves in front of every chain.
    // W3 / W4 have no registers to spare for that: their chains start from zero and the conversion is 16 v_cvt_f32_i32 per tile.
    constexpr bool MAGIC = BITS <= 2;
    constexpr int CI = MAGIC ? 0x40400000 : 0;
    p16i_t cinit = {CI, CI, CI, CI, CI, CI, CI, CI, CI, CI, CI, CI, CI, CI, CI, CI};
    if (MAGIC) asm volatile("" : "+v"(cinit));
    auto build_av = [&](int rt, p4i_t (&av)[4]) {       // A operands of tile row rt: the joint plane index of (row, table) selects the operand row
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if constexpr ((TMAC_G2_KO & 16) != 0) { const uint4 q = wv[0][rt][0]; av[ks] = (p4i_t){(int)q.x, (int)q.y, (int)q.z, (int)(q.w + ks)}; continue; }
            if constexpr (BITS == 1) {             // dword ks of the unit = tables 2 ks, 2 ks + 1
                const uint4 q = wv[0][rt][0];
                const uint4 g = pat_row2(ks == 0 ? q.x : ks == 1 ? q.y : ks == 2 ? q.z : q.w);
                av[ks] = (p4i_t){(int)g.x, (int)g.y, (int)g.z, (int)g.w};
            } else if constexpr (BITS == 3) {      // dwords 3 ks .. 3 ks + 2 of the unit's twelve
                const uint32_t dw[12] = {wv[0][rt][0].x, wv[0][rt][0].y, wv[0][rt][0].z, wv[0][rt][0].w, wv[0][rt][1].x, wv[0][rt][1].y,
                                         wv[0][rt][1].z, wv[0][rt][1].w, wv[0][rt][2].x, wv[0][rt][2].y, wv[0][rt][2].z, wv[0][rt][2].w};
                const uint4 g0 = pat_row2(dw[3 * ks]), g1 = pat_row2(dw[3 * ks + 1]), g2 = pat_row2(dw[3 * ks + 2]);
                av[ks] = (p4i_t){(int)(g0.x + (g0.z << 1) + (g1.x << 2)), (int)(g0.y + (g0.w << 1) + (g1.y << 2)),
                                 (int)(g1.z + (g2.x << 1) + (g2.z << 2)), (int)(g1.w + (g2.y << 1) + (g2.w << 2))};
            } else if constexpr (BITS == 2) {
                const uint4 q = wv[ks >> 1][rt][0];
                const uint2 t0 = pat_row((ks & 1) ? q.z : q.x), t1 = pat_row((ks & 1) ? q.w : q.y);
                av[ks] = (p4i_t){(int)t0.x, (int)t0.y, (int)t1.x, (int)t1.y};
            } else {
                const uint4 q = wv[ks >> 1][rt][ks & 1];
                const uint2 a0 = pat_row(q.x), a1 = pat_row(q.y), b0 = pat_row(q.z), b1 = pat_row(q.w);
                av[ks] = (p4i_t){(int)(a0.x + (a1.x << 2)), (int)(a0.y + (a1.y << 2)), (int)(b0.x + (b1.x << 2)), (int)(b0.y + (b1.y << 2))};
            }
        }
    };
    // One act group (64 activations = four 32-deep MFMA steps) per iteration.  The order below interleaves the units a step keeps
    // busy -- the vector-memory path (B operands, weights), LDS (operand-row gathers, row scales), the matrix core and the VALU (fp32
    // scale chain) -- inside ONE wave: the operands of tile row 1 are fetched while the MFMAs of tile row 0 run, the fp32 chain of a
    // tile runs under the MFMAs of the next.  sched_barrier keeps the compiler from regrouping the phases (and a B load from being
    // hoisted above the last chain that reads its registers: that costs registers of its own and a copy at the loop's end).
    for (int kk = k_lo; kk < k_end; ++kk) {
        const int g = kk >> apg_sh;
        const bool glast = (kk & apg_m) == apg_m;          // last act group of its weight group
        const bool more = glast && g + 1 < g_hi;
        const bool next = kk + 1 < k_end;
        const int cbuf = (g - g_lo) & 1;
        PSTAMP(kk - k_lo, 0);
        // the older wave of a SIMD wins every issue conflict (its four K ranges finish ~15 % earlier and then wait at the
        // reduction): alternate the priority between the two waves of a SIMD step by step
        if (NWV == 8) { if (((kk - k_lo) ^ (w >> 2)) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        PSTAMP_IN(kk - k_lo, 1);
        if (more) write_staged(g + 1);
        // A operands and row scales of tile row 0
        p4i_t av0[4], av1[4];
        p2f_t sc0[8];
        build_av(0, av0);
        read_rows(cbuf, 0, 0, sc0);
        // column values of the act group (loaded one step ahead): v = x * H + hlbx with x = comb * 2^-22, H = (ls / 2) * 2^22,
        // hlbx = lb / 2  [- 15 * (entry sum) * (ls / 2) for the +15 operand bias of W4]
        float H[2], hlbx[2], lb[2];
        int bias[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float hls = __uint_as_float(cn[nt][0]);
            lb[nt] = __uint_as_float(cn[nt][3]);
            H[nt] = MAGIC ? __fmul_rn(hls, 4194304.0f) : hls;
            hlbx[nt] = __uint_as_float(cn[nt][1]);
            bias[nt] = 0;
            if (BIASB) {
                const float hs15 = __fmul_rn((float)BIASB, __uint_as_float(cn[nt][2]));
                hlbx[nt] = __fmaf_rn(-hs15, hls, hlbx[nt]);
                bias[nt] = (int)hs15;
            }
        }
        const int kn = next ? kk + 1 : kk;
        // every step, no condition: the scales of the group behind the NEXT step's group -- at the last act group of g that is g + 2,
        // before that g + 1 once more (the same dword again)
        { const int gs2 = ((kn >> apg_sh) + 1 < g_hi) ? (kn >> apg_sh) + 1 : g_hi - 1; load_staged(gs2); }
        load_cols(kn); load_weights(kn, 0);
        __builtin_amdgcn_sched_barrier(0);
        PSTAMP_IN(kk - k_lo, 2);

        auto chain = [&](const p4i_t (&av)[4], int nt, p16i_t& c) {      // one 32 x 32 tile of the act group: four dependent MFMAs
            if constexpr ((TMAC_G2_KO & 64) != 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = cinit[r] + av[r & 3][r >> 2] + bv[nt][r & 3][r >> 2];
                return;
            }
            c = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[0], bv[nt][0], cinit, 0, 0, 0);
#pragma unroll
            for (int ks = 1; ks < 4; ++ks) c = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[ks], bv[nt][ks], c, 0, 0, 0);
        };
        // C += ((comb / 2) ls + lb / 2) scale [+ zero lb, once per weight group], cf. k_gemv_quad / k_gemm_onehot (W4)
        auto epilogue = [&](int rt, int nt, const p16i_t& c, const p2f_t (&sc)[8]) {
            // three passes of independent instructions over the lane's 16 values (in place: int32 -> fp32 -> scaled), not
            // 8 dependent three-instruction chains through one temporary.  Packed fp32: as 48 plain instructions the tile was 3 % slower
            // (profiles/r06_prefill_unpacked.txt)
            if (TMAC_G2_KO & 4) { facc[rt][nt][0] += (p2f_t){__int_as_float(c[0]), __int_as_float(c[5])}; return; }
            const p2f_t h2 = {H[nt], H[nt]}, b2 = {hlbx[nt], hlbx[nt]};
            p2f_t x[8];
#pragma unroll
            for (int r2 = 0; r2 < 8; ++r2)
                x[r2] = MAGIC ? (p2f_t){__int_as_float(c[2 * r2]), __int_as_float(c[2 * r2 + 1])} - (p2f_t){3.0f, 3.0f}
                              : (p2f_t){(float)c[2 * r2], (float)c[2 * r2 + 1]};
#pragma unroll
            for (int r2 = 0; r2 < 8; ++r2) x[r2] = __builtin_elementwise_fma(x[r2], h2, b2);
#pragma unroll
            for (int r2 = 0; r2 < 8; ++r2) facc[rt][nt][r2] = __builtin_elementwise_fma(x[r2], sc[r2], facc[rt][nt][r2]);
            if (DUMP) {
                const int nn = n0 + nt * 32 + j;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int orow = row0 + 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * kb;
                    if (nn < a.N && orow < Mw) a.dump[((size_t)nn * Mw + orow) * G + kk] = c[r] - CI - bias[nt];
                }
            }
        };
        // The zero-point term of a weight group, zero[o] * (sum of its lut_biases)[n] (tbl.cc:497-505 regrouped), is a rank-1 update of the
        // 64 x 64 tile: two weight groups make one v_mfma_f32_32x32x2_f32 per 32 x 32 tile (k = the group's parity; lane (kb, j) holds row /
        // column j of the group kb), i.e. 4 matrix-core instructions per two weight groups instead of 64 packed fp32 ones (rounds 2-5).
        // fp32 products and sums as before, in another order.
        auto zero_stage = [&]() {              // (at the last act group of a weight group)
            const float* zp = reinterpret_cast<const float*>(plds + sc_wave + cbuf * 512 + 256);
            const bool mine = kb == cbuf;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const float z = zp[32 * rt + j];
                za[rt] = mine ? z : (cbuf ? za[rt] : 0.f);
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const float l = __fadd_rn(lbs[nt], lb[nt]);
                zb[nt] = mine ? l : (cbuf ? zb[nt] : 0.f);
            }
        };
        auto zero_flush = [&]() {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    p16f_t c;
#pragma unroll
                    for (int r2 = 0; r2 < 8; ++r2) { c[2 * r2] = facc[rt][nt][r2].x; c[2 * r2 + 1] = facc[rt][nt][r2].y; }
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(za[rt], zb[nt], c, 0, 0, 0);
#pragma unroll
                    for (int r2 = 0; r2 < 8; ++r2) facc[rt][nt][r2] = (p2f_t){c[2 * r2], c[2 * r2 + 1]};
                }
        };

        // tile pipeline: the MFMAs of tile t + 1 run under the fp32 chain of tile t (two accumulator sets)
        constexpr bool SC2 = BITS <= 2;        // W1, W2: tile row 1's scales in registers of their own, fetched early (W3 / W4 have none to spare)
        p16i_t ca, cb;
        p2f_t sc1s[SC2 ? 8 : 1];
        p2f_t (&sc1)[8] = *reinterpret_cast<p2f_t (*)[8]>(SC2 ? &sc1s[0] : &sc0[0]);
        chain(av0, 0, ca);
        chain(av0, 1, cb);
        __builtin_amdgcn_sched_barrier(0);
        build_av(1, av1);
        if (SC2) read_rows(cbuf, 0, 1, sc1);
        load_weights(kn, 1);
        __builtin_amdgcn_sched_barrier(0);
        PSTAMP_IN(kk - k_lo, 3);
        epilogue(0, 0, ca, sc0);
        __builtin_amdgcn_sched_barrier(0);
        chain(av1, 0, ca);
        load_b(kn, 0);                         // (n tile 0's operands have been read by their last chain)
        __builtin_amdgcn_sched_barrier(0);
        epilogue(0, 1, cb, sc0);
        if (!SC2) read_rows(cbuf, 0, 1, sc1);
        __builtin_amdgcn_sched_barrier(0);
        chain(av1, 1, cb);
        load_b(kn, 1);
        __builtin_amdgcn_sched_barrier(0);
        PSTAMP_IN(kk - k_lo, 4);
        epilogue(1, 0, ca, sc1);
        __builtin_amdgcn_sched_barrier(0);
        epilogue(1, 1, cb, sc1);
        if (ZP && glast) { zero_stage(); if (cbuf || g + 1 >= g_hi) zero_flush(); }     // (behind the act group's own terms: one branch at the end of the block)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) lbs[nt] = glast ? 0.f : __fadd_rn(lbs[nt], lb[nt]);
    }

    __builtin_amdgcn_s_setprio(0);
    // ---- reduce the K ranges through LDS (the operand rows are free now) and store ------------------
    PSTAMP(1, 5);
    __syncthreads();
    PSTAMP(1, 6);
    {
        unsigned char* red = plds + w * 16384;                // [n 64][o 64] fp32, 16-byte slots XOR-swizzled by n & 15
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int nl = nt * 32 + j, slot = (8 * rt + 2 * q4 + kb) ^ (nl & 15);
                    *reinterpret_cast<p4f_t*>(red + nl * 256 + slot * 16) =
                        (p4f_t){facc[rt][nt][2 * q4][0], facc[rt][nt][2 * q4][1], facc[rt][nt][2 * q4 + 1][0], facc[rt][nt][2 * q4 + 1][1]};
                }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16 / P_NWV; ++i) {
        const int idx = tid + 64 * P_NWV * i, nl = idx >> 4, sl = idx & 15;
        const unsigned char* p = plds + nl * 256 + ((sl ^ (nl & 15)) * 16);
        p4f_t v = *reinterpret_cast<const p4f_t*>(p);
#pragma unroll
        for (int ww = 1; ww < P_NWV; ++ww) v += *reinterpret_cast<const p4f_t*>(p + ww * 16384);
        const int n = n0 + nl, o = row0 + sl * 4;
        if (n < a.N && o < Mw) {
            if (a.out_f16) {
                const __half2 h0 = __floats2half2_rn(v.x, v.y), h1 = __floats2half2_rn(v.z, v.w);
                *reinterpret_cast<uint2*>(reinterpret_cast<__half*>(M.C) + (size_t)n * Mw + o) =
                    make_uint2(*reinterpret_cast<const uint32_t*>(&h0), *reinterpret_cast<const uint32_t*>(&h1));
            } else {
                *reinterpret_cast<p4f_t*>(reinterpret_cast<float*>(M.C) + (size_t)n * Mw + o) = v;
            }
        }
    }
    PSTAMP(2, 5);
#undef PSTAMP_IN
#undef PSTAMP
}

// ---------------------------------------------------------------------------------------------
// The unified-scale flavour (BitNet: m_groups >= 1, one act group per activation row; tbl_g4_int8_int32_update +
// qgemm.py:170-174) on the same operands: nothing is scaled per act group, so the int32 tiles accumulate over the wave's whole
// K range, the eight ranges are added as integers (exact, any order) and the scale-final expression
//   C = ((comb / 2) * lut_scale[n] + lut_bias[n] / 2) * scale[o / (Mw / m_groups)]
// runs once per output -- bit for bit what the GEMV kernel and the oracle compute from the per-plane totals (comb / 2 =
// sum_p alpha_p * total_p is exact in fp32: |comb| < 2^24).  1- to 4-bit weights (round 4; 2-bit only before): the operand rows
// of k_gemm_planes per width -- 3- and 4-bit rows carry +7 / +15 per operand byte, which leaves as BIASB x (sum of ALL half-table
// entries of activation row n), an integer the row-wise LUT build provides (colv[2][n], int32 bits).
// The LUT image comes from k_preprocess_pairs_row (tmac_quad.hip), colv holds lut_scales | lut_biases | entry sum with one act group.
template <int BITS, bool DUMP, int NWV>
__global__ __launch_bounds__(64 * NWV, 8 / NWV) void k_gemm_planes_us(Gemm2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char plds[];
    constexpr int P_NWV = NWV;
    constexpr int NJ = BITS;
    constexpr bool ODD = (BITS & 1) != 0;      // see k_gemm_planes: 16-byte table entries, a lane holds one whole unit per tile row
    constexpr int WPU = ODD ? BITS : BITS / 2;
    constexpr int WUN = ODD ? 1 : 2;
    constexpr int BIASB = BITS == 4 ? 15 : BITS == 3 ? 7 : 0;
    const Shape& s = a.s;
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int kb = lane >> 5, j = lane & 31;
    const int xcd = blockIdx.x & 7, qid = blockIdx.x >> 3;
    int bx = xcd + 8 * (qid / a.gy);
    const int by = qid % a.gy;
    if (bx >= a.gx) return;
    int mi = 0;
    while (mi + 1 < a.nmat && bx >= a.m[mi].wg_end) ++mi;
    if (mi > 0) bx -= a.m[mi - 1].wg_end;
    const GemmMat M = a.m[mi];
    const int Mw = M.Mw, nk = s.K / 64, nu = s.K / 32, nst = (nu + 63) >> 6, nq = (Mw + 3) >> 2;
    const int row0 = bx * 64, n0 = by * 64;
    const int k_lo = (w * nk) / P_NWV, k_end = ((w + 1) * nk) / P_NWV;        // 64-activation steps of this wave

    const uint32_t psel = 0x0c0c0000u | ((4u + (lane & 3)) << 8);
    const uint32_t copyoff = ODD ? (uint32_t)(lane & 15) * 16u : (uint32_t)j * 8u;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(M.W), (short)0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.bimg), (short)0, 0x7fffffff, 0x00020000);
    int wvoff[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int quad = min((row0 >> 2) + rt * 8 + (j >> 2), nq - 1);
        wvoff[rt] = ODD ? (quad * nst * NJ * 64 + kb) * 16
                        : ((quad * nst * NJ + (BITS == 2 ? kb : 2 * kb)) * 64) * 16;
    }
    uint4 wv[WUN][2][WPU];
    // round 6, as in k_gemm_planes: the B operands go global -> registers, the 16 bytes of (unit, pair, row) straight to the lane whose
    // operand they are, into the registers of the step before as soon as its MFMAs have read them (the row-wise image addresses with
    // four scalar offsets per step)
    p4i_t bv[2][4];
    const int bdvoff = ((ODD ? 4 * kb : 2 * kb) * a.Npad + n0 + j) * 16;
    auto load_b = [&](int kk, int ks) {
        const int so = (ODD ? 8 * kk + ks : (2 * kk + (ks >> 1)) * 4 + (ks & 1)) * a.Npad * 16;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const u32x4q v = __builtin_amdgcn_raw_buffer_load_b128(rs_b, bdvoff + nt * 512, so, 0);
            bv[nt][ks] = (p4i_t){(int)v[0], (int)v[1], (int)v[2], (int)v[3]};
        }
    };
    auto load_weights = [&](int kk, int rt) {
#pragma unroll
        for (int ul = 0; ul < WUN; ++ul) {
            const int u = 2 * kk + ul, so = ((u >> 6) * NJ * 64 + (u & 63)) * 16;
#pragma unroll
            for (int q = 0; q < WPU; ++q) {
                const u32x4q v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wvoff[rt], so + q * 1024, 0);
                wv[ul][rt][q] = make_uint4(v[0], v[1], v[2], v[3]);
            }
        }
    };
    typedef unsigned int p2u_t __attribute__((ext_vector_type(2)));
    typedef unsigned int p4u_t __attribute__((ext_vector_type(4)));
    auto pat_row = [&](uint32_t d) -> uint2 {
        const uint32_t ad = __builtin_amdgcn_perm(d, copyoff, psel);
        const p2u_t v = *(__attribute__((address_space(3))) const p2u_t*)(uintptr_t)ad;   // absolute LDS address (see k_gemm_planes)
        return make_uint2(v.x, v.y);
    };
    auto pat_row2 = [&](uint32_t d) -> uint4 {   // odd widths: the operand rows of the byte's two nibbles
        const uint32_t ad = __builtin_amdgcn_perm(d, copyoff, psel);
        const p4u_t v = *(__attribute__((address_space(3))) const p4u_t*)(uintptr_t)ad;
        return make_uint4(v.x, v.y, v.z, v.w);
    };
    const bool work = k_lo < k_end;
    if (work) {
        load_weights(k_lo, 0); load_weights(k_lo, 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) load_b(k_lo, ks);
    }
    {   // joint-index operand rows, as in k_gemm_planes
        const int b = tid & 255, i0 = b & 15, i1 = b >> 4;
        uint32_t lo = 0, hi = 0, lo1 = 0, hi1 = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (ODD) {
                const int v0 = (BITS == 3 ? 1 : 0) + (e == (i0 & 7) ? ((i0 & 8) ? -1 : 1) : 0);
                const int v1 = (BITS == 3 ? 1 : 0) + (e == (i1 & 7) ? ((i1 & 8) ? -1 : 1) : 0);
                const uint32_t b0 = (uint32_t)(v0 & 0xff) << (8 * (e & 3)), b1 = (uint32_t)(v1 & 0xff) << (8 * (e & 3));
                if (e < 4) { lo |= b0; lo1 |= b1; } else { hi |= b0; hi1 |= b1; }
            } else {
                int v = (BITS == 4) ? 3 : 0;
                if (e == (i0 & 7)) v += (i0 & 8) ? -1 : 1;
                if (e == (i1 & 7)) v += (i1 & 8) ? -2 : 2;
                const uint32_t by8 = (uint32_t)(v & 0xff) << (8 * (e & 3));
                if (e < 4) lo |= by8; else hi |= by8;
            }
        }
        if (!ODD) { lo1 = lo; hi1 = hi; }
        if constexpr (NWV == 4) {
            uint4* pt = reinterpret_cast<uint4*>(plds) + b * 16;
#pragma unroll
            for (int c = 0; c < 16; ++c) pt[(c + b) & 15] = make_uint4(lo, hi, lo1, hi1);
        } else {
            uint4* pt = reinterpret_cast<uint4*>(plds) + b * 16 + (tid >> 8) * 8;
#pragma unroll
            for (int c = 0; c < 8; ++c) pt[(c + b) & 7] = make_uint4(lo, hi, lo1, hi1);
        }
    }
    __syncthreads();

    p16i_t acc[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][nt][r] = 0;
    for (int kk = k_lo; kk < k_end; ++kk) {
        const bool next = kk + 1 < k_end;
        if (NWV == 8) { if (((kk - k_lo) ^ (w >> 2)) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        p4i_t av[2][4];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if constexpr (BITS == 1) {             // dword ks of the unit = tables 2 ks, 2 ks + 1
                    const uint4 q = wv[0][rt][0];
                    const uint4 g = pat_row2(ks == 0 ? q.x : ks == 1 ? q.y : ks == 2 ? q.z : q.w);
                    av[rt][ks] = (p4i_t){(int)g.x, (int)g.y, (int)g.z, (int)g.w};
                } else if constexpr (BITS == 3) {      // dwords 3 ks .. 3 ks + 2 of the unit's twelve
                    const uint32_t dw[12] = {wv[0][rt][0].x, wv[0][rt][0].y, wv[0][rt][0].z, wv[0][rt][0].w, wv[0][rt][1].x, wv[0][rt][1].y,
                                             wv[0][rt][1].z, wv[0][rt][1].w, wv[0][rt][2].x, wv[0][rt][2].y, wv[0][rt][2].z, wv[0][rt][2].w};
                    const uint4 g0 = pat_row2(dw[3 * ks]), g1 = pat_row2(dw[3 * ks + 1]), g2 = pat_row2(dw[3 * ks + 2]);
                    av[rt][ks] = (p4i_t){(int)(g0.x + (g0.z << 1) + (g1.x << 2)), (int)(g0.y + (g0.w << 1) + (g1.y << 2)),
                                         (int)(g1.z + (g2.x << 1) + (g2.z << 2)), (int)(g1.w + (g2.y << 1) + (g2.w << 2))};
                } else if constexpr (BITS == 2) {
                    const uint4 q = wv[ks >> 1][rt][0];
                    const uint2 t0 = pat_row((ks & 1) ? q.z : q.x), t1 = pat_row((ks & 1) ? q.w : q.y);
                    av[rt][ks] = (p4i_t){(int)t0.x, (int)t0.y, (int)t1.x, (int)t1.y};
                } else {
                    const uint4 q = wv[ks >> 1][rt][ks & 1];
                    const uint2 a0 = pat_row(q.x), a1 = pat_row(q.y), b0 = pat_row(q.z), b1 = pat_row(q.w);
                    av[rt][ks] = (p4i_t){(int)(a0.x + (a1.x << 2)), (int)(a0.y + (a1.y << 2)), (int)(b0.x + (b1.x << 2)), (int)(b0.y + (b1.y << 2))};
                }
            }
        const int kn = next ? kk + 1 : kk;     // (the last step fetches its own operands again -- one basic block, counted waits)
        load_weights(kn, 0); load_weights(kn, 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[rt][nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[rt][ks], bv[nt][ks], acc[rt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0); load_b(kn, ks); __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    {
        unsigned char* red = plds + w * 16384;                // [n 64][o 64] int32, 16-byte slots XOR-swizzled by n & 15
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int nl = nt * 32 + j, slot = (8 * rt + 2 * q4 + kb) ^ (nl & 15);
                    *reinterpret_cast<p4i_t*>(red + nl * 256 + slot * 16) =
                        (p4i_t){acc[rt][nt][4 * q4], acc[rt][nt][4 * q4 + 1], acc[rt][nt][4 * q4 + 2], acc[rt][nt][4 * q4 + 3]};
                }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16 / P_NWV; ++i) {
        const int idx = tid + 64 * P_NWV * i, nl = idx >> 4, sl = idx & 15;
        const unsigned char* p = plds + nl * 256 + ((sl ^ (nl & 15)) * 16);
        p4i_t v = *reinterpret_cast<const p4i_t*>(p);
#pragma unroll
        for (int ww = 1; ww < P_NWV; ++ww) v += *reinterpret_cast<const p4i_t*>(p + ww * 16384);
        const int n = n0 + nl, o = row0 + sl * 4;
        if (n < a.N && o < Mw) {
            const float ls = a.colv[n], hlb = __fmul_rn(a.colv[a.Npad + n], 0.5f);
            // the biased operand bytes of 3- / 4-bit rows added BIASB x (every half-table entry of row n) to each total
            const int bias = BIASB ? BIASB * __float_as_int(a.colv[2 * a.Npad + n]) : 0;
            float r[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = v[e] - bias;
                if (DUMP) a.dump[(size_t)n * Mw + o + e] = c;
                const float t = __fmul_rn((float)c, 0.5f);
                const float x = __fadd_rn(__fmul_rn(t, ls), hlb);
                r[e] = __fmul_rn(x, q_ld_scale(M.SC, a.sc_f16, (o + e) / (Mw / s.m_groups)));
            }
            if (a.out_f16) {
                const __half2 h0 = __floats2half2_rn(r[0], r[1]), h1 = __floats2half2_rn(r[2], r[3]);
                *reinterpret_cast<uint2*>(reinterpret_cast<__half*>(M.C) + (size_t)n * Mw + o) =
                    make_uint2(*reinterpret_cast<const uint32_t*>(&h0), *reinterpret_cast<const uint32_t*>(&h1));
            } else {
                *reinterpret_cast<p4f_t*>(reinterpret_cast<float*>(M.C) + (size_t)n * Mw + o) = (p4f_t){r[0], r[1], r[2], r[3]};
            }
        }
    }
}

bool gemm_planes_us_supported(const Shape& s) {
    return s.lay == 2 && s.bits >= 1 && s.bits <= 4 && s.K % 64 == 0 && s.Mw % 4 == 0 && s.m_groups >= 1 && s.ags == s.K && s.Mw % s.m_groups == 0;
}

bool gemm_planes_supported(const Shape& s) {
    if (s.lay != 2 || s.bits < 1 || s.bits > 4 || s.K % 64 != 0 || s.Mw % 4 != 0) return false;
    if (s.m_groups >= 1) return gemm_planes_us_supported(s);             // unified scale: k_gemm_planes_us
    const int apg = s.gs / 64;
    return s.ags == 64 && s.gs >= 64 && s.gs % 64 == 0 && s.K % s.gs == 0 && (apg & (apg - 1)) == 0;
}

hipError_t launch_gemm_planes(const Gemm2Args& a_in, hipStream_t st) {
    if (!gemm_planes_supported(a_in.s) || a_in.nmat < 1 || a_in.nmat > 4 || (a_in.dump && a_in.nmat != 1)) return hipErrorInvalidValue;
    if (a_in.N < 1 || a_in.Npad % 64 != 0 || a_in.Npad < ((a_in.N + 63) & ~63)) return hipErrorInvalidValue;
    // buffer instructions address with 32-bit byte offsets: the LUT image and every weight matrix must stay below 2 GB
    if ((size_t)2 * a_in.s.K * a_in.Npad >= ((size_t)1 << 31)) return hipErrorInvalidValue;
    for (int i = 0; i < a_in.nmat; ++i)
        if ((size_t)((a_in.m[i].Mw + 3) / 4) * ((a_in.s.K / 32 + 63) / 64) * a_in.s.bits * 1024 >= ((size_t)1 << 31)) return hipErrorInvalidValue;
    Gemm2Args a = a_in;
    int gx = 0;
    for (int i = 0; i < a.nmat; ++i) {
        if (a.m[i].Mw % 4 != 0) return hipErrorInvalidValue;
        gx += (a.m[i].Mw + 63) / 64;
        a.m[i].wg_end = gx;
    }
    a.apg_shift = 0;
    while ((64 << a.apg_shift) < a.s.gs) ++a.apg_shift;
    a.gx = gx;
    a.gy = (a.N + 63) / 64;
    // form: one tile per CU at most -> eight waves on it; more -> two four-wave workgroups per CU (see PForm)
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 256;
        n_cu = v;
    }
    // (rounds 3-5: 4-bit weights took the four-wave form only from four tiles per CU on -- its 16 operand-row copies cost them two-way bank
    // conflicts and a shift per gather, twice as many as 2-bit weights have; with 32 copies in both forms one rule serves every width:
    // q/k/v W4 at N = 256, three tiles per CU, 8.20 -> 7.88 ms per 256 tokens with the four-wave form)
    const int nwv = a.form == 1 ? 8 : a.form == 2 ? 4 : (gx * a.gy > n_cu ? 4 : 8);
    dim3 g(((gx + 7) & ~7) * a.gy), b(64 * nwv);
    const bool us = a.s.m_groups >= 1;
    const int lds_bytes = nwv == 8 ? (us ? PFormU<8>::LDS_BYTES : PForm<8>::LDS_BYTES) : (us ? PFormU<4>::LDS_BYTES : PForm<4>::LDS_BYTES);
#define PLAUNCH(KERNEL) do { \
        static bool attr_set = false; \
        if (!attr_set) { \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
            if (e != hipSuccess) return e; \
            attr_set = true; \
        } \
        hipLaunchKernelGGL((KERNEL), g, b, lds_bytes, st, a); } while (0)
#define PLW(...) do { if (nwv == 8) PLAUNCH((__VA_ARGS__, 8>)); else PLAUNCH((__VA_ARGS__, 4>)); } while (0)
    if (a.s.m_groups >= 1) {
#define PLU(B) do { if (a.dump) PLW(k_gemm_planes_us<B, true); else PLW(k_gemm_planes_us<B, false); } while (0)
        switch (a.s.bits) {
            case 1: PLU(1); break;
            case 2: PLU(2); break;
            case 3: PLU(3); break;
            default: PLU(4); break;
        }
#undef PLU
        return hipGetLastError();
    }
#define PL3(B, Z, D) do { if (a.sc_f16) PLW(k_gemm_planes<B, Z, D, true); else PLW(k_gemm_planes<B, Z, D, false); } while (0)
#define PL2(B, Z) do { if (a.dump) PL3(B, Z, true); else PL3(B, Z, false); } while (0)
    switch (a.s.bits) {
        case 1: if (a.s.zero_point) PL2(1, true); else PL2(1, false); break;
        case 2: if (a.s.zero_point) PL2(2, true); else PL2(2, false); break;
        case 3: if (a.s.zero_point) PL2(3, true); else PL2(3, false); break;
        default: if (a.s.zero_point) PL2(4, true); else PL2(4, false); break;
    }
#undef PL2
#undef PL3
#undef PLW
#undef PLAUNCH
    return hipGetLastError();
}

}  // namespace tmac
