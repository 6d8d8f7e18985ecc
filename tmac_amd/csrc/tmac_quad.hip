// tmac_quad.hip — k_gemv_quad: fused LUT build + GEMV (N = 1 decode), a wave owns a row quad.  Production path.
//
// Reference semantics: lut_ctor.cc:38-266 (LUT build, bit-exact), tbl.cc:435-529 / 586-628 (lookup, exact integer sums,
// per-act-group fp32 scale-apply), qgemm.py:170-174,192-206 (bit-plane combine, scale-final).  Work decomposition:
// a WAVE owns a row quad (4 output rows): its 64 lanes hold 64 consecutive 8-table units (QUAD layout,
// tmac_layout.h), so every weight instruction reads 1 KiB contiguous, the LUT in LDS is read conflict-free (lane u
// reads 16 B at u*16), and the reduction over K is wave-local (DPP within rows of 16 lanes + 2 cross-row moves).
//   WPQ = 1     : one wave does all of K for its quad                       (many quads: q/k/v, gate/up)
//   WPQ = 2,3,4 : that many waves split the K steps of a quad, combined in LDS  (few quads: o, down)
// Workgroups (512 / 768 / 1024 threads) are persistent, build the LUT once while their first weight fragments are in
// flight, then walk their (quad, step) list.  Template parameters: BITS 2|4; ZP zero points; SM 0 per-group scales /
// 2 unified scale applied last (BitNet); LUTSRC 1 build the LUT in-kernel / 0 copy the image k_preprocess wrote;
// NR tables built per thread; FT threads; WPQ waves per quad; DUMP integer tap; ACC 1 v_mfma_i32_16x16x64_i8
// accumulate / 0 v_mqsad_pk_u16_u8 (A/B variant).  DESIGN.md 4.1, 4.2, 4.6.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "tmac_core.h"
#include "tmac_kernels.h"
#include "tmac_fastdiv.h"
#include "tmac_quad_core.h"

namespace tmac {


// weights of (local quad lq, step st) + the lane's scales (rows beta0, beta0+1 of its unit's scale group)
template <int BITS, bool ZP, int SM, int ACC, bool SCF16>
__device__ __forceinline__ void load_q(QFrag<BITS>& f, const FusedArgs& a, const FusedMat& M, int lq, int st, int nst, int lane) {
    constexpr int NJ = 8 * BITS / 8;
    constexpr int per = ZP ? 2 : 1;
    const int u = st * 64 + lane;
    // Scale words go through scalar locals and are stored to f.sraw with constant subscripts at the end: stores to
    // different elements in the two dtype branches get sunk into one store with a run-time subscript, which
    // keeps the whole fragment in scratch memory.
    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    if (SM == 0 && ACC == 1) {
        // epilogue role of this lane: row beta = lane & 3, units st*64 + 16g + 4*lg .. +3 (g = (lane & 15) >> 2, lg = lane >> 4)
        // sraw[2*gi], sraw[2*gi+1]: scale (, zero) of act-group pair gi; f16 packs both into sraw[2*gi]
        // Loads are unconditional with clamped indices (no exec-mask branches around them); units past K read zero
        // tables from LDS, so what the clamped lanes load never reaches a result.
        // scale group of unit st*64 + c: st * (64 >> gs_shift) + (c >> gs_shift)  (a scale group never straddles a
        // step: gs <= 2048); the lane part is loop-invariant, the step part scalar, the address a uniform base plus a
        // 32-bit lane offset
        const int c0 = 4 * (lane & 12) + 4 * (lane >> 4);
        const uint32_t sg_step = (uint32_t)st * (64u >> a.gs_shift);
        const uint32_t row0 = (uint32_t)lq * (uint32_t)a.nsg;
        const char* scb = reinterpret_cast<const char*>(M.SC);
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            uint32_t v0 = 0, v1 = 0;
            if (gi == 0 || a.gs_shift < 2) {      // gs >= 128: both act-group pairs of the lane share the scale group (the epilogue reads pair 0)
                const uint32_t sg = min(sg_step + (uint32_t)((c0 + 2 * gi) >> a.gs_shift), (uint32_t)a.nsg - 1u);
                const uint32_t sidx = ((row0 + sg) * 4 + (lane & 3)) * per;
                if (SCF16) {
                    const char* ph = scb + (size_t)(sidx * 2u);
                    if (ZP) v0 = *reinterpret_cast<const uint32_t*>(ph);
                    else v0 = *reinterpret_cast<const unsigned short*>(ph);
                } else {
                    const uint32_t* p32 = reinterpret_cast<const uint32_t*>(scb + (size_t)(sidx * 4u));
                    v0 = p32[0];
                    if (ZP) v1 = p32[1];
                }
            }
            if (gi == 0) { r0 = v0; r1 = v1; } else { r2 = v0; r3 = v1; }
        }
    }
    if (ACC == 1 || u < a.nu) {   // nst*64 lanes of every step exist in the QUAD layout (zero padded)
        const uint4* wp = M.W + (size_t)((uint32_t)(lq * nst + st) * (uint32_t)(NJ * 64)) + lane;   // uniform base + lane
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const u32x4q v = __builtin_nontemporal_load(reinterpret_cast<const u32x4q*>(wp + (size_t)j * 64));
            f.wd[4 * j] = v.x; f.wd[4 * j + 1] = v.y; f.wd[4 * j + 2] = v.z; f.wd[4 * j + 3] = v.w;
        }
        if (SM == 0 && ACC == 0) {
            // rows beta0, beta0+1 of this unit's scale group: fp32 -> one word per element, f16 -> two per word
            const int sg = u >> a.gs_shift;
            const uint32_t sidx = (((uint32_t)lq * (uint32_t)a.nsg + (uint32_t)sg) * 4 + 2 * (lane & 1)) * per;
            if (SCF16) {
                const uint32_t* p32 = reinterpret_cast<const uint32_t*>(reinterpret_cast<const __half*>(M.SC) + sidx);
                r0 = p32[0];
                if (ZP) r1 = p32[1];
            } else {
                const uint32_t* p32 = reinterpret_cast<const uint32_t*>(M.SC) + sidx;
                r0 = p32[0]; r1 = p32[1];
                if (ZP) { r2 = p32[2]; r3 = p32[3]; }
            }
        }
    }
    if (SM == 0) { f.sraw[0] = r0; f.sraw[1] = r1; f.sraw[2] = r2; f.sraw[3] = r3; }
}


// ACC 0: v_mqsad_pk_u16_u8 accumulate (VALU).  ACC 1: v_mfma_i32_16x16x64_i8 accumulate (matrix pipe), see
// k_gemv_fused for the operand construction; with 64 lanes = 64 units of one quad, source lane l = 16g + i and
// D[i][4g+beta] lands in lane l' = 16*(i/4) + 4g + beta, register i%4: lane l' owns output row beta and the four
// units 16g + 4*(l'/16) .. +3 of the step (two act groups, one 128-wide scale group).
template <int BITS, bool ZP, int SM, int LUTSRC, int NR, int FT, int WPQ, bool DUMP, int ACC, bool SCF16, bool EARLY>
__global__ __launch_bounds__(FT) void k_gemv_quad(FusedArgs a) {
    extern __shared__ uint4 lds[];
    const unsigned long long t_entry = DUMP ? __builtin_amdgcn_s_memtime() : 0ull;   // before the first kernel-argument load
    constexpr int NWV = FT / 64, IPI = NWV / WPQ;
    const Shape& s = a.s;
    // the wave index is uniform but the compiler cannot prove it from threadIdx: readfirstlane moves every
    // per-wave quantity (quad, step, base pointers, loop control) to SGPRs / the scalar ALU
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int n = blockIdx.y;
    const int T = s.K / 4, nu = a.nu, G = a.G, nst = (nu + 63) >> 6;
    const int tstride = nst * 64 + 1;                            // whole steps: units past K hold zero tables
    uint4* tab = lds;                                            // [4][tstride]
    const int GP = nst * 32;                                     // act groups of the padded steps (>= G)
    // MFMA accumulate with per-group scales: the epilogue works on the plane-combined integer (see compute_mfma), which
    // wants ls / 2 and lb / 2 -- stored halved (exact) instead of being halved per use
    constexpr bool HALVES = (ACC == 1 && SM != 2);
    constexpr float LSK = HALVES ? 0.5f : 1.0f;
    float* l_ls = reinterpret_cast<float*>(lds + 4 * tstride);   // [GP]  (groups past K: 0)
    float* l_lb = l_ls + GP;                                     // [GP]
    float* l_red = l_lb + GP;                                    // [2][NWV][4][4] partials (WPQ == 2 / SM 2)
    float* l_scr = l_red + 2 * NWV * 16;                         // SM 2 build scratch: [NWV] maxima + [T/8] chunk sums
    // Per-matrix fields: a cursor (one for the prefetch, one for the consumption) caches the current matrix' pointers
    // and quad range in SGPRs and advances when a quad index leaves the range (quads only grow).  Looking the matrix
    // up per step — run-time indexing of a.m[], or select chains over four copies — costs tens of scalar
    // instructions per step, and the scalar pipe issues no faster than the vector pipe.
    const int nmat = a.nmat;
    const int total_q = a.m[nmat - 1].nb_end;                    // cumulative QUAD counts
    struct MatCur { FusedMat m; int base, mi, end; };          // end: first quad past this matrix (INT_MAX for the last)
    auto seek = [&](MatCur& c, int gq) {
        while (gq >= c.end) { c.base = c.m.nb_end; ++c.mi; c.m = a.m[c.mi]; c.end = (c.mi + 1 < nmat) ? c.m.nb_end : 0x7fffffff; }
    };
    MatCur pc = {a.m[0], 0, 0, nmat > 1 ? a.m[0].nb_end : 0x7fffffff}, cc = pc;

#define QSTAMP(i) do { if (DUMP && a.stamps && lane == 0 && (w == 0 || w == NWV - 1)) a.stamps[((size_t)blockIdx.x * 2 + (w ? 1 : 0)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    QSTAMP(0);
    if (DUMP && a.stamps && lane == 0 && (w == 0 || w == NWV - 1)) a.stamps[((size_t)blockIdx.x * 2 + (w ? 1 : 0)) * 8 + 7] = t_entry;

    // ---- 1. activation loads for the LUT build (issued first: vmcnt retires in order) ----------
    // A lane builds the two consecutive tables 2p, 2p+1 (8 activations, one 16-byte fp16 load) of pair p = r*FT + tid:
    // one uint4 LDS store, and the act-group scale (abs-max over 8 lanes, two exact divisions) is computed once per two
    // tables.
    constexpr int NP = NR / 2;                                   // pairs per thread
    const int P = T / 2;
    uint32_t xr[NP][8];
    if (LUTSRC == 1) {
#pragma unroll
        for (int r = 0; r < NP; ++r) {
            const int p = min(r * FT + tid, P - 1);     // clamped, not predicated (see load_q)
            if (a.act_f16) {
                const uint4 v = reinterpret_cast<const uint4*>(reinterpret_cast<const __half*>(a.B) + (size_t)n * s.K)[p];
                xr[r][0] = v.x; xr[r][1] = v.y; xr[r][2] = v.z; xr[r][3] = v.w;
            } else {
                const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(a.B) + (size_t)n * s.K) + 2 * (size_t)p;
                const uint4 v0 = src[0], v1 = src[1];
                xr[r][0] = v0.x; xr[r][1] = v0.y; xr[r][2] = v0.z; xr[r][3] = v0.w;
                xr[r][4] = v1.x; xr[r][5] = v1.y; xr[r][6] = v1.z; xr[r][7] = v1.w;
            }
        }
    }
    auto unpack = [&](int r, float (&x)[8]) {
        if (a.act_f16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const __half2 hh = *reinterpret_cast<const __half2*>(&xr[r][i]);
                x[2 * i] = __low2float(hh); x[2 * i + 1] = __high2float(hh);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __uint_as_float(xr[r][i]);
        }
    };

    // ---- 2. this wave's work: quads slot, slot + stride, ...; steps h, h + WPQ, ... of each ------
    const int slot0 = blockIdx.x * IPI + w / WPQ, h = w % WPQ, stride = gridDim.x * IPI;
    constexpr int RING = (BITS <= 2) ? 4 : 2;      // weight fragments in flight per wave (register budget)
    QFrag<BITS> f0, f1, f2, f3;
    int p_q = (h < nst) ? slot0 : total_q, p_st = h;    // prefetch cursor (a wave without steps, WPQ > nst, never issues)
    auto issue = [&](QFrag<BITS>& f) {
        if (p_q < total_q) {
            seek(pc, p_q);
            load_q<BITS, ZP, SM, ACC, SCF16>(f, a, pc.m, p_q - pc.base, p_st, nst, lane);
            p_st += WPQ;
            if (p_st >= nst) { p_st = h; p_q += stride; }
        }
    };
    // When the weight loads are issued matters more than anything else in this kernel (measured, profiles/r01_tune_quad.txt):
    // issued first, as a prefetch, they occupy the CU's 64 B/clk load path for 1-2 us while the VALU idles, and the LUT
    // build cannot start before the wave's own loads have all returned (its s_waitcnt is vmcnt(0): loads sit under
    // wave-dependent branches).  So: wait for the activation registers alone (the empty asm reads them, the compiler
    // puts its s_waitcnt in front), THEN issue the fragments — the LUT build below is pure VALU and overlaps the weight
    // stream — or, with two workgroups per CU competing for the load path, after the LUT build.
    constexpr bool early = LUTSRC == 1 && EARLY;      // chosen by the launcher: at most one workgroup per CU
    if (LUTSRC == 1) {
#pragma unroll
        for (int r = 0; r < NP; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < 4 || !a.act_f16) asm volatile("" :: "v"(xr[r][i]));
    }
    if (early) {
        issue(f0); issue(f1);
        if (RING == 4) { issue(f2); issue(f3); }
    }
    QSTAMP(1);

    // ---- 3. LUT into LDS (all FT threads) ------------------------------------------------------
    if (LUTSRC == 0) {
        const uint4* src = reinterpret_cast<const uint4*>(a.qlut_lds) + (size_t)n * 4 * a.tstride;   // image stride (k_preprocess)
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4)
            for (int u = tid; u < nu; u += FT) {
                uint4 v = src[j4 * a.tstride + u];
                if (ACC == 1) { v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u; }
                tab[j4 * tstride + u] = v;
            }
        if (SM == 2) { if (tid == 0) { l_ls[0] = a.lut_scales[n]; l_lb[0] = a.lut_biases[n]; } }
        else for (int i = tid; i < G; i += FT) { l_ls[i] = __fmul_rn(LSK, a.lut_scales[(size_t)n * G + i]); l_lb[i] = __fmul_rn(LSK, a.lut_biases[(size_t)n * G + i]); }
    } else {
        float gscale = 0.f, gtinv = 0.f;
        if (SM == 2) {
            // One act group = the whole row: the scale needs a maximum over K, and lut_biases is ONE fp32 chain over the K/32
            // 8-table chunk sums in order (lut_ctor.cc:157,218) -- 270 dependent adds for K = 8640, a microsecond of a single
            // lane.  Neither the chunk sums nor the chain depend on the scale, so: pass 1 computes maxima AND chunk sums
            // (LUT[0] of a table is -(((x0+x1)+x2)+x3), the same adds q_table8 performs), one barrier, then lane 0 of the last
            // wave walks the chain while every other wave already builds its tables.
            float mx = 0.f;
#pragma unroll
            for (int r = 0; r < NP; ++r) {
                const int p = r * FT + tid;
                if (p < P) {
                    float x[8];
                    unpack(r, x);
                    mx = fmaxf(mx, __fadd_rn(__fadd_rn(fabsf(x[0]), fabsf(x[1])), __fadd_rn(fabsf(x[2]), fabsf(x[3]))));
                    mx = fmaxf(mx, __fadd_rn(__fadd_rn(fabsf(x[4]), fabsf(x[5])), __fadd_rn(fabsf(x[6]), fabsf(x[7]))));
                    float va = -__fadd_rn(__fadd_rn(__fadd_rn(x[0], x[1]), x[2]), x[3]);
                    float vb = -__fadd_rn(__fadd_rn(__fadd_rn(x[4], x[5]), x[6]), x[7]);
                    va = __fadd_rn(va, qdpp_f<0x4E>(va));      // lane ^ 2: v0+v4 | v2+v6      (lut_ctor.cc:25-31)
                    vb = __fadd_rn(vb, qdpp_f<0x4E>(vb));      //           v1+v5 | v3+v7
                    va = __fadd_rn(va, qdpp_f<0xB1>(va));      // lane ^ 1: (v0+v4)+(v2+v6)
                    vb = __fadd_rn(vb, qdpp_f<0xB1>(vb));      //           (v1+v5)+(v3+v7)
                    if ((p & 3) == 0) l_scr[NWV + (p >> 2)] = __fadd_rn(va, vb);
                }
            }
            mx = q_row_allmax(mx);
            mx = q_xor_max_f(mx);
            if (lane == 0) l_scr[w] = mx;
            __syncthreads();
            mx = l_scr[0];
#pragma unroll
            for (int i = 1; i < NWV; ++i) mx = fmaxf(mx, l_scr[i]);
            gscale = __fdiv_rn(mx, 127.0f);
            gtinv = (gscale != 0.0f) ? __fdiv_rn(1.0f, gscale) : 0.0f;
            if (tid == FT - 64) {   // the last wave holds the fewest pairs (none when K/8 <= FT - 64)
                // the LDS reads are batched (16-byte reads, unrolled) so that only the dependent adds are serial
                float biases = 0.0f;
                const float4* cs = reinterpret_cast<const float4*>(l_scr + NWV);
                const int nc = T / 8;
                int c = 0;
#pragma unroll 4
                for (; c + 4 <= nc; c += 4) {
                    const float4 v4 = cs[c >> 2];
                    biases = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(biases, v4.x), v4.y), v4.z), v4.w);
                }
                for (; c < nc; ++c) biases = __fadd_rn(biases, l_scr[NWV + c]);
                l_ls[0] = gscale;
                l_lb[0] = biases;
            }
        }
#pragma unroll
        for (int r = 0; r < NP; ++r) {
            const int p = r * FT + tid;
            if (p < P) {   // T % 16 == 0: the 8 lanes of an act group are valid or invalid as a whole
                float x[8];
                unpack(r, x);
                float scales, t_scales;
                if (SM == 2) { scales = gscale; t_scales = gtinv; }
                else {
                    const float s0 = __fadd_rn(__fadd_rn(fabsf(x[0]), fabsf(x[1])), __fadd_rn(fabsf(x[2]), fabsf(x[3])));
                    const float s1 = __fadd_rn(__fadd_rn(fabsf(x[4]), fabsf(x[5])), __fadd_rn(fabsf(x[6]), fabsf(x[7])));
                    const float mx = q_half_allmax(fmaxf(s0, s1));
                    scales = div127(mx);
                    t_scales = (scales != 0.0f) ? rcp_exact(scales) : 0.0f;
                }
                uint32_t lo0, hi0, lo1, hi1;
                float La, Lb;
                q_table8<ACC == 1>(x[0], x[1], x[2], x[3], t_scales, lo0, hi0, La);
                q_table8<ACC == 1>(x[4], x[5], x[6], x[7], t_scales, lo1, hi1, Lb);
                // tables 2p, 2p+1 = the uint4 (j4 = p & 3) of unit p >> 2
                tab[(p & 3) * tstride + (p >> 2)] = make_uint4(lo0, hi0, lo1, hi1);
                // lut_biases (lut_ctor.cc:25-31): per 8-table chunk ((v0+v4)+(v2+v6)) + ((v1+v5)+(v3+v7)), v_i = LUT[0] of table i
                // = -L15; the 4 lanes of a chunk hold (v0,v1), (v2,v3), (v4,v5), (v6,v7)
                float va = -La, vb = -Lb;
                va = __fadd_rn(va, qdpp_f<0x4E>(va));      // lane ^ 2: v0+v4 | v2+v6
                vb = __fadd_rn(vb, qdpp_f<0x4E>(vb));      //           v1+v5 | v3+v7
                va = __fadd_rn(va, qdpp_f<0xB1>(va));      // lane ^ 1: (v0+v4)+(v2+v6)
                vb = __fadd_rn(vb, qdpp_f<0xB1>(vb));      //           (v1+v5)+(v3+v7)
                const float v = __fadd_rn(va, vb);
                if (SM != 2) {
                    const float c1 = qdpp_f<0x104>(v);      // row_shl:4: the second chunk of the act group
                    if ((p & 7) == 0) {
                        l_ls[p >> 3] = __fmul_rn(LSK, scales);
                        l_lb[p >> 3] = __fmul_rn(LSK, __fadd_rn(__fadd_rn(0.0f, v), c1));
                    }
                }
            }
        }
    }
    {   // zero tables for the units between K and the end of the last 64-unit step
        const uint32_t z = (ACC == 1) ? 0u : 0x80808080u;
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4)
            for (int u = nu + tid; u < nst * 64; u += FT) tab[j4 * tstride + u] = make_uint4(z, z, z, z);
        if (SM != 2)
            for (int i = G + tid; i < GP; i += FT) { l_ls[i] = 0.f; l_lb[i] = 0.f; }
    }
    QSTAMP(2);
    if (!early) {
        issue(f0); issue(f1);
        if (RING == 4) { issue(f2); issue(f3); }
    }
    __syncthreads();
    QSTAMP(3);
    if (a.lut_tap && blockIdx.x == 0) {
        for (int i = tid; i < (SM == 2 ? 1 : G); i += FT) { a.lut_tap[(size_t)n * 2 * G + i] = __fmul_rn(1.0f / LSK, l_ls[i]); a.lut_tap[(size_t)n * 2 * G + G + i] = __fmul_rn(1.0f / LSK, l_lb[i]); }
    }

    // ---- 4. stream this wave's quads -----------------------------------------------------------
    float cacc[2][BITS];
    int32_t iacc[BITS][4];
    auto reset_acc = [&]() {
#pragma unroll
        for (int pl = 0; pl < BITS; ++pl) {
            cacc[0][pl] = 0.f; cacc[1][pl] = 0.f;
#pragma unroll
            for (int be = 0; be < 4; ++be) iacc[pl][be] = 0;
        }
    };
    reset_acc();
    const int beta0 = 2 * (lane & 1);

    auto compute = [&](const QFrag<BITS>& f, int st, int Mw_m, int lq) {
        const int u = st * 64 + lane;
        if (u >= nu) return;                 // nu is even: both lanes of a pair are valid or not
        uint32_t tb[16];
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
            const uint4 v = tab[j4 * tstride + u];
            tb[4 * j4] = v.x; tb[4 * j4 + 1] = v.y; tb[4 * j4 + 2] = v.z; tb[4 * j4 + 3] = v.w;
        }
        SegAcc<BITS, 0> acc;
        acc.reset();
        accumulate_tables<BITS, 0, 8>(f.wd, tb, acc);
        if (SM == 2) {
#pragma unroll
            for (int pl = 0; pl < BITS; ++pl)
#pragma unroll
                for (int be = 0; be < 4; ++be) iacc[pl][be] += acc.ps(pl, be, 8);
            return;
        }
        const int kk = u >> 1;
        const float ls = l_ls[kk], lb = l_lb[kk];
#pragma unroll
        for (int pl = 0; pl < BITS; ++pl) {
            uint32_t lo = (uint32_t)acc.a[pl], hi = (uint32_t)(acc.a[pl] >> 32);
            lo += qdpp_u<0xB1>(lo);          // the two 8-table halves of the act group: lanes (l, l^1)
            hi += qdpp_u<0xB1>(hi);
            const uint32_t mine = (lane & 1) ? hi : lo;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int32_t ps = 127 * 16 - (int32_t)((mine >> (16 * i)) & 0xffff);
                if (DUMP && a.dump) {
                    const int o = 4 * lq + beta0 + i;
                    if (o < Mw_m) a.dump[((size_t)n * Mw_m * BITS + mrow(o, pl, BITS)) * G + kk] = ps;
                }
                const float v = (pl == 0) ? __fmaf_rn((float)ps, ls, lb) : __fmul_rn((float)ps, ls);
                float c = __fmaf_rn(v, qfrag_scale<ZP>(f.sraw, SCF16, i, 0), cacc[i][pl]);
                if (ZP && pl == 0) c = __fmaf_rn(qfrag_scale<ZP>(f.sraw, SCF16, i, 1), __fmul_rn(2.0f, lb), c);
                cacc[i][pl] = c;
            }
        }
    };

    // ---- ACC == 1 ------------------------------------------------------------------------------
    qv4i_t bsel;
    {
        const int jrel = (lane & 15) - 4 * (lane >> 4);
        const uint32_t be = (jrel >= 0 && jrel < 4) ? (0x01u << (8 * jrel)) : 0u, bo = (jrel >= 0 && jrel < 4) ? (0xfeu << (8 * jrel)) : 0u;   // +1 | -2
        bsel = (qv4i_t){(int)be, (int)bo, (int)be, (int)bo};
    }
    uint32_t k3 = 0x03020100u;
    asm volatile("" : "+v"(k3));     // keep the selector constant in a VGPR (operand of v_and_or_b32)
    auto compute_mfma = [&](const QFrag<BITS>& f, int st, int Mw_m, int lq) {
        const int u = st * 64 + lane;
        uint32_t tb[16];
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
            const uint4 v = tab[j4 * tstride + u];            // units past K read the zero tables: no contribution
            tb[4 * j4] = v.x; tb[4 * j4 + 1] = v.y; tb[4 * j4 + 2] = v.z; tb[4 * j4 + 3] = v.w;
        }
        qv4i_t c[BITS];
#pragma unroll
        for (int pl = 0; pl < BITS; ++pl) c[pl] = (qv4i_t){0, 0, 0, 0};
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
#pragma unroll
            for (int pl = 0; pl < BITS; ++pl) {
                uint32_t pa, ma, pb, mb;
                const int qa = (2 * tp) * BITS + pl, qb = (2 * tp + 1) * BITS + pl;
                if (qa & 1) q_lookup4_pm<1>(f.wd[qa >> 1], tb[4 * tp], tb[4 * tp + 1], k3, pa, ma);
                else q_lookup4_pm<0>(f.wd[qa >> 1], tb[4 * tp], tb[4 * tp + 1], k3, pa, ma);
                if (qb & 1) q_lookup4_pm<1>(f.wd[qb >> 1], tb[4 * tp + 2], tb[4 * tp + 3], k3, pb, mb);
                else q_lookup4_pm<0>(f.wd[qb >> 1], tb[4 * tp + 2], tb[4 * tp + 3], k3, pb, mb);
                c[pl] = __builtin_amdgcn_mfma_i32_16x16x64_i8((qv4i_t){(int)pa, (int)ma, (int)pb, (int)mb}, bsel, c[pl], 0, 0, 0);
            }
        }
        if (SM == 2) {      // unified scale: keep the exact integer sum of this lane's row (lane & 3), all units
            // The MFMA results must have landed before a VALU instruction reads them (no hardware interlock: up to 18 wait states after an
            // 8-pass MFMA; with one bit-plane the compiler's hazard recogniser left the two a single wait state apart across the loop branch and
            // W1 unified-scale results were wrong).  ONE wait for all planes, behind the last MFMA of the step (the planes' chains are
            // interleaved, so the others finished earlier):
            // a wait per plane cost (BITS - 1) x 19 idle cycles per item
            if constexpr (BITS == 1) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(c[0]));
            else if constexpr (BITS == 2) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(c[0]), "+v"(c[1]));
            else if constexpr (BITS == 3) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]));
            else asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
#pragma unroll
            for (int pl = 0; pl < BITS; ++pl) iacc[pl][0] += (c[pl].x + c[pl].y) + (c[pl].z + c[pl].w);
            return;
        }
        const int ub4 = st * 64 + 4 * (lane & 12) + 4 * (lane >> 4);
        const int o = 4 * lq + (lane & 3);
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int ug = ub4 + 2 * gi;
            {   // act groups past K have zero tables and zero LUT scale/bias: they add exactly 0, no guard needed
                const int kk = ug >> 1;
                const float hls = l_ls[kk], hlb = l_lb[kk];           // ls / 2, lb / 2 (HALVES)
                const bool first = (gi == 0) || (a.gs_shift >= 2);   // static register indices only: a run-time
                float sc, zr = 0.f;                                   // subscript would put sraw in scratch memory
                if (SCF16) {
                    const uint32_t wv = first ? f.sraw[0] : f.sraw[2];
                    sc = __half2float(__ushort_as_half((unsigned short)(wv & 0xffff)));
                    if (ZP) zr = __half2float(__ushort_as_half((unsigned short)(wv >> 16)));
                } else {
                    sc = __uint_as_float(first ? f.sraw[0] : f.sraw[2]);
                    if (ZP) zr = __uint_as_float(first ? f.sraw[1] : f.sraw[3]);
                }
                // Bit-planes are combined as integers before the one conversion and scale chain of the act group:
                //   sum_p alpha_p [(ps_p ls + [p = 0] lb) scale + [p = 0] zero 2 lb]        (tbl.cc:479-526, kernels.cc:1068)
                // = ((sum_p 2^p ps_p) (ls / 2) + lb / 2) scale + (2 zero) (lb / 2)          alpha_p = 2^(p-1); |sum| < 2^15
                // -- the same real number with fewer roundings (fp32 contract 1e-3; the integer sums stay the tap).
                int32_t comb = 0;
#pragma unroll
                for (int pl = BITS - 1; pl >= 0; --pl) {
                    const int32_t ps = (gi == 0) ? (c[pl].x + c[pl].y) : (c[pl].z + c[pl].w);
                    if (DUMP && a.dump && o < Mw_m && ug < nu) a.dump[((size_t)n * Mw_m * BITS + mrow(o, pl, BITS)) * G + kk] = ps;
                    comb = (pl == BITS - 1) ? ps : (int32_t)(((uint32_t)comb << 1) + (uint32_t)ps);   // Horner: v_lshl_add_u32
                }
                const float v = __fmaf_rn((float)comb, hls, hlb);
                float cc = __fmaf_rn(v, sc, cacc[0][0]);
                if (ZP) cc = __fmaf_rn(__fadd_rn(zr, zr), hlb, cc);
                cacc[0][0] = cc;
            }
        }
    };

    // reduce one quad over the 64 lanes (same-parity lanes of a row by DPP, rows by ds_bpermute), combine the
    // WPQ waves through LDS (double-buffered by iteration parity), store 4 outputs
    int parity = 0;
    auto finish_quad = [&](bool have, const FusedMat& M, int lq) {
        float* red = l_red + parity * (NWV * 16);
        if (ACC == 1 && SM != 2) {   // per-group scales: fp32 partials
            float acc = 0.f;
            if (have) {
                acc = cacc[0][0];                             // planes already combined (compute_mfma)
                acc = __fadd_rn(acc, qdpp_f<0x124>(acc));     // lanes with the same beta: rotate by 4, 8 within the row
                acc = __fadd_rn(acc, qdpp_f<0x128>(acc));
                acc = q_xor_add_f(acc);
            }
            if (WPQ == 1) {
                const int o = 4 * lq + lane;
                if (have && lane < 4 && o < M.Mw) q_st_out(M.C, a.out_f16, (size_t)n * M.Mw + o, acc);
            } else {
                if (lane < 4) red[w * 4 + lane] = acc;
                __syncthreads();
                if (have && h == 0 && lane < 4) {
                    float t = red[w * 4 + lane];
#pragma unroll
                    for (int ww = 1; ww < WPQ; ++ww) t = __fadd_rn(t, red[(w + ww) * 4 + lane]);
                    const int o = 4 * lq + lane;
                    if (o < M.Mw) q_st_out(M.C, a.out_f16, (size_t)n * M.Mw + o, t);
                }
            }
        } else if (SM != 2) {
            float part[2] = {0.f, 0.f};
            if (have) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float acc = __fmul_rn(cacc[i][0], 0.5f);
#pragma unroll
                    for (int pl = 1; pl < BITS; ++pl) acc = __fadd_rn(acc, __fmul_rn(cacc[i][pl], q_alpha(pl)));
                    acc = __fadd_rn(acc, qdpp_f<0x4E>(acc));      // lane ^ 2
                    acc = __fadd_rn(acc, qdpp_f<0x124>(acc));     // rotate by 4, 8 within the 16-lane row
                    acc = __fadd_rn(acc, qdpp_f<0x128>(acc));
                    acc = q_xor_add_f(acc);
                    part[i] = acc;
                }
            }
            if (WPQ == 1) {
                if (have && lane < 2) {
                    const int o = 4 * lq + 2 * lane;
                    if (o < M.Mw) q_st_out(M.C, a.out_f16, (size_t)n * M.Mw + o, part[0]);
                    if (o + 1 < M.Mw) q_st_out(M.C, a.out_f16, (size_t)n * M.Mw + o + 1, part[1]);
                }
            } else {
                if (lane < 2) { red[w * 4 + 2 * lane] = part[0]; red[w * 4 + 2 * lane + 1] = part[1]; }
                __syncthreads();
                if (have && h == 0 && lane < 4) {
                    float t = red[w * 4 + lane];
#pragma unroll
                    for (int ww = 1; ww < WPQ; ++ww) t = __fadd_rn(t, red[(w + ww) * 4 + lane]);
                    const int o = 4 * lq + lane;
                    if (o < M.Mw) q_st_out(M.C, a.out_f16, (size_t)n * M.Mw + o, t);
                }
            }
        } else {
            int32_t* redi = reinterpret_cast<int32_t*>(red);
            int32_t tot[BITS];
#pragma unroll
            for (int pl = 0; pl < BITS; ++pl) {
                if (ACC == 1) {       // every lane holds a partial of row lane & 3: rotate by 4, 8 within the row, then rows
                    uint32_t v = have ? (uint32_t)iacc[pl][0] : 0u;
                    v += qdpp_u<0x124>(v);
                    v += qdpp_u<0x128>(v);
                    v = q_xor_add_u(v);
                    tot[pl] = (int32_t)v;
                } else {
                    int32_t mine = 0;
#pragma unroll
                    for (int be = 0; be < 4; ++be) {
                        int32_t v = have ? iacc[pl][be] : 0;
#pragma unroll
                        for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
                        if (lane == be) mine = v;
                    }
                    tot[pl] = mine;
                }                     // lane be (< 4) holds row be's integer sum of this wave
            }
            if (WPQ > 1) {
                if (lane < 4)
#pragma unroll
                    for (int pl = 0; pl < BITS; ++pl) redi[(w * 4 + lane) * 4 + pl] = tot[pl];
                __syncthreads();
            }
            if (have && h == 0 && lane < 4) {
                const int o = 4 * lq + lane;
                if (o < M.Mw) {
                    float acc = 0.f;
#pragma unroll
                    for (int pl = 0; pl < BITS; ++pl) {
                        int32_t cb = tot[pl];
                        if (WPQ > 1)
#pragma unroll
                            for (int ww = 1; ww < WPQ; ++ww) cb += redi[((w + ww) * 4 + lane) * 4 + pl];
                        if (DUMP && a.dump) a.dump[(size_t)n * M.Mw * BITS + mrow(o, pl, BITS)] = cb;
                        const float t = __fmul_rn((float)cb, q_alpha(pl));
                        acc = (pl == 0) ? t : __fadd_rn(acc, t);
                    }
                    const float v = __fadd_rn(__fmul_rn(acc, l_ls[0]), __fmul_rn(l_lb[0], 0.5f));
                    q_st_out(M.C, a.out_f16, (size_t)n * M.Mw + o, __fmul_rn(v, q_ld_scale(M.SC, a.sc_f16, o / (M.Mw / s.m_groups))));
                }
            }
        }
        parity ^= 1;
        reset_acc();
    };

    // The fragment ring is consumed in issue order.  The (quad, step) work list of this wave is walked by one
    // cursor in a loop unrolled over the ring, so every fragment has a fixed role in the loop body (a run-time ring
    // position makes the compiler merge four control-flow paths with ~40 register copies per step).  A quad is
    // closed (reduce + store, workgroup barrier for WPQ > 1) when the cursor leaves it; every wave closes the
    // same number of quads, with or without work, so the barriers inside finish_quad stay matched.
    int c_it = 0, c_st = h, lq = 0;
    bool have = slot0 < total_q && h < nst;
    if (have) { seek(cc, slot0); lq = slot0 - cc.base; }
    if (blockIdx.x * IPI < total_q) {      // uniform: this workgroup has at least one quad iteration
#define QSTEP(F)                                                                                              \
        while (!(have && c_st < nst)) {                                                                       \
            if (c_it == 0) QSTAMP(4);                                                                         \
            finish_quad(have, cc.m, lq);                                                                      \
            if (c_it == 0) QSTAMP(5);                                                                         \
            ++c_it;                                                                                           \
            if (blockIdx.x * IPI + c_it * stride >= total_q) goto q_done;                                     \
            const int gq = slot0 + c_it * stride;                                                             \
            have = gq < total_q && h < nst;                                                                   \
            if (have) { seek(cc, gq); lq = gq - cc.base; }                                                                   \
            c_st = h;                                                                                         \
        }                                                                                                     \
        if (ACC == 1) compute_mfma(F, c_st, cc.m.Mw, lq); else compute(F, c_st, cc.m.Mw, lq);                   \
        issue(F);                                                                                             \
        c_st += WPQ;
        for (;;) {
            QSTEP(f0)
            QSTEP(f1)
            if (RING == 4) {
                QSTEP(f2)
                QSTEP(f3)
            }
        }
#undef QSTEP
    }
q_done:
    QSTAMP(6);
}

// ---------------------------------------------------------------------------------------------
#if !defined(TMAC_QUAD_BITS) || !defined(TMAC_QUAD_SCF16)
#error "compile with -DTMAC_QUAD_BITS=1..4 -DTMAC_QUAD_SCF16=0|1 (one translation unit per combination keeps the build parallel)"
#endif
constexpr bool QSCF16 = TMAC_QUAD_SCF16 != 0;   // weight scales stored as fp16 (1) / fp32 (0): a kernel template parameter

#if TMAC_QUAD_BITS == 2 && TMAC_QUAD_SCF16 == 0
// ---------------------------------------------------------------------------------------------
// Prefill LUT build for the fused entry point: the quad kernel's build phase as a kernel of its own.  One lane builds
// the two tables of pair p (8 activations, one 16-byte fp16 load) of activation row n with the packed-fp32 sequence of
// q_table8, the act group (64 activations = 8 lanes) shares its abs-max by DPP, and the result goes straight into the
// unit-major half-table image the one-hot GEMM stages from (same bytes as k_preprocess writes there; QLUT, scales and
// biases bit-identical to lut_ctor.cc by the same argument as in k_gemv_quad).  k_preprocess keeps one workgroup per
// act group with 16 of 64 lanes building tables and writes three layouts; this one writes the image only.
// ---------------------------------------------------------------------------------------------
template <bool F16, bool ALL>
__global__ __launch_bounds__(256) void k_preprocess_pairs(const void* __restrict__ B, uint4* __restrict__ qlut_lds,
                                                          float* __restrict__ lut_scales, float* __restrict__ lut_biases,
                                                          int K, int tstride, uint4* __restrict__ qlut_ref,
                                                          uint2* __restrict__ qlut_dev, size_t qdev_u4_per_row) {
    const int P = K / 8, G = K / 64, n = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;                     // P % 8 == 0: the 8 lanes of an act group leave together
    float x[8];
    if (F16) {
        const uint4 v = reinterpret_cast<const uint4*>(reinterpret_cast<const __half*>(B) + (size_t)n * K)[p];
        const uint32_t r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const __half2 hh = *reinterpret_cast<const __half2*>(&r[i]);
            x[2 * i] = __low2float(hh); x[2 * i + 1] = __high2float(hh);
        }
    } else {
        const float4* src = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(B) + (size_t)n * K) + 2 * (size_t)p;
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
    q_table8<false>(x[0], x[1], x[2], x[3], t_scales, lo0, hi0, La);     // biased bytes, as the image holds them
    q_table8<false>(x[4], x[5], x[6], x[7], t_scales, lo1, hi1, Lb);
    qlut_lds[((size_t)n * 4 + (p & 3)) * tstride + (p >> 2)] = make_uint4(lo0, hi0, lo1, hi1);
    if (ALL) {
        // the other two layouts of the workspace (tmac_hip_preprocessor_dev serves every consumer):
        // 16-table-segment half tables for k_gemv_lo -- tables 2p, 2p+1 are one uint4 of segment p >> 3
        const int seg = p >> 3, j8 = p & 7;
        *reinterpret_cast<uint4*>(qlut_dev + ((size_t)n * qdev_u4_per_row + qlut_dev_u4_index(seg, j8)) * 2) = make_uint4(lo0, hi0, lo1, hi1);
        // and the reference's int8 [K/4][16]: entries 0..7 signed, 8..15 = -entry(15 - j) (lut_ctor.cc:152-155); on biased
        // bytes U in [1, 255] the bytewise negation 256 - U is one word subtraction without borrows
        const uint32_t sg = 0x80808080u, rev = 0x00010203u;
        const uint32_t n0l = __builtin_amdgcn_perm(0u, 0x01010100u - lo0, rev), n0h = __builtin_amdgcn_perm(0u, 0x01010100u - hi0, rev);
        const uint32_t n1l = __builtin_amdgcn_perm(0u, 0x01010100u - lo1, rev), n1h = __builtin_amdgcn_perm(0u, 0x01010100u - hi1, rev);
        uint4* r = qlut_ref + ((size_t)n * (K / 4) + 2 * (size_t)p);
        r[0] = make_uint4(lo0 ^ sg, hi0 ^ sg, n0h ^ sg, n0l ^ sg);
        r[1] = make_uint4(lo1 ^ sg, hi1 ^ sg, n1h ^ sg, n1l ^ sg);
    }
    float va = -La, vb = -Lb;               // lut_biases, lut_ctor.cc:25-31 (see k_gemv_quad)
    va = __fadd_rn(va, qdpp_f<0x4E>(va));
    vb = __fadd_rn(vb, qdpp_f<0x4E>(vb));
    va = __fadd_rn(va, qdpp_f<0xB1>(va));
    vb = __fadd_rn(vb, qdpp_f<0xB1>(vb));
    const float v = __fadd_rn(va, vb);
    const float c1 = qdpp_f<0x104>(v);
    if ((p & 7) == 0) {
        lut_scales[(size_t)n * G + (p >> 3)] = scales;
        lut_biases[(size_t)n * G + (p >> 3)] = __fadd_rn(__fadd_rn(0.0f, v), c1);
    }
}

// The same for one act group per row (act_group_size = K, the unified-scale / BitNet flavour): a workgroup owns an
// activation row.  Pass 1: abs-max of the row and the 8-table chunk sums of lut_biases (neither depends on the scale);
// then one lane walks the reference's sequential fp32 chain over the chunk sums (lut_ctor.cc:157,218) while the other
// waves quantise their tables -- the build phase of k_gemv_quad's SM == 2 path as a kernel of its own.
template <bool F16, bool ALL, int PT>
__global__ __launch_bounds__(PT) void k_preprocess_pairs_row(const void* __restrict__ B, uint4* __restrict__ qlut_lds,
                                                               float* __restrict__ lut_scales, float* __restrict__ lut_biases,
                                                               int K, int tstride, uint4* __restrict__ qlut_ref,
                                                               uint2* __restrict__ qlut_dev, size_t qdev_u4_per_row,
                                                               uint4* __restrict__ bimg, float* __restrict__ colv, int Npad) {
    extern __shared__ float prs[];          // [PT/64] wave maxima | [K/32] chunk sums
    constexpr int NWV = PT / 64;
    const int P = K / 8, n = blockIdx.x, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    float* l_mx = prs;
    float* l_cs = prs + NWV;
    int* l_hs = reinterpret_cast<int*>(prs + NWV + K / 32);      // sum of all signed half-table entries of the row (LUT image only)
    if (tid == 0) *l_hs = 0;
    constexpr int NPR = 3;                   // pairs per thread: K <= 24 * PT
    float x[NPR][8];
    float mx = 0.f;
#pragma unroll
    for (int r = 0; r < NPR; ++r) {
        const int p = r * PT + tid;
        if (p < P) {
            if (F16) {
                const uint4 v = reinterpret_cast<const uint4*>(reinterpret_cast<const __half*>(B) + (size_t)n * K)[p];
                const uint32_t rr[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const __half2 hh = *reinterpret_cast<const __half2*>(&rr[i]);
                    x[r][2 * i] = __low2float(hh); x[r][2 * i + 1] = __high2float(hh);
                }
            } else {
                const float4* src = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(B) + (size_t)n * K) + 2 * (size_t)p;
                const float4 a0 = src[0], a1 = src[1];
                x[r][0] = a0.x; x[r][1] = a0.y; x[r][2] = a0.z; x[r][3] = a0.w; x[r][4] = a1.x; x[r][5] = a1.y; x[r][6] = a1.z; x[r][7] = a1.w;
            }
            mx = fmaxf(mx, __fadd_rn(__fadd_rn(fabsf(x[r][0]), fabsf(x[r][1])), __fadd_rn(fabsf(x[r][2]), fabsf(x[r][3]))));
            mx = fmaxf(mx, __fadd_rn(__fadd_rn(fabsf(x[r][4]), fabsf(x[r][5])), __fadd_rn(fabsf(x[r][6]), fabsf(x[r][7]))));
            float va = -__fadd_rn(__fadd_rn(__fadd_rn(x[r][0], x[r][1]), x[r][2]), x[r][3]);
            float vb = -__fadd_rn(__fadd_rn(__fadd_rn(x[r][4], x[r][5]), x[r][6]), x[r][7]);
            va = __fadd_rn(va, qdpp_f<0x4E>(va));
            vb = __fadd_rn(vb, qdpp_f<0x4E>(vb));
            va = __fadd_rn(va, qdpp_f<0xB1>(va));
            vb = __fadd_rn(vb, qdpp_f<0xB1>(vb));
            if ((p & 3) == 0) l_cs[p >> 2] = __fadd_rn(va, vb);
        }
    }
    mx = q_row_allmax(mx);
    mx = q_xor_max_f(mx);
    if (lane == 0) l_mx[w] = mx;
    __syncthreads();
    mx = l_mx[0];
#pragma unroll
    for (int i = 1; i < NWV; ++i) mx = fmaxf(mx, l_mx[i]);
    const float gscale = __fdiv_rn(mx, 127.0f);
    const float gtinv = (gscale != 0.0f) ? __fdiv_rn(1.0f, gscale) : 0.0f;
    if (tid == PT - 64) {
        float biases = 0.0f;
        const int nc = K / 32;
        for (int c = 0; c < nc; ++c) biases = __fadd_rn(biases, l_cs[c]);
        lut_scales[n] = gscale;
        lut_biases[n] = biases;
        if (colv) { colv[n] = gscale; colv[Npad + n] = biases; }      // the layout k_gemm_planes_us reads (tmac_gemm2.hip)
    }
    int hsum = 0;
#pragma unroll
    for (int r = 0; r < NPR; ++r) {
        const int p = r * PT + tid;
        if (p < P) {
            uint32_t lo0, hi0, lo1, hi1;
            float La, Lb;
            q_table8<false>(x[r][0], x[r][1], x[r][2], x[r][3], gtinv, lo0, hi0, La);
            q_table8<false>(x[r][4], x[r][5], x[r][6], x[r][7], gtinv, lo1, hi1, Lb);
            qlut_lds[((size_t)n * 4 + (p & 3)) * tstride + (p >> 2)] = make_uint4(lo0, hi0, lo1, hi1);
            if (bimg) {  // signed half tables, [unit][pair][n]: what the plane-combined GEMM streams
                bimg[(size_t)p * Npad + n] = make_uint4(lo0 ^ 0x80808080u, hi0 ^ 0x80808080u, lo1 ^ 0x80808080u, hi1 ^ 0x80808080u);
                // the 16 entries are biased bytes U = entry + 128: sum of the bytes - 16 * 128
                hsum += (int)__builtin_amdgcn_sad_u8(lo0, 0u, __builtin_amdgcn_sad_u8(hi0, 0u, __builtin_amdgcn_sad_u8(lo1, 0u, __builtin_amdgcn_sad_u8(hi1, 0u, 0u)))) - 2048;
            }
            if (ALL) {
                const int seg = p >> 3, j8 = p & 7;
                *reinterpret_cast<uint4*>(qlut_dev + ((size_t)n * qdev_u4_per_row + qlut_dev_u4_index(seg, j8)) * 2) = make_uint4(lo0, hi0, lo1, hi1);
                const uint32_t sg = 0x80808080u, rev = 0x00010203u;
                const uint32_t n0l = __builtin_amdgcn_perm(0u, 0x01010100u - lo0, rev), n0h = __builtin_amdgcn_perm(0u, 0x01010100u - hi0, rev);
                const uint32_t n1l = __builtin_amdgcn_perm(0u, 0x01010100u - lo1, rev), n1h = __builtin_amdgcn_perm(0u, 0x01010100u - hi1, rev);
                uint4* rp = qlut_ref + ((size_t)n * (K / 4) + 2 * (size_t)p);
                rp[0] = make_uint4(lo0 ^ sg, hi0 ^ sg, n0h ^ sg, n0l ^ sg);
                rp[1] = make_uint4(lo1 ^ sg, hi1 ^ sg, n1h ^ sg, n1l ^ sg);
            }
        }
    }
    if (colv) {   // (uniform) the row's entry sum, int32 bits: what the +7 / +15 operand bytes of 3- / 4-bit rows add per unit of it
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) hsum += __shfl_xor(hsum, off);
        if (lane == 0) atomicAdd(l_hs, hsum);
        __syncthreads();
        if (tid == 0) colv[2 * (size_t)Npad + n] = __int_as_float(*l_hs);
    }
}

hipError_t launch_preprocess_pairs_row(const void* B, int act_f16, void* qlut_lds, float* lut_scales, float* lut_biases, int K, int N,
                                       int8_t* qlut_ref, void* qlut_dev, size_t qdev_u4_per_row, void* bimg, float* colv, int Npad, hipStream_t st) {
    constexpr int PT = 512;
    if (K % 64 != 0 || N < 1 || K > 24 * PT || ((qlut_ref == nullptr) != (qlut_dev == nullptr))) return hipErrorInvalidValue;
    const int tstride = (((K / 32) + 15) & ~15) + 1;
    const size_t shmem = sizeof(float) * (PT / 64 + K / 32 + 1);
    dim3 g(N), b(PT);
#define PLR(F, A) hipLaunchKernelGGL((k_preprocess_pairs_row<F, A, PT>), g, b, shmem, st, B, (uint4*)qlut_lds, lut_scales, lut_biases, K, tstride, \
                                     (uint4*)qlut_ref, (uint2*)qlut_dev, qdev_u4_per_row, (uint4*)bimg, colv, Npad)
    if (qlut_ref) { if (act_f16) PLR(true, true); else PLR(false, true); }
    else { if (act_f16) PLR(true, false); else PLR(false, false); }
#undef PLR
    return hipGetLastError();
}

hipError_t launch_preprocess_pairs(const void* B, int act_f16, void* qlut_lds, float* lut_scales, float* lut_biases, int K, int N,
                                   int8_t* qlut_ref, void* qlut_dev, size_t qdev_u4_per_row, hipStream_t st) {
    if (K % 64 != 0 || N < 1 || ((qlut_ref == nullptr) != (qlut_dev == nullptr))) return hipErrorInvalidValue;
    const int tstride = (((K / 32) + 15) & ~15) + 1;
    dim3 g((K / 8 + 255) / 256, N), b(256);
#define PL(F, A) hipLaunchKernelGGL((k_preprocess_pairs<F, A>), g, b, 0, st, B, (uint4*)qlut_lds, lut_scales, lut_biases, K, tstride, \
                                    (uint4*)qlut_ref, (uint2*)qlut_dev, qdev_u4_per_row)
    if (qlut_ref) { if (act_f16) PL(true, true); else PL(false, true); }
    else { if (act_f16) PL(true, false); else PL(false, false); }
#undef PL
    return hipGetLastError();
}

bool gemv_quad_supported(const Shape& s) {
    if (s.bits < 1 || s.bits > 4 || s.K % 64 != 0 || s.K > 24576 || s.Mw % 4 != 0) return false;
    if (s.m_groups >= 1) return s.ags == s.K && s.Mw % s.m_groups == 0;
    const int gu = s.gs / 32;
    return s.ags == 64 && s.gs >= 64 && s.gs % 64 == 0 && s.K % s.gs == 0 && (gu & (gu - 1)) == 0;
}

#endif

static size_t quad_lds_bytes(const Shape& s, int nwv) {
    const int nu = s.K / 32, nst = (nu + 63) / 64, G = s.K / s.ags;
    return (size_t)4 * (nst * 64 + 1) * 16 + sizeof(float) * (2 * (nst * 32 > G ? nst * 32 : G) + 2 * nwv * 16 + nwv + s.K / 32);
}

void fused_precompute(FusedArgs& a);   // tmac_fused.hip

// Instantiation policy (compile time): the v_mqsad accumulate (ACC 0, A/B variant 7), the prebuilt-LUT source
// (LUTSRC 0) and the integer tap (DUMP) exist for 512-thread workgroups only.
template <int BITS, bool ZP, int SM, int LUTSRC, int FT, int WPQ>
static hipError_t qlaunch_nr(const FusedArgs& a, int total_q, int N, hipStream_t st) {
    constexpr int IPI = FT / 64 / WPQ;
    const size_t shmem = quad_lds_bytes(a.s, FT / 64);
    int gx = (total_q + IPI - 1) / IPI;
    const int cap = (FT > 512) ? 256 : 512;       // persistent: <= 1 (FT = 768, 1024) / 2 (FT = 512) workgroups per CU (measured best)
    if (gx > cap) gx = cap;
    dim3 g(gx, N), b(FT);
    const int T = a.s.K / 4;
    constexpr int A1 = 1;
#define QLE(NRV, DV, AV, EV) hipLaunchKernelGGL((k_gemv_quad<BITS, ZP, SM, LUTSRC, NRV, FT, WPQ, DV, AV, QSCF16, EV>), g, b, shmem, st, a)
#define QL(NRV, DV, AV) QLE(NRV, DV, AV, false)
    const bool early = LUTSRC == 1 && gx <= 256;   // one workgroup per CU: overlap the weight stream with the LUT build (see the kernel)
    const bool two = (LUTSRC == 0 || T <= 2 * FT);
    if (!two && T > 6 * FT) return hipErrorInvalidValue;
    if constexpr (FT == 512) {
        const int acc = a.acc_mfma ? 1 : 0;
        if (a.dump) {
            if (acc) { if (two) QL(2, true, A1); else if constexpr (LUTSRC == 1) QL(6, true, A1); }
            else { if (two) QL(2, true, 0); else if constexpr (LUTSRC == 1) QL(6, true, 0); }
        } else {
            if (acc) {
                if constexpr (LUTSRC == 1) {
                    if (early) { if (two) QLE(2, false, A1, true); else QLE(6, false, A1, true); }
                    else { if (two) QL(2, false, A1); else QL(6, false, A1); }
                } else QL(2, false, A1);
            } else { if (two) QL(2, false, 0); else if constexpr (LUTSRC == 1) QL(6, false, 0); }
        }
    } else {
        if (a.dump || LUTSRC == 0 || !a.acc_mfma) return hipErrorInvalidValue;
        if constexpr (LUTSRC == 1) { if (two) QLE(2, false, A1, true); else QLE(6, false, A1, true); }   // grids of these sizes never exceed one workgroup per CU
    }
#undef QL
#undef QLE
    return hipGetLastError();
}

template <int BITS, bool ZP, int SM, int LUTSRC>
static hipError_t qlaunch_cfg(const FusedArgs& a, int total_q, int N, int force_ft, int force_wpq, hipStream_t st) {
    // Configuration choice, from tools/tune_quad.py on MI355X (profiles/r01_tune_quad.txt, us per launch in a graph):
    //   W2: o 4096x4096: (512,2) 4.3 | qkv 12288x4096: (512,2) 6.5, (512,1) 6.6, (1024,1) 6.9 | gate_up 22016x4096:
    //   (512,1) 9.2, (512,2) 10.2 | down 4096x11008: (768,3) 6.9, (1024,4) 7.2, (512,2) 7.3, (512,1) 8.8.
    //   W4 (tune_quad.py 0 4): o (512,2) | qkv (512,2) | gate_up (512,1) | down (512,2).
    const int nst = (a.s.K / 32 + 63) / 64;
    // two waves per quad up to one quad per wave slot of the chip (4096), one beyond
    int best_ft = 512, best_wpq = (total_q <= 4096 && nst >= 2) ? 2 : 1;
    // ... except where two waves per quad would leave the chip between one and two workgroups per CU (shards of a
    // row-split model: 3 x 2048 and 2 x 2752 rows measured 7 % faster with one, profiles/history/r01_autotune_shards_before_heuristic.txt)
    if (best_wpq == 2 && total_q > 1024 && total_q < 2048) best_wpq = 1;
    if (BITS == 2 && total_q <= 1024 && nst >= 4 && !(a.dump || LUTSRC == 0 || !a.acc_mfma)) {
        // long rows, few quads: 3 waves per quad when that splits the steps evenly, else 4
        if (nst % 3 == 0 && a.s.K / 4 <= 6 * 768) { best_ft = 768; best_wpq = 3; }
        else { best_ft = 1024; best_wpq = 4; }
    }
    // One balanced pass: if twelve-wave workgroups with 1, 2 or 3 waves per quad cover the quads in a single pass over
    // 75-100 % of the CUs, every CU gets the same work, the LUT is built once per CU and the weights can be issued early.
    // Matches every case the tuner found on the llama-2-7B shapes and their 2-/4-/8-way row shards
    // (profiles/history/r01_autotune_shards_before_heuristic.txt and the runs after it): q/k/v 3 x 4096 rows -> (768,1) 5.6 against 6.1 us; 3 x 2048 -> (768,2);
    // gate/up 2 x 5504 -> (768,1) 5.45 against 5.9; 2 x 2752 -> (768,2); down 4096 x 11008 -> (768,3).
    if (!(a.dump || LUTSRC == 0 || !a.acc_mfma) && a.s.K / 4 <= 6 * 768) {
        for (int wq = 1; wq <= 3; ++wq) {
            if (wq > nst || nst % wq || (wq == 3 && BITS > 2)) continue;   // (W4 long rows measure better on (512,2): tune_quad.py 0 4)
            const int wgs = (total_q * wq + 11) / 12;
            if (wgs > 192 && wgs <= 256) { best_ft = 768; best_wpq = wq; break; }
        }
    }
    double best = 0.0;
    if (a.s.K / 4 > 6 * 512 && best_ft == 512 && !(a.dump || LUTSRC == 0 || !a.acc_mfma)) best_ft = 1024;   // LUT build: <= 6 tables per thread
    if (force_ft) { best_ft = force_ft; if (!force_wpq && best_wpq > 2) best_wpq = 2; }
    if (force_wpq) best_wpq = force_wpq;
    const bool need512 = a.dump || LUTSRC == 0 || !a.acc_mfma;
    if (best_ft == 768 || best_wpq == 3) {        // 12 waves: 3 per quad (balanced when the row has 3k steps), or 12 / 6 quads per workgroup
        if (best_ft != 768 || need512 || a.s.K / 4 > 6 * 768) return hipErrorInvalidValue;
        if (best_wpq == 3) return qlaunch_nr<BITS, ZP, SM, LUTSRC, 768, 3>(a, total_q, N, st);
        if (best_wpq == 1) return qlaunch_nr<BITS, ZP, SM, LUTSRC, 768, 1>(a, total_q, N, st);
        if (best_wpq == 2) return qlaunch_nr<BITS, ZP, SM, LUTSRC, 768, 2>(a, total_q, N, st);
        return hipErrorInvalidValue;
    }
    if ((need512 && best_ft != 512) || (best_wpq == 4 && best_ft != 1024) || (best_ft != 512 && best_ft != 1024) ||
        (best_wpq != 1 && best_wpq != 2 && best_wpq != 4) || a.s.K / 4 > 6 * best_ft)
        best = 1e30;
    if (best >= 1e30) return hipErrorInvalidValue;
    if (best_ft == 512) return best_wpq == 1 ? qlaunch_nr<BITS, ZP, SM, LUTSRC, 512, 1>(a, total_q, N, st)
                                             : qlaunch_nr<BITS, ZP, SM, LUTSRC, 512, 2>(a, total_q, N, st);
    if (best_wpq == 4) return qlaunch_nr<BITS, ZP, SM, LUTSRC, 1024, 4>(a, total_q, N, st);
    return best_wpq == 1 ? qlaunch_nr<BITS, ZP, SM, LUTSRC, 1024, 1>(a, total_q, N, st)
                         : qlaunch_nr<BITS, ZP, SM, LUTSRC, 1024, 2>(a, total_q, N, st);
}

template <int BITS, int LUTSRC>
static hipError_t qlaunch_b(const FusedArgs& a, int total_q, int N, int fft, int fwpq, hipStream_t st) {
    if (a.s.m_groups >= 1) {   // unified scale: one scalar read per output, dtype stays a run-time flag (fp32 TU only)
        if constexpr (QSCF16) return hipErrorInvalidValue;
        else return qlaunch_cfg<BITS, false, 2, LUTSRC>(a, total_q, N, fft, fwpq, st);
    }
    return a.s.zero_point ? qlaunch_cfg<BITS, true, 0, LUTSRC>(a, total_q, N, fft, fwpq, st)
                          : qlaunch_cfg<BITS, false, 0, LUTSRC>(a, total_q, N, fft, fwpq, st);
}

// a.m[i].nb_end must hold cumulative QUAD counts.  force_ft / force_wpq: 0 = heuristic (A/B knobs)
#define QENTRY_(b, h) launch_gemv_quad_b##b##_h##h
#define QENTRY(b, h) QENTRY_(b, h)
#define QENTRY_DECL(b, h) hipError_t QENTRY_(b, h)(const FusedArgs& a, int total_q, int N, bool build_lut, int force_ft, int force_wpq, hipStream_t st)
QENTRY_DECL(1, 0); QENTRY_DECL(1, 1); QENTRY_DECL(2, 0); QENTRY_DECL(2, 1); QENTRY_DECL(3, 0); QENTRY_DECL(3, 1); QENTRY_DECL(4, 0); QENTRY_DECL(4, 1);
hipError_t QENTRY(TMAC_QUAD_BITS, TMAC_QUAD_SCF16)(const FusedArgs& a, int total_q, int N, bool build_lut, int force_ft, int force_wpq, hipStream_t st) {
    return build_lut ? qlaunch_b<TMAC_QUAD_BITS, 1>(a, total_q, N, force_ft, force_wpq, st)
                     : qlaunch_b<TMAC_QUAD_BITS, 0>(a, total_q, N, force_ft, force_wpq, st);
}
#if TMAC_QUAD_BITS == 2 && TMAC_QUAD_SCF16 == 0
hipError_t launch_gemv_quad(const FusedArgs& a_in, int N, bool build_lut, int force_ft, int force_wpq, hipStream_t st) {
    if (!gemv_quad_supported(a_in.s) || a_in.nmat < 1 || a_in.nmat > 4) return hipErrorInvalidValue;
    FusedArgs a = a_in;
    fused_precompute(a);
    const int total_q = a.m[a.nmat - 1].nb_end;
    const bool h = a.sc_f16 && a.s.m_groups < 1;
#define QDISP(B) return h ? launch_gemv_quad_b##B##_h1(a, total_q, N, build_lut, force_ft, force_wpq, st) \
                          : launch_gemv_quad_b##B##_h0(a, total_q, N, build_lut, force_ft, force_wpq, st)
    switch (a.s.bits) {
        case 1: QDISP(1);
        case 2: QDISP(2);
        case 3: QDISP(3);
        default: QDISP(4);
    }
#undef QDISP
}
#endif

}  // namespace tmac
