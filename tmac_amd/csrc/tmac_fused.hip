// tmac_fused.hip — the production GEMV kernel: LUT construction fused into the lookup kernel,
// several weight matrices that share one activation vector in ONE launch (q/k/v, gate/up).
//
// Why (measured on MI355X, profiles/r01_*): a llama-2-7B decode step is 224 GEMVs of 4.7-12.7 MB.
// A pure 12.7 MB streaming kernel costs ~3.6 us back to back and an empty kernel ~2.6 us, so the
// per-launch boundary, not the bytes, dominates.  Fusing the preprocessor (lut_ctor.cc) into the GEMV
// and batching matrices that share their input takes a layer from 11 launches to 4, and the LUT
// build (fp32 VALU + LDS writes, ~0.5-1 us) runs while the first weight loads are in flight.
//
// Workgroup = 512 threads = 8 waves, owns 16 output rows (4 row quads) over the whole K.
//   lane = rl*16 + ul   (row quad rl, unit lane ul);  unit = 8 tables = 32 activations (ts = 8 layout)
//   wave w, step i handle unit u = i*128 + w*16 + ul; two neighbouring lanes (ul, ul^1) hold the two
//   halves of one 64-activation act group and exchange their packed integer sums with one DPP add, so
//   the per-group integer partial sum stays exact before the fp32 scale-apply (tbl.cc:464-492).
// LDS: half tables [4][nu_pad+1] x 16 B (j4-major: a wave-instruction reads 256 contiguous bytes,
//   row quads broadcast), LUT scales/biases, reduction scratch.  <= 26 KB for K = 11008.
//
//   LUTSRC 0: tables copied from the workspace (standalone preprocessor ran before)  -> tmac_hip_qgemm_dev
//   LUTSRC 1: tables built in-kernel from the activations (bit-exact with k_preprocess) -> tmac_hip_qgemm_fused_dev
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "tmac_core.h"
#include "tmac_kernels.h"

namespace tmac {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float f_alpha(int p) { return p == 0 ? 0.5f : (p == 1 ? 1.0f : (p == 2 ? 2.0f : 4.0f)); }

__device__ __forceinline__ int f_rne_sat_int8(float x) {
    const float r = rintf(x);
    int i = (r >= -2147483648.0f && r < 2147483648.0f) ? (int)r : INT32_MIN;
    return max(-128, min(127, i));
}

__device__ __forceinline__ float ld_scale(const void* p, int f16, size_t i) {
    return f16 ? __half2float(reinterpret_cast<const __half*>(p)[i]) : reinterpret_cast<const float*>(p)[i];
}

__device__ __forceinline__ void st_out(void* C, int f16, size_t i, float v) {
    if (f16) reinterpret_cast<__half*>(C)[i] = __float2half_rn(v);
    else reinterpret_cast<float*>(C)[i] = v;
}

// Cross-lane moves as DPP modifiers (no LDS round trip, unlike __shfl_xor -> ds_bpermute_b32).
//   quad_perm [1,0,3,2] = 0xB1 (lane^1)   quad_perm [2,3,0,1] = 0x4E (lane^2)
//   row_shl:n = 0x100+n (lane i reads lane i+n of its 16-lane row)   row_ror:n = 0x120+n
//   row_half_mirror = 0x141 (i -> 7-i within 8)   row_mirror = 0x140 (i -> 15-i within 16)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
// max over the 16 lanes of a DPP row, result in every lane (max is commutative and idempotent)
__device__ __forceinline__ float row_allmax(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    return v;
}

#define TMAC_STAMP(i) do { if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)

constexpr int FT = 512;        // threads per workgroup
constexpr int FW = FT / 64;    // waves
constexpr int UPS = FW * KL;   // units per step (128)

template <int BITS>
struct WFrag {
    uint32_t wd[8 * BITS / 2];
    uint32_t sraw[4];     // raw scale / zero-point words of this lane's two rows, fetched WITH the weights
};                        // (converted at use: converting at load time would stall on the load)

// scale (which = 0) or zero point (which = 1) of row i (0/1) from the raw words
template <bool ZP>
__device__ __forceinline__ float frag_scale(const uint32_t (&sraw)[4], int f16, int i, int which) {
    const int e = i * (ZP ? 2 : 1) + which;          // element index within the lane's 2*per values
    if (f16) {
        const uint32_t wv = sraw[e >> 1];
        return __half2float(__ushort_as_half((unsigned short)((e & 1) ? (wv >> 16) : (wv & 0xffff))));
    }
    return __uint_as_float(sraw[e]);
}

// weights (non-temporal, 1 KiB per wave-instruction) + the lane's scale values for unit u
template <int BITS, bool ZP, int SM, int ACC>
__device__ __forceinline__ void load_w(WFrag<BITS>& f, const FusedArgs& a, const FusedMat& M, int b, int ub, int u, int rl,
                                       int ul) {
    constexpr int NJ = 8 * BITS / 8;
    const bool src_valid = u < a.nu;     // ragged last step: lanes past K load no weights
    if (src_valid) {
        const uint4* wp = M.W + ((size_t)(b * a.nsb + ub) * NJ * RL + rl) * KL + ul;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + (size_t)j * RL * KL));
            f.wd[4 * j] = v.x; f.wd[4 * j + 1] = v.y; f.wd[4 * j + 2] = v.z; f.wd[4 * j + 3] = v.w;
        }
    }
    // scale words go through scalar locals and reach f.sraw with constant subscripts: stores to different elements
    // in the two dtype branches would be sunk into one store with a run-time subscript (-> scratch memory)
    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    if (SM == 0 && ACC == 1) {
        // epilogue lane l' = 16*lg + 4*rlp + bp owns row (rlp, bp) and units ub*16 + 4*lg .. +3 (2 act groups);
        // sraw[2*gi], sraw[2*gi+1]: scale (, zero) of act-group pair gi; f16 packs both into sraw[2*gi]
        constexpr int per = ZP ? 2 : 1;
        const int lane = rl * KL + ul, lg = lane >> 4, rlp = (lane & 15) >> 2, bp = lane & 3;
        const int ub4 = ub * KL + 4 * lg;
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            if (gi == 1 && a.gs_shift >= 2) break;               // gs >= 128: both act groups share the scale group
            const int sg = (ub4 + 2 * gi) >> a.gs_shift;
            uint32_t v0 = 0, v1 = 0;
            if (ub4 + 2 * gi < a.nu) {
                const size_t sidx = ((((size_t)b * a.nsg + sg) * RL + rlp) * 4 + bp) * per;
                if (a.sc_f16) {
                    const __half* ph = reinterpret_cast<const __half*>(M.SC) + sidx;
                    if (ZP) v0 = *reinterpret_cast<const uint32_t*>(ph);
                    else v0 = *reinterpret_cast<const unsigned short*>(ph);
                } else {
                    const uint32_t* p32 = reinterpret_cast<const uint32_t*>(M.SC) + sidx;
                    v0 = p32[0];
                    if (ZP) v1 = p32[1];
                }
            }
            if (gi == 0) { r0 = v0; r1 = v1; } else { r2 = v0; r3 = v1; }
        }
    } else if (SM == 0 && src_valid) {
        constexpr int per = ZP ? 2 : 1;
        const int sg = u >> a.gs_shift;                        // scale group = u*32 / gs
        const size_t sidx = ((((size_t)b * a.nsg + sg) * RL + rl) * 4 + 2 * (ul & 1)) * per;
        // 2*per consecutive elements: 4 B (f16) / 8 B (f16 zp, f32) / 16 B (f32 zp), naturally aligned
        if (a.sc_f16) {
            const uint32_t* p32 = reinterpret_cast<const uint32_t*>(reinterpret_cast<const __half*>(M.SC) + sidx);
            r0 = p32[0];
            if (ZP) r1 = p32[1];
        } else {
            const uint32_t* p32 = reinterpret_cast<const uint32_t*>(M.SC) + sidx;
            r0 = p32[0]; r1 = p32[1];
            if (ZP) { r2 = p32[2]; r3 = p32[3]; }
        }
    }
    if (SM == 0) { f.sraw[0] = r0; f.sraw[1] = r1; f.sraw[2] = r2; f.sraw[3] = r3; }
}

// SM 0: per-(row, group) scales (+ zero points), act group 64.   SM 2: unified scale applied last (ags == K).
typedef int v4i_t __attribute__((ext_vector_type(4)));

// ACC 0: v_mqsad_pk_u16_u8 accumulate on the VALU (biased tables).
// ACC 1: v_mfma_i32_16x16x64_i8 accumulate on the matrix pipe (signed tables): each lane's 4 operand
//        registers {plus(ta), minus(ta), plus(tb), minus(tb)} x 4 row bytes are summed, per lane and row
//        byte, against the constant selector B[(g,q,beta)][j] = (q even ? +1 : -1) * [j == 4g+beta]; the
//        result C[unit lane][4*row quad + beta] lands transposed: lane l' = 16*(unit/4) + 4*rl + beta holds,
//        in its 4 accumulator registers, ONE output row and FOUR consecutive units (= 2 act groups, one
//        128-wide scale group), which is exactly what the fp32 epilogue wants.  VALU cost per 4 lookups
//        drops from ~37 to ~23 cycles (measured rates: v_perm_b32 4.8, v_mqsad 17.5, simple VALU 2.8 cycles).
template <int BITS, bool ZP, int SM, int LUTSRC, int NR, int ACC>
__global__ __launch_bounds__(FT) void k_gemv_fused(FusedArgs a) {
    extern __shared__ uint4 lds[];
    const Shape& s = a.s;  // s.Mw is not meaningful here (per-matrix Mw in a.m[])
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, rl = lane >> 4, ul = lane & 15;
    const int n = blockIdx.y;
    const int T = s.K / 4, nu = a.nu, G = a.G, tstride = a.tstride;   // host-precomputed: no integer divides here
    uint4* tab = lds;                                             // [4][tstride]
    float* l_ls = reinterpret_cast<float*>(lds + 4 * tstride);    // [G]   (SM2: [1])
    float* l_lb = l_ls + G;                                       // [G]
    float* l_red = l_lb + G;                                      // [FW][RL][4] floats / ints, then build scratch
    float* l_scr = l_red + 2 * FW * RL * 4 * 4;                   // SM2 build: [FW] maxima + [T/8] chunk sums

    TMAC_STAMP(0);
    // ---- persistent row-block loop: global block gb = blockIdx.x + i*gridDim.x -> (matrix, local block) ----
    const int total_nb = a.m[a.nmat - 1].nb_end;
    const int nsteps = (nu + UPS - 1) / UPS;
    auto locate = [&](int gb, int& mi, int& bl) {
        mi = 0;
#pragma unroll
        for (int i = 1; i < 4; ++i)
            if (i < a.nmat && gb >= a.m[i - 1].nb_end) mi = i;
        bl = gb - (mi ? a.m[mi - 1].nb_end : 0);
    };

    // ---- 1. activation loads for the LUT build (issued FIRST: vmcnt retires in order) -----------
    uint32_t xr[NR][4];
    if (LUTSRC == 1) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int t = r * FT + tid;
            if (t < T) {
                if (a.act_f16) {
                    const uint2 v = reinterpret_cast<const uint2*>(reinterpret_cast<const __half*>(a.B) + (size_t)n * s.K)[t];
                    xr[r][0] = v.x; xr[r][1] = v.y;
                } else {
                    const uint4 v = reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(a.B) + (size_t)n * s.K)[t];
                    xr[r][0] = v.x; xr[r][1] = v.y; xr[r][2] = v.z; xr[r][3] = v.w;
                }
            }
        }
    }

    // ---- 2. weight fragment ring (items = (row block, step) pairs in execution order); first issue after the LUT build --
    WFrag<BITS> f0, f1;
    int p_gb = blockIdx.x, p_step = 0;   // prefetch cursor
    auto issue = [&](WFrag<BITS>& f) {
        if (p_gb < total_nb) {
            int mi, bl;
            locate(p_gb, mi, bl);
            const int u = p_step * UPS + w * KL + ul;
            if (p_step * UPS + w * KL < nu) load_w<BITS, ZP, SM, ACC>(f, a, a.m[mi], bl, p_step * FW + w, u, rl, ul);
            if (++p_step == nsteps) { p_step = 0; p_gb += gridDim.x; }
        }
    };
    TMAC_STAMP(1);
    // ---- 3. LUT into LDS ----------------------------------------------------------------------
    if (LUTSRC == 0) {
        const uint4* src = reinterpret_cast<const uint4*>(a.qlut_lds) + (size_t)n * 4 * tstride;
        for (int i = tid; i < 4 * tstride; i += FT) {
            uint4 v = src[i];
            if (ACC == 1) { v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u; }  // biased -> signed
            tab[i] = v;
        }
        if (SM == 2) { if (tid == 0) { l_ls[0] = a.lut_scales[n]; l_lb[0] = a.lut_biases[n]; } }
        else for (int i = tid; i < G; i += FT) { l_ls[i] = a.lut_scales[(size_t)n * G + i]; l_lb[i] = a.lut_biases[(size_t)n * G + i]; }
    } else {
        // fp32 arithmetic identical to k_preprocess (lut_ctor.cc:38-266); see that kernel for the citations
        float gscale = 0.f, gtinv = 0.f;
        if (SM == 2) {   // one act group == all of K: block-wide max first
            float mx = 0.f;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int t = r * FT + tid;
                if (t < T) {
                    float x0, x1, x2, x3;
                    if (a.act_f16) {
                        const __half2 h0 = *reinterpret_cast<const __half2*>(&xr[r][0]), h1 = *reinterpret_cast<const __half2*>(&xr[r][1]);
                        x0 = __low2float(h0); x1 = __high2float(h0); x2 = __low2float(h1); x3 = __high2float(h1);
                    } else { x0 = __uint_as_float(xr[r][0]); x1 = __uint_as_float(xr[r][1]); x2 = __uint_as_float(xr[r][2]); x3 = __uint_as_float(xr[r][3]); }
                    mx = fmaxf(mx, __fadd_rn(__fadd_rn(fabsf(x0), fabsf(x1)), __fadd_rn(fabsf(x2), fabsf(x3))));
                }
            }
            mx = row_allmax(mx);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            if (lane == 0) l_scr[w] = mx;
            __syncthreads();
            mx = l_scr[0];
#pragma unroll
            for (int i = 1; i < FW; ++i) mx = fmaxf(mx, l_scr[i]);
            gscale = __fdiv_rn(mx, 127.0f);
            gtinv = (gscale != 0.0f) ? __fdiv_rn(1.0f, gscale) : 0.0f;
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int t = r * FT + tid;
            if (t < T) {   // T % 16 == 0: a 16-lane act group is valid or invalid as a whole
                float x0, x1, x2, x3;
                if (a.act_f16) {
                    const __half2 h0 = *reinterpret_cast<const __half2*>(&xr[r][0]), h1 = *reinterpret_cast<const __half2*>(&xr[r][1]);
                    x0 = __low2float(h0); x1 = __high2float(h0); x2 = __low2float(h1); x3 = __high2float(h1);
                } else { x0 = __uint_as_float(xr[r][0]); x1 = __uint_as_float(xr[r][1]); x2 = __uint_as_float(xr[r][2]); x3 = __uint_as_float(xr[r][3]); }
                float scales, t_scales;
                if (SM == 2) { scales = gscale; t_scales = gtinv; }
                else {
                    const float mx = row_allmax(__fadd_rn(__fadd_rn(fabsf(x0), fabsf(x1)), __fadd_rn(fabsf(x2), fabsf(x3))));
                    scales = __fdiv_rn(mx, 127.0f);
                    t_scales = (scales != 0.0f) ? __fdiv_rn(1.0f, scales) : 0.0f;
                }
                // the 8 odd entries L[g] = ((x0 +- x1) +- x2) +- x3
                const float a_p = __fadd_rn(x0, x1), a_m = __fsub_rn(x0, x1);
                const float L1 = __fsub_rn(__fsub_rn(a_m, x2), x3), L3 = __fsub_rn(__fsub_rn(a_p, x2), x3);
                const float L5 = __fsub_rn(__fadd_rn(a_m, x2), x3), L7 = __fsub_rn(__fadd_rn(a_p, x2), x3);
                const float L9 = __fadd_rn(__fsub_rn(a_m, x2), x3), L11 = __fadd_rn(__fsub_rn(a_p, x2), x3);
                const float L13 = __fadd_rn(__fadd_rn(a_m, x2), x3), L15 = __fadd_rn(__fadd_rn(a_p, x2), x3);
                // half table entries j = 0..7: even j -> -L[15-j]
                const float e[8] = {-L15, L1, -L13, L3, -L11, L5, -L9, L7};
                uint32_t lo = 0, hi = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    lo |= (uint32_t)(max(f_rne_sat_int8(__fmul_rn(e[i], t_scales)), -127) + 128) << (8 * i);
                    hi |= (uint32_t)(max(f_rne_sat_int8(__fmul_rn(e[4 + i], t_scales)), -127) + 128) << (8 * i);
                }
                if (ACC == 1) { lo ^= 0x80808080u; hi ^= 0x80808080u; }   // signed entries for the MFMA path
                const int u = t >> 3, tl = t & 7;
                reinterpret_cast<uint2*>(tab + (tl >> 1) * tstride + u)[tl & 1] = make_uint2(lo, hi);
                // bias: chunk (8 tables) horizontal add in the reference order, then sequential over chunks
                // (lut_ctor.cc:25-31): lane 8c gets ((v0+v4)+(v2+v6)) + ((v1+v5)+(v3+v7)); row_shl:n reads lane i+n
                float v = -L15;
                v = __fadd_rn(v, dpp_f<0x104>(v));
                v = __fadd_rn(v, dpp_f<0x102>(v));
                v = __fadd_rn(v, dpp_f<0x101>(v));
                if (SM == 2) {
                    if ((t & 7) == 0) l_scr[FW + (t >> 3)] = v;
                } else {
                    const float c1 = dpp_f<0x108>(v);
                    if ((t & 15) == 0) {
                        l_ls[t >> 4] = scales;
                        l_lb[t >> 4] = __fadd_rn(__fadd_rn(0.0f, v), c1);
                    }
                }
            }
        }
        if (SM == 2) {
            __syncthreads();
            if (tid == 0) {
                float biases = 0.0f;
                for (int c = 0; c < T / 8; ++c) biases = __fadd_rn(biases, l_scr[FW + c]);
                l_ls[0] = gscale;
                l_lb[0] = biases;
            }
        }
    }
    // the weight fragments are issued only now: up front they would occupy the CU's load path while the VALU idles and
    // hold back the LUT build (measured on k_gemv_quad)
    issue(f0);
    issue(f1);
    TMAC_STAMP(2);
    __syncthreads();
    TMAC_STAMP(3);
    if (a.lut_tap && blockIdx.x == 0) {   // parity tap: the LUT scales/biases this kernel built
        for (int i = tid; i < (SM == 2 ? 1 : G); i += FT) { a.lut_tap[(size_t)n * 2 * G + i] = l_ls[i]; a.lut_tap[(size_t)n * 2 * G + G + i] = l_lb[i]; }
    }

    // ---- 4. lookups over this workgroup's row blocks ------------------------------------------
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
    const int beta0 = 2 * (ul & 1);

    auto compute = [&](const WFrag<BITS>& f, int u, int Mw_m, int bl) {
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
            // the two halves of the act group live in lanes (ul, ul^1): packed u16 sums add without carry
            uint32_t lo = (uint32_t)acc.a[pl], hi = (uint32_t)(acc.a[pl] >> 32);
            lo += dpp_u<0xB1>(lo);
            hi += dpp_u<0xB1>(hi);
            const uint32_t mine = (ul & 1) ? hi : lo;   // rows beta0, beta0+1
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int32_t ps = 127 * 16 - (int32_t)((mine >> (16 * i)) & 0xffff);
                if (a.dump) {
                    const int o = 4 * (bl * RL + rl) + beta0 + i;
                    if (o < Mw_m) a.dump[((size_t)n * Mw_m * BITS + mrow(o, pl, BITS)) * G + kk] = ps;
                }
                const float v = (pl == 0) ? __fmaf_rn((float)ps, ls, lb) : __fmul_rn((float)ps, ls);
                float c = __fmaf_rn(v, frag_scale<ZP>(f.sraw, a.sc_f16, i, 0), cacc[i][pl]);
                if (ZP && pl == 0) c = __fmaf_rn(frag_scale<ZP>(f.sraw, a.sc_f16, i, 1), __fmul_rn(2.0f, lb), c);
                cacc[i][pl] = c;
            }
        }
    };

    // ---- ACC == 1: matrix-pipe accumulate --------------------------------------------------------
    const int lg = lane >> 4, rlp = (lane & 15) >> 2, bp = lane & 3;   // epilogue role of this lane (see ACC note)
    v4i_t bsel;
    {
        const int jrel = (lane & 15) - 4 * (lane >> 4);                 // selector column of this lane as B operand
        const uint32_t be = (jrel >= 0 && jrel < 4) ? (0x01u << (8 * jrel)) : 0u;
        const uint32_t bo = (jrel >= 0 && jrel < 4) ? (0xffu << (8 * jrel)) : 0u;
        bsel = (v4i_t){(int)be, (int)bo, (int)be, (int)bo};
    }
    auto compute_mfma = [&](const WFrag<BITS>& f, int step, int Mw_m, int bl) {
        const int u = step * UPS + w * KL + ul;                         // this lane's unit as a SOURCE of lookups
        uint32_t tb[16];
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
            uint4 v = make_uint4(0, 0, 0, 0);                           // units past K contribute zero tables
            if (u < nu) v = tab[j4 * tstride + u];
            tb[4 * j4] = v.x; tb[4 * j4 + 1] = v.y; tb[4 * j4 + 2] = v.z; tb[4 * j4 + 3] = v.w;
        }
        v4i_t c[BITS];
#pragma unroll
        for (int pl = 0; pl < BITS; ++pl) c[pl] = (v4i_t){0, 0, 0, 0};
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
#pragma unroll
            for (int pl = 0; pl < BITS; ++pl) {
                uint32_t pa, ma, pb, mb;
                const int qa = (2 * tp) * BITS + pl, qb = (2 * tp + 1) * BITS + pl;   // nibble quads of tables 2tp, 2tp+1
                if (qa & 1) lookup4_pm<1>(f.wd[qa >> 1], tb[4 * tp], tb[4 * tp + 1], pa, ma);
                else lookup4_pm<0>(f.wd[qa >> 1], tb[4 * tp], tb[4 * tp + 1], pa, ma);
                if (qb & 1) lookup4_pm<1>(f.wd[qb >> 1], tb[4 * tp + 2], tb[4 * tp + 3], pb, mb);
                else lookup4_pm<0>(f.wd[qb >> 1], tb[4 * tp + 2], tb[4 * tp + 3], pb, mb);
                c[pl] = __builtin_amdgcn_mfma_i32_16x16x64_i8((v4i_t){(int)pa, (int)ma, (int)pb, (int)mb}, bsel, c[pl], 0, 0, 0);
            }
        }
        // epilogue: this lane now holds, for output row (rlp, bp), the sums of units ub4 .. ub4+3
        const int ub4 = step * UPS + w * KL + 4 * lg;
        const int o = 4 * (bl * RL + rlp) + bp;
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int ug = ub4 + 2 * gi;
            if (ug < nu) {
                const int kk = ug >> 1;
                const float ls = l_ls[kk], lb = l_lb[kk];
                const bool first = (gi == 0) || (a.gs_shift >= 2);     // which prefetched scale group (constant subscripts)
                float sc, zr = 0.f;
                if (a.sc_f16) {
                    const uint32_t wv = first ? f.sraw[0] : f.sraw[2];
                    sc = __half2float(__ushort_as_half((unsigned short)(wv & 0xffff)));
                    if (ZP) zr = __half2float(__ushort_as_half((unsigned short)(wv >> 16)));
                } else {
                    sc = __uint_as_float(first ? f.sraw[0] : f.sraw[2]);
                    if (ZP) zr = __uint_as_float(first ? f.sraw[1] : f.sraw[3]);
                }
#pragma unroll
                for (int pl = 0; pl < BITS; ++pl) {
                    const int32_t ps = (gi == 0) ? (c[pl].x + c[pl].y) : (c[pl].z + c[pl].w);   // two 8-table halves
                    if (a.dump && o < Mw_m) a.dump[((size_t)n * Mw_m * BITS + mrow(o, pl, BITS)) * G + kk] = ps;
                    const float v = (pl == 0) ? __fmaf_rn((float)ps, ls, lb) : __fmul_rn((float)ps, ls);
                    float cc = __fmaf_rn(v, sc, cacc[0][pl]);
                    if (ZP && pl == 0) cc = __fmaf_rn(zr, __fmul_rn(2.0f, lb), cc);
                    cacc[0][pl] = cc;
                }
            }
        }
    };

    // reduce over unit lanes and waves, store 16 outputs; l_red is double-buffered by block parity so
    // one barrier per row block suffices
    int parity = 0;
    auto finish_block = [&](const FusedMat& M, int bl) {
        float* red = l_red + parity * (FW * RL * 4 * 4);
        if (ACC == 1 && SM != 2) {
            float acc = __fmul_rn(cacc[0][0], 0.5f);
#pragma unroll
            for (int pl = 1; pl < BITS; ++pl) acc = __fadd_rn(acc, __fmul_rn(cacc[0][pl], f_alpha(pl)));
            acc = __fadd_rn(acc, __shfl_xor(acc, 16, 64));   // the 4 lane groups hold different unit quads of the row
            acc = __fadd_rn(acc, __shfl_xor(acc, 32, 64));
            if (lane < 16) red[w * 16 + lane] = acc;         // lane = 4*rlp + bp = row within the block
            __syncthreads();
            if (tid < RL * 4) {
                const int o = bl * 16 + tid;
                float t = red[tid];
#pragma unroll
                for (int ww = 1; ww < FW; ++ww) t = __fadd_rn(t, red[ww * 16 + tid]);
                if (o < M.Mw) st_out(M.C, a.out_f16, (size_t)n * M.Mw + o, t);
            }
        } else if (SM != 2) {
            float part[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float acc = __fmul_rn(cacc[i][0], 0.5f);
#pragma unroll
                for (int pl = 1; pl < BITS; ++pl) acc = __fadd_rn(acc, __fmul_rn(cacc[i][pl], f_alpha(pl)));
                // sum over the 8 same-parity lanes of the row: lane^2, then rotate by 4 and by 8
                acc = __fadd_rn(acc, dpp_f<0x4E>(acc));
                acc = __fadd_rn(acc, dpp_f<0x124>(acc));
                acc = __fadd_rn(acc, dpp_f<0x128>(acc));
                part[i] = acc;
            }
            if (ul < 2) {
                red[(w * RL + rl) * 4 + beta0] = part[0];
                red[(w * RL + rl) * 4 + beta0 + 1] = part[1];
            }
            __syncthreads();
            if (tid < RL * 4) {
                const int o = bl * 16 + tid;
                float acc = red[tid];
#pragma unroll
                for (int ww = 1; ww < FW; ++ww) acc = __fadd_rn(acc, red[ww * RL * 4 + tid]);
                if (o < M.Mw) st_out(M.C, a.out_f16, (size_t)n * M.Mw + o, acc);
            }
        } else {
            int32_t* redi = reinterpret_cast<int32_t*>(red);
#pragma unroll
            for (int pl = 0; pl < BITS; ++pl)
#pragma unroll
                for (int be = 0; be < 4; ++be) {
                    int32_t v = iacc[pl][be];
#pragma unroll
                    for (int m = 1; m < KL; m <<= 1) v += __shfl_xor(v, m, 64);
                    if (ul == 0) redi[((w * RL + rl) * 4 + pl) * 4 + be] = v;
                }
            __syncthreads();
            if (tid < RL * 4) {
                const int r = tid >> 2, be = tid & 3, o = bl * 16 + tid;
                if (o < M.Mw) {
                    float acc = 0.f;
#pragma unroll
                    for (int pl = 0; pl < BITS; ++pl) {
                        int32_t cb = 0;
#pragma unroll
                        for (int ww = 0; ww < FW; ++ww) cb += redi[((ww * RL + r) * 4 + pl) * 4 + be];
                        if (a.dump) a.dump[(size_t)n * M.Mw * BITS + mrow(o, pl, BITS)] = cb;
                        const float t = __fmul_rn((float)cb, f_alpha(pl));
                        acc = (pl == 0) ? t : __fadd_rn(acc, t);
                    }
                    const float v = __fadd_rn(__fmul_rn(acc, l_ls[0]), __fmul_rn(l_lb[0], 0.5f));
                    st_out(M.C, a.out_f16, (size_t)n * M.Mw + o, __fmul_rn(v, ld_scale(M.SC, a.sc_f16, o / (M.Mw / s.m_groups))));
                }
            }
        }
        parity ^= 1;
        reset_acc();
    };

    int c_gb = blockIdx.x, c_step = 0;   // compute cursor (uniform across the workgroup)
    while (c_gb < total_nb) {
        int mi, bl;
        locate(c_gb, mi, bl);
        {
            const int u = c_step * UPS + w * KL + ul;
            if (ACC == 1 && SM != 2) { if (c_step * UPS + w * KL < nu) compute_mfma(f0, c_step, a.m[mi].Mw, bl); }
            else if (u < nu) compute(f0, u, a.m[mi].Mw, bl);
            if (++c_step == nsteps) { finish_block(a.m[mi], bl); c_step = 0; c_gb += gridDim.x; }
            issue(f0);
        }
        if (c_gb >= total_nb) break;
        locate(c_gb, mi, bl);
        {
            const int u = c_step * UPS + w * KL + ul;
            if (ACC == 1 && SM != 2) { if (c_step * UPS + w * KL < nu) compute_mfma(f1, c_step, a.m[mi].Mw, bl); }
            else if (u < nu) compute(f1, u, a.m[mi].Mw, bl);
            if (++c_step == nsteps) { finish_block(a.m[mi], bl); c_step = 0; c_gb += gridDim.x; }
            issue(f1);
        }
    }
    TMAC_STAMP(4);
}

// ---------------------------------------------------------------------------------------------
bool gemv_fused_supported(const Shape& s) {
    if (s.bits < 1 || s.bits > 4 || s.K % 64 != 0 || s.K > 16384) return false;
    if (s.m_groups >= 1) return s.ags == s.K && s.Mw % s.m_groups == 0;
    const int gu = s.gs / 32;   // scale group in units; must be a power of two (shift instead of divide)
    return s.ags == 64 && s.gs >= 64 && s.gs % 64 == 0 && s.K % s.gs == 0 && (gu & (gu - 1)) == 0;
}

void fused_precompute(FusedArgs& a) {
    const Shape& s = a.s;
    a.nu = s.K / 32;
    a.nsb = (a.nu + KL - 1) / KL;
    a.tstride = ((a.nu + 15) & ~15) + 1;
    a.G = s.K / s.ags;
    a.nsg = s.gs > 0 ? s.K / s.gs : 1;
    a.gs_shift = 0;
    if (s.gs > 0) for (int g = s.gs / 32; g > 1; g >>= 1) ++a.gs_shift;
}

size_t fused_lds_bytes(const Shape& s) {
    const int nu = s.K / 32, nu_pad = (nu + 15) & ~15, G = s.K / s.ags;
    return (size_t)4 * (nu_pad + 1) * 16 + sizeof(float) * (2 * G + 2 * FW * RL * 4 * 4 + FW + s.K / 32);
}

size_t qlut_lds_u4(int K) {
    const int nu = K / 32, nu_pad = (nu + 15) & ~15;
    return (size_t)4 * (nu_pad + 1);
}

template <int BITS, bool ZP, int SM, int LUTSRC, int ACC>
static hipError_t launch_nr(const FusedArgs& a, int total_nb, int N, hipStream_t st) {
    const size_t shmem = fused_lds_bytes(a.s);
    // persistent workgroups: at most 2 per CU (256 CUs); each builds the LUT once and walks its row blocks
    dim3 g(total_nb < 512 ? total_nb : 512, N), b(FT);
    const int T = a.s.K / 4;
    if (LUTSRC == 0 || T <= 2 * FT) hipLaunchKernelGGL((k_gemv_fused<BITS, ZP, SM, LUTSRC, 2, ACC>), g, b, shmem, st, a);
    else if (T <= 6 * FT) hipLaunchKernelGGL((k_gemv_fused<BITS, ZP, SM, LUTSRC, 6, ACC>), g, b, shmem, st, a);
    else hipLaunchKernelGGL((k_gemv_fused<BITS, ZP, SM, LUTSRC, 8, ACC>), g, b, shmem, st, a);
    return hipGetLastError();
}

template <int BITS, int LUTSRC>
static hipError_t launch_b(const FusedArgs& a, int total_nb, int N, hipStream_t st) {
    if (a.s.m_groups >= 1) return launch_nr<BITS, false, 2, LUTSRC, 0>(a, total_nb, N, st);
    if (a.acc_mfma)
        return a.s.zero_point ? launch_nr<BITS, true, 0, LUTSRC, 1>(a, total_nb, N, st) : launch_nr<BITS, false, 0, LUTSRC, 1>(a, total_nb, N, st);
    return a.s.zero_point ? launch_nr<BITS, true, 0, LUTSRC, 0>(a, total_nb, N, st) : launch_nr<BITS, false, 0, LUTSRC, 0>(a, total_nb, N, st);
}

hipError_t launch_gemv_fused(const FusedArgs& a_in, int N, bool build_lut, hipStream_t st) {
    if (!gemv_fused_supported(a_in.s) || a_in.nmat < 1 || a_in.nmat > 4) return hipErrorInvalidValue;
    FusedArgs a = a_in;
    fused_precompute(a);
    const int total_nb = a.m[a.nmat - 1].nb_end;
#define DISPATCH(B)                                                     \
    case B: return build_lut ? launch_b<B, 1>(a, total_nb, N, st) : launch_b<B, 0>(a, total_nb, N, st);
    switch (a.s.bits) {
        DISPATCH(1) DISPATCH(2) DISPATCH(3) DISPATCH(4)
    }
#undef DISPATCH
    return hipErrorInvalidValue;
}

}  // namespace tmac
