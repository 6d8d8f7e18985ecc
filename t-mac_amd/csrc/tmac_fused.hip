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

constexpr int FT = 512;        // threads per workgroup
constexpr int FW = FT / 64;    // waves
constexpr int UPS = FW * KL;   // units per step (128)

template <int BITS>
struct WFrag { uint32_t wd[8 * BITS / 2]; };

template <int BITS>
__device__ __forceinline__ void load_w(WFrag<BITS>& f, const uint4* W, const Shape& s, int b, int ub, int rl, int ul) {
    constexpr int NJ = 8 * BITS / 8;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(W + weight_u4_index(s, b, ub, j, rl, ul)));
        f.wd[4 * j] = v.x; f.wd[4 * j + 1] = v.y; f.wd[4 * j + 2] = v.z; f.wd[4 * j + 3] = v.w;
    }
}

// SM 0: per-(row, group) scales (+ zero points), act group 64.   SM 2: unified scale applied last (ags == K).
template <int BITS, bool ZP, int SM, int LUTSRC, int NR>
__global__ __launch_bounds__(FT) void k_gemv_fused(FusedArgs a) {
    extern __shared__ uint4 lds[];
    const Shape s = a.s;  // s.Mw is not meaningful here (per-matrix Mw in a.m[])
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, rl = lane >> 4, ul = lane & 15;
    const int n = blockIdx.y;
    const int T = s.K / 4, nu = s.K / 32, nu_pad = (nu + 15) & ~15, G = s.K / s.ags;
    const int tstride = nu_pad + 1;
    uint4* tab = lds;                                             // [4][tstride]
    float* l_ls = reinterpret_cast<float*>(lds + 4 * tstride);    // [G]   (SM2: [1])
    float* l_lb = l_ls + G;                                       // [G]
    float* l_red = l_lb + G;                                      // [FW][RL][4] floats / ints, then build scratch
    float* l_scr = l_red + FW * RL * 4 * 4;                       // SM2 build: [FW] maxima + [T/8] chunk sums

    // ---- which matrix / row block ---------------------------------------------------------------
    int mi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < a.nmat && (int)blockIdx.x >= a.m[i - 1].nb_end) mi = i;
    const FusedMat& M = a.m[mi];
    const int b = blockIdx.x - (mi ? a.m[mi - 1].nb_end : 0);
    Shape sm = s;
    sm.Mw = M.Mw;
    const int rq = b * RL + rl;

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

    // ---- 2. first weight fragments ------------------------------------------------------------
    const int nsteps = (nu + UPS - 1) / UPS;
    WFrag<BITS> f0, f1;
    {
        const int u = w * KL + ul;
        if (u < nu) load_w<BITS>(f0, M.W, sm, b, w, rl, ul);
        if (UPS + u < nu) load_w<BITS>(f1, M.W, sm, b, FW + w, rl, ul);
    }

    // ---- 3. LUT into LDS ----------------------------------------------------------------------
    if (LUTSRC == 0) {
        const uint4* src = reinterpret_cast<const uint4*>(a.qlut_lds) + (size_t)n * 4 * tstride;
        for (int i = tid; i < 4 * tstride; i += FT) tab[i] = src[i];
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
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
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
                    float mx = __fadd_rn(__fadd_rn(fabsf(x0), fabsf(x1)), __fadd_rn(fabsf(x2), fabsf(x3)));
#pragma unroll
                    for (int m = 1; m < 16; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
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
                const int u = t >> 3, tl = t & 7;
                reinterpret_cast<uint2*>(tab + (tl >> 1) * tstride + u)[tl & 1] = make_uint2(lo, hi);
                // bias: chunk (8 tables) horizontal add in the reference order, then sequential over chunks
                float v = -L15;
                v = __fadd_rn(v, __shfl_xor(v, 4, 64));
                v = __fadd_rn(v, __shfl_xor(v, 2, 64));
                v = __fadd_rn(v, __shfl_xor(v, 1, 64));
                if (SM == 2) {
                    if ((t & 7) == 0) l_scr[FW + (t >> 3)] = v;
                } else {
                    const float c1 = __shfl_xor(v, 8, 64);
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
    __syncthreads();

    // ---- 4. lookups --------------------------------------------------------------------------
    float cacc[2][BITS];
    int32_t iacc[BITS][4];
#pragma unroll
    for (int pl = 0; pl < BITS; ++pl) {
        cacc[0][pl] = 0.f; cacc[1][pl] = 0.f;
#pragma unroll
        for (int be = 0; be < 4; ++be) iacc[pl][be] = 0;
    }
    const int beta0 = 2 * (ul & 1);

    auto compute = [&](const WFrag<BITS>& f, int u) {
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
        const int sg = (u * 32) / s.gs;
        const size_t sidx = dev_scale_index(sm, b, sg, rl, beta0, 0);
        const int per = ZP ? 2 : 1;
#pragma unroll
        for (int pl = 0; pl < BITS; ++pl) {
            // the two halves of the act group live in lanes (ul, ul^1): packed u16 sums add without carry
            uint32_t lo = (uint32_t)acc.a[pl], hi = (uint32_t)(acc.a[pl] >> 32);
            lo += __shfl_xor(lo, 1, 64);
            hi += __shfl_xor(hi, 1, 64);
            const uint32_t mine = (ul & 1) ? hi : lo;   // rows beta0, beta0+1
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int32_t ps = 127 * 16 - (int32_t)((mine >> (16 * i)) & 0xffff);
                if (a.dump) {
                    const int o = 4 * rq + beta0 + i;
                    if (o < M.Mw) a.dump[((size_t)n * M.Mw * BITS + mrow(o, pl, BITS)) * G + kk] = ps;
                }
                const float v = (pl == 0) ? __fmaf_rn((float)ps, ls, lb) : __fmul_rn((float)ps, ls);
                float c = __fmaf_rn(v, ld_scale(M.SC, a.sc_f16, sidx + i * per), cacc[i][pl]);
                if (ZP && pl == 0) c = __fmaf_rn(ld_scale(M.SC, a.sc_f16, sidx + i * per + 1), __fmul_rn(2.0f, lb), c);
                cacc[i][pl] = c;
            }
        }
    };

    for (int i = 0; i < nsteps; i += 2) {
        const int u0 = i * UPS + w * KL + ul;
        if (u0 < nu) compute(f0, u0);
        const int u2 = u0 + 2 * UPS;
        if (u2 < nu) load_w<BITS>(f0, M.W, sm, b, (i + 2) * FW + w, rl, ul);
        const int u1 = u0 + UPS;
        if (u1 < nu) compute(f1, u1);
        const int u3 = u1 + 2 * UPS;
        if (u3 < nu) load_w<BITS>(f1, M.W, sm, b, (i + 3) * FW + w, rl, ul);
    }

    // ---- 5. reduce over unit lanes and waves, store ------------------------------------------
    if (SM != 2) {
        float part[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float acc = __fmul_rn(cacc[i][0], 0.5f);
#pragma unroll
            for (int pl = 1; pl < BITS; ++pl) acc = __fadd_rn(acc, __fmul_rn(cacc[i][pl], f_alpha(pl)));
#pragma unroll
            for (int m = 2; m < KL; m <<= 1) acc = __fadd_rn(acc, __shfl_xor(acc, m, 64));
            part[i] = acc;
        }
        if (ul < 2) {
            l_red[(w * RL + rl) * 4 + beta0] = part[0];
            l_red[(w * RL + rl) * 4 + beta0 + 1] = part[1];
        }
        __syncthreads();
        if (tid < RL * 4) {
            const int o = b * 16 + tid;
            float acc = l_red[tid];
#pragma unroll
            for (int ww = 1; ww < FW; ++ww) acc = __fadd_rn(acc, l_red[ww * RL * 4 + tid]);
            if (o < M.Mw) st_out(M.C, a.out_f16, (size_t)n * M.Mw + o, acc);
        }
    } else {
        int32_t* l_redi = reinterpret_cast<int32_t*>(l_red);
#pragma unroll
        for (int pl = 0; pl < BITS; ++pl)
#pragma unroll
            for (int be = 0; be < 4; ++be) {
                int32_t v = iacc[pl][be];
#pragma unroll
                for (int m = 1; m < KL; m <<= 1) v += __shfl_xor(v, m, 64);
                if (ul == 0) l_redi[((w * RL + rl) * 4 + pl) * 4 + be] = v;
            }
        __syncthreads();
        if (tid < RL * 4) {
            const int r = tid >> 2, be = tid & 3, o = b * 16 + tid;
            if (o < M.Mw) {
                float acc = 0.f;
#pragma unroll
                for (int pl = 0; pl < BITS; ++pl) {
                    int32_t cb = 0;
#pragma unroll
                    for (int ww = 0; ww < FW; ++ww) cb += l_redi[((ww * RL + r) * 4 + pl) * 4 + be];
                    if (a.dump) a.dump[(size_t)n * M.Mw * BITS + mrow(o, pl, BITS)] = cb;
                    const float t = __fmul_rn((float)cb, f_alpha(pl));
                    acc = (pl == 0) ? t : __fadd_rn(acc, t);
                }
                const float v = __fadd_rn(__fmul_rn(acc, l_ls[0]), __fmul_rn(l_lb[0], 0.5f));
                st_out(M.C, a.out_f16, (size_t)n * M.Mw + o, __fmul_rn(v, ld_scale(M.SC, a.sc_f16, o / (M.Mw / s.m_groups))));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
bool gemv_fused_supported(const Shape& s) {
    if (s.bits < 1 || s.bits > 4 || s.K % 64 != 0 || s.K > 16384) return false;
    if (s.m_groups >= 1) return s.ags == s.K && s.Mw % s.m_groups == 0;
    return s.ags == 64 && s.gs >= 64 && s.gs % 64 == 0 && s.K % s.gs == 0;
}

size_t fused_lds_bytes(const Shape& s) {
    const int nu = s.K / 32, nu_pad = (nu + 15) & ~15, G = s.K / s.ags;
    return (size_t)4 * (nu_pad + 1) * 16 + sizeof(float) * (2 * G + FW * RL * 4 * 4 + FW + s.K / 32);
}

size_t qlut_lds_u4(int K) {
    const int nu = K / 32, nu_pad = (nu + 15) & ~15;
    return (size_t)4 * (nu_pad + 1);
}

template <int BITS, bool ZP, int SM, int LUTSRC>
static hipError_t launch_nr(const FusedArgs& a, int total_nb, int N, hipStream_t st) {
    const size_t shmem = fused_lds_bytes(a.s);
    dim3 g(total_nb, N), b(FT);
    const int T = a.s.K / 4;
    if (LUTSRC == 0 || T <= 2 * FT) hipLaunchKernelGGL((k_gemv_fused<BITS, ZP, SM, LUTSRC, 2>), g, b, shmem, st, a);
    else if (T <= 6 * FT) hipLaunchKernelGGL((k_gemv_fused<BITS, ZP, SM, LUTSRC, 6>), g, b, shmem, st, a);
    else hipLaunchKernelGGL((k_gemv_fused<BITS, ZP, SM, LUTSRC, 8>), g, b, shmem, st, a);
    return hipGetLastError();
}

template <int BITS, int LUTSRC>
static hipError_t launch_b(const FusedArgs& a, int total_nb, int N, hipStream_t st) {
    if (a.s.m_groups >= 1) return launch_nr<BITS, false, 2, LUTSRC>(a, total_nb, N, st);
    return a.s.zero_point ? launch_nr<BITS, true, 0, LUTSRC>(a, total_nb, N, st) : launch_nr<BITS, false, 0, LUTSRC>(a, total_nb, N, st);
}

hipError_t launch_gemv_fused(const FusedArgs& a, int N, bool build_lut, hipStream_t st) {
    if (!gemv_fused_supported(a.s) || a.nmat < 1 || a.nmat > 4) return hipErrorInvalidValue;
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
