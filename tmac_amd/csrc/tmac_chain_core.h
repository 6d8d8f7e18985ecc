// tmac_chain_core.h -- device pieces shared by the persistent decode kernels: k_decode_chain (tmac_chain.hip: dependent calls with
// in-kernel hand-offs) and k_gemv_stream (tmac_stream.hip: independent calls, LUT images prebuilt).  Weight fragments of one
// (row quad, 64-unit step) item, their issue, and the item's lookups + MFMA adder + scale chain (tbl.cc:445-526).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "tmac_quad_core.h"
#include "tmac_chain.h"

namespace tmac {

typedef const ChainOp* cop_ptr;   // descriptors: the workgroup's copy in LDS
// Pointers read from the LDS copy are generic to the compiler: without the explicit global address space it emits flat
// loads / stores, which also count on lgkmcnt -- every LDS wait would then wait for global memory.
#define TMAC_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ TMAC_GLOBAL T* as_global(T* p) { return (TMAC_GLOBAL T*)p; }
// A value read from the LDS copy is the same in every lane, but the compiler treats an LDS load as divergent: it
// computes with it in VGPRs and wraps buffer resources in waterfall loops.  readfirstlane states the uniformity.
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
template <typename T>
__device__ __forceinline__ T* uni(T* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

template <int BITS>
struct CFrag {
    u32x4q wq[BITS];     // the lane's 16 bytes of every bit-plane block of the item (dword 4 j + c of the item = wq[j][c])
    uint32_t s0, s1;     // the lane's scale (, zero) of the step's scale group: fp16 pair in s0, or fp32 in s0 (, s1)
};

// weights of (global quad gq, step st) + the lane's scale: the epilogue role of a lane is row lane & 3, units
// st*64 + 16g + 4*lg .. +3 (see k_gemv_quad); scale groups span >= 4 units, so one scale group per lane and step.
// Lanes whose unit lies past K skip the weight load (their LUT entries are zero tables: whatever the registers hold
// contributes exactly 0) -- the zero padding of the last step is stored but never fetched.
template <int BITS, bool ZP, bool SCF16, int SM>
__device__ __forceinline__ void c_issue(CFrag<BITS>& f, __amdgpu_buffer_rsrc_t rs, int woff, const TMAC_GLOBAL char* scq, int nsg, int gsh, int nu,
                                        int st, int lane, uint32_t lane16) {
    constexpr int per = ZP ? 2 : 1;
    constexpr int esz = SCF16 ? 2 : 4;
    const int c0 = 4 * (lane & 12) + 4 * (lane >> 4);
    const uint32_t sg = min((uint32_t)st * (64u >> gsh) + (uint32_t)(c0 >> gsh), (uint32_t)nsg - 1u);
    const uint32_t boff = (sg * 4 + (lane & 3)) * (per * esz);          // scq already points at the quad's first scale group
    uint32_t r0 = 0, r1 = 0;
    if (SM == 0) {                       // (the unified scale is applied once per output, in the epilogue)
        if (SCF16) {
            if (ZP) r0 = *reinterpret_cast<const TMAC_GLOBAL uint32_t*>(scq + boff);
            else r0 = *reinterpret_cast<const TMAC_GLOBAL unsigned short*>(scq + boff);
        } else {
            const TMAC_GLOBAL uint32_t* p32 = reinterpret_cast<const TMAC_GLOBAL uint32_t*>(scq + boff);
            r0 = p32[0];
            if (ZP) r1 = p32[1];
        }
    }
    f.s0 = r0; f.s1 = r1;
    if (st * 64 + lane < nu) {
        // Buffer loads: resource (matrix base) and the fragment's byte offset in SGPRs, the lane's byte offset in a VGPR of
        // its own (lane16, made opaque at kernel entry).  No VALU instruction takes part: when the address arithmetic
        // (a rematerialised lane << 4, or a 64-bit add) lands in a dead ring register, that VALU write makes the
        // compiler wait for every earlier load that might still target the register -- it serialised the fragments of a
        // ring, one full memory latency each (1.4-2.8 us per op, profiles/r02_chain_prefetch_ab.txt B).
        const int soff = woff + st * (BITS * 1024);
#pragma unroll
        for (int j = 0; j < BITS; ++j) {
            f.wq[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lane16, soff + j * 1024, 2 /* nt */);
        }
    }
}


// The same for an item whose 64 units all lie below K (every item but a ragged last step): the lane's scale group is
// (st * 64 >> gsh) + (c0 >> gsh) -- the first term is uniform and goes into the scale POINTER on the scalar unit (scq_item), the second is a
// per-lane constant of the op (v_sc0, with the row's offset folded in): no vector arithmetic, no exec mask (round 6).
template <int BITS, bool ZP, bool SCF16, int SM>
__device__ __forceinline__ void c_issue_full(CFrag<BITS>& f, __amdgpu_buffer_rsrc_t rs, int soff, const TMAC_GLOBAL char* scq_item, uint32_t v_sc0, uint32_t lane16) {
    uint32_t r0 = 0, r1 = 0;
    if (SM == 0) {
        if (SCF16) {
            if (ZP) r0 = *reinterpret_cast<const TMAC_GLOBAL uint32_t*>(scq_item + v_sc0);
            else r0 = *reinterpret_cast<const TMAC_GLOBAL unsigned short*>(scq_item + v_sc0);
        } else {
            const TMAC_GLOBAL uint32_t* p32 = reinterpret_cast<const TMAC_GLOBAL uint32_t*>(scq_item + v_sc0);
            r0 = p32[0];
            if (ZP) r1 = p32[1];
        }
    }
    f.s0 = r0; f.s1 = r1;
#pragma unroll
    for (int j = 0; j < BITS; ++j) f.wq[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lane16, soff + j * 1024, 2 /* nt */);
}

// ---- TMAC_RING_STATIC: the loads of an item as ONE unconditional, branch-free sequence ---------------------------------------------
// With every ring slot (re)filled at fixed program points by exactly the same loads in the same order, the compiler's waitcnt pass can
// count the loads issued behind the one it needs and wait per slot (s_waitcnt vmcnt(n)); with conditional issue it waits for the whole
// ring.  The operands come from c_item_operands() for a real item, or describe a dummy (null resource: every lane out of range, zeros
// without a fetch; the scale word from any mapped address).  Lanes past K re-read lane 0's 16 bytes (one line the wave fetches anyway;
// their LUT entries are zero tables) instead of sitting under an exec-masked branch.
struct CItemOps {
    __amdgpu_buffer_rsrc_t rs;      // matrix (or null) resource
    int soff;                       // byte offset of the fragment's first bit-plane block
    const TMAC_GLOBAL char* sc;     // first scale group of the quad
    uint32_t boff;                  // per lane: byte offset of its scale (, zero) word(s)
    uint32_t l16;                   // per lane: byte offset inside a 1 KB block
};
template <int BITS, bool ZP, bool SCF16, int SM>
__device__ __forceinline__ void c_item_operands(CItemOps& o, __amdgpu_buffer_rsrc_t rs, int woff, const TMAC_GLOBAL char* scq, int nsg, int gsh, int nu,
                                                int st, int lane, uint32_t lane16) {
    constexpr int per = ZP ? 2 : 1;
    constexpr int esz = SCF16 ? 2 : 4;
    const int c0 = 4 * (lane & 12) + 4 * (lane >> 4);
    const uint32_t sg = min((uint32_t)st * (64u >> gsh) + (uint32_t)(c0 >> gsh), (uint32_t)nsg - 1u);
    o.rs = rs; o.sc = scq;
    o.boff = (sg * 4 + (lane & 3)) * (per * esz);
    o.l16 = (st * 64 + lane < nu) ? lane16 : 0u;
    o.soff = woff + st * (BITS * 1024);
}
template <int BITS, bool ZP, bool SCF16, int SM>
__device__ __forceinline__ void c_issue_static(CFrag<BITS>& f, const CItemOps& o) {
    uint32_t r0 = 0, r1 = 0;
    if (SM == 0) {
        if (SCF16) {
            if (ZP) r0 = *reinterpret_cast<const TMAC_GLOBAL uint32_t*>(o.sc + o.boff);
            else r0 = *reinterpret_cast<const TMAC_GLOBAL unsigned short*>(o.sc + o.boff);
        } else {
            const TMAC_GLOBAL uint32_t* p32 = reinterpret_cast<const TMAC_GLOBAL uint32_t*>(o.sc + o.boff);
            r0 = p32[0];
            if (ZP) r1 = p32[1];
        }
    }
    f.s0 = r0; f.s1 = r1;
#pragma unroll
    for (int j = 0; j < BITS; ++j) f.wq[j] = __builtin_amdgcn_raw_buffer_load_b128(o.rs, (int)o.l16, o.soff + j * 1024, 2 /* nt */);
}

// ---- stores and barriers the compiler's waitcnt pass does not see (TMAC_RING_STATIC) ----------------------------------------------
// On gfx9 stores count on vmcnt like loads, and loads and stores complete out of order with respect to each other: as soon as a store is
// pending anywhere on a path, the pass stops counting and waits with vmcnt(0) in front of the next use of a loaded register -- which
// drains the weight ring.  (Real hardware: loads return in order among themselves; a pending store can only make a counted wait longer.)
// Likewise __syncthreads() is a workgroup fence + s_barrier, and the fence is s_waitcnt vmcnt(0) lgkmcnt(0): what the barriers of the
// kernel order is LDS (tables, partial sums), never global memory.
__device__ __forceinline__ void c_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void c_store_b16(unsigned long long gaddr, uint32_t v) { asm volatile("global_store_short %0, %1, off" :: "v"(gaddr), "v"(v) : "memory"); }
__device__ __forceinline__ void c_store_b32(unsigned long long gaddr, uint32_t v) { asm volatile("global_store_dword %0, %1, off" :: "v"(gaddr), "v"(v) : "memory"); }
__device__ __forceinline__ void c_store_b64(unsigned long long gaddr, unsigned long long v) { asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(gaddr), "v"(v) : "memory"); }
__device__ __forceinline__ void c_store_b64_sc1(unsigned long long gaddr, unsigned long long v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(gaddr), "v"(v) : "memory"); }
__device__ __forceinline__ void c_store_b64_sys(unsigned long long gaddr, unsigned long long v) { asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(gaddr), "v"(v) : "memory"); }

// One 64-unit step of a row quad: lookups (v_perm_b32 on the half tables), v_mfma_i32_16x16x64_i8 as the adder, then the
// two act groups of the lane's output row through the fp32 scale chain (compute_mfma of k_gemv_quad, SM = 0), or -- SM = 2 --
// the exact int32 sum of the lane's row over all units, per bit-plane (tbl.cc:586-628).
// Per-group scales: the adder's selector matrix of plane p weighs `all` +2^p and `neg` -2^(p+1) (c_selectors), so ONE accumulator
// receives sum_p 2^p PS_p -- the integer the scale chain wants (Sigma 2^p PS_p, exact in int32: |.| <= 15 * 16 * 127 per act group) --
// instead of one accumulator per plane combined by shifts and adds afterwards.  lane16 = 16 * lane and lk4 = 4 * (2 * (lane & 12) +
// 2 * (lane >> 4)) are the lane's byte offsets into the table rows and the act groups' scale arrays (computed once per kernel: what
// remains per item is one v_add per LDS read).
constexpr int IMG2_ROW = 65;                      // uint4 per table row of a 64-unit step
constexpr int IMG2_STEP = 4 * IMG2_ROW;           // uint4 per step
// uint4 index of (unit, row j4) / byte offset of an item's first table row in the step-major layout
__host__ __device__ inline int img2_index(int unit, int j4) { return (unit >> 6) * IMG2_STEP + j4 * IMG2_ROW + (unit & 63); }
#ifndef TMAC_IMG2_SC
#define TMAC_IMG2_SC 0      // A/B knob, 1: LUT scales / biases of the step-major image interleaved per pair of act groups {ls0, ls1, lb0, lb1}, one 16-byte
                            // read per item instead of two 8-byte ones.  Measured SLOWER (profiles/r06_stream_image.txt): equal for the (quad x 64 units) form,
                            // -10 % for the quarter-walk form (64 lanes reading four 16-byte addresses); the two arrays ls[], lb[] stay
#endif
template <int BITS>
struct CSel { qv4i_t p[BITS]; };
template <int BITS, int SM>
__device__ __forceinline__ void c_selectors(CSel<BITS>& sel, int lane) {
    const int jrel = (lane & 15) - 4 * (lane >> 4);
    const bool on = jrel >= 0 && jrel < 4;
#pragma unroll
    for (int pl = 0; pl < BITS; ++pl) {
        const uint32_t wp = (SM == 0) ? (1u << pl) : 1u;                        // unified scales keep one exact total per plane
        const uint32_t be = on ? (wp << (8 * jrel)) : 0u, bo = on ? (((0x100u - 2u * wp) & 0xffu) << (8 * jrel)) : 0u;   // +2^p | -2^(p+1)
        sel.p[pl] = (qv4i_t){(int)be, (int)bo, (int)be, (int)bo};
    }
}

// TAP (parity instantiations only): the integers of the lane's two act groups, comb = sum_p 2^p PS_p, go to tap_row[act group] (tap_row: the
// lane's output row in the launch's tap buffer, G act groups per row) exactly as they enter the fp32 chain.
// IMG2 (round 6, both persistent kernels): the tables in the STEP-MAJOR layout [64-unit step][4][IMG2_ROW = 65] uint4 -- the four table rows of
// a lane are 1040 bytes apart whatever K is, i.e. immediate offsets of ONE address (tb_off: the item's uniform byte offset into the tables,
// + lane16) instead of four address computations with a run-time row stride (one of them a quarter-rate 64-bit multiply-add).  The 65th
// uint4 of a row is padding: the LUT build of k_decode_chain stores the four rows of a unit from four neighbouring lanes (16 bytes apart
// in the banks, not on top of each other).  (TMAC_IMG2_SC: an A/B knob, below.)
template <int BITS, bool ZP, bool SCF16, int SM, bool TAP = false, bool IMG2 = false>
__device__ __forceinline__ void c_compute(const CFrag<BITS>& f, const uint4* tab, int tstride, const float* l_ls, const float* l_lb,
                                          int ub, uint32_t lane16, uint32_t lk4, const CSel<BITS>& sel, uint32_t k3, float& cacc, int32_t (&iacc)[BITS],
                                          int32_t* tap_row = nullptr, int G = 0, int tb_off = 0) {
    // ub: first unit of the item's table rows (64 x step for a (quad, 64-unit step) item, 16 x quarter-step for k_gemv_stream's quarter-walk form,
    // where lane16 = 16 (lane & 15) and lk4 = 8 (lane >> 4)); act groups ub / 2 + lk4 / 4 + {0, 1}
    uint32_t tb[16];
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4) {
        // units past K read the zero tables: no contribution
        const uint4 v = IMG2 ? *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(tab) + tb_off + lane16 + j4 * (16 * IMG2_ROW))
                             : *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(tab + (j4 * tstride + ub)) + lane16);
        tb[4 * j4] = v.x; tb[4 * j4 + 1] = v.y; tb[4 * j4 + 2] = v.z; tb[4 * j4 + 3] = v.w;
    }
    constexpr int NACC = (SM == 0) ? 1 : BITS;
    qv4i_t c[NACC];
#pragma unroll
    for (int pl = 0; pl < NACC; ++pl) c[pl] = (qv4i_t){0, 0, 0, 0};
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
#pragma unroll
        for (int pl = 0; pl < BITS; ++pl) {
            uint32_t pa, ma, pb, mb;
            const int qa = (2 * tp) * BITS + pl, qb = (2 * tp + 1) * BITS + pl;
            if (qa & 1) q_lookup4_pm<1>(f.wq[qa >> 3][(qa >> 1) & 3], tb[4 * tp], tb[4 * tp + 1], k3, pa, ma);
            else q_lookup4_pm<0>(f.wq[qa >> 3][(qa >> 1) & 3], tb[4 * tp], tb[4 * tp + 1], k3, pa, ma);
            if (qb & 1) q_lookup4_pm<1>(f.wq[qb >> 3][(qb >> 1) & 3], tb[4 * tp + 2], tb[4 * tp + 3], k3, pb, mb);
            else q_lookup4_pm<0>(f.wq[qb >> 3][(qb >> 1) & 3], tb[4 * tp + 2], tb[4 * tp + 3], k3, pb, mb);
            constexpr int acc = 0;
            qv4i_t& cd = c[(SM == 0) ? acc : pl];
            cd = __builtin_amdgcn_mfma_i32_16x16x64_i8((qv4i_t){(int)pa, (int)ma, (int)pb, (int)mb}, sel.p[pl], cd, 0, 0, 0);
        }
    }
    if constexpr (SM == 2) {
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
        for (int pl = 0; pl < BITS; ++pl) iacc[pl] += (c[pl].x + c[pl].y) + (c[pl].z + c[pl].w);
        return;
    } else {
        float sc, zr = 0.f;
        if (SCF16) {
            sc = __half2float(__ushort_as_half((unsigned short)(f.s0 & 0xffff)));
            if (ZP) zr = __half2float(__ushort_as_half((unsigned short)(f.s0 >> 16)));
        } else {
            sc = __uint_as_float(f.s0);
            if (ZP) zr = __uint_as_float(f.s1);
        }
        // act groups ub / 2 + lk4 / 4 + {0, 1}: ls / 2 and lb / 2 (groups past K hold zeros)
        float2 hls2, hlb2;
        if constexpr (IMG2 && TMAC_IMG2_SC) {
            const float4 q4 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(l_ls + 2 * (ub >> 1)) + 2 * lk4);
            hls2 = make_float2(q4.x, q4.y); hlb2 = make_float2(q4.z, q4.w);
        } else {
            hls2 = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(l_ls + (ub >> 1)) + lk4);
            hlb2 = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(l_lb + (ub >> 1)) + lk4);
        }
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const float hls = gi ? hls2.y : hls2.x, hlb = gi ? hlb2.y : hlb2.x;
            // sum_p alpha_p [(ps_p ls + [p = 0] lb) scale + [p = 0] zero 2 lb] = ((sum_p 2^p ps_p)(ls / 2) + lb / 2) scale + (2 zero)(lb / 2)
            const int32_t comb = (gi == 0) ? (c[0].x + c[0].y) : (c[0].z + c[0].w);
            if constexpr (TAP) {
                const int kk = (ub >> 1) + (int)(lk4 >> 2) + gi;
                if (tap_row && kk < G) tap_row[kk] = comb;
            }
            const float v = __fmaf_rn((float)comb, hls, hlb);
            float cc = __fmaf_rn(v, sc, cacc);
            if (ZP) cc = __fmaf_rn(__fadd_rn(zr, zr), hlb, cc);
            cacc = cc;
        }
    }
}


}  // namespace tmac
