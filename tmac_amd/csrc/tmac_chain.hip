// tmac_chain.hip — k_decode_chain: a recorded sequence of fused decode GEMV groups (N = 1) as ONE persistent launch.
//
// Why: a decoded token of llama-2-7B is 128 dependent launches of 4.7-25 MB each; as separate launches every one of them
// pays the dependent-dispatch gap, fetches its activations, builds its LUT and only then starts its weight stream, so HBM
// idles through most of a launch (0.78 ms per token against 0.45 ms for launches that merely read the
// bytes).  Here one workgroup per CU walks the whole op list, and inside a workgroup the phases of a call -- fetch the
// activations, build the LUT, look up, combine and publish -- belong to different WAVES, so that they overlap across calls:
//   * NBW builder waves run up to one call ahead of the lookups: for every block of 64 LUT pairs (a quarter of a 64-unit step,
//     512 activations) they poll the hand-off granules (data is the flag: self-tagged 8-byte granules {generation, 2 x fp16},
//     written through with sc1 stores and read with sc1 loads; cdna_hip_programming.md Guideline 16, recipe R2), build the
//     tables (lut_ctor.cc:120-215) into the LDS buffer of the call's parity and raise the block's flag.  While the lookup waves
//     are busy with call i, the builders already sit in the polls of call i + 1: the fabric round trip of the poll, the table
//     build and the lookups of the previous call overlap instead of following each other;
//   * NLW lookup waves walk the (row quad, 64-unit step) items of a call (tbl.cc:445-462 on v_perm_b32, the MFMA as adder,
//     the per-act-group scale chain tbl.cc:479-526).  An item needs the four block flags of its step only, so the lookups of
//     step 0 start while the later steps' tables are still being built or their activations have not arrived yet;
//   * ONE publisher wave combines the partial sums of quads whose steps are split over several waves (in wave order, as
//     k_gemv_quad does), stores the outputs and publishes the granules of a workgroup iteration with one store instruction.
//     Lookup waves leave their partials in one of NPAR reduction buffers and go on -- nobody waits at a barrier.
// The roles synchronise through flags in LDS (values that only grow: call index, iteration count), never through s_barrier.
// Row quads are dealt out iteration-major (tmac_chain.h): the rows of a call complete in ascending order chip-wide.
// Arithmetic is that of k_gemv_quad (tmac_quad.hip) — same LUT build, same lookup + MFMA adder, same scale chain, same
// lane/wave decomposition for a given number of waves per quad — so results are bit-identical to the per-launch path with
// 64 * NLW-thread workgroups.
// Scope: 1- to 4-bit weights, fp16 activations; SM = 0: per-group scales with act groups of 64 and scale groups >= 128 (the
// GPTQ-style path, tbl.cc:323-532); SM = 2: unified scale(s), one act group per row, exact int32 totals and the scale-final
// epilogue (BitNet: tbl.cc:536-630, qgemm.py:170-174,192-206).
// Deadlock freedom: every wait is for something an EARLIER stage produces -- builders wait for granules (publishers of earlier
// calls, any workgroup) and for the LDS buffer (own publisher, two calls back), lookup waves for block flags (own builders, same
// call) and a free reduction buffer (own publisher, NPAR iterations back), the publisher for the lookup waves' arrival -- so by
// induction over the call index everything completes provided all workgroups are resident; the grid is one workgroup per CU and
// the kernel's register / LDS footprint admits exactly one.  Every spin is bounded (ChainArgs::spin_limit) and reports through ctl[2].
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "tmac_quad_core.h"
#include "tmac_chain.h"

// A/B knobs (tools/build_variant.sh): s_sleep count in the LDS flag waits, issue priority of the builder / publisher waves
#ifndef TMAC_CHAIN_SPIN_SLEEP
#define TMAC_CHAIN_SPIN_SLEEP 2
#endif
#ifndef TMAC_CHAIN_AUX_PRIO
#define TMAC_CHAIN_AUX_PRIO 3
#endif

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
    uint32_t wd[4 * BITS];
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
            const u32x4q v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lane16, soff + j * 1024, 2 /* nt */);
            f.wd[4 * j] = v.x; f.wd[4 * j + 1] = v.y; f.wd[4 * j + 2] = v.z; f.wd[4 * j + 3] = v.w;
        }
    }
}

// One 64-unit step of a row quad: lookups (v_perm_b32 on the half tables), v_mfma_i32_16x16x64_i8 as the adder, then the
// two act groups of the lane's output row through the fp32 scale chain (compute_mfma of k_gemv_quad, SM = 0), or -- SM = 2 --
// the exact int32 sum of the lane's row over all units, per bit-plane (tbl.cc:586-628).
template <int BITS, bool ZP, bool SCF16, int SM>
__device__ __forceinline__ void c_compute(const CFrag<BITS>& f, const uint4* tab, int tstride, const float* l_ls, const float* l_lb,
                                          int st, int lane, qv4i_t bsel, uint32_t k3, float& cacc, int32_t (&iacc)[BITS]) {
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
    if (SM == 2) {
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
    }
    float sc, zr = 0.f;
    if (SCF16) {
        sc = __half2float(__ushort_as_half((unsigned short)(f.s0 & 0xffff)));
        if (ZP) zr = __half2float(__ushort_as_half((unsigned short)(f.s0 >> 16)));
    } else {
        sc = __uint_as_float(f.s0);
        if (ZP) zr = __uint_as_float(f.s1);
    }
    const int ub4 = st * 64 + 4 * (lane & 12) + 4 * (lane >> 4);
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
        const int kk = (ub4 + 2 * gi) >> 1;
        const float hls = l_ls[kk], hlb = l_lb[kk];           // ls / 2, lb / 2; groups past K hold zeros
        // sum_p alpha_p [(ps_p ls + [p = 0] lb) scale + [p = 0] zero 2 lb] = ((sum_p 2^p ps_p)(ls / 2) + lb / 2) scale + (2 zero)(lb / 2)
        int32_t comb = 0;
#pragma unroll
        for (int pl = BITS - 1; pl >= 0; --pl) {
            const int32_t ps = (gi == 0) ? (c[pl].x + c[pl].y) : (c[pl].z + c[pl].w);
            comb = (pl == BITS - 1) ? ps : (int32_t)(((uint32_t)comb << 1) + (uint32_t)ps);
        }
        const float v = __fmaf_rn((float)comb, hls, hlb);
        float cc = __fmaf_rn(v, sc, cacc);
        if (ZP) cc = __fmaf_rn(__fadd_rn(zr, zr), hlb, cc);
        cacc = cc;
    }
}

// The hand-off granules of LUT pairs (one pair = two consecutive row quads = 8 activations): loads and their wait in one statement
// (cdna_hip_programming.md 5.7, form (i)).  Agent scope (sc1) bypasses this CU's L1 and sees what other XCDs wrote through; chains that
// span several GPUs poll at system scope (sc0 sc1): the granules then arrive over xGMI from the peers' producers.
// Three or six pairs per call: all blocks of a builder's batch fly together (polled one after the other, every block would cost
// a fabric round trip of its own).
#define TMAC_POLL_FN(SUFFIX, SC)                                                                                                \
    __device__ __forceinline__ void c_poll3##SUFFIX(const uint4* const (&p)[6], u32x4q (&v)[12]) {                              \
        asm volatile("global_load_dwordx4 %0, %6, off " SC "\n\t"                                                             \
                     "global_load_dwordx4 %1, %6, off offset:16 " SC "\n\t"                                                   \
                     "global_load_dwordx4 %2, %7, off " SC "\n\t"                                                             \
                     "global_load_dwordx4 %3, %7, off offset:16 " SC "\n\t"                                                   \
                     "global_load_dwordx4 %4, %8, off " SC "\n\t"                                                             \
                     "global_load_dwordx4 %5, %8, off offset:16 " SC "\n\t"                                                   \
                     "s_waitcnt vmcnt(0)"                                                                                       \
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5])                             \
                     : "v"(p[0]), "v"(p[1]), "v"(p[2]) : "memory");                                                             \
    }
TMAC_POLL_FN(, "sc1")
TMAC_POLL_FN(_sys, "sc0 sc1")
#undef TMAC_POLL_FN

// ---- flags in LDS.  Values only grow (call index + 1, iteration count + 1), writers store them AFTER the data they announce
// (LDS executes a wave's instructions in order), readers branch on them BEFORE touching that data; the asm memory clobbers keep the
// compiler from moving LDS accesses across them.  Relaxed accesses: a workgroup-scope release / acquire would also wait for the
// wave's global loads in flight (the weight ring). ----
__device__ __forceinline__ unsigned lds_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#define TMAC_CBAR() asm volatile("" ::: "memory")
// four consecutive flag words (16-byte aligned) all equal to `want`; every lane reads the same address
__device__ __forceinline__ bool lds_flags4_eq(const unsigned* p, unsigned want) {
    u32x4q v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p) : "memory");
    return __builtin_amdgcn_readfirstlane((v.x == want) & (v.y == want) & (v.z == want) & (v.w == want)) != 0;
}
// One block of the LUT (64 pairs = 128 tables, lut_ctor.cc:120-215) from the 8 fp16 activations of the lane's pair (xw: 4 dwords).
// SM 0: act groups of 64 = 8 pairs = 8 lanes (K is a multiple of 64, so a group is never cut by the end of K); SM 2: the row's scale.
template <int SM>
__device__ __forceinline__ void c_build_block(uint4* tab, int tstride, float* l_ls, float* l_lb, int P, int blk, int lane, const uint32_t (&xw)[4],
                                              float gscale, float gtinv) {
    const int p = blk * 64 + lane;
    const bool act = p < P;
    float x[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t wv = act ? xw[q] : 0u;
        const __half2 hh = *reinterpret_cast<const __half2*>(&wv);
        x[2 * q] = __low2float(hh); x[2 * q + 1] = __high2float(hh);
    }
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
    q_table8<true>(x[0], x[1], x[2], x[3], t_scales, lo0, hi0, La);
    q_table8<true>(x[4], x[5], x[6], x[7], t_scales, lo1, hi1, Lb);
    // units past K get zero tables / zero LUT scales: the padded lanes of the last step contribute exactly 0
    tab[(p & 3) * tstride + (p >> 2)] = act ? make_uint4(lo0, hi0, lo1, hi1) : make_uint4(0u, 0u, 0u, 0u);
    if (SM != 2) {
        // lut_biases (lut_ctor.cc:25-31): per 8-table chunk ((v0+v4)+(v2+v6)) + ((v1+v5)+(v3+v7)), v_i = -L15 of table i
        float va = -La, vb = -Lb;
        va = __fadd_rn(va, qdpp_f<0x4E>(va));
        vb = __fadd_rn(vb, qdpp_f<0x4E>(vb));
        va = __fadd_rn(va, qdpp_f<0xB1>(va));
        vb = __fadd_rn(vb, qdpp_f<0xB1>(vb));
        const float v = __fadd_rn(va, vb);
        const float c1 = qdpp_f<0x104>(v);      // row_shl:4: the second chunk of the act group
        if ((p & 7) == 0) {
            l_ls[p >> 3] = act ? __fmul_rn(0.5f, scales) : 0.f;
            l_lb[p >> 3] = act ? __fmul_rn(0.5f, __fadd_rn(__fadd_rn(0.0f, v), c1)) : 0.f;
        }
    }
}
// unified scale: what a block contributes to the row's lut_scales (its maximum |x0|+|x1|+|x2|+|x3| over the tables) and to lut_biases
// (the chunk sums of lut_ctor.cc:25-31, one per 4 pairs) -- neither depends on the scale
__device__ __forceinline__ void c_block_stats(float* l_us, int P, int blk, int lane, const uint32_t (&xw)[4]) {
    const int p = blk * 64 + lane;
    float mx = 0.f;
    float x[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t wv = p < P ? xw[q] : 0u;
        const __half2 hh = *reinterpret_cast<const __half2*>(&wv);
        x[2 * q] = __low2float(hh); x[2 * q + 1] = __high2float(hh);
    }
    mx = fmaxf(mx, __fadd_rn(__fadd_rn(fabsf(x[0]), fabsf(x[1])), __fadd_rn(fabsf(x[2]), fabsf(x[3]))));
    mx = fmaxf(mx, __fadd_rn(__fadd_rn(fabsf(x[4]), fabsf(x[5])), __fadd_rn(fabsf(x[6]), fabsf(x[7]))));
    float va = -__fadd_rn(__fadd_rn(__fadd_rn(x[0], x[1]), x[2]), x[3]);
    float vb = -__fadd_rn(__fadd_rn(__fadd_rn(x[4], x[5]), x[6]), x[7]);
    va = __fadd_rn(va, qdpp_f<0x4E>(va));      // lane ^ 2: v0+v4 | v2+v6      (lut_ctor.cc:25-31)
    vb = __fadd_rn(vb, qdpp_f<0x4E>(vb));      //           v1+v5 | v3+v7
    va = __fadd_rn(va, qdpp_f<0xB1>(va));      // lane ^ 1: (v0+v4)+(v2+v6)
    vb = __fadd_rn(vb, qdpp_f<0xB1>(vb));      //           (v1+v5)+(v3+v7)
    if ((p & 3) == 0 && p < P) l_us[CHAIN_US_FLOATS + (p >> 2)] = __fadd_rn(va, vb);
    mx = q_row_allmax(mx);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (lane == 0) l_us[blk] = mx;
}
// unified scale: lut_scales of the row from the blocks' maxima (every wave that needs it computes the same value)
__device__ __forceinline__ float c_row_scale(const float* l_us, int nreal, int lane) {
    float mx = lane < nreal ? l_us[lane] : 0.f;          // nreal <= CHAIN_MAX_BLK <= 64
    mx = q_row_allmax(mx);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    return div127(mx);
}

// layout of the synchronisation words (CHAIN_SYNC_WORDS, zeroed at kernel entry)
constexpr int SY_BLK = 0;                               // [2][CHAIN_MAX_BLK]  state of block b of the LUT of a call with that parity: 4 (call + 1) + 1 its raw
                                                        //   activations are parked in its table slots, + 2 a wave has claimed the build, + 3 built
constexpr int SY_ARR = 2 * CHAIN_MAX_BLK;               // [NPAR][16]          lookup wave w has left its partials of iteration g in buffer g % NPAR: g + 1
constexpr int SY_MISC = SY_ARR + CHAIN_NPAR * 16;
constexpr int SY_CONS = SY_MISC + 16;                   // [NPAR]              the publisher is done with buffer g % NPAR of iteration g: g + 1
constexpr int SY_PDONE = SY_MISC + 4;                   //                     every lookup wave and the publisher are done with call i: i + 1
constexpr int SY_ABORT = SY_MISC + 5;                   //                     a wait of this workgroup gave up: stop waiting everywhere
constexpr int SY_PUBG = SY_MISC + 3;                    //                     workgroup iterations (all calls) this workgroup has published so far
constexpr int SY_BIAS = SY_MISC + 6;                    // [2]                 unified scale: lut_biases of the call with that parity is final: call + 1
constexpr int SY_BSYNC = SY_MISC + 8;                   // [8]                 unified scale: builder j has left its maximum and chunk sums: call + 1
static_assert(CHAIN_NPAR == 4 && CHAIN_NBW <= 8 && CHAIN_NLW <= 16, "layout of the synchronisation words");

template <int BITS, bool ZP, bool SCF16, int SM>
__global__ __launch_bounds__(CHAIN_FT) void k_decode_chain(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    constexpr int NLW = CHAIN_NLW, NBW = CHAIN_NBW, NPAR = CHAIN_NPAR;
    constexpr int RING = (BITS <= 2) ? 4 : 2;       // weight fragments in flight per lookup wave
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int bx = blockIdx.x, gx = gridDim.x;
    const unsigned gen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(a.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const unsigned long long par_off = (gen & 1u) ? a.arena_half : 0ull;    // this launch's half of the hand-off arena
    unsigned* l_sync = reinterpret_cast<unsigned*>(lds + 2 * (size_t)a.buf_u4);
    float* l_red = reinterpret_cast<float*>(l_sync + CHAIN_SYNC_WORDS);     // [NPAR][NLW][4][CHAIN_RED] partials of split quads (SM 2: per bit-plane)
    // all op descriptors into LDS: a field is then a ds_read away instead of a scalar-cache miss
    uint4* l_ops = reinterpret_cast<uint4*>(l_red + NPAR * NLW * 4 * CHAIN_RED);
    {
        const uint4* gsrc = reinterpret_cast<const uint4*>(a.ops);
        constexpr int U4 = sizeof(ChainOp) / 16;
        for (int idx = tid; idx < a.nops * U4; idx += CHAIN_FT) l_ops[idx] = gsrc[idx];
        for (int idx = tid; idx < CHAIN_SYNC_WORDS; idx += CHAIN_FT) l_sync[idx] = 0u;
        __syncthreads();                       // the only workgroup barrier of the kernel
    }
    const cop_ptr ops = reinterpret_cast<cop_ptr>(l_ops);
    unsigned* const l_abort = l_sync + SY_ABORT;
    const unsigned lds_limit = a.spin_limit > (1u << 27) ? 0xffffffffu : a.spin_limit * 16u;     // LDS spins are ~10 x shorter than a poll
    const bool sys = a.npeer > 0 || a.poll_mode == 1;                       // granules written by other GPUs: polls at system scope

    // stamps: s_memrealtime (100 MHz, one clock for the whole device; s_memtime counts per XCD with unrelated offsets)
#define CSTAMPV(i, k, v) do { if (a.stamps && lane == 0) a.stamps[((size_t)(i) * gx + bx) * 16 + (k)] = (v); } while (0)
#define CSTAMP(i, k) CSTAMPV(i, k, __builtin_amdgcn_s_memrealtime())
    // a bounded wait on flags in LDS: `cond` is re-evaluated (it reads LDS) until it holds, the workgroup has given up, or the limit
    // is reached -- then the error word is set (bit 31 | call << 8 | wave) and every other wait of the launch falls through
#define LDS_WAIT(cond, i)                                                                                                         \
    do {                                                                                                                          \
        unsigned spins_ = 0;                                                                                                      \
        while (!(cond)) {                                                                                                         \
            if (lds_ld(l_abort) != 0u) break;                                                                                     \
            if (++spins_ >= lds_limit) {                                                                                          \
                if (lane == 0) { atomicOr(a.ctl + 2, 0x80000000u | ((unsigned)(i) << 8) | (unsigned)w); lds_st(l_abort, 1u); }     \
                break;                                                                                                            \
            }                                                                                                                     \
            __builtin_amdgcn_s_sleep(TMAC_CHAIN_SPIN_SLEEP);                                                                      \
        }                                                                                                                         \
        TMAC_CBAR();                                                                                                              \
    } while (0)

    // the row quads of workgroup iteration `it` of a call: n quads from lo on (iteration-major partition, tmac_chain.h)
    struct Part { int niter, base, rem, cA, cB, nA, nB; };
    auto part_of = [&](cop_ptr d) __attribute__((always_inline)) {
        Part p;
        p.niter = uni(d->niter); p.base = uni(d->sb_base); p.rem = uni(d->sb_rem);
        const int perA = uni(d->perA), exA = uni(d->exA), perB = uni(d->perB), exB = uni(d->exB);
        p.nA = perA + (bx < exA ? 1 : 0); p.nB = perB + (bx < exB ? 1 : 0);
        p.cA = bx * perA + min(bx, exA);                 // this workgroup's offset inside a super-block of sb_base + 1 quads
        p.cB = p.rem + bx * perB + min(bx, exB);         // ... of sb_base quads, plus the sb_rem extra quads in front of it
        return p;
    };
    auto part_lo = [&](const Part& p, int it) __attribute__((always_inline)) {     // super-block it starts at it * base + min(it, rem)
        return it * p.base + (it < p.rem ? it + p.cA : p.cB);
    };

    // ---- The LUT of call i (lut_ctor.cc:120-215), block by block (block b = LUT pairs 64 b .. 64 b + 63; pair p = tables 2p, 2p + 1
    // from activations 8p .. 8p + 7; four blocks per 64-unit step).  Whoever CLAIMS a block fetches its activations -- for a
    // handed-over vector the poll that detects the granules IS the load -- builds its tables into the LDS buffer of the call's
    // parity and raises its flag.  A block has two candidates: lookup wave b % NLW, which claims on entering the call (a wave that
    // would only wait for the LUT builds it instead: in a chain of strictly dependent calls all lookup waves sit here when the
    // activations arrive, and they arrive in registers), and builder b % NBW, which claims when the vector is due and works through
    // whatever is left (lookup waves still busy with the previous call: the builders run ahead).  Block state, per LUT buffer:
    // below 4 (i + 1): not claimed for call i; + 0 claimed, being fetched (unified scale); + 1 raw activations parked, statistics
    // left (unified scale); + 2 claimed for the build; + 3 built.
    // Unified scale (SM 2): the row's scale is a maximum over ALL blocks, so the build of any block waits until every block is
    // parked; the claimer parks the raw activations in the block's own table slots (same 16 bytes per pair).
    // Runs the candidates c0, c0 + cs, ... (< nblk), three per batch (their polls fly together); all_batches: until none is left.
    auto lut_duty = [&](int i, cop_ptr d, int c0, int cs, bool all_batches, unsigned long long& polls, bool stamp, auto&& after_fetch) __attribute__((always_inline)) {
        const int tstride = uni(d->tstride), nst = uni(d->nst), GP = uni(d->GP);
        uint4* tab = lds + (size_t)(i & 1) * a.buf_u4;
        float* l_ls = reinterpret_cast<float*>(tab + 4 * tstride);
        float* l_lb = l_ls + GP;
        float* l_us = l_ls;                       // SM 2: [blk] the blocks' maxima, [CHAIN_US_FLOATS + c] the chunk sums of the bias chain
        unsigned* bfl = l_sync + SY_BLK + (i & 1) * CHAIN_MAX_BLK;
        const unsigned base = 4u * ((unsigned)i + 1u);
        const int P = uni(d->K) / 8, nblk = 4 * nst, nreal = (P + 63) >> 6;
        const bool gran = (uni(d->in_gran) & 1) != 0;
        const uint4* in4 = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(uni(d->in)) + (gran ? par_off : 0ull));
        bool armed = false;                       // the once-per-call waits are behind this wave
        // The LDS buffer of this parity -- tables AND block states -- belongs to call i - 2 until that call is through (lookups finished,
        // outputs combined): a wave that ran ahead (no rows of its own in the calls in between) must not even claim before that.
        if (i >= 2 && c0 < nblk) LDS_WAIT(lds_ld(l_sync + SY_PDONE) + 1u >= (unsigned)i, i);
        // state below `below` -> `to` (compare-and-swap by lane 0)
        auto claim = [&](int b, unsigned below, unsigned to) __attribute__((always_inline)) {
            unsigned won = 0u;
            if (lane == 0) {
                const unsigned old = lds_ld(bfl + b);
                if (old < below) won = atomicCAS(bfl + b, old, to) == old ? 1u : 0u;
            }
            return __builtin_amdgcn_readfirstlane(won) != 0u;
        };
        for (int first = c0; first < nblk; first += 3 * cs) {
            int blk[3];
            bool any = false;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int c = first + k * cs;
                blk[k] = (c < nblk && claim(c, base, SM == 2 ? base : base + 2u)) ? c : -1;
                any |= blk[k] >= 0;
            }
            if (any) {
                if (!armed) {
                    // First poll: not before this workgroup itself has published the iteration that completes the rows this call
                    // reads (the workgroups run within a fraction of a microsecond of each other, so that is when the last rows
                    // appear chip-wide) -- a poll that samples the image just too early costs a whole fabric round trip.
                    if (gran) {
                        const unsigned sg = (unsigned)uni(d->src_g);
                        if (sg && (all_batches || !(a.poll_mode & 2))) LDS_WAIT(lds_ld(l_sync + SY_PUBG) >= sg, i);     // (A/B knob: poll_mode bit 1 = lookup waves poll at once)
                        for (int z = 0; z < a.poll_delay; ++z) __builtin_amdgcn_s_sleep(1);      // A/B knob: wait before a call's first poll
                    }
                    if (stamp) CSTAMP(i, 11);
                    armed = true;
                }
                // ---- fetch: the 8 activations of pair 64 blk + lane as 4 dwords of fp16 pairs, for the claimed blocks that reach below K
                uint32_t xw[3][4];
                const uint4* gp[6];
                bool need[3];
                int nfetch = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const bool real = blk[k] >= 0 && blk[k] * 64 < P;
                    const int p = (real ? blk[k] : 0) * 64 + lane;               // absent blocks re-read block 0 (result ignored)
                    need[k] = real && p < P;
                    gp[k] = in4 + (gran ? 2 : 1) * (size_t)min(p, P - 1);
                    gp[k + 3] = gp[k];
                    nfetch += real ? 1 : 0;
                }
                if (nfetch > 0) {
                    if (gran) {
                        u32x4q v[12];
                        unsigned spins = 0;
                        for (;;) {
                            ++polls;
                            if (sys) c_poll3_sys(gp, v); else c_poll3(gp, v);
                            bool ok = true;
#pragma unroll
                            for (int k = 0; k < 3; ++k)
                                ok = ok & (!need[k] || ((v[2 * k].x == gen) & (v[2 * k].z == gen) & (v[2 * k + 1].x == gen) & (v[2 * k + 1].z == gen)));
                            if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                            if (lds_ld(l_abort) != 0u) break;
                            ++spins;
                            if ((spins & 1023u) == 0u) {      // something is slow or broken: look at the error word, give up past the limit
                                const unsigned err = __hip_atomic_load(a.ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (err != 0u || spins >= a.spin_limit) {
                                    if (lane == 0) {
                                        if (err == 0u) atomicOr(a.ctl + 2, 0x80000000u | ((unsigned)i << 8) | (unsigned)w);
                                        lds_st(l_abort, 1u);
                                    }
                                    break;
                                }
                            }
                            for (int z = 0; z < a.poll_sleep; ++z) __builtin_amdgcn_s_sleep(1);
                        }
#pragma unroll
                        for (int k = 0; k < 3; ++k) { xw[k][0] = v[2 * k].y; xw[k][1] = v[2 * k].w; xw[k][2] = v[2 * k + 1].y; xw[k][3] = v[2 * k + 1].w; }
                    } else {
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const u32x4q v = *(const TMAC_GLOBAL u32x4q*)(gp[k]);
                            xw[k][0] = v.x; xw[k][1] = v.y; xw[k][2] = v.z; xw[k][3] = v.w;
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 3; ++k) { xw[k][0] = 0u; xw[k][1] = 0u; xw[k][2] = 0u; xw[k][3] = 0u; }
                }
                if (stamp && first == c0) CSTAMP(i, 1);
                if (first == c0) after_fetch();
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (blk[k] >= 0) {
                        if (SM == 2) {
                            const int p = blk[k] * 64 + lane;
                            if (p < P) tab[(p & 3) * tstride + (p >> 2)] = make_uint4(xw[k][0], xw[k][1], xw[k][2], xw[k][3]);
                            if (blk[k] * 64 < P) c_block_stats(l_us, P, blk[k], lane, xw[k]);
                            TMAC_CBAR();
                            if (lane == 0) lds_st(bfl + blk[k], base + 1u);
                        } else {
                            c_build_block<SM>(tab, tstride, l_ls, l_lb, P, blk[k], lane, xw[k], 0.f, 0.f);
                            TMAC_CBAR();
                            if (lane == 0) lds_st(bfl + blk[k], base + 3u);
                        }
                    }
                }
            }
            if (!all_batches) break;
        }
        if (SM == 2) {
            // every block parked (by whoever claimed it): the row's scale is known; then build what is parked among this wave's candidates
            for (int b = 0; b < nreal; ++b) LDS_WAIT(lds_ld(bfl + b) >= base + 1u, i);
            const float gs = c_row_scale(l_us, nreal, lane);
            const float gt = (gs != 0.0f) ? rcp_exact(gs) : 0.0f;
            int tried = 0;
            for (int c = c0; c < nblk; c += cs) {
                if (claim(c, base + 2u, base + 2u)) {        // parked (base + 1; every block is at least that by now) -> claimed for the build
                    const int p = c * 64 + lane;
                    uint32_t xw[4] = {0u, 0u, 0u, 0u};
                    if (p < P) { const uint4 v = tab[(p & 3) * tstride + (p >> 2)]; xw[0] = v.x; xw[1] = v.y; xw[2] = v.z; xw[3] = v.w; }
                    c_build_block<SM>(tab, tstride, l_ls, l_lb, P, c, lane, xw, gs, gt);
                    TMAC_CBAR();
                    if (lane == 0) lds_st(bfl + c, base + 3u);
                }
                if (!all_batches && ++tried >= 3) break;
            }
        }
    };

    if (w < NLW) {
        // =========================================== lookup waves ===========================================
        qv4i_t bsel;
        {
            const int jrel = (lane & 15) - 4 * (lane >> 4);
            const uint32_t be = (jrel >= 0 && jrel < 4) ? (0x01u << (8 * jrel)) : 0u, bo = (jrel >= 0 && jrel < 4) ? (0xfeu << (8 * jrel)) : 0u;   // +1 | -2
            bsel = (qv4i_t){(int)be, (int)bo, (int)be, (int)bo};
        }
        uint32_t k3 = 0x03020100u;
        asm volatile("" : "+v"(k3));
        uint32_t lane16 = (uint32_t)lane * 16u;
        asm volatile("" : "+v"(lane16));
        CFrag<BITS> ring[RING];
        unsigned g = 0;                                                  // workgroup iterations closed so far (all calls)
        for (int i = 0; i < a.nops; ++i) {
            const cop_ptr d = ops + i;
            if (w == 0) CSTAMP(i, 0);
            const int tstride = uni(d->tstride), nu = uni(d->nu), nst = uni(d->nst), GP = uni(d->GP);
            const uint4* tab = lds + (size_t)(i & 1) * a.buf_u4;          // [4][tstride]
            const float* l_ls = reinterpret_cast<const float*>(tab + 4 * tstride);   // [GP] ls / 2 (groups past K: 0)
            const float* l_lb = l_ls + GP;                                 // [GP] lb / 2
            const unsigned* bfl = l_sync + SY_BLK + (i & 1) * CHAIN_MAX_BLK;
            const unsigned base = 4u * ((unsigned)i + 1u);
            // role of this wave: quad qs (of ipi) of every iteration of its workgroup; steps h, h + wpq, ... of each
            const int wpq = uni(d->wpq), inv = uni(d->wpq_inv);
            const int qs = (w * inv) >> 16, h = w - qs * wpq;             // w / wpq, w % wpq for w < 16
            const Part pt = part_of(d);
            const int my_q = qs < pt.nB ? pt.niter : (qs < pt.nA ? pt.rem : 0);      // iterations (a prefix) in which this wave has a quad
            const int nsteps = h < nst ? ((nst - h + wpq - 1) * inv) >> 16 : 0;      // steps h, h + wpq, ... < nst
            const int nsg = uni(d->nsg), gsh = uni(d->gs_shift);
            // Work items of this wave in this op: my_q quads x nsteps steps, walked by an issue cursor and a lookup cursor.
            // What depends on the quad alone -- its matrix (compares against the op's cumulative quad counts), the buffer
            // resource of that matrix, the byte offset of the quad's weights, its first scale group -- is resolved when the
            // cursor enters the quad, not per item: scalar instructions are issued by ONE unit per CU.
            const int qe0 = uni(d->q_end[0]), qe1 = uni(d->q_end[1]), qe2 = uni(d->q_end[2]);
            const int n_items = my_q * nsteps;
            __amdgpu_buffer_rsrc_t q_rs = __builtin_amdgcn_make_buffer_rsrc(static_cast<uint4*>(nullptr), (short)0, 0, 0x00020000);
            const TMAC_GLOBAL char* q_sc = nullptr;
            int q_woff = 0, q_res = -1;
            int i_it = 0, i_st = h, issued = 0;
            auto issue_next = [&](CFrag<BITS>& f) __attribute__((always_inline)) {
                if (issued < n_items) {
                    if (q_res != i_it) {
                        const int gqi = part_lo(pt, i_it) + qs;
                        const int mi = (gqi >= qe0 ? 1 : 0) + (gqi >= qe1 ? 1 : 0) + (gqi >= qe2 ? 1 : 0);
                        const int lq = gqi - (gqi >= qe2 ? qe2 : (gqi >= qe1 ? qe1 : (gqi >= qe0 ? qe0 : 0)));
                        q_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(uni(d->m[mi].W)), (short)0, 0x7fffffff, 0x00020000);
                        q_sc = as_global(uni(reinterpret_cast<const char*>(d->m[mi].SC))) + (size_t)lq * (size_t)(nsg * 4 * (ZP ? 2 : 1) * (SCF16 ? 2 : 4));
                        q_woff = lq * nst * (BITS * 1024);
                        q_res = i_it;
                    }
                    c_issue<BITS, ZP, SCF16, SM>(f, q_rs, q_woff, q_sc, nsg, gsh, nu, i_st, lane, lane16);
                    ++issued;
                    i_st += wpq;
                    if (i_st >= nst) { i_st = h; ++i_it; }
                }
            };
            // an item of step s needs the four blocks of that step only (tables of units 64 s .. 64 s + 63, their act groups' scales)
            unsigned ready = 0;                                           // bit s: step s of this call's LUT has been seen complete
            auto ensure = [&](int s) __attribute__((always_inline)) {
                if (!((ready >> s) & 1u)) {
                    LDS_WAIT(lds_flags4_eq(bfl + 4 * s, base + 3u), i);
                    ready |= 1u << s;
                }
            };

            // ---- 1. the first weight fragment(s) go out at once; the rest of the ring once the LUT exists -- the polls share this
            // CU's in-order vector-memory queue with the weight loads, and a poll behind a full ring (96 KB per CU) returns
            // microseconds late (profiles/r02_chain_prefetch_ab.txt A, r03_chain_knobs.txt).  Then this wave's share of the LUT:
            // the blocks it can still claim (none when the builders ran ahead). ----
            const int isf = a.issue_first >= 0 ? a.issue_first : (uni(d->in_gran) >> 8);
#pragma unroll
            for (int k = 0; k < RING; ++k)
                if (k < isf) issue_next(ring[k]);
            {
                // a wave that fetches a block sends the rest of its ring right behind the successful poll, in front of the table build
                // (the weights stream in meanwhile, as in k_gemv_quad); a wave without a block keeps the queue clear for the others' polls
                unsigned long long polls = 0;
                bool rest_out = false;
                lut_duty(i, d, w, NLW, false, polls, w == 0, [&]() __attribute__((always_inline)) {
#pragma unroll
                    for (int k = 0; k < RING; ++k)
                        if (k >= isf) issue_next(ring[k]);
                    rest_out = true;
                });
                if (n_items > 0) ensure(h);
                if (w == 0) CSTAMP(i, 2);
                if (!rest_out) {
#pragma unroll
                    for (int k = 0; k < RING; ++k)
                        if (k >= isf) issue_next(ring[k]);
                }
            }

            // ---- 2. lookups.  Items are consumed in issue order, ring slot = item ordinal mod RING (static register roles: the loop is
            // unrolled over the ring).  A workgroup iteration (ipi consecutive quads) is closed by finish(): every lookup wave leaves
            // its quad's partial sums in the iteration's reduction buffer, raises its arrival flag and goes on; the publisher wave
            // combines the wpq partials of each quad in wave order and publishes.  Every lookup wave closes all niter iterations. ----
            int c_it = 0;
            int32_t iacc[BITS];
#pragma unroll
            for (int pl = 0; pl < BITS; ++pl) iacc[pl] = 0;
            auto finish = [&](bool have, float cacc) __attribute__((always_inline)) {
                const unsigned slot = g & (unsigned)(NPAR - 1);
                if (g >= (unsigned)NPAR) LDS_WAIT(lds_ld(l_sync + SY_CONS + slot) + (unsigned)NPAR > g, i);     // the publisher has emptied this buffer
                float* red = l_red + slot * (NLW * 4 * CHAIN_RED);
                if (SM == 2) {
                    // exact integer totals of the lane's row (lane & 3): lanes of a DPP row by rotation, rows by two cross-row moves
                    int32_t* redi = reinterpret_cast<int32_t*>(red);
#pragma unroll
                    for (int pl = 0; pl < BITS; ++pl) {
                        uint32_t v = have ? (uint32_t)iacc[pl] : 0u;
                        v += qdpp_u<0x124>(v);
                        v += qdpp_u<0x128>(v);
                        v += (uint32_t)__shfl_xor((int)v, 16, 64);
                        v += (uint32_t)__shfl_xor((int)v, 32, 64);
                        if (lane < 4) redi[(w * 4 + lane) * CHAIN_RED + pl] = (int32_t)v;
                        iacc[pl] = 0;
                    }
                } else {
                    float acc = 0.f;
                    if (have) {
                        acc = cacc;
                        acc = __fadd_rn(acc, qdpp_f<0x124>(acc));     // lanes with the same row: rotate by 4, 8 within the DPP row
                        acc = __fadd_rn(acc, qdpp_f<0x128>(acc));
                        acc = __fadd_rn(acc, __shfl_xor(acc, 16, 64));
                        acc = __fadd_rn(acc, __shfl_xor(acc, 32, 64));
                    }
                    if (lane < 4) red[(w * 4 + lane) * CHAIN_RED] = acc;
                }
                TMAC_CBAR();
                if (lane == 0) lds_st(l_sync + SY_ARR + slot * 16 + w, g + 1u);
                ++g;
                ++c_it;
            };

            int c_st = h;
            float cacc = 0.f;
            if (n_items > 0) {
                int left = n_items;
                while (left > 0) {
#pragma unroll
                    for (int k = 0; k < RING; ++k) {
                        ensure(c_st);
                        c_compute<BITS, ZP, SCF16, SM>(ring[k], tab, tstride, l_ls, l_lb, c_st, lane, bsel, k3, cacc, iacc);
                        issue_next(ring[k]);               // refill this slot with the item RING places ahead, if there is one
                        c_st += wpq;
                        if (c_st >= nst) {
                            finish(true, cacc);
                            cacc = 0.f;
                            c_st = h;
                        }
                        if (--left == 0) break;
                    }
                }
            }
            if (w == 0) CSTAMP(i, 4);
            if (w == NLW - 1) CSTAMP(i, 9);
            while (c_it < pt.niter) finish(false, 0.f);
            if (w == 0) CSTAMP(i, 10);
        }
    } else if (w < NLW + NBW) {
        // =========================================== builder waves ===========================================
        const int j = w - NLW;
        __builtin_amdgcn_s_setprio(TMAC_CHAIN_AUX_PRIO);           // few waves, always on somebody's critical path: ahead of the (spinning or streaming) lookup waves
        for (int i = 0; i < a.nops; ++i) {
            const cop_ptr d = ops + i;
            if (j == 0) CSTAMP(i, 6);
            // A builder may not claim before the vector is due (the workgroup has published what the call reads): until then the blocks
            // belong to the lookup waves that arrive -- they fetch into registers and build in parallel, no detour -- and from then on to
            // whoever comes first.  Vectors that were published long ago (or are external) are due at once: the builders run ahead.
            if (i >= 2) LDS_WAIT(lds_ld(l_sync + SY_PDONE) + 1u >= (unsigned)i, i);
            if (uni(d->in_gran) & 1) { const unsigned sg = (unsigned)uni(d->src_g); if (sg) LDS_WAIT(lds_ld(l_sync + SY_PUBG) >= sg, i); }
            unsigned long long polls = 0;
            lut_duty(i, d, j, NBW, true, polls, false, [&]() {});
            if (j == 0) { CSTAMP(i, 3); CSTAMPV(i, 7, polls); }
        }
    } else {
        // =========================================== publisher wave ===========================================
        unsigned g = 0;
        __builtin_amdgcn_s_setprio(TMAC_CHAIN_AUX_PRIO);
        for (int i = 0; i < a.nops; ++i) {
            const cop_ptr d = ops + i;
            const int tstride = uni(d->tstride);
            const float* l_us = reinterpret_cast<const float*>(lds + (size_t)(i & 1) * a.buf_u4 + 4 * tstride);
            const int wpq = uni(d->wpq), ipi = uni(d->ipi), nm = uni(d->nmat);
            const Part pt = part_of(d);
            // everything the epilogue needs from the descriptor, read while the lookups still run (a field is an LDS round trip)
            int qe_[4], mw_[4];
            unsigned long long c_[4], gr_[4], sc_[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                qe_[mi] = 0x7fffffff; mw_[mi] = 4; c_[mi] = 0ull; gr_[mi] = 0ull; sc_[mi] = 0ull;
                if (mi < nm) {
                    qe_[mi] = uni(d->m[mi].q_end); mw_[mi] = uni(d->m[mi].Mw);
                    c_[mi] = reinterpret_cast<unsigned long long>(uni(d->m[mi].C));
                    gr_[mi] = reinterpret_cast<unsigned long long>(uni(d->m[mi].GR));
                    sc_[mi] = reinterpret_cast<unsigned long long>(uni(d->m[mi].SC));
                }
            }
            const int mg = SM == 2 ? uni(d->m_groups) : 1;
            const int qs = lane >> 2, row = lane & 3;
            float us_ls = 0.f, us_lb = 0.f;           // SM 2: lut_scales, lut_biases of the row
            for (int it = 0; it < pt.niter; ++it) {
                const unsigned slot = g & (unsigned)(NPAR - 1);
                const unsigned* arr = l_sync + SY_ARR + slot * 16;
                const float* red = l_red + slot * (NLW * 4 * CHAIN_RED);
                const int g0 = part_lo(pt, it);                        // first quad of this workgroup iteration
                const int n_it = it < pt.rem ? pt.nA : pt.nB;
                const int gql = g0 + qs;
                const bool mine = qs < ipi && qs < n_it;               // the 4 lanes of a quad decide together
                // the quad's matrix: compares against the op's cumulative quad counts
                const int mi_l = (gql >= qe_[0] ? 1 : 0) + (gql >= qe_[1] ? 1 : 0) + (gql >= qe_[2] ? 1 : 0);
                const int lq = gql - (gql >= qe_[2] ? qe_[2] : (gql >= qe_[1] ? qe_[1] : (gql >= qe_[0] ? qe_[0] : 0)));
                const unsigned long long cbase = mi_l == 0 ? c_[0] : mi_l == 1 ? c_[1] : mi_l == 2 ? c_[2] : c_[3];
                const unsigned long long gbase = mi_l == 0 ? gr_[0] : mi_l == 1 ? gr_[1] : mi_l == 2 ? gr_[2] : gr_[3];
                float uscale = 0.f;
                if (SM == 2) {
                    // the row's unified scale (qgemm.py:170-174): fetched now, needed after the arrivals
                    const int Mwm = mi_l == 0 ? mw_[0] : mi_l == 1 ? mw_[1] : mi_l == 2 ? mw_[2] : mw_[3];
                    const unsigned long long sb = mi_l == 0 ? sc_[0] : mi_l == 1 ? sc_[1] : mi_l == 2 ? sc_[2] : sc_[3];
                    const int gq = (mg == 1 || !mine) ? 0 : (4 * lq + row) / (Mwm / mg);
                    if (mine) uscale = SCF16 ? __half2float(__ushort_as_half(reinterpret_cast<const TMAC_GLOBAL unsigned short*>(sb)[gq]))
                                             : reinterpret_cast<const TMAC_GLOBAL float*>(sb)[gq];
                    if (it == 0) {
                        // lut_scales from the blocks' maxima; lut_biases: ONE fp32 chain over the K/32 chunk sums in order (lut_ctor.cc:157,218;
                        // 270 dependent adds at K = 8640) -- walked here, by the wave that needs it, while the lookups run
                        const unsigned* bfl = l_sync + SY_BLK + (i & 1) * CHAIN_MAX_BLK;
                        const unsigned base = 4u * ((unsigned)i + 1u);
                        const int nreal = (uni(d->K) / 8 + 63) >> 6;
                        for (int b = 0; b < nreal; ++b) LDS_WAIT(lds_ld(bfl + b) >= base + 1u, i);
                        us_ls = c_row_scale(l_us, nreal, lane);
                        float biases = 0.0f;
                        const float4* cs = reinterpret_cast<const float4*>(l_us + CHAIN_US_FLOATS);      // 16-byte reads, unrolled: only the adds are serial
                        const int nc = uni(d->nu);                                                         // K / 32 chunks
                        int c = 0;
#pragma unroll 4
                        for (; c + 4 <= nc; c += 4) {
                            const float4 v4 = cs[c >> 2];
                            biases = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(biases, v4.x), v4.y), v4.z), v4.w);
                        }
                        for (; c < nc; ++c) biases = __fadd_rn(biases, l_us[CHAIN_US_FLOATS + c]);
                        us_lb = biases;
                    }
                }
                LDS_WAIT(lds_flags4_eq(arr, g + 1u) && lds_flags4_eq(arr + 4, g + 1u) && (NLW <= 8 || lds_flags4_eq(arr + 8, g + 1u)) &&
                         (NLW <= 12 || lds_flags4_eq(arr + 12, g + 1u)), i);
                if (it == pt.niter - 1) CSTAMP(i, 8);
                float t = 0.f;
                if (mine) {
                    if (SM == 2) {
                        int32_t cb[BITS];
#pragma unroll
                        for (int pl = 0; pl < BITS; ++pl) cb[pl] = 0;
                        const int32_t* redi = reinterpret_cast<const int32_t*>(red);
                        for (int ww = 0; ww < wpq; ++ww)
#pragma unroll
                            for (int pl = 0; pl < BITS; ++pl) cb[pl] += redi[((qs * wpq + ww) * 4 + row) * CHAIN_RED + pl];
                        // scale-final (qgemm.py:170-174,192-206), as k_gemv_quad's epilogue: C = ((sum_p float(cb_p) alpha_p) ls + lb / 2) Scale
                        float acc = 0.f;
#pragma unroll
                        for (int pl = 0; pl < BITS; ++pl) {
                            const float tp = __fmul_rn((float)cb[pl], q_alpha(pl));
                            acc = (pl == 0) ? tp : __fadd_rn(acc, tp);
                        }
                        const float v = __fadd_rn(__fmul_rn(acc, us_ls), __fmul_rn(us_lb, 0.5f));
                        t = __fmul_rn(v, uscale);
                    } else {
                        t = red[((qs * wpq) * 4 + row) * CHAIN_RED];
                        for (int ww = 1; ww < wpq; ++ww) t = __fadd_rn(t, red[((qs * wpq + ww) * 4 + row) * CHAIN_RED]);
                    }
                }
                TMAC_CBAR();
                // The fp16 output is the fp32 result rounded once more (as k_gemv_quad stores it and as the oracle is
                // compared): without the barrier the compiler fuses the last multiplication with the conversion
                // (v_fma_mixlo_f16: ONE rounding of the exact product), which differs on exact fp16 ties.
                asm volatile("" : "+v"(t));
                // granule = {generation, fp16 row | fp16 next row << 16}, 8 bytes, write-through: even rows store
                const uint32_t hb = (uint32_t)__half_as_ushort(__float2half_rn(t));
                const uint32_t nb = qdpp_u<0xB1>(hb);                    // quad_perm [1,0,3,2]: the neighbour's value
                if (mine) {
                    const size_t oi = (size_t)(4 * lq + row);
                    if (a.out_f16) reinterpret_cast<TMAC_GLOBAL unsigned short*>(cbase)[oi] = (unsigned short)hb;
                    else reinterpret_cast<TMAC_GLOBAL float*>(cbase)[oi] = t;
                    if (gbase && !(row & 1)) {
                        const unsigned long long gv = ((unsigned long long)(hb | (nb << 16)) << 32) | gen;
                        TMAC_GLOBAL unsigned long long* dst = reinterpret_cast<TMAC_GLOBAL unsigned long long*>(gbase + par_off) + 2 * (size_t)lq + (row >> 1);
                        __hip_atomic_store(dst, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        // row-sharded chains: the same granule into the hand-off arena of every other rank (identical layout on
                        // every rank: the peer's address is its arena base plus this address' offset), system scope over xGMI
                        const unsigned long long off = reinterpret_cast<unsigned long long>(dst) - a.arena_base;
                        for (int pe = 0; pe < a.npeer; ++pe)
                            __hip_atomic_store(reinterpret_cast<TMAC_GLOBAL unsigned long long*>(a.peer_base[pe] + off), gv,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
                TMAC_CBAR();
                if (lane == 0) { lds_st(l_sync + SY_CONS + slot, g + 1u); lds_st(l_sync + SY_PUBG, g + 1u); }
                ++g;
            }
            CSTAMP(i, 5);
            // every lookup wave has closed the call's last iteration (its last read of this call's LUT came before) and the epilogue's
            // reads of the buffer are done: the builders may overwrite it (call i + 2)
            if (lane == 0) lds_st(l_sync + SY_PDONE, (unsigned)i + 1u);
        }
    }
#undef CSTAMP
#undef CSTAMPV
#undef LDS_WAIT

    // the last workgroup out advances the generation: every workgroup has read it by then (the workgroup's waves exit one by one:
    // the count is per wave)
    if (lane == 0) {
        const unsigned total = (unsigned)gx * (unsigned)(CHAIN_FT / 64);
        const unsigned old = __hip_atomic_fetch_add(a.ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == total - 1u) {
            __hip_atomic_store(a.ctl + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.ctl, gen + 1u == 0u ? 1u : gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

#if !defined(TMAC_CHAIN_BITS)
#error "compile with -DTMAC_CHAIN_BITS=1..4 (one translation unit per weight width keeps the build parallel)"
#endif
constexpr int CB = TMAC_CHAIN_BITS;

template <bool ZP, bool SCF16, int SM>
static hipError_t chain_launch_one(const ChainArgs& a, int grid, size_t lds_bytes, hipStream_t st, int* resident) {
    auto* kern = &k_decode_chain<CB, ZP, SCF16, SM>;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    if (resident)       // query only: workgroups of this kernel one CU can hold (the chain needs >= 1 on EVERY CU at once)
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(resident, reinterpret_cast<const void*>(kern), CHAIN_FT, lds_bytes);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(CHAIN_FT), lds_bytes, st, a);
    return hipGetLastError();
}

#define TMAC_CHAIN_LAUNCHER(NAME)                                                                                               \
    hipError_t NAME(const ChainArgs& a, bool zp, bool sc_f16, int sm, int grid, size_t lds_bytes, hipStream_t st, int* resident) { \
        if (a.nops < 1 || grid < 1) return hipErrorInvalidValue;                                                                  \
        if (sm == 2) return sc_f16 ? chain_launch_one<false, true, 2>(a, grid, lds_bytes, st, resident)                          \
                                   : chain_launch_one<false, false, 2>(a, grid, lds_bytes, st, resident);                        \
        if (sm != 0) return hipErrorInvalidValue;                                                                                 \
        if (zp) return sc_f16 ? chain_launch_one<true, true, 0>(a, grid, lds_bytes, st, resident)                                \
                              : chain_launch_one<true, false, 0>(a, grid, lds_bytes, st, resident);                              \
        return sc_f16 ? chain_launch_one<false, true, 0>(a, grid, lds_bytes, st, resident)                                       \
                      : chain_launch_one<false, false, 0>(a, grid, lds_bytes, st, resident);                                     \
    }
#if TMAC_CHAIN_BITS == 1
TMAC_CHAIN_LAUNCHER(launch_decode_chain_b1)
#elif TMAC_CHAIN_BITS == 2
TMAC_CHAIN_LAUNCHER(launch_decode_chain_b2)
#elif TMAC_CHAIN_BITS == 3
TMAC_CHAIN_LAUNCHER(launch_decode_chain_b3)
#else
TMAC_CHAIN_LAUNCHER(launch_decode_chain_b4)
#endif

}  // namespace tmac
