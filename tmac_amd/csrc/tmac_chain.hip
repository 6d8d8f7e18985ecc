// tmac_chain.hip — k_decode_chain: a recorded sequence of fused decode GEMV groups (N = 1) as ONE persistent launch.
//
// Why: a decoded token of llama-2-7B is 128 dependent launches of 4.7-25 MB each; as separate launches every one of them
// pays the dependent-dispatch gap, fetches its activations, builds its LUT and only then starts its weight stream, so HBM
// idles through most of a launch (0.78 ms per token against 0.45 ms for launches that merely read the
// bytes).  Here one workgroup per CU walks the whole op list:
//   * no dependent dispatch, no ramp-up and drain per op, activations read once (the poll that detects them IS the load);
//     an op's first weight fragment is in flight before its polls, the rest streams in during the LUT build;
//   * the hand-off is in-kernel: an op's outputs are published as self-tagged 8-byte granules {generation, 2 x fp16}
//     with write-through (sc1) stores; the consumers' LUT build reads exactly those granules with sc1 loads and spins
//     until every tag carries this launch's generation (data is the flag: no counter, no fence, no drain of the weight
//     loads in flight).  cdna_hip_programming.md Guideline 16, recipe R2;
//   * no dispatch gap, no grid barrier: a workgroup only ever waits for data it needs.
// Arithmetic is that of k_gemv_quad (tmac_quad.hip) — same LUT build (lut_ctor.cc:120-215), same lookup + MFMA adder
// (tbl.cc:445-462), same per-act-group scale chain (tbl.cc:479-526), same lane/wave decomposition for a given number of
// waves per quad — so results are bit-identical to the per-launch path with 768-thread workgroups.
// Scope: 1- to 4-bit weights, fp16 activations; SM = 0: per-group scales with act groups of 64 and scale groups >= 128 (the
// GPTQ-style path, tbl.cc:323-532); SM = 2: unified scale(s), one act group per row, exact int32 totals and the scale-final
// epilogue (BitNet: tbl.cc:536-630, qgemm.py:170-174,192-206).
// Deadlock freedom: workgroups process ops in order and producers never wait for consumers, so by induction over the op
// index everything completes provided all workgroups are resident; the grid is one workgroup per CU and the kernel's
// register / LDS footprint admits exactly one.  Every spin is bounded (ChainArgs::spin_limit) and reports through ctl[2].
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "tmac_chain_core.h"

namespace tmac {

// The hand-off granules of up to three LUT pairs (two consecutive row quads = 8 activations each), all rounds of a thread
// in flight together; loads and their wait in one statement (cdna_hip_programming.md 5.7, form (i)).  Agent scope (sc1)
// bypasses this CU's L1 and sees what other XCDs wrote through; chains that span several GPUs poll at system scope (sc0 sc1):
// the granules then arrive over xGMI from the peers' producers.
#define TMAC_POLL_FNS(SUFFIX, SC)                                                                                              \
    __device__ __forceinline__ void c_poll1##SUFFIX(const uint4* p0, u32x4q (&v)[8]) {                                          \
        asm volatile("global_load_dwordx4 %0, %2, off " SC "\n\t"                                                             \
                     "global_load_dwordx4 %1, %2, off offset:16 " SC "\n\t"                                                   \
                     "s_waitcnt vmcnt(0)"                                                                                       \
                     : "=&v"(v[0]), "=&v"(v[1]) : "v"(p0) : "memory");                                                          \
    }                                                                                                                           \
    __device__ __forceinline__ void c_poll2##SUFFIX(const uint4* p0, const uint4* p1, u32x4q (&v)[8]) {                         \
        asm volatile("global_load_dwordx4 %0, %4, off " SC "\n\t"                                                             \
                     "global_load_dwordx4 %1, %4, off offset:16 " SC "\n\t"                                                   \
                     "global_load_dwordx4 %2, %5, off " SC "\n\t"                                                             \
                     "global_load_dwordx4 %3, %5, off offset:16 " SC "\n\t"                                                   \
                     "s_waitcnt vmcnt(0)"                                                                                       \
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p0), "v"(p1) : "memory");                       \
    }                                                                                                                           \
    __device__ __forceinline__ void c_poll3##SUFFIX(const uint4* p0, const uint4* p1, const uint4* p2, u32x4q (&v)[8]) {        \
        asm volatile("global_load_dwordx4 %0, %6, off " SC "\n\t"                                                             \
                     "global_load_dwordx4 %1, %6, off offset:16 " SC "\n\t"                                                   \
                     "global_load_dwordx4 %2, %7, off " SC "\n\t"                                                             \
                     "global_load_dwordx4 %3, %7, off offset:16 " SC "\n\t"                                                   \
                     "global_load_dwordx4 %4, %8, off " SC "\n\t"                                                             \
                     "global_load_dwordx4 %5, %8, off offset:16 " SC "\n\t"                                                   \
                     "s_waitcnt vmcnt(0)"                                                                                       \
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]) : "v"(p0), "v"(p1), "v"(p2) : "memory"); \
    }
#define TMAC_POLL4_FN(SUFFIX, SC)                                                                                               \
    __device__ __forceinline__ void c_poll4##SUFFIX(const uint4* p0, const uint4* p1, const uint4* p2, const uint4* p3, u32x4q (&v)[8]) { \
        asm volatile("global_load_dwordx4 %0, %8, off " SC "\n\t"                                                             \
                     "global_load_dwordx4 %1, %8, off offset:16 " SC "\n\t"                                                   \
                     "global_load_dwordx4 %2, %9, off " SC "\n\t"                                                             \
                     "global_load_dwordx4 %3, %9, off offset:16 " SC "\n\t"                                                   \
                     "global_load_dwordx4 %4, %10, off " SC "\n\t"                                                            \
                     "global_load_dwordx4 %5, %10, off offset:16 " SC "\n\t"                                                  \
                     "global_load_dwordx4 %6, %11, off " SC "\n\t"                                                            \
                     "global_load_dwordx4 %7, %11, off offset:16 " SC "\n\t"                                                  \
                     "s_waitcnt vmcnt(0)"                                                                                       \
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])   \
                     : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");                                                          \
    }
TMAC_POLL_FNS(, "sc1")
TMAC_POLL_FNS(_sys, "sc0 sc1")
TMAC_POLL4_FN(, "sc1")
TMAC_POLL4_FN(_sys, "sc0 sc1")
#undef TMAC_POLL4_FN
#undef TMAC_POLL_FNS
// the same for plain activations (in memory since before the launch)
__device__ __forceinline__ void c_ext3(const uint4* p0, const uint4* p1, const uint4* p2, u32x4q (&v)[8]) {
    asm volatile("global_load_dwordx4 %0, %3, off\n\t"
                 "global_load_dwordx4 %1, %4, off\n\t"
                 "global_load_dwordx4 %2, %5, off\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]) : "v"(p0), "v"(p1), "v"(p2) : "memory");
}

// XF: the chain holds vector transforms (tmac_hip_chain_xform) -- a kernel of its own, so that chains without them keep their registers.
// TAP: the parity tap (tmac_hip_chain_set_tap), an instance of its own next to XF (end of round 6: inside the XF instance its conditional
// stores -- per item and act group -- cost the decoder pattern, which needs XF and sets no tap, 2.8 %: 1.060 -> 1.031 ms per token)
template <int BITS, bool ZP, bool SCF16, int SM, bool XF, bool TAP = false>
__global__ __launch_bounds__(CHAIN_FT) void k_decode_chain(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    constexpr int FT = CHAIN_FT, NWV = CHAIN_NWV;
    constexpr int RING = (BITS <= 2) ? 4 : 2;       // fragments per ring; two rings (current op / next op)
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int bx = blockIdx.x, gx = gridDim.x;
    const unsigned gx_inv = gx == 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)gx - 1u) / (unsigned)gx);      // ceil(2^32 / gx); one workgroup: p mod 1 = 0 = bx below (2^32 does not fit)
    const unsigned gen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(a.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const unsigned long long par_off = (gen & 1u) ? a.arena_half : 0ull;    // this launch's half of the hand-off arena (tmac_chain.h)
    float* l_red = reinterpret_cast<float*>(lds + 2 * (size_t)a.buf_u4);    // [2][NWV][4][CHAIN_RED] partials of split quads (SM 2: per bit-plane)
    // all op descriptors into LDS (16 uint4 each): a field is then a ds_read away instead of a scalar-cache miss
    uint4* l_ops = lds + 2 * (size_t)a.buf_u4 + (2 * NWV * 4 * CHAIN_RED * sizeof(float)) / 16;
    {
        const uint4* gsrc = reinterpret_cast<const uint4*>(a.ops);
        for (int idx = tid; idx < a.nops * (int)(sizeof(ChainOp) / 16); idx += FT) l_ops[idx] = gsrc[idx];
        __syncthreads();
    }
    const cop_ptr ops = reinterpret_cast<cop_ptr>(l_ops);
    float* l_xf = reinterpret_cast<float*>(l_ops + (size_t)a.nops * (sizeof(ChainOp) / 16));     // [2][16] partial sums of a transform's mean square (by op parity)
    // vectors of the transforms, 16-byte pieces in the order lanes of a wave hold them (piece h of the pair (round r, tpair) at
    // ((2 r + h) * PS + tpair) * 16 bytes, PS = the op's pairs per round rounded up to whole waves: chain_xf_region_floats)
    float* l_carry = l_xf + 32;                                                                     // [a.carry_floats] the t a NORM transform keeps
    float* l_tmp = l_carry + a.carry_floats;                                                        // [a.tmp_floats] this op's t (NORM, not kept) or x (GLU)
    float* l_gam = l_tmp + a.tmp_floats;                                                            // [a.gam_floats] this op's norm weights
    float* l_ext = l_gam + a.gam_floats;                                                            // [a.ext_floats] this op's activations when they are fp32 in memory
    bool aborted = false;                                                   // a hand-off timed out somewhere: stop waiting

    // stamps: s_memrealtime (100 MHz, one clock for the whole device; s_memtime counts per XCD with unrelated offsets)
// (stamps 0-4 and 7 by the wave that polls and builds LUT block 0 -- the LAST wave, see tpair below -- 5 and 6 by wave 0, the publisher)
// (TMAC_CHAIN_STAMPS, tmac_chain.h: the hooks exist in profiling builds only -- seven conditional stores per call cost the token 3 %)
#define CSTAMPV(i, k, v) do { if (TMAC_CHAIN_STAMPS && a.stamps && tid == (((k) == 5 || (k) == 6) ? 0 : FT - 64)) a.stamps[((size_t)(i) * gx + bx) * 8 + (k)] = (v); } while (0)
#define CSTAMP(i, k) CSTAMPV(i, k, __builtin_amdgcn_s_memrealtime())

    CSel<BITS> sel;
    c_selectors<BITS, SM>(sel, lane);
    uint32_t k3 = 0x03020100u;
    asm volatile("" : "+v"(k3));
    uint32_t lane16 = (uint32_t)lane * 16u;
    asm volatile("" : "+v"(lane16));
    uint32_t lk4 = 4u * (uint32_t)(2 * (lane & 12) + 2 * (lane >> 4));      // the lane's two act groups inside a step's 32 (c_compute)
    asm volatile("" : "+v"(lk4));

    // LUT pair of this thread in round r: r * FT + tpair.  The waves take the 64-pair blocks of a round in REVERSE order (lanes in
    // order: the build's DPP sums depend on it): wave 0 -- which combines and publishes every workgroup iteration and therefore enters
    // an op last -- gets the block that exists least often (K = 4096: blocks 0..7 go to waves 11..4), so the barrier behind the LUT
    // build waits for waves that started polling on time, not for the publisher (A/B knob TMAC_CHAIN_PAIR_ORDER=0: waves in order)
#ifndef TMAC_CHAIN_PAIR_ORDER
#define TMAC_CHAIN_PAIR_ORDER 1
#endif
    const int tpair = TMAC_CHAIN_PAIR_ORDER ? (NWV - 1 - w) * 64 + lane : tid;
    // ... and the lookup roles likewise: LOGICAL wave wl = NWV - 1 - w takes quad slot wl / wpq, steps wl % wpq, ... -- when a workgroup
    // iteration has fewer (quad, step) slots than waves, the idle ones include wave 0.  The partial sums are filed under the logical
    // index, so the combination order (k_gemv_quad's) does not change.
#ifndef TMAC_CHAIN_BIAS_WAVE
#define TMAC_CHAIN_BIAS_WAVE 0       // unified scale: the wave that walks the lut_biases chain behind the LUT barrier (A/B knob)
#endif
#ifndef TMAC_CHAIN_ROLE_ORDER
#define TMAC_CHAIN_ROLE_ORDER 1
#endif
    const int wl = TMAC_CHAIN_ROLE_ORDER ? NWV - 1 - w : w;
    // per-op role of this wave: quad qs (of ipi) of every iteration of its workgroup; steps h, h + wpq, ... of each
    // Row quads are dealt to the workgroups as contiguous, balanced ranges: workgroup b owns q_per (+ 1 for the first q_extra
    // workgroups) consecutive quads -- every CU streams its share of every op (800 quads over 256 CUs: 3 or 4 each, not 6 on
    // 134 of them), and a workgroup's outputs stay together in the hand-off image (one or two stores per granule line).
    struct Role { int q_lo, cnt, qs, ipi, h, wpq, nst, my_iter, nquads, nsteps; };
    auto role_of = [&](cop_ptr d) __attribute__((always_inline)) {
        Role r;
        r.wpq = uni(d->wpq);
        const int ipi = uni(d->ipi), inv = uni(d->wpq_inv);
        const int qs = (wl * inv) >> 16;                                 // wl / wpq for wl < 12
        r.h = wl - qs * r.wpq;
        r.qs = qs; r.ipi = ipi;
        r.nst = uni(d->nst);
        const int qper = uni(d->q_per), qex = uni(d->q_extra);
        const int unit = (XF && uni(d->epi)) ? 2 : 1;                     // GLU in the producer: ranges in pairs (gate quad, up quad)
        r.q_lo = unit * (bx * qper + min(bx, qex));
        r.cnt = unit * (qper + (bx < qex ? 1 : 0));
        const int iinv = uni(d->ipi_inv);                                // x / ipi = (x * ipi_inv) >> 16 for the small x here
        r.my_iter = ((r.cnt + ipi - 1) * iinv) >> 16;                    // iterations the WORKGROUP runs (barriers)
        r.nquads = qs < r.cnt ? (((r.cnt - 1 - qs) * iinv) >> 16) + 1 : 0;   // quads this wave works on: q_lo + qs + it * ipi < q_lo + cnt
        r.nsteps = r.h < r.nst ? ((r.nst - r.h + r.wpq - 1) * inv) >> 16 : 0;   // steps h, h + wpq, ... < nst
        return r;
    };

    CFrag<BITS> ring[RING];
    int parity = 0;
    for (int i = 0; i < a.nops; ++i) {
        const cop_ptr d = ops + i;
        CSTAMP(i, 0);
        const int tstride = uni(d->tstride), nu = uni(d->nu), nst = uni(d->nst), G = uni(d->G), GP = uni(d->GP);
        uint4* tab = lds + (size_t)(i & 1) * a.buf_u4;                // [steps][4][65] (tmac_chain_core.h: the step-major layout)
#ifndef TMAC_CHAIN_IMG2
#define TMAC_CHAIN_IMG2 1          // A/B: 0 = the [4][tstride] table layout of rounds 1-5
#endif
#ifndef TMAC_CHAIN_ISSUE_FULL
#define TMAC_CHAIN_ISSUE_FULL 0    // A/B: 1 = items below K through c_issue_full (scale addressing on the scalar unit, no exec mask): -5 VALU per item and 3 % SLOWER in the
                                   // dependent chain (0.672 -> 0.695 ms: the branch in front of the loads moves the compiler's waits; profiles/r06_chain_experiments.txt) -- it pays in k_gemv_stream only
#endif
        float* l_ls = reinterpret_cast<float*>(tab + (TMAC_CHAIN_IMG2 ? IMG2_STEP * nst : 4 * tstride));   // [GP] ls / 2 (groups past K: 0)
        float* l_lb = l_ls + GP;                                     // [GP] lb / 2
        // SM 2 uses the same floats as: [0] lut_scales, [1] lut_biases, [2 .. 2+NWV) per-wave maxima, [16 .. 48) the unified
        // scales of the op's matrices (m_groups <= CHAIN_US_MAX_GROUPS each), [CHAIN_US_FLOATS .. + K/32) the chunk sums of the bias chain
        float* l_us = l_ls;
        const int P = uni(d->K) / 8;                                      // LUT pairs: tables 2p, 2p+1 from activations 8p .. 8p+7
        const Role ro = role_of(d);
        const int wpq = ro.wpq, h = ro.h;
        const int nsg = uni(d->nsg), gsh = uni(d->gs_shift), ipi = uni(d->ipi), nm = uni(d->nmat);
        // Work items of this wave in this op: nquads quads x nsteps steps, walked by an issue cursor and a lookup cursor.
        // What depends on the quad alone -- its matrix (compares against the op's cumulative quad counts), the buffer
        // resource of that matrix, the byte offset of the quad's weights, its first scale group -- is resolved when the
        // cursor enters the quad, not per item: scalar instructions are issued by ONE unit per CU, and 12 waves x ~60 of
        // them per fragment were 0.4 us per fragment issued (profiles/r02_chain_prefetch_ab.txt C).
        const int qe0 = uni(d->q_end[0]), qe1 = uni(d->q_end[1]), qe2 = uni(d->q_end[2]);
        const bool epi = XF && uni(d->epi) != 0;                              // matrices 0 / 1 dealt in pairs, silu(gate) * up published (tmac_chain.h)
        const int n_items = ro.nquads * ro.nsteps;
        __amdgpu_buffer_rsrc_t q_rs = __builtin_amdgcn_make_buffer_rsrc(static_cast<uint4*>(nullptr), (short)0, 0, 0x00020000);
        const TMAC_GLOBAL char* q_sc = nullptr;
        const TMAC_GLOBAL char* q_scm = nullptr;
        int q_woff = 0, q_res = -1, q_mi = -1;
        // (c_issue_full) log2 of a scale group's bytes per row quad; the lane's constant part of a scale offset; is the last 64-unit step ragged?
        constexpr int SCSH = (ZP ? 1 : 0) + (SCF16 ? 3 : 4);
        const uint32_t v_sc0 = (((uint32_t)(4 * (lane & 12) + 4 * (lane >> 4)) >> gsh) << SCSH) + (uint32_t)(lane & 3) * (uint32_t)((ZP ? 2 : 1) * (SCF16 ? 2 : 4));
        const bool rag = (nu & 63) != 0;
        int i_it = 0, i_st = h, issued = 0;
        auto issue_next = [&](CFrag<BITS>& f) __attribute__((always_inline)) {
            if (issued < n_items) {
                if (q_res != i_it) {
                    const int gqi = ro.q_lo + ro.qs + i_it * ro.ipi;
                    const int mi = epi ? (gqi & 1) : (gqi >= qe0 ? 1 : 0) + (gqi >= qe1 ? 1 : 0) + (gqi >= qe2 ? 1 : 0);
                    const int lq = epi ? (gqi >> 1) : gqi - (gqi >= qe2 ? qe2 : (gqi >= qe1 ? qe1 : (gqi >= qe0 ? qe0 : 0)));
                    if (mi != q_mi) {       // the matrix' pointers: two LDS reads + five readfirstlane -- per matrix a wave enters, not per quad
                        q_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(uni(d->m[mi].W)), (short)0, 0x7fffffff, 0x00020000);
                        q_scm = as_global(uni(reinterpret_cast<const char*>(d->m[mi].SC)));
                        q_mi = mi;
                    }
                    q_sc = q_scm + (size_t)lq * (size_t)(nsg * 4 * (ZP ? 2 : 1) * (SCF16 ? 2 : 4));
                    q_woff = lq * nst * (BITS * 1024);
                    q_res = i_it;
                }
                if (!TMAC_CHAIN_ISSUE_FULL || (rag && i_st == nst - 1)) c_issue<BITS, ZP, SCF16, SM>(f, q_rs, q_woff, q_sc, nsg, gsh, nu, i_st, lane, lane16);
                else c_issue_full<BITS, ZP, SCF16, SM>(f, q_rs, q_woff + i_st * (BITS * 1024), q_sc + ((size_t)((i_st << 6) >> gsh) << SCSH), v_sc0, lane16);
                ++issued;
                i_st += wpq;
                if (i_st >= nst) { i_st = h; ++i_it; }
            }
        };

        // ---- 1. this op's activations.  The CU's vector-memory queue is empty here (the previous op's lookups consumed
        // everything it had in flight).  Measured the other way round -- next op's weights prefetched behind the current op's
        // lookups -- every publish and every poll sat behind 20-100 KB of queued weight loads per CU: 3-4 us per hand-off
        // (profiles/r02_chain_prefetch_ab.txt).  ONE weight fragment per wave goes out in front of the polls, the rest of the
        // ring once the activations have arrived: the poll returns after a fabric round trip plus 24 KB per CU, and the
        // compiler-visible wait behind it does not hold the LUT build until ALL of the op's weights have landed -- the build
        // overlaps the stream.  The whole ring in front measured 5-7 % slower even for the ops whose stream outlasts the
        // hand-off (profiles/r03_chain_knobs.txt A, B); the count sits in the op descriptor, ChainArgs::issue_first overrides it. ----
        const int isf = a.issue_first >= 0 ? a.issue_first : (uni(d->in_gran) >> 8);
#pragma unroll
        for (int k = 0; k < RING; ++k)
            if (k < isf) issue_next(ring[k]);
        if (SM == 2) {             // the unified scales of this op's matrices: a handful of floats, parked in LDS for the epilogue
            const int mg = uni(d->m_groups);
            if (tid < nm * mg) {
                const int mi = tid / mg, g = tid - mi * mg;
                const void* scp = d->m[mi].SC;
                l_us[16 + mi * CHAIN_US_MAX_GROUPS + g] =
                    SCF16 ? __half2float(__ushort_as_half(as_global(reinterpret_cast<const unsigned short*>(scp))[g]))
                          : as_global(reinterpret_cast<const float*>(scp))[g];
            }
        }
        // hand-off image | fp32 vector in memory (known to the XF instance only: chains without it keep their code) | fp16 vector in memory
        const bool gran = (uni(d->in_gran) & 1) != 0, ext32 = XF && (uni(d->in_gran) & 2) != 0;
        const int nr = (P + FT - 1) / FT;                            // rounds of FT pairs (<= NRMAX, checked on the host)
        constexpr int NRMAX = 3;
        uint32_t xw[NRMAX][4];
        // a GLU transform (x = silu(in) * in2, tmac_hip_chain_xform) reads a second vector of the same kind -- both handed over or both in
        // memory -- in at most two rounds; its granules are polled together with the first vector's (one fabric round trip for both)
        const int xk = XF ? uni(d->xf_kind) : 0;
        const bool glu = XF && xk == 2;
        // NORM: the residual and the weight vector are in memory since before the launch -- their loads (16 bytes each) go out in front of
        // the polls and cross the fabric while the hand-off is awaited.  xa: the residual (NORM) or the second vector's fp16 pairs (GLU):
        // one set of registers for the two transforms
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4 xa[XF ? 2 : 1];                    // GLU: the second vector's fp16 pairs of the two rounds
        const int xPS = (min(P, FT) + 63) & ~63;
        const int xfl = XF ? uni(d->xf_flags) : 0;
        float* xt = nullptr;                     // NORM: the t = in + residual of this op (the kept vector's place, or scratch); GLU: x (scratch)
        auto xslot = [&](float* region, int r, int h) __attribute__((always_inline)) {
            return reinterpret_cast<f32x4*>(region) + ((2 * r + h) * xPS + tpair);
        };
        if (XF && xk == 1) {
            // The residual and the norm weights are in memory since before the launch: they go global -> LDS without touching a register
            // (buffer_load ... lds, 16 bytes per lane: lane l of a wave lands at M0 + 16 l), issued in front of the polls and complete
            // when the polls are (loads return in order).  Held in registers across the polls and the reduction -- 32 VGPRs -- the kernel
            // spilled, and a spill behind the weight ring waits for the weights.
            xt = (xfl & 4) ? l_carry : l_tmp;
            typedef __attribute__((address_space(3))) void* lds_vp;
            const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void*)(uni(d->res)), (short)0, 0x7fffffff, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)(uni(d->gamma)), (short)0, 0x7fffffff, 0x00020000);
            const bool ext_res = uni(d->res) != nullptr && !(xfl & 2), has_g = uni(d->gamma) != nullptr;
            const uint32_t b_t = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)xt;
            const uint32_t b_g = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)l_gam;
            const int wfirst = __builtin_amdgcn_readfirstlane(tpair - lane);
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (r < nr && wfirst < xPS) {        // (whole waves past the round's pairs have no slots: PS is the round's pairs in whole waves)
                    const int pc = min(r * FT + tpair, P - 1);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t lo = (uint32_t)(((2 * r + h) * xPS + wfirst) * 16);
                        if (ext_res) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_r, (lds_vp)(uintptr_t)(b_t + lo), 16, (2 * pc + h) * 16, 0, 0, 0);
                        if (has_g) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_vp)(uintptr_t)(b_g + lo), 16, (2 * pc + h) * 16, 0, 0, 0);
                    }
                }
        }
        unsigned long long polls = 0;
        {
            // pair of round r: p = r * FT + tid; past the end the address is clamped and the result ignored
            const uint4* in4 = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(uni(d->in)) + (gran ? par_off : 0ull));
            const int p0 = min(tpair, P - 1), p1 = min(FT + tpair, P - 1), p2 = min(2 * FT + tpair, P - 1);
            const bool n0 = tpair < P, n1 = FT + tpair < P, n2 = 2 * FT + tpair < P;
            u32x4q v[8];
            if (gran && glu) {
                const uint4* in5 = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(uni(d->in2)) + par_off);
                const uint4 *g0 = in4 + 2 * (size_t)p0, *g1 = in4 + 2 * (size_t)p1, *h0 = in5 + 2 * (size_t)p0, *h1 = in5 + 2 * (size_t)p1;
                unsigned spins = 0;
                for (int z = 0; z < a.poll_delay; ++z) __builtin_amdgcn_s_sleep(1);
                for (;;) {
                    ++polls;
                    bool ok;
                    const bool sys = a.npeer > 0 || a.poll_mode == 1;
                    if (nr == 1) {
                        if (sys) c_poll2_sys(g0, h0, v); else c_poll2(g0, h0, v);
                        ok = !n0 || ((v[0].x == gen) & (v[0].z == gen) & (v[1].x == gen) & (v[1].z == gen) &
                                     (v[2].x == gen) & (v[2].z == gen) & (v[3].x == gen) & (v[3].z == gen));
                    } else {
                        if (sys) c_poll4_sys(g0, g1, h0, h1, v); else c_poll4(g0, g1, h0, h1, v);
                        ok = (!n0 || ((v[0].x == gen) & (v[0].z == gen) & (v[1].x == gen) & (v[1].z == gen) &
                                      (v[4].x == gen) & (v[4].z == gen) & (v[5].x == gen) & (v[5].z == gen))) &
                             (!n1 || ((v[2].x == gen) & (v[2].z == gen) & (v[3].x == gen) & (v[3].z == gen) &
                                      (v[6].x == gen) & (v[6].z == gen) & (v[7].x == gen) & (v[7].z == gen)));
                    }
                    if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                    if (aborted) break;
                    ++spins;
                    if ((spins & 1023u) == 0u) {
                        const unsigned err = __hip_atomic_load(a.ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (err != 0u || spins >= a.spin_limit) {
                            if (err == 0u && lane == 0) atomicOr(a.ctl + 2, 0x80000000u | ((unsigned)i << 8) | (unsigned)w);
                            aborted = true;
                        }
                    }
                    for (int z = 0; z < a.poll_sleep; ++z) __builtin_amdgcn_s_sleep(1);
                }
                auto w2f = [](uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3) __attribute__((always_inline)) {
                    return (f32x4){__uint_as_float(a0), __uint_as_float(a1), __uint_as_float(a2), __uint_as_float(a3)};
                };
                if (nr == 1) {
                    xw[0][0] = v[0].y; xw[0][1] = v[0].w; xw[0][2] = v[1].y; xw[0][3] = v[1].w;
                    xa[0] = w2f(v[2].y, v[2].w, v[3].y, v[3].w);
                } else {
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        xw[r][0] = v[2 * r].y; xw[r][1] = v[2 * r].w; xw[r][2] = v[2 * r + 1].y; xw[r][3] = v[2 * r + 1].w;
                        xa[r] = w2f(v[4 + 2 * r].y, v[4 + 2 * r].w, v[5 + 2 * r].y, v[5 + 2 * r].w);
                    }
                }
            } else if (gran) {
                const uint4 *g0 = in4 + 2 * (size_t)p0, *g1 = in4 + 2 * (size_t)p1, *g2 = in4 + 2 * (size_t)p2;
                unsigned spins = 0;
                for (int z = 0; z < a.poll_delay; ++z) __builtin_amdgcn_s_sleep(1);      // A/B knob: wait before the first poll
                for (;;) {
                    ++polls;
                    bool ok;
                    const bool sys = a.npeer > 0 || a.poll_mode == 1;      // granules written by other GPUs: system scope
                    if (nr == 1) {
                        if (sys) c_poll1_sys(g0, v); else c_poll1(g0, v);
                        ok = !n0 || ((v[0].x == gen) & (v[0].z == gen) & (v[1].x == gen) & (v[1].z == gen));
                    } else if (nr == 2) {
                        if (sys) c_poll2_sys(g0, g1, v); else c_poll2(g0, g1, v);
                        ok = (!n0 || ((v[0].x == gen) & (v[0].z == gen) & (v[1].x == gen) & (v[1].z == gen))) &
                             (!n1 || ((v[2].x == gen) & (v[2].z == gen) & (v[3].x == gen) & (v[3].z == gen)));
                    } else {
                        if (sys) c_poll3_sys(g0, g1, g2, v); else c_poll3(g0, g1, g2, v);
                        ok = (!n0 || ((v[0].x == gen) & (v[0].z == gen) & (v[1].x == gen) & (v[1].z == gen))) &
                             (!n1 || ((v[2].x == gen) & (v[2].z == gen) & (v[3].x == gen) & (v[3].z == gen))) &
                             (!n2 || ((v[4].x == gen) & (v[4].z == gen) & (v[5].x == gen) & (v[5].z == gen)));
                    }
                    if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                    if (aborted) break;
                    ++spins;
                    if ((spins & 1023u) == 0u) {      // something is slow or broken: look at the error word, give up past the limit
                        const unsigned err = __hip_atomic_load(a.ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (err != 0u || spins >= a.spin_limit) {
                            if (err == 0u && lane == 0) atomicOr(a.ctl + 2, 0x80000000u | ((unsigned)i << 8) | (unsigned)w);
                            aborted = true;
                        }
                    }
                    if (a.poll_grid > 0) {
                        // A/B: all workgroups re-poll at the same instants of the device clock (one phase for the whole chip instead of 256)
                        const unsigned rem = (unsigned)a.poll_grid - ((unsigned)__builtin_amdgcn_s_memrealtime() & ((unsigned)a.poll_grid - 1u));
                        const int nz = (int)((rem * 10u) >> 5);                 // ticks of 10 ns -> s_sleep 1 rounds of ~32 ns
                        for (int z = 0; z < nz; ++z) __builtin_amdgcn_s_sleep(1);
                    } else
                    for (int z = 0; z < a.poll_sleep; ++z) __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int r = 0; r < NRMAX; ++r) { xw[r][0] = v[2 * r].y; xw[r][1] = v[2 * r].w; xw[r][2] = v[2 * r + 1].y; xw[r][3] = v[2 * r + 1].w; }
            } else if (ext32) {
                // fp32 activations in memory (a caller whose graph is fp32, e.g. ggml): global -> LDS without registers, 32 bytes per pair,
                // read back as the fp32 values the LUT is built from (no fp16 in between: what the per-launch path does with them)
                typedef __attribute__((address_space(3))) void* lds_vp;
                const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc((void*)(uni(d->in)), (short)0, 0x7fffffff, 0x00020000);
                const uint32_t b_e = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)l_ext;
                const int wfirst = __builtin_amdgcn_readfirstlane(tpair - lane);
#pragma unroll
                for (int r = 0; r < NRMAX; ++r)
                    if (r < nr && wfirst < xPS) {
                        const int pc = min(r * FT + tpair, P - 1);
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_i, (lds_vp)(uintptr_t)(b_e + (uint32_t)(((2 * r + h) * xPS + wfirst) * 16)), 16,
                                                                     (2 * pc + h) * 16, 0, 0, 0);
                    }
#pragma unroll
                for (int r = 0; r < NRMAX; ++r) xw[r][0] = xw[r][1] = xw[r][2] = xw[r][3] = 0u;
            } else {
                c_ext3(in4 + p0, in4 + p1, in4 + p2, v);
#pragma unroll
                for (int r = 0; r < NRMAX; ++r) { xw[r][0] = v[r].x; xw[r][1] = v[r].y; xw[r][2] = v[r].z; xw[r][3] = v[r].w; }
                if (glu) {
                    const uint4* in5 = uni(reinterpret_cast<const uint4*>(d->in2));
                    c_ext3(in5 + p0, in5 + p1, in5 + p1, v);
#pragma unroll
                    for (int r = 0; r < 2; ++r) xa[r] = (f32x4){__uint_as_float(v[r].x), __uint_as_float(v[r].y), __uint_as_float(v[r].z), __uint_as_float(v[r].w)};
                }
            }
        }
        CSTAMP(i, 1);
        CSTAMPV(i, 7, polls);
        // Nothing is in flight here (the polls carry their own waits, invisible to the compiler).  Saying so with a wait
        // the compiler SEES resets its scoreboard: otherwise every register that was a load destination anywhere in the
        // op loop counts as possibly pending, and VALU writes to such registers (LUT build temporaries, store operands)
        // get conservative s_waitcnt vmcnt(n) in front of them -- waits for this op's weights in the middle of the LUT build.
        __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0), expcnt / lgkmcnt untouched
        // ---- 2. this wave's first RING (quad, step) items: the weights stream in during the LUT build ----
        // A wave stalls 0.3-2 us in front of these loads while the CU's memory queue is full (stamps 1 -> 3: the op's stream, not instruction
        // issue), and its tables wait behind that stall.  ISSUE_SPLIT = n: only n fragments go out here, the rest of the ring behind the
        // wave's part of the LUT build.  Measured (profiles/r04b_chain_issue_split.txt, same box): with 3- / 4-bit weights (a ring of two
        // fragments) and with unified scales (two passes over the vector before the tables) n = 0 is 2.5 % faster -- the fragment in front
        // of the polls keeps the memory side busy through the build; 1- / 2-bit per-group chains are fastest with the whole ring here.
#ifndef TMAC_CHAIN_ISSUE_SPLIT
#define TMAC_CHAIN_ISSUE_SPLIT (-1)
#endif
        constexpr int ISSUE_SPLIT = TMAC_CHAIN_ISSUE_SPLIT >= 0 ? TMAC_CHAIN_ISSUE_SPLIT : ((BITS >= 3 || SM == 2) ? 0 : RING);
#pragma unroll
        for (int k = 0; k < RING; ++k)
            if (k >= isf && k < isf + ISSUE_SPLIT) issue_next(ring[k]);

        CSTAMP(i, 3);

        // ---- 3. LUT into LDS (lut_ctor.cc:120-215, as in k_gemv_quad) ----
        // the activations as fp32: the fp16 vector as it is, or -- tmac_hip_chain_xform -- a vector transform of it, computed here where
        // every workgroup holds the whole vector anyway (a decoder's residual add + RMSNorm in front of q/k/v and gate/up, its
        // silu(gate) * up in front of the down projection): the calls of a layer chain up without a kernel in between
        float xrs = 1.0f;                        // NORM: 1 / rms
        // A transformed vector waits in LDS for the table build (each lane reads back what it wrote: no barrier of its own): held in
        // registers -- two rounds x 8 fp32 -- the kernel spilled, with a vmcnt(0) in front of the spill.
        if constexpr (XF) {
        if (xk == 2) {
            // GLU: x = silu(in) * in2, fp32
            xt = l_tmp;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int p = r * FT + tpair;
                if (r < nr && p < P) {
                    f32x4 x0, x1;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const __half2 ha = *reinterpret_cast<const __half2*>(&xw[r][q]);
                        const uint32_t w2 = __float_as_uint(xa[r][q]);
                        const __half2 hh = *reinterpret_cast<const __half2*>(&w2);
                        const float u0 = __low2float(hh), u1 = __high2float(hh);
                        const float a0 = __low2float(ha), a1 = __high2float(ha);
                        // silu(a) = a / (1 + exp(-a)): hardware exp2 and reciprocal (1 ulp each; the transform is specified to a tolerance)
                        const float y0 = __fmul_rn(__fmul_rn(a0, __builtin_amdgcn_rcpf(__fadd_rn(1.0f, __expf(-a0)))), u0);
                        const float y1 = __fmul_rn(__fmul_rn(a1, __builtin_amdgcn_rcpf(__fadd_rn(1.0f, __expf(-a1)))), u1);
                        if (q < 2) { x0[2 * q] = y0; x0[2 * q + 1] = y1; } else { x1[2 * q - 4] = y0; x1[2 * q - 3] = y1; }
                    }
                    *xslot(xt, r, 0) = x0; *xslot(xt, r, 1) = x1;
                }
            }
        } else if (xk == 1) {
            // NORM: t = in + residual (memory fp32, or the t an earlier NORM of this launch kept in LDS); x = t * rsqrt(mean(t^2) + eps) *
            // gamma (or x = t without gamma); t optionally kept for a later op and / or written to memory (the residual stream that
            // outlives the launch; each workgroup writes a stripe of it).  Up to two rounds of pairs (K <= 12288).
            TMAC_GLOBAL f32x4* rout4 = (TMAC_GLOBAL f32x4*)(uni(d->res_out));
            const bool has_res = uni(d->res) != nullptr || (xfl & 2);
            float* rsrc = (xfl & 2) ? l_carry : xt;
            float ss = 0.f;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int p = r * FT + tpair;
                if (r < nr && p < P) {
                    f32x4 t0 = (f32x4){0.f, 0.f, 0.f, 0.f}, t1 = t0;
                    if (has_res) { t0 = *xslot(rsrc, r, 0); t1 = *xslot(rsrc, r, 1); }
                    f32x4 i0 = (f32x4){0.f, 0.f, 0.f, 0.f}, i1 = i0;
                    if (ext32) { i0 = *xslot(l_ext, r, 0); i1 = *xslot(l_ext, r, 1); }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const __half2 hh = *reinterpret_cast<const __half2*>(&xw[r][q]);
                        const float v0 = ext32 ? (q < 2 ? i0[2 * q] : i1[2 * q - 4]) : __low2float(hh);
                        const float v1 = ext32 ? (q < 2 ? i0[2 * q + 1] : i1[2 * q - 3]) : __high2float(hh);
                        const float u0 = __fadd_rn(v0, q < 2 ? t0[2 * q] : t1[2 * q - 4]);
                        const float u1 = __fadd_rn(v1, q < 2 ? t0[2 * q + 1] : t1[2 * q - 3]);
                        if (q < 2) { t0[2 * q] = u0; t0[2 * q + 1] = u1; } else { t1[2 * q - 4] = u0; t1[2 * q - 3] = u1; }
                        ss = __fmaf_rn(u0, u0, ss);
                        ss = __fmaf_rn(u1, u1, ss);
                    }
                    *xslot(xt, r, 0) = t0; *xslot(xt, r, 1) = t1;
                    // one writer per pair, whatever the grid: pair p belongs to workgroup p mod gx (p / gx through the 32-bit reciprocal:
                    // exact for p < 2^16, the grid sizes of any device or partition)
                    if (rout4 && (gx == 1 || p - (int)__umulhi((unsigned)p, gx_inv) * gx == bx)) {
                        rout4[2 * (size_t)p] = t0;
                        rout4[2 * (size_t)p + 1] = t1;
                    }
                }
            }
            if (uni(d->gamma) != nullptr) {
                // sum over the wave: four DPP steps inside the rows of 16 lanes, the four row sums through readlane (__shfl_xor is a
                // ds_bpermute round trip per step: six in a row cost 0.2 us here)
                ss = __fadd_rn(ss, qdpp_f<0xB1>(ss)); ss = __fadd_rn(ss, qdpp_f<0x4E>(ss));
                ss = __fadd_rn(ss, qdpp_f<0x141>(ss)); ss = __fadd_rn(ss, qdpp_f<0x140>(ss));     // row_half_mirror, row_mirror
                const int ssb = __builtin_bit_cast(int, ss);
                const float sw = __fadd_rn(__fadd_rn(__builtin_bit_cast(float, __builtin_amdgcn_readlane(ssb, 0)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(ssb, 16))),
                                           __fadd_rn(__builtin_bit_cast(float, __builtin_amdgcn_readlane(ssb, 32)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(ssb, 48))));
                float* xfb = l_xf + (i & 1) * 16;                    // two sets by op parity: no second barrier needed
                if (lane == 0) xfb[w] = sw;
                // LDS-only barrier: __syncthreads() carries a workgroup fence = s_waitcnt vmcnt(0) on gfx9, i.e. it would wait here for the
                // weight fragments issued a moment ago (a memory round trip in the open, before the tables are even started)
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                float tot = 0.f;
#pragma unroll
                for (int ww = 0; ww < NWV; ++ww) tot += xfb[ww];
                xrs = __builtin_amdgcn_rsqf(__fmaf_rn(tot, __builtin_amdgcn_rcpf((float)(8 * P)), __uint_as_float((uint32_t)uni(d->eps_bits))));
            }
        }
        }   // XF
        auto unpack = [&](int r, float (&x)[8]) __attribute__((always_inline)) {
            if (XF && xk != 0 && r < 2) {
                const f32x4 t0 = *xslot(xt, r, 0), t1 = *xslot(xt, r, 1);
                if (xk == 1 && uni(d->gamma) != nullptr) {
                    const f32x4 g0 = *xslot(l_gam, r, 0), g1 = *xslot(l_gam, r, 1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        x[e] = __fmul_rn(__fmul_rn(t0[e], xrs), g0[e]);
                        x[4 + e] = __fmul_rn(__fmul_rn(t1[e], xrs), g1[e]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { x[e] = t0[e]; x[4 + e] = t1[e]; }
                }
            } else if (ext32) {
                const f32x4 t0 = *xslot(l_ext, r, 0), t1 = *xslot(l_ext, r, 1);
#pragma unroll
                for (int e = 0; e < 4; ++e) { x[e] = t0[e]; x[4 + e] = t1[e]; }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const __half2 hh = *reinterpret_cast<const __half2*>(&xw[r][q]);
                    x[2 * q] = __low2float(hh); x[2 * q + 1] = __high2float(hh);
                }
            }
        };
        float gscale = 0.f, gtinv = 0.f;
        if (SM == 2) {
            // One act group = the whole row (qgemm.py:93-96): the scale is a maximum over K and lut_biases ONE fp32 chain over the
            // K/32 chunk sums in order (lut_ctor.cc:157,218).  Neither the chunk sums nor the chain depend on the scale: pass 1
            // forms maxima and chunk sums, one barrier, every wave builds its tables; the chain is walked behind the second barrier.
            float mx = 0.f;
#pragma unroll
            for (int r = 0; r < NRMAX; ++r) {
                const int p = r * FT + tpair;
                if (r < nr && p < P) {
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
                    if ((p & 3) == 0) l_us[CHAIN_US_FLOATS + (p >> 2)] = __fadd_rn(va, vb);
                }
            }
#ifdef TMAC_CHAIN_KO_PASS1      /* timing experiment (wrong scale): what the cross-wave maximum and its barrier cost */
            mx = q_row_allmax(mx);
            mx = q_xor_max_f(mx);
#else
            mx = q_row_allmax(mx);
            mx = q_xor_max_f(mx);
            if (lane == 0) l_us[2 + w] = mx;
            __syncthreads();
            mx = l_us[2];
#pragma unroll
            for (int ww = 1; ww < NWV; ++ww) mx = fmaxf(mx, l_us[2 + ww]);
#endif
            gscale = div127(mx);
            gtinv = (gscale != 0.0f) ? rcp_exact(gscale) : 0.0f;
            if (tid == 0) l_us[0] = gscale;
        }
#pragma unroll
        for (int r = 0; r < NRMAX; ++r) {
            const int p = r * FT + tpair;
            if (r < nr && p < P) {
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
                q_table8<true>(x[0], x[1], x[2], x[3], t_scales, lo0, hi0, La);
                q_table8<true>(x[4], x[5], x[6], x[7], t_scales, lo1, hi1, Lb);
                tab[TMAC_CHAIN_IMG2 ? img2_index(p >> 2, p & 3) : (p & 3) * tstride + (p >> 2)] = make_uint4(lo0, hi0, lo1, hi1);
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
                        l_ls[p >> 3] = __fmul_rn(0.5f, scales);
                        l_lb[p >> 3] = __fmul_rn(0.5f, __fadd_rn(__fadd_rn(0.0f, v), c1));
                    }
                }
            }
        }
        {   // zero tables / zero LUT scales for the units between K and the end of the last 64-unit step
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4)
                for (int u = nu + tid; u < nst * 64; u += FT) tab[TMAC_CHAIN_IMG2 ? img2_index(u, j4) : j4 * tstride + u] = make_uint4(0u, 0u, 0u, 0u);
            if (SM != 2)
                for (int g = G + tid; g < GP; g += FT) { l_ls[g] = 0.f; l_lb[g] = 0.f; }
        }
        if (ISSUE_SPLIT < RING) {
#pragma unroll
            for (int k = 0; k < RING; ++k)
                if (k >= isf + ISSUE_SPLIT) issue_next(ring[k]);
        }
        CSTAMP(i, 4);
        __syncthreads();
        CSTAMP(i, 2);
        if (SM == 2 && tid == (TMAC_CHAIN_PAIR_ORDER ? TMAC_CHAIN_BIAS_WAVE * 64 : FT - 64)) {
            // lut_biases: ONE fp32 chain over the K/32 chunk sums in order (lut_ctor.cc:157,218; 270 dependent adds at K = 8640).
            // Only the epilogue needs it, so it is walked here, behind the barrier that releases the lookups, by lane 0 of the
            // last wave -- the wave with the fewest pairs to build and, when quads are split or a workgroup owns fewer than 12,
            // the least (or no) lookup work; finish() has a barrier before the first reader.
            float biases = 0.0f;
            const float4* cs = reinterpret_cast<const float4*>(l_us + CHAIN_US_FLOATS);      // 16-byte reads, unrolled: only the adds are serial
            const int nc = nu;                                                                 // K / 32 chunks
            int c = 0;
#pragma unroll 4
            for (; c + 4 <= nc; c += 4) {
                const float4 v4 = cs[c >> 2];
                biases = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(biases, v4.x), v4.y), v4.z), v4.w);
            }
            for (; c < nc; ++c) biases = __fadd_rn(biases, l_us[CHAIN_US_FLOATS + c]);
            l_us[1] = biases;
        }

        // ---- 4. lookups.  Items are consumed in issue order, ring slot = item ordinal mod RING (static register roles:
        // the loop is unrolled over the ring).  A workgroup iteration (ipi consecutive quads) is closed by finish(): every
        // wave leaves its quad's partial sums in LDS, one barrier, and wave 0 combines the wpq partials of each quad (in wave
        // order, as k_gemv_quad does), stores the outputs and publishes the granules of all ipi quads with ONE store
        // instruction -- a granule line is then written by one or two stores, not by sixteen 8-byte write-throughs that each
        // invalidate the line in every polling XCD.  Every wave closes my_iter iterations, with or without work. ----
        int c_it = 0;
        int32_t iacc[BITS];
#pragma unroll
        for (int pl = 0; pl < BITS; ++pl) iacc[pl] = 0;
        auto finish = [&](bool have, float cacc) __attribute__((always_inline)) {
            float* red = l_red + parity * (NWV * 4 * CHAIN_RED);
            if (SM == 2) {
                // exact integer totals of the lane's row (lane & 3): lanes of a DPP row by rotation, rows by two cross-row moves
                int32_t* redi = reinterpret_cast<int32_t*>(red);
#pragma unroll
                for (int pl = 0; pl < BITS; ++pl) {
                    uint32_t v = have ? (uint32_t)iacc[pl] : 0u;
                    v += qdpp_u<0x124>(v);
                    v += qdpp_u<0x128>(v);
                    v = q_xor_add_u(v);
                    if (lane < 4) redi[(wl * 4 + lane) * CHAIN_RED + pl] = (int32_t)v;
                    iacc[pl] = 0;
                }
            } else {
                float acc = 0.f;
                if (have) {
                    acc = cacc;
                    acc = __fadd_rn(acc, qdpp_f<0x124>(acc));     // lanes with the same row: rotate by 4, 8 within the DPP row
                    acc = __fadd_rn(acc, qdpp_f<0x128>(acc));
                    acc = q_xor_add_f(acc);
                }
                if (lane < 4) red[(wl * 4 + lane) * CHAIN_RED] = acc;
            }
            // wave 0 resolves where its lanes' rows go BEFORE the barrier (it is waited for by every consumer of these rows: what it
            // does behind the barrier is on the chip's critical path, what it does in front of it is not): lane = (quad qs, row)
            const int p_qs = lane >> 2, p_row = lane & 3;
            const int p_g0 = ro.q_lo + c_it * ipi;                    // first quad of this workgroup iteration
            const int p_gql = p_g0 + p_qs;
            const bool p_mine = p_qs < ipi && p_gql < ro.q_lo + ro.cnt;   // the 4 lanes of a quad decide together
            int p_mi = 0, p_lq = 0, p_Mw = 4;
            unsigned long long p_c = 0ull, p_gr = 0ull;
            if (w == 0) {
                const int e0 = uni(d->q_end[0]), e1 = uni(d->q_end[1]), e2 = uni(d->q_end[2]);      // (INT_MAX from the last matrix on)
                p_mi = epi ? (p_gql & 1) : (p_gql >= e0 ? 1 : 0) + (p_gql >= e1 ? 1 : 0) + (p_gql >= e2 ? 1 : 0);
                p_lq = epi ? (p_gql >> 1) : p_gql - (p_gql >= e2 ? e2 : (p_gql >= e1 ? e1 : (p_gql >= e0 ? e0 : 0)));
                const ChainMat* mp = &d->m[p_mine ? p_mi : 0];        // the workgroup's copy in LDS: a per-lane ds_read
                p_c = reinterpret_cast<unsigned long long>(mp->C);
                p_gr = reinterpret_cast<unsigned long long>(mp->GR);
                p_Mw = mp->Mw;
            }
            __syncthreads();
            if (w == 0) {
                const int qs = p_qs, row = p_row;
                const bool mine = p_mine;
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
                        if (TAP && a.tap)
#pragma unroll
                            for (int pl = 0; pl < BITS; ++pl) a.tap[a.tap_off[i] + (size_t)(4 * p_gql + row) * BITS + pl] = cb[pl];
                        // scale-final (qgemm.py:170-174,192-206), as k_gemv_quad's epilogue: C = ((sum_p float(cb_p) alpha_p) ls + lb / 2) Scale
                        float acc = 0.f;
#pragma unroll
                        for (int pl = 0; pl < BITS; ++pl) {
                            const float tp = __fmul_rn((float)cb[pl], q_alpha(pl));
                            acc = (pl == 0) ? tp : __fadd_rn(acc, tp);
                        }
                        const float v = __fadd_rn(__fmul_rn(acc, l_us[0]), __fmul_rn(l_us[1], 0.5f));
                        const int mg = uni(d->m_groups);
                        const int g = mg == 1 ? 0 : (4 * p_lq + row) / (p_Mw / mg);
                        t = __fmul_rn(v, l_us[16 + p_mi * CHAIN_US_MAX_GROUPS + g]);
                    } else {
                        t = red[((qs * wpq) * 4 + row) * CHAIN_RED];
                        for (int ww = 1; ww < wpq; ++ww) t = __fadd_rn(t, red[((qs * wpq + ww) * 4 + row) * CHAIN_RED]);
                    }
                }
                // The fp16 output is the fp32 result rounded once more (as k_gemv_quad stores it and as the oracle is
                // compared): without the barrier the compiler fuses the last multiplication with the conversion
                // (v_fma_mixlo_f16: ONE rounding of the exact product), which differs on exact fp16 ties.
                asm volatile("" : "+v"(t));
                // granule = {generation, fp16 row | fp16 next row << 16}, 8 bytes, write-through: even rows store
                const uint32_t hb = (uint32_t)__half_as_ushort(__float2half_rn(t));
                uint32_t pb = hb;                                        // what the hand-off image carries: the fp16 output, or ...
                if (epi) {
                    // ... silu(gate) * up: the up quad of this row pair sits four lanes up (slot qs + 1, same DPP row); from the fp16
                    // values a reader would see, fp32 arithmetic, rounded to the fp16 the image holds
                    const uint32_t ub = qdpp_u<0x104>(hb);               // row_shl:4
                    const float gv = __half2float(__ushort_as_half((unsigned short)hb)), uv = __half2float(__ushort_as_half((unsigned short)ub));
                    const float xv = __fmul_rn(__fmul_rn(gv, __builtin_amdgcn_rcpf(__fadd_rn(1.0f, __expf(-gv)))), uv);
                    if (p_mi == 0) pb = (uint32_t)__half_as_ushort(__float2half_rn(xv));
                }
                const uint32_t nb = qdpp_u<0xB1>(pb);                    // quad_perm [1,0,3,2]: the neighbour's value
                if (mine) {
                    const size_t oi = (size_t)(4 * p_lq + row);
                    if (a.out_f16) reinterpret_cast<TMAC_GLOBAL unsigned short*>(p_c)[oi] = (unsigned short)hb;
                    else reinterpret_cast<TMAC_GLOBAL float*>(p_c)[oi] = t;
                    if (p_gr && !(row & 1)) {
                        const unsigned long long gv = ((unsigned long long)(pb | (nb << 16)) << 32) | gen;
                        TMAC_GLOBAL unsigned long long* dst = reinterpret_cast<TMAC_GLOBAL unsigned long long*>(p_gr + par_off) + 2 * (size_t)p_lq + (row >> 1);
                        __hip_atomic_store(dst, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        // row-sharded chains: the same granule into the hand-off arena of every other rank (identical layout on
                        // every rank: the peer's address is its arena base plus this address' offset), system scope over xGMI
                        const unsigned long long off = reinterpret_cast<unsigned long long>(dst) - a.arena_base;
                        for (int pe = 0; pe < a.npeer; ++pe)
                            __hip_atomic_store(reinterpret_cast<TMAC_GLOBAL unsigned long long*>(a.peer_base[pe] + off), gv,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            }
            parity ^= 1;
            ++c_it;
        };

        int c_st = h;
        float cacc = 0.f;
        if (n_items > 0) {
            int left = n_items;
            while (left > 0) {
#pragma unroll
                for (int k = 0; k < RING; ++k) {
                    // (the instance with the extensions is also the parity tap's: ChainArgs::tap)
                    c_compute<BITS, ZP, SCF16, SM, TAP, TMAC_CHAIN_IMG2 != 0>(ring[k], tab, tstride, l_ls, l_lb, c_st * 64, lane16, lk4, sel, k3, cacc, iacc,
                                                             (TAP && a.tap) ? a.tap + a.tap_off[i] + (size_t)(4 * (ro.q_lo + ro.qs + c_it * ipi) + (lane & 3)) * G : nullptr, G,
                                                             c_st * (16 * IMG2_STEP));
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
        CSTAMP(i, 5);
        while (c_it < ro.my_iter) finish(false, 0.f);
        if (TMAC_CHAIN_STAMPS && a.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CSTAMP(i, 6); }
    }
#undef CSTAMP
#undef CSTAMPV

    // the last workgroup out advances the generation: every workgroup has read it by then
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(a.ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)gx - 1u) {
            __hip_atomic_store(a.ctl + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.ctl, gen + 1u == 0u ? 1u : gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

#if !defined(TMAC_CHAIN_BITS)
#error "compile with -DTMAC_CHAIN_BITS=1..4 (one translation unit per weight width keeps the build parallel)"
#endif
constexpr int CB = TMAC_CHAIN_BITS;

template <bool ZP, bool SCF16, int SM, bool XF, bool TAP>
static hipError_t chain_launch_x(const ChainArgs& a, int grid, size_t lds_bytes, hipStream_t st, int* resident) {
    auto* kern = &k_decode_chain<CB, ZP, SCF16, SM, XF, TAP>;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    if (resident)       // query only: workgroups of this kernel one CU can hold (the chain needs >= 1 on EVERY CU at once)
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(resident, reinterpret_cast<const void*>(kern), CHAIN_FT, lds_bytes);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(CHAIN_FT), lds_bytes, st, a);
    return hipGetLastError();
}

template <bool ZP, bool SCF16, int SM>
static hipError_t chain_launch_one(const ChainArgs& a, int grid, size_t lds_bytes, hipStream_t st, int* resident) {
    if (a.tap) return chain_launch_x<ZP, SCF16, SM, true, true>(a, grid, lds_bytes, st, resident);      // (the tap's instance carries the extensions too)
    return a.xforms ? chain_launch_x<ZP, SCF16, SM, true, false>(a, grid, lds_bytes, st, resident) : chain_launch_x<ZP, SCF16, SM, false, false>(a, grid, lds_bytes, st, resident);
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
