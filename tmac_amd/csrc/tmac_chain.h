// tmac_chain.h — descriptors of k_decode_chain (tmac_chain.hip): a recorded sequence of fused decode GEMV groups
// (tmac_hip_qgemm_fused_dev calls with N = 1) executed by ONE persistent launch.  See the kernel for the protocol.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tmac {

#ifndef TMAC_CHAIN_FT
#define TMAC_CHAIN_FT 768
#endif
constexpr int CHAIN_FT = TMAC_CHAIN_FT;   // threads per workgroup (12 waves = 3 per SIMD, one workgroup per CU; 1024 measured 6 % slower)
constexpr int CHAIN_NWV = CHAIN_FT / 64;
constexpr int CHAIN_RED = 4;              // words per (wave, row) in the split-quad reduction buffer: one fp32 partial, or (unified scale) one int32 per bit-plane
constexpr int CHAIN_US_FLOATS = 48;       // unified-scale flavour: floats in front of the chunk sums in a LUT buffer's scale area (see k_decode_chain)
constexpr int CHAIN_US_MAX_GROUPS = 8;    // unified scales per matrix the kernel parks in LDS

struct ChainMat {
    const uint4* W;      // QUAD layout weights
    const void* SC;      // QUAD layout scales
    void* C;             // user-visible output vector [Mw] (out dtype)
    uint4* GR;           // hand-off image of this output for a later op of the chain (nullptr: nobody consumes it):
                         //   one uint4 per row quad = two 8-byte granules {tag, fp16 row 0 | fp16 row 1 << 16}, {tag, rows 2 | 3}
    int Mw;
    int q_end;           // cumulative number of row quads up to and including this matrix
};

struct ChainOp {
    ChainMat m[4];
    int q_end[4];        // m[k].q_end again, contiguous (one scalar load), INT_MAX from the last matrix on
    const void* in;      // in_gran: the hand-off image (uint4 [K/4]) written earlier in this launch; else activations [K] fp16
    int in_gran;         // bit 0: `in` is a hand-off image; bit 1: `in` is fp32 in memory ([K] floats); bits 8..: weight fragments per wave issued in front of the polls (host's choice for this op)
    int nmat;
    int K, nu, nst, tstride, G, GP, nsg, gs_shift;
    int wpq, ipi;        // waves per row quad, row quads per workgroup iteration (12 / wpq)
    int wpq_inv;         // ceil(65536 / wpq): wave / wpq = (wave * wpq_inv) >> 16
    int total_q;
    int q_per, q_extra;  // total_q = q_per * grid + q_extra: workgroup b owns the q_per + (b < q_extra) consecutive quads from b * q_per + min(b, q_extra)
    int m_groups;        // unified-scale flavour: scales per matrix (rows split into m_groups equal runs; qgemm.py:170-174), else 0
    int ipi_inv;         // ceil(65536 / ipi)
    // vector transform of the activations inside the LUT build (tmac_hip_chain_xform; xf_kind 0: none)
    const void* in2;     // GLU (xf_kind 2): x = silu(in) * in2 -- a second vector of the same kind as `in` (hand-off image / fp16 memory)
    const float* res;    // NORM (xf_kind 1): t = in + res (fp32 [K] in memory; nullptr: none, or the carry: xf_flags bit 1)
    const float* gamma;  //   x = t * rsqrt(mean(t^2) + eps) * gamma (fp32 [K]); nullptr: x = t
    float* res_out;      //   t is also written here (fp32 [K]); nullptr: not written
    int xf_kind;
    int xf_flags;        // bit 1: the residual is the t kept by an earlier NORM of this launch (LDS); bit 2: keep this op's t
    int eps_bits;        // eps as the bits of a float
    int epi;             // 1: this op's matrices 0 and 1 are gate and up of a GLU whose only reader is a later op of the chain: the row quads of
                         //    the two are dealt in pairs (workgroup ranges in units of two: q_per / q_extra count pairs; virtual quad v = matrix v & 1,
                         //    quad v >> 1) and matrix 0's hand-off image carries silu(gate) * up, computed once per row by the publishing wave
    // stream mode (tmac_stream.hip): the op's prebuilt LUT image in global memory (layout of the LDS LUT buffer) and its size in uint4, whole KB
    const void* img;
    int img_u4;
    int wg_lo;           // stream mode: the op is served by the row ranges (workgroups) wg_lo .. wg_lo + wg_cnt - 1 of the launch; q_per / q_extra / wpq / ipi
                         // are those of total_q quads dealt to wg_cnt ranges, range wg_lo + b owning what workgroup b owns above (0 in k_decode_chain)
    // sizeof == 320: the kernel keeps a copy of all descriptors in LDS (uint4 copies)
};
static_assert(sizeof(ChainOp) == 320, "ChainOp is copied to LDS in 16-byte pieces");

struct ChainArgs {
    const ChainOp* ops;            // device memory; every workgroup copies them to LDS at kernel entry (a descriptor field read through
                                   // the scalar cache misses it -- 128 x 256 B per token, once each -- and costs ~0.4 us on the critical path)
    int nops;
    unsigned* ctl;                 // [0] generation (tag of this launch; its parity selects the half of the hand-off arena), [1] workgroups finished, [2] error word
    int out_f16;
    unsigned spin_limit;           // polls of one hand-off before a wave gives up and sets ctl[2]
    int buf_u4;                    // uint4 per LDS LUT buffer (two buffers, by op parity)
    int poll_sleep;                // s_sleep 1 (64 cycles) count between two polls of a hand-off (A/B knob)
    int poll_delay;                // s_sleep 1 count before the first poll of a hand-off (A/B knob)
    int issue_first;               // A/B knob: >= 0 overrides the per-op number of weight fragments issued before the polls for the activations
    int poll_mode;                 // A/B knob: 0 polls at agent scope (sc1) | 1 at system scope (sc0 sc1; always with peers)
    int poll_grid;                 // A/B knob: re-polls wait for the next multiple of this many 10 ns ticks of the device clock (power of two; 0: off)
    // row-sharded chains over several GPUs (one process per GPU): every rank holds the hand-off images of ALL ranks' rows in one
    // arena with the same layout; a producer stores its granules into its own arena and, through IPC mappings, into every peer's
    unsigned long long arena_base;     // this rank's arena (device address): two halves of arena_half bytes, used by generation parity -- a rank that
                                       // is one launch ahead of a peer writes the other half, never the image the peer is still polling
    unsigned long long arena_half;
    unsigned long long peer_base[7];   // the other ranks' arenas as mapped into this process
    int npeer;
    int xforms;                    // some op carries a vector transform (tmac_hip_chain_xform): the kernel instance that knows them
    int carry_floats;              // LDS floats for the vector a NORM transform keeps for a later op of the launch (0: no op does)
    int tmp_floats, gam_floats;    // LDS floats for a transform's vector of the current op (NORM's t / GLU's x) and for NORM's weights
    int ext_floats;                // LDS floats for an op's activations when they come as fp32 from memory
    int32_t* tap;                  // parity tap (tmac_hip_chain_set_tap; runs the TAP instance of the kernel): the integers that enter the float part, per op at
    const unsigned long long* tap_off;   //   tap + tap_off[op]: per-group scales int32 [4 * total_q rows][K / 64] comb = sum_p 2^p PS_p; unified scales [rows][bits] totals
    unsigned long long* stamps;    // optional [nops][grid][8] of wave 0, s_memrealtime (100 MHz): 0 op entry, 1 activations complete, 2 LUT built
                                   // (barrier passed), 3 current ring landed, 5 last quad published, 6 everything in flight landed, 7 polls
};

// one translation unit per weight width (tmac_chain.hip with -DTMAC_CHAIN_BITS=b).  sm: 0 per-group scales, 2 unified scale.
// resident != nullptr: no launch -- returns how many workgroups of that kernel one CU can hold with lds_bytes of LDS.
hipError_t launch_decode_chain_b1(const ChainArgs& a, bool zp, bool sc_f16, int sm, int grid, size_t lds_bytes, hipStream_t st, int* resident);
hipError_t launch_decode_chain_b2(const ChainArgs& a, bool zp, bool sc_f16, int sm, int grid, size_t lds_bytes, hipStream_t st, int* resident);
hipError_t launch_decode_chain_b3(const ChainArgs& a, bool zp, bool sc_f16, int sm, int grid, size_t lds_bytes, hipStream_t st, int* resident);
hipError_t launch_decode_chain_b4(const ChainArgs& a, bool zp, bool sc_f16, int sm, int grid, size_t lds_bytes, hipStream_t st, int* resident);
inline hipError_t launch_decode_chain(const ChainArgs& a, int bits, bool zp, bool sc_f16, int sm, int grid, size_t lds_bytes, hipStream_t st,
                                      int* resident = nullptr) {
    switch (bits) {
        case 1: return launch_decode_chain_b1(a, zp, sc_f16, sm, grid, lds_bytes, st, resident);
        case 2: return launch_decode_chain_b2(a, zp, sc_f16, sm, grid, lds_bytes, st, resident);
        case 3: return launch_decode_chain_b3(a, zp, sc_f16, sm, grid, lds_bytes, st, resident);
        case 4: return launch_decode_chain_b4(a, zp, sc_f16, sm, grid, lds_bytes, st, resident);
        default: return hipErrorInvalidValue;
    }
}
// LDS: two LUT buffers of buf_u4 uint4 each ([4][tstride] half tables + the act groups' scales / biases, or the unified-scale
// scratch), the split-quad reduction buffer, the op descriptors, the transforms' scratch and carry
inline int chain_buf_u4(int K) {
    const int nu = K / 32, nst = (nu + 63) / 64;
    return 4 * 65 * nst + (2 * nst * 32 * 4 + 15) / 16 + CHAIN_US_FLOATS / 4;       // tables [steps][4][65] uint4 (tmac_chain_core.h: IMG2), ls / 2, lb / 2 per act group | unified-scale scratch
}
// LDS floats of one transform vector of a K-vector: 16-byte pieces, two per pair, rounds of CHAIN_FT pairs padded to whole waves
inline int chain_xf_region_floats(int K) {
    const int P = K / 8, nr = (P + CHAIN_FT - 1) / CHAIN_FT, PS = ((P < CHAIN_FT ? P : CHAIN_FT) + 63) & ~63;
    return nr * 2 * PS * 4;
}
inline size_t chain_lds_bytes(int buf_u4, int nops, int xf_floats = 0) {
    return (size_t)2 * buf_u4 * 16 + sizeof(float) * 2 * CHAIN_NWV * 4 * CHAIN_RED + sizeof(ChainOp) * (size_t)nops + sizeof(float) * (32 + (size_t)xf_floats);
}

// Profiling stamps of k_decode_chain (tmac_hip_chain_set_stamps, bench.py --stamps, tools/chain_stamps.py): compiled in by
// -DTMAC_CHAIN_STAMPS=1 only (tools/build_variant.sh st "-DTMAC_CHAIN_STAMPS=1").  Round 6: the idle hooks -- seven conditional stores per call,
// which also cost the compiler its count of the loads in flight -- were 3 % of the dependent token (0.671 -> 0.651 ms; BitNet-3B 0.547 -> 0.529).
#ifndef TMAC_CHAIN_STAMPS
#define TMAC_CHAIN_STAMPS 0
#endif

// ---- stream mode: a recording in which no op consumes another's output (tmac_stream.hip) ----
#ifndef TMAC_STREAM_NLW
#define TMAC_STREAM_NLW 12
#endif
constexpr int STREAM_NLW = TMAC_STREAM_NLW;           // lookup waves per workgroup (the roles of k_decode_chain: waves per quad must divide it)
constexpr int STREAM_FT = (STREAM_NLW + 1) * 64;      // + the loader wave
struct StreamArgs {
    const ChainOp* ops;
    int nops;
    int out_f16;
    int buf_u4;                    // uint4 per LDS LUT buffer (two buffers, by visit parity), a multiple of 64
    int nsplit;                    // workgroups per row range (1 or 2): workgroup (range, part) takes the visits part, part + nsplit, ... of its class
    // The SCHEDULE (round 6): the calls are independent, so not every row range has to visit every op.  The grid's row ranges form `ncls`
    // classes of consecutive ranges (range b is in class b * ncls / ranges); the host deals every op to an aligned block of classes -- the
    // ranges wg_lo .. wg_lo + wg_cnt - 1 share its rows -- and writes, per class, the list of ops it visits.  An op that is small for 256 CUs
    // is then served by a fraction of them with proportionally more rows each, while the other classes work on other ops: the costs of a
    // visit (two barriers, the image, the waves' op change, idle waves in a short op) are paid per (workgroup, visit), not per byte.
    int ncls;                      // classes (a power of two <= ranges; 1: every range visits every op)
    int vmax;                      // records per class in `roles`
    const int* nvis;               // [ncls] visits of each class
    const int* roles;              // [ncls][vmax][STREAM_ROLE_INTS]: what a lookup wave needs to enter a visit, worked out by the host (layout below)
    int32_t* tap;                  // parity tap, as ChainArgs::tap (the TAP instantiation of the kernel)
    const unsigned long long* tap_off;
    unsigned long long* stamps;    // profiling builds only, else ignored: [workgroups][lookup waves][8].  -DTMAC_STREAM_STAMPS=2: cycle sums 0 waiting for weights,
                                   // 1 lookups + refill, 2 partial sums, 3 closing barriers, 4 op change, 5 A barriers; 6 items, 7 first-to-last cycles.
                                   // =1 (the kernel's own resources): 4 first-to-last cycles, 5 / 6 first / last s_memtime, 7 XCC_ID << 32 | HW_ID
};
// A lookup wave enters every op twice (its issue cursor, then its lookup cursor, RING items behind).  What it needs there -- its share of the
// op's quads and steps, the op's geometry -- depends on the op, the wave and, through the one-more-quad rule only, on the workgroup: the host
// tabulates it per (op, wave) and the wave reads its record with two scalar loads instead of a dozen descriptor fields from LDS and ~80
// scalar instructions (every instruction of a wave costs >= 4 cycles of ITS time, and all waves of a workgroup change ops together).
// Per visit: 16 common ints, then 4 ints per lookup wave (logical index wl).
enum { SR_NST = 0, SR_IPI, SR_NSG, SR_GSH, SR_NU, SR_QE0, SR_QE1, SR_QE2, SR_QPER, SR_QEXTRA, SR_IT_LO, SR_IT_HI, SR_TSTRIDE, SR_OP /* index of the visited op */, SR_WPQ,
       SR_WLO /* first row range of the op's block */, SR_COMMON };
enum { SRW_NQ = 0 /* quads of the wave: workgroups with q_per | q_per + 1 quads << 16 */, SRW_NSTEPS, SRW_H, SRW_QS, SRW_INTS };
constexpr int STREAM_ROLE_INTS = SR_COMMON + SRW_INTS * STREAM_NLW;
// image / LDS buffer of one op, whole KB: the layout of k_decode_chain's LUT buffer
inline int stream_img_u4(int K) { return (chain_buf_u4(K) + 63) & ~63; }
inline size_t stream_lds_bytes(int buf_u4, int nops, bool qw = false) {
    size_t b = (size_t)2 * buf_u4 * 16 + sizeof(float) * 2 * STREAM_NLW * (qw ? 16 : 4) * CHAIN_RED + sizeof(ChainOp) * (size_t)nops;
#ifdef TMAC_STREAM_STAMPS
    b += 32 * (STREAM_NLW + 1);        // the profiling build's cycle sums
#endif
    return b;
}
hipError_t launch_lut_images(const ChainOp* d_ops, int nops, int max_nst, int sm, bool sc_f16, hipStream_t st);
hipError_t launch_gemv_stream(const StreamArgs& a, int bits, bool zp, bool sc_f16, int sm, bool qw, int grid, size_t lds_bytes, hipStream_t st);

}  // namespace tmac
