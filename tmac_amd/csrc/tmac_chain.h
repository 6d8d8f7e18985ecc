// tmac_chain.h — descriptors of k_decode_chain (tmac_chain.hip): a recorded sequence of fused decode GEMV groups
// (tmac_hip_qgemm_fused_dev calls with N = 1) executed by ONE persistent launch.  See the kernel for the protocol.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tmac {

// Wave roles of a workgroup (k_decode_chain): NLW lookup waves walk the (row quad, 64-unit step) items of every call; NBW builder
// waves fetch the activations of a call (the poll that detects a handed-over vector IS its load) and build its LUT, 64 LUT
// pairs = a quarter of a 64-unit step per block, one call ahead of the lookups; ONE publisher wave combines the partial sums of
// split quads and stores / publishes the outputs.  The roles meet through flags in LDS only -- no workgroup barrier in the call loop.
#ifndef TMAC_CHAIN_NLW
#define TMAC_CHAIN_NLW 12
#endif
#ifndef TMAC_CHAIN_NBW
#define TMAC_CHAIN_NBW 3
#endif
constexpr int CHAIN_NLW = TMAC_CHAIN_NLW;       // lookup waves: the launch configuration k_gemv_quad results are bit-identical with (64 * NLW threads)
constexpr int CHAIN_NBW = TMAC_CHAIN_NBW;       // builder waves
constexpr int CHAIN_NWV = CHAIN_NLW;            // (host side: waves that share the row quads of a workgroup iteration)
constexpr int CHAIN_LT = CHAIN_NLW * 64;        // threads of the equivalent per-launch configuration (tmac_hip_chain_threads)
constexpr int CHAIN_FT = (CHAIN_NLW + CHAIN_NBW + 1) * 64;   // threads per workgroup actually launched
static_assert(CHAIN_FT <= 1024, "a workgroup has at most 16 waves");
constexpr int CHAIN_RED = 4;              // words per (wave, row) in the split-quad reduction buffer: one fp32 partial, or (unified scale) one int32 per bit-plane
constexpr int CHAIN_NPAR = 4;             // reduction buffers in flight (lookup waves run up to this many workgroup iterations ahead of the publisher)
constexpr int CHAIN_US_FLOATS = 48;       // unified-scale flavour: floats in front of the chunk sums in a LUT buffer's scale area (see k_decode_chain)
constexpr int CHAIN_US_MAX_GROUPS = 8;    // unified scales per matrix the kernel parks in LDS
constexpr int CHAIN_MAX_K = 18432;        // 36 blocks of 64 LUT pairs
constexpr int CHAIN_MAX_BLK = 40;         // block flags per LUT buffer (4 per 64-unit step, K <= CHAIN_MAX_K: 36)
// words of the synchronisation area in LDS (see k_decode_chain)
constexpr int CHAIN_SYNC_WORDS = 2 * CHAIN_MAX_BLK + CHAIN_NPAR * 16 + 32;

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
    int in_gran;         // bit 0: `in` is a hand-off image; bits 8..: weight fragments per wave issued before the call's first LUT slice is known to be ready
    int nmat;
    int K, nu, nst, tstride, G, GP, nsg, gs_shift;
    int wpq, ipi;        // waves per row quad, row quads per workgroup iteration (NLW / wpq)
    int wpq_inv;         // ceil(65536 / wpq): wave / wpq = (wave * wpq_inv) >> 16
    int total_q;
    // Row quads are dealt out iteration-major: the call's quads are cut into `niter` consecutive super-blocks of sb_base (+ 1 for the
    // first sb_rem) quads, super-block `it` is shared by all workgroups in contiguous, balanced ranges -- per (+ 1 for the first ex
    // workgroups) quads each, <= ipi -- and is what iteration `it` of every workgroup computes.  So the rows of a call are produced in
    // ascending order chip-wide (a consumer's first LUT slices complete first) and every workgroup owns the same share of every iteration.
    int niter, sb_base, sb_rem;
    int perA, exA;       // super-blocks it < sb_rem (sb_base + 1 quads)
    int perB, exB;       // the others (sb_base quads)
    int m_groups;        // unified-scale flavour: scales per matrix (rows split into m_groups equal runs; qgemm.py:170-174), else 0
    int src_g;           // in_gran: 1 + the number of workgroup iterations (all calls of the chain, in order) up to and including the one that
                         // completes the rows this call reads -- the builders' first poll waits for their own workgroup to have published it
    int pad[3];
    // sizeof == 288: the kernel keeps a copy of all descriptors in LDS (uint4 copies)
};
static_assert(sizeof(ChainOp) == 288, "ChainOp is copied to LDS in 16-byte pieces");

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
    // row-sharded chains over several GPUs (one process per GPU): every rank holds the hand-off images of ALL ranks' rows in one
    // arena with the same layout; a producer stores its granules into its own arena and, through IPC mappings, into every peer's
    unsigned long long arena_base;     // this rank's arena (device address): two halves of arena_half bytes, used by generation parity -- a rank that
                                       // is one launch ahead of a peer writes the other half, never the image the peer is still polling
    unsigned long long arena_half;
    unsigned long long peer_base[7];   // the other ranks' arenas as mapped into this process
    int npeer;
    unsigned long long* stamps;    // optional [nops][grid][16], s_memrealtime (100 MHz): 0 lookup wave 0 enters the call, 1 builder 0 has the activations of
                                   // its first batch, 2 lookup wave 0 sees its first LUT step, 3 builder 0 finds nothing left to build, 4 lookup wave 0 done with
                                   // its last item, 5 publisher: last rows of the call published, 6 builder 0 starts on the call, 7 poll rounds of builder 0,
                                   // 8 publisher sees the last iteration's arrivals, 9 last lookup wave done, 10 lookup wave 0 leaves the call,
                                   // 11 builder 0 issues its first poll, 12 builder 0 has parked and offered its blocks
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
// scratch), the synchronisation words, the split-quad reduction buffers, the op descriptors
inline int chain_buf_u4(int K) {
    const int nu = K / 32, nst = (nu + 63) / 64;
    return 4 * (nst * 64 + 1) + (2 * nst * 32 * 4 + 15) / 16 + CHAIN_US_FLOATS / 4;
}
inline size_t chain_lds_bytes(int buf_u4, int nops) {
    return (size_t)2 * buf_u4 * 16 + sizeof(unsigned) * CHAIN_SYNC_WORDS + sizeof(float) * CHAIN_NPAR * CHAIN_NLW * 4 * CHAIN_RED +
           sizeof(ChainOp) * (size_t)nops;
}

}  // namespace tmac
