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
    int in_gran;
    int nmat;
    int K, nu, nst, tstride, G, GP, nsg, gs_shift;
    int wpq, ipi;        // waves per row quad, row quads per workgroup iteration (12 / wpq)
    int wpq_inv;         // ceil(65536 / wpq): wave / wpq = (wave * wpq_inv) >> 16
    int total_q;
    int it_full, it_rem; // total_q = it_full * (grid * ipi) + it_rem: iterations every workgroup runs / quads of the last, partial one
    int pad_[2];         // sizeof == 256: the kernel keeps a copy of all descriptors in LDS (uint4 copies)
};
static_assert(sizeof(ChainOp) == 256, "ChainOp is copied to LDS in 16-byte pieces");

struct ChainArgs {
    const ChainOp* ops;            // device memory; every workgroup copies them to LDS at kernel entry (a descriptor field read through
                                   // the scalar cache misses it -- 128 x 256 B per token, once each -- and costs ~0.4 us on the critical path)
    int nops;
    unsigned* ctl;                 // [0] generation (tag of this launch), [1] workgroups finished, [2] error word
    int out_f16;
    unsigned spin_limit;           // polls of one hand-off before a wave gives up and sets ctl[2]
    int buf_u4;                    // uint4 per LDS LUT buffer (two buffers, by op parity)
    int poll_sleep;                // s_sleep 1 (64 cycles) count between two polls of a hand-off (A/B knob)
    int poll_delay;                // s_sleep 1 count before the first poll of a hand-off (A/B knob)
    int issue_first;               // A/B knob: issue an op's weights before polling for its activations
    int poll_mode;                 // A/B knob: 0 dwordx4 sc1 | 1 dwordx4 sc0 sc1 | 2 dwordx4 nt | 3 dwordx4 plain
    unsigned long long* stamps;    // optional [nops][grid][8] of wave 0, s_memrealtime (100 MHz): 0 op entry, 1 activations complete, 2 LUT built
                                   // (barrier passed), 3 current ring landed, 5 last quad published, 6 everything in flight landed, 7 polls
};

hipError_t launch_decode_chain(const ChainArgs& a, int bits, bool zp, bool sc_f16, int grid, size_t lds_bytes, hipStream_t st);
size_t chain_lds_bytes(int buf_u4, int nops);
int chain_buf_u4(int K);

}  // namespace tmac
