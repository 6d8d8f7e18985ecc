// tmac_chain.h — descriptors of k_decode_chain (tmac_chain.hip): a recorded sequence of fused decode GEMV groups
// (tmac_hip_qgemm_fused_dev calls with N = 1) executed by ONE persistent launch.  See the kernel for the protocol.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tmac {

constexpr int CHAIN_FT = 768;            // threads per workgroup (12 waves, one workgroup per CU)
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
    const void* in;      // in_gran: the hand-off image (uint4 [K/4]) written earlier in this launch; else activations [K] fp16
    int in_gran;
    int nmat;
    int K, nu, nst, tstride, G, GP, nsg, gs_shift;
    int wpq, ipi;        // waves per row quad, row quads per workgroup iteration (12 / wpq)
    int wpq_inv;         // ceil(65536 / wpq): wave / wpq = (wave * wpq_inv) >> 16
    int total_q;
    int it_full, it_rem; // total_q = it_full * (grid * ipi) + it_rem: iterations every workgroup runs / quads of the last, partial one
};

struct ChainArgs {
    const ChainOp* ops;            // device memory, read through the scalar cache (constant address space)
    int nops;
    unsigned* ctl;                 // [0] generation (tag of this launch), [1] workgroups finished, [2] error word
    int out_f16;
    unsigned spin_limit;           // polls of one hand-off before a wave gives up and sets ctl[2]
    int buf_u4;                    // uint4 per LDS LUT buffer (two buffers, by op parity)
    unsigned long long* stamps;    // optional [nops][grid][8] of wave 0, s_memrealtime (100 MHz): 0 op entry, 1 activations complete, 2 LUT built
                                   // (barrier passed), 3 current ring landed, 5 last quad published, 6 everything in flight landed, 7 polls
};

hipError_t launch_decode_chain(const ChainArgs& a, int bits, bool zp, bool sc_f16, int grid, size_t lds_bytes, hipStream_t st);
size_t chain_lds_bytes(int buf_u4);
int chain_buf_u4(int K);

}  // namespace tmac
