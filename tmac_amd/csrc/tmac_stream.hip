// tmac_stream.hip -- k_gemv_stream: a recorded sequence of INDEPENDENT decode GEMV groups (N = 1) as one persistent launch, and
// k_lut_images, the LUT build of all of them as one launch in front of it.
//
// What it is for: SURVEY 8(d)'s headline measurement -- back-to-back GEMVs over rotating distinct weights, none reading another's
// output -- and any caller whose calls carry no data dependence (tmac_hip_chain_end recognises such a recording: no op consumes an
// earlier op's output).  The reference's own call structure is the model: the preprocessor runs once per activation vector
// (tmac_gemm_wrapper.h:170-195), qgemm_lut then only looks up (:197-228), and its GEMV byte count includes the QLUT as an INPUT
// (SURVEY 8d).  k_decode_chain cannot use that freedom (every call's table depends on the previous call's output); here
//   * k_lut_images builds every op's tables ONCE (not once per CU) into a global image that has the layout of the LDS LUT buffer;
//   * a SERVICE wave (wave 12 of each 13-wave workgroup) moves op i+1's image global -> LDS (buffer_load ... lds, no registers, no VALU)
//     while the twelve LOOKUP waves work on op i -- the tables are never on a lookup wave's path -- and combines the partial sums of split
//     quads and stores the outputs behind each closing barrier, so no lookup wave is waited for while it stores;
//   * a lookup wave's weight fragments form ONE FIFO across all ops: consume the oldest, refill the slot with the wave's next item --
//     of this op or of the ops behind it -- so the ring always holds RING items and the stream never stops at an op boundary.  The
//     loads of every item are the same unconditional sequence (c_issue_static), barriers and stores are asm: the compiler counts the
//     loads behind the one it needs and waits per slot (s_waitcnt vmcnt((RING - 1) * loads per item)) instead of draining the ring
//     (profiles/r05_ring_static.txt);
//   * TWO workgroups per CU for 1- to 3-bit weights (they need <= 64 VGPRs and <= 80 SGPRs per wave to be co-resident), each walking every other op: a lookup wave is bounded by its own serial latency per item
//     (table reads, eight dependent MFMAs, scale chain), so twice the waves on a CU -- not a deeper ring, not fewer barriers, both measured
//     flat -- is what fills the gaps: headline 3.43 -> 2.93 us per call, gate/up 5.35 -> 4.52 (profiles/r05_stream_knockouts.txt);
//   * no hand-off, no polls, no spin: nothing waits for another workgroup, so residency is not a correctness condition.
// Arithmetic per item and per row is k_decode_chain's / k_gemv_quad's (c_compute, the same wave / lane decomposition, the same
// combination order of split quads), the tables are q_table8's: outputs are bit-identical to the other N = 1 paths for the same waves
// per quad -- in the (quad x 64 units) form; the quarter-walk form (QW below, round 6) keeps the integers and changes the order of a
// row's fp32 partial sums.  Round 6 also added the SCHEDULE (tmac_chain.h, StreamArgs: calls dealt to classes of row ranges) and the
// step-major LUT image (tmac_chain_core.h, IMG2).  Scope: 1- to 4-bit QUAD-layout weights with per-group scales (act groups of 64) or unified scales (SM = 2: one act group per row,
// exact int32 totals per bit-plane, scale-final in the service wave; tables by k_lut_images_us); fp16 or fp32 activations.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "tmac_chain_core.h"

// Two translation units (build time): -DTMAC_STREAM_QW_TU=0 (default) holds the LUT-image kernels and the (quad x 64 units) form of
// k_gemv_stream, =1 its quarter-walk form.
#ifndef TMAC_STREAM_QW_TU
#define TMAC_STREAM_QW_TU 0
#endif

namespace tmac {

#if !TMAC_STREAM_QW_TU
// ---------------------------------------------------------------------------------------------
// LUT images: block (x, y) builds pairs 256 x .. 256 x + 255 (= the 64 units of step x) of op y -- the build phase of k_gemv_quad /
// k_preprocess_pairs (lut_ctor.cc:120-215,240-256) writing what k_gemv_stream's LUT buffer holds, in the STEP-MAJOR layout (round 6;
// c_compute's IMG2): [64-unit step][4][65] uint4 of signed half tables (pair p = unit p >> 2, row p & 3), then ls / 2 and lb / 2 per act
// group (32 per step each); zero tables / zero scales for the units between K and the end of the last step.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lut_images(const ChainOp* __restrict__ ops) {
    const ChainOp& d = ops[blockIdx.y];
    const int K = d.K, P = K / 8, nst = d.nst;
    if ((int)blockIdx.x >= nst) return;
    const int p = blockIdx.x * 256 + threadIdx.x;
    uint4* tab = reinterpret_cast<uint4*>(const_cast<void*>(d.img));
    float* l_sc = reinterpret_cast<float*>(tab + IMG2_STEP * nst);
    uint4* tslot = tab + img2_index(p >> 2, p & 3);                                      // unit p >> 2, row p & 3
#if TMAC_IMG2_SC
    float* scslot = l_sc + 4 * (p >> 4) + ((p >> 3) & 1);                               // act group p >> 3: its pair, its half
#else
    float* scslot = l_sc + (p >> 3);                                                    // ls[GP] then lb[GP]
#endif
    if (p >= P) {                           // P % 8 == 0: the 8 lanes of an act group are valid or padding together
        *tslot = make_uint4(0u, 0u, 0u, 0u);
        if ((p & 7) == 0) { scslot[0] = 0.f; scslot[TMAC_IMG2_SC ? 2 : 32 * nst] = 0.f; }
        return;
    }
    float x[8];
    if (d.in_gran & 2) {
        const float4* src = reinterpret_cast<const float4*>(d.in) + 2 * (size_t)p;
        const float4 a0 = src[0], a1 = src[1];
        x[0] = a0.x; x[1] = a0.y; x[2] = a0.z; x[3] = a0.w; x[4] = a1.x; x[5] = a1.y; x[6] = a1.z; x[7] = a1.w;
    } else {
        const uint4 v = reinterpret_cast<const uint4*>(d.in)[p];
        const uint32_t r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const __half2 hh = *reinterpret_cast<const __half2*>(&r[i]);
            x[2 * i] = __low2float(hh); x[2 * i + 1] = __high2float(hh);
        }
    }
    const float s0 = __fadd_rn(__fadd_rn(fabsf(x[0]), fabsf(x[1])), __fadd_rn(fabsf(x[2]), fabsf(x[3])));
    const float s1 = __fadd_rn(__fadd_rn(fabsf(x[4]), fabsf(x[5])), __fadd_rn(fabsf(x[6]), fabsf(x[7])));
    const float mx = q_half_allmax(fmaxf(s0, s1));
    const float scales = div127(mx);
    const float t_scales = (scales != 0.0f) ? rcp_exact(scales) : 0.0f;
    uint32_t lo0, hi0, lo1, hi1;
    float La, Lb;
    q_table8<true>(x[0], x[1], x[2], x[3], t_scales, lo0, hi0, La);
    q_table8<true>(x[4], x[5], x[6], x[7], t_scales, lo1, hi1, Lb);
    *tslot = make_uint4(lo0, hi0, lo1, hi1);
    float va = -La, vb = -Lb;               // lut_biases, lut_ctor.cc:25-31 (see k_gemv_quad)
    va = __fadd_rn(va, qdpp_f<0x4E>(va));
    vb = __fadd_rn(vb, qdpp_f<0x4E>(vb));
    va = __fadd_rn(va, qdpp_f<0xB1>(va));
    vb = __fadd_rn(vb, qdpp_f<0xB1>(vb));
    const float v = __fadd_rn(va, vb);
    const float c1 = qdpp_f<0x104>(v);
    if ((p & 7) == 0) {
        scslot[0] = __fmul_rn(0.5f, scales);
        scslot[TMAC_IMG2_SC ? 2 : 32 * nst] = __fmul_rn(0.5f, __fadd_rn(__fadd_rn(0.0f, v), c1));
    }
}

// Unified-scale calls (BitNet: one act group = the whole row, qgemm.py:93-96, 170-174): ONE block per op.  The scale is a maximum over K and
// lut_biases ONE fp32 chain over the K / 32 chunk sums in order (lut_ctor.cc:157,218) -- k_decode_chain's SM = 2 build, bit for bit: pass 1
// maxima and chunk sums, pass 2 the tables with the row's scale, the chain by one lane.  Image: the tables as above, then float 0 = lut_scales,
// float 1 = lut_biases, floats 16 + 8 m + g = unified scale g of the op's matrix m (k_decode_chain's LDS layout; static data copied here so that
// the service wave's scale-final epilogue reads LDS only: a global load per combination sat between two workgroup barriers and stretched
// every BitNet visit, profiles/r06_stream_schedule.txt).
template <bool SCF16>
__global__ __launch_bounds__(256) void k_lut_images_us(const ChainOp* __restrict__ ops) {
    __shared__ float s_cs[768];                 // chunk sums (K <= 24576)
    __shared__ float s_mx[4];
    const ChainOp& d = ops[blockIdx.x];
    const int K = d.K, P = K / 8, nu = d.nu, nst = d.nst;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    uint4* tab = reinterpret_cast<uint4*>(const_cast<void*>(d.img));
    float* l_us = reinterpret_cast<float*>(tab + IMG2_STEP * nst);
    auto load8 = [&](int p, float (&x)[8]) __attribute__((always_inline)) {
        if (d.in_gran & 2) {
            const float4* src = reinterpret_cast<const float4*>(d.in) + 2 * (size_t)p;
            const float4 a0 = src[0], a1 = src[1];
            x[0] = a0.x; x[1] = a0.y; x[2] = a0.z; x[3] = a0.w; x[4] = a1.x; x[5] = a1.y; x[6] = a1.z; x[7] = a1.w;
        } else {
            const uint4 v = reinterpret_cast<const uint4*>(d.in)[p];
            const uint32_t r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const __half2 hh = *reinterpret_cast<const __half2*>(&r[i]);
                x[2 * i] = __low2float(hh); x[2 * i + 1] = __high2float(hh);
            }
        }
    };
    float mx = 0.f;
    for (int p = tid; p < P; p += 256) {        // (P % 4 == 0: the four lanes of a chunk are valid together)
        float x[8];
        load8(p, x);
        mx = fmaxf(mx, __fadd_rn(__fadd_rn(fabsf(x[0]), fabsf(x[1])), __fadd_rn(fabsf(x[2]), fabsf(x[3]))));
        mx = fmaxf(mx, __fadd_rn(__fadd_rn(fabsf(x[4]), fabsf(x[5])), __fadd_rn(fabsf(x[6]), fabsf(x[7]))));
        float va = -__fadd_rn(__fadd_rn(__fadd_rn(x[0], x[1]), x[2]), x[3]);
        float vb = -__fadd_rn(__fadd_rn(__fadd_rn(x[4], x[5]), x[6]), x[7]);
        va = __fadd_rn(va, qdpp_f<0x4E>(va));      // lane ^ 2: v0+v4 | v2+v6      (lut_ctor.cc:25-31)
        vb = __fadd_rn(vb, qdpp_f<0x4E>(vb));
        va = __fadd_rn(va, qdpp_f<0xB1>(va));      // lane ^ 1: (v0+v4)+(v2+v6)
        vb = __fadd_rn(vb, qdpp_f<0xB1>(vb));
        if ((p & 3) == 0) s_cs[p >> 2] = __fadd_rn(va, vb);
    }
    mx = q_row_allmax(mx);
    mx = q_xor_max_f(mx);
    if (lane == 0) s_mx[w] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
    const float gscale = div127(mx);
    const float gtinv = (gscale != 0.0f) ? rcp_exact(gscale) : 0.0f;
    for (int p = tid; p < P; p += 256) {
        float x[8];
        load8(p, x);
        uint32_t lo0, hi0, lo1, hi1;
        float La, Lb;
        q_table8<true>(x[0], x[1], x[2], x[3], gtinv, lo0, hi0, La);
        q_table8<true>(x[4], x[5], x[6], x[7], gtinv, lo1, hi1, Lb);
        tab[img2_index(p >> 2, p & 3)] = make_uint4(lo0, hi0, lo1, hi1);
    }
    for (int j4 = 0; j4 < 4; ++j4)
        for (int u = nu + tid; u < nst * 64; u += 256) tab[img2_index(u, j4)] = make_uint4(0u, 0u, 0u, 0u);
    if (tid == 0) {
        float biases = 0.0f;
        for (int c = 0; c < nu; ++c) biases = __fadd_rn(biases, s_cs[c]);
        l_us[0] = gscale;
        l_us[1] = biases;
    }
    if (tid >= 64 && tid < 64 + 4 * CHAIN_US_MAX_GROUPS) {
        const int mi = (tid - 64) / CHAIN_US_MAX_GROUPS, g = (tid - 64) % CHAIN_US_MAX_GROUPS;
        if (mi < d.nmat && g < d.m_groups)
            l_us[16 + mi * CHAIN_US_MAX_GROUPS + g] = SCF16 ? __half2float(reinterpret_cast<const __half*>(d.m[mi].SC)[g]) : reinterpret_cast<const float*>(d.m[mi].SC)[g];
    }
}

hipError_t launch_lut_images(const ChainOp* d_ops, int nops, int max_nst, int sm, bool sc_f16, hipStream_t st) {
    if (nops < 1 || max_nst < 1) return hipErrorInvalidValue;
    if (sm == 2 && sc_f16) hipLaunchKernelGGL(k_lut_images_us<true>, dim3(nops), dim3(256), 0, st, d_ops);
    else if (sm == 2) hipLaunchKernelGGL(k_lut_images_us<false>, dim3(nops), dim3(256), 0, st, d_ops);
    else hipLaunchKernelGGL(k_lut_images, dim3(max_nst, nops), dim3(256), 0, st, d_ops);
    return hipGetLastError();
}
#endif   // !TMAC_STREAM_QW_TU

// ---------------------------------------------------------------------------------------------
#ifndef TMAC_STREAM_KO
#define TMAC_STREAM_KO 0          // timing experiments only (wrong results): 1 no combine / store, 2 no finish at all, 4 no image loads, 8 no lookups
#endif
// RING: weight fragments in flight per lookup wave.  MINW: waves per SIMD the register allocation must leave room for -- 4 with one workgroup
// per CU (13 waves), 8 with TWO (StreamArgs::nsplit = 2: the workgroups of a CU take alternate ops -- twice the waves hide a wave's own
// serial latency per item, which is what bounds the kernel: profiles/r05_stream_knockouts.txt).  8, not the 7 that 26 waves need: the second
// workgroup of a CU is placed only with <= 64 VGPRs per wave (measured: profiles/r05_stream_stamps.txt), which is what 8 waves per SIMD ask for.
// Two workgroups share a CU only while the kernel's SGPR allocation stays at 80 (74 + VCC, FLAT_SCRATCH, XNACK_MASK; granule 16): with 96 the
// second workgroup of a CU waited for the first one to end (measured with the profiling build's place-and-time stamps:
// profiles/r05_stream_stamps.txt), so the count is capped (amdgpu_num_sgpr(n) leaves n - 8 to the kernel; the compiler parks what does not
// fit in VGPR lanes: 13 lane moves in the whole kernel with 74, 92 with 64).
#define TMAC_STREAM_ATTR __attribute__((amdgpu_num_sgpr(82)))
// QW, the QUARTER-WALK form (round 6).  An item of the form above is one row quad x 64 units: a K whose last 64-unit step is ragged (11008 =
// 5.4 steps, BitNet's 3200 = 1.6, 8640 = 4.2) pays whole items for it -- 10 to 22 % of all lookups on zero tables.  The adder hands every
// lane of k-block kb (lanes 16 kb .. 16 kb + 15) to the output lanes L with (L >> 2 & 3) == kb and nothing crosses k-blocks, so the four
// k-blocks of an item need not belong to one row quad: here an item is FOUR CONSECUTIVE ROW QUADS (a "group": 16 rows) x 16 units (a quarter
// step), k-block kb = quad 4 g + kb of the group.  Same 2 KB of W2 weights, same instructions, but K is walked in quarters: ceil(nu / 16) x 16
// units per quad instead of ceil(nu / 64) x 64 (11008: 352 instead of 384; 3200: 112 instead of 128; 8640: 272 instead of 320).  The lane's
// accumulator still belongs to ONE row for the whole walk (row 4 kb' + beta of the group for output lane L = 16 ug + 4 kb' + beta), a lane's
// 16 bytes are the SAME bytes of the unchanged QUAD layout (voffset = its quad's offset + 16 (lane & 15), soff advances by quarters: no per-
// item address arithmetic, no padding mask -- a ragged last quarter reads stored zero nibbles), the tables of a quarter are read by all four
// k-blocks (lanes 16 kb + j read unit 16 t + j).  Rows are dealt in groups (q_end / q_per / q_extra / the roles count groups; every matrix
// has a multiple of 16 rows: the host checks), the partial sums of a wave are 16 per item walk, the wave-wide reduction is two cross-row
// steps instead of four.  Integers are the same integers (tapped); the fp32 partial sums of a row are added in another order than
// k_gemv_quad's, so per-group-scale outputs are specified to the oracle's tolerance (<= 1e-3, measured <= 2e-5) like every N = 1 kernel
// against the reference, and no longer bit-identical to the stand-alone launch; unified-scale outputs stay bit-identical (exact totals).
template <int BITS, bool ZP, bool SCF16, int RING, int MINW, int SM, bool TAP, bool QW>
__global__ __launch_bounds__(STREAM_FT, MINW) TMAC_STREAM_ATTR void k_gemv_stream(StreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    constexpr int NWV = STREAM_NLW;                     // lookup waves; wave NWV is the loader
    constexpr int RPW = QW ? 16 : 4;                    // rows (partial sums) a wave hands over per closed iteration
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    // nsplit workgroups share every row range: workgroup (bx, part) walks the ops part, part + nsplit, ... with the rows of range bx.  They
    // are independent of each other (nothing in this kernel waits for another workgroup); whether they share a CU is the scheduler's business.
    const int nsplit = a.nsplit, part = (int)blockIdx.x % nsplit, bx = (int)blockIdx.x / nsplit;
    // the schedule (tmac_chain.h): row range bx belongs to class bx * ncls / ranges and walks that class' visit list -- the ops whose rows the
    // host dealt to a block of classes containing it; the two workgroups of a range take alternate visits
    const int cls = __builtin_amdgcn_readfirstlane(bx * a.ncls / ((int)gridDim.x / nsplit));
    const int* my_roles = a.roles + (size_t)cls * a.vmax * STREAM_ROLE_INTS;
    const int nops = max((a.nvis[cls] - part + nsplit - 1) / nsplit, 0);    // this workgroup's visits: local j = visit j * nsplit + part of the class
    if (nops == 0) return;
    float* l_red = reinterpret_cast<float*>(lds + 2 * (size_t)a.buf_u4);    // [2][NWV][RPW][CHAIN_RED] partials of split quads / groups
    uint4* l_ops = lds + 2 * (size_t)a.buf_u4 + (2 * NWV * RPW * CHAIN_RED * sizeof(float)) / 16;
    {
        constexpr int U = (int)(sizeof(ChainOp) / 16);
        const uint4* gsrc = reinterpret_cast<const uint4*>(a.ops);
        for (int idx = tid; idx < nops * U; idx += STREAM_FT) {
            const int jl = idx / U, r = idx - jl * U;
            l_ops[idx] = gsrc[my_roles[(size_t)(jl * nsplit + part) * STREAM_ROLE_INTS + SR_OP] * U + r];
        }
        __syncthreads();
    }
    const cop_ptr ops = reinterpret_cast<cop_ptr>(l_ops);
    // barriers of op i, executed by all thirteen waves: A(i) -- the tables of op i are in LDS and everybody has left op i - 1 -- then one
    // per workgroup iteration of the op (finish()).  The number of iterations depends on the workgroup alone.
    auto wg_iters = [&](cop_ptr d) __attribute__((always_inline)) {
        const int qper = uni(d->q_per), qex = uni(d->q_extra), ipi = uni(d->ipi), iinv = uni(d->ipi_inv), bl = bx - uni(d->wg_lo);
        const int cnt = qper + (bl < qex ? 1 : 0);
        return ((cnt + ipi - 1) * iinv) >> 16;
    };

    if (w == NWV) {
        // ---- the loader: op j's image global -> LDS buffer j & 1, one op ahead of the lookups.  1 KB per instruction, lane l lands at
        // M0 + 16 l; images are padded to whole KB (zeros), so is the buffer.  The buffer of op j + 1 was op j - 1's: free since A(j). ----
        typedef __attribute__((address_space(3))) void* lds_vp;
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint4*)lds;
        auto load_image = [&](int j) __attribute__((always_inline)) {
            const cop_ptr d = ops + j;
            const int n16 = uni(d->img_u4);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(uni(d->img)), (short)0, n16 * 16, 0x00020000);
            const uint32_t base = lds0 + (uint32_t)(j & 1) * (uint32_t)a.buf_u4 * 16u;
            for (int o = 0; o < n16; o += 64)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vp)(uintptr_t)(base + (uint32_t)o * 16u), 16, (o + lane) * 16, 0, 0, 0);
        };
        // ... and the combiner: after the barrier that closes a workgroup iteration, lane (quad slot, row) of this wave sums the wpq partials of
        // its row (in wave order, as k_gemv_quad does) and stores the output -- the lookup waves go straight on.  The combination of an op's LAST
        // iteration is deferred behind the next op's A barrier (the partials sit in the other half of the double-buffered area by then), so
        // that A never waits for it: between A and the op's first closing barrier the lookup waves have at least an item to do.
        int s_par = 0;
        auto combine = [&](int j, int t, int par) __attribute__((always_inline)) {
            const cop_ptr d = ops + j;
            const int qper = uni(d->q_per), qex = uni(d->q_extra), ipi = uni(d->ipi), wpq = uni(d->wpq), bl = bx - uni(d->wg_lo);
            const int q_lo = bl * qper + min(bl, qex), cnt = qper + (bl < qex ? 1 : 0);
            // lane = (slot of the iteration, row of the slot's quad / group): 16 quads x 4 rows, or (QW) 4 groups x 16 rows per pass
            constexpr int SPP = 64 / RPW;                 // slots per pass
            for (int pass = 0; pass * SPP < ipi; ++pass) {
            const int p_qs = pass * SPP + (int)((unsigned)lane / RPW), p_row = lane & (RPW - 1);
            const int p_gql = q_lo + t * ipi + p_qs;
            if (p_qs < ipi && p_gql < q_lo + cnt) {
                const int e0 = uni(d->q_end[0]), e1 = uni(d->q_end[1]), e2 = uni(d->q_end[2]);
                const int p_mi = (p_gql >= e0 ? 1 : 0) + (p_gql >= e1 ? 1 : 0) + (p_gql >= e2 ? 1 : 0);
                const int p_lq = p_gql - (p_gql >= e2 ? e2 : (p_gql >= e1 ? e1 : (p_gql >= e0 ? e0 : 0)));
                const unsigned long long p_c = reinterpret_cast<unsigned long long>(d->m[p_mi].C);
                const float* red = l_red + par * (NWV * RPW * CHAIN_RED);
                float v;
                if (SM == 2) {
                    // exact int32 totals per bit-plane, then scale-final (qgemm.py:170-174,192-206) as k_decode_chain's epilogue:
                    // C = ((sum_p float(cb_p) alpha_p) lut_scales + lut_biases / 2) Scale; the two LUT scalars sit behind the op's tables in LDS
                    int32_t cb[BITS];
#pragma unroll
                    for (int pl = 0; pl < BITS; ++pl) cb[pl] = 0;
                    const int32_t* redi = reinterpret_cast<const int32_t*>(red);
                    for (int ww = 0; ww < wpq; ++ww)
#pragma unroll
                        for (int pl = 0; pl < BITS; ++pl) cb[pl] += redi[((p_qs * wpq + ww) * RPW + p_row) * CHAIN_RED + pl];
                    if constexpr (TAP) {
                        if (a.tap) {          // the exact totals per bit-plane, as they enter scale-final
                            const int opi = my_roles[(size_t)(j * nsplit + part) * STREAM_ROLE_INTS + SR_OP];
#pragma unroll
                            for (int pl = 0; pl < BITS; ++pl) a.tap[a.tap_off[opi] + (size_t)(RPW * p_gql + p_row) * BITS + pl] = cb[pl];
                        }
                    }
                    float acc = 0.f;
#pragma unroll
                    for (int pl = 0; pl < BITS; ++pl) {
                        const float tp = __fmul_rn((float)cb[pl], q_alpha(pl));
                        acc = (pl == 0) ? tp : __fadd_rn(acc, tp);
                    }
                    const float* l_us = reinterpret_cast<const float*>(lds + (size_t)(j & 1) * a.buf_u4 + IMG2_STEP * (size_t)uni(d->nst));
                    const float vv = __fadd_rn(__fmul_rn(acc, l_us[0]), __fmul_rn(l_us[1], 0.5f));
                    const int mg = uni(d->m_groups);
                    const int g = mg == 1 ? 0 : (RPW * p_lq + p_row) / (d->m[p_mi].Mw / mg);
                    v = __fmul_rn(vv, l_us[16 + p_mi * CHAIN_US_MAX_GROUPS + g]);       // the matrix' unified scale: in the image (k_lut_images_us)
                } else {
                v = red[((p_qs * wpq) * RPW + p_row) * CHAIN_RED];
                for (int ww = 1; ww < wpq; ++ww) v = __fadd_rn(v, red[((p_qs * wpq + ww) * RPW + p_row) * CHAIN_RED]);
                }
                asm volatile("" : "+v"(v));       // the fp16 output is the fp32 result rounded once more (no fused convert: k_decode_chain)
                const size_t oi = (size_t)(RPW * p_lq + p_row);
                if (a.out_f16) c_store_b16(p_c + 2 * oi, (uint32_t)__half_as_ushort(__float2half_rn(v)));
                else c_store_b32(p_c + 4 * oi, __float_as_uint(v));
            }
            }
        };
        load_image(0);
        int prev_nit = 0;
        for (int j = 0; j < nops; ++j) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            c_lds_barrier();                                  // A(j)
            if (!(TMAC_STREAM_KO & 1) && j > 0 && prev_nit > 0) combine(j - 1, prev_nit - 1, s_par ^ 1);
            if (!(TMAC_STREAM_KO & 4) && j + 1 < nops) load_image(j + 1);
            const int nit = wg_iters(ops + j);
            if (!(TMAC_STREAM_KO & 2))
                for (int t = 0; t < nit; ++t) {
                    c_lds_barrier();                          // closes workgroup iteration t of op j: partials in half s_par
                    if (!(TMAC_STREAM_KO & 1) && t + 1 < nit) combine(j, t, s_par);
                    s_par ^= 1;
                }
            prev_nit = nit;
        }
        if (!(TMAC_STREAM_KO & 1) && prev_nit > 0) combine(nops - 1, prev_nit - 1, s_par ^ 1);
        return;
    }

    CSel<BITS> sel;
    c_selectors<BITS, SM>(sel, lane);
    uint32_t k3 = 0x03020100u;
    asm volatile("" : "+v"(k3));
    uint32_t lane16 = QW ? (uint32_t)(lane & 15) * 16u : (uint32_t)lane * 16u;       // the lane's 16 bytes of a table row / weight block
    asm volatile("" : "+v"(lane16));
    uint32_t lk4 = QW ? 8u * (uint32_t)(lane >> 4)                           // QW: act groups 2 ug, 2 ug + 1 of the quarter (output lane 16 ug + 4 kb + beta)
                      : 4u * (uint32_t)(2 * (lane & 12) + 2 * (lane >> 4));      // the lane's two act groups inside a step's 32 (c_compute)
    asm volatile("" : "+v"(lk4));
    const int wl = NWV - 1 - w;             // logical wave index of the roles and of the partial sums (k_decode_chain's order)

    // a wave's share of an op -- quad slot qs of every workgroup iteration, steps h, h + wpq, ... (k_decode_chain's role_of) -- and the op's
    // geometry: the host's record (tmac_chain.h, STREAM_ROLE_INTS), two scalar loads.  The wait is part of the statement: the compiler does
    // not know of these loads (an LDS wait it counts can only become longer by them).
    typedef int sr16 __attribute__((ext_vector_type(16)));
    typedef int sr4 __attribute__((ext_vector_type(4)));
    auto load_role = [&](int jl, sr16& rc, sr4& rw) __attribute__((always_inline)) {
        const int* rp = my_roles + (size_t)(jl * nsplit + part) * STREAM_ROLE_INTS;
        const int* rpw = rp + SR_COMMON + SRW_INTS * wl;
        asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx4 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(rc), "=&s"(rw) : "s"(rp), "s"(rpw) : "memory");
    };
    auto items_of = [&](const sr16& rc, const sr4& rw) __attribute__((always_inline)) {
        const int nq = bx - rc[SR_WLO] < rc[SR_QEXTRA] ? (int)((unsigned)rw[SRW_NQ] >> 16) : (rw[SRW_NQ] & 0xffff);
        return nq * rw[SRW_NSTEPS];
    };

    // ---- the issue cursor: this wave's items of ops 0, 1, ... in order, RING items ahead of the lookups ----
    CFrag<BITS> ring[RING];
    const __amdgpu_buffer_rsrc_t null_rs = __builtin_amdgcn_make_buffer_rsrc(static_cast<uint4*>(nullptr), (short)0, 0, 0x00020000);   // every lane out of range: zeros, no fetch
    const TMAC_GLOBAL char* ops_g = as_global(reinterpret_cast<const char*>(a.ops));     // a dummy's scale word comes from a mapped address
    const uint32_t dummy_boff = (uint32_t)(lane & 3) * (uint32_t)((ZP ? 2 : 1) * (SCF16 ? 2 : 4));
    uint32_t v_sc0 = dummy_boff;                      // per op: the lane's scale offset inside an item whose units all lie below K (refill)
    bool i_rag = false;                               // per op: the last step (quarter) is ragged
    uint32_t v_wq = lane16, v_scq = dummy_boff;      // QW, per op: the lane's weight offset (its k-block's quad of the group + 16 (lane & 15)) and scale offset (its output quad + row)
    // The cursor's state is what an item costs: every instruction of this lambda is paid once per item and wave (>= 4 cycles each).  The quad
    // is tracked as a running global index with the current matrix' quad range around it (quads ascend within an op: one compare says
    // whether the matrix changes); the lane's part of the scale address that depends on the op alone (c0 >> gs_shift) is kept per op.
    int i_op = -1, i_left = 0, i_st = 0, i_h = 0, i_wpq = 1, i_nst = 1, i_nsg1 = 0, i_gsh = 0, i_nu = 0, i_gq = 0, i_ipi = 1, i_n64 = 1;
    int q_lo = 0, q_hi = 0, q_woff = 0;
    bool q_stale = true;
    cop_ptr i_d = ops;
    __amdgpu_buffer_rsrc_t q_rs = null_rs;
    const TMAC_GLOBAL char* q_sc = ops_g;
    const TMAC_GLOBAL char* q_scm = ops_g;
    int q_scstride = 0;
    const uint32_t c0 = QW ? (uint32_t)(4 * (lane >> 4)) : (uint32_t)(4 * (lane & 12) + 4 * (lane >> 4));       // the lane's first unit inside a step / quarter (c_issue)
    uint32_t c0g = c0;                                                       // c0 >> gs_shift of the issue cursor's op
    constexpr int SC_SHIFT = (ZP ? 1 : 0) + (SCF16 ? 3 : 4);               // log2 of a scale group's bytes per row quad: 4 rows x (scale [, zero]) x 2 | 4
    auto refill = [&](CFrag<BITS>& f) __attribute__((always_inline)) {
        while (i_left == 0 && i_op < nops) {            // enter the next op in which this wave has items
            ++i_op;
            if (i_op < nops) {
                i_d = ops + i_op;
                sr16 rc; sr4 rw;
                load_role(i_op, rc, rw);
                i_left = items_of(rc, rw);
                i_st = rw[SRW_H]; i_h = rw[SRW_H]; i_wpq = rc[SR_WPQ]; i_nst = rc[SR_NST]; i_ipi = rc[SR_IPI];
                const int bl = bx - rc[SR_WLO];
                i_gq = bl * rc[SR_QPER] + min(bl, rc[SR_QEXTRA]) + rw[SRW_QS];
                i_nsg1 = rc[SR_NSG] - 1; i_gsh = rc[SR_GSH]; i_nu = rc[SR_NU];
                q_scstride = rc[SR_NSG] << SC_SHIFT;
                c0g = c0 >> i_gsh;
                if constexpr (QW) {
                    i_n64 = (rc[SR_TSTRIDE] - 1) >> 6;                   // 64-unit steps per quad in the weight layout
                    v_wq = (uint32_t)(lane >> 4) * (uint32_t)(i_n64 * (BITS * 1024)) + lane16;
                    v_scq = (uint32_t)((lane >> 2) & 3) * (uint32_t)q_scstride + dummy_boff;
                    v_sc0 = (c0g << SC_SHIFT) + v_scq;
                    i_rag = (i_nu & 15) != 0;
                } else {
                    v_sc0 = (c0g << SC_SHIFT) | dummy_boff;
                    i_rag = (i_nu & 63) != 0;
                }
                q_hi = 0; q_stale = true;                 // (no matrix yet: the first quad takes the slow path)
            }
        }
        const bool real = i_left > 0;
        if (real && q_stale) {                            // what depends on the quad alone
            if (i_gq >= q_hi) {                           // another matrix of the op (per matrix a wave enters, not per quad): its quad range and pointers
                const int e0 = uni(i_d->q_end[0]), e1 = uni(i_d->q_end[1]), e2 = uni(i_d->q_end[2]);
                const int mi = (i_gq >= e0 ? 1 : 0) + (i_gq >= e1 ? 1 : 0) + (i_gq >= e2 ? 1 : 0);
                q_lo = i_gq >= e2 ? e2 : (i_gq >= e1 ? e1 : (i_gq >= e0 ? e0 : 0));
                q_hi = i_gq >= e2 ? 0x7fffffff : (i_gq >= e1 ? e2 : (i_gq >= e0 ? e1 : e0));
                q_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(uni(i_d->m[mi].W)), (short)0, 0x7fffffff, 0x00020000);
                q_scm = as_global(uni(reinterpret_cast<const char*>(i_d->m[mi].SC)));
            }
            const int lq = QW ? 4 * (i_gq - q_lo) : i_gq - q_lo;           // (QW: the group's first quad)
            q_sc = q_scm + (size_t)lq * (size_t)q_scstride;
            q_woff = lq * (QW ? i_n64 : i_nst) * (BITS * 1024);
            q_stale = false;
        }
        CItemOps io;
        if (real) {
            // c_item_operands with the op's constants folded: scale group min(st * (64 >> gsh) + (c0 >> gsh), nsg - 1); lanes whose unit lies past K
            // re-read lane 0's 16 bytes (their tables are zero tables)
            // The lane's scale group is (first unit of the item >> gsh) + c0g: a uniform term, added to the scale POINTER by the scalar unit, and a
            // per-lane constant of the op (v_sc0) -- no vector arithmetic per item.  Only a ragged LAST step (quarter) has lanes past K, whose
            // group is clamped (and, in the 64-unit form, whose weight offset re-reads lane 0's bytes): the vector form is kept for that item.
            io.rs = q_rs;
            const int ub = QW ? i_st << 4 : i_st << 6;
            if (i_rag && i_st == i_nst - 1) {
                const uint32_t sg = min((uint32_t)(ub >> i_gsh) + c0g, (uint32_t)i_nsg1);
                io.sc = q_sc;
                io.boff = QW ? (sg << SC_SHIFT) + v_scq : ((sg << SC_SHIFT) | dummy_boff);
                io.l16 = QW ? v_wq : ((lane < i_nu - ub) ? lane16 : 0u);
            } else {
                io.sc = q_sc + ((size_t)(ub >> i_gsh) << SC_SHIFT);
                io.boff = v_sc0;
                io.l16 = QW ? v_wq : lane16;
            }
            io.soff = QW ? q_woff + (i_st >> 2) * (BITS * 1024) + (i_st & 3) * 256 : q_woff + i_st * (BITS * 1024);
        } else { io.rs = null_rs; io.soff = 0; io.sc = ops_g; io.boff = dummy_boff; io.l16 = 0u; }      // behind the last op: keeps the FIFO's depth
        c_issue_static<BITS, ZP, SCF16, SM>(f, io);
        if (real) {
            --i_left;
            i_st += i_wpq;
            if (i_st >= i_nst) { i_st = i_h; i_gq += i_ipi; q_stale = true; }
        }
    };
#pragma unroll
    for (int k = 0; k < RING; ++k) refill(ring[k]);

    // ---- the lookup cursor ----
    int c_op = -1, c_left = 0, c_it = 0, c_st = 0, parity = 0, my_iter = 0;
    int tstride = 1, nst = 1, wpq = 1, h = 0;
    int t_gq0 = 0, t_ipi = 1, t_G = 0;              // TAP: the wave's first quad of the op, quads per workgroup iteration, act groups per row
    int32_t* t_base = nullptr;
    uint4* tab = lds;
    float* l_ls = reinterpret_cast<float*>(lds);
    float* l_lb = l_ls;
    float cacc = 0.f;
    int32_t iacc[BITS];
#pragma unroll
    for (int pl = 0; pl < BITS; ++pl) iacc[pl] = 0;

#ifdef TMAC_STREAM_STAMPS
    // profiling build: where a lookup wave's cycles go (s_memtime; each stamp costs a scalar memory round trip, so the sums are upper bounds).
    // The sums live in LDS (behind the descriptors: stream_lds_bytes reserves the room in these builds) and the clock is kept in 32 bits, so
    // that the build keeps the kernel's register footprint -- with 20 more SGPRs the two workgroups of a CU were no longer co-resident.
    uint32_t* t_lds = reinterpret_cast<uint32_t*>(l_ops) + (size_t)nops * (sizeof(ChainOp) / 4) + (size_t)w * 8;
    if (lane < 8) t_lds[lane] = 0;
    uint32_t t_prev = (uint32_t)__builtin_readcyclecounter();
#if TMAC_STREAM_STAMPS >= 2
    if (lane == 0) t_lds[7] = 0u - t_prev;
#else
    if (lane == 0) { t_lds[6] = t_prev; t_lds[7] = (uint32_t)(__builtin_readcyclecounter() >> 32); }
#endif
#if TMAC_STREAM_STAMPS >= 2
#define TMAC_ST(i) do { const uint32_t t_now = (uint32_t)__builtin_readcyclecounter(); if (lane == 0) (void)__hip_atomic_fetch_add(&t_lds[i], t_now - t_prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); t_prev = t_now; } while (0)
#else
#define TMAC_ST(i) do { } while (0)         // level 1: life time and place of the workgroup only
#endif
#else
#define TMAC_ST(i) do { } while (0)
#endif
    // closes a workgroup iteration of op c_op: the wave's partial sums to LDS, one barrier (the service wave combines the wpq partials of
    // every quad of the iteration and stores the outputs)
    auto finish = [&](bool have, float acc_in) __attribute__((always_inline)) {
        if (TMAC_STREAM_KO & 2) { ++c_it; return; }
        float* red = l_red + parity * (NWV * RPW * CHAIN_RED);
        if (SM == 2) {
            // exact integer totals of the lane's row (lane & 3), per bit-plane: lanes of a DPP row by rotation, rows by two cross-row moves
            // (QW: the 16 lanes of a DPP row are 16 different rows of the group: the cross-row moves only)
            int32_t* redi = reinterpret_cast<int32_t*>(red);
#pragma unroll
            for (int pl = 0; pl < BITS; ++pl) {
                uint32_t v = have ? (uint32_t)iacc[pl] : 0u;
                if constexpr (!QW) {
                    v += qdpp_u<0x124>(v);
                    v += qdpp_u<0x128>(v);
                }
                v = q_xor_add_u(v);
                if (lane < RPW) redi[(wl * RPW + lane) * CHAIN_RED + pl] = (int32_t)v;
                iacc[pl] = 0;
            }
        } else {
        float acc = 0.f;
        if (have) {
            acc = acc_in;
            if constexpr (!QW) {
                acc = __fadd_rn(acc, qdpp_f<0x124>(acc));     // lanes with the same row: rotate by 4, 8 within the DPP row
                acc = __fadd_rn(acc, qdpp_f<0x128>(acc));
            }
            acc = q_xor_add_f(acc);
        }
        if (lane < RPW) red[(wl * RPW + lane) * CHAIN_RED] = acc;
        }
        TMAC_ST(2);
        c_lds_barrier();                          // the service wave combines and stores behind it
        TMAC_ST(3);
        parity ^= 1;
        ++c_it;
    };
    // the lookup cursor leaves op c_op (remaining barriers of the op) and enters the next op in which this wave has items, passing the
    // barriers of the ops in between
    bool done = false;
    auto advance = [&]() __attribute__((always_inline)) {
        for (;;) {
            if (c_op >= 0) while (c_it < my_iter) finish(false, 0.f);
            ++c_op;
            if (c_op >= nops) { done = true; return; }
            sr16 rc; sr4 rw;
            load_role(c_op, rc, rw);
            my_iter = bx - rc[SR_WLO] < rc[SR_QEXTRA] ? rc[SR_IT_HI] : rc[SR_IT_LO];
            tstride = rc[SR_TSTRIDE]; nst = rc[SR_NST]; wpq = rc[SR_WPQ]; h = rw[SRW_H];
            tab = lds + (size_t)(c_op & 1) * a.buf_u4;
            l_ls = reinterpret_cast<float*>(tab + IMG2_STEP * ((tstride - 1) >> 6));    // behind the tables ([steps][4][65] uint4): ls / 2, then lb / 2 per act group
            l_lb = TMAC_IMG2_SC ? l_ls : l_ls + ((tstride - 1) >> 1);
            c_left = items_of(rc, rw); c_it = 0; c_st = h;
            if constexpr (TAP) {
                const int bl = bx - rc[SR_WLO];
                t_gq0 = bl * rc[SR_QPER] + min(bl, rc[SR_QEXTRA]) + rw[SRW_QS]; t_ipi = rc[SR_IPI]; t_G = rc[SR_NU] >> 1;
                t_base = a.tap ? a.tap + a.tap_off[rc[SR_OP]] : nullptr;
            }
            TMAC_ST(4);
            c_lds_barrier();                          // A(c_op): the loader has this op's tables in LDS
            TMAC_ST(5);
            if (c_left > 0) return;
        }
    };

    advance();                                        // op 0 (or the first op in which this wave has items)
    // Rounds of RING slots; the loop is left at the END of a round only.  Once the wave is past its last item the remaining slots of the
    // round skip the lookups and still refill (a dummy by then): with an exit -- or the op change -- in front of a slot's lookups, the
    // compiler's structured control flow routes those paths through the loop's latch, the waitcnt pass merges "this slot's refill is the
    // youngest load" into the head of the round and stops counting there (s_waitcnt vmcnt(1) in front of slot 0).
    while (!done) {
#pragma unroll
        for (int k = 0; k < RING; ++k) {
            if (!done) {
#if defined(TMAC_STREAM_STAMPS) && TMAC_STREAM_STAMPS >= 2
                {   // the wait the compiler places in front of the slot's first lookup, made explicit: the slot's loads are older than the other slots' refills
                    constexpr int L = BITS + (SCF16 || !ZP ? 1 : 2);
                    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((RING - 1) * L) : "memory");
                    TMAC_ST(0);
                    if (lane == 0) (void)__hip_atomic_fetch_add(&t_lds[6], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
#endif
                if (TMAC_STREAM_KO & 8) cacc += __uint_as_float(ring[k].wq[0].x ^ ring[k].wq[BITS - 1].w ^ ring[k].s0); else
                c_compute<BITS, ZP, SCF16, SM, TAP, true>(ring[k], tab, tstride, l_ls, l_lb, QW ? c_st * 16 : c_st * 64, lane16, lk4, sel, k3, cacc, iacc,
                                                          (TAP && t_base) ? t_base + (size_t)(RPW * (t_gq0 + c_it * t_ipi) + (lane & (RPW - 1))) * t_G : nullptr, t_G,
                                                          QW ? (c_st >> 2) * (16 * IMG2_STEP) + (c_st & 3) * 256 : c_st * (16 * IMG2_STEP));
                asm volatile("" : "+v"(cacc));        // the item's scale chain ends before the slot is refilled (the scale word keeps its register)
            }
            refill(ring[k]);
            TMAC_ST(1);
            if (!done) {
                --c_left;
                c_st += wpq;
                if (c_st >= nst) {
                    finish(true, cacc);
                    cacc = 0.f;
                    c_st = h;
                }
                if (c_left == 0) advance();           // the op's last item: on to the next op BEHIND the refill (every path into the next slot has issued the same loads)
                TMAC_ST(4);
            }
        }
    }
#ifdef TMAC_STREAM_STAMPS
    if (a.stamps && lane == 0) {
        const unsigned long long t_last = __builtin_readcyclecounter();
        unsigned long long* o = a.stamps + ((size_t)blockIdx.x * NWV + w) * 8;
#if TMAC_STREAM_STAMPS >= 2
        for (int i = 0; i < 7; ++i) o[i] = t_lds[i];
        uint32_t hw_id2;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id2));
        o[7] = (unsigned long long)(uint32_t)(t_lds[7] + (uint32_t)t_last) | ((unsigned long long)hw_id2 << 32);      // + where the wave ran (SIMD_ID: bits 5:4)
#else
        const unsigned long long t_first = ((unsigned long long)t_lds[7] << 32) | t_lds[6];
        o[4] = t_last - t_first;
        uint32_t hw_id, xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        o[5] = t_first; o[6] = t_last; o[7] = ((unsigned long long)xcc_id << 32) | hw_id;       // where and when: are the workgroups of a pair co-resident?
#endif
    }
#endif
#undef TMAC_ST

}

template <int BITS, bool QWF>
static hipError_t stream_launch_b(const StreamArgs& a, bool zp, bool sc_f16, int sm, int grid, size_t lds_bytes, hipStream_t st) {
    constexpr int R2 = (BITS <= 3) ? 2 : 1;      // two workgroups per CU: <= 64 VGPRs
    constexpr int R1 = (BITS <= 2) ? 4 : 2;      // one workgroup per CU: ring depth 2..8 measured flat (profiles/r05_stream_knockouts.txt)
#define TMAC_SL2(Z, H, R, MW, S) do { \
        auto* kern = a.tap ? &k_gemv_stream<BITS, Z, H, R1, 4, S, true, QWF> : &k_gemv_stream<BITS, Z, H, R, MW, S, false, QWF>; \
        if (lds_bytes > 64 * 1024) { \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
            if (e != hipSuccess) return e; \
        } \
        hipLaunchKernelGGL(kern, dim3(grid * a.nsplit), dim3(STREAM_FT), lds_bytes, st, a); \
        return hipGetLastError(); } while (0)
#define TMAC_SL(Z, H, S) do { if (a.nsplit >= 2) TMAC_SL2(Z, H, R2, 8, S); else TMAC_SL2(Z, H, R1, 4, S); } while (0)
    if (sm == 2) { if (sc_f16) TMAC_SL(false, true, 2); else TMAC_SL(false, false, 2); }      // unified scales: no zero points (qgemm.py:170-174)
    if (zp) { if (sc_f16) TMAC_SL(true, true, 0); else TMAC_SL(true, false, 0); }
    if (sc_f16) TMAC_SL(false, true, 0);
    TMAC_SL(false, false, 0);
#undef TMAC_SL
#undef TMAC_SL2
}

#if TMAC_STREAM_QW_TU
hipError_t launch_gemv_stream_qw(const StreamArgs& a, int bits, bool zp, bool sc_f16, int sm, int grid, size_t lds_bytes, hipStream_t st) {
    switch (bits) {
        case 1: return stream_launch_b<1, true>(a, zp, sc_f16, sm, grid, lds_bytes, st);
        case 2: return stream_launch_b<2, true>(a, zp, sc_f16, sm, grid, lds_bytes, st);
        case 3: return stream_launch_b<3, true>(a, zp, sc_f16, sm, grid, lds_bytes, st);
        case 4: return stream_launch_b<4, true>(a, zp, sc_f16, sm, grid, lds_bytes, st);
        default: return hipErrorInvalidValue;
    }
}
#else
hipError_t launch_gemv_stream_qw(const StreamArgs& a, int bits, bool zp, bool sc_f16, int sm, int grid, size_t lds_bytes, hipStream_t st);
hipError_t launch_gemv_stream(const StreamArgs& a, int bits, bool zp, bool sc_f16, int sm, bool qw, int grid, size_t lds_bytes, hipStream_t st) {
    if (a.nops < 1 || grid < 1 || a.nsplit < 1 || a.nsplit > 4 || (sm != 0 && sm != 2)) return hipErrorInvalidValue;
    if (qw) return launch_gemv_stream_qw(a, bits, zp, sc_f16, sm, grid, lds_bytes, st);
    switch (bits) {
        case 1: return stream_launch_b<1, false>(a, zp, sc_f16, sm, grid, lds_bytes, st);
        case 2: return stream_launch_b<2, false>(a, zp, sc_f16, sm, grid, lds_bytes, st);
        case 3: return stream_launch_b<3, false>(a, zp, sc_f16, sm, grid, lds_bytes, st);
        case 4: return stream_launch_b<4, false>(a, zp, sc_f16, sm, grid, lds_bytes, st);
        default: return hipErrorInvalidValue;
    }
}
#endif

}  // namespace tmac
