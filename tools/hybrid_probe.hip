// hybrid_probe.hip — can the two lookup mechanisms of this repo run side by side?  (round-5 groundwork; a probe, not product code)
//
// The decode kernels look tables up with v_perm_b32 on 8-entry half tables (VALU: 11 instructions per 8 lookups incl. index and sign
// handling, tmac_quad_core.h q_lookup4_pm) and add with v_mfma_i32_16x16x64_i8.  The prefill kernel turns a weight BYTE (both bit-planes
// of a table for a row) into an 8-byte operand row with one v_perm (address) + one ds_read_b64 from a 256-entry table kept in LDS in 32
// copies (no bank conflict), and lets the matrix core do lookup and sum.  Each mechanism alone sits at about the HBM roofline at 100 %
// of ITS unit (VALU resp. the LDS crossbar): DESIGN.md 4.1, 8.  This probe measures, for one (row quad, 64-unit step) item per wave --
// 32 bytes of 2-bit weights per lane, 2 KB per wave, as in k_decode_chain -- the time per item with 12 waves per CU when the item's
// tables are looked up (0) all by v_perm, (1) all by LDS gathers, (2) half and half, instruction streams interleaved in every wave.
// If (2) lands well below both, a hybrid item is worth building; if it lands near max(0, 1) / 2 + overheads, it is not.
// Results are meaningless as numbers (synthetic weights, one constant B operand): only instruction mix and dependences are real.
//   hipcc --offload-arch=gfx950 -O3 -o tools/hybrid_probe tools/hybrid_probe.hip && ./tools/hybrid_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t b, uint32_t c) { return (a & b) | c; }

// the decode kernels' signed lookup of four nibbles (tmac_quad_core.h): `all` and the negated ones, weighed +1 / -2 by the MFMA's B operand
template <int H>
__device__ __forceinline__ void lookup4_pm(uint32_t w, uint32_t tab_lo, uint32_t tab_hi, uint32_t k3, uint32_t& all, uint32_t& neg) {
    const uint32_t x = H ? (w >> 4) : w;
    all = __builtin_amdgcn_perm(tab_hi, tab_lo, x & 0x07070707u);
    const uint32_t sel3 = and_or(x >> 1, 0x04040404u, k3);
    neg = __builtin_amdgcn_perm(all, 0u, sel3);
}

// MODE 0: v_perm for all 8 tables of the lane's unit; 1: LDS gathers for all; 2: tables 0-3 by v_perm, 4-7 by gathers
template <int MODE>
__global__ __launch_bounds__(768) void k_probe(uint32_t* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // 256 entries x 32 copies x 8 bytes = 64 KB of operand rows
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 256 * 32 * 2; i += 768) reinterpret_cast<uint32_t*>(lds)[i] = (uint32_t)i * 2654435761u;
    __syncthreads();
    // synthetic state: 8 dwords of weights (one unit: 8 tables x 2 planes x 4 rows), 16 dwords of half tables, the adder's B operand
    uint32_t w[8], tb[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = (uint32_t)(tid * 8 + i) * 0x9e3779b9u;
#pragma unroll
    for (int i = 0; i < 16; ++i) tb[i] = (uint32_t)(lane * 16 + i) * 0x85ebca6bu;
    const v4i bsel = {0x01010101, (int)0xfefefefe, 0x01010101, (int)0xfefefefe};
    uint32_t k3 = 0x03020100u;
    asm volatile("" : "+v"(k3));
    const uint32_t copyoff = (uint32_t)(lane & 31) * 8u;
    const uint32_t psel = 0x0c0c0000u | ((4u + (lane & 3)) << 8);            // byte 0 = copy offset, byte 1 = byte (lane & 3) of the dword
    v4i c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, g0 = {0, 0, 0, 0}, g1 = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        // (new "weights" per item: one VALU per dword, the stand-in for the ring's arrival)
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] += 0x01030507u + (uint32_t)it;
        constexpr int NPERM = MODE == 0 ? 4 : MODE == 1 ? 0 : 2;             // table PAIRS looked up by v_perm (4 pairs = 8 tables per unit)
        // ---- v_perm path: per table pair and plane: two lookup4_pm, one MFMA as the adder (c_compute of tmac_chain.hip)
#pragma unroll
        for (int tp = 0; tp < NPERM; ++tp)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                uint32_t pa, ma, pb, mb;
                const int qa = (2 * tp) * 2 + pl, qb = (2 * tp + 1) * 2 + pl;
                if (qa & 1) lookup4_pm<1>(w[qa >> 1], tb[4 * tp], tb[4 * tp + 1], k3, pa, ma); else lookup4_pm<0>(w[qa >> 1], tb[4 * tp], tb[4 * tp + 1], k3, pa, ma);
                if (qb & 1) lookup4_pm<1>(w[qb >> 1], tb[4 * tp + 2], tb[4 * tp + 3], k3, pb, mb); else lookup4_pm<0>(w[qb >> 1], tb[4 * tp + 2], tb[4 * tp + 3], k3, pb, mb);
                if (pl == 0) c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8((v4i){(int)pa, (int)ma, (int)pb, (int)mb}, bsel, c0, 0, 0, 0);
                else c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8((v4i){(int)pa, (int)ma, (int)pb, (int)mb}, bsel, c1, 0, 0, 0);
            }
        // ---- gather path: a weight dword = the joint plane indices of ONE table for the lane's 4 rows: per row one address v_perm + one
        // ds_read_b64 (8 operand bytes = the table's 8 half-table entries weighed by both planes); two tables fill an MFMA's 16-byte A operand,
        // the tables themselves are the B operand (a constant here).  4 rows x 2 tables = 4 MFMAs per table pair.
#pragma unroll
        for (int tp = NPERM; tp < 4; ++tp)
#pragma unroll
            for (int row = 0; row < 4; ++row) {
                const uint32_t selr = 0x0c0c0000u | ((4u + row) << 8);
                const uint32_t a0 = __builtin_amdgcn_perm(w[2 * tp], copyoff, selr), a1 = __builtin_amdgcn_perm(w[2 * tp + 1], copyoff, selr);
                const u2 r0 = *(__attribute__((address_space(3))) const u2*)(uintptr_t)a0;
                const u2 r1 = *(__attribute__((address_space(3))) const u2*)(uintptr_t)a1;
                if (row & 1) g1 = __builtin_amdgcn_mfma_i32_16x16x64_i8((v4i){(int)r0.x, (int)r0.y, (int)r1.x, (int)r1.y}, (v4i){(int)tb[4 * tp], (int)tb[4 * tp + 1], (int)tb[4 * tp + 2], (int)tb[4 * tp + 3]}, g1, 0, 0, 0);
                else g0 = __builtin_amdgcn_mfma_i32_16x16x64_i8((v4i){(int)r0.x, (int)r0.y, (int)r1.x, (int)r1.y}, (v4i){(int)tb[4 * tp], (int)tb[4 * tp + 1], (int)tb[4 * tp + 2], (int)tb[4 * tp + 3]}, g0, 0, 0, 0);
            }
        (void)psel;
    }
    const v4i s = c0 + c1 + g0 + g1;
    out[(size_t)blockIdx.x * 768 + tid] = (uint32_t)(s.x ^ s.y ^ s.z ^ s.w);
}

template <int MODE>
static double run(uint32_t* out, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipLaunchKernelGGL((k_probe<MODE>), dim3(256), dim3(768), 65536, 0, out, 16);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_probe<MODE>), dim3(256), dim3(768), 65536, 0, out, iters);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return (double)best * 1e3 / iters;    // us per item (every wave of the chip processes one item per iteration)
}

int main() {
    uint32_t* out; CK(hipMalloc((void**)&out, (size_t)256 * 768 * 4));
    const int iters = 4000;
    const double t0 = run<0>(out, iters), t1 = run<1>(out, iters), t2 = run<2>(out, iters);
    const double bytes = 256.0 * 12 * 2048;   // weight bytes the chip's 3072 waves stand for per item round
    printf("item = 2 KB of 2-bit weights per wave (64 lookups per lane), 12 waves per CU, 256 CUs; us per item round, equivalent weight stream\n");
    printf("all v_perm (VALU)            %.3f us  %.2f TB/s\n", t0, bytes / t0 * 1e-6);
    printf("all LDS gathers (+ MFMA)     %.3f us  %.2f TB/s\n", t1, bytes / t1 * 1e-6);
    printf("half / half, interleaved     %.3f us  %.2f TB/s\n", t2, bytes / t2 * 1e-6);
    return 0;
}
