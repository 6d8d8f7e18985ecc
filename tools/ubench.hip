// ubench.hip — micro-benchmarks that size the floor for tiny HBM-streaming kernels on MI355X.
//   ./ubench  -> prints per-launch time of: empty kernel, pure streaming reads of 4.7 MB / 12.7 MB with
//   several (blocks, threads, loads-per-thread) shapes, rotating over > 256 MB of buffers (MALL-cold).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

// each thread reads NL uint4, block-strided so every wave-instruction is 1 KiB contiguous
template <int NL, bool NT>
__global__ void k_stream(const u32x4* __restrict__ src, uint32_t* __restrict__ out) {
    const size_t base = (size_t)blockIdx.x * blockDim.x * NL + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 v[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) v[i] = NT ? __builtin_nontemporal_load(src + base + (size_t)i * blockDim.x) : src[base + (size_t)i * blockDim.x];
#pragma unroll
    for (int i = 0; i < NL; ++i) acc ^= v[i];
    uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (r == 0x12345678u) out[blockIdx.x] = r;   // practically never: keeps the loads alive, no store traffic
}

template <int NL, bool NT>
static void run(const char* tag, std::vector<u32x4*>& bufs, size_t bytes, int threads, uint32_t* out) {
    const size_t n4 = bytes / 16;
    const int blocks = (int)(n4 / ((size_t)threads * NL));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) for (auto b : bufs) hipLaunchKernelGGL((k_stream<NL, NT>), dim3(blocks), dim3(threads), 0, 0, b, out);
    CK(hipDeviceSynchronize());
    const int reps = 4;
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) for (auto b : bufs) hipLaunchKernelGGL((k_stream<NL, NT>), dim3(blocks), dim3(threads), 0, 0, b, out);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / (reps * bufs.size());
    printf("%-28s bytes=%8zu blocks=%5d thr=%4d NL=%2d nt=%d : %7.3f us/launch  %7.1f GB/s\n", tag, bytes, blocks, threads, NL, (int)NT, us, bytes / us * 1e-3);
}

// ---- the same floors as a replayed hipGraph of dependent kernel nodes (how bench.py launches the decode step) ----
template <int NL>
static double graph_chain(hipStream_t st, std::vector<u32x4*>& bufs, size_t bytes, int threads, uint32_t* out, bool empty) {
    const size_t n4 = bytes / 16;
    const int blocks = (int)(n4 / ((size_t)threads * NL));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (auto b : bufs) {
        if (empty) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, (int*)nullptr);
        else hipLaunchKernelGGL((k_stream<NL, true>), dim3(blocks), dim3(threads), 0, st, b, out);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 1e30;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms * 1e3 / bufs.size() < best) best = ms * 1e3 / bufs.size();
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return best;
}

static int graph_main() {
    // bytes per launch of the four llama-2-7B W2 decode launches (weights + scales/zeros)
    struct { const char* name; size_t bytes; } L[] = {{"o        4096x4096      ", 4743424}, {"down     4096x11008     ", 12734128},
                                                       {"qkv      3 x 4096x4096  ", 3 * 4743424}, {"gate_up  2 x 11008x4096 ", 2 * 12697600}};
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    uint32_t* out; CK(hipMalloc((void**)&out, 1 << 20));
    std::vector<u32x4*> one(64, nullptr);
    printf("replayed hipGraph of dependent kernel nodes, per node:\n");
    printf("  empty kernel (256 x 256 threads)            : %6.2f us\n", graph_chain<4>(st, one, 1 << 20, 256, out, true));
    for (auto& l : L) {
        const size_t bytes = (l.bytes + 65535) / 65536 * 65536;
        const int nbuf = (int)((size_t)(800u << 20) / bytes) + 1;   // > 3 x the 256 MB MALL
        std::vector<u32x4*> bufs(nbuf);
        for (auto& b : bufs) { CK(hipMalloc((void**)&b, bytes)); CK(hipMemset(b, 0x5a, bytes)); }
        CK(hipDeviceSynchronize());
        double best = 1e30; int bt = 0, bn = 0;
        const double a = graph_chain<4>(st, bufs, bytes, 256, out, false); if (a < best) { best = a; bt = 256; bn = 4; }
        const double b8 = graph_chain<8>(st, bufs, bytes, 256, out, false); if (b8 < best) { best = b8; bt = 256; bn = 8; }
        const double c = graph_chain<4>(st, bufs, bytes, 512, out, false); if (c < best) { best = c; bt = 512; bn = 4; }
        const double d = graph_chain<8>(st, bufs, bytes, 512, out, false); if (d < best) { best = d; bt = 512; bn = 8; }
        const double e = graph_chain<2>(st, bufs, bytes, 256, out, false); if (e < best) { best = e; bt = 256; bn = 2; }
        printf("  pure read of %s %9zu B : %6.2f us  (%6.1f GB/s; best of 5 shapes: %d threads x %d uint4)\n", l.name, l.bytes, best, l.bytes / best * 1e-3, bt, bn);
        for (auto b : bufs) CK(hipFree(b));
    }
    return 0;
}

int rate_main();
// ---- L2 -> LDS streaming probe: what one CU gets from an L2-resident image with a bounded number of bytes in flight -----
// Every wave repeats: issue NB 1 KB global_load_lds (buffer form, scalar offsets), wait for all of them.  All workgroups
// read the SAME image (as the prefill GEMM's workgroups of one XCD do), at a wave-specific rotating offset.
template <int NB, bool TOLDS>
__global__ __launch_bounds__(512) void k_dma(const uint32_t* img, uint32_t img_bytes, int iters, uint32_t* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dl[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(img), (short)0, 0x7fffffff, 0x00020000);
    uint32_t off = (uint32_t)((w * 131 + blockIdx.x * 17) % 1024) * 1024u, acc = 0;
    u32x4 r[NB];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const uint32_t so = (off + b * 65536u) % img_bytes;
            if (TOLDS) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(uintptr_t)(w * NB * 1024 + b * 1024), 16, lane * 16, so, 0, 0);
            else r[b] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, so, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!TOLDS)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc += r[b].x;
        off = (off + 8192u) % img_bytes;
    }
    if (TOLDS) acc = *reinterpret_cast<const uint32_t*>(dl + threadIdx.x * 4);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int NB, bool TOLDS>
static void dma_run(const uint32_t* img, uint32_t bytes, int waves, uint32_t* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 200;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dma<NB, TOLDS>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * NB * 1024));
    hipLaunchKernelGGL((k_dma<NB, TOLDS>), dim3(256), dim3(64 * waves), 8 * NB * 1024, 0, img, bytes, 10, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_dma<NB, TOLDS>), dim3(256), dim3(64 * waves), 8 * NB * 1024, 0, img, bytes, iters, out);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us_it = ms * 1e3 / iters, kb = (double)waves * NB;
    printf("%-9s image %5.1f MB, %d waves/CU x %2d KB in flight: %6.2f us per round, %6.1f GB/s per CU, %5.2f TB/s chip\n", TOLDS ? "-> LDS" : "-> VGPR",
           bytes / 1048576.0, waves, NB, us_it, kb * 1024 / us_it * 1e-3, kb * 1024 * 256 / us_it * 1e-6);
}

int dma_main() {
    uint32_t* img; CK(hipMalloc((void**)&img, 64u << 20)); CK(hipMemset(img, 1, 64u << 20));
    uint32_t* out; CK(hipMalloc((void**)&out, 256 * 512 * 4));
    for (uint32_t mb : {2u, 8u, 64u}) {
        dma_run<8, true>(img, mb << 20, 8, out);
        dma_run<8, true>(img, mb << 20, 4, out);
        dma_run<8, true>(img, mb << 20, 1, out);
        dma_run<4, true>(img, mb << 20, 8, out);
        dma_run<16, true>(img, mb << 20, 8, out);
        dma_run<8, false>(img, mb << 20, 8, out);
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && argv[1][0] == 'd') return dma_main();
    if (argc > 1 && argv[1][0] == 'g') return graph_main();
    if (argc > 1) return rate_main();
    const size_t big = 12734128 / 16384 * 16384 + 16384, small = 4743424 / 16384 * 16384 + 16384;
    const int nbuf = 32;   // 32 x 12.7 MB = 407 MB > 256 MB MALL
    std::vector<u32x4*> bufs(nbuf);
    for (auto& b : bufs) { CK(hipMalloc((void**)&b, big)); CK(hipMemset(b, 0x5a, big)); }
    uint32_t* out; CK(hipMalloc((void**)&out, 1 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0, (int*)nullptr);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0, (int*)nullptr);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("empty kernel back-to-back: %.3f us/launch\n", ms);
    run<4, true>("stream 12.7MB", bufs, big, 256, out);
    run<8, true>("stream 12.7MB", bufs, big, 256, out);
    run<12, true>("stream 12.7MB", bufs, big, 256, out);
    run<4, true>("stream 12.7MB", bufs, big, 512, out);
    run<8, true>("stream 12.7MB", bufs, big, 512, out);
    run<8, false>("stream 12.7MB", bufs, big, 512, out);
    run<6, true>("stream 12.7MB", bufs, big, 1024, out);
    run<2, true>("stream 12.7MB", bufs, big, 256, out);
    run<4, true>("stream 4.7MB", bufs, small, 256, out);
    run<2, true>("stream 4.7MB", bufs, small, 256, out);
    run<4, true>("stream 4.7MB", bufs, small, 512, out);
    run<2, true>("stream 4.7MB", bufs, small, 512, out);
    run<1, true>("stream 4.7MB", bufs, small, 256, out);
    return 0;
}

// ---- instruction-rate probes (v_perm_b32, v_mqsad_pk_u16_u8, sdwa byte add) --------------------------
template <int WHICH>
__global__ void k_rate(uint32_t* out, int iters) {
    __shared__ unsigned char lut[2048];
    if (WHICH == 5) { for (int i = threadIdx.x; i < 2048; i += blockDim.x) lut[i] = (unsigned char)(i * 37 + 11); __syncthreads(); }
    uint32_t a = threadIdx.x * 2654435761u, b = a ^ 0x9e3779b9u, c = a + 7, d = b + 11;
    unsigned long long q0 = a, q1 = b, q2 = c, q3 = d;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (WHICH == 0) {  // 4 independent v_perm_b32 chains
                a = __builtin_amdgcn_perm(a, b, 0x07060100u); b = __builtin_amdgcn_perm(b, c, 0x03020504u);
                c = __builtin_amdgcn_perm(c, d, 0x01000302u); d = __builtin_amdgcn_perm(d, a, 0x05040706u);
            } else if (WHICH == 1) {  // 4 independent v_mqsad_pk_u16_u8 chains
                q0 = __builtin_amdgcn_mqsad_pk_u16_u8(q1, 0xffu, q0); q1 = __builtin_amdgcn_mqsad_pk_u16_u8(q2, 0xffu, q1);
                q2 = __builtin_amdgcn_mqsad_pk_u16_u8(q3, 0xffu, q2); q3 = __builtin_amdgcn_mqsad_pk_u16_u8(q0, 0xffu, q3);
            } else if (WHICH == 2) {  // 4 independent v_add_u32 (baseline)
                a += b; b += c; c += d; d += a;
            } else if (WHICH == 3) {  // byte-select adds (sdwa)
                a += (b >> 8) & 0xff; b += (c >> 16) & 0xff; c += d >> 24; d += a & 0xff;
            } else if (WHICH == 4) {  // 4 independent ds_bpermute_b32 chains: one dword from a data-dependent lane per instruction
                a = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(a & 0xfcu), (int)b); b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(b & 0xfcu), (int)c);
                c = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(c & 0xfcu), (int)d); d = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(d & 0xfcu), (int)a);
            } else if (WHICH == 5) {  // 4 independent LDS byte gathers (ds_read_u8 at a data-dependent address in a 2 KB table)
                a = lut[a & 2047]; b = lut[(b + a) & 2047]; c = lut[(c + b) & 2047]; d = lut[(d + c) & 2047];
            } else if (WHICH == 6) {  // v_pk_fma_f32, 4 independent chains on 64-bit register pairs
                asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n\tv_pk_fma_f32 %1, %1, %4, %1\n\tv_pk_fma_f32 %2, %2, %4, %2\n\tv_pk_fma_f32 %3, %3, %4, %3"
                             : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(q0));
            } else if (WHICH == 7) {  // v_cvt_f32_i32
                asm volatile("v_cvt_f32_i32 %0, %0\n\tv_cvt_f32_i32 %1, %1\n\tv_cvt_f32_i32 %2, %2\n\tv_cvt_f32_i32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (WHICH == 8) {  // v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %1, %0\n\tv_fma_f32 %1, %1, %2, %1\n\tv_fma_f32 %2, %2, %3, %2\n\tv_fma_f32 %3, %3, %0, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (WHICH == 9) {  // v_pk_add_f32
                asm volatile("v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4"
                             : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(q0));
            } else if (WHICH == 10) { // v_mov_b64
                asm volatile("v_mov_b64 %0, %1\n\tv_mov_b64 %1, %2\n\tv_mov_b64 %2, %3\n\tv_mov_b64 %3, %0" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
            } else if (WHICH == 11) { // v_mov_b32
                asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else {                  // v_add_u32 through asm (the compiler cannot merge the chain)
                asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %1, %1, %2\n\tv_add_u32 %2, %2, %3\n\tv_add_u32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3);
}

template <int WHICH>
static void rate(const char* name, uint32_t* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000, blocks = 256 * 8, threads = 256;   // 8 waves per SIMD worth of work queued
    hipLaunchKernelGGL((k_rate<WHICH>), dim3(blocks), dim3(threads), 0, 0, out, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_rate<WHICH>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double insts = (double)blocks * (threads / 64) * iters * 16 * 4;   // wave-instructions
    printf("%-24s %.2f G wave-instr/s chip-wide -> %.2f cycles per wave-instr per SIMD at 2.4 GHz (1024 SIMDs)\n", name,
           insts / (ms * 1e-3) * 1e-9, 1024 * 2.4e9 / (insts / (ms * 1e-3)));
}

int rate_main() {
    uint32_t* out; CK(hipMalloc((void**)&out, 256 * 8 * 256 * 4));
    rate<2>("v_add_u32", out);
    rate<0>("v_perm_b32", out);
    rate<1>("v_mqsad_pk_u16_u8", out);
    rate<3>("v_add_u32_sdwa (byte)", out);
    rate<4>("ds_bpermute_b32", out);
    rate<5>("ds_read_u8 (LDS gather)", out);
    rate<12>("v_add_u32 (asm)", out);
    rate<11>("v_mov_b32", out);
    rate<10>("v_mov_b64", out);
    rate<7>("v_cvt_f32_i32", out);
    rate<8>("v_fma_f32", out);
    rate<6>("v_pk_fma_f32", out);
    rate<9>("v_pk_add_f32", out);
    printf("lookups per wave-instruction: v_perm_b32 on a half table 4 (x64 lanes) at 11 VALU per 8 lookups incl. index and sign handling; "
           "ds_bpermute_b32 1 (a dword from one lane); ds_read_u8 1\n");
    return 0;
}
