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

int main() {
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
