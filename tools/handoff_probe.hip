// handoff_probe.hip — what does ONE in-kernel hand-off between two workgroups cost on an otherwise idle MI355X?  (a probe, not product code)
//
// k_decode_chain hands a call's outputs to the next call through self-tagged 8-byte granules: agent-scope write-through stores (sc1),
// agent-scope L1-bypassing loads that spin until the tag is current.  The stamps of the running chain put a successful
// poll at ~1 us beyond the weight fragment queued in front of it.  This probe measures the floor of that number: two workgroups play
// ping-pong through two granules -- A stores tag i and spins on B's granule, B spins on A's and answers -- 2000 round trips, one lane each,
// s_memrealtime (100 MHz) around the loop; half a round trip = one hand-off.  Variants: the two workgroups on DIFFERENT XCDs (workgroup
// ids 0 and 1: consecutive ids go round the 8 XCDs) or on the SAME XCD (ids 0 and 8); agent scope (sc1) or system scope (sc0 sc1);
// ordinary (coarse-grained) device memory or fine-grained device memory (what the row-sharded chain's arena uses).
//   hipcc --offload-arch=gfx950 -O3 -o tools/handoff_probe tools/handoff_probe.hip && ./tools/handoff_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <bool SYS>
__device__ __forceinline__ void put(unsigned long long* p, unsigned long long v) {
    if (SYS) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool SYS>
__device__ __forceinline__ unsigned long long get(unsigned long long* p) {
    return SYS ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// buf[0]: A's granule, buf[32]: B's granule (256 bytes apart); a / b: the two workgroup ids; out[0] = ticks of the whole loop, out[1] = give-ups
template <bool SYS>
__global__ void k_pingpong(unsigned long long* buf, int iters, int a, int b, unsigned long long base, unsigned long long* out) {
    if (threadIdx.x != 0) return;
    const int me = (int)blockIdx.x;
    if (me != a && me != b) return;
    unsigned long long* mine = buf + (me == a ? 0 : 32);
    unsigned long long* theirs = buf + (me == a ? 32 : 0);
    unsigned long long t0 = 0, fails = 0;
    if (me == a) t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 1; i <= iters; ++i) {
        const unsigned long long tag = base + (unsigned long long)i;
        if (me == a) put<SYS>(mine, tag);
        unsigned spins = 0;
        while (get<SYS>(theirs) != tag) { if (++spins > (1u << 20)) { ++fails; break; } }
        if (fails) break;                                   // (the partner then runs into its own limit once and leaves as well)
        if (me == b) put<SYS>(mine, tag);
    }
    if (me == a) { out[0] = __builtin_amdgcn_s_memrealtime() - t0; out[1] = fails; }
}

template <bool SYS>
static void run(const char* mem, unsigned long long* buf, unsigned long long* out, int a, int b, unsigned long long base) {
    const int iters = 2000;
    CK(hipMemset(out, 0, 16));
    hipLaunchKernelGGL((k_pingpong<SYS>), dim3(16), dim3(64), 0, 0, buf, iters, a, b, base, out);
    CK(hipDeviceSynchronize());
    unsigned long long h[2];
    CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
    printf("%-14s %-18s workgroups %d <-> %-2d (%s): %.3f us per hand-off (half a round trip)%s\n", mem, SYS ? "system (sc0 sc1)" : "agent (sc1)", a, b,
           (a & 7) == (b & 7) ? "same XCD" : "other XCD", (double)h[0] * 0.01 / iters / 2.0, h[1] ? "  [SPIN LIMIT HIT]" : "");
}

int main() {
    unsigned long long *coarse, *fine, *out;
    CK(hipMalloc((void**)&coarse, 4096));
    CK(hipExtMallocWithFlags((void**)&fine, 4096, hipDeviceMallocFinegrained));
    CK(hipMalloc((void**)&out, 64));
    CK(hipMemset(coarse, 0, 4096)); CK(hipMemset(fine, 0, 4096));
    CK(hipDeviceSynchronize());
    unsigned long long base = 1000;
    for (int rep = 0; rep < 2; ++rep) {
        run<false>("coarse-grained", coarse, out, 0, 1, base); base += 10000;
        run<true>("coarse-grained", coarse, out, 0, 1, base); base += 10000;
        run<false>("fine-grained", fine, out, 0, 1, base); base += 10000;
        run<true>("fine-grained", fine, out, 0, 1, base); base += 10000;
        run<false>("coarse-grained", coarse, out, 0, 8, base); base += 10000;
        run<false>("fine-grained", fine, out, 0, 8, base); base += 10000;
        run<false>("coarse-grained", coarse, out, 0, 4, base); base += 10000;
    }
    return 0;
}
