// divcheck.hip — exhaustive check (all 2^31 non-negative fp32 bit patterns) that the short division sequences
// used by the in-kernel LUT build equal the correctly rounded IEEE results the reference computes on the CPU
// (lut_ctor.cc: scales = absmax / 127, t_scales = 1 / scales):
//   div127(mx)  = Markstein: q0 = mx*y, r = fma(-127, q0, mx), q = fma(r, y, q0),  y = RN(1/127)
//   rcp_exact(s) = v_rcp_f32 + two Newton/residual steps + the all-ones-mantissa special case
// Prints mismatch counts per biased exponent.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off divcheck.hip -o divcheck
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../t-mac_amd/csrc/tmac_fastdiv.h"

__global__ void k_check(unsigned long long* bad127, unsigned long long* badrcp) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < 0x7f800000ull; b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const float r1 = __fdiv_rn(x, 127.0f), f1 = tmac::div127(x);
        if (__float_as_uint(r1) != __float_as_uint(f1)) atomicAdd(&bad127[b >> 23], 1ull);
        if (b != 0) {
            const float r2 = __fdiv_rn(1.0f, x), f2 = tmac::rcp_exact(x);
            if (__float_as_uint(r2) != __float_as_uint(f2)) atomicAdd(&badrcp[b >> 23], 1ull);
        }
    }
}

int main() {
    unsigned long long *d, h[512];
    if (hipMalloc((void**)&d, sizeof(h)) != hipSuccess) { printf("no device\n"); return 1; }
    hipMemset(d, 0, sizeof(h));
    hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, d, d + 256);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long t1 = 0, t2 = 0;
    for (int e = 0; e < 256; ++e) {
        t1 += h[e]; t2 += h[256 + e];
        if (h[e] || h[256 + e]) printf("exp %3d: div127 mismatches %llu, rcp mismatches %llu\n", e, h[e], h[256 + e]);
    }
    printf("TOTAL div127 mismatches %llu, rcp_exact mismatches %llu (of 2^31 inputs each)\n", t1, t2);
    return 0;
}
