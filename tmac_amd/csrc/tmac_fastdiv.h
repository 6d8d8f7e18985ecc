// tmac_fastdiv.h — short, correctly rounded replacements for the two divisions of the LUT build
// (lut_ctor.cc:56-60: scales = absmax / 127, t_scales = scales ? 1 / scales : 0).  hipcc expands __fdiv_rn into
// the 13-instruction v_div_scale / v_div_fmas / v_div_fixup sequence; these are 3 and 9 instructions.
// Exactness (bit-equal to IEEE division for every fp32 input) is checked exhaustively: div127 on the CPU and the
// GPU, rcp_exact on the GPU against __fdiv_rn (tools/divcheck.hip, result in profiles/).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tmac {

// x / 127, correctly rounded (Markstein: y = RN(1/127); 127 is not an all-ones significand)
__device__ __forceinline__ float div127(float x) {
    const float y = 1.0f / 127.0f;                   // constant-folded: RN(1/127) = 0x1.020408p-7
    const float q0 = __fmul_rn(x, y);
    const float r = __fmaf_rn(-127.0f, q0, x);
    return __fmaf_rn(r, y, q0);
}

// 1 / s, correctly rounded, for s in [2^-100, 2^100]; outside (and never for finite activations of sane magnitude)
// the IEEE sequence is used
__device__ __forceinline__ float rcp_exact(float s) {
    const uint32_t sb = __float_as_uint(s);
    if (sb - 0x0d800000u > 0x64000000u) return __fdiv_rn(1.0f, s);   // biased exponent outside [27, 227]
    const float y0 = __builtin_amdgcn_rcpf(s);
    const float e = __fmaf_rn(-s, y0, 1.0f);
    const float y1 = __fmaf_rn(e, y0, y0);
    const float r0 = __fmaf_rn(-s, y1, 1.0f);
    float q = __fmaf_rn(r0, y1, y1);
    // all-ones significand: 1/s sits just above a rounding midpoint the residual iteration cannot see;
    // RN(1/((2 - 2^-23) 2^k)) = (1 + 2^-23) 2^(-k-1), whose bit pattern is 0x7F000000 - bits(s)
    if ((sb & 0x007fffffu) == 0x007fffffu) q = __uint_as_float(0x7F000000u - sb);
    return q;
}

}  // namespace tmac
