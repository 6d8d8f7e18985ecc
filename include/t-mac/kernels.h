// kernels.h — stand-in for the header T-MAC's code generator emits (deploy/compile.py:52-71,196-199; e.g.
// deploy/tuned/aarch64-llama-2-7b-2bit/kernels.h): the two dispatchers the rest of T-MAC and the llama.cpp fork call,
// and the shape-named kernels of the four checked-in tuned sets.  Here they are exported by libtmac_hip.so (the
// kernels are HIP code inside the library, not inline CPU code), with the reference's signatures and return
// convention: 0 = ok, -1 = no kernel for that shape.
#pragma once
#include "../tmac_hip.h"   // qgemm_lut_int8, preprocessor_int8 and the shape-named symbols

#ifdef __cplusplus
extern "C" {
#endif
#define TMAC_DECL_Q(bm, k, n, b) int32_t qgemm_lut_t1_int8_m##bm##_k##k##_n##n##_b##b(void* A, void* LUT, void* Scales, void* LUT_Scales, void* LUT_Biases, void* C);
#define TMAC_DECL_P(m, k, n, b) int32_t preprocessor_t1_int8_m##m##_k##k##_n##n##_b##b(void* B, void* LUT_Scales, void* LUT_Biases, void* QLUT);
/* aarch64-llama-2-7b-2bit */
TMAC_DECL_Q(128, 4096, 1, 2) TMAC_DECL_Q(128, 11008, 1, 2)
TMAC_DECL_P(8192, 4096, 1, 2) TMAC_DECL_P(22016, 4096, 1, 2) TMAC_DECL_P(8192, 11008, 1, 2)
/* aarch64-llama-2-7b-4bit */
TMAC_DECL_Q(1024, 4096, 1, 4) TMAC_DECL_Q(256, 4096, 1, 4) TMAC_DECL_Q(256, 11008, 1, 4)
TMAC_DECL_P(16384, 4096, 1, 4) TMAC_DECL_P(44032, 4096, 1, 4) TMAC_DECL_P(16384, 11008, 1, 4)
/* aarch64-llama-3-8b-2bit */
TMAC_DECL_Q(256, 4096, 1, 2) TMAC_DECL_Q(512, 4096, 1, 2) TMAC_DECL_Q(128, 14336, 1, 2)
TMAC_DECL_P(28672, 4096, 1, 2) TMAC_DECL_P(8192, 14336, 1, 2) TMAC_DECL_P(2048, 4096, 1, 2)
/* aarch64-hf-bitnet-3b */
TMAC_DECL_Q(128, 8640, 1, 2) TMAC_DECL_Q(128, 3200, 1, 2) TMAC_DECL_Q(320, 3200, 1, 2)
TMAC_DECL_P(6400, 8640, 1, 2) TMAC_DECL_P(17280, 3200, 1, 2) TMAC_DECL_P(6400, 3200, 1, 2)
#undef TMAC_DECL_Q
#undef TMAC_DECL_P
#ifdef __cplusplus
}
#endif
