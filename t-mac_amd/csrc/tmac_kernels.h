// tmac_kernels.h — host-callable launchers of the gfx950 kernels (implemented in tmac_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tmac_layout.h"

namespace tmac {

enum Dtype { F32 = 0, F16 = 1 };

// GEMV kernel variants (A/B-able at run time through tmac_hip_set_variant)
enum Variant {
    V_AUTO = 0,
    V_LO_MQSAD = 1,  // tiled kernel, v_mqsad_pk_u16_u8 accumulate
    V_LO_SDWA = 2,   // tiled kernel, byte-select adds
    V_REF_LAYOUT = 3 // generic kernel on the reference blobs
};

struct GemvArgs {
    Shape s;
    const void* W;        // device layout weights (uint4)        | reference blob for V_REF_LAYOUT
    const void* SC;       // device layout scales (sc_dtype)      | reference blob (float_type = sc_dtype)
    Dtype sc_dtype;
    const void* qlut_dev; // uint4 [N][qlut_dev_u4]
    const int8_t* qlut_ref; // int8 [N][K/4][16] (V_REF_LAYOUT only)
    const float* lut_scales;
    const float* lut_biases;
    void* C;              // [N][Mw]
    Dtype out_dtype;
    int32_t* ps_dump;     // optional int32 tap (device)
    int N;
};

hipError_t launch_selftest(const uint32_t* in, uint32_t* out, int n, hipStream_t st);
hipError_t launch_retile_weights(const uint8_t* A_ref, void* Wd, const Shape& s, hipStream_t st);
hipError_t launch_retile_scales(const void* S_ref, Dtype in_dt, void* Sd, Dtype out_dt, const Shape& s, hipStream_t st);
hipError_t launch_preprocess(const void* B, Dtype act_dt, int8_t* qlut_ref, void* qlut_dev, float* lut_scales,
                             float* lut_biases, int K, int N, int ags, size_t qdev_u4_per_row, hipStream_t st);
hipError_t launch_qlut_ref_to_dev(const int8_t* qlut_ref, void* qlut_dev, int K, int N, size_t qdev_u4_per_row, hipStream_t st);
// returns hipErrorInvalidValue when the variant does not cover the configuration
hipError_t launch_gemv(const GemvArgs& a, Variant v, hipStream_t st);
bool gemv_lo_supported(const Shape& s);

}  // namespace tmac
