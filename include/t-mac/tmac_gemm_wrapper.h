// t-mac/tmac_gemm_wrapper.h — source-compatible replacement of the reference's runtime wrapper
// (include/t-mac/tmac_gemm_wrapper.h:79-347, no-TVM branch) on top of libtmac_hip.so.
//
// A llama.cpp build configured with -DGGML_TMAC=ON includes "t-mac/tmac_gemm_wrapper.h" and calls
//   TMAC::TMACGeMMWrapper<T>::{set_workspace, get_kcfg, llama_cpp_init, llama_cpp_compute}
// with HOST pointers (tools/run_pipeline.py:181-188).  Pointing its include path at this directory and
// linking libtmac_hip.so keeps those call sites unchanged: llama_cpp_init -> preprocessor_int8,
// llama_cpp_compute -> qgemm_lut_int8 (one M-tile per call, as the reference), both executed on the GPU.
// T must be float on x86 hosts (the reference's float_type, python/t_mac/intrins/tbl.cc:10-15).
#pragma once

#include <cstdio>
#include <cstdlib>
#include <string>
#include <type_traits>

#include "../tmac_hip.h"

namespace TMAC {

constexpr size_t kAllocAlignment = 64;

struct TMACGeMMConfig {
  int bm;
  int simd_n_in;
  int simd_n_out;
  int kfactor;
  int group_size;
  int lut_scales_size;
  int scales_size;
  int n_tile_num;
};

template <typename T, int g = 4>
class TMACGeMMWrapper {
  static_assert(std::is_same<T, float>::value, "libtmac_hip's host-pointer ABI uses the x86 float_type (fp32)");
  static_assert(g == 4, "T-MAC LUT group size is 4");

public:
  TMACGeMMWrapper(int n_threads, int act_group_size, const std::string& kcfg_file, const std::string& /*library_file*/)
      : _act_group_size(act_group_size), _allocated(false), _qlut(nullptr), _lut_scales(nullptr), _lut_biases(nullptr) {
    (void)n_threads;
    // kcfg path: argument, else $TMAC_KCFG_FILE, else the TMAC_KCFG_FILE compile definition the CMake package sets
    // (reference: tmac_gemm_wrapper.h:40-56)
    const char* path = kcfg_file.empty() ? nullptr : kcfg_file.c_str();
#ifdef TMAC_KCFG_FILE
#define TMAC_STR2_(x) #x
#define TMAC_STR_(x) TMAC_STR2_(x)
    if (!path && !std::getenv("TMAC_KCFG_FILE")) path = TMAC_STR_(TMAC_KCFG_FILE);
#undef TMAC_STR_
#undef TMAC_STR2_
#endif
    if (tmac_hip_load_kcfg(path) != 0) {
      std::fprintf(stderr, "TMAC: %s\n", tmac_hip_last_error());
      std::abort();  // reference: LOG(FATAL) << "Please set TMAC_KCFG_FILE environment variable"
    }
  }
  TMACGeMMWrapper() : TMACGeMMWrapper(1, 32, "", "") {}

  void set_num_threads(int) {}  // no TVM threadpool

  // Activation (B): NxK.  Main thread only, as in the reference.
  void llama_cpp_init(void* B, void* qlut, void* lut_scales, void* lut_biases, int M, int K, int N, int bits) {
    int ret = preprocessor_int8(M * bits, K, N, bits, B, lut_scales, lut_biases, qlut);
    if (ret != 0) std::fprintf(stderr, "error calling preprocessor (m=%d, k=%d, n=%d, b=%d): %s\n", M, K, N, bits, tmac_hip_last_error());
  }

  // One M-tile: M = bm / bits, A / scales / C are the tile pointers (tmac_gemm_wrapper.h:197-228).
  void llama_cpp_compute(void* A, void* scales, void* qlut, void* lut_scales, void* lut_biases, void* C, int M, int K, int N, int bits) {
    int ret = qgemm_lut_int8(M * bits, K, N, bits, A, qlut, scales, lut_scales, lut_biases, C);
    if (ret != 0) std::fprintf(stderr, "error calling qgemm_lut (m=%d, k=%d, n=%d, b=%d): %s\n", M, K, N, bits, tmac_hip_last_error());
  }

  TMACGeMMConfig get_kcfg(int M, int K, int N, int bits) {
    tmac_kcfg c{};
    tmac_hip_get_kcfg(M, K, N, bits, &c);
    return {c.bm, c.simd_n_in, c.simd_n_out, c.kfactor, c.group_size, c.lut_scales_size, c.scales_size, c.n_tile_num};
  }

  // Main thread only.  Host staging buffers with the reference's sizes and alignment (:257-270).
  void set_workspace(int maxK, int maxN) {
    posix_memalign(&_qlut, kAllocAlignment, (size_t)maxN * maxK / g * (1 << g) * sizeof(int8_t));
    posix_memalign(&_lut_scales, kAllocAlignment, (size_t)maxN * maxK / _act_group_size * sizeof(T));
    posix_memalign(&_lut_biases, kAllocAlignment, (size_t)maxN * maxK / _act_group_size * sizeof(T));
    _allocated = true;
  }

  ~TMACGeMMWrapper() {
    if (_allocated) { free(_qlut); free(_lut_scales); free(_lut_biases); }
  }

private:
  int _act_group_size;
  bool _allocated;
  void* _qlut;
  void* _lut_scales;
  void* _lut_biases;
};

}  // namespace TMAC
