/* ref_driver.c — TEST INFRASTRUCTURE: runs a reference-style per-tile kernel over all M-tiles with an
 * OpenMP static schedule, i.e. the way llama.cpp splits T-MAC's tiles over its threads
 * (include/t-mac/tmac_gemm_wrapper.h:197-199 "split the blocks in llama.cpp and pass the right ptr").
 * Used only by bench.py's cpu_baseline leg to time oracle/_ref (or the scalar port) on the host cores. */
#include <omp.h>
#include <stddef.h>
#include <stdint.h>

typedef int32_t (*qgemm_tile_fn)(void* A, void* LUT, void* Scales, void* LUT_Scales, void* LUT_Biases, void* C);

int ref_run_tiles_omp(qgemm_tile_fn fn, uint8_t* A, size_t a_stride, void* LUT, float* S, size_t s_stride, void* LS,
                      void* LB, float* C, size_t c_stride, int ntiles, int nthreads) {
    int rc = 0;
#pragma omp parallel for schedule(static) num_threads(nthreads) reduction(| : rc)
    for (int t = 0; t < ntiles; ++t)
        rc |= fn(A + (size_t)t * a_stride, LUT, S + (size_t)t * s_stride, LS, LB, C + (size_t)t * c_stride);
    return rc;
}

/* N > 1 as the reference computes it: its N = 1 kernel looped over the activation rows (python/t_mac/ops/qgemm.py:183-190,228-231: per-row
 * LUT / LUT_Scales / LUT_Biases / C).  Rows inside the tile loop: a tile's weights stay in cache for all rows, the most favourable
 * order for the CPU.  LUT rows lut_stride bytes apart, LS / LB rows g floats apart, C rows Mw floats apart. */
int ref_run_tiles_rows_omp(qgemm_tile_fn fn, uint8_t* A, size_t a_stride, uint8_t* LUT, size_t lut_stride, float* S, size_t s_stride, float* LS,
                           float* LB, size_t g, float* C, size_t c_stride, size_t Mw, int ntiles, int nrows, int nthreads) {
    int rc = 0;
#pragma omp parallel for schedule(static) num_threads(nthreads) reduction(| : rc)
    for (int t = 0; t < ntiles; ++t)
        for (int n = 0; n < nrows; ++n)
            rc |= fn(A + (size_t)t * a_stride, LUT + (size_t)n * lut_stride, S + (size_t)t * s_stride, LS + (size_t)n * g, LB + (size_t)n * g,
                     C + (size_t)n * Mw + (size_t)t * c_stride);
    return rc;
}

/* The int32 / scale-final path (BitNet on x86: the reference selects tbl_g4_int8_int32_update there, tools/run_pipeline.py:409-412).
 * fn = ref_tile_cbits_int32 of oracle/_ref/libtmac_ref_intrins.so (the reference's own intrinsic in the generated glue's k_outer loop);
 * the bit-plane combine and the three float operations of scale-final are the glue of python/t_mac/ops/qgemm.py:170-174,192-206,
 * restated here (the reference has no compiled x86 exemplar of it). */
typedef int32_t (*cbits_i32_fn)(int bits, int kfactor, int bm, int K, void* A_tile, void* LUT, int32_t* CBits);

int ref_run_tiles_int32_omp(cbits_i32_fn fn, int bits, int kfactor, int bm, int K, uint8_t* A, size_t a_stride, void* LUT,
                            float lut_scale, float lut_bias, float scale, float* C, int ntiles, int nthreads) {
    int rc = 0;
    if (bm > 1024 || bits < 1 || bits > 4) return -1;
#pragma omp parallel for schedule(static) num_threads(nthreads) reduction(| : rc)
    for (int t = 0; t < ntiles; ++t) {
        int32_t cb[1024] __attribute__((aligned(32)));
        rc |= fn(bits, kfactor, bm, K, A + (size_t)t * a_stride, LUT, cb);
        const int rows = bm / bits;
        for (int m = 0; m < rows; ++m) {
            float acc = 0.f;
            for (int b = 0; b < bits; ++b) {
                const float alpha = b == 0 ? 0.5f : (b == 1 ? 1.0f : (b == 2 ? 2.0f : 4.0f));
                const float term = (float)cb[(m / 8) * 8 * bits + b * 8 + m % 8] * alpha;
                acc = b == 0 ? term : acc + term;
            }
            C[(size_t)t * rows + m] = (acc * lut_scale + lut_bias * 0.5f) * scale;
        }
    }
    return rc;
}

int ref_max_threads(void) { return omp_get_max_threads(); }
