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

int ref_max_threads(void) { return omp_get_max_threads(); }
