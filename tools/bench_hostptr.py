#!/usr/bin/env python3
"""PCIe-inclusive rate of the zero-change route: the reference-named host-pointer entry points
preprocessor_int8 / qgemm_lut_int8 called exactly as llama.cpp calls the reference -- one preprocessor call and one
qgemm call per 128-row M-tile, all pointers in host memory.  Tile weights are uploaded on first use and cached by
pointer; every call still stages the LUT to the device and the tile's outputs back and synchronises.
usage: python tools/bench_hostptr.py"""
import ctypes as C
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tmac_amd  # noqa: E402

L = tmac_amd.lib()
rng = np.random.default_rng(0)
for Mw, K in [(4096, 4096), (4096, 11008)]:
    bits, bm = 2, 128
    ntile = Mw * bits // bm
    with tempfile.TemporaryDirectory() as td:
        ini = os.path.join(td, "kcfg.ini")
        open(ini, "w").write(f"[qgemm_lut_t1_int8_m{Mw * bits}_k{K}_n1_b2]\nbm = 128\nsimd_n_in = 16\nsimd_n_out = 8\nkfactor = 16\n"
                             f"group_size = 128\nlut_scales_size = {K // 64}\nscales_size = {Mw * K // 128 * 2}\nn_tile_num = {ntile}\n")
        tmac_amd.binding.check(L.tmac_hip_load_kcfg(ini.encode()))
    A = rng.integers(0, 256, size=(ntile, K // 4 * bm // 2), dtype=np.uint8)
    S = np.abs(rng.standard_normal((ntile, K // 128 * 64 * 2))).astype(np.float32)
    B = rng.standard_normal(K).astype(np.float32)
    ls = np.zeros(K // 64, np.float32); lb = np.zeros(K // 64, np.float32); q = np.zeros((K // 4, 16), np.int8)
    out = np.zeros(Mw, np.float32)
    p = lambda a: C.c_void_p(a.ctypes.data)

    # pointer objects made once: numpy's .ctypes.data costs more than a served tile call
    pB, pls, plb, pq = p(B), p(ls), p(lb), p(q)
    pA = [p(A[t]) for t in range(ntile)]; pS = [p(S[t]) for t in range(ntile)]
    pC = [C.c_void_p(out.ctypes.data + 4 * 64 * t) for t in range(ntile)]

    def gemv():
        assert L.preprocessor_int8(Mw * bits, K, 1, bits, pB, pls, plb, pq) == 0
        for t in range(ntile):
            assert L.qgemm_lut_int8(bm, K, 1, bits, pA[t], pq, pS[t], pls, plb, pC[t]) == 0

    L.tmac_hip_debug_host_runs.argtypes = [C.c_int]
    for runs in (0, 1):
        L.tmac_hip_debug_host_runs(runs)
        L.tmac_hip_cache_clear()
        gemv(); gemv()              # uploads and caches the tiles; the second pass groups them into a run
        reps = 5
        t0 = time.perf_counter()
        for r in range(reps):
            B[:] = rng.standard_normal(K).astype(np.float32)     # a new activation vector per GEMV, as in a model
            gemv()
        dt = (time.perf_counter() - t0) / reps
        nbytes = Mw * K * bits // 8 + S.size * 2
        print(f"host-pointer route {Mw}x{K} W2, {'whole runs' if runs else 'tile by tile'}: {dt * 1e6:9.1f} us per GEMV ({ntile} tile calls, "
              f"{dt * 1e6 / ntile:6.1f} us each) = {nbytes / dt / 1e9:6.2f} GB/s of weight bytes, PCIe staging and synchronisation included")
L.tmac_hip_cache_clear()
