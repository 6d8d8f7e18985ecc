"""Checks the operand layout model of v_mfma_i32_16x16x64_i8 used by the MFMA-accumulate path."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tmac_amd
L = tmac_amd.lib()
rng = np.random.default_rng(0)
x = rng.integers(0, 2**32, size=(64, 8), dtype=np.uint64).astype(np.uint32)
out = np.zeros((64, 4), np.int32)
tmac_amd.binding.check(L.tmac_hip_selftest_mfma(x.ctypes.data, out.ctypes.data))
A = x[:, :4].copy().view(np.int8).reshape(64, 16).astype(np.int64)   # lane, 16 bytes (reg-major, byte-minor)
B = x[:, 4:].copy().view(np.int8).reshape(64, 16).astype(np.int64)
# model: C[i][j] = sum_g sum_{16 bytes} A[lane 16g+i][.] * B[lane 16g+j][.];  D lane l holds col l%16, rows 4*(l//16)+r
C = np.zeros((16, 16), np.int64)
for i in range(16):
    for j in range(16):
        C[i, j] = sum(int((A[16 * g + i] * B[16 * g + j]).sum()) for g in range(4))
got = np.zeros((16, 16), np.int64)
for l in range(64):
    for r in range(4):
        got[4 * (l // 16) + r, l % 16] = out[l, r]
print("model matches hardware:", np.array_equal(C, got))
if not np.array_equal(C, got):
    print("transposed?", np.array_equal(C.T, got))
    print(C[:4, :4]); print(got[:4, :4])
