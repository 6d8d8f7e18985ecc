import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tmac_amd
L = tmac_amd.lib()
x = np.concatenate([np.arange(64, dtype=np.uint32), 1000 + np.arange(64, dtype=np.uint32)])
out = np.zeros((64, 4), np.uint32)
L.tmac_hip_selftest_permlane.argtypes = [C.c_void_p, C.c_void_p]
print("rc", L.tmac_hip_selftest_permlane(x.ctypes.data, out.ctypes.data))
for name, col in [("p16.first", 0), ("p16.second", 1), ("p32.first", 2), ("p32.second", 3)]:
    print(name, out[:, col].reshape(4, 16)[:, :3].tolist(), "(first 3 lanes of each 16-lane row)")
