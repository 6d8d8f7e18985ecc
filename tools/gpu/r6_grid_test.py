import torch
torch.cuda.init()
import sys, os
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
import tmac_amd as tm
from test_gpu_chain import Model
from test_gpu_stream import INDEP, US_OPS
L = tm.lib()
for grid in (1, 3, 7, 13, 250):
    for qw in ("0", "1"):
        os.environ["TMAC_STREAM_QW"] = qw
        tm.binding.check(L.tmac_hip_reset_state())
        tm.binding.check(L.tmac_hip_debug_chain_grid(grid))
        for ops, kw in ((INDEP, {}), (US_OPS, dict(mg=1, zp=False))):
            m = Model(tm, ops, seed=5, **kw)
            ch = m.record(); assert ch.stream
            ch.launch(); m.check(ch); m.check_tap(ch)
            ch.free(); m.free()
        tm.binding.check(L.tmac_hip_debug_chain_grid(0))
        print("grid", grid, "qw", qw, "ok", flush=True)
