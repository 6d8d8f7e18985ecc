"""Phase timeline of the fused kernel from s_memtime stamps (debug tool; run on the GPU box)."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tmac_amd
from tmac_amd import KCfg, F16
L = tmac_amd.lib()
L.tmac_hip_debug_stamps.argtypes = [C.c_void_p]
dev = torch.device("cuda")
wr = tmac_amd.TMACGeMMWrapper(act_group_size=64); wr.set_workspace(11008, 1)
for name, Mw, K, cnt in [("o", 4096, 4096, 1), ("qkv", 4096, 4096, 3), ("gate_up", 11008, 4096, 2), ("down", 4096, 11008, 1)]:
    ws = []
    for _ in range(cnt):
        A = torch.randint(0, 256, (Mw * 2 // 128, K // 4, 64), dtype=torch.uint8, device=dev)
        S = (torch.randn((Mw * 2 // 128, K // 128, 8, 2, 8), device=dev) * 0.01).half().contiguous()
        ws.append(tmac_amd.Weights(A, S, Mw, K, 2, KCfg.make(Mw, K, 2, 128), scales_dtype=F16, dev_dtype=F16, on_device=True))
    x = torch.randn(K, device=dev).half()
    outs = [torch.empty(Mw, dtype=torch.float16, device=dev) for _ in range(cnt)]
    nb = 1024
    st = torch.zeros((nb, 8), dtype=torch.int64, device=dev)
    for _ in range(3): wr.fused(ws, x, outs, 1)
    torch.cuda.synchronize()
    L.tmac_hip_debug_stamps(st.data_ptr())
    wr.fused(ws, x, outs, 1)
    torch.cuda.synchronize()
    L.tmac_hip_debug_stamps(None)
    s = st.cpu().numpy().astype(np.float64)
    s = s[s[:, 0] > 0]
    nb = len(s)
    t0 = s[:, 0].min()
    rel = (s[:, :7] - t0)
    d = np.diff(s[:, :7], axis=1)
    print(f"== {name} Mw={Mw}x{cnt} K={K}: blocks={nb}; stamps in s_memtime ticks")
    print("   kernel span (last end - first start):", rel[:, 6].max())
    print("   block start spread: min/median/max", rel[:, 0].min(), np.median(rel[:, 0]), rel[:, 0].max())
    print("   phase medians [issue loads, build LUT, barrier, first quad lookups, first quad reduce+store, remaining quads]:", np.median(d, axis=0))
    print("   phase p90:", np.percentile(d, 90, axis=0))
    print("   kernel entry -> first kernel-argument value available (ticks), median/p90:", np.median(s[:, 0] - s[:, 7]), np.percentile(s[:, 0] - s[:, 7], 90))
    print("   workgroup duration median/max:", np.median(rel[:, 6] - rel[:, 0]), (rel[:, 6] - rel[:, 0]).max())
    for w in ws: w.free()
