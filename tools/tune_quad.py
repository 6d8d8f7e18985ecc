"""Times every (threads per workgroup, waves per quad) configuration of k_gemv_quad on the llama-2-7B W2 shapes.
Back-to-back launches over distinct weight sets (HBM-cold), hipEvent timing; prints us per launch."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tmac_amd
from tmac_amd import KCfg, F16
L = tmac_amd.lib()
L.tmac_hip_debug_quad_config.argtypes = [C.c_int, C.c_int]
dev = torch.device("cuda")
wr = tmac_amd.TMACGeMMWrapper(act_group_size=64); wr.set_workspace(11008, 1)
NSET = 12
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
BITS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
BM = {1: 128, 2: 128, 3: 192, 4: 256}[BITS]
L.tmac_hip_set_variant(variant)
for name, Mw, K, cnt in [("o", 4096, 4096, 1), ("qkv", 4096, 4096, 3), ("gate_up", 11008, 4096, 2), ("down", 4096, 11008, 1)]:
    sets = []
    for _ in range(NSET):
        ws = []
        for _ in range(cnt):
            A = torch.randint(0, 256, (Mw * BITS // BM, K // 4, BM // 2), dtype=torch.uint8, device=dev)
            S = (torch.randn((Mw * BITS // BM, K // 128, BM // BITS // 8, 2, 8), device=dev) * 0.01).half().contiguous()
            ws.append(tmac_amd.Weights(A, S, Mw, K, BITS, KCfg.make(Mw, K, BITS, BM), scales_dtype=F16, dev_dtype=F16, on_device=True))
        sets.append(ws)
    x = torch.randn(K, device=dev).half()
    outs = [torch.empty(Mw, dtype=torch.float16, device=dev) for _ in range(cnt)]
    res = {}
    for ft, wpq in [(0, 0), (512, 1), (512, 2), (768, 1), (768, 2), (768, 3), (1024, 1), (1024, 2), (1024, 4)]:
        L.tmac_hip_debug_quad_config(ft, wpq)
        try:
            for ws in sets[:2]: wr.fused(ws, x, outs, 1)
        except Exception as e:
            res[(ft, wpq)] = None; continue
        torch.cuda.synchronize()
        # capture the NSET launches into a hipGraph so the host launch path (~9 us per ctypes call) is out of the timing
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for ws in sets: wr.fused(ws, x, outs, 1)
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for ws in sets: wr.fused(ws, x, outs, 1)
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for rep in range(6):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / NSET)
        res[(ft, wpq)] = best
        del g
    L.tmac_hip_debug_quad_config(0, 0)
    print(f"{name:8s} Mw={Mw}x{cnt} K={K}: " + "  ".join(f"({ft},{wpq})={'n/a' if v is None else f'{v:.2f}'}" for (ft, wpq), v in res.items()))
    for ws in sets:
        for w in ws: w.free()
