"""Crossover between the GEMV row loop and k_gemm_planes for few activation rows (llama-2-7B shapes, W2 / W4): LUT build + GEMM
through the fused entry point, hipGraph replay.  usage: bench_small_n.py [bits]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tmac_amd
from tmac_amd import KCfg, F16
L = tmac_amd.lib()
dev = torch.device("cuda")
BITS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
BM = {1: 64, 2: 128, 3: 192, 4: 256}[BITS]
wr = tmac_amd.TMACGeMMWrapper(act_group_size=64)


def timeit(fn, reps=20):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


print("bits =", BITS, " us per call (LUT build + GEMM): rows | row loop | k_gemm_planes | default choice")
for name, Mw, K, nshare in [("o", 4096, 4096, 1), ("qkv", 4096, 4096, 3), ("gate_up", 11008, 4096, 2), ("down", 4096, 11008, 1)]:
    ws = []
    for _ in range(nshare):
        A = torch.randint(0, 256, (Mw * BITS // BM, K // 4, BM // 2), dtype=torch.uint8, device=dev)
        S = (torch.randn((Mw * BITS // BM, K // 128, BM // BITS // 8, 2, 8), device=dev) * 0.01).half().contiguous()
        ws.append(tmac_amd.Weights(A, S, Mw, K, BITS, KCfg.make(Mw, K, BITS, BM), scales_dtype=F16, dev_dtype=F16, on_device=True))
    line = []
    for N in (2, 3, 4, 6, 8, 10, 12, 16):
        outs = [torch.empty(N, Mw, dtype=torch.float16, device=dev) for _ in range(nshare)]
        x = torch.randn(N, K, device=dev).half()
        L.tmac_hip_set_gemm_min_n(0)
        t_loop = timeit(lambda: wr.fused(ws, x, outs, N))
        L.tmac_hip_set_gemm_min_n(1)
        t_gemm = timeit(lambda: wr.fused(ws, x, outs, N))
        L.tmac_hip_set_gemm_min_n(32)
        t_auto = timeit(lambda: wr.fused(ws, x, outs, N))
        line.append(f"{N}: {t_loop:.1f} | {t_gemm:.1f} | {t_auto:.1f}")
    L.tmac_hip_set_gemm_min_n(32)
    print(f"{name:8s} " + "   ".join(line))
    for w in ws:
        w.free()
