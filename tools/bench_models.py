"""Per-launch GEMV time of the fused kernel on the other BASELINE model shapes (graph replay over distinct weight
sets, hipEvent): llama-2-7B W4 (config 3) and BitNet-b1.58-3B (config 4: 2-bit ternary, unified scale, act group = K).
Prints us per launch and GB/s of algorithmic bytes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tmac_amd
from tmac_amd import KCfg, F16, F32
dev = torch.device("cuda")
NSET = 12

def run(tag, shapes, bits, bm, gs, ags_of, zp, m_groups):
    tot_us = 0.0; tot_b = 0
    for name, Mw, K, cnt, per_layer in shapes:
        ags = ags_of(K)
        wr = tmac_amd.TMACGeMMWrapper(act_group_size=ags); wr.set_workspace(K, 1)
        cfg = KCfg.make(Mw, K, bits, bm, 16, gs, ags, zp, m_groups)
        sets = []
        for _ in range(NSET):
            ws = []
            for _ in range(cnt):
                A = torch.randint(0, 256, (Mw * bits // bm, K // 4, bm // 2), dtype=torch.uint8, device=dev)
                if m_groups >= 1:
                    S = torch.full((m_groups,), 0.01, device=dev, dtype=torch.float32)
                    ws.append(tmac_amd.Weights(A, S, Mw, K, bits, cfg, scales_dtype=F32, dev_dtype=F32, on_device=True))
                else:
                    S = (torch.randn((Mw * bits // bm, K // gs, bm // bits // 8, 2 if zp else 1, 8), device=dev) * 0.01).half().contiguous()
                    ws.append(tmac_amd.Weights(A, S, Mw, K, bits, cfg, scales_dtype=F16, dev_dtype=F16, on_device=True))
            sets.append(ws)
        x = torch.randn(K, device=dev).half()
        outs = [torch.empty(Mw, dtype=torch.float16, device=dev) for _ in range(cnt)]
        for ws in sets[:2]: wr.fused(ws, x, outs, 1)
        torch.cuda.synchronize()
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
        nsc = (m_groups * 4) if m_groups >= 1 else Mw * (K // gs) * (2 if zp else 1) * 2
        b = cnt * (Mw * K * bits // 8 + nsc + Mw * 2) + K // 4 * 16 + (K // ags) * 4
        print(f"{tag} {name:8s} Mw={Mw}x{cnt} K={K} bits={bits}: {best:7.2f} us/launch  {b / best * 1e-3:7.1f} GB/s")
        tot_us += per_layer * best; tot_b += per_layer * b
        for ws in sets:
            for w in ws: w.free()
    return tot_us, tot_b

t, b = run("llama-2-7b-W4", [("o", 4096, 4096, 1, 1), ("qkv", 4096, 4096, 3, 1), ("gate_up", 11008, 4096, 2, 1), ("down", 4096, 11008, 1, 1)],
           4, 256, 128, lambda K: 64, True, -1)
print(f"llama-2-7B W4A16-style decode: {32 * t * 1e-3:.3f} ms/token of GEMV launches, {b / t * 1e-3:.0f} GB/s")
t, b = run("bitnet-3b", [("o", 3200, 3200, 1, 1), ("qkv", 3200, 3200, 3, 1), ("gate_up", 8640, 3200, 2, 1), ("down", 3200, 8640, 1, 1)],
           2, 128, 128, lambda K: K, False, 1)
print(f"BitNet-b1.58-3B decode: {26 * t * 1e-3:.3f} ms/token of GEMV launches, {b / t * 1e-3:.0f} GB/s")

# ---- BitNet prefill: the unified-scale flavour of the one-hot GEMM against the GEMV row loop (fused entry point) ----
def bitnet_prefill(N):
    L = tmac_amd.lib()
    tot = {"gemm": 0.0, "loop": 0.0}
    for name, Mw, K, cnt in [("o", 3200, 3200, 1), ("qkv", 3200, 3200, 3), ("gate_up", 8640, 3200, 2), ("down", 3200, 8640, 1)]:
        wr = tmac_amd.TMACGeMMWrapper(act_group_size=K); wr.set_workspace(K, N)
        cfg = KCfg.make(Mw, K, 2, 128, 16, 128, K, False, 1, N)
        ws = []
        for _ in range(cnt):
            A = torch.randint(0, 256, (Mw * 2 // 128, K // 4, 64), dtype=torch.uint8, device=dev)
            S = torch.full((1,), 0.01, device=dev, dtype=torch.float32)
            ws.append(tmac_amd.Weights(A, S, Mw, K, 2, cfg, scales_dtype=F32, dev_dtype=F32, on_device=True))
        x = torch.randn(N, K, device=dev).half()
        outs = [torch.empty(N, Mw, dtype=torch.float16, device=dev) for _ in range(cnt)]
        res = {}
        for tag, mn in (("gemm", 16), ("loop", 0)):
            L.tmac_hip_set_gemm_min_n(mn)
            wr.fused(ws, x, outs, N); torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): wr.fused(ws, x, outs, N)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / 5)
            res[tag] = best; tot[tag] += best
        L.tmac_hip_set_gemm_min_n(32)
        print(f"bitnet-3b prefill N={N} {name:8s} Mw={Mw}x{cnt} K={K}: LUT build + one-hot GEMM {res['gemm']:8.1f} us | LUT build + GEMV row loop {res['loop']:8.1f} us")
        for w in ws: w.free()
        L.tmac_hip_cache_clear()
    print(f"BitNet-b1.58-3B prefill, {N} tokens, 26 layers of mpGEMMs: GEMM {26 * tot['gemm'] * 1e-3:.2f} ms ({N / (26 * tot['gemm']) * 1e6:.0f} tokens/s), "
          f"row loop {26 * tot['loop'] * 1e-3:.2f} ms ({N / (26 * tot['loop']) * 1e6:.0f} tokens/s)")

bitnet_prefill(256)
