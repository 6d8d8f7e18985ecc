"""Prefill (N > 1) timing of qgemm_lut on the llama-2-7B W2 shapes: preprocessor (LUT build for N rows) + qgemm,
one-hot MFMA GEMM (k_gemm_onehot) against the GEMV kernel looped over the rows, plus a dense fp16 torch.matmul of
the same shape for scale.  hipEvent timing, eager launches (kernels are 50+ us).  usage: bench_prefill.py [N]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tmac_amd
from tmac_amd import KCfg, F16
L = tmac_amd.lib()
dev = torch.device("cuda")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
BITS = int(sys.argv[2]) if len(sys.argv) > 2 else 2          # 2 or 4
BM = 128 if BITS == 2 else 256
wr = tmac_amd.TMACGeMMWrapper(act_group_size=64); wr.set_workspace(11008, N)

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best

tot = {"gemm": 0.0, "loop": 0.0, "pre": 0.0, "dense": 0.0}
for name, Mw, K, cnt in [("qkv/o", 4096, 4096, 4), ("gate/up", 11008, 4096, 2), ("down", 4096, 11008, 1)]:
    A = torch.randint(0, 256, (Mw * BITS // BM, K // 4, BM // 2), dtype=torch.uint8, device=dev)
    S = (torch.randn((Mw * BITS // BM, K // 128, BM // BITS // 8, 2, 8), device=dev) * 0.01).half().contiguous()
    w = tmac_amd.Weights(A, S, Mw, K, BITS, KCfg.make(Mw, K, BITS, BM), scales_dtype=F16, dev_dtype=F16, on_device=True)
    x = torch.randn(N, K, device=dev).half()
    out = torch.empty(N, Mw, dtype=torch.float16, device=dev)
    Wd = torch.randn(Mw, K, device=dev).half()
    t_pre = timeit(lambda: wr.llama_cpp_init(x, Mw, K, N, BITS))
    L.tmac_hip_set_gemm_min_n(1)
    t_gemm = timeit(lambda: wr.llama_cpp_compute(w, out, N))
    L.tmac_hip_set_gemm_min_n(0)
    t_loop = timeit(lambda: wr.llama_cpp_compute(w, out, N), reps=3)
    L.tmac_hip_set_gemm_min_n(32)
    t_fused = timeit(lambda: wr.fused([w], x, [out], N))     # pair-wise LUT build (image only) + GEMM in one call
    if cnt > 1:   # the matrices of a model that share this activation block (q/k/v: 3, gate/up: 2) in ONE fused call
        nshare = 3 if name == "qkv/o" else cnt
        ws_ = [w] + [tmac_amd.Weights(A, S, Mw, K, BITS, KCfg.make(Mw, K, BITS, BM), scales_dtype=F16, dev_dtype=F16, on_device=True) for _ in range(nshare - 1)]
        outs_ = [out] + [torch.empty_like(out) for _ in range(nshare - 1)]
        t_multi = timeit(lambda: wr.fused(ws_, x, outs_, N))
        print(f"         {nshare} matrices sharing the activations in one fused call: {t_multi:8.1f} us = {t_multi / nshare:7.1f} us per matrix "
              f"(single calls: {t_fused:7.1f} us each)")
        tot["multi"] = tot.get("multi", 0.0) + t_multi + (t_fused if name == "qkv/o" else 0.0)   # q/k/v together + o alone
        for w_ in ws_[1:]:
            w_.free()
    else:
        tot["multi"] = tot.get("multi", 0.0) + t_fused
    t_dense = timeit(lambda: torch.matmul(x, Wd.t()))
    ops = 2.0 * (Mw * BITS) * (K / 4 * 8) * N      # MFMA work actually issued: 8-entry half tables
    print(f"{name:8s} Mw={Mw} K={K} N={N} W{BITS}: preprocessor {t_pre:8.1f} us | one-hot MFMA gemm {t_gemm:8.1f} us "
          f"({ops / t_gemm * 1e-6:7.1f} int8 TOP/s, {2.0 * Mw * K * N / t_gemm * 1e-6:6.1f} dense-equivalent TFLOP/s) | "
          f"gemv loop {t_loop:9.1f} us | dense fp16 matmul {t_dense:7.1f} us | fused entry (LUT build + gemm) {t_fused:7.1f} us")
    tot["gemm"] += cnt * t_gemm; tot["loop"] += cnt * t_loop; tot["dense"] += cnt * t_dense
    tot["fusedpre"] = tot.get("fusedpre", 0.0) + (t_fused - t_gemm) * (1 if name != "qkv/o" else 2)
    tot["pre"] += t_pre * (1 if name != "qkv/o" else 2)     # one LUT build per distinct activation tensor
    w.free()
t = 32 * tot["multi"]
print(f"llama-2-7B prefill, {N} tokens, 32 layers of mpGEMMs, fused entry with q/k/v and gate/up batched:  {t * 1e-3:9.2f} ms  -> {N / t * 1e6:10.0f} tokens/s")
t = 32 * (tot["gemm"] + tot["fusedpre"])
print(f"llama-2-7B prefill, {N} tokens, 32 layers of mpGEMMs, fused entry (pair-wise LUT build): {t * 1e-3:9.2f} ms  -> {N / t * 1e6:10.0f} tokens/s")
for k in ("gemm", "loop", "dense"):
    t = 32 * (tot[k] + (tot["pre"] if k != "dense" else 0.0))
    print(f"llama-2-7B prefill, {N} tokens, 32 layers of mpGEMMs, {k:5s}: {t * 1e-3:9.2f} ms  -> {N / t * 1e6:10.0f} tokens/s")
