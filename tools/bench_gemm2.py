"""Per-shape timing of the N > 1 kernels on the llama-2-7B shapes: LUT image build and k_gemm_planes through the fused
entry point (one call = k_lut_image + k_gemm_planes) and the GEMM alone, replayed from a hipGraph.
usage: bench_gemm2.py [N] [bits] [kernel: 0 = k_gemm_planes, form by tile count (default), 1 = k_gemm_onehot, 2 / 3 = k_gemm_planes with
eight- / four-wave workgroups]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tmac_amd
from tmac_amd import KCfg, F16
L = tmac_amd.lib()
dev = torch.device("cuda")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
BITS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
BM = {1: 64, 2: 128, 3: 192, 4: 256}[BITS]
KERNEL = int(sys.argv[3]) if len(sys.argv) > 3 else 0
tmac_amd.binding.check(L.tmac_hip_debug_gemm_kernel(KERNEL))
L.tmac_hip_set_gemm_min_n(1)
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


print("N =", N, " bits =", BITS, " kernel =", ["k_gemm_planes (auto form)", "k_gemm_onehot", "k_gemm_planes, 8 waves", "k_gemm_planes, 4 waves"][KERNEL])
for name, Mw, K, nshare in [("o", 4096, 4096, 1), ("qkv", 4096, 4096, 3), ("gate_up", 11008, 4096, 2), ("down", 4096, 11008, 1)]:
    ws, outs = [], []
    for _ in range(nshare):
        A = torch.randint(0, 256, (Mw * BITS // BM, K // 4, BM // 2), dtype=torch.uint8, device=dev)
        S = (torch.randn((Mw * BITS // BM, K // 128, BM // BITS // 8, 2, 8), device=dev) * 0.01).half().contiguous()
        ws.append(tmac_amd.Weights(A, S, Mw, K, BITS, KCfg.make(Mw, K, BITS, BM), scales_dtype=F16, dev_dtype=F16, on_device=True))
        outs.append(torch.empty(N, Mw, dtype=torch.float16, device=dev))
    x = torch.randn(N, K, device=dev).half()
    t = timeit(lambda: wr.fused(ws, x, outs, N))
    if nshare == 1:   # the GEMM alone: LUT built once by the split entry point, then only tmac_hip_qgemm_dev in the graph
        wr.set_workspace(K, N)
        wr.llama_cpp_init(x, Mw, K, N, BITS)
        tg = timeit(lambda: wr.llama_cpp_compute(ws[0], outs[0], N))
        print(f"{name:8s} gemm alone {tg:8.1f} us")
    ops = 2.0 * Mw * nshare * (K / 4 * 8) * N
    print(f"{name:8s} {nshare} x {Mw} x {K}: LUT image + gemm {t:8.1f} us  ({ops / t * 1e-6:7.1f} int8 TOP/s issued, {ops / t * 1e-6 / 4404 * 100:5.1f} % of the 32x32x32 i8 ceiling)")
    for w in ws:
        w.free()
