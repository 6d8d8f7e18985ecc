"""Step stamps of k_gemm_planes (workgroup 0): where an act-group step of each wave spends its time.
NOTE (round 6): needs a profiling build of the library (tools/build_variant_obj.sh st tmac_gemm2 "-DTMAC_G2_STAMPS=1", TMAC_HIP_LIB=.../libtmac_hip_vst.so):
the default build compiles the hook out (it cost the W4 prefill line 1 %).  The stamps INSIDE a step (-DTMAC_G2_STEP_STAMPS=1 builds) distort the direct form -- their stores make the compiler drain
all loads behind every stamp, a step takes twice as long; the per-step durations of a normal build (stamp 0 only) remain meaningful.
usage: gemm2_stamps.py [Mw K N]   (W2 g128 zero points, fp16 scales / activations / outputs)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tmac_amd
from tmac_amd import KCfg, F16
L = tmac_amd.lib()
dev = torch.device("cuda")
Mw, K, N = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 4096, 256)
BITS, BM = 2, 128
wr = tmac_amd.TMACGeMMWrapper(act_group_size=64)
A = torch.randint(0, 256, (Mw * BITS // BM, K // 4, BM // 2), dtype=torch.uint8, device=dev)
S = (torch.randn((Mw * BITS // BM, K // 128, BM // BITS // 8, 2, 8), device=dev) * 0.01).half().contiguous()
w = tmac_amd.Weights(A, S, Mw, K, BITS, KCfg.make(Mw, K, BITS, BM), scales_dtype=F16, dev_dtype=F16, on_device=True)
x = torch.randn(N, K, device=dev).half()
out = torch.empty(N, Mw, dtype=torch.float16, device=dev)
for _ in range(3):
    wr.fused([w], x, [out], N)
torch.cuda.synchronize()
buf = torch.zeros(8 * 64 * 8, dtype=torch.int64, device=dev)
tmac_amd.binding.check(L.tmac_hip_debug_gemm_stamps(buf.data_ptr()))
wr.fused([w], x, [out], N)
torch.cuda.synchronize()
L.tmac_hip_debug_gemm_stamps(None)
st = buf.cpu().numpy().reshape(8, 64, 8).astype(np.int64)
t0 = st[:, 0, 5].min()
us = lambda v: (v - t0) / 100.0
print(f"{Mw} x {K}, N = {N}: workgroup 0, times in us from the first wave's start (s_memrealtime, 10 ns)")
for wv in range(8):
    s = st[wv]
    nsteps = int((s[:, 0] > 0).sum())
    print(f"wave {wv}: start {us(s[0,5]):6.2f}  operand rows built {us(s[0,6]):6.2f}  loop end {us(s[1,5]):6.2f}  reduce barrier {us(s[1,6]):6.2f}  stored {us(s[2,5]):6.2f}   steps {nsteps}")
    tops = [us(s[k, 0]) for k in range(nsteps)] + [us(s[1, 5])]
    print("    step durations: " + " ".join(f"{tops[k + 1] - tops[k]:5.2f}" for k in range(nsteps)))
    if nsteps > 2 and (s[1:nsteps, 1] > 0).all():
        # inside a step (round 6, direct form -- no hand-placed wait any more): top -> [more: staged scales to LDS] -> tile row 0's gathers, column
        # values, the next act group's loads issued -> first two chains + tile row 1's gathers and weights -> fp32 chain 0, chains 2 and 3 with the
        # next B operands behind them, fp32 chain 1 -> fp32 chains 2, 3 (+ zero-point update)
        k = np.arange(1, nsteps - 1)
        seg = [(s[k, 1] - s[k, 0]), (s[k, 2] - s[k, 1]), (s[k, 3] - s[k, 2]), (s[k, 4] - s[k, 3]), (s[k + 1, 0] - s[k, 4])]
        print("    mean of steps 1..n-2 (us): step top %.2f | gathers of tile row 0 + loads issued %.2f | chains 0,1 + gathers / weights of tile row 1 %.2f | fp32 chains 0,1 + chains 2,3 + B loads %.2f | fp32 chains 2,3 %.2f"
              % tuple(float(x.mean()) / 100.0 for x in seg))
