"""Independent GEMVs of the headline shape (llama-2-7B W2 down projection, 4096 x 11008, distinct weights > MALL) issued on S streams
round-robin -- eager and as ONE hipGraph with S parallel branches -- against the single-stream sequence bench.py's headline_gemv times.
SURVEY 8d's measurement is "N back-to-back launches in one stream / HIP graph over rotating distinct weight buffers": the calls carry no
data dependence, so a graph may run them side by side.  Prints us per GEMV and the fraction of the 8 TB/s HBM peak.
usage: bench_concurrent.py [NSET] ; knobs per line: (threads, waves per quad) of k_gemv_quad, 0 0 = the library's choice."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tmac_amd
from tmac_amd import KCfg, F16
dev = torch.device("cuda")
NSET = int(sys.argv[1]) if len(sys.argv) > 1 else 48
Mw, K, bits, bm, gs, ags = 4096, 11008, 2, 128, 128, 64
L = tmac_amd.lib()
wr = tmac_amd.TMACGeMMWrapper(act_group_size=ags); wr.set_workspace(K, 1)
cfg = KCfg.make(Mw, K, bits, bm, 16, gs, ags, True, -1)
sets = []
for _ in range(NSET):
    A = torch.randint(0, 256, (Mw * bits // bm, K // 4, bm // 2), dtype=torch.uint8, device=dev)
    S = (torch.randn((Mw * bits // bm, K // gs, bm // bits // 8, 2, 8), device=dev) * 0.01).half().contiguous()
    sets.append([tmac_amd.Weights(A, S, Mw, K, bits, cfg, scales_dtype=F16, dev_dtype=F16, on_device=True)])
xs = [torch.randn(K, device=dev).half() for _ in range(NSET)]
outs = [[torch.empty(Mw, dtype=torch.float16, device=dev)] for _ in range(NSET)]
hb = Mw * K * bits // 8 + Mw * (K // gs) * 2 * 2 + K // 4 * 16 + (K // ags) * 4 + Mw * 2


def issue(streams):
    for i, ws in enumerate(sets):
        wr.fused(ws, xs[i], outs[i], 1, stream=streams[i % len(streams)])


def measure(S, graph):
    cur = torch.cuda.current_stream()
    streams = [torch.cuda.Stream() for _ in range(S)] if S > 1 else [cur]

    def fork_join():
        if S > 1:
            for s in streams: s.wait_stream(torch.cuda.current_stream())
        issue(streams if S > 1 else [torch.cuda.current_stream()])
        if S > 1:
            for s in streams: torch.cuda.current_stream().wait_stream(s)
    fork_join(); torch.cuda.synchronize()
    g = None
    if graph:
        side = torch.cuda.Stream(); side.wait_stream(cur)
        with torch.cuda.stream(side):
            fork_join()
        cur.wait_stream(side); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fork_join()
        g.replay(); torch.cuda.synchronize()
    ts = []
    for rep in range(8):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay() if g is not None else fork_join()
        e1.record(); torch.cuda.synchronize()
        if rep >= 2:
            ts.append(e0.elapsed_time(e1) * 1e3 / NSET)
    return float(np.mean(ts)), float(np.min(ts))


ref = None
for ft, wpq in ((0, 0), (512, 2), (512, 1)):
    tmac_amd.binding.check(L.tmac_hip_debug_quad_config(ft, wpq))
    for graph in (True, False):
        for S in (1, 2, 3, 4):
            try:
                mean, best = measure(S, graph)
            except Exception as e:
                print(f"cfg=({ft},{wpq}) streams={S} graph={graph}: failed {e!r}")
                continue
            print(f"cfg=({ft},{wpq}) streams={S} {'graph' if graph else 'eager'}: {mean:6.2f} us/GEMV (best {best:6.2f})  "
                  f"{hb / mean * 1e-3:7.1f} GB/s  frac {hb / mean * 1e-3 / 8000:.3f}", flush=True)
    # results identical whatever the schedule
    torch.cuda.synchronize()
    cur_out = torch.stack([o[0] for o in outs]).clone()
    if ref is None:
        ref = cur_out
    else:
        print("  outputs equal to the first configuration's:", bool(torch.equal(ref, cur_out)))
tmac_amd.binding.check(L.tmac_hip_debug_quad_config(0, 0))
