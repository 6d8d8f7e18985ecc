"""Back-to-back independent GEMVs (SURVEY 8d's headline measurement) through the recorded-sequence API in stream mode
(k_lut_images + k_gemv_stream): NL matrices of one shape with distinct weights (> MALL in total) and a DISTINCT activation vector per
call, one launch; us per GEMV (hipEvent pair, mean / best of 10), fraction of the 8 TB/s HBM peak on SURVEY 8d's algorithmic bytes, and
a bit-comparison of every call's output with the same call launched on its own (tmac_hip_qgemm_fused_dev, the chain's configuration).
usage: bench_stream.py [shape ...]   shape = MwxK[xCNT][:bits]   default: 4096x11008 4096x4096 11008x4096x2 4096x4096x3
env: NL (calls per launch, default 32), TMAC_CHAIN_STREAM=0 measures k_decode_chain on the same recording, FORCE_WPQ=n forces the waves per row quad,
TMAC_STREAM_SPLIT=1 one workgroup per CU; STAMPS=1|2 with a profiling build of the library (tools/build_variant.sh x "-DTMAC_STREAM_STAMPS=1|2",
TMAC_HIP_LIB=.../libtmac_hip_x.so): 1 = where and when every workgroup ran (co-residency), 2 = where a lookup wave's cycles go."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tmac_amd
from tmac_amd import KCfg, F16
dev = torch.device("cuda")
NL = int(os.environ.get("NL", "32"))
L = tmac_amd.lib()
if os.environ.get("FORCE_WPQ"):        # A/B: waves per row quad of the recorded calls (tmac_hip_debug_chain_config)
    tmac_amd.binding.check(L.tmac_hip_debug_chain_config(int(os.environ["FORCE_WPQ"]), 0))
gen = torch.Generator(device=dev); gen.manual_seed(7)


def algorithmic_bytes(Mw, K, bits, gs=128, ags=64, zp=True):
    return Mw * K * bits // 8 + Mw * (K // gs) * (2 if zp else 1) * 2 + K // 4 * 16 + (K // ags) * 4 + Mw * 2


def run(Mw, K, cnt, bits):
    bm = {1: 64, 2: 128, 3: 192, 4: 256}[bits]
    wr = tmac_amd.TMACGeMMWrapper(act_group_size=64); wr.set_workspace(K, 1)
    cfg = KCfg.make(Mw, K, bits, bm, 16, 128, 64, True, -1)
    sets, xs, outs = [], [], []
    c = 1.0 / np.sqrt(2.5 * K)
    for _ in range(NL):
        ws = []
        for _ in range(cnt):
            A = torch.randint(0, 256, (Mw * bits // bm, K // 4, bm // 2), dtype=torch.uint8, device=dev, generator=gen)
            S = (torch.randn((Mw * bits // bm, K // 128, bm // bits // 8, 2, 8), device=dev, generator=gen) * c).half().contiguous()
            ws.append(tmac_amd.Weights(A, S, Mw, K, bits, cfg, scales_dtype=F16, dev_dtype=F16, on_device=True))
        sets.append(ws)
        xs.append(torch.randn(K, device=dev, generator=gen).half())
        outs.append([torch.zeros(Mw, dtype=torch.float16, device=dev) for _ in range(cnt)])
    with wr.record_chain() as rec:
        for i in range(NL):
            wr.fused(sets[i], xs[i], outs[i], 1, act_dtype=F16, out_dtype=F16)
    ch = rec.chain
    stamps = None
    if os.environ.get("STAMPS"):       # = the level of a -DTMAC_STREAM_STAMPS=1|2 build (tmac_chain.h, StreamArgs::stamps)
        stamps = torch.zeros((512, 12, 8), dtype=torch.int64, device=dev)
        ch.set_stamps(stamps)
    ts = []
    SB = int(os.environ.get("SB", "1"))        # SB > 1: that many replays back to back per event pair (sustained clocks, launch gaps included), as bench.py times it
    for r in range(13):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(SB):
            ch.launch()
        e1.record(); torch.cuda.synchronize()
        if r >= 3:
            ts.append(e0.elapsed_time(e1) * 1e3 / (NL * SB))
    assert ch.status() == 0
    got = [[o.clone() for o in os_] for os_ in outs]
    same = True
    for i in range(NL):
        if not getattr(ch, "quarter_walk", False):
            L.tmac_hip_debug_quad_config(ch.threads, ch.wpq(i))
        ref = [torch.empty_like(o) for o in outs[i]]
        wr.fused(sets[i], xs[i], ref, 1, act_dtype=F16, out_dtype=F16)
        torch.cuda.synchronize()
        if getattr(ch, "quarter_walk", False):     # quarter-walk form: another fp32 summation order (fp16 outputs: within 2 ulp of the stand-alone launch)
            same = same and all(float((a.float() - b.float()).abs().max()) <= 2e-3 * float(b.float().abs().max()) for a, b in zip(got[i], ref))
        else:
            same = same and all(torch.equal(a, b) for a, b in zip(got[i], ref))
    L.tmac_hip_debug_quad_config(0, 0)
    hb = cnt * algorithmic_bytes(Mw, K, bits) - (cnt - 1) * (K // 4 * 16 + (K // 64) * 4)
    mean, best = float(np.mean(ts)), float(np.min(ts))
    print(f"{Mw}x{K}x{cnt} W{bits} {('stream-qw' if getattr(ch, 'quarter_walk', False) else 'stream') if ch.stream else 'chain '} wpq={ch.wpq(0)}: {mean:6.2f} us/call (best {best:6.2f})  {hb / mean * 1e-3:7.1f} GB/s  "
          f"frac {hb / mean * 1e-3 / 8000:.3f}  bit-identical to the stand-alone launches: {same}", flush=True)
    if stamps is not None:
        raw = stamps.cpu().numpy()
        if os.environ["STAMPS"] == "2":                               # a -DTMAC_STREAM_STAMPS=2 build
            hwid = raw[:, :, 7] >> 32
            raw = raw.copy(); raw[:, :, 7] &= 0xffffffff
            st = raw.astype(np.float64)
            simd = (hwid >> 4) & 3
            act = st[:, :, 6] > 0
            for sd in range(4):
                m = act & (simd == sd)
                if m.any():
                    print(f"      SIMD {sd}: {m.sum() / len(st):.2f} busy lookup waves per workgroup, {st[:, :, 1][m].sum() / st[:, :, 6][m].sum():.0f} cycles per item, "
                          f"closing-barrier share {100 * st[:, :, 3][m].mean() / st[:, :, 7][m].mean():.0f} %")
            print("      lookup waves per SIMD, first workgroups:", [np.bincount(simd[i], minlength=4).tolist() for i in range(0, min(len(simd), 6))])
            busy = st[st[:, :, 6] > 0]                                # waves with items
            tot = busy[:, 7].mean()
            pc = [100 * busy[:, i].mean() / tot for i in range(6)]
            print(f"    stamps: {len(busy)} waves with items, {busy[:, 6].mean():.1f} items each, {tot:.0f} cycles first to last: waiting for weights {pc[0]:.0f} %, "
                  f"lookups + refill {pc[1]:.0f} % ({busy[:, 1].sum() / busy[:, 6].sum():.0f} cycles per item), partial sums {pc[2]:.0f} %, closing barriers {pc[3]:.0f} %, "
                  f"op change {pc[4]:.0f} %, A barriers {pc[5]:.0f} %")
        else:                                                         # =1: where and when the workgroups ran
            raw = raw[raw[:, :, 4].max(axis=1) > 0]
            hw = raw[:, 0, 7]
            cu = ((hw >> 32) & 0xf) * 1024 + ((hw >> 13) & 0x7) * 128 + ((hw >> 12) & 1) * 64 + ((hw >> 8) & 0xf)      # xcc, se, sh, cu
            t0 = raw[:, :, 5].min(axis=1); t1 = raw[:, :, 6].max(axis=1)
            ov = []
            for c in np.unique(cu):
                ix = np.nonzero(cu == c)[0]
                if len(ix) == 2:
                    a, b = ix
                    ov.append(max(0, min(t1[a], t1[b]) - max(t0[a], t0[b])) / max(1, min(t1[a] - t0[a], t1[b] - t0[b])))
            print(f"    {len(np.unique(cu))} distinct CUs for {len(raw)} workgroups, {(t1 - t0).mean():.0f} cycles each; {len(ov)} CUs with two: overlap of the pair's life times {np.mean(ov) if ov else 0:.2f}")
    ch.free()
    for ws in sets:
        for w in ws:
            w.free()


shapes = sys.argv[1:] or ["4096x11008", "4096x4096", "11008x4096x2", "4096x4096x3"]
for s in shapes:
    bits = 2
    if ":" in s:
        s, b = s.split(":"); bits = int(b)
    p = [int(v) for v in s.split("x")]
    run(p[0], p[1], p[2] if len(p) > 2 else 1, bits)
