#!/usr/bin/env python3
"""Reads gpurun_out/chain_stamps.npy (bench.py --stamps: s_memrealtime stamps [calls][workgroups][8] of wave 0 of
k_decode_chain) and prints where a decoded token's time goes, call kind by call kind."""
import sys
import numpy as np

raw = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/chain_stamps.npy")
st = raw[:, :, :7].astype(np.float64) * 0.01
nops, nwg, _ = st.shape
names = ["qkv", "o", "gate_up", "down"]
t0 = st[0, :, 0].min()
print(f"{nops} calls, {nwg} workgroups, launch span {st[:, :, 6].max() - t0:.1f} us")
pub = st[:, :, 5]                      # last publish of wave 0 of each workgroup
complete = pub.max(axis=1)             # (approximately: other waves may publish slightly later)
lo, hi = 8, nops - 8
print("kind      period | entry->in  polls | in->issued->built->barrier | lut->pub | pub spread (max-min, p95-p50) | in-done minus last pub (min/mean/max) | in spread")
for k, name in enumerate(names):
    idx = np.arange(lo + (k - lo) % 4, hi, 4)
    per = np.mean(complete[idx] - complete[idx - 1])
    s = st[idx]
    prev = pub[idx - 1]
    lastp = prev.max(axis=1)
    d = s[:, :, 1] - lastp[:, None]
    print(f"{name:8s} {per:7.2f} | {np.mean(s[:, :, 1] - s[:, :, 0]):7.2f} {np.mean(raw[idx, :, 7]):6.2f} | {np.mean(s[:, :, 3] - s[:, :, 1]):5.2f} {np.mean(s[:, :, 4] - s[:, :, 3]):5.2f} {np.mean(s[:, :, 2] - s[:, :, 4]):5.2f} | "
          f"{np.mean(s[:, :, 5] - s[:, :, 2]):6.2f} | "
          f"{np.mean(prev.max(axis=1) - prev.min(axis=1)):5.2f} {np.mean(np.percentile(prev, 95, axis=1) - np.median(prev, axis=1)):5.2f} | "
          f"{np.mean(d.min(axis=1)):5.2f} {np.mean(d):5.2f} {np.mean(d.max(axis=1)):5.2f} | {np.mean(s[:, :, 1].max(axis=1) - s[:, :, 1].min(axis=1)):5.2f}")
late = (pub[lo:hi] - np.median(pub[lo:hi], axis=1, keepdims=True)).mean(axis=0)
print("mean lateness of a workgroup's publish vs the median: min %.2f max %.2f; by XCD (wg %% 8):" % (late.min(), late.max()),
      " ".join("%.2f" % late[x::8].mean() for x in range(8)))
print("layer period: %.2f us" % ((complete[hi - 1] - complete[lo - 1]) / ((hi - lo) / 4)))
