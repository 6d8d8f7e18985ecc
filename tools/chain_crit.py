#!/usr/bin/env python3
"""critical-path view of gpurun_out/chain_stamps.npy: for every call, the workgroup that published last"""
import sys
import numpy as np
raw = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/chain_stamps.npy")
st = raw[:, :, :7].astype(np.float64) * 0.01
names = ["qkv", "o", "gate_up", "down"]
pub = st[:, :, 5]
acc = {n: [] for n in names}
for i in range(8, st.shape[0] - 8):
    c = int(np.argmax(pub[i])); prev = pub[i - 1].max()
    e, inn, iss, built, lut, p = st[i, c, 0], st[i, c, 1], st[i, c, 3], st[i, c, 4], st[i, c, 2], st[i, c, 5]
    acc[names[i % 4]].append([p - prev, e - prev, inn - e, raw[i, c, 7], iss - inn, built - iss, lut - built, p - lut])
for n in names:
    a = np.array(acc[n]).mean(axis=0)
    print("%-8s period %.2f = entry after previous call's last publish %.2f + wait %.2f (polls %.1f) + issue %.2f + build %.2f + barrier %.2f + lookups %.2f" % (n, *a))
