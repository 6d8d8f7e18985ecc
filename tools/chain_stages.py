#!/usr/bin/env python3
"""Stage-by-stage means of the chain's stamps (layout: tmac_chain.h), us after the previous call's last publish anywhere (T)."""
import sys
import numpy as np
raw = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/chain_stamps.npy")
names = sys.argv[2].split(",") if len(sys.argv) > 2 else ["qkv", "o", "gate_up", "down"]
st = raw.astype(np.float64) * 0.01
pub = st[:, :, 5]
nk = len(names)
for k, name in enumerate(names):
    acc = []
    for i in range(8 + (k - 8) % nk, st.shape[0] - 8, nk):
        T = pub[i - 1].max(); rel = st[i] - T
        acc.append([pub[i].max() - T] + [rel[:, j].mean() for j in (6, 11, 1, 12, 3, 0, 2, 4, 9, 10, 8, 5)])
    a = np.array(acc).mean(axis=0)
    print("%-8s period %.2f | B: start %.2f poll1 %.2f acts %.2f offered %.2f idle %.2f | L0: enter %.2f step1 %.2f done %.2f, last wave done %.2f, L0 leaves %.2f | P: arrivals %.2f published %.2f" % (name, *a))
