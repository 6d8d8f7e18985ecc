#!/usr/bin/env python3
"""Critical-path view of gpurun_out/chain_stamps.npy (bench.py --stamps): s_memrealtime stamps [calls][workgroups][8] of the
wave-specialised k_decode_chain (layout: tmac_chain.h) -- 0 lookup wave 0 enters the call, 1 builder 0 has the activations of its
first batch, 2 lookup wave 0 sees its first LUT step, 3 builder 0 has built its last block, 4 lookup wave 0 is done with its last
item, 5 the publisher has published the call's last rows, 6 builder 0 starts on the call (LDS buffer free), 7 poll rounds of builder 0.
Times are relative to the moment the PREVIOUS call's last rows were published anywhere on the chip (T)."""
import sys
import numpy as np
raw = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/chain_stamps.npy")
names = sys.argv[2].split(",") if len(sys.argv) > 2 else ["qkv", "o", "gate_up", "down"]
st = raw[:, :, :7].astype(np.float64) * 0.01
pub = st[:, :, 5]
nk = len(names)
print("kind      period | builder: start  acts-in  built (polls) | lookups: enter  1st-step  done | publish   [mean over workgroups / critical workgroup; us after T]")
for k, name in enumerate(names):
    rows, crit = [], []
    for i in range(8 + (k - 8) % nk, st.shape[0] - 8, nk):
        T = pub[i - 1].max()
        c = int(np.argmax(pub[i]))
        rel = st[i] - T
        rows.append([pub[i].max() - T, rel[:, 6].mean(), rel[:, 1].mean(), rel[:, 3].mean(), raw[i, :, 7].mean(), rel[:, 0].mean(), rel[:, 2].mean(), rel[:, 4].mean(), rel[:, 5].mean()])
        crit.append([rel[c, 6], rel[c, 1], rel[c, 3], raw[i, c, 7], rel[c, 0], rel[c, 2], rel[c, 4], rel[c, 5]])
    a, b = np.array(rows).mean(axis=0), np.array(crit).mean(axis=0)
    print("%-8s %6.2f  | %6.2f %6.2f %6.2f (%.1f) | %6.2f %6.2f %6.2f | %6.2f" % (name, *a))
    print("%-8s   crit  | %6.2f %6.2f %6.2f (%.1f) | %6.2f %6.2f %6.2f | %6.2f" % ("", *b))
lo, hi = 8, st.shape[0] - 8
comp = pub.max(axis=1)
print("period per %d calls: %.2f us" % (nk, (comp[hi - 1] - comp[lo - 1]) / ((hi - lo) / nk)))
late = (pub[lo:hi] - np.median(pub[lo:hi], axis=1, keepdims=True)).mean(axis=0)
print("mean lateness of a workgroup's publish vs the median: min %.2f max %.2f; by XCD (wg %% 8):" % (late.min(), late.max()),
      " ".join("%.2f" % late[x::8].mean() for x in range(8)))
