#!/usr/bin/env python3
"""tmac_hip_autotune_fused on the llama-2-7B W2 decode matrix sets as they are sharded over 1, 2, 4, 8 ranks:
what the launch heuristic leaves on the table for shapes it was not derived on.  Prints one line per (world, set).

    python tools/autotune_shapes.py [bits]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tmac_amd  # noqa: E402
from tmac_amd import F16, KCfg  # noqa: E402

bits = int(sys.argv[1]) if len(sys.argv) > 1 else 2
BM = {1: 128, 2: 128, 3: 192, 4: 256}[bits]
MATS = [("qkv", 4096, 4096, 3), ("o", 4096, 4096, 1), ("gate_up", 11008, 4096, 2), ("down", 4096, 11008, 1)]
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(0)
wr = tmac_amd.TMACGeMMWrapper(act_group_size=64)
rpt = BM // bits
for world in (1, 2, 4, 8):
    for name, Mw, K, cnt in MATS:
        Mloc = ((Mw // rpt + world - 1) // world) * rpt
        cfg = KCfg.make(Mloc, K, bits, BM, 16, 128, 64, True)
        ws = []
        for _ in range(cnt):
            A = torch.randint(0, 256, (Mloc * bits // BM, K // 4, BM // 2), dtype=torch.uint8, device=dev, generator=gen)
            S = (torch.randn((Mloc * bits // BM, K // 128, rpt // 8, 2, 8), device=dev, generator=gen) / np.sqrt(2.5 * K)).half().contiguous()
            ws.append(tmac_amd.Weights(A, S, Mloc, K, bits, cfg, scales_dtype=F16, dev_dtype=F16, on_device=True))
        r = wr.autotune(ws, F16, F16)
        print(f"world={world} {name:8s} rows/rank={Mloc:6d} x{cnt} K={K:6d}: heuristic {r['heuristic_us']:.2f} us, "
              f"best ({r['ft']},{r['wpq']}) {r['us']:.2f} us", flush=True)
        for w in ws:
            w.free()
        tmac_amd.lib().tmac_hip_tune_clear()
