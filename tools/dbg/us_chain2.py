import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import tmac_amd
import test_gpu_chain as T
tm = tmac_amd
L = tm.lib()
L.tmac_hip_debug_chain_config(0, 1 << 17)
for seed in (43, 44, 45):
    m = T.Model(tm, T.UNIFIED[:1], bits=2, zp=False, dev_f16=False, seed=seed, mg=1)
    chain = m.record()
    want = None
    for rep in range(6):
        chain.launch(); torch.cuda.synchronize()
        assert chain.status() == 0
        if want is None:
            want = [w.astype(np.float16) for w in m.oracle_outputs(0, m.x_ext[0].float().cpu().numpy())]
        bad = []
        for mi in range(3):
            a = m.outs[0][mi].cpu().numpy()
            idx = np.nonzero(a.view(np.uint16) != want[mi].view(np.uint16))[0]
            bad += [(mi, int(j), float(a[j]), float(want[mi][j])) for j in idx]
        print("seed", seed, "rep", rep, "wpq", chain.wpq(0), "mismatches", bad[:8])
    chain.free(); m.free()
