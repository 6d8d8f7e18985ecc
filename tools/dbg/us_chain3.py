import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import tmac_amd
import test_gpu_chain as T
tm = tmac_amd
L = tm.lib()
L.tmac_hip_debug_chain_config(0, 1 << 17)
m = T.Model(tm, T.UNIFIED[:1], bits=2, zp=False, dev_f16=False, seed=43, mg=1)
chain = m.record()
buf = torch.zeros(chain.nops * chain.grid * 8, dtype=torch.int64, device="cuda")
chain.set_stamps(buf)
chain.launch(); torch.cuda.synchronize()
raw = buf.cpu().numpy().reshape(chain.nops, chain.grid, 8)[0, :, 4].astype(np.uint64)
ls = (raw & np.uint64(0xffffffff)).astype(np.uint32).view(np.float32)
lb = (raw >> np.uint64(32)).astype(np.uint32).view(np.float32)
x = m.x_ext[0].float().cpu().numpy()
q, lso, lbo = T.orc.preprocessor(x[None, :], 3200)
print("oracle ls", lso[0, 0], hex(lso[0, 0:1].view(np.uint32)[0]), "lb", lbo[0, 0], hex(lbo[0, 0:1].view(np.uint32)[0]))
act = raw != 0
print("chain ls values", {hex(v) for v in ls[act].view(np.uint32)}, "lb values", {hex(v) for v in lb[act].view(np.uint32)}, "workgroups", int(act.sum()))

r = buf.cpu().numpy().reshape(chain.nops, chain.grid, 8)[0, 167]
a4, a6 = int(np.uint64(r[4])), int(np.uint64(r[6]))
def fl(u): return np.array([u & 0xffffffff], np.uint32).view(np.float32)[0]
def s16(u): return u - 65536 if u >= 32768 else u
print("t", repr(fl(a4)), "cb0", s16((a4 >> 32) & 0xffff), "cb1", s16((a4 >> 48) & 0xffff), "v", repr(fl(a6)), "scale", repr(fl(a6 >> 32)))
print("out", m.outs[0][2][182].item())
