import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import tmac_amd
import test_gpu_chain as T
tm = tmac_amd
L = tm.lib()
L.tmac_hip_debug_chain_config(0, 1 << 17)
m = T.Model(tm, T.UNIFIED, bits=2, zp=False, dev_f16=False, seed=43, mg=1)
chain = m.record()
chain.launch(); torch.cuda.synchronize()
print("status", chain.status(), "wpq", [chain.wpq(i) for i in range(chain.nops)])
got = [[o.clone() for o in os_] for os_ in m.outs]
for i, (K, rows, src) in enumerate(m.ops):
    x = m.x_ext[i] if src is None else got[src[0]][src[1]]
    L.tmac_hip_debug_quad_config(chain.threads, chain.wpq(i))
    ref = [torch.empty_like(o) for o in got[i]]
    m.wr.fused(m.ws[i], x, ref, 1, act_dtype=tm.F16); torch.cuda.synchronize()
    L.tmac_hip_debug_quad_config(0, 0)
    want = m.oracle_outputs(i, x.float().cpu().numpy())
    for mi in range(len(rows)):
        a, b, w = got[i][mi].cpu().numpy(), ref[mi].cpu().numpy(), want[mi].astype(np.float16)
        bad = np.nonzero(a.view(np.uint16) != b.view(np.uint16))[0]
        bado = np.nonzero(a.view(np.uint16) != w.view(np.uint16))[0]
        bads = np.nonzero(b.view(np.uint16) != w.view(np.uint16))[0]
        print(f"op {i} mat {mi} rows {rows[mi]}: chain!=standalone {len(bad)} chain!=oracle {len(bado)} standalone!=oracle {len(bads)}", bad[:12], bado[:12])
        if len(bado):
            j = bado[0]; print("   e.g.", j, a[j], b[j], w[j], want[mi][j])

# which single-ulp perturbation reproduces the chain's value of op 0 matrix 2 row 182?
i, mi, j = 0, 2, 182
K, rows, src = m.ops[i]
x = m.x_ext[i].float().cpu().numpy()
q, ls, lb = T.orc.preprocessor(x[None, :], K)
A, S = m.host[i][mi]
Cc, cb = T.orc.qgemm_scale_final(A, q, S, ls[:, 0], lb[:, 0], rows[mi], K, 1, 2, T.BITS_BM[2], T.KF, 1)
print("oracle fp32", Cc[0][j], "ls", ls[0, 0], "lb", lb[0, 0], "S", S, "cb shape", cb.shape)
f = np.float32
def final(acc, l_s, l_b, sc):
    return f(f(f(acc * l_s) + f(l_b * f(0.5))) * sc)
# planes of output row j in M-space: rows mrow(j, p)
def mrow(o, p, bits): return (o // 8) * 8 * bits + p * 8 + (o % 8)
c0, c1 = int(cb[0][mrow(j, 0, 2)]), int(cb[0][mrow(j, 1, 2)])
acc = f(f(f(c0) * f(0.5)) + f(f(c1) * f(1.0)))
base = final(acc, ls[0, 0], lb[0, 0], S[0])
print("cb", c0, c1, "recomputed", base, np.float16(base))
for name, dl, db in [("ls+1", 1, 0), ("ls-1", -1, 0), ("lb+1", 0, 1), ("lb-1", 0, -1)]:
    l2 = (ls[0, 0:1].view(np.uint32) + np.uint32(dl)).view(np.float32)[0] if dl >= 0 else (ls[0, 0:1].view(np.uint32) - np.uint32(1)).view(np.float32)[0]
    b2 = (lb[0, 0:1].view(np.uint32) + np.uint32(db)).view(np.float32)[0] if db >= 0 else (lb[0, 0:1].view(np.uint32) - np.uint32(1)).view(np.float32)[0]
    v = final(acc, l2, b2, S[0])
    print(name, v, np.float16(v))
