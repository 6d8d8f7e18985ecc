"""Row-sharded decode chain across processes: every rank's producers store their hand-off granules into the arenas of ALL ranks
through IPC-mapped memory (tmac_hip_chain_record_gather / _export / _connect, include/tmac_hip.h); the partition is the
reference's own -- M-tiles split over callers, every caller needs the whole LUT (`include/t-mac/tmac_gemm_wrapper.h:197-199`,
`python/t_mac/ops/qgemm.py:268-273`) -- K is never split.

The single-GPU test box cannot hold two ranks on two devices, and RCCL refuses two ranks on one device; IPC does not.  So two
PROCESSES share device 0, each with a chain of 96 workgroups (tmac_hip_debug_chain_grid: both persistent kernels must be resident
together), exchange their blobs through files, launch together, and every rank must end up with the rows of the unsharded chain
bit for bit (same waves per row quad => same fp32 summation order; rows are independent)."""
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, time
root, rank, world, d, bits, mg = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
sys.path.insert(0, root)
import numpy as np, torch
import tmac_amd
from oracle import oracle as orc
tm = tmac_amd
L = tm.lib()
tm.binding.check(L.tmac_hip_init(0))
KF, GS, AGS = 16, 128, 64
bm = {2: 128, 4: 256}[bits]
zp = mg < 1
# ops: (K, full rows Mw); each op's single output is gathered and feeds the next op
OPS = [(1024, 2048), (2048, 1024), (1024, 4096), (4096, 1024), (1024, 512)]

def barrier(tag):
    open(os.path.join(d, f"{tag}.{rank}"), "w").close()
    t0 = time.time()
    while not all(os.path.exists(os.path.join(d, f"{tag}.{r}")) for r in range(world)):
        if time.time() - t0 > 120: raise SystemExit("barrier timeout " + tag)
        time.sleep(0.002)

def build(shard_rank, shard_world, grid, wpq):
    tm.binding.check(L.tmac_hip_reset_state())
    tm.binding.check(L.tmac_hip_debug_chain_grid(grid))
    tm.binding.check(L.tmac_hip_debug_chain_config(wpq, 1 << 18))
    wr = tm.TMACGeMMWrapper(act_group_size=AGS)
    ws, outs, gath, host = [], [], [], []
    for i, (K, Mw) in enumerate(OPS):
        ags = K if mg >= 1 else AGS
        case = orc.make_case(900 + i, Mw, K, bits=bits, gs=GS, ags=ags, zero_point=zp, m_groups=mg, fp16_values=True)
        if mg >= 1:
            case["w"] = np.random.default_rng(70 + i).integers(1, 4, size=(Mw, K), dtype=np.uint8)      # ternary: zero-mean
            S = (case["sc"] * 0 + 1.0 / np.sqrt(2.0 * K / 3.0)).astype(np.float16).astype(np.float32)
        else:
            c = 1.0 / np.sqrt(2.5 * K)
            case["sc"] = (case["sc"] * c).astype(np.float16).astype(np.float32)
            lvl = (2 ** bits - 1) / 2.0 - 2 ** (bits - 1)
            case["zr"] = (case["zr"] * c + lvl * case["sc"]).astype(np.float16).astype(np.float32)
        rows = Mw // shard_world
        sl = slice(shard_rank * rows, (shard_rank + 1) * rows)
        A = orc.preprocess_weights(case["w"][sl], bits, bm, KF)
        Sb = S if mg >= 1 else orc.preprocess_scales(case["sc"][sl], case["zr"][sl], bits, bm)
        cfg = tm.KCfg.make(rows, K, bits, bm, KF, GS, ags, zp, mg)
        ws.append(wr.register_weights(A, Sb, rows, K, bits, cfg, scales_dtype=tm.F32, dev_dtype=tm.F16))
        host.append((A, Sb))
        outs.append(torch.zeros(rows, dtype=torch.float16, device="cuda"))
        gath.append(torch.zeros(Mw, dtype=torch.float16, device="cuda"))
    x0 = torch.from_numpy(np.random.default_rng(5).standard_normal(OPS[0][0]).astype(np.float32)).cuda().half()
    with wr.record_chain() as rec:
        x = x0
        for i in range(len(OPS)):
            wr.fused([ws[i]], x, [outs[i]], 1, act_dtype=tm.F16)
            if shard_world > 1:
                wr.record_gather(outs[i], gath[i], outs[i].numel() * 2, shard_rank, shard_world)
                x = gath[i]
            else:
                x = outs[i]
    return rec.chain, outs, ws, host, x0

# ---- the sharded chain of this rank: grid 96, two waves per quad
chain, outs, ws, _, _ = build(rank, world, 96, 2)
blob = chain.export()
open(os.path.join(d, f"blob.{rank}.tmp"), "wb").write(blob)
os.rename(os.path.join(d, f"blob.{rank}.tmp"), os.path.join(d, f"blob.{rank}"))
barrier("exported")
chain.connect([open(os.path.join(d, f"blob.{r}"), "rb").read() for r in range(world)])
res = []
for rep in range(3):
    for o in outs: o.fill_(float(rep))
    torch.cuda.synchronize()
    barrier(f"launch{rep}")
    chain.launch()
    torch.cuda.synchronize()
    assert chain.status() == 0, f"rank {rank}: a hand-off timed out (rep {rep})"
    res.append([o.clone() for o in outs])
barrier("done")
chain.free()
# ---- reference: the unsharded chain in this process alone (same waves per quad), rank after rank so that it has the device to itself
for turn in range(world):
    if turn == rank:
        ref_chain, ref_outs, _, host, x0 = build(0, 1, 0, 2)
        ref_chain.launch(); torch.cuda.synchronize()
        assert ref_chain.status() == 0
        for i, (K, Mw) in enumerate(OPS):
            rows = Mw // world
            want = ref_outs[i][rank * rows:(rank + 1) * rows]
            assert bool(torch.isfinite(want.float()).all()) and float(want.float().abs().max()) > 0
            for rep in range(3):
                assert torch.equal(res[rep][i], want), f"rank {rank} op {i} rep {rep}: sharded chain != unsharded chain"
            # and against the oracle directly (lut_ctor.cc / tbl.cc restated in oracle/tmac_oracle.c) on the vector the sharded chain consumed:
            # this rank's rows of the full matrix's result, 1e-3 of max |C| (fp16 outputs)
            xin = (x0 if i == 0 else ref_outs[i - 1]).float().cpu().numpy()[None, :]
            A, Sb = host[i]
            if mg >= 1:
                q, ls, lb = orc.preprocessor(xin, K)
                full = orc.qgemm_scale_final(A, q, Sb, ls[:, 0], lb[:, 0], Mw, K, 1, bits, bm, KF, mg)[0][0]
            else:
                q, ls, lb = orc.preprocessor(xin, AGS)
                full = orc.qgemm_float(A, q, Sb, ls, lb, Mw, K, 1, bits, bm, KF, GS, AGS, zp)[0]
            got = res[0][i].float().cpu().numpy()
            err = float(np.abs(got - full[rank * rows:(rank + 1) * rows]).max() / max(np.abs(full).max(), 1e-30))
            assert err <= 1e-3, f"rank {rank} op {i}: sharded chain vs oracle {err}"
        ref_chain.free()
    barrier(f"ref{turn}")
print("rank", rank, "ok")
'''


@pytest.mark.parametrize("bits,mg", [(2, -1), (4, -1), (2, 1)])
def test_two_processes_share_one_device(bits, mg):
    with tempfile.TemporaryDirectory() as d:
        script = os.path.join(d, "worker.py")
        open(script, "w").write(WORKER)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        ps = [subprocess.Popen([sys.executable, script, ROOT, str(r), "2", d, str(bits), str(mg)], env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
        outs = []
        for p in ps:
            try:
                outs.append(p.communicate(timeout=300)[0].decode())
            except subprocess.TimeoutExpired:
                p.kill()
                outs.append("TIMEOUT\n" + p.communicate()[0].decode())
        assert all(p.returncode == 0 for p in ps), "\n".join(o[-3000:] for o in outs)
