"""The multi-GPU exchange step through the C-ABI (tmac_hip_comm_*, RCCL resolved at run time).

With one GPU: a communicator of one rank, all-gather == copy, and a row-sharded GEMV whose "gathered" activation vector
feeds the next LUT build (the sequence bench.py --gpus N runs per call).  With two or more GPUs visible: two processes, one
per GPU, row shards of one matrix, all-gather of the produced halves through the library, every rank holding the whole
vector bit-identical to the single-GPU result (the integer path needs no reduction: K is never split)."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tm():
    import torch
    import tmac_amd
    assert torch.cuda.is_available()
    return tmac_amd


def test_single_rank_allgather(tm):
    import torch
    comm = tm.Comm(tm.Comm.unique_id(), 0, 1)
    src = torch.arange(4096, dtype=torch.float16, device="cuda")
    dst = torch.zeros_like(src)
    comm.allgather(src, dst, src.numel() * 2)
    torch.cuda.synchronize()
    assert torch.equal(src, dst)
    comm.destroy()


WORKER = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
rank, world, idfile = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
torch.cuda.set_device(rank)
import tmac_amd
from oracle import oracle as orc
tmac_amd.binding.check(tmac_amd.lib().tmac_hip_init(rank))
if rank == 0:
    uid = tmac_amd.Comm.unique_id()
    open(idfile + ".tmp", "wb").write(uid); os.rename(idfile + ".tmp", idfile)
else:
    import time
    while not os.path.exists(idfile): time.sleep(0.05)
    uid = open(idfile, "rb").read()
comm = tmac_amd.Comm(uid, rank, world)
Mw, K, bits, bm, kf, gs, ags = 1024, 2048, 2, 128, 16, 128, 64
case = orc.make_case(7, Mw, K, bits=bits, fp16_values=True)
A = orc.preprocess_weights(case["w"], bits, bm, kf); S = orc.preprocess_scales(case["sc"], case["zr"], bits, bm)
wr = tmac_amd.TMACGeMMWrapper(act_group_size=ags)
rows = Mw // world; tiles = rows * bits // bm; t0 = rank * tiles                      # this rank's tiles: contiguous in the reference layout
w = wr.register_weights(A[t0:t0 + tiles], S[t0:t0 + tiles], rows, K, bits, tmac_amd.KCfg.make(rows, K, bits, bm, kf, gs, ags, True))
x = torch.from_numpy(case["B"][0]).cuda().half()
part = torch.empty(rows, dtype=torch.float16, device="cuda")
wr.fused([w], x, [part], 1)
whole = torch.empty(Mw, dtype=torch.float16, device="cuda")
comm.allgather(part, whole, rows * 2)
torch.cuda.synchronize()
wfull = wr.register_weights(A, S, Mw, K, bits, tmac_amd.KCfg.make(Mw, K, bits, bm, kf, gs, ags, True))
ref = torch.empty(Mw, dtype=torch.float16, device="cuda")
wr.fused([wfull], x, [ref], 1)
torch.cuda.synchronize()
assert torch.equal(whole, ref), "gathered row shards differ from the single-GPU result"
comm.destroy()
print("rank", rank, "ok")
'''


def test_two_ranks_row_shards(tm):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's single-GPU test box has one)")
    with tempfile.TemporaryDirectory() as d:
        idfile = os.path.join(d, "id")
        script = os.path.join(d, "worker.py")
        open(script, "w").write(WORKER)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        ps = [subprocess.Popen([sys.executable, script, ROOT, str(r), "2", idfile], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
        outs = [p.communicate(timeout=300)[0].decode() for p in ps]
        assert all(p.returncode == 0 for p in ps), outs
