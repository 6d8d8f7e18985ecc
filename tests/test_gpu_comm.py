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


IPC_WORKER = r'''
import os, sys, time
root, rank, world, d = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
sys.path.insert(0, root)
import numpy as np, torch
import tmac_amd
from oracle import oracle as orc
tm = tmac_amd
tm.binding.check(tm.lib().tmac_hip_init(0))                  # both ranks on device 0 (RCCL refuses that; IPC windows do not)
bits, bm, kf, gs, ags, N = 2, 128, 16, 128, 64, 48
OPS = [(1024, 2048), (2048, 1024)]                           # (K, Mw): the second mpGEMM consumes the gathered block of the first

def barrier(tag):
    open(os.path.join(d, f"{tag}.{rank}"), "w").close()
    t0 = time.time()
    while not all(os.path.exists(os.path.join(d, f"{tag}.{r}")) for r in range(world)):
        if time.time() - t0 > 120: raise SystemExit("barrier timeout " + tag)
        time.sleep(0.002)

wr = tm.TMACGeMMWrapper(act_group_size=ags)
host, shard, full = [], [], []
for i, (K, Mw) in enumerate(OPS):
    case = orc.make_case(300 + i, Mw, K, N=N, bits=bits, fp16_values=True)
    c = 1.0 / np.sqrt(2.5 * K)
    sc = (case["sc"] * c).astype(np.float16).astype(np.float32)
    zr = (case["zr"] * c + ((2 ** bits - 1) / 2.0 - 2 ** (bits - 1)) * sc).astype(np.float16).astype(np.float32)
    A = orc.preprocess_weights(case["w"], bits, bm, kf); S = orc.preprocess_scales(sc, zr, bits, bm)
    rows = Mw // world; tiles = rows * bits // bm; t0 = rank * tiles
    shard.append(wr.register_weights(A[t0:t0 + tiles], S[t0:t0 + tiles], rows, K, bits, tm.KCfg.make(rows, K, bits, bm, kf, gs, ags, True, -1, N), dev_dtype=tm.F16))
    full.append(wr.register_weights(A, S, Mw, K, bits, tm.KCfg.make(Mw, K, bits, bm, kf, gs, ags, True, -1, N), dev_dtype=tm.F16))
    host.append((A, S, case))
x0 = torch.from_numpy(host[0][2]["B"]).cuda().half()          # [N][K0]
# ---- the communicator: windows exported through files
maxb = max(N * (Mw // world) * 2 for K, Mw in OPS)
comm = tm.Comm.ipc(maxb, rank, world)
open(os.path.join(d, f"blob.{rank}.tmp"), "wb").write(comm.export()); os.rename(os.path.join(d, f"blob.{rank}.tmp"), os.path.join(d, f"blob.{rank}"))
barrier("exported")
comm.connect([open(os.path.join(d, f"blob.{r}"), "rb").read() for r in range(world)])
res = []
for rep in range(3):                                           # three rounds: the window halves and the generation flags are reused
    x = x0
    outs = []
    for i, (K, Mw) in enumerate(OPS):
        rows = Mw // world
        part = torch.empty((N, rows), dtype=torch.float16, device="cuda")
        wr.fused([shard[i]], x, [part], N, act_dtype=tm.F16)
        g = torch.empty((world, N, rows), dtype=torch.float16, device="cuda")
        comm.allgather(part, g, N * rows * 2)
        x = g.permute(1, 0, 2).reshape(N, Mw).contiguous()    # the [N][Mw] block every rank's next LUT build needs whole
        outs.append(x)
    torch.cuda.synchronize()
    assert comm.status() == 0, f"rank {rank}: a part did not arrive (rep {rep})"
    res.append(outs)
    barrier(f"rep{rep}")
# ---- the unsharded computation in this process, and the oracle
x = x0
for i, (K, Mw) in enumerate(OPS):
    ref = torch.empty((N, Mw), dtype=torch.float16, device="cuda")
    wr.fused([full[i]], x, [ref], N, act_dtype=tm.F16)
    torch.cuda.synchronize()
    for rep in range(3):
        assert torch.equal(res[rep][i], ref), f"rank {rank} op {i} rep {rep}: gathered row shards differ from the unsharded result"
    A, S, case = host[i]
    q, ls, lb = orc.preprocessor(x.float().cpu().numpy(), ags)
    want = orc.qgemm_float(A, q, S, ls, lb, Mw, K, N, bits, bm, kf, gs, ags, True)
    got = ref.float().cpu().numpy()
    assert float(np.abs(got - want).max() / np.abs(want).max()) <= 1e-3
    x = ref
comm.destroy()
print("rank", rank, "ok")
'''


def test_prefill_exchange_over_ipc_two_processes_one_device(tm):
    """BASELINE configs[4]'s exchange step (N > 1: the [N][rows] block of every rank gathered before the next LUT build) through the
    IPC transport of the communicator, two PROCESSES on one device: gathered row shards bit-identical to the unsharded mpGEMM (rows
    are independent, K is never split) and within 1e-3 of the oracle, three rounds (window halves and generation flags reused)."""
    with tempfile.TemporaryDirectory() as d:
        script = os.path.join(d, "worker.py")
        open(script, "w").write(IPC_WORKER)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        ps = [subprocess.Popen([sys.executable, script, ROOT, str(r), "2", d], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
        outs = []
        for p in ps:
            try:
                outs.append(p.communicate(timeout=300)[0].decode())
            except subprocess.TimeoutExpired:
                p.kill()
                outs.append("TIMEOUT\n" + p.communicate()[0].decode())
        assert all(p.returncode == 0 for p in ps), "\n".join(o[-3000:] for o in outs)


@pytest.mark.parametrize("workload,extra,nranks", [("llama2-7b-w2", [], 2), ("bitnet-3b", [], 2), ("llama2-7b-w2-prefill", ["--comm", "ipc"], 2),
                                                   ("llama2-7b-w2", [], 4), ("llama2-7b-w2", [], 8), ("llama2-7b-w2-prefill", ["--comm", "ipc"], 4),
                                                   ("llama2-7b-w2", ["--pattern", "independent"], 2), ("llama2-7b-w2", ["--pattern", "independent"], 8),
                                                   ("bitnet-3b", ["--pattern", "independent"], 4)])
def test_bench_ranks_share_the_device(tm, workload, extra, nranks):
    """bench.py's N > 1 orchestration end to end, launched exactly as the driver launches it (torch.distributed.run, one rank per
    "GPU"), with both ranks on the one device of the test box (--share-device: gloo instead of RCCL for bootstrap and timing, the CUs
    divided between the ranks' persistent kernels): row-sharded weights, the recorded exchange steps, IPC export / connect of the
    hand-off arenas over torch.distributed, the two trial launches, the timed launches, ONE JSON line from rank 0 -- with 2, 4 and 8
    ranks (ragged tile splits: gate / up's 172 reference tiles over 8 ranks).  Decode: the
    row-sharded chain (the mode fails instead of falling back); prefill: the IPC transport of the exchange step."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29600 + (os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--share-device", "--workload", workload, "--layers", "2", "--steps", "5", "--warmup", "2",
           "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:] + "\n" + r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == nranks and d["value"] > 0 and d["steps"] == 5
    assert d["config"]["parallelism"] == f"row-shard x{nranks}"
    # the preflight: one step of the timed path reproduced by every rank's communication-free emulation (same bits: same launch configuration)
    independent = "independent" in extra
    assert d["preflight"]["ok"], d["preflight"]
    if independent:
        # row-sharded STREAM mode: every rank streams its shard of the independent calls, nothing is exchanged; the
        # quarter-walk form's per-group-scale outputs are within the tolerance of the emulation, unified-scale ones bit-identical
        assert d["config"]["path"] == "chain" and d["config"]["pattern"] == "independent", d["config"]
        assert "stream" in d["roofline"]["kernel"] or "k_gemv_stream" in d["roofline"]["kernel"], d["roofline"]["kernel"]
        assert d["preflight"]["max_rel_diff"] <= 2e-3, d["preflight"]
        return
    if not workload.endswith("prefill"):
        # the default decode line of N ranks: value = the token's calls as independent calls on every rank's row shard (no exchange: the mode
        # that scales), and beside it the dependent chain with its in-kernel exchange over IPC-mapped arenas, preflight and all
        assert d["config"]["path"] == "chain" and d["config"]["pattern"] == "independent", d["config"]
        assert d["preflight"]["max_rel_diff"] <= 2e-3, d["preflight"]
        d = dict(d["dependent_chain"], prefill_scaling_headline=d.get("prefill_scaling_headline"), activations_finite=d.get("activations_finite", True),
                 config={"path": "chain"})
        assert "error" not in d, d
        assert d["n_gpus"] == nranks and d["value"] > 0
    assert d["preflight"]["bit_identical_on_every_rank"], d["preflight"]
    if not workload.endswith("prefill"):
        assert d["config"]["path"] == "chain", d["config"]
        assert d.get("activations_finite", True)
        # ... and a multi-GPU decode run reports the prefill twin of the same matrices as its scaling headline
        h = d["prefill_scaling_headline"]
        assert "error" not in h and h["value"] > 0 and h["n_gpus"] == nranks and h["workload"].endswith("prefill-256"), h
