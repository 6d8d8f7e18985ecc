"""N > 1 host path on CPU: two gloo ranks own tile-aligned row shards of one GEMV, compute them
(the oracle stands in for the GPU kernel here — this test is about the sharding math and the
exchange step), all-gather the outputs and must reproduce the single-rank result bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, Mw, K, bits, bm, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    import tmac_amd.sharding as sh
    case = orc.make_case(3, Mw, K, bits=bits)
    A = orc.preprocess_weights(case["w"], bits, bm, 16)
    S = orc.preprocess_scales(case["sc"], case["zr"], bits, bm)
    shard = sh.plan_row_shards(Mw, bits, bm, world)[rank]
    (wb, we), (sb, se) = sh.shard_blob_ranges(shard, K, bits, bm, 128, True, 4)
    if shard.rows:       # (a rank past the ragged tail owns no tile: it still takes part in the equal-size exchange)
        A_loc = np.frombuffer(A.tobytes()[wb:we], np.uint8).reshape(shard.tile_count, K // 4, bm // 2)
        S_loc = np.frombuffer(S.tobytes()[sb:se], np.float32).reshape(shard.tile_count, K // 128, -1)
    # every rank builds the LUT from the (already gathered) activation vector: replicated preprocessing
    qlut, ls, lb = orc.preprocessor(case["B"], 64)
    out = np.zeros((1, shard.padded_rows), np.float32)
    if shard.rows:
        out[:, :shard.rows] = orc.qgemm_float(A_loc, qlut, S_loc, ls, lb, shard.rows, K, 1, bits, bm, 16, 128, 64, True)
    # exchange step: equal-size (padded) all-gather; ranks with fewer tiles contribute zeros past their rows
    gathered = [torch.empty_like(torch.from_numpy(out)) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(out))
    if rank == 0:
        ref = orc.qgemm_float(A, qlut, S, ls, lb, Mw, K, 1, bits, bm, 16, 128, 64, True)
        shards = sh.plan_row_shards(Mw, bits, bm, world)
        got = np.concatenate([g.numpy()[:, :s_.rows] for s_, g in zip(shards, gathered)], -1)
        q.put((np.array_equal(got.view(np.uint32), ref.view(np.uint32)), got.shape, ref.shape))
    dist.barrier()
    dist.destroy_process_group()


# 8 tiles (even) / 11 tiles (ragged) over 2 ranks; 11 tiles over 4 ranks (3, 3, 3, 2); 3 tiles over 4 ranks (one rank owns nothing)
@pytest.mark.parametrize("Mw,K,bits,bm,world", [(512, 1024, 2, 128, 2), (704, 512, 2, 128, 2), (704, 512, 2, 128, 4), (192, 512, 2, 128, 4)])
def test_row_sharding_matches_single_rank(Mw, K, bits, bm, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, Mw, K, bits, bm, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok, gs, rs = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert gs == rs and ok


def _worker_prefill(rank, world, port, Mw, K, N, bits, bm, q):
    """two chained mpGEMMs with N activation rows (BASELINE configs[4] in small): rank r owns row shards of both weight
    matrices; the exchange step is bench.py's -- all-gather of the [N][padded rows] block into [world][N][rows], permuted to
    [N][world * rows] and trimmed to the logical width, which is the next call's activation block"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    import tmac_amd.sharding as sh
    GS = 64                                  # weight group 64: a ragged width (an odd number of 64-row tiles) can be the next call's K

    def mats(seed, rows, cols):
        case = orc.make_case(seed, rows, cols, N=N, bits=bits, gs=GS)
        case["sc"] *= np.float32(1.0 / np.sqrt(2.5 * cols))
        case["zr"] = (case["zr"] * np.float32(1.0 / np.sqrt(2.5 * cols)) - np.float32(0.5) * case["sc"]).astype(np.float32)
        return case, orc.preprocess_weights(case["w"], bits, bm, 16), orc.preprocess_scales(case["sc"], case["zr"], bits, bm)

    def gemm(A, S, x, rows, cols):
        qlut, ls, lb = orc.preprocessor(np.ascontiguousarray(x, np.float32), 64)
        return orc.qgemm_float(A, qlut, S, ls, lb, rows, cols, N, bits, bm, 16, GS, 64, True)

    def sharded(A, S, x, rows, cols):
        shard = sh.plan_row_shards(rows, bits, bm, world)[rank]
        (wb, we), (sb, se) = sh.shard_blob_ranges(shard, cols, bits, bm, GS, True, 4)
        out = np.zeros((N, shard.padded_rows), np.float32)
        if shard.rows:
            A_loc = np.frombuffer(A.tobytes()[wb:we], np.uint8).reshape(shard.tile_count, cols // 4, bm // 2)
            S_loc = np.frombuffer(S.tobytes()[sb:se], np.float32).reshape(shard.tile_count, cols // GS, -1)
            out[:, :shard.rows] = gemm(A_loc, S_loc, x, shard.rows, cols)
        pieces = [torch.empty((N, shard.padded_rows), dtype=torch.float32) for _ in range(world)]
        dist.all_gather(pieces, torch.from_numpy(out))
        g = torch.stack(pieces)                                                          # [world][N][rows], as all_gather_into_tensor fills it
        return g.permute(1, 0, 2).reshape(N, -1)[:, :rows].contiguous().numpy()        # bench.py's trim to the logical width

    c1, A1, S1 = mats(11, Mw, K)
    c2, A2, S2 = mats(12, K, Mw)             # consumes the first call's [N][Mw] block
    x0 = c1["B"]
    y1 = sharded(A1, S1, x0, Mw, K)
    y2 = sharded(A2, S2, y1, K, Mw)
    if rank == 0:
        r1 = gemm(A1, S1, x0, Mw, K)
        r2 = gemm(A2, S2, r1, K, Mw)
        q.put((np.array_equal(y1.view(np.uint32), r1.view(np.uint32)) and np.array_equal(y2.view(np.uint32), r2.view(np.uint32)), y2.shape, r2.shape))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("Mw,K,N,bits,bm", [(512, 256, 5, 2, 128), (704, 128, 3, 2, 128)])    # even / ragged tile split
def test_two_rank_prefill_block_exchange(Mw, K, N, bits, bm):
    """world size 2, N > 1: the row-sharded mpGEMM pair with the activation-block exchange reproduces the single-rank result
    bit for bit (no reduction anywhere: K is never split)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_prefill, args=(r, 2, port, Mw, K, N, bits, bm, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, gs, rs = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert gs == rs and ok


def test_shard_plan_covers_every_tile_once():
    import tmac_amd.sharding as sh
    for Mw, bits, bm, world in [(4096, 2, 128, 8), (11008, 2, 128, 8), (11008, 2, 128, 4), (4096, 4, 256, 3), (3200, 2, 320, 4)]:
        shards = sh.plan_row_shards(Mw, bits, bm, world)
        assert sum(s.rows for s in shards) == Mw
        assert all(s.padded_rows == shards[0].padded_rows for s in shards)
        cur = 0
        for s in shards:
            assert s.row_begin == min(cur, Mw) and s.tile_count >= 0
            cur += s.rows
