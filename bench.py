#!/usr/bin/env python3
"""bench.py — W2A8 LUT-GEMV throughput of the T-MAC hot path on MI355X (BASELINE.json metric).

Workload (config.workload = "llama-2-7b-w2a8-decode-all-layers", BASELINE.json configs[1]):
one "step" = one decoded token's worth of the hot path = for each of 32 layers
    preprocessor(x0)  -> q,k,v   3 x qgemm_lut (4096 x 4096)
    preprocessor(x1)  -> o       1 x qgemm_lut (4096 x 4096)
    preprocessor(x2)  -> gate,up 2 x qgemm_lut (11008 x 4096)
    preprocessor(x3)  -> down    1 x qgemm_lut (4096 x 11008)
with W2 weights (group 128, zero points, act group 64; python/t_mac/model_utils.py:27-32,
tools/run_pipeline.py:405-419), fp16 activations/scales/outputs, fp32 accumulation, every layer's
weights distinct (1.62 GB of 2-bit planes, > the 256 MB Infinity Cache).  The launches are chained by
real data (x1 = q, x2 = o, x3 = gate, next x0 = down) on one stream, as a decoder would issue them.
Synthetic data: uniform random weights, |N(0,1)|-shaped scales sized so activations stay O(1).

value = algorithmic bytes of the 224 GEMVs per step (SURVEY.md 8d formula) / step time, GB/s.

--gpus N > 1 (launched through torch.distributed.run, one rank per GPU, RCCL): weight ROWS are sharded
over the ranks (tile-aligned), the integer path needs no reduction, and the only exchange step is an
all-gather of each produced activation vector (fp16, 8-22 KB) before the next LUT build; total work is
fixed, so "scaling" is "strong".
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec
LAYERS = 32
# (name, Mw, K, count per layer, input slot)
MATS = [("qkv", 4096, 4096, 3, 0), ("o", 4096, 4096, 1, 1), ("gate_up", 11008, 4096, 2, 2), ("down", 4096, 11008, 1, 3)]
BITS, GS, AGS, BM, KF = 2, 128, 64, 128, 16


def algorithmic_bytes(Mw, K, bits=BITS, gs=GS, ags=AGS, zp=True, N=1):
    """weight planes + fp16 scales(/zeros) + int8 QLUT + fp16 LUT scales/biases + fp16 out"""
    return Mw * K * bits // 8 + Mw * (K // gs) * (2 if zp else 1) * 2 + N * (K // 4) * 16 + N * (K // ags) * 4 + N * Mw * 2


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--layers", type=int, default=LAYERS, help=argparse.SUPPRESS)   # debugging only
    ap.add_argument("--variant", type=int, default=0, help="GEMV kernel variant (0 auto, 1 mqsad, 2 sdwa)")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket the headline GEMV with events")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--path", choices=["chain", "fused", "split"], default="chain",
                    help="chain: the token's 128 fused calls recorded once and executed by ONE persistent launch "
                         "(k_decode_chain: in-kernel hand-off of the activation vectors, weights of the next call streaming "
                         "in behind the current one); fused: LUT build inside the GEMV kernel, q/k/v and gate/up batched "
                         "(4 launches/layer, hipGraph replay); split: preprocessor + one GEMV launch per matrix (11 launches/layer)")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the check of one layer's outputs (the launches being timed, at full size) against the oracle")
    ap.add_argument("--stamps", action="store_true", help="chain path: report per-call times from in-kernel s_memtime stamps")
    ap.add_argument("--autotune", action="store_true",
                    help="measure the kernel's launch configurations on this rank's shard shapes before the run instead of "
                         "trusting the built-in heuristic (tmac_hip_autotune_fused; outside the timed region)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--eager-collectives", action="store_true",
                    help="multi-GPU: launch eagerly instead of capturing the RCCL all-gathers into the hipGraph "
                         "(~47 us of host time per launch + collective, 6 ms per token)")
    ap.add_argument("--force-dist", action="store_true", help=argparse.SUPPRESS)   # debugging: take the multi-GPU code path with 1 rank
    return ap.parse_args()


def cpu_baseline(seconds=6.0):
    """The reference kernel (oracle/_ref, built from /root/reference by oracle/Makefile) — or, if that
    prebuilt file is absent, our scalar port — timed on this box's host cores over a bounded sample: the
    three W2 shapes of one llama-2-7B layer (one matrix each).  Tiles are split over threads with an OpenMP
    static schedule exactly as llama.cpp splits them (tmac_gemm_wrapper.h:197-199); best of >= 5 after a
    warm-up (deploy/benchmark.cc:36-45 method).  Test-infrastructure code used as a reported baseline."""
    import ctypes as C
    from oracle import oracle as orc
    cores = os.cpu_count() or 1
    rng = np.random.default_rng(0)
    shapes = [(4096, 4096), (11008, 4096), (4096, 11008)]
    setname = "aarch64-llama-2-7b-2bit"
    kind = "reference" if orc.have_ref(setname) else "port"
    drv = C.CDLL(os.path.join(ROOT, "oracle", "libref_driver.so")) if kind == "reference" else None
    work = []
    for Mw, K in shapes:
        M = Mw * BITS
        A = rng.integers(0, 256, size=(M // BM, K // 4, BM // 2), dtype=np.uint8)
        S = np.abs(rng.standard_normal((M // BM, K // GS, BM // BITS * 2))).astype(np.float32)
        Bv = rng.standard_normal((1, K)).astype(np.float32)
        work.append((Mw, K, A, S, Bv))
    total_bytes = sum(algorithmic_bytes(Mw, K) for Mw, K in shapes)

    def run_once(nthreads):
        t0 = time.perf_counter()
        for Mw, K, A, S, Bv in work:
            if kind == "reference":
                L = orc.ref_lib(setname)
                G = K // AGS
                ls = np.zeros(G, np.float32); lb = np.zeros(G, np.float32); q = np.zeros((K // 4, 16), np.int8)
                pre = getattr(L, f"preprocessor_t1_int8_m{8192 if Mw == 4096 else 22016}_k{K}_n1_b2")
                pre(orc._p(Bv), orc._p(ls), orc._p(lb), orc._p(q))
                qg = getattr(L, f"qgemm_lut_t1_int8_m{BM}_k{K}_n1_b2")
                ntiles = Mw * BITS // BM
                Cout = np.zeros(Mw, np.float32)
                rc = drv.ref_run_tiles_omp(C.cast(qg, C.c_void_p), orc._p(A), C.c_size_t(A[0].nbytes), orc._p(q), orc._p(S),
                                           C.c_size_t(S[0].size), orc._p(ls), orc._p(lb), orc._p(Cout),
                                           C.c_size_t(BM // BITS), ntiles, nthreads)
                assert rc == 0
            else:
                q, ls, lb = orc.preprocessor(Bv, AGS)
                orc.qgemm_float(A, q, S, ls, lb, Mw, K, 1, BITS, BM, KF, GS, AGS, True)
        return time.perf_counter() - t0

    out = {}
    thread_counts = sorted({1, min(cores, 8), min(cores, 32), min(cores, 64), cores}) if kind == "reference" else [1]
    for nthreads in thread_counts:
        run_once(nthreads)
        best, t_end, reps = 1e9, time.perf_counter() + seconds / len(thread_counts), 0
        while time.perf_counter() < t_end or reps < 5:
            best = min(best, run_once(nthreads)); reps += 1
        out[nthreads] = total_bytes / best / 1e9
    used = max(out, key=out.get)
    return {"value": round(out[used], 3), "unit": "GB/s", "cores": used, "kind": kind, "host_cores": cores,
            "by_threads_GBps": {str(k): round(v, 3) for k, v in out.items()},
            "sample": "one 4096x4096, one 11008x4096 and one 4096x11008 W2 g128 zp GEMV (preprocessor + all tiles), "
                      "OpenMP static tile split, best of >=5; value = best thread count"}


def main():
    args = parse()
    # The contract is ONE JSON line on stdout.  Libraries underneath write banners to file descriptor 1 (RCCL prints its
    # version block there under torchrun), so everything else that reaches fd 1 is sent to stderr and the original stdout
    # is kept for the result line alone.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} must be launched with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dist_on = world > 1 or args.force_dist
    if dist_on and args.path == "chain":
        args.path = "fused"            # the persistent chain is a single-GPU launch; row shards exchange through RCCL between launches
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"), RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import tmac_amd
    from tmac_amd import KCfg, F16
    L = tmac_amd.lib()
    tmac_amd.binding.check(L.tmac_hip_init(local_rank))
    tmac_amd.binding.check(L.tmac_hip_set_variant(args.variant))
    dev = torch.device("cuda", local_rank)
    gen = torch.Generator(device=dev); gen.manual_seed(1234)   # same weights on every rank, sliced by rank below
    wr = tmac_amd.TMACGeMMWrapper(act_group_size=AGS)
    wr.set_workspace(11008, 1)

    # ---- synthetic, row-sharded weights, registered (re-tiled on the GPU) once --------------------
    rpt = BM // BITS                                  # 64 output rows per reference tile
    layers = []
    host_l0 = {}
    bytes_per_step = 0
    shard_rows = {}
    for name, Mw, K, cnt, slot in MATS:
        ntiles = Mw // rpt
        tiles_per_rank = (ntiles + world - 1) // world          # ragged split -> padded with extra synthetic rows
        shard_rows[name] = tiles_per_rank * rpt
        bytes_per_step += cnt * algorithmic_bytes(Mw, K)
    bytes_per_step *= args.layers
    for li in range(args.layers):
        mats = {}
        for name, Mw, K, cnt, slot in MATS:
            Mloc = shard_rows[name]
            cfg = KCfg.make(Mloc, K, BITS, BM, KF, GS, AGS, True)
            c = 1.0 / np.sqrt(2.25 * K)           # E[((w - 1.5) s - n)^2] = (1.25 + 1) c^2 for w uniform in 0..3: unit gain per GEMV
            ws = []
            for _ in range(cnt):
                A = torch.randint(0, 256, (Mloc * BITS // BM, K // 4, BM // 2), dtype=torch.uint8, device=dev, generator=gen)
                S = (torch.randn((Mloc * BITS // BM, K // GS, rpt // 8, 2, 8), device=dev, generator=gen) * c)
                S[:, :, :, 0, :].abs_()
                # zero = (mean weight level - 2^(b-1)) * scale + noise: the real weight (w - 2^(b-1)) * scale - zero has mean 0, so
                # a common component of the activations is not amplified from layer to layer (it grew 16x per GEMV and the
                # chained vectors overflowed fp16 after a few layers)
                S[:, :, :, 1, :] += S[:, :, :, 0, :] * ((2 ** BITS - 1) / 2.0 - 2 ** (BITS - 1))
                S = S.half().contiguous()
                ws.append(tmac_amd.Weights(A, S, Mloc, K, BITS, cfg, scales_dtype=F16, dev_dtype=F16, on_device=True))
                if li == 0 and not args.no_verify:
                    host_l0.setdefault(name, []).append((A.cpu().numpy(), S.float().cpu().numpy().reshape(Mloc * BITS // BM, K // GS, -1)))
                del A, S
            mats[name] = ws
        layers.append(mats)
    torch.cuda.synchronize()

    # activations: x[slot] full vectors (fp16); per-matrix local outputs
    xdim = {0: 4096, 1: 4096, 2: 4096, 3: 11008}
    x = {s: torch.randn(xdim[s], device=dev, generator=gen).half() for s in xdim}
    outs = {name: [torch.empty(shard_rows[name], dtype=torch.float16, device=dev) for _ in range(cnt)]
            for name, Mw, K, cnt, slot in MATS}
    gathered = {name: torch.empty(shard_rows[name] * world, dtype=torch.float16, device=dev) for name, *_ in MATS}
    nxt = {"qkv": 1, "o": 2, "gate_up": 3, "down": 0}
    logical = {"qkv": 4096, "o": 4096, "gate_up": 11008, "down": 4096}
    ev_pairs = []
    use_ev = not args.no_kernel_events

    # launch-configuration tuning on this rank's shard shapes (one measurement per distinct matrix set; the table is keyed
    # by shape, so layer 0 stands for all layers).  Outside the timed region, like the reference's offline autotvm tuning.
    tuned = {}
    if args.path == "fused" and args.autotune and args.variant == 0:
        for name, Mw, K, cnt, slot in MATS:
            r = wr.autotune(layers[0][name], F16, F16)
            tuned[name] = [r["ft"], r["wpq"], round(r["us"], 2), round(r["heuristic_us"], 2)]
        torch.cuda.synchronize()

    fused_calls = args.path in ("chain", "fused")

    def step(record):
        for li in range(args.layers):
            mats = layers[li]
            for name, Mw, K, cnt, slot in MATS:
                if fused_calls:
                    wr.fused(mats[name], x[slot], outs[name], 1, act_dtype=F16, out_dtype=F16)
                else:
                    wr.llama_cpp_init(x[slot], Mw, K, 1, BITS, act_dtype=F16)
                    for i in range(cnt):
                        wr.llama_cpp_compute(mats[name][i], outs[name][i], 1, out_dtype=F16)
                # exchange step: the first output of the group becomes the next activation vector
                if dist_on:
                    dist.all_gather_into_tensor(gathered[name], outs[name][0])
                    x[nxt[name]] = gathered[name][:logical[name]]
                else:
                    x[nxt[name]] = outs[name][0]

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    # One step (128 launches + the all-gathers) is captured into a hipGraph and replayed.  With RCCL collectives inside,
    # capture was exercised with one rank only on the development box (1.27 ms per step against 6.0 ms eager): if
    # capture raises, the run falls back to eager launches; if the first replay does not finish, a watchdog ends the
    # process instead of hanging the node.
    use_graph = (not args.no_graph) and not (dist_on and args.eager_collectives) and args.path != "chain"
    graph = None
    chain = None
    if args.path == "chain":
        # record the token's calls once (they are not launched while recording); one launch per step from here on
        step(False)                                  # leaves x[0] = the down projection's output, as in a decode loop
        torch.cuda.synchronize()
        with wr.record_chain() as rec:
            step(False)
        chain = rec.chain
        if args.stamps:
            stamp_buf = torch.zeros(chain.nops * chain.grid * 8, dtype=torch.int64, device=dev)
            chain.set_stamps(stamp_buf)
    if use_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step(False)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            if dist_on:
                # ProcessGroupNCCL's watchdog thread polls the events of earlier collectives; under the default (global)
                # capture mode such a query from another thread invalidates the capture and kills the process (seen on the
                # development box).  Let it reap what has completed, then capture in thread-local mode, where only this
                # thread's calls are policed.
                time.sleep(1.0)
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    step(False)
            else:
                with torch.cuda.graph(graph):
                    step(False)
        except Exception as e:   # capture not supported with this RCCL / torch build: measured eagerly instead
            if not dist_on:
                raise
            sys.stderr.write(f"bench.py: graph capture with collectives failed ({e!r}); launching eagerly\n")
            graph = None
            use_graph = False
            try:
                torch.cuda.synchronize()
            except Exception:
                pass

    def first_step():
        # every rank runs exactly one step here whether its capture succeeded or not, so that the ranks stay aligned on
        # the number of collectives issued; a failed capture can leave a sticky HIP error behind, hence one retry
        if graph is not None:
            graph.replay()
        else:
            try:
                step(False)
            except tmac_amd.binding.TMACHipError:
                torch.cuda.synchronize()
                step(False)
        torch.cuda.synchronize()

    if dist_on:
        import threading
        done = threading.Event()

        def watchdog():
            if not done.wait(240.0):
                sys.stderr.write("bench.py: the first step with RCCL all-gathers did not complete; "
                                 "rerun with --eager-collectives\n")
                sys.stderr.flush()
                os._exit(3)
        threading.Thread(target=watchdog, daemon=True).start()
        first_step()
        done.set()

    def run_step():
        if chain is not None:
            chain.launch()
        elif graph is not None:
            graph.replay()
        else:
            step(False)

    for _ in range(args.warmup):
        run_step()
    barrier()
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        run_step()
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    ev_ms_per_step = ev0.elapsed_time(ev1) / args.steps      # hipEvent pair on the launch stream around the timed region
    if chain is not None and chain.status() != 0:
        raise SystemExit("bench.py: a hand-off inside the decode chain timed out; outputs invalid")
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3

    # ---- roofline of the dominant kernel: the GEMV on the headline shape (4096 x 11008 W2) ----------
    # hipEvent pair (on the launch stream) around back-to-back launches of that kernel over all layers'
    # distinct weights (32 x 11.3 MB > MALL), so the figure includes the inter-kernel boundary.
    roof = None
    traffic, traffic_src = None, None
    try:    # HBM bytes per launch of the dominant kernel from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE, x2 gfx950 correction)
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            tj = json.load(f)
        ent = tj.get("k_decode_chain" if args.path == "chain" else "k_gemv_quad_headline")
        if ent and (args.path != "chain" or ent.get("layers") == args.layers):
            traffic, traffic_src = ent["bytes_per_launch"], ent.get("source")
    except Exception:
        pass
    if args.path == "chain":
        ach = bytes_per_step / (ev_ms_per_step * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "k_decode_chain: one persistent launch per decoded token (all %d GEMVs, LUT builds and "
                                          "in-kernel hand-offs of the llama-2-7B W2 g128 zp layer stack)" % (7 * args.layers),
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": bytes_per_step, "avg_launch_us": round(ev_ms_per_step * 1e3, 2),
                "launches_timed": args.steps,
                "timing": "hipEvent pair on the launch stream around the %d timed launches" % args.steps}
        if args.stamps:
            raw = stamp_buf.cpu().numpy().reshape(chain.nops, chain.grid, 8)
            try:
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                np.save(os.path.join(ROOT, "gpurun_out", "chain_stamps.npy"), raw)
            except Exception:
                pass
            st = raw[:, :, :7].astype(np.float64) * 0.01                     # s_memrealtime: 100 MHz -> us (wave 0 of every workgroup)
            names = [m[0] for m in MATS]
            ends = st[:, :, 5].max(axis=1)                                    # a call is complete when its last row quad is published
            dur = ends - np.concatenate([[st[0, :, 0].min()], ends[:-1]])
            per = {}
            for k, name in enumerate(names):
                sel = st[k::4]
                per[name] = {"us": round(float(np.mean(dur[k::4])), 3),
                             "wait_input_us": round(float(np.mean(sel[:, :, 1] - sel[:, :, 0])), 3),
                             "lut_build_us": round(float(np.mean(sel[:, :, 2] - sel[:, :, 1])), 3),
                             "lookups_us": round(float(np.mean(sel[:, :, 5] - sel[:, :, 2])), 3),
                             "polls": round(float(np.mean(raw[k::4, :, 7])), 2)}
            hb = algorithmic_bytes(4096, 11008)
            roof["per_call_from_stamps"] = per
            roof["headline_gemv"] = {"shape": "4096x11008 W2 g128 zp (the down projection inside the launch)", "us": per["down"]["us"],
                                     "GBps": round(hb / (per["down"]["us"] * 1e-6) / 1e9, 1) if per["down"]["us"] > 0 else None,
                                     "frac": round(hb / (per["down"]["us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if per["down"]["us"] > 0 else None}
    elif use_ev:
        xin = torch.randn(11008, device=dev, generator=gen).half()
        wr.llama_cpp_init(xin, 4096, 11008, 1, BITS, act_dtype=F16)
        reps, skip = 10, 3            # the first replays run while the clocks settle after the timed region
        durs = []

        def headline_launches():
            for li in range(args.layers):
                if args.path == "fused":
                    wr.fused(layers[li]["down"], xin, outs["down"], 1, act_dtype=F16, out_dtype=F16)
                else:
                    wr.llama_cpp_compute(layers[li]["down"][0], outs["down"][0], 1, out_dtype=F16)

        rgraph = None
        if use_graph:   # same launch mechanism as the timed region: the 32 launches replayed from a hipGraph
            headline_launches()
            torch.cuda.synchronize()
            rgraph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(rgraph):
                headline_launches()
        for r in range(reps + skip):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            if rgraph is not None:
                rgraph.replay()
            else:
                headline_launches()
            e1.record()
            torch.cuda.synchronize()
            if r >= skip:
                durs.append(e0.elapsed_time(e1) * 1e-3 / args.layers)
        durs = np.array(durs)
        hb = algorithmic_bytes(shard_rows["down"], 11008)
        ach = hb / float(np.mean(durs)) / 1e9
        # floor of this launch structure, measured the same way: a kernel that only READS the same number of bytes
        # (distinct buffers per launch, > MALL in total) and an empty kernel, as dependent nodes of a replayed graph
        floor = None
        if use_graph:
            nb = (hb + 4095) // 4096 * 4096
            bufs = [torch.empty(nb, dtype=torch.uint8, device=dev).fill_(0x5a) for _ in range(args.layers)]
            sink = torch.zeros(4096, dtype=torch.uint8, device=dev)
            cs = torch.cuda.current_stream().cuda_stream

            def floor_time(nbytes):
                def launches():
                    for b in bufs:
                        tmac_amd.binding.check(L.tmac_hip_debug_stream_read(b.data_ptr(), nbytes, sink.data_ptr(), cs))
                launches(); torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    cs2 = torch.cuda.current_stream().cuda_stream
                    for b in bufs:
                        tmac_amd.binding.check(L.tmac_hip_debug_stream_read(b.data_ptr(), nbytes, sink.data_ptr(), cs2))
                ts = []
                for r in range(8):
                    f0 = torch.cuda.Event(enable_timing=True); f1 = torch.cuda.Event(enable_timing=True)
                    f0.record(); g.replay(); f1.record(); torch.cuda.synchronize()
                    if r >= 3:
                        ts.append(f0.elapsed_time(f1) * 1e-3 / len(bufs))
                return float(np.mean(ts))
            floor = {"pure_read_same_bytes_us": round(floor_time(hb // 16 * 16) * 1e6, 3),
                     "near_empty_launch_us": round(floor_time(4096) * 1e6, 3)}
            del bufs
            # and of the whole step: the same 4 launches per layer, each only reading its matrices' bytes (distinct buffers)
            sizes = [cnt * algorithmic_bytes(shard_rows[name], K) // 16 * 16 for name, Mw, K, cnt, slot in MATS]
            sbufs = [[torch.empty(sz, dtype=torch.uint8, device=dev).fill_(0x5a) for sz in sizes] for _ in range(args.layers)]

            def step_reads(stream):
                for lb_ in sbufs:
                    for b_, sz in zip(lb_, sizes):
                        tmac_amd.binding.check(L.tmac_hip_debug_stream_read(b_.data_ptr(), sz, sink.data_ptr(), stream))
            step_reads(cs); torch.cuda.synchronize()
            sg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(sg):
                step_reads(torch.cuda.current_stream().cuda_stream)
            ts = []
            for r in range(8):
                f0 = torch.cuda.Event(enable_timing=True); f1 = torch.cuda.Event(enable_timing=True)
                f0.record(); sg.replay(); f1.record(); torch.cuda.synchronize()
                if r >= 3:
                    ts.append(f0.elapsed_time(f1))
            floor["pure_read_step_ms"] = round(float(np.mean(ts)), 4)
            del sbufs, sg
        roof = {"bound": "hbm", "kernel": ("k_gemv_quad, LUT build fused" if args.path == "fused" else "k_gemv_quad, LUT prebuilt") + ", headline shape 4096x11008 W2 g128 zp", "achieved": round(ach, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": hb, "avg_launch_us": round(float(np.mean(durs)) * 1e6, 3),
                "min_launch_us": round(float(np.min(durs)) * 1e6, 3), "launches_timed": reps * args.layers,
                "launch_floor": floor,
                "timing": "hipEvent pair on the launch stream around %d back-to-back launches (distinct weights, %s), mean of 10" % (args.layers, "hipGraph replay" if use_graph else "eager")}

    # ---- verification (outside the timed region): the launches being timed, at full size, against the oracle ----------
    verified = None
    if not args.no_verify and world == 1 and host_l0:
        from oracle import oracle as orc
        vx = torch.randn(4096, device=dev, generator=gen).half()
        vouts = {name: [torch.zeros(Mw, dtype=torch.float16, device=dev) for _ in range(cnt)] for name, Mw, K, cnt, slot in MATS}
        vin = {"qkv": vx, "o": vouts["qkv"][0], "gate_up": vouts["o"][0], "down": vouts["gate_up"][0]}

        def vcalls():
            for name, Mw, K, cnt, slot in MATS:
                if fused_calls:
                    wr.fused(layers[0][name], vin[name], vouts[name], 1, act_dtype=F16, out_dtype=F16)
                else:
                    wr.llama_cpp_init(vin[name], Mw, K, 1, BITS, act_dtype=F16)
                    for i in range(cnt):
                        wr.llama_cpp_compute(layers[0][name][i], vouts[name][i], 1, out_dtype=F16)
        if args.path == "chain":
            with wr.record_chain() as vrec:
                vcalls()
            vrec.chain.launch()
            torch.cuda.synchronize()
            ok = vrec.chain.status() == 0
            vrec.chain.free()
        else:
            vcalls()
            torch.cuda.synchronize()
            ok = True
        worst = 0.0
        for name, Mw, K, cnt, slot in MATS:
            q, ls, lb = orc.preprocessor(vin[name].float().cpu().numpy()[None, :], AGS)
            for i in range(cnt):
                A, S = host_l0[name][i]
                ref = orc.qgemm_float(A, q, S, ls, lb, Mw, K, 1, BITS, BM, KF, GS, AGS, True)[0]
                got = vouts[name][i].float().cpu().numpy()
                worst = max(worst, float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)))
        verified = {"ok": bool(ok and worst <= 1e-3), "max_rel_err": float("%.3g" % worst), "tolerance": 1e-3,
                    "what": "layer 0's seven GEMVs (q/k/v, o, gate/up, down at full size, chained) through the timed path vs oracle/ (fp16 outputs)"}
        if not verified["ok"]:
            sys.stderr.write("bench.py: VERIFICATION FAILED: %r\n" % (verified,))

    if rank == 0:
        res = {
            "metric": "W2A8 GEMV GB/s (llama-2-7B all-layer decode, N=1)",
            "value": round(bytes_per_step / (ms_per_step * 1e-3) / 1e9, 2),
            "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "int8",
            "data": "synthetic",
            "tokens_per_s": round(1e3 / ms_per_step, 1),
            "frac_of_hbm_peak": round(bytes_per_step / (ms_per_step * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), 4),
            "config": {"workload": "llama-2-7b-w2a8-decode-all-layers", "layers": args.layers,
                       "gemv_per_step": 7 * args.layers, "launches_per_step": 1 if args.path == "chain" else (4 if args.path == "fused" else 11) * args.layers, "path": args.path,
                       "autotune": tuned,
                       "algorithmic_bytes_per_step": bytes_per_step, "weights": "W2 g128 zero-point, act_group 64",
                       "parallelism": f"row-shard x{world}" if world > 1 else "single GPU",
                       "kernel_variant": args.variant, "launch": "one persistent launch per step" if args.path == "chain" else ("hipGraph replay" if use_graph else "eager")},
            "roofline": roof,
            "verified": verified,
            "event_ms_per_step": round(ev_ms_per_step, 4),
            "cpu_baseline": None,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the baseline is a reported extra, never a reason to lose the GPU number
                res["cpu_baseline"] = {"error": repr(e)}
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(res) + "\n").encode())
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
