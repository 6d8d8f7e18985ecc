#!/usr/bin/env python3
"""bench.py — throughput of T-MAC's LUT mpGEMM hot path on MI355X (BASELINE.json metric and configs).

Default workload (config.workload = "llama-2-7b-w2a8-decode-all-layers", BASELINE.json configs[1]): one "step" = one
decoded token's worth of the hot path = for each of 32 layers
    preprocessor(x0)  -> q,k,v   3 x qgemm_lut (4096 x 4096)
    preprocessor(x1)  -> o       1 x qgemm_lut (4096 x 4096)
    preprocessor(x2)  -> gate,up 2 x qgemm_lut (11008 x 4096)
    preprocessor(x3)  -> down    1 x qgemm_lut (4096 x 11008)
with W2 weights (group 128, zero points, act group 64; python/t_mac/model_utils.py:27-32, tools/run_pipeline.py:405-419),
fp16 activations/scales/outputs, fp32 accumulation, every layer's weights distinct (1.62 GB of 2-bit planes, > the 256 MB
Infinity Cache).  The calls are chained by real data (x1 = q, x2 = o, x3 = gate, next x0 = down), as a decoder issues them.
Synthetic data: uniform random weights, |N(0,1)|-shaped scales and zero points sized so that the chained activations stay O(1).

--workload selects the other BASELINE configurations (same JSON schema, each with its own roofline):
    llama2-7b-w4          configs[2]: the same shapes with 4-bit GPTQ-style weights (zero points), N = 1
    bitnet-3b             configs[3]: BitNet-b1.58-3B shapes (3200/8640), ternary in 2 bits, one scale, act group = K, N = 1
    llama2-7b-w2-prefill  configs[4]: the W2 shapes at N = 256 (plane-combined one-hot MFMA GEMM; roofline bound "mfma")
    llama2-7b-w4-prefill  the W4 shapes at N = 256
    bitnet-3b-prefill     the BitNet shapes at N = 256 (k_gemm_planes_us: int32 totals over the whole K, scale-final once per output)

--path: chain = the step's fused calls recorded once and executed by ONE persistent launch (k_decode_chain; default where
the configuration is covered); fused = one launch per fused call, replayed as a hipGraph; split = preprocessor + one
launch per matrix.  value = algorithmic bytes of the step's GEMVs (SURVEY.md 8d formula) / step time, GB/s (decode), or
tokens/s (prefill).

--gpus N > 1 (launched through torch.distributed.run, one rank per GPU, RCCL): weight ROWS are sharded over the ranks
(tile-aligned), the integer path needs no reduction, and the only exchange step is an all-gather of each produced activation
block before the next LUT build; total work is fixed, so "scaling" is "strong".
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

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_I8_PEAK_TOPS = 4404.0   # cdna_hip_programming.md: v_mfma_i32_32x32x32_i8 microbenchmark ceiling (the instruction k_gemm_planes issues; 16x16x64: 3944)
KF = 16

LLAMA = [("qkv", 4096, 4096, 3, 0), ("o", 4096, 4096, 1, 1), ("gate_up", 11008, 4096, 2, 2), ("down", 4096, 11008, 1, 3)]
BITNET = [("qkv", 3200, 3200, 3, 0), ("o", 3200, 3200, 1, 1), ("gate_up", 8640, 3200, 2, 2), ("down", 3200, 8640, 1, 3)]
# (name, Mw, K, count per layer, input slot); python/t_mac/model_utils.py:27-54
WORKLOADS = {
    "llama2-7b-w2": dict(tag="llama-2-7b-w2a8-decode-all-layers", mats=LLAMA, layers=32, bits=2, bm=128, gs=128, ags=64, zp=True, mg=-1, N=1,
                         metric="W2A8 GEMV GB/s (llama-2-7B all-layer decode, N=1)", weights="W2 g128 zero-point, act_group 64"),
    "llama2-7b-w4": dict(tag="llama-2-7b-w4a16-decode-all-layers", mats=LLAMA, layers=32, bits=4, bm=256, gs=128, ags=64, zp=True, mg=-1, N=1,
                         metric="W4 (GPTQ-style) GEMV GB/s (llama-2-7B all-layer decode, N=1)", weights="W4 g128 zero-point, act_group 64"),
    "bitnet-3b": dict(tag="bitnet-b1.58-3b-decode-all-layers", mats=BITNET, layers=26, bits=2, bm=128, gs=0, ags=0, zp=False, mg=1, N=1,
                      metric="W1.58A8 GEMV GB/s (BitNet-b1.58-3B all-layer decode, N=1)", weights="ternary in 2 bits, one scale, act_group = K (int32 path)"),
    "llama2-7b-w2-prefill": dict(tag="llama-2-7b-w2a8-prefill-256", mats=LLAMA, layers=32, bits=2, bm=128, gs=128, ags=64, zp=True, mg=-1, N=256,
                                 metric="W2A8 prefill tokens/s (llama-2-7B all-layer mpGEMM, N=256)", weights="W2 g128 zero-point, act_group 64"),
    "bitnet-3b-prefill": dict(tag="bitnet-b1.58-3b-prefill-256", mats=BITNET, layers=26, bits=2, bm=128, gs=0, ags=0, zp=False, mg=1, N=256,
                              metric="W1.58A8 prefill tokens/s (BitNet-b1.58-3B all-layer mpGEMM, N=256)", weights="ternary in 2 bits, one scale, act_group = K (int32 path)"),
    "llama2-7b-w4-prefill": dict(tag="llama-2-7b-w4a16-prefill-256", mats=LLAMA, layers=32, bits=4, bm=256, gs=128, ags=64, zp=True, mg=-1, N=256,
                                 metric="W4 (GPTQ-style) prefill tokens/s (llama-2-7B all-layer mpGEMM, N=256)", weights="W4 g128 zero-point, act_group 64"),
}


def algorithmic_bytes(Mw, K, bits, gs, ags, zp, mg, N=1):
    """weight planes + fp16 scales(/zeros) + int8 QLUT + fp16 LUT scales/biases + fp16 out"""
    sc = mg * 4 if mg >= 1 else Mw * (K // gs) * (2 if zp else 1) * 2
    g = 1 if mg >= 1 else K // ags
    return Mw * K * bits // 8 + sc + N * (K // 4) * 16 + N * g * 4 + N * Mw * 2


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="llama2-7b-w2")
    ap.add_argument("--layers", type=int, default=None, help=argparse.SUPPRESS)   # debugging only
    ap.add_argument("--variant", type=int, default=0, help="GEMV kernel variant (0 auto, 1 mqsad, 2 sdwa)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--path", choices=["auto", "chain", "fused", "split"], default="auto",
                    help="auto: chain where k_decode_chain covers the configuration (N = 1; row-sharded over the ranks with --gpus N), else fused")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the check of one layer's outputs (the launches being timed, at full size) against the oracle")
    ap.add_argument("--pattern", choices=["auto", "chained", "independent", "decoder"], default="auto",
                    help="independent: SURVEY 8(d)'s measurement -- the token's GEMVs back to back over distinct weights, every call reading a vector "
                         "that is resident before the launch (LUT builds included): the timed workload of a decode line since round 6; chained: "
                         "the calls linked by real data (x1 = q, x2 = o, x3 = gate, next x0 = down: every call waits for its predecessor inside one "
                         "persistent launch) -- rounds 1-5 timed this stricter form, a default run still measures it and reports it as "
                         "`dependent_chain`; auto (default): independent + the extras of the default line; "
                         "decoder: the calls as a decoder issues them -- residual add + RMSNorm and silu(gate) * up inside the chain "
                         "(tmac_hip_chain_xform), one launch per segment o -> gate/up -> down -> next q/k/v, an outside kernel (stand-in for "
                         "attention) between the segments; hipGraph replay of the token's launches")
    ap.add_argument("--no-decoder-pattern", action="store_true", help="skip the decoder-pattern measurement of the default line")
    ap.add_argument("--stream-calls", type=int, default=96, help="independent GEMVs per launch of roofline.stream_core / stream_by_shape (distinct weight sets)")
    ap.add_argument("--clock-ramp-ms", type=float, default=100.0, help="decode workloads: GPU milliseconds of the step run untimed in front of the warm-up steps (0: off)")
    ap.add_argument("--no-stream-core", action="store_true", help="skip roofline.stream_core (profiling passes: keeps the kernel's statistics to the timed launches)")
    ap.add_argument("--stamps", action="store_true", help="chain path: report per-call times from in-kernel stamps (costs ~6 %%; needs a profiling build of the library: tools/build_variant.sh st \"-DTMAC_CHAIN_STAMPS=1\", TMAC_HIP_LIB=tmac_amd/lib/ko/libtmac_hip_st.so)")
    ap.add_argument("--floors", action="store_true", help="fused path: also time launches that only read the same bytes")
    ap.add_argument("--autotune", action="store_true", help="fused path: measure the launch configurations first (tmac_hip_autotune_fused)")
    ap.add_argument("--no-graph", action="store_true", help="fused/split: launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--eager-collectives", action="store_true", help="multi-GPU: do not capture the RCCL all-gathers into the hipGraph")
    ap.add_argument("--comm", choices=["torch", "lib", "ipc"], default="torch",
                    help="multi-GPU exchange step: torch.distributed (ProcessGroupNCCL = RCCL; default, the path exercised so far), the "
                         "library's own communicator (tmac_hip_comm_*: RCCL through the C-ABI, bootstrapped over torch.distributed), or its "
                         "IPC transport (windows mapped by every peer, no RCCL; tests/test_gpu_comm.py runs it with two processes on one device)")
    ap.add_argument("--no-prefill-headline", action="store_true", help="decode runs: skip the N = 256 prefill measurement of the same matrices reported next to the decode line (prefill_twin on one GPU, prefill_scaling_headline on several)")
    ap.add_argument("--force-dist", action="store_true", help=argparse.SUPPRESS)   # debugging: take the multi-GPU code path with 1 rank
    ap.add_argument("--share-device", action="store_true",
                    help="test mode (tests/test_gpu_comm.py): every rank uses device 0 -- RCCL refuses that, so torch.distributed runs on gloo with host "
                         "tensors (bootstrap and timing only) and the row-sharded chain's workgroups are divided between the ranks; exercises the whole "
                         "N > 1 orchestration (row shards, recorded exchange steps, IPC export / connect, trial launches) on a one-GPU box.  Times mean nothing.")
    ap.add_argument("--gemm-kernel", type=int, default=0, choices=[0, 1, 2, 3],
                    help="prefill A/B: tmac_hip_debug_gemm_kernel (0 auto, 1 k_gemm_onehot, 2 / 3 k_gemm_planes with eight- / four-wave workgroups)")
    a = ap.parse_args()
    wl = WORKLOADS[a.workload]
    if a.steps is None:
        a.steps = 1000 if wl["N"] == 1 else 50      # ~0.7 s / ~0.4 s of timed GPU work: long enough for an outside observer (rocm-smi sampling) to see
    if a.warmup is None:
        a.warmup = 20 if wl["N"] == 1 else 5
    if a.layers is None:
        a.layers = wl["layers"]
    return a


CPU_SETS = {
    # workload -> (prebuilt set under deploy/tuned/, bits, bm, [(Mw, K, preprocessor M key)], shapes' sample text)
    "llama2-7b-w2": ("aarch64-llama-2-7b-2bit", 2, 128, [(4096, 4096, 8192), (11008, 4096, 22016), (4096, 11008, 8192)]),
    # the set's 4096 x 4096 kernel is the bm = 1024 one, whose stack accumulator is sized for fp16 and overflows with float_type = float
    # (SURVEY.md 8c): the set's bm = 256 kernel of the same K serves both K = 4096 shapes
    "llama2-7b-w4": ("aarch64-llama-2-7b-4bit", 4, 256, [(4096, 4096, 16384), (11008, 4096, 44032), (4096, 11008, 16384)]),
}


def _usable_cores():
    """host threads this process may run on (affinity mask; the largest OpenMP team below additionally stops at 128: a 256-thread team
    on a container's share of a 256-thread host collapsed to 0.1 GB/s in rounds 4 / 5)"""
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except Exception:
        return os.cpu_count() or 1


def _cpu_quota():
    """CPUs' worth of time the container's cgroup grants (cpu.max / cfs quota), or None: reported with the baseline -- a team larger than
    the quota still measured faster here (216 GB/s with 64 threads against 175 with 16 under a quota of 16), so it does not cap the list"""
    for quota_path, period_path in (("/sys/fs/cgroup/cpu.max", None), ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
        try:
            with open(quota_path) as f:
                parts = f.read().split()
            if period_path is None:
                q, per = parts[0], parts[1]
            else:
                q = parts[0]
                with open(period_path) as f:
                    per = f.read().split()[0]
            return None if q == "max" or int(q) <= 0 else round(int(q) / int(per), 2)
        except Exception:
            continue
    return None


def cpu_baseline(workload="llama2-7b-w2", seconds=6.0):
    """The reference's own CPU code (oracle/_ref, built from /root/reference by oracle/Makefile) — or, if that prebuilt file is
    absent, our scalar port — timed on this box's host cores over a bounded sample: one matrix of each of the three shapes of
    a layer.  Tiles are split over threads with an OpenMP static schedule exactly as llama.cpp splits them
    (tmac_gemm_wrapper.h:197-199); best of >= 5 after a warm-up (deploy/benchmark.cc:36-45 method).  llama W2 / W4: the
    checked-in prebuilt kernels (deploy/tuned/aarch64-llama-2-7b-{2,4}bit/kernels.cc).  BitNet: the int32 intrinsic
    (tbl_g4_int8_int32_update, what the reference selects on x86) in the generated glue's loop + the scale-final expression.
    Test-infrastructure code used as a reported baseline."""
    import ctypes as C
    from oracle import oracle as orc
    cores = _usable_cores()
    rng = np.random.default_rng(0)
    wl = WORKLOADS[workload]
    bitnet = wl["mg"] >= 1
    GS, AGS = 128, 64
    if bitnet:
        BITS, BM = 2, 128
        shapes = [(3200, 3200, 0), (8640, 3200, 0), (3200, 8640, 0)]
        kind = "reference" if orc.have_ref("intrins") else "port"
        what = "tbl_g4_int8_int32_update + lut_ctor intrinsics (python/t_mac/intrins) in the generated glue's loops, scale-final glue restated in oracle/ref_driver.c"
    else:
        setname, BITS, BM, shapes = CPU_SETS[workload]
        kind = "reference" if orc.have_ref(setname) else "port"
        what = f"prebuilt kernels deploy/tuned/{setname}/kernels.cc"
    drv = C.CDLL(os.path.join(ROOT, "oracle", "libref_driver.so")) if kind == "reference" else None
    # SURVEY 8(d) asks for rotating distinct weights, as the GPU side has them: one set of the layer's three shapes is 4-27 MB and would
    # stay in the host's L3 when re-run (225 GB/s was a cache number).  >= 512 MB of distinct sets are walked round-robin,
    # so every run streams its weights from DRAM; the cache-resident figure (set 0 re-run) is reported beside it.
    def make_set():
        ws_ = []
        for Mw, K, pm in shapes:
            M = Mw * BITS
            A = rng.integers(0, 256, size=(M // BM, K // 4, BM // 2), dtype=np.uint8)
            S = (np.array([1.0 / K], np.float32) if bitnet else
                 np.abs(rng.standard_normal((M // BM, K // GS, BM // BITS * 2))).astype(np.float32))
            Bv = rng.standard_normal((1, K)).astype(np.float32)
            ws_.append((Mw, K, pm, A, S, Bv))
        return ws_
    total_bytes = sum(algorithmic_bytes(Mw, K, BITS, GS, K if bitnet else AGS, not bitnet, 1 if bitnet else -1) for Mw, K, _ in shapes)
    nsets = max(2, -(-(512 << 20) // total_bytes)) if kind == "reference" else 1
    sets = [make_set() for _ in range(nsets)]
    turn = [0]

    def run_once(nthreads, rotate=True):
        work = sets[turn[0] % nsets] if rotate else sets[0]
        turn[0] += 1
        t0 = time.perf_counter()
        for Mw, K, pm, A, S, Bv in work:
            ntiles = Mw * BITS // BM
            Cout = np.zeros(Mw, np.float32)
            if kind == "reference" and bitnet:
                L = orc.ref_lib("intrins")
                ls = np.zeros(1, np.float32); lb = np.zeros(1, np.float32); q = np.zeros((K // 4, 16), np.int8)
                L.ref_preprocessor(K, K, orc._p(Bv), orc._p(ls), orc._p(lb), orc._p(q))
                rc = drv.ref_run_tiles_int32_omp(C.cast(L.ref_tile_cbits_int32, C.c_void_p), BITS, KF, BM, K, orc._p(A), C.c_size_t(A[0].nbytes),
                                                 orc._p(q), C.c_float(float(ls[0])), C.c_float(float(lb[0])), C.c_float(float(S[0])),
                                                 orc._p(Cout), ntiles, nthreads)
                assert rc == 0
            elif kind == "reference":
                L = orc.ref_lib(setname)
                G = K // AGS
                ls = np.zeros(G, np.float32); lb = np.zeros(G, np.float32); q = np.zeros((K // 4, 16), np.int8)
                pre = getattr(L, f"preprocessor_t1_int8_m{pm}_k{K}_n1_b{BITS}")
                pre(orc._p(Bv), orc._p(ls), orc._p(lb), orc._p(q))
                qg = getattr(L, f"qgemm_lut_t1_int8_m{BM}_k{K}_n1_b{BITS}")
                rc = drv.ref_run_tiles_omp(C.cast(qg, C.c_void_p), orc._p(A), C.c_size_t(A[0].nbytes), orc._p(q), orc._p(S),
                                           C.c_size_t(S[0].size), orc._p(ls), orc._p(lb), orc._p(Cout),
                                           C.c_size_t(BM // BITS), ntiles, nthreads)
                assert rc == 0
            elif bitnet:
                q, ls, lb = orc.preprocessor(Bv, K)
                orc.qgemm_scale_final(A, q, S, ls[:, 0], lb[:, 0], Mw, K, 1, BITS, BM, KF, 1)
            else:
                q, ls, lb = orc.preprocessor(Bv, AGS)
                orc.qgemm_float(A, q, S, ls, lb, Mw, K, 1, BITS, BM, KF, GS, AGS, True)
        return time.perf_counter() - t0

    out = {}
    # (the largest team stops at 128: on a 256-thread host whose container gets a share of them the full team measured 0.1 GB/s in rounds 4 / 5)
    thread_counts = sorted({1, min(cores, 8), min(cores, 32), min(cores, 64), min(cores, 128)}) if kind == "reference" else [1]
    for nthreads in thread_counts:
        run_once(nthreads)
        best, t_end, reps = 1e9, time.perf_counter() + seconds / len(thread_counts), 0
        while time.perf_counter() < t_end or reps < 5:
            best = min(best, run_once(nthreads)); reps += 1
        out[nthreads] = total_bytes / best / 1e9
    used = max(out, key=out.get)
    resident = None
    if nsets > 1:
        run_once(used, rotate=False)
        resident = round(total_bytes / min(run_once(used, rotate=False) for _ in range(5)) / 1e9, 3)
    return {"value": round(out[used], 3), "unit": "GB/s", "cores": used, "kind": kind, "host_cores": cores, "host_cores_total": os.cpu_count(), "cgroup_cpu_quota": _cpu_quota(),
            "by_threads_GBps": {str(k): round(v, 3) for k, v in out.items()}, "code": what,
            "weights": "%d distinct sets of the three shapes (%.0f MB) walked round-robin: every run streams its weights from DRAM" % (nsets, nsets * total_bytes / 1e6) if nsets > 1 else "one set",
            "cache_resident_GBps": resident,
            "sample": "one GEMV of each of the layer's three shapes (%s; preprocessor + all tiles, bm = %d), OpenMP static tile split, "
                      "best of >=5 per thread count over rotating weight sets; value = best thread count" % (", ".join(f"{m}x{k}" for m, k, _ in shapes), BM)}


def cpu_baseline_prefill(workload, seconds=8.0, rows=8):
    """N > 1 on the CPU = the reference's N = 1 kernel looped over the activation rows (python/t_mac/ops/qgemm.py:183-190,228-231).
    Timed on a bounded sample -- `rows` activation rows through one matrix of each of the layer's three shapes (preprocessor per row +
    all tiles, rows inside the tile loop so a tile's weights stay in cache: oracle/ref_driver.c) -- and scaled to the workload's unit:
    tokens/s = 1 / (layers x per-row time of a layer's seven matrices).  llama W2 / W4 only (the checked-in prebuilt kernel sets)."""
    import ctypes as C
    from oracle import oracle as orc
    wl = WORKLOADS[workload]
    base = workload.replace("-prefill", "")
    if base not in CPU_SETS:
        return {"note": "no compiled reference kernel set for this workload's N > 1 loop; its per-row rate is the decode workload's cpu_baseline"}
    setname, BITS, BM, shapes = CPU_SETS[base]
    if not orc.have_ref(setname):
        return {"note": "oracle/_ref is not built here"}
    cores = _usable_cores()
    rng = np.random.default_rng(0)
    GS, AGS = 128, 64
    drv = C.CDLL(os.path.join(ROOT, "oracle", "libref_driver.so"))
    L = orc.ref_lib(setname)
    counts = {(m, k): c for (nm, m, k, c, sl) in wl["mats"]}
    work = []
    for Mw, K, pm in shapes:
        M = Mw * BITS
        A = rng.integers(0, 256, size=(M // BM, K // 4, BM // 2), dtype=np.uint8)
        S = np.abs(rng.standard_normal((M // BM, K // GS, BM // BITS * 2))).astype(np.float32)
        Bv = rng.standard_normal((rows, K)).astype(np.float32)
        work.append((Mw, K, pm, A, S, Bv, counts[(Mw, K)]))

    def run_once(nthreads):
        per_layer = 0.0
        for Mw, K, pm, A, S, Bv, cnt in work:
            G = K // AGS
            ls = np.zeros((rows, G), np.float32); lb = np.zeros((rows, G), np.float32); q = np.zeros((rows, K // 4, 16), np.int8)
            Cout = np.zeros((rows, Mw), np.float32)
            pre = getattr(L, f"preprocessor_t1_int8_m{pm}_k{K}_n1_b{BITS}")
            qg = getattr(L, f"qgemm_lut_t1_int8_m{BM}_k{K}_n1_b{BITS}")
            t0 = time.perf_counter()
            for n in range(rows):
                pre(orc._p(Bv[n]), orc._p(ls[n]), orc._p(lb[n]), orc._p(q[n]))
            t1 = time.perf_counter()
            rc = drv.ref_run_tiles_rows_omp(C.cast(qg, C.c_void_p), orc._p(A), C.c_size_t(A[0].nbytes), orc._p(q), C.c_size_t(q[0].nbytes),
                                            orc._p(S), C.c_size_t(S[0].size), orc._p(ls), orc._p(lb), C.c_size_t(G), orc._p(Cout),
                                            C.c_size_t(BM // BITS), C.c_size_t(Mw), Mw * BITS // BM, rows, nthreads)
            t2 = time.perf_counter()
            assert rc == 0
            # one LUT per activation vector (q/k/v and gate/up share theirs), one tile pass per matrix
            per_layer += ((t1 - t0) + cnt * (t2 - t1)) / rows
        return per_layer

    out = {}
    thread_counts = sorted({1, min(cores, 8), min(cores, 32), min(cores, 64), min(cores, 128)})
    for nthreads in thread_counts:
        run_once(nthreads)
        best, t_end, reps = 1e9, time.perf_counter() + seconds / len(thread_counts), 0
        while time.perf_counter() < t_end or reps < 3:
            best = min(best, run_once(nthreads)); reps += 1
        out[nthreads] = 1.0 / (wl["layers"] * best)
    used = max(out, key=out.get)
    return {"value": round(out[used], 2), "unit": "tokens/s", "cores": used, "kind": "reference", "host_cores": cores,
            "by_threads_tokens_per_s": {str(k): round(v, 2) for k, v in out.items()},
            "code": f"prebuilt kernels deploy/tuned/{setname}/kernels.cc, N = 1 kernel looped over the rows (qgemm.py:183-190)",
            "sample": "%d activation rows through one matrix of each of the layer's three shapes (%s): preprocessor per row + all tiles (rows inside "
                      "the tile loop, OpenMP static tile split), best of >=3, scaled to %d layers x seven matrices; value = best thread count"
                      % (rows, ", ".join(f"{m}x{k}" for m, k, _ in shapes), wl["layers"])}


def measure_stream_calls(c, roof, name, Mw, K):
    """roofline.stream_core / stream_by_shape: SURVEY 8(d)'s headline measurement -- back-to-back INDEPENDENT GEMVs of one shape over rotating
    distinct weights (> MALL in total), a distinct activation vector per call: the recording runs in stream mode (tables once per call by
    k_lut_images, lookups by k_gemv_stream; both launches timed).  c: the run's context (run()); name / Mw / K: the target shape (MATS[3])."""
    # Calls per launch: args.stream_calls (default 96) DISTINCT weight sets -- the layers' own matrices plus extra synthetic ones of the
    # same shape, so that neither the MALL nor L2 ever holds a matrix when its call comes round again (96 x 4.2 MB for the smallest
    # shape) -- because a launch has a fixed cost (k_lut_images ~5.5 us, two launch boundaries, the persistent kernel's ramp and tail:
    # ~19 us measured, profiles/r06_stream_schedule.txt) that 32 calls of ~2.4 us do not amortise; the 32-call figure is kept beside it.
    SB = 10                      # replays back to back per event pair: launches overlap their predecessors' tails as in any timed loop

    def time_stream_calls(mi, ncalls, verify):
        name_, Mw_, K_, cnt_, slot_ = c.MATS[mi]
        extra = [[c.new_weights(c.shard_rows[name_], K_, c.cfg_of(name_)) for _ in range(cnt_)] for _ in range(max(ncalls - c.args.layers, 0))]
        sets = [c.layers[li][name_] for li in range(min(c.args.layers, ncalls))] + extra
        xs_ = [c.torch.randn(K_, device=c.dev, generator=c.gen).half() for _ in range(ncalls)]
        os_ = [[c.torch.empty(c.shard_rows[name_], dtype=c.torch.float16, device=c.dev) for _ in range(cnt_)] for _ in range(ncalls)]
        with c.wr.record_chain() as rec_:
            for i_ in range(ncalls):
                c.wr.fused(sets[i_], xs_[i_], os_[i_], 1, act_dtype=c.F16, out_dtype=c.F16)
        dur_ = []
        for r in range(11):
            e0 = c.torch.cuda.Event(enable_timing=True); e1 = c.torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(SB):
                rec_.chain.launch()
            e1.record()
            c.torch.cuda.synchronize()
            if r >= 3:
                dur_.append(e0.elapsed_time(e1) * 1e-3 / (SB * ncalls))
        ok_ = rec_.chain.status() == 0
        same_ = None
        if verify:
            # the launch's outputs against the same calls launched one by one (whose integer path the parity tests tap): same bits
            # (the quarter-walk form of k_gemv_stream adds a row's fp32 partial sums in another order: per-group-scale outputs within
            # 2e-3 of the stand-alone launch's fp16 values -- its integers are tapped against the oracle in tests/test_gpu_stream.py)
            same_ = True
            qw_ = bool(getattr(rec_.chain, "quarter_walk", False))
            for i_ in (0, ncalls // 2, ncalls - 1):
                if not qw_:
                    c.L.tmac_hip_debug_quad_config(rec_.chain.threads, rec_.chain.wpq(i_))
                ref_ = [c.torch.empty_like(o) for o in os_[i_]]
                c.wr.fused(sets[i_], xs_[i_], ref_, 1, act_dtype=c.F16, out_dtype=c.F16)
                c.torch.cuda.synchronize()
                if qw_ and c.MG < 1:
                    same_ = same_ and all(float((a_.float() - b_.float()).abs().max()) <= 2e-3 * float(a_.float().abs().max()) for a_, b_ in zip(ref_, os_[i_]))
                else:
                    same_ = same_ and all(bool(c.torch.equal(a_, b_)) for a_, b_ in zip(ref_, os_[i_]))
            c.L.tmac_hip_debug_quad_config(0, 0)
        mode_ = ("k_lut_images + k_gemv_stream (stream mode%s)" % (", qw" if getattr(rec_.chain, "quarter_walk", False) else "")) if getattr(rec_.chain, "stream", False) else "k_decode_chain"
        rec_.chain.free()
        for ws_ in extra:
            for w_ in ws_:
                w_.free()
        hb_ = cnt_ * algorithmic_bytes(Mw_, K_, c.BITS, c.GS, c.ags_of(K_), c.ZP, c.MG) - (cnt_ - 1) * (K_ // 4 * 16 + (K_ // c.ags_of(K_)) * 4)
        return float(np.mean(dur_)) * 1e6, float(np.min(dur_)) * 1e6, hb_, ok_, same_, mode_

    ncalls = max(c.args.stream_calls, 2)
    sus, smin, hb_s, sok, ssame, smode = time_stream_calls(3, ncalls, True)
    roof["stream_core"] = {"what": "%d independent GEMVs %s (%dx%d, %d distinct weight sets, a distinct activation vector each) recorded once, launched as %s: "
                                   "tables built once per call, lookups with the tables prebuilt, no hand-offs" % (ncalls, name, Mw, K, ncalls, smode),
                           "calls_per_launch": ncalls,
                           "us_per_gemv": round(sus, 3), "min_us": round(smin, 3), "GBps": round(hb_s / sus * 1e-3, 1),
                           "frac": round(hb_s / sus * 1e-3 / HBM_PEAK_GBS, 4), "ok": bool(sok and ssame),
                           "matches_single_launches": ssame, "form": "quarter-walk (outputs within 2e-3 of the stand-alone launches; integers tapped in tests)" if "qw" in smode else "quad x 64 units (bit-identical to the stand-alone launches)",
                           "timing": "hipEvent pair around 10 back-to-back replays (LUT build launches included), mean of 8 pairs"}
    if ncalls != c.args.layers:
        s32, m32, _, ok32, _, _ = time_stream_calls(3, c.args.layers, False)
        roof["stream_core"]["at_%d_calls_per_launch" % c.args.layers] = {"us_per_gemv": round(s32, 3), "min_us": round(m32, 3),
                                                                       "frac": round(hb_s / s32 * 1e-3 / HBM_PEAK_GBS, 4), "ok": ok32}
    # the same measurement for the layer's other three calls (q/k/v fused, o, gate/up fused): what the shape costs in stream mode
    try:
        by_shape = {}
        for mi2 in range(3):
            name2, Mw2, K2, cnt2, slot2 = c.MATS[mi2]
            us2, min2, hb2, ok2, _, _ = time_stream_calls(mi2, ncalls, False)
            by_shape[name2] = {"shape": "%d x %dx%d" % (cnt2, Mw2, K2), "calls_per_launch": ncalls, "us_per_call": round(us2, 3), "GBps": round(hb2 / us2 * 1e-3, 1),
                               "frac": round(hb2 / us2 * 1e-3 / HBM_PEAK_GBS, 4), "ok": ok2}
        roof["stream_by_shape"] = by_shape
    except c.tmac_amd.binding.TMACHipError as e:
        roof["stream_by_shape"] = {"error": repr(e)}


def measure_independent_pattern(c, roof):
    """roofline.independent_pattern of a CHAINED run: the token's mpGEMMs as independent calls (what --pattern independent, the default, times);
    with several ranks every rank streams its row shard, nothing is exchanged, the slowest rank's time counts"""
    try:
        ix = {s_: c.torch.randn(c.xdim[s_], device=c.dev, generator=c.gen).half() for s_ in c.xdim}
        iouts = [{n_: [c.torch.empty_like(o) for o in c.outs[n_]] for n_ in c.outs} for _ in range(c.args.layers)]
        with c.wr.record_chain() as irec:
            for li in range(c.args.layers):
                c.calls(c.layers[li], ix, iouts[li], exchange=False, link=False)
        for _ in range(2):
            irec.chain.launch()
        c.barrier()
        idur = []
        for r in range(8):
            e0 = c.torch.cuda.Event(enable_timing=True); e1 = c.torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                irec.chain.launch()
            e1.record()
            c.torch.cuda.synchronize()
            if r >= 3:
                idur.append(e0.elapsed_time(e1) / 10)
        iok = irec.chain.status() == 0
        imode = ("stream mode%s" % (", quarter-walk form" if getattr(irec.chain, "quarter_walk", False) else "")) if getattr(irec.chain, "stream", False) else "k_decode_chain"
        irec.chain.free()
        ims = float(np.mean(idur))
        if c.dist_on:
            t_ = c.torch.tensor([ims, 0.0 if iok else 1.0], dtype=c.torch.float64, device=c.dev)
            c.d_all_reduce(t_, c.dist.ReduceOp.MAX)
            ims, iok = float(t_[0].item()), float(t_[1].item()) == 0.0
        roof["independent_pattern"] = {"what": "the token's %d mpGEMMs as independent calls (each reads a vector that is in memory before the launch), %s%s"
                                               % (7 * c.args.layers, imode, "" if c.world == 1 else "; %d ranks, each over its row shard, no exchange; slowest rank's time" % c.world),
                                       "n_gpus": c.world, "ms_per_token": round(ims, 4),
                                       "GBps": round(c.bytes_per_step / (ims * 1e-3) / 1e9, 1),
                                       "frac": round(c.bytes_per_step / (ims * 1e-3) / 1e9 / (HBM_PEAK_GBS * c.world), 4), "ok": iok}
        del iouts
    except Exception as e:
        roof["independent_pattern"] = {"error": repr(e)}


def verify_against_oracle(c):
    """outside the timed region: the launches being timed, at full size (layer 0), against the oracle; returns the line's `verified` object"""
    verified = None
    if c.host_l0:
        from oracle import oracle as orc
        vshape = (lambda k: (k,)) if c.decode else (lambda k: (c.N, k))
        vx0 = c.torch.randn(vshape(c.MATS[0][2]), device=c.dev, generator=c.gen).half()
        vlink = c.args.pattern != "independent"         # independent pattern: every call of the layer reads a resident vector of its own
        vx = {c.MATS[0][4]: vx0} if vlink else {s_: c.torch.randn(vshape(c.xdim[s_]), device=c.dev, generator=c.gen).half() for s_ in c.xdim}
        if not vlink:
            vx0 = vx[c.MATS[0][4]]
        vouts = {name: [c.torch.zeros(vshape(Mw), dtype=c.torch.float16, device=c.dev) for _ in range(cnt)] for name, Mw, K, cnt, slot in c.MATS}
        ok = True
        if c.args.path == "chain":
            with c.wr.record_chain() as vrec:
                c.calls(c.layers[0], vx, vouts, exchange=False, link=vlink)
            vrec.chain.launch()
            c.torch.cuda.synchronize()
            ok = vrec.chain.status() == 0
            vrec.chain.free()
        else:
            c.calls(c.layers[0], vx, vouts, exchange=False, link=vlink)
            c.torch.cuda.synchronize()
        worst = 0.0
        rows = [0] if c.decode else [0, c.N - 1]          # prefill: two of the N activation rows (the oracle takes seconds per row)
        for name, Mw, K, cnt, slot in c.MATS:
            src = vx[slot] if not vlink else (vx0 if slot == c.MATS[0][4] else vouts[[n_ for n_ in c.nxt if c.nxt[n_] == slot][0]][0])     # what this call consumed
            xin_h = src.float().cpu().numpy().reshape(-1, K)[rows]
            q, ls, lb = orc.preprocessor(xin_h, c.ags_of(K))
            for i in range(cnt):
                A, S = c.host_l0[name][i]
                if c.MG >= 1:
                    ref, _ = orc.qgemm_scale_final(A, q, S, ls[:, 0], lb[:, 0], Mw, K, len(rows), c.BITS, c.BM, KF, c.MG)
                else:
                    ref = orc.qgemm_float(A, q, S, ls, lb, Mw, K, len(rows), c.BITS, c.BM, KF, c.GS, c.ags_of(K), c.ZP)
                got = vouts[name][i].float().cpu().numpy().reshape(-1, Mw)[rows]
                worst = max(worst, float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)))
        verified = {"ok": bool(ok and worst <= 1e-3), "max_rel_err": float("%.3g" % worst), "tolerance": 1e-3,
                    "what": "layer 0's seven mpGEMMs (q/k/v, o, gate/up, down at full size, %s) through the timed path vs oracle/ "
                            "(fp16 outputs%s)" % ("chained" if vlink else "independent calls", "" if c.decode else "; activation rows 0 and N-1")}
        if not verified["ok"]:
            sys.stderr.write("bench.py: VERIFICATION FAILED: %r\n" % (verified,))

    return verified


def measure_dense_fp16(c, roof, ev_ms_per_step):
    """prefill lines: roofline.dense_fp16_baseline -- a dense fp16 torch.matmul (hipBLASLt) of the same shapes on this box, replayed from a hipGraph
    like the timed path"""
    try:
        dW = {name: [c.torch.randn(c.shard_rows[name], K, device=c.dev, generator=c.gen).half() for _ in range(cnt)] for name, Mw, K, cnt, slot in c.MATS}
        dX = {slot: c.torch.randn(c.N, K, device=c.dev, generator=c.gen).half() for name, Mw, K, cnt, slot in c.MATS}
        dO = {name: [c.torch.empty(c.N, c.shard_rows[name], dtype=c.torch.float16, device=c.dev) for _ in range(cnt)] for name, Mw, K, cnt, slot in c.MATS}

        def dense_step():
            for _ in range(c.args.layers):
                for name, Mw, K, cnt, slot in c.MATS:
                    for i in range(cnt):
                        c.torch.matmul(dX[slot], dW[name][i].t(), out=dO[name][i])
        dense_step(); c.torch.cuda.synchronize()
        dside = c.torch.cuda.Stream(); dside.wait_stream(c.torch.cuda.current_stream())
        with c.torch.cuda.stream(dside):
            dense_step()
        c.torch.cuda.current_stream().wait_stream(dside); c.torch.cuda.synchronize()
        dg = c.torch.cuda.CUDAGraph()
        with c.torch.cuda.graph(dg, stream=dside):
            dense_step()
        dts = []
        for r in range(8):
            f0 = c.torch.cuda.Event(enable_timing=True); f1 = c.torch.cuda.Event(enable_timing=True)
            f0.record(); dg.replay(); f1.record(); c.torch.cuda.synchronize()
            if r >= 3:
                dts.append(f0.elapsed_time(f1))
        dms = float(np.mean(dts))
        roof["dense_fp16_baseline"] = {"ms_per_step": round(dms, 4), "tokens_per_s": round(c.N / (dms * 1e-3), 1),
                                       "this_over_dense": round(dms / ev_ms_per_step, 3),
                                       "what": "torch.matmul (hipBLASLt) fp16 [N, K] x [K, rows] of the same %d matrices per layer (ONE weight set reused by every layer: "
                                               "%.0f MB of fp16 weights, MALL-resident -- the 2-bit path streams distinct weights per layer), hipGraph replay, mean of 5; "
                                               "it reads 8 x the weight bytes per matrix and cannot produce the reference's integer sums"
                                               % (sum(m[3] for m in c.MATS), sum(m[3] * c.shard_rows[m[0]] * m[2] for m in c.MATS) * 2 / 1e6)}
        del dW, dX, dO, dg
    except Exception as e:
        roof["dense_fp16_baseline"] = {"error": repr(e)}


# ---- the decoder pattern: the same matrices as a decoder issues them (cx: the run's context) ----
# ---- the decoder pattern: the same 224 matrices as a decoder issues them.  A real layer has an operator outside
# the hot path between q/k/v and o (attention) and element-wise operators between the other mpGEMMs; the latter run inside the
# chain's LUT builds (tmac_hip_chain_xform), the former ends the persistent launch: one launch per segment.
def build_decoder_pattern(cx):
    H = cx.MATS[1][2]                      # hidden size = K of the o projection
    f16 = lambda n_: cx.torch.zeros(n_, dtype=cx.torch.float16, device=cx.dev)
    gam = [(cx.torch.ones(H, dtype=cx.torch.float32, device=cx.dev), cx.torch.ones(H, dtype=cx.torch.float32, device=cx.dev)) for _ in range(cx.args.layers)]
    hs = [cx.torch.randn(H, device=cx.dev, generator=cx.gen)] + [cx.torch.zeros(H, dtype=cx.torch.float32, device=cx.dev) for _ in range(cx.args.layers)]
    attn = f16(H)
    bo = [dict(o=[f16(cx.shard_rows["o"])], gate_up=[f16(cx.shard_rows["gate_up"]) for _ in range(2)], down=[f16(cx.shard_rows["down"])],
               qkv=[f16(cx.shard_rows["qkv"]) for _ in range(3)]) for _ in range(cx.args.layers + 1)]
    x0 = hs[0].half()
    chains_d = []
    xf_sel = os.environ.get("TMAC_BENCH_DECODER_XF", "all")      # measurement only: all | none | norm | glu (which transforms the segments carry)
    do_norm, do_glu = xf_sel in ("all", "norm"), xf_sel in ("all", "glu")
    with cx.wr.record_chain() as r0:
        if do_norm:
            cx.wr.chain_xform("norm", gamma=gam[0][0], eps=1e-5)
        cx.wr.fused(cx.layers[0]["qkv"], x0, bo[0]["qkv"], 1, act_dtype=cx.F16, out_dtype=cx.F16)
    chains_d.append(r0.chain)
    for li in range(cx.args.layers):
        b, last = bo[li + 1], li == cx.args.layers - 1
        with cx.wr.record_chain() as rc:
            cx.wr.fused(cx.layers[li]["o"], attn, b["o"], 1, act_dtype=cx.F16, out_dtype=cx.F16)
            if do_norm:
                cx.wr.chain_xform("norm", residual=hs[li], gamma=gam[li][1], eps=1e-5, keep=True)
            cx.wr.fused(cx.layers[li]["gate_up"], b["o"][0], b["gate_up"], 1, act_dtype=cx.F16, out_dtype=cx.F16)
            if do_glu:
                cx.wr.chain_xform("glu", in2=b["gate_up"][1])
            cx.wr.fused(cx.layers[li]["down"], b["gate_up"][0], b["down"], 1, act_dtype=cx.F16, out_dtype=cx.F16)
            if not last:
                if do_norm:
                    cx.wr.chain_xform("norm", residual=cx.wr.CARRY, gamma=gam[li + 1][0], eps=1e-5, residual_out=hs[li + 1])
                cx.wr.fused(cx.layers[li + 1]["qkv"], b["down"][0], b["qkv"], 1, act_dtype=cx.F16, out_dtype=cx.F16)
        chains_d.append(rc.chain)
    keep_alive = (gam, hs, attn, bo, x0)

    def dstep(segments=True):
        if segments:
            chains_d[0].launch()
        for li in range(cx.args.layers):
            attn.copy_(bo[li]["qkv"][0])          # the outside operator: a kernel of the stream between two segments (stand-in for attention)
            if segments:
                chains_d[li + 1].launch()
    return dstep, chains_d, keep_alive

def time_outside_ops(cx, dstep, reps=10):
    """the stand-ins for attention alone (graph replay of the same 32 copies): the part of the decoder pattern's time that is not this library's"""
    try:
        side = cx.torch.cuda.Stream()
        side.wait_stream(cx.torch.cuda.current_stream())
        with cx.torch.cuda.stream(side):
            dstep(False)
        cx.torch.cuda.current_stream().wait_stream(side)
        cx.torch.cuda.synchronize()
        g = cx.torch.cuda.CUDAGraph()
        with cx.torch.cuda.graph(g, stream=side):
            dstep(False)
    except Exception:
        cx.torch.cuda.synchronize()
        return None
    durs = []
    for r in range(reps + 3):
        e0 = cx.torch.cuda.Event(enable_timing=True); e1 = cx.torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record()
        cx.torch.cuda.synchronize()
        if r >= 3:
            durs.append(e0.elapsed_time(e1))
    return float(np.mean(durs))

def time_decoder_pattern(cx, reps=10):
    dstep, chains_d, keep_alive = build_decoder_pattern(cx)
    dstep(); cx.torch.cuda.synchronize()
    ok = all(c.status() == 0 for c in chains_d)
    g = None
    try:
        side = cx.torch.cuda.Stream()
        side.wait_stream(cx.torch.cuda.current_stream())
        with cx.torch.cuda.stream(side):
            dstep()
        cx.torch.cuda.current_stream().wait_stream(side)
        cx.torch.cuda.synchronize()
        g = cx.torch.cuda.CUDAGraph()
        with cx.torch.cuda.graph(g, stream=side):
            dstep()
    except Exception as e:
        sys.stderr.write(f"bench.py: decoder pattern not captured ({e!r}); eager launches\n")
        g = None
        cx.torch.cuda.synchronize()
    durs = []
    for r in range(reps + 3):
        e0 = cx.torch.cuda.Event(enable_timing=True); e1 = cx.torch.cuda.Event(enable_timing=True)
        e0.record(); (g.replay() if g is not None else dstep()); e1.record()
        cx.torch.cuda.synchronize()
        if r >= 3:
            durs.append(e0.elapsed_time(e1))
    ok = ok and all(c.status() == 0 for c in chains_d)
    finite_d = bool(cx.torch.isfinite(keep_alive[3][-1]["down"][0].float()).all().item())
    outside = time_outside_ops(cx, dstep) if g is not None else None
    for c in chains_d:
        c.free()
    return float(np.mean(durs)), ok and finite_d, len(chains_d), g is not None, outside


def run(args, env):
    """one workload, measured as the contract says; returns the result dict on rank 0 (None elsewhere).  env: what main() set up once
    per process (torch.distributed, the kept stdout)"""
    import torch
    import torch.distributed as dist
    wl = WORKLOADS[args.workload]
    MATS, BITS, BM, GS, ZP, MG, N = wl["mats"], wl["bits"], wl["bm"], wl["gs"], wl["zp"], wl["mg"], wl["N"]
    decode = N == 1
    world, rank, local_rank, dist_on = env["world"], env["rank"], env["local_rank"], env["dist_on"]
    chain_ok = decode and args.variant == 0
    if args.pattern == "auto":
        # a decode line times SURVEY 8(d)'s measurement: the token's GEMVs back to back over distinct weights, every call reading a resident
        # vector (main() then measures the dependent chain of the same matrices and reports it beside the value)
        args.pattern = "independent" if (decode and not args.stamps) else "chained"       # (--stamps: k_decode_chain's per-call stamps)
    if args.path == "auto":
        args.path = "chain" if chain_ok else "fused"
    if args.path == "chain" and not chain_ok:
        raise SystemExit("bench.py: --path chain covers N = 1")

    # the few collectives of the bootstrap and the timing: on the device through RCCL, or -- test mode -- through host tensors on gloo
    def d_all_reduce(t, op):
        if args.share_device:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op)

    def d_all_gather_into(out, inp):
        if args.share_device:
            parts = [torch.empty(inp.shape, dtype=inp.dtype) for _ in range(world)]
            dist.all_gather(parts, inp.cpu())
            out.copy_(torch.cat([p.reshape(-1) for p in parts]).reshape(out.shape))
        else:
            dist.all_gather_into_tensor(out, inp)
    import tmac_amd
    from tmac_amd import KCfg, F16, F32
    L = tmac_amd.lib()
    tmac_amd.binding.check(L.tmac_hip_init(local_rank))
    if args.share_device and world > 1:
        if args.comm == "lib":
            raise SystemExit("bench.py: --share-device cannot use RCCL (--comm lib)")
        # both persistent kernels must be resident at once: the CUs are divided between the ranks
        tmac_amd.binding.check(L.tmac_hip_debug_chain_grid(max(8, (256 - 32 * world) // world)))
    tmac_amd.binding.check(L.tmac_hip_set_variant(args.variant))
    tmac_amd.binding.check(L.tmac_hip_debug_gemm_kernel(args.gemm_kernel))
    dev = torch.device("cuda", local_rank)
    gen = torch.Generator(device=dev); gen.manual_seed(1234)   # same weights on every rank, sliced by rank below
    ags_of = (lambda K: K) if MG >= 1 else (lambda K: wl["ags"])
    maxK = max(m[2] for m in MATS)
    wr = tmac_amd.TMACGeMMWrapper(act_group_size=ags_of(maxK))
    wr.set_workspace(maxK, N)

    # ---- synthetic, row-sharded weights, registered (re-tiled on the GPU) once --------------------
    rpt = BM // BITS                                  # output rows per reference tile
    layers, host_l0, shard_rows = [], {}, {}
    bytes_per_step = 0
    ops_per_step = 0.0
    for name, Mw, K, cnt, slot in MATS:
        ntiles = Mw // rpt
        tiles_per_rank = (ntiles + world - 1) // world          # ragged split -> padded with extra synthetic rows
        shard_rows[name] = tiles_per_rank * rpt
        bytes_per_step += cnt * algorithmic_bytes(Mw, K, BITS, GS, ags_of(K), ZP, MG, N)
        ops_per_step += cnt * 2.0 * Mw * (K / 4 * 8) * N        # int8 MFMA work k_gemm_planes issues: one operand row per OUTPUT row (planes combined), 8-entry half tables
    bytes_per_step *= args.layers
    ops_per_step *= args.layers
    lvl = (2 ** BITS - 1) / 2.0 - 2 ** (BITS - 1)                # mean weight level minus the offset 2^(b-1)
    wvar = (4 ** BITS - 1) / 12.0                                 # variance of a uniform b-bit level
    keep_host = not args.no_verify and world == 1
    def new_weights(Mloc, K, cfg, keep=None):
        """one synthetic matrix in the reference layout, registered (re-tiled on the GPU); keep: list that receives the host copy"""
        A = torch.randint(0, 256, (Mloc * BITS // BM, K // 4, BM // 2), dtype=torch.uint8, device=dev, generator=gen)
        if MG >= 1:
            # BitNet's weights are ternary, {-1, 0, 1} stored as levels {1, 2, 3} of the 2-bit format (level 0 unused).  In the
            # reference layout (weights.py:57-73) byte lanes 0-7 of every 16 hold plane-0 nibbles and lanes 8-15 the plane-1
            # nibbles of the same outputs, so the constraint is bytewise: plane1 set with probability 43/64, plane0 = random |
            # ~plane1 gives P(1) = 21/64, P(2) = P(3) = 43/128: mean level - 2 = 1/128, variance 0.664 -- zero-mean enough
            # that a common component of the activations is not amplified (gain 0.9 at K = 8640), and with the scale
            # 0.98 / sqrt(0.664 K) the chained vectors keep unit size through all layers.
            lanes = A.view(-1, 16)
            r = [torch.randint(0, 256, (lanes.shape[0], 8), dtype=torch.uint8, device=dev, generator=gen) for _ in range(5)]
            p1 = lanes[:, 8:] | (r[0] & (r[1] | (r[2] & (r[3] | r[4]))))
            lanes[:, 8:] = p1
            lanes[:, :8] |= ~p1
            del r, p1
            S = torch.full((MG,), 0.98 / float(np.sqrt(0.664 * K)), device=dev, dtype=torch.float32)
            w = tmac_amd.Weights(A, S, Mloc, K, BITS, cfg, scales_dtype=F32, dev_dtype=F32, on_device=True)
            if keep is not None:
                keep.append((A.cpu().numpy(), S.cpu().numpy()))
        else:
            # real weight = (w - 2^(b-1)) scale - zero.  zero = (mean level - 2^(b-1)) scale + noise makes it zero-mean,
            # so a common component of the activations is not amplified from call to call (with independent zeros
            # it grew 16x per GEMV and the chained vectors overflowed fp16 after a few layers); c: unit gain
            c = 1.0 / np.sqrt((wvar + 1.0) * K)
            S = torch.randn((Mloc * BITS // BM, K // GS, rpt // 8, 2 if ZP else 1, 8), device=dev, generator=gen) * c
            S[:, :, :, 0, :].abs_()
            if ZP:
                S[:, :, :, 1, :] += S[:, :, :, 0, :] * lvl
            S = S.half().contiguous()
            w = tmac_amd.Weights(A, S, Mloc, K, BITS, cfg, scales_dtype=F16, dev_dtype=F16, on_device=True)
            if keep is not None:
                keep.append((A.cpu().numpy(), S.float().cpu().numpy().reshape(Mloc * BITS // BM, K // GS, -1)))
        return w

    def cfg_of(name):
        Mw, K = next((m[1], m[2]) for m in MATS if m[0] == name)
        return KCfg.make(shard_rows[name], K, BITS, BM, KF, GS if GS else 128, ags_of(K), ZP, MG, N)

    for li in range(args.layers):
        mats = {}
        for name, Mw, K, cnt, slot in MATS:
            cfg = cfg_of(name)
            mats[name] = [new_weights(shard_rows[name], K, cfg, host_l0.setdefault(name, []) if (li == 0 and keep_host) else None) for _ in range(cnt)]
        layers.append(mats)
    torch.cuda.synchronize()

    # activations: x[slot] full blocks [N][K] (fp16); per-matrix local outputs [N][rows of this rank]
    xdim = {slot: K for name, Mw, K, cnt, slot in MATS}
    shp = (lambda k: (k,)) if decode else (lambda k: (N, k))
    x = {s: torch.randn(shp(xdim[s]), device=dev, generator=gen).half() for s in xdim}
    outs = {name: [torch.empty(shp(shard_rows[name]), dtype=torch.float16, device=dev) for _ in range(cnt)]
            for name, Mw, K, cnt, slot in MATS}
    gathered = {name: torch.empty((world,) + shp(shard_rows[name]), dtype=torch.float16, device=dev) for name, *_ in MATS}
    nxt = {"qkv": 1, "o": 2, "gate_up": 3, "down": 0}
    logical = {name: Mw for name, Mw, K, cnt, slot in MATS}

    tuned = {}
    if args.path == "fused" and args.autotune and args.variant == 0 and decode:
        for name, Mw, K, cnt, slot in MATS:
            r = wr.autotune(layers[0][name], F16, F16)
            tuned[name] = [r["ft"], r["wpq"], round(r["us"], 2), round(r["heuristic_us"], 2)]
        torch.cuda.synchronize()

    fused_calls = args.path in ("chain", "fused")
    lib_comm = None
    if dist_on and args.comm == "lib":
        # rank 0 creates the RCCL id, torch.distributed carries the 128 bytes to the other ranks (bootstrap only)
        idt = torch.zeros(tmac_amd.Comm.ID_BYTES, dtype=torch.uint8, device=dev)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(tmac_amd.Comm.unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, 0)
        lib_comm = tmac_amd.Comm(bytes(idt.cpu().numpy().tobytes()), rank, world)
    elif dist_on and args.comm == "ipc":
        # every rank's window is mapped by all peers; torch.distributed carries the 128-byte blobs (bootstrap only)
        maxb = max(N * shard_rows[name] * 2 for name, *_ in MATS)
        lib_comm = tmac_amd.Comm.ipc(maxb, rank, world)
        blob = torch.frombuffer(bytearray(lib_comm.export()), dtype=torch.uint8).to(dev)
        blobs = torch.empty(world * tmac_amd.Comm.BLOB_BYTES, dtype=torch.uint8, device=dev)
        d_all_gather_into(blobs, blob)
        lib_comm.connect(bytes(blobs.cpu().numpy().tobytes()))

    recording = [False]          # inside wr.record_chain(): exchange steps are recorded, not executed

    def calls(mats, xin, out_of, exchange=True, link=None):
        """the hot-path calls of one layer; xin / out_of: dicts of input blocks per slot / output lists per matrix group"""
        for name, Mw, K, cnt, slot in MATS:
            if fused_calls:
                wr.fused(mats[name], xin[slot], out_of[name], N, act_dtype=F16, out_dtype=F16)
            else:
                wr.llama_cpp_init(xin[slot], Mw, K, N, BITS, act_group_size=ags_of(K), act_dtype=F16)
                for i in range(cnt):
                    wr.llama_cpp_compute(mats[name][i], out_of[name][i], N, out_dtype=F16)
            # exchange step: the first output of the group becomes the next activation block.  (--pattern independent: every call reads a
            # vector that is resident before the launch; rows stay sharded and NOTHING is exchanged -- the mode that scales by construction)
            exchange = exchange and args.pattern != "independent"
            if dist_on and exchange and recording[0] and lib_comm is None:
                # row-sharded chain: the exchange step becomes part of the in-kernel hand-off (tmac_hip_chain_record_gather)
                wr.record_gather(out_of[name][0], gathered[name], out_of[name][0].numel() * 2, rank, world)
                xin[nxt[name]] = gathered[name].reshape(-1)[:logical[name]]
            elif dist_on and exchange:
                if lib_comm is not None:
                    lib_comm.allgather(out_of[name][0], gathered[name], out_of[name][0].numel() * 2)
                else:
                    d_all_gather_into(gathered[name], out_of[name][0])
                g = gathered[name]
                xin[nxt[name]] = (g.reshape(-1)[:logical[name]] if decode
                                  else g.permute(1, 0, 2).reshape(N, -1)[:, :logical[name]].contiguous())
            elif (args.pattern != "independent") if link is None else link:
                xin[nxt[name]] = out_of[name][0]

    # independent pattern: nothing orders the calls, so every layer gets output buffers of its own
    outs_l = [outs] + [{name: [torch.empty_like(o) for o in outs[name]] for name in outs} for _ in range(args.layers - 1)] \
        if args.pattern == "independent" else [outs] * args.layers

    step_in = [None]            # the tensor the most recent step() read as its first activation block (a replayed graph / chain keeps reading THAT one)

    def step():
        step_in[0] = x[MATS[0][4]]
        for li in range(args.layers):
            calls(layers[li], x, outs_l[li])

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    # what the measurement functions outside run() need of this run (measure_*, verify_against_oracle, the decoder pattern)
    import types
    ctx = types.SimpleNamespace(args=args, torch=torch, dist=dist, tmac_amd=tmac_amd, F16=F16, dev=dev, gen=gen, wr=wr, L=L, layers=layers, MATS=MATS,
                                BITS=BITS, BM=BM, GS=GS, ZP=ZP, MG=MG, N=N, decode=decode, world=world, dist_on=dist_on, ags_of=ags_of, cfg_of=cfg_of,
                                new_weights=new_weights, shard_rows=shard_rows, calls=calls, outs=outs, xdim=xdim, nxt=nxt, barrier=barrier,
                                d_all_reduce=d_all_reduce, bytes_per_step=bytes_per_step, host_l0=host_l0)
    # ---- launch mechanism ---------------------------------------------------------------------------------------------
    # (the fused entry point's per-stream LUT workspace for N > 1 must not be allocated inside a capture: the warm-up step and
    # the capture below run on the same side stream, so the workspace exists and has its final size when capture starts)
    graph, chain, stamp_buf = None, None, None
    dpat = None
    if args.pattern == "decoder":
        if args.path != "chain" or dist_on:
            raise SystemExit("bench.py: --pattern decoder runs on the chain path of one GPU")
    if args.path == "chain" and args.pattern != "decoder":
        # record the token's calls once (they are not launched while recording); one launch per step from here on
        step()                                       # leaves x[0] = the down projection's output, as in a decode loop
        torch.cuda.synchronize()
        def all_ranks_ok(ok):
            if not dist_on:
                return ok
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            d_all_reduce(t, dist.ReduceOp.MIN)
            return bool(t.item())
        recording[0] = True
        why = ""
        try:
            with wr.record_chain() as rec:
                step()
            chain = rec.chain
        except tmac_amd.binding.TMACHipError as e:
            if not dist_on:
                raise
            why = str(e)
        recording[0] = False
        if dist_on:
            # Every decision below is taken by ALL ranks together (a rank on its own path would leave the others in a collective).
            # The ranks exchange the IPC handles of their hand-off arenas (bootstrap only); every producer then stores its granules
            # into all of them (system-scope stores over xGMI).  Two trial launches prove the hand-off before anything is timed.
            ok = all_ranks_ok(chain is not None)
            if ok and args.pattern == "independent":
                pass                      # nothing is handed over between ranks: every rank's recording is a stream of its own (no arenas to connect)
            elif ok:
                try:
                    blob = torch.frombuffer(bytearray(chain.export()), dtype=torch.uint8).to(dev)
                except tmac_amd.binding.TMACHipError as e:
                    why, blob = str(e), torch.zeros(tmac_amd.DecodeChain.BLOB_BYTES, dtype=torch.uint8, device=dev)
                blobs = torch.empty(world * tmac_amd.DecodeChain.BLOB_BYTES, dtype=torch.uint8, device=dev)
                d_all_gather_into(blobs, blob)
                try:
                    if not why:
                        chain.connect(bytes(blobs.cpu().numpy().tobytes()))
                except tmac_amd.binding.TMACHipError as e:
                    why = str(e)
                ok = all_ranks_ok(not why)
            if ok:
                barrier()
                for _ in range(2):
                    chain.launch()
                torch.cuda.synchronize()
                st = chain.status()
                if st:
                    why = "a hand-off across ranks timed out in the trial launches (error word %#x)" % st
                ok = all_ranks_ok(st == 0)
            if not ok and args.share_device:
                raise SystemExit(f"bench.py --share-device: the row-sharded chain did not come up ({why or 'another rank failed'})")
            if not ok:
                if rank == 0:
                    sys.stderr.write(f"bench.py: no row-sharded chain on this node ({why or 'another rank failed'}); per-launch path with RCCL all-gathers instead\n")
                if chain is not None:
                    chain.free()
                chain, args.path = None, "fused"
        if chain is not None and args.stamps:
            stamp_buf = torch.zeros(chain.nops * chain.grid * 8, dtype=torch.int64, device=dev)
            chain.set_stamps(stamp_buf)
    use_graph = (not args.no_graph) and not (dist_on and args.eager_collectives) and args.path != "chain"
    if use_graph:
        # One step (the launches + the all-gathers) is captured into a hipGraph and replayed.  With RCCL collectives inside,
        # capture was exercised with one rank only on the development box: if capture raises, the run falls back to eager
        # launches; if the first replay does not finish, a watchdog ends the process instead of hanging the node.
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            if dist_on:
                # ProcessGroupNCCL's watchdog thread polls the events of earlier collectives; under the default (global)
                # capture mode such a query from another thread invalidates the capture and kills the process.  Let it reap
                # what has completed, then capture in thread-local mode, where only this thread's calls are policed.
                time.sleep(1.0)
                with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                    step()
            else:
                with torch.cuda.graph(graph, stream=side):
                    step()
        except Exception as e:   # capture not supported with this RCCL / torch build: measured eagerly instead
            if not dist_on:
                raise
            sys.stderr.write(f"bench.py: graph capture with collectives failed ({e!r}); launching eagerly\n")
            graph, use_graph = None, False
            try:
                torch.cuda.synchronize()
            except Exception:
                pass

    if args.pattern == "decoder":
        dstep, dchains, dkeep = build_decoder_pattern(ctx)
        dstep(); torch.cuda.synchronize()
        assert all(c.status() == 0 for c in dchains), "a hand-off inside a decoder segment timed out"
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            dstep()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            dstep()
        dpat = dict(chains=dchains, keep=dkeep, dstep=dstep)

    def run_step():
        if chain is not None:
            chain.launch()
        elif graph is not None:
            graph.replay()
        else:
            step()

    if dist_on:
        import threading
        done = threading.Event()

        def watchdog():
            if not done.wait(240.0):
                sys.stderr.write("bench.py: the first step with RCCL all-gathers did not complete; rerun with --eager-collectives\n")
                sys.stderr.flush()
                os._exit(3)
        threading.Thread(target=watchdog, daemon=True).start()
        try:
            run_step()
        except tmac_amd.binding.TMACHipError:
            torch.cuda.synchronize()
            run_step()
        torch.cuda.synchronize()
        done.set()

    # ---- multi-GPU preflight (before anything is timed): which path and transport every rank took, and one step of the path being timed
    # against a communication-free emulation of the same step on every rank.  All ranks hold the same synthetic shard (same seed), so the
    # all-gather of an output is that output repeated `world` times: each rank can replay the whole step alone -- the same calls launched
    # one by one with the chain's launch configuration, the gather replaced by a local repeat -- and must arrive at the same last-layer
    # outputs as the distributed step (same bits when the paths share the configuration; a wrong or stale exchange is off by O(1)).
    preflight = None
    if dist_on and dpat is None:
        path_txt = ("row-sharded stream mode: every rank runs k_lut_images + k_gemv_stream over its row shard of the independent calls; nothing is exchanged"
                    if (chain is not None and args.pattern == "independent") else
                    "row-sharded persistent chain: hand-off granules stored into every rank's IPC-mapped arena (system scope over xGMI)" if chain is not None else
                    "one launch per fused call + all-gather per exchange step over " +
                    {"torch": "torch.distributed (ProcessGroupNCCL = RCCL)", "lib": "tmac_hip_comm (RCCL through the C-ABI)", "ipc": "tmac_hip_comm IPC windows"}[args.comm] +
                    (", replayed from a hipGraph" if graph is not None else ", eager"))
        sys.stderr.write(f"bench.py preflight: rank {rank}/{world} device {local_rank}: {path_txt}\n")
        def preflight_once():
            src_key = MATS[0][4]
            # what the step about to run reads first: a replay reads the tensor of the recorded / captured call, an eager step the current one
            xin0 = (step_in[0] if (chain is not None or graph is not None) else x[src_key]).clone()
            run_step()
            torch.cuda.synchronize()
            got = {name: [o.clone() for o in outs_l[-1][name]] for name in outs}
            xe = {src_key: xin0} if args.pattern != "independent" else dict(x)
            tmp = {name: [torch.empty_like(o) for o in outs[name]] for name in outs}
            opi = 0
            for li in range(args.layers):
                for name, Mw, K, cnt, slot in MATS:
                    if chain is not None and not getattr(chain, "quarter_walk", False):
                        L.tmac_hip_debug_quad_config(chain.threads, chain.wpq(opi))
                    wr.fused(layers[li][name], xe[slot], tmp[name], N, act_dtype=F16, out_dtype=F16)
                    o0 = tmp[name][0]
                    if args.pattern != "independent":
                        xe[nxt[name]] = (o0.repeat(world)[:logical[name]] if decode else o0.repeat(1, world)[:, :logical[name]].contiguous())
                    opi += 1
            L.tmac_hip_debug_quad_config(0, 0)
            torch.cuda.synchronize()
            same = all(bool(torch.equal(a_, b_)) for name in outs for a_, b_ in zip(got[name], tmp[name]))
            worst = max(float((a_.float() - b_.float()).abs().max() / b_.float().abs().max().clamp_min(1e-30)) for name in outs for a_, b_ in zip(got[name], tmp[name]))
            okf = torch.tensor([1 if (same or worst <= 1e-2) else 0, 1 if same else 0], dtype=torch.int32, device=dev)
            d_all_reduce(okf, dist.ReduceOp.MIN)
            wt = torch.tensor([worst], dtype=torch.float64, device=dev)
            d_all_reduce(wt, dist.ReduceOp.MAX)
            sys.stderr.write(f"bench.py preflight: rank {rank}: step reproduced {'bit for bit' if same else 'to %.3g' % worst}\n")
            return bool(okf[0].item()), bool(okf[1].item()), float("%.3g" % wt.item())
        try:
            ok_, bits_, worst_ = preflight_once()
            tries = 1
            if not ok_:             # once more before the line says so (every rank takes the same decision: the flags are all-reduced)
                ok_, bits_, worst_ = preflight_once()
                tries = 2
            preflight = {"ok": ok_, "bit_identical_on_every_rank": bits_, "max_rel_diff": worst_, "attempts": tries,
                         "what": "one step of the timed path vs the same %d calls launched one by one on each rank with the all-gathers replaced by local repeats "
                                 "(identical synthetic shards), last layer's outputs" % (len(MATS) * args.layers), "path": path_txt}
            if not ok_:
                sys.stderr.write("bench.py: PREFLIGHT FAILED twice: the distributed step differs from its single-rank emulation (%r); the line carries preflight.ok = false\n" % (preflight,))
        except tmac_amd.binding.TMACHipError as e:
            preflight = {"error": repr(e)}
    # Clock ramp (round 6).  After the set-up above the GPU sits at idle clocks, and its power management takes tens of ms of sustained load to
    # reach the steady state: the first 25 launches of the stream-mode step average 0.35 ms, launches 50+ run 0.28 ms (the latency-bound chain:
    # 0.69 -> 0.67; profiles/r06_clock_ramp.txt).  A serving loop lives in the steady state, and W = 5 warm-up steps (what the driver passes) end
    # inside the ramp, so the step is first run, untimed, for --clock-ramp-ms (default 100 ms of GPU time); the W warm-up steps and the K timed
    # steps follow as the contract says.  The line reports the ramp and what the first K steps after idle cost (`clock_ramp`).
    clock_ramp = None
    if args.clock_ramp_ms > 0 and decode:
        cr0 = torch.cuda.Event(enable_timing=True); cr1 = torch.cuda.Event(enable_timing=True); cr2 = torch.cuda.Event(enable_timing=True)
        barrier()
        cr0.record()
        for _ in range(args.steps):
            run_step()
        cr1.record()
        torch.cuda.synchronize()
        first_ms = cr0.elapsed_time(cr1)
        n_ramp = args.steps
        while True:
            for _ in range(min(args.steps, 32)):
                run_step()
            n_ramp += min(args.steps, 32)
            cr2.record()
            torch.cuda.synchronize()
            if cr0.elapsed_time(cr2) >= args.clock_ramp_ms or n_ramp >= 100000:
                break
        clock_ramp = {"untimed_steps": n_ramp, "untimed_ms": round(cr0.elapsed_time(cr2), 1), "first_%d_steps_after_idle_ms_per_step" % args.steps: round(first_ms / args.steps, 4),
                      "what": "the step run untimed until the GPU's clocks reach their steady state under this load, in front of the W warm-up and K timed steps"}
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
    if dpat is not None and any(c.status() != 0 for c in dpat["chains"]):
        raise SystemExit("bench.py: a hand-off inside a decoder segment timed out; outputs invalid")
    if dpat is not None and args.stamps and len(dpat["chains"]) > 2:
        # one segment (layer 1's) launched on its own with stamps: where its time goes (stderr; the timed numbers above are not affected)
        sc = dpat["chains"][2]
        sb = torch.zeros(sc.nops * sc.grid * 8, dtype=torch.int64, device=dev)
        sc.set_stamps(sb)
        for _ in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); sc.launch(); e1.record(); torch.cuda.synchronize()
        st = sb.cpu().numpy().reshape(sc.nops, sc.grid, 8).astype(np.float64) * 0.01
        t0 = st[0, :, 0].min()
        sys.stderr.write("decoder segment with stamps: %.2f us by events; first entry -> last publish %.2f us\n" % (e0.elapsed_time(e1) * 1e3, st[-1, :, 5].max() - t0))
        for oi in range(sc.nops):
            sys.stderr.write("  op %d: entry %.2f..%.2f  activations %.2f..%.2f  LUT built %.2f..%.2f  published %.2f..%.2f (us after the first entry; min..max over workgroups)\n"
                             % (oi, st[oi, :, 0].min() - t0, st[oi, :, 0].max() - t0, st[oi, :, 1].min() - t0, st[oi, :, 1].max() - t0,
                                st[oi, :, 2].min() - t0, st[oi, :, 2].max() - t0, st[oi, :, 5].min() - t0, st[oi, :, 5].max() - t0))
        sc.set_stamps(None)
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        d_all_reduce(t, dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    finite = bool(torch.isfinite((dpat["keep"][3][-1]["down"][0] if dpat is not None else outs["down"][0]).float()).all().item())     # the chained activations stayed finite

    def time_headline_launches(graph_ok):
        """back-to-back stand-alone launches of the GEMV on the headline shape (the down projection of every layer: distinct
        weights > MALL), replayed from a hipGraph where possible, inside a hipEvent pair; seconds per launch, 10 samples"""
        name, Mw, K, cnt, slot = MATS[3]
        xin = torch.randn(K, device=dev, generator=gen).half()
        wr.llama_cpp_init(xin, Mw, K, 1, BITS, act_group_size=ags_of(K), act_dtype=F16)
        hl_out = [torch.empty(shard_rows[name], dtype=torch.float16, device=dev)]
        reps, skip, durs = 10, 3, []

        def headline_launches():
            for li in range(args.layers):
                if fused_calls:
                    wr.fused(layers[li][name], xin, hl_out, 1, act_dtype=F16, out_dtype=F16)
                else:
                    wr.llama_cpp_compute(layers[li][name][0], hl_out[0], 1, out_dtype=F16)
        rgraph = None
        if graph_ok:
            headline_launches()
            torch.cuda.synchronize()
            rgraph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(rgraph):
                headline_launches()
        for r in range(reps + skip):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            rgraph.replay() if rgraph is not None else headline_launches()
            e1.record()
            torch.cuda.synchronize()
            if r >= skip:
                durs.append(e0.elapsed_time(e1) * 1e-3 / args.layers)
        return np.array(durs), reps

    # ---- roofline of the dominant kernel ------------------------------------------------------------------------------
    traffic, traffic_src = None, None
    kkey = {"chain": "k_decode_chain", "fused": "k_gemv_quad_headline", "split": "k_gemv_quad_headline"}[args.path] if decode else "k_gemm_planes"
    is_stream = decode and args.path == "chain" and chain is not None and bool(getattr(chain, "stream", False))
    if is_stream:
        kkey = "k_gemv_stream"          # --pattern independent: the recording runs as k_lut_images + k_gemv_stream
    try:    # HBM bytes per launch of that kernel from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE, x2 gfx950 correction)
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            ent = json.load(f).get(args.workload, {}).get(kkey)
        if ent and ent.get("layers", args.layers) == args.layers:
            traffic, traffic_src = ent["bytes_per_launch"], ent.get("source")
    except Exception:
        pass
    roof = None
    if not decode:
        ach = ops_per_step / world / (ev_ms_per_step * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": "k_gemm_planes (plane-combined one-hot int8 MFMA GEMM; LUT build kernels included in the time)",
                "achieved": round(ach, 1), "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s (int8 MFMA work issued: 2 x output rows x (K/4 tables x 8 half-table entries) x N)",
                "frac": round(ach / MFMA_I8_PEAK_TOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "ops_per_step": ops_per_step, "timing": "hipEvent pair on the launch stream around the %d timed steps" % args.steps}
        # The MFMA work counted above is the one-hot operand's (8 half-table entries per 4 activations: 2 x the dense contraction).  What a
        # user compares with is the DENSE equivalent -- 2 Mw K N flop per matrix -- against the dense bf16 / fp16 MFMA peak, and a dense fp16
        # matmul of the same shapes on this box, replayed from a hipGraph like the timed path.
        dense_flop = sum(cnt * 2.0 * shard_rows[name] * K * N for name, Mw, K, cnt, slot in MATS) * args.layers
        roof["dense_equivalent"] = {"TFLOPs": round(dense_flop / (ev_ms_per_step * 1e-3) / 1e12, 1), "peak": 2500.0,
                                    "frac": round(dense_flop / (ev_ms_per_step * 1e-3) / 1e12 / 2500.0, 4),
                                    "what": "2 x rows x K x N per matrix over the same time, against the dense bf16 MFMA peak (MI355X_MICROARCH.md)"}
        if not dist_on:
            measure_dense_fp16(ctx, roof, ev_ms_per_step)
    elif args.path == "chain":
        ach = bytes_per_step / (ev_ms_per_step * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": ("k_lut_images + k_gemv_stream: the token's %d GEMVs as independent calls, tables prebuilt once per call, one "
                                           "persistent launch for the lookups" % (7 * args.layers)) if is_stream else
                                          ("k_decode_chain: one persistent launch per decoded token (all %d GEMVs, LUT builds and in-kernel "
                                           "hand-offs of the layer stack)" % (7 * args.layers)) if dpat is None else
                                          ("k_decode_chain, decoder pattern: %d launches per token (first q/k/v; per layer the segment o -> [+ residual, RMSNorm] -> "
                                           "gate/up -> [silu(gate) * up] -> down -> [+ residual, RMSNorm] -> next q/k/v with the element-wise operators inside "
                                           "the LUT builds) and a copy kernel between the segments (stand-in for attention), replayed from a hipGraph"
                                           % len(dpat["chains"])),
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": bytes_per_step, "avg_launch_us": round(ev_ms_per_step * 1e3, 2), "launches_timed": args.steps,
                "timing": "hipEvent pair on the launch stream around the %d timed launches" % args.steps}
        # BASELINE.json's target shape as a kernel of its own (outside the timed region): the down projection of every layer,
        # LUT build fused, back-to-back launches over distinct weights replayed from a hipGraph
        name, Mw, K, cnt, slot = MATS[3]
        hdurs, _ = time_headline_launches(True)
        hb = algorithmic_bytes(Mw, K, BITS, GS, ags_of(K), ZP, MG)
        hus = float(np.mean(hdurs)) * 1e6
        roof["headline_gemv"] = {"shape": f"{Mw}x{K} N=1 (k_gemv_quad, LUT build fused, one launch per GEMV)", "us": round(hus, 3),
                                 "min_us": round(float(np.min(hdurs)) * 1e6, 3), "algorithmic_bytes": hb,
                                 "GBps": round(hb / hus * 1e-3, 1), "frac": round(hb / hus * 1e-3 / HBM_PEAK_GBS, 4),
                                 "timing": "hipEvent pair around %d back-to-back launches (distinct weights, hipGraph replay), mean of 10" % args.layers}
        # The streaming core alone: ONE persistent launch over the headline GEMV of every layer (distinct weights >
        # MALL), every call fed by the same external vector -- no hand-off anywhere, so what is left is activation fetch, LUT build,
        # lookups, reduction and publish of a workgroup, call after call.  Separates "issue / structure bound" from "latency bound".
        if not dist_on and not args.no_stream_core:
            # SURVEY 8(d)'s headline measurement: back-to-back INDEPENDENT GEMVs of the target shape over rotating distinct weights (> MALL in
            # total), a distinct activation vector per call.  Nothing is handed over, so the recording runs in stream mode: tables once per
            # call by k_lut_images (the reference's llama_cpp_init), lookups by k_gemv_stream (its llama_cpp_compute); both launches timed.
            measure_stream_calls(ctx, roof, name, Mw, K)
        # ... and the whole token's 224 matrices as independent calls (what --pattern independent times): every call reads a resident vector.
        # With several ranks: every rank streams ITS row shard of every matrix, nothing is exchanged (rows split, K whole, vectors resident on
        # every rank) -- the decode-side workload that scales by construction, reported beside the dependent chain's value as
        # roofline.independent_pattern (n_gpus ranks, aggregate bytes over the slowest rank's time)
        if (not args.no_stream_core) and decode and args.path == "chain" and dpat is None and args.pattern != "independent":
            measure_independent_pattern(ctx, roof)
        # the same matrices as a decoder issues them (outside the timed region; --pattern decoder makes it the timed workload)
        if not dist_on and dpat is None and args.pattern == "chained" and not args.no_decoder_pattern:
            try:
                dms, dok, dn, dgraphed, dout = time_decoder_pattern(ctx)
                roof["decoder_pattern"] = {"what": "one launch per segment (o -> gate/up -> down -> next q/k/v; residual add + RMSNorm and silu(gate) * up inside the "
                                                   "LUT builds), a copy kernel between the segments as stand-in for attention",
                                           "ms_per_token": round(dms, 4), "launches": dn,
                                           "outside_ms": None if dout is None else round(dout, 4),      # the 32 stand-in copies alone (same graph without the segments)
                                           "GBps": round(bytes_per_step / (dms * 1e-3) / 1e9, 1),
                                           "frac": round(bytes_per_step / (dms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "ok": dok,
                                           "timing": "hipEvent pair around a %s of the token's launches, mean of 10" % ("hipGraph replay" if dgraphed else "eager sequence")}
            except Exception as e:
                roof["decoder_pattern"] = {"error": repr(e)}
        if args.stamps and chain is not None:
            raw = stamp_buf.cpu().numpy().reshape(chain.nops, chain.grid, 8)
            try:
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                np.save(os.path.join(ROOT, "gpurun_out", "chain_stamps.npy"), raw)
            except Exception:
                pass
            st = raw[:, :, :7].astype(np.float64) * 0.01                     # s_memrealtime: 100 MHz -> us (wave 0 of every workgroup)
            ends = st[:, :, 5].max(axis=1)                                    # a call is complete when its last row quad is published
            dur = ends - np.concatenate([[st[0, :, 0].min()], ends[:-1]])
            per = {}
            for k, (name, Mw, K, cnt, slot) in enumerate(MATS):
                sel = st[k::4]
                per[name] = {"us": round(float(np.mean(dur[k::4])), 3),
                             "wait_input_us": round(float(np.mean(sel[:, :, 1] - sel[:, :, 0])), 3),
                             "lut_build_us": round(float(np.mean(sel[:, :, 2] - sel[:, :, 1])), 3),
                             "lookups_us": round(float(np.mean(sel[:, :, 5] - sel[:, :, 2])), 3),
                             "polls": round(float(np.mean(raw[k::4, :, 7])), 2)}
            name, Mw, K, cnt, slot = MATS[3]
            hb = algorithmic_bytes(Mw, K, BITS, GS, ags_of(K), ZP, MG)
            roof["per_call_from_stamps"] = per
            roof["headline_gemv_inside_chain"] = {"shape": f"{Mw}x{K} (the down projection inside the launch)", "us": per["down"]["us"],
                                                  "GBps": round(hb / (per["down"]["us"] * 1e-6) / 1e9, 1), "frac": round(hb / (per["down"]["us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
    else:
        # per-launch paths: the headline GEMV is the dominant kernel
        name, Mw, K, cnt, slot = MATS[3]
        durs, reps = time_headline_launches(use_graph)
        hb = algorithmic_bytes(shard_rows[name], K, BITS, GS, ags_of(K), ZP, MG)
        ach = hb / float(np.mean(durs)) / 1e9
        floor = None
        if use_graph and args.floors:
            # floor of this launch structure, measured the same way: launches that only READ the same number of bytes
            # (distinct buffers, > MALL in total) and near-empty launches, as dependent nodes of a replayed graph
            nb = (hb + 4095) // 4096 * 4096
            bufs = [torch.empty(nb, dtype=torch.uint8, device=dev).fill_(0x5a) for _ in range(args.layers)]
            sink = torch.zeros(4096, dtype=torch.uint8, device=dev)

            def floor_time(nbytes):
                cs = torch.cuda.current_stream().cuda_stream
                for b in bufs:
                    tmac_amd.binding.check(L.tmac_hip_debug_stream_read(b.data_ptr(), nbytes, sink.data_ptr(), cs))
                torch.cuda.synchronize()
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
            floor = {"pure_read_same_bytes_us": round(floor_time(hb // 16 * 16) * 1e6, 3), "near_empty_launch_us": round(floor_time(4096) * 1e6, 3)}
            del bufs
        roof = {"bound": "hbm", "kernel": ("k_gemv_quad, LUT build fused" if fused_calls else "k_gemv_quad, LUT prebuilt") + f", headline shape {Mw}x{K}",
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": hb, "avg_launch_us": round(float(np.mean(durs)) * 1e6, 3),
                "min_launch_us": round(float(np.min(durs)) * 1e6, 3), "launches_timed": reps * args.layers, "launch_floor": floor,
                "timing": "hipEvent pair on the launch stream around %d back-to-back launches (distinct weights, %s), mean of 10"
                          % (args.layers, "hipGraph replay" if use_graph else "eager")}

    # ---- verification (outside the timed region): the launches being timed, at full size, against the oracle ----------
    verified = verify_against_oracle(ctx)

    if rank == 0:
        sec = ms_per_step * 1e-3
        res = {
            "metric": wl["metric"],
            "value": round(bytes_per_step / sec / 1e9, 2) if decode else round(N / sec, 1),
            "unit": "GB/s" if decode else "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "int8",
            "data": "synthetic",
            "tokens_per_s": round(N / sec, 1),
            "algorithmic_GBps": round(bytes_per_step / sec / 1e9, 2),
            "frac_of_hbm_peak": round(bytes_per_step / sec / 1e9 / (HBM_PEAK_GBS * world), 4),
            "config": {"workload": wl["tag"], "layers": args.layers, "N": N,
                       "gemm_per_step": 7 * args.layers,
                       "launches_per_step": (len(dpat["chains"]) if dpat is not None else (2 if is_stream else 1)) if args.path == "chain" else (4 if fused_calls else 11) * args.layers + (0 if decode else 4 * args.layers), "path": args.path,
                       "pattern": args.pattern,
                       **({"outside_ms": (lambda v: None if v is None else round(v, 4))(time_outside_ops(ctx, dpat["dstep"]))} if dpat is not None else {}),
                       "autotune": tuned,
                       "algorithmic_bytes_per_step": bytes_per_step, "weights": wl["weights"],
                       "parallelism": f"row-shard x{world}" if world > 1 else "single GPU",
                       "kernel_variant": args.variant,
                       "launch": (("k_lut_images (the LUT builds of all calls) + ONE persistent launch of k_gemv_stream per step" + (" and rank; rows sharded, nothing exchanged" if world > 1 else ""))
                                  if is_stream else
                                  "one persistent launch per step" + (" and rank, hand-off across ranks through IPC-mapped arenas" if world > 1 else ""))
                                 if args.path == "chain" else ("hipGraph replay" if use_graph else "eager")},
            "roofline": roof,
            "clock_ramp": clock_ramp,
            "preflight": preflight,
            "verified": verified,
            "activations_finite": finite,
            "event_ms_per_step": round(ev_ms_per_step, 4),
            "cpu_baseline": None,
        }
        if world == 1 and not args.no_cpu_baseline and decode:
            try:
                res["cpu_baseline"] = cpu_baseline(args.workload)
            except Exception as e:  # the baseline is a reported extra, never a reason to lose the GPU number
                res["cpu_baseline"] = {"error": repr(e)}
        elif world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline_prefill(args.workload)
            except Exception as e:
                res["cpu_baseline"] = {"error": repr(e)}
    else:
        res = None
    # release this workload's device memory (a multi-GPU decode run measures the prefill twin next, in the same process)
    try:
        if chain is not None:
            chain.free()
        for mats in layers:
            for ws in mats.values():
                for w_ in ws:
                    w_.free()
        torch.cuda.synchronize()
    except Exception:
        pass
    return res


PREFILL_TWIN = {"llama2-7b-w2": "llama2-7b-w2-prefill", "llama2-7b-w4": "llama2-7b-w4-prefill", "bitnet-3b": "bitnet-3b-prefill"}


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
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"), RANK="0", WORLD_SIZE="1")
        if args.share_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    env = dict(world=world, rank=rank, local_rank=local_rank, dist_on=dist_on)
    auto = args.pattern == "auto"
    decode_wl = WORKLOADS[args.workload]["N"] == 1
    res = run(args, env)
    # The DEPENDENT form of the same token (rounds 1-5's timed workload): the 224 calls linked by real data -- x1 = q, x2 = o, x3 = gate, next
    # x0 = down -- so that every call waits for its predecessor's outputs inside ONE persistent launch (k_decode_chain: in-kernel hand-offs,
    # LUT build per call and CU; with ranks: hand-off granules stored into every rank's IPC-mapped arena).  Measured in the same run, same
    # matrices' shapes and bytes, verified against the oracle, reported as `dependent_chain`; the decoder pattern (one launch per segment
    # between two attentions) rides in that run.
    if auto and decode_wl and args.path in ("auto", "chain") and args.variant == 0 and not args.stamps:
        import copy
        a2 = copy.copy(args)
        a2.pattern, a2.path = "chained", "chain"
        a2.steps, a2.warmup, a2.no_cpu_baseline, a2.no_stream_core = max(args.steps // 3, 50), 10, True, True
        try:
            r2 = run(a2, env)
            if res is not None and r2 is not None:
                rr = dict(r2.get("roofline") or {})
                dp = rr.pop("decoder_pattern", None)
                rr.pop("independent_pattern", None)
                res["dependent_chain"] = {"what": "the same %d mpGEMMs chained by real data (every call consumes an earlier call's output inside one persistent launch of "
                                                  "k_decode_chain): the timed workload of rounds 1-5" % r2["config"]["gemm_per_step"],
                                          "ms_per_step": r2["ms_per_step"], "value": r2["value"], "unit": r2["unit"], "steps": r2["steps"], "n_gpus": r2["n_gpus"],
                                          "frac_of_hbm_peak": r2.get("frac_of_hbm_peak"), "roofline": rr, "verified": r2.get("verified"),
                                          "preflight": r2.get("preflight"), "launch": r2["config"]["launch"]}
                if dp is not None:
                    res["roofline"]["decoder_pattern"] = dp
        except BaseException as e:      # the line's value is never lost to the extra measurement
            if res is not None:
                res["dependent_chain"] = {"error": repr(e)}
    # Scaling headline.  Decode is a chain of dependent GEMVs whose hand-off and LUT build do not shrink with the number of GPUs (DESIGN 6:
    # 1.3 / 1.6 / 1.9 x at 2 / 4 / 8 by the model); the prefill GEMM of the same matrices (N = 256, BASELINE config 5) is compute-bound and
    # splits by rows with one all-gather per call -- so a multi-GPU run also measures it, on the same ranks, and rank 0 reports both.
    if world > 1 and WORKLOADS[args.workload]["N"] == 1 and args.workload in PREFILL_TWIN and not args.no_prefill_headline:
        import copy
        a2 = copy.copy(args)
        a2.workload, a2.path, a2.pattern = PREFILL_TWIN[args.workload], "auto", "chained"
        a2.steps, a2.warmup, a2.no_cpu_baseline, a2.no_verify, a2.stamps = 10, 2, True, True, False
        a2.layers = min(args.layers, WORKLOADS[a2.workload]["layers"])
        try:
            r2 = run(a2, env)
            head = None if r2 is None else {k: r2.get(k) for k in ("metric", "value", "unit", "n_gpus", "ms_per_step", "steps", "scaling", "dtype")}
            if head is not None:
                head["workload"] = r2["config"]["workload"]
                head["launch"] = r2["config"]["launch"]
                head["preflight"] = r2.get("preflight")
        except BaseException as e:      # the decode number is never lost to the extra measurement
            head = {"error": repr(e)}
        if res is not None:
            res["prefill_scaling_headline"] = head
    # One GPU: the prefill twin rides along in the default line too (N = 256 over the same matrices, with its graph-replayed dense fp16 baseline),
    # so that the driver's own run carries a prefill number, not only the builder's.  Outside the timed region.
    if (world == 1 and not dist_on and res is not None and WORKLOADS[args.workload]["N"] == 1 and args.workload in PREFILL_TWIN
            and not args.no_prefill_headline and auto and not args.stamps):
        import copy
        a2 = copy.copy(args)
        a2.workload, a2.path, a2.pattern = PREFILL_TWIN[args.workload], "auto", "chained"
        a2.steps, a2.warmup, a2.no_cpu_baseline, a2.no_verify = 30, 5, True, True
        a2.layers = min(args.layers, WORKLOADS[a2.workload]["layers"])
        try:
            r2 = run(a2, env)
            r2r = r2.get("roofline") or {}
            res["prefill_twin"] = {"workload": r2["config"]["workload"], "ms_per_step": r2["ms_per_step"], "value": r2["value"], "unit": r2["unit"],
                                   "steps": r2["steps"], "frac_of_int8_mfma_peak": r2r.get("frac"), "dense_equivalent": r2r.get("dense_equivalent"),
                                   "dense_fp16_baseline": r2r.get("dense_fp16_baseline")}
        except BaseException as e:      # the decode number is never lost to the extra measurement
            res["prefill_twin"] = {"error": repr(e)}
    # ... and so do the other two decode configurations of BASELINE.json (W4 GPTQ-style, BitNet-b1.58-3B): the same two measurements (the token's
    # calls as independent calls; as one dependent chain), 200 launches each, outside the timed region, so that the driver's run carries them
    if (world == 1 and not dist_on and res is not None and args.workload == "llama2-7b-w2" and auto and args.path in ("auto", "chain")
            and not args.no_prefill_headline and not args.stamps and args.layers == WORKLOADS[args.workload]["layers"]):
        import copy
        res["other_decode_workloads"] = {}
        for w2 in ("llama2-7b-w4", "bitnet-3b"):
            ent = {}
            for pat in ("independent", "chained"):
                a2 = copy.copy(args)
                a2.workload, a2.pattern, a2.path, a2.steps, a2.warmup, a2.no_cpu_baseline, a2.no_verify = w2, pat, "chain", 200, 10, True, True
                a2.no_stream_core, a2.no_decoder_pattern, a2.layers = True, True, WORKLOADS[w2]["layers"]
                try:
                    r2 = run(a2, env)
                    ent["independent" if pat == "independent" else "dependent_chain"] = {
                        "ms_per_step": r2["ms_per_step"], "value": r2["value"], "unit": r2["unit"], "steps": r2["steps"], "frac": (r2.get("roofline") or {}).get("frac")}
                except BaseException as e:
                    ent[pat] = {"error": repr(e)}
            ent["workload"] = WORKLOADS[w2]["tag"]
            res["other_decode_workloads"][w2] = ent
    if rank == 0 and res is not None:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(res) + "\n").encode())
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
