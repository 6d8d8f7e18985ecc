"""Stream mode of the recorded call sequence (k_lut_images + k_gemv_stream, tmac_amd/csrc/tmac_stream.hip): a recording in which no
call consumes another call's output -- SURVEY 8(d)'s back-to-back GEMVs -- runs with the tables prebuilt once per call (the reference's
own structure: llama_cpp_init, then lookups only, tmac_gemm_wrapper.h:170-228), a loader wave staging the next call's tables and a
weight prefetch that crosses call boundaries.

Bars (the harness of test_gpu_chain.py): every call's outputs (a) BIT-IDENTICAL to the same call launched on its own through
tmac_hip_qgemm_fused_dev with the chain's launch configuration -- whose integer path test_gpu_parity.py taps bit for bit against the
oracle and the reference-made goldens -- and (b) within 1e-3 of the oracle (lut_ctor.cc / tbl.cc restated in oracle/tmac_oracle.c).
"""
import os

import numpy as np
import pytest

from test_gpu_chain import Model, tm, _short_spin      # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["auto", "quad64"])
def _form(request, monkeypatch):
    """k_gemv_stream has two forms (tmac_stream.hip): row quad x 64 units per item -- per-group-scale outputs bit-identical to the
    stand-alone launches -- and the quarter-walk form (16 rows x 16 units; taken by default when some K of the recording has a ragged last
    step and every matrix has a multiple of 16 rows), whose integers are the same (check_tap) and whose fp32 outputs are held to the
    oracle's tolerance.  Every test runs with the default choice and with the first form forced."""
    if request.param == "quad64":
        monkeypatch.setenv("TMAC_STREAM_QW", "0")


def _run(tm, ops, reps=2, **kw):
    m = Model(tm, ops, **kw)
    chain = m.record()
    assert chain.stream, "a recording without data flow between its calls is a stream"
    for rep in range(reps):
        chain.launch()
        m.check(chain, oracle_ops=None if rep == 0 else [])
    m.check_tap(chain)          # the integers inside k_gemv_stream itself against the oracle (tmac_hip_chain_set_tap)
    chain.free()
    m.free()


# (K, rows of the matrices, None): a lone call; few / many quads per workgroup; split quads (K >= 2 steps of 2048 activations);
# a K whose last step is partly padding (2688 = 1.3 steps, 6144 = 3 steps); a fused triple; more matrices than workgroups have quads for
INDEP = [
    (1024, [512, 256], None),
    (2688, [640], None),
    (4096, [1024, 1024, 1024], None),
    (128, [64], None),
    (6144, [256], None),
    (4096, [4224, 4224], None),
    (256, [128, 64, 64, 128], None),
    (3200, [3200], None),
]


@pytest.mark.parametrize("bits,zp,dev_f16", [(2, True, True), (2, False, False), (4, True, True), (4, False, False), (1, True, True), (3, False, True)])
def test_stream_of_independent_calls(tm, bits, zp, dev_f16):
    _run(tm, INDEP, bits=bits, zp=zp, dev_f16=dev_f16, seed=3)


def test_stream_single_call_and_repeats(tm):
    _run(tm, [(4096, [4096], None)], reps=3, seed=5)
    _run(tm, [(11008, [1024], None)], reps=3, seed=6)       # the headline's K: two rounds of table pairs, six steps (the last one 3/8 full)


@pytest.mark.parametrize("bits", [2, 4])
def test_stream_full_size_llama_shapes(tm, bits):
    """stream mode at BASELINE's full sizes (llama-2-7B: the 4096 x 11008 target shape, gate/up 2 x 11008 x 4096, q/k/v 3 x 4096 x 4096,
    o), two calls of each so that the schedule deals them to classes of row ranges: every output against the stand-alone launch (bits),
    the oracle (1e-3), and k_gemv_stream's own integers against the oracle (array_equal)"""
    ops = [(11008, [4096], None), (4096, [11008, 11008], None), (4096, [4096, 4096, 4096], None), (4096, [4096], None)] * 2
    _run(tm, ops, reps=1, bits=bits, seed=31 + bits)


@pytest.mark.parametrize("mg", [-1, 1])
def test_stream_largest_K(tm, mg, monkeypatch):
    """K = 16384 and 18432 (the persistent kernels' limit, 8 x 3 x 768): the second LUT buffer starts above 64 KB of LDS -- the loader wave's
    buffer_load-to-LDS M0 base, the long chunk-sum / bias chain of k_lut_images_us -- with K = 16384 two workgroups per CU still fit
    (2 x 72 KB), with 18432 one; per-group and unified scales.  k_gemv_quad has no launch configuration of its own for these K
    with the chain's workgroup size, so the bars are: the oracle (1e-3; unified scales: bits), k_gemv_stream's own integers against the
    oracle (array_equal), and the same recording through k_decode_chain (TMAC_CHAIN_STREAM=0) bit for bit."""
    import torch
    from test_gpu_chain import rel_err
    monkeypatch.setenv("TMAC_STREAM_NCLS", "1")      # every row range visits every call, as k_decode_chain's workgroups do (same waves per quad)
    monkeypatch.setenv("TMAC_STREAM_QW", "0")        # (the quarter-walk form of these K is covered by the oracle and the tap in the other tests)
    for ops, seed in (([(16384, [256], None), (16384, [64, 128], None), (16384, [1024], None)], 41), ([(18432, [128, 128], None), (18432, [512], None)], 42)):
        m = Model(tm, ops, mg=mg, seed=seed)
        s = m.record()
        assert s.stream
        s.launch(); torch.cuda.synchronize()
        assert s.status() == 0
        got = [[o.clone() for o in os_] for os_ in m.outs]
        for i in range(len(ops)):
            want = m.oracle_outputs(i, m.x_ext[i].float().cpu().numpy())
            for k in range(len(want)):
                g = got[i][k].cpu().numpy()
                if mg >= 1:
                    assert np.array_equal(g.view(np.uint16), want[k].astype(np.float16).view(np.uint16))
                assert rel_err(g.astype(np.float32), want[k]) <= 1e-3
        m.check_tap(s)
        for os_ in m.outs:
            for o in os_:
                o.zero_()
        monkeypatch.setenv("TMAC_CHAIN_STREAM", "0")
        c = m.record()
        monkeypatch.delenv("TMAC_CHAIN_STREAM")
        assert not c.stream
        c.launch(); torch.cuda.synchronize()
        assert c.status() == 0
        for x, y in zip(got, m.outs):
            for p, q in zip(x, y):
                assert torch.equal(p, q)
        s.free(); c.free(); m.free()


def test_stream_fp32_outputs_and_fp32_activations(tm):
    _run(tm, INDEP[:5], out_f16=False, seed=7)
    _run(tm, INDEP[:5], ext_f32=True, seed=8)


@pytest.mark.parametrize("grid", [1, 7, 8, 13, 96, 200])
def test_stream_on_fewer_workgroups(tm, grid):
    """fewer workgroups than CUs (another kernel holds the rest; a partitioned device): more quads per workgroup, several workgroup
    iterations per call, waves without items in some calls"""
    tm.binding.check(tm.lib().tmac_hip_debug_chain_grid(grid))
    try:
        _run(tm, INDEP, seed=9)
    finally:
        tm.binding.check(tm.lib().tmac_hip_debug_chain_grid(0))


def test_long_stream(tm):
    """260 calls in one recording: the workgroups' descriptor copies fill a third of LDS, the role table has 260 records, each of the two
    workgroups of a CU walks 130 ops"""
    _run(tm, [(128, [64], None), (256, [64, 64], None), (640, [128], None), (1024, [256, 64], None)] * 65, reps=1, seed=15)


def _random_stream(seed):
    rng = np.random.default_rng(seed)
    ops = []
    for _ in range(int(rng.integers(2, 10))):
        K = int(rng.choice([128, 256, 640, 1024, 2688, 3200, 4096, 6144, 11008]))
        rows = [int(rng.choice([64, 128, 256, 640, 1024, 2688, 3200, 4224])) for _ in range(int(rng.integers(1, 5)))]
        ops.append((K, rows, None))
    return ops


@pytest.mark.parametrize("seed", range(int(os.environ.get("TMAC_FUZZ_STREAMS", "10"))))
def test_random_streams(tm, seed):
    rng = np.random.default_rng(2000 + seed)
    bits, zp, dev_f16 = int(rng.integers(1, 5)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    mg = int(rng.choice([-1, -1, 1, 2]))        # per-group scales, or 1 / 2 unified scales per matrix (every row count above is even)
    _run(tm, _random_stream(seed), bits=bits, zp=zp, dev_f16=dev_f16, mg=mg, seed=80 + seed)


def test_stream_equals_the_ordinary_chain(tm, monkeypatch):
    """TMAC_CHAIN_STREAM=0 keeps k_decode_chain for the same recording: both launches, same bits -- with the schedule in which every row
    range visits every call (TMAC_STREAM_NCLS=1), as k_decode_chain's workgroups do: a call dealt to fewer ranges may take another number of
    waves per row quad, i.e. another order of its fp32 partial sums (the stand-alone comparison above follows the chain's choice)"""
    import torch
    monkeypatch.setenv("TMAC_STREAM_NCLS", "1")
    monkeypatch.setenv("TMAC_STREAM_QW", "0")        # ... and the item form whose order of fp32 partial sums is k_decode_chain's
    m = Model(tm, INDEP, seed=11)
    s = m.record()
    assert s.stream
    s.launch(); torch.cuda.synchronize()
    a = [[o.clone() for o in os_] for os_ in m.outs]
    for os_ in m.outs:
        for o in os_:
            o.zero_()
    monkeypatch.setenv("TMAC_CHAIN_STREAM", "0")
    c = m.record()
    assert not c.stream
    c.launch(); torch.cuda.synchronize()
    assert c.status() == 0
    for x, y in zip(a, m.outs):
        for p, q in zip(x, y):
            assert torch.equal(p, q)
    s.free(); c.free(); m.free()


def test_dependent_recordings_are_not_streams(tm):
    m = Model(tm, [(1024, [1024], None), (1024, [256], (0, 0))], seed=12)
    c = m.record()
    assert not c.stream
    c.launch(); m.check(c)
    c.free(); m.free()


# Unified scales (BitNet: one act group per row, int32 totals + scale-final: tbl.cc:536-630, qgemm.py:170-174): k_lut_images_us builds the
# row's scale and the sequential bias chain once per call, the service wave applies scale-final.  Same two bars; the oracle bar is exact
# here (the harness compares unified-scale outputs with the oracle's fp32 result rounded once).
US_OPS = [
    (1024, [512, 256], None),
    (3200, [3200], None),                      # BitNet-3B's q / k / v / o shape
    (8640, [640], None),                       # ... its down projection's K: 4.2 steps, 270 chunk sums in the bias chain
    (3200, [1088, 1088], None),
    (128, [64], None),
    (6144, [256], None),
]


@pytest.mark.parametrize("bits,mg,dev_f16,ternary", [(2, 1, True, True), (2, 1, False, False), (1, 1, True, False), (3, 2, True, False), (4, 1, False, False), (2, 4, True, False)])
def test_stream_of_unified_scale_calls(tm, bits, mg, dev_f16, ternary):
    _run(tm, US_OPS, bits=bits, mg=mg, dev_f16=dev_f16, ternary=ternary, seed=21)


def test_stream_unified_scales_fp32_and_small_grids(tm):
    _run(tm, US_OPS[:4], mg=1, out_f16=False, seed=22)
    _run(tm, US_OPS[:4], mg=1, ext_f32=True, seed=23)
    tm.binding.check(tm.lib().tmac_hip_debug_chain_grid(96))
    try:
        _run(tm, US_OPS, mg=1, ternary=True, seed=24)
    finally:
        tm.binding.check(tm.lib().tmac_hip_debug_chain_grid(0))


def test_unified_scale_stream_equals_the_ordinary_chain(tm, monkeypatch):
    import torch
    m = Model(tm, US_OPS, mg=1, ternary=True, seed=25)
    s = m.record()
    assert s.stream
    s.launch(); torch.cuda.synchronize()
    a = [[o.clone() for o in os_] for os_ in m.outs]
    for os_ in m.outs:
        for o in os_:
            o.zero_()
    monkeypatch.setenv("TMAC_CHAIN_STREAM", "0")
    c = m.record()
    assert not c.stream
    c.launch(); torch.cuda.synchronize()
    assert c.status() == 0
    for x, y in zip(a, m.outs):
        for p, q in zip(x, y):
            assert torch.equal(p, q)
    s.free(); c.free(); m.free()


def test_stream_launch_inside_a_hip_graph(tm):
    import torch
    m = Model(tm, INDEP[:4], seed=14)
    c = m.record()
    assert c.stream
    c.launch(); torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        c.launch()
    for os_ in m.outs:
        for o in os_:
            o.zero_()
    g.replay()
    m.check(c)
    c.free(); m.free()


# Deferred launches (tmac_hip_defer / tmac_hip_flush, round 6): a caller that does not record queues its N = 1 calls; a flush launches the queue
# as ONE stream-mode launch; a call that reads or overwrites anything a queued call writes flushes the queue first.  Results must be those of
# launching the calls in order.
@pytest.mark.parametrize("seed", range(6))
def test_deferred_launches_equal_in_order_launches(tm, seed):
    import ctypes as C
    import torch
    from test_gpu_chain import rel_err
    rng = np.random.default_rng(300 + seed)
    # a random sequence with dependences: ~half of the calls read an earlier call's output (RAW: must flush), some rewrite an output buffer
    n = int(rng.integers(4, 9))
    ops = []
    for i in range(n):
        feeders = [(j, m) for j in range(i) for m in range(len(ops[j][1])) if ops[j][1][m] % 128 == 0]
        if feeders and rng.random() < 0.5:
            src = feeders[int(rng.integers(len(feeders)))]
            K = ops[src[0]][1][src[1]]
        else:
            src, K = None, int(rng.choice([256, 1024, 2688, 4096]))
        rows = [int(rng.choice([128, 256, 1024])) for _ in range(int(rng.integers(1, 4)))]
        ops.append((K, rows, src))
    m = Model(tm, ops, seed=400 + seed)
    L = tm.lib()
    # in-order launches
    m.issue(); torch.cuda.synchronize()
    want = [[o.clone() for o in os_] for os_ in m.outs]
    for os_ in m.outs:
        for o in os_:
            o.zero_()
    tm.binding.check(L.tmac_hip_defer(1))
    try:
        for rep in range(3):                       # the second and third token hit the cache of recordings
            m.issue()
            tm.binding.check(L.tmac_hip_flush(None))
            torch.cuda.synchronize()
            for a_, b_ in zip(want, m.outs):
                for p_, q_ in zip(a_, b_):
                    # (a batch runs k_gemv_stream, whose waves per quad / quarter-walk form may differ from the stand-alone launch's: fp16 ulps)
                    assert rel_err(q_.float().cpu().numpy(), p_.float().cpu().numpy()) <= 2e-3
        st = [C.c_uint64(0) for _ in range(4)]
        tm.binding.check(L.tmac_hip_defer_stats(*[C.byref(x) for x in st]))
        flushes, hits, streams, singles = [int(x.value) for x in st]
        assert flushes >= 3 and hits >= 2 * (flushes // 3) - 1, (flushes, hits, streams, singles)
    finally:
        tm.binding.check(L.tmac_hip_defer(0))
    # ... and against the oracle, call by call, on the vectors the deferred run consumed
    for i, (K, rows, src) in enumerate(ops):
        x = m.x_of(i).float().cpu().numpy()
        ref = m.oracle_outputs(i, x)
        for k in range(len(rows)):
            assert rel_err(m.outs[i][k].float().cpu().numpy(), ref[k]) <= 1e-3
    m.free()


def test_deferred_batch_of_mixed_widths_is_one_stream_launch_per_configuration(tm):
    """qgemm.py:98-116 allows any mix of widths and zero-point settings between the matrices a caller multiplies; a recording refuses the
    mix (one kernel instantiation per launch), the deferred queue regroups it: the calls of a batch are independent"""
    import ctypes as C
    import torch
    from test_gpu_chain import rel_err
    ops = [(1024, [256, 512], None), (2688, [128], None), (4096, [1024], None)]
    models = [Model(tm, ops, bits=2, zp=True, seed=31), Model(tm, ops, bits=4, zp=True, seed=32), Model(tm, ops, bits=2, zp=False, dev_f16=False, seed=33),
              Model(tm, [(3200, [640], None)], bits=2, mg=1, seed=34),           # (one call alone in its configuration -> launched by itself)
              Model(tm, ops[:2], bits=1, zp=True, seed=35)]                      # (two calls: cheaper one by one than as a stream launch, profiles/r06_stream_small_batches.txt)
    L = tm.lib()

    def issue_interleaved():
        for i in range(len(ops)):
            for m in models:
                if i < len(m.ops):
                    m.wr.fused(m.ws[i], m.x_of(i), m.outs[i], 1, act_dtype=m.act_dtype(i))

    issue_interleaved(); torch.cuda.synchronize()
    want = [[[o.clone() for o in os_] for os_ in m.outs] for m in models]
    for m in models:
        for os_ in m.outs:
            for o in os_:
                o.zero_()
    st = [C.c_uint64(0) for _ in range(4)]
    tm.binding.check(L.tmac_hip_defer_stats(*[C.byref(x) for x in st]))
    before = [int(x.value) for x in st]
    tm.binding.check(L.tmac_hip_defer(1))
    try:
        for rep in range(2):
            issue_interleaved()
            tm.binding.check(L.tmac_hip_flush(None))
            torch.cuda.synchronize()
            for wm, m in zip(want, models):
                for a_, b_ in zip(wm, m.outs):
                    for p_, q_ in zip(a_, b_):
                        assert rel_err(q_.float().cpu().numpy(), p_.float().cpu().numpy()) <= 2e-3
        tm.binding.check(L.tmac_hip_defer_stats(*[C.byref(x) for x in st]))
        flushes, hits, streams, singles = [int(x.value) - b for x, b in zip(st, before)]
        assert (flushes, hits, streams, singles) == (2, 1, 6, 6), (flushes, hits, streams, singles)      # three configurations of three calls + a lone call + a pair, twice
    finally:
        tm.binding.check(L.tmac_hip_defer(0))
    for m in models:
        for i, (K, rows, src) in enumerate(m.ops):
            ref = m.oracle_outputs(i, m.x_of(i).float().cpu().numpy())
            for k in range(len(rows)):
                assert rel_err(m.outs[i][k].float().cpu().numpy(), ref[k]) <= 1e-3
        m.free()
