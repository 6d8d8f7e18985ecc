"""The persistent decode chain (k_decode_chain, tmac_hip_chain_*): a recorded sequence of fused N = 1 calls executed by one
launch, outputs handed from call to call inside the kernel.

Bar: every call's outputs (a) BIT-IDENTICAL to the same call launched on its own through tmac_hip_qgemm_fused_dev with
768-thread workgroups and the chain's number of waves per row quad, fed the activation vector the chain actually produced,
and (b) within 1e-3 (max-norm, fp16 outputs) of the oracle run on that activation vector (lut_ctor.cc / tbl.cc restated in
oracle/tmac_oracle.c).  (a) ties the chain to the kernel whose integer path is tapped bit for bit in test_gpu_parity.py;
(b) ties it to the reference independently of that kernel.
"""
import os

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

BITS_BM = {1: 64, 2: 128, 3: 192, 4: 256}      # (1-bit: 64 rows per tile so that the 64-row matrices below stay tileable)
KF, GS, AGS = 16, 128, 64


@pytest.fixture(scope="module")
def tm():
    import torch
    import tmac_amd
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    assert tmac_amd.lib().tmac_hip_device_count() > 0
    return tmac_amd


@pytest.fixture(autouse=True)
def _short_spin(tm):
    # after conftest's per-test reset: a broken hand-off fails in ~0.2 s, not 2 s
    tm.binding.check(tm.lib().tmac_hip_debug_chain_config(0, 1 << 17))


def rel_err(c, ref):
    return float(np.abs(c.astype(np.float64) - ref.astype(np.float64)).max() / max(np.abs(ref).max(), 1e-30))


class Model:
    """ops: list of (K, [Mw, ...], src) with src = None (external activations) or (op index, matrix index).
    mg = -1: per-group scales (+ zero points), act groups of 64 (tbl.cc:323-532); mg >= 1: unified scale(s), one act group
    per row, int32 totals + scale-final (BitNet: tbl.cc:536-630, qgemm.py:170-174)."""

    def __init__(self, tm, ops, bits=2, zp=True, dev_f16=True, seed=0, out_f16=True, mg=-1, ternary=False, ext_f32=False):
        import torch
        self.tm, self.ops, self.bits, self.zp, self.mg = tm, ops, bits, zp and mg < 1, mg
        self.ext_f32 = ext_f32                   # vectors in memory are fp32 (a caller with an fp32 graph); handed-over ones stay fp16
        zp = self.zp
        bm = BITS_BM[bits]
        self.wr = tm.TMACGeMMWrapper(act_group_size=AGS)
        rng = np.random.default_rng(seed)
        self.host, self.ws, self.outs, self.x_ext = [], [], [], {}
        for i, (K, rows, src) in enumerate(ops):
            hs, ws, os_ = [], [], []
            for m, Mw in enumerate(rows):
                ags = K if mg >= 1 else AGS
                case = orc.make_case(1000 * seed + 10 * i + m, Mw, K, bits=bits, gs=GS, ags=ags, zero_point=zp, m_groups=mg, fp16_values=True)
                if mg >= 1 and bits == 2 and ternary:
                    # BitNet's own data: ternary weights {-1, 0, 1} stored as levels {1, 2, 3} of the 2-bit format: zero-mean,
                    # variance 2/3 -- a chain of any depth keeps O(1) activations with scales around 1 / sqrt(2 K / 3)
                    case["w"] = np.random.default_rng(5000 + 1000 * seed + 10 * i + m).integers(1, 4, size=(Mw, K), dtype=np.uint8)
                A = orc.preprocess_weights(case["w"], bits, bm, KF)
                if mg >= 1 and bits == 2 and ternary:
                    S = ((0.8 + 0.4 * np.minimum(case["sc"], 1.0)) / np.sqrt(2.0 * K / 3.0)).astype(np.float16).astype(np.float32)
                elif mg >= 1:
                    # no zero points: the weights' mean level is -1/2 scale, the common component of x grows 0.5 scale K per op;
                    # 0.6 / sqrt(K) keeps a chain a few ops deep inside fp16
                    S = ((0.5 + case["sc"]) * (0.6 / np.sqrt(K))).astype(np.float16).astype(np.float32)
                else:
                    c = 1.0 / np.sqrt(2.5 * K)           # keeps the chained activations O(1)
                    case["sc"] = (case["sc"] * c).astype(np.float16).astype(np.float32)
                    if zp:                               # zero-mean real weights: a common component of x is not amplified op after op
                        lvl = (2 ** bits - 1) / 2.0 - 2 ** (bits - 1)
                        case["zr"] = (case["zr"] * c + lvl * case["sc"]).astype(np.float16).astype(np.float32)
                    else:                                # without zero points the mean level is -1/2 scale: damp the chain instead
                        case["sc"] = (case["sc"] * (4.0 / np.sqrt(K))).astype(np.float16).astype(np.float32)
                    S = orc.preprocess_scales(case["sc"], case["zr"] if zp else None, bits, bm)
                cfg = tm.KCfg.make(Mw, K, bits, bm, KF, GS, ags, zp, mg)
                ws.append(self.wr.register_weights(A, S, Mw, K, bits, cfg, scales_dtype=tm.F32, dev_dtype=tm.F16 if dev_f16 else tm.F32))
                hs.append((A, S))
                os_.append(torch.zeros(Mw, dtype=torch.float16 if out_f16 else torch.float32, device="cuda"))
            self.host.append(hs); self.ws.append(ws); self.outs.append(os_)
            if src is None:
                self.x_ext[i] = torch.from_numpy(rng.standard_normal(K).astype(np.float32)).cuda()
                if not ext_f32:
                    self.x_ext[i] = self.x_ext[i].half()

    def x_of(self, i):
        src = self.ops[i][2]
        return self.x_ext[i] if src is None else self.outs[src[0]][src[1]]

    def act_dtype(self, i):
        return self.tm.F32 if self.ext_f32 and self.ops[i][2] is None else self.tm.F16

    def issue(self):
        for i in range(len(self.ops)):
            self.wr.fused(self.ws[i], self.x_of(i), self.outs[i], 1, act_dtype=self.act_dtype(i))

    def record(self):
        with self.wr.record_chain() as rec:
            self.issue()
        return rec.chain

    def oracle_outputs(self, i, x):
        """the oracle's fp32 outputs of op i on the activation vector x (numpy fp32 [K])"""
        K, rows, _ = self.ops[i]
        xb = x[None, :]
        out = []
        if self.mg >= 1:
            q, ls, lb = orc.preprocessor(xb, K)
            for m, Mw in enumerate(rows):
                A, S = self.host[i][m]
                Cc, _ = orc.qgemm_scale_final(A, q, S, ls[:, 0], lb[:, 0], Mw, K, 1, self.bits, BITS_BM[self.bits], KF, self.mg)
                out.append(Cc[0])
        else:
            q, ls, lb = orc.preprocessor(xb, AGS)
            for m, Mw in enumerate(rows):
                A, S = self.host[i][m]
                out.append(orc.qgemm_float(A, q, S, ls, lb, Mw, K, 1, self.bits, BITS_BM[self.bits], KF, GS, AGS, self.zp)[0])
        return out

    def check(self, chain, oracle_ops=None):
        """after chain.launch(): compare every op with its stand-alone launch and with the oracle"""
        import torch
        tm = self.tm
        L = tm.lib()
        torch.cuda.synchronize()
        assert chain.status() == 0, "a hand-off inside the chain timed out"
        got = [[o.clone() for o in os_] for os_ in self.outs]
        for i, (K, rows, src) in enumerate(self.ops):
            x = self.x_ext[i] if src is None else got[src[0]][src[1]]
            assert x.numel() == K
            assert bool(torch.isfinite(x.float()).all()) and float(x.float().abs().max()) > 0, f"op {i}: degenerate activations"
            # (a) the same call on its own, the chain's threads per workgroup and waves per quad
            # (a quarter-walk stream splits quarters, not 64-unit steps, over its waves: its waves per group need not exist as a stand-alone
            # configuration -- the default launch serves: exact totals for unified scales, a tolerance for per-group scales, below)
            if not getattr(chain, "quarter_walk", False):
                L.tmac_hip_debug_quad_config(chain.threads, chain.wpq(i))
            ref = [torch.empty_like(o) for o in got[i]]
            try:
                self.wr.fused(self.ws[i], x, ref, 1, act_dtype=self.act_dtype(i))
                torch.cuda.synchronize()
            finally:
                L.tmac_hip_debug_quad_config(0, 0)
            # (k_gemv_stream's quarter-walk form adds a row's fp32 partial sums in another order than the stand-alone launch: per-group-scale
            # outputs are then held to the stand-alone launch within 1e-4 here, to the oracle below, and their integers by check_tap)
            qw_float = getattr(chain, "quarter_walk", False) and self.mg < 1
            for m in range(len(rows)):
                a, b = got[i][m].cpu().numpy(), ref[m].cpu().numpy()
                if qw_float:
                    assert rel_err(a.astype(np.float32), b.astype(np.float32)) <= (1e-4 if a.dtype == np.float32 else 2e-3), f"op {i} matrix {m}: quarter-walk stream vs stand-alone launch"
                    continue
                assert np.array_equal(a.view(np.uint16 if a.dtype == np.float16 else np.uint32),
                                      b.view(np.uint16 if b.dtype == np.float16 else np.uint32)), f"op {i} matrix {m}: chain != stand-alone launch"
            # (b) the oracle on the activation vector the chain produced
            if oracle_ops is None or i in oracle_ops:
                want = self.oracle_outputs(i, x.float().cpu().numpy())
                for m in range(len(rows)):
                    g = got[i][m].cpu().numpy()
                    if self.mg >= 1 and g.dtype == np.float16:
                        # the int32 totals are exact and scale-final is three individually rounded fp32 operations: the fp16
                        # outputs are the oracle's fp32 values rounded once -- bit for bit
                        assert np.array_equal(g.view(np.uint16), want[m].astype(np.float16).view(np.uint16)), f"op {i} matrix {m} vs oracle (bits)"
                    assert rel_err(g.astype(np.float32), want[m]) <= 1e-3, f"op {i} matrix {m} vs oracle"

    def check_tap(self, chain):
        """The INTEGERS of every call, written by the persistent launch itself (tmac_hip_chain_set_tap), against the oracle on the activation
        vector the call consumed: per-group scales comb[row][act group] = sum_p 2^p PS_p (PS_p: tbl.cc:445-462 per bit-plane), unified
        scales the exact per-plane totals (tbl.cc:586-628) -- array_equal, no tolerance.  north_star: "bit-exact for the integer
        LUT/accumulate path", checked inside k_decode_chain / k_gemv_stream, not through another kernel."""
        import torch
        nops = len(self.ops)
        total, _ = chain.tap_layout(nops)
        buf = torch.full((total,), -(2 ** 31), dtype=torch.int32, device="cuda")
        chain.set_tap(buf)
        try:
            chain.launch()
            torch.cuda.synchronize()
            assert chain.status() == 0
        finally:
            chain.set_tap(None)
        tap = buf.cpu().numpy()
        got = [[o.clone() for o in os_] for os_ in self.outs]
        bm = BITS_BM[self.bits]
        for i, (K, rows, src) in enumerate(self.ops):
            x = (self.x_ext[i] if src is None else got[src[0]][src[1]]).float().cpu().numpy()
            off, cnt = chain.tap_layout(i)
            ags = K if self.mg >= 1 else AGS
            q, _, _ = orc.preprocessor(x[None, :], ags)
            per_row = self.bits if self.mg >= 1 else K // 64
            assert cnt == sum(rows) * per_row
            r0 = 0
            for m, Mw in enumerate(rows):
                A, _ = self.host[i][m]
                PS = orc.partial_sums(A, q[0], Mw, K, self.bits, bm, KF, ags)           # [M-space rows][K / ags]
                o = np.arange(Mw)
                planes = [PS[(o // 8) * 8 * self.bits + p * 8 + (o % 8)] for p in range(self.bits)]     # weight row o, plane p (weights.py:57-87)
                g = tap[off + r0 * per_row: off + (r0 + Mw) * per_row].reshape(Mw, per_row)
                if self.mg >= 1:
                    want = np.stack([pl[:, 0] for pl in planes], axis=1)
                else:
                    want = sum(pl.astype(np.int64) << p for p, pl in enumerate(planes))
                assert np.array_equal(g.astype(np.int64), want), f"op {i} matrix {m}: integer tap of the persistent kernel != oracle"
                r0 += Mw

    def free(self):
        for ws in self.ws:
            for w in ws:
                w.free()


SMALL = [
    (1024, [512, 256], None),          # q/k-style pair from an external vector
    (512, [1024], (0, 0)),             # one matrix, 8 quads per ... few quads: waves split the steps
    (1024, [2688, 512, 128], (1, 0)),  # three matrices
    (2688, [1024], (2, 0)),            # K = 2688: 84 units = one whole step + a ragged one (the K = 11008 case in small)
    (1024, [64], (3, 0)),              # fewer quads than workgroups
    (256, [256, 256], (0, 1)),         # consumes an older output (not the previous op's)
    (256, [4096], (5, 1)),
    (4096, [1024, 1024], (6, 0)),      # two full steps
    (1024, [512], None),               # an external vector in the middle of the chain
]


@pytest.mark.parametrize("bits,zp,dev_f16", [(2, True, True), (2, False, False), (4, True, True), (2, True, False), (4, False, True)])
def test_small_chain(tm, bits, zp, dev_f16):
    import torch
    m = Model(tm, SMALL, bits=bits, zp=zp, dev_f16=dev_f16, seed=bits + 2 * zp)
    chain = m.record()
    assert chain.nops == len(SMALL)
    for rep in range(3):                      # replays: the generation tag advances, stale granules must not match
        for os_ in m.outs:
            for o in os_:
                o.fill_(float(rep))
        chain.launch()
        m.check(chain, oracle_ops=None if rep == 0 else [])
    m.check_tap(chain)
    chain.free()
    m.free()


EXT32 = [
    (1024, [512, 256], None),          # fp32 vector in memory
    (512, [1024], (0, 0)),             # handed over (fp16 granules)
    (2688, [1024], None),              # ragged last step
    (6400, [4096], None),              # two rounds of pairs (enough rows that the chain's waves per quad exist as a launch of its own)
    (1024, [64, 64], (1, 0)),
    (12800, [4096], None),             # three rounds
]


@pytest.mark.parametrize("bits,zp,mg", [(2, True, -1), (4, False, -1), (2, False, 1), (3, True, -1)])
def test_chain_fp32_activations_from_memory(tm, bits, zp, mg):
    """calls whose activations are fp32 vectors in memory (ggml's graphs are fp32): no fp16 in between -- the LUT is built from the
    same fp32 values the call launched on its own builds it from, outputs bit-identical to it, and within 1e-3 of the oracle"""
    m = Model(tm, EXT32, bits=bits, zp=zp, seed=40 + bits, mg=mg, ext_f32=True)
    chain = m.record()
    for rep in range(2):
        chain.launch()
        m.check(chain, oracle_ops=None if rep == 0 else [])
    chain.free()
    m.free()


@pytest.mark.parametrize("bits,zp", [(1, True), (3, True), (3, False), (1, False)])
def test_small_chain_one_and_three_bit_weights(tm, bits, zp):
    """W1 / W3 (the 1-/3-bit tiles of python/t_mac/ops/qgemm.py:98-116) through the persistent chain"""
    m = Model(tm, SMALL, bits=bits, zp=zp, dev_f16=True, seed=20 + bits + zp)
    chain = m.record()
    for rep in range(2):
        chain.launch()
        m.check(chain, oracle_ops=None if rep == 0 else [])
    chain.free()
    m.free()


# unified scale (BitNet): K = 640 .. 8640 (one to five 64-unit steps, K = 8640 with the ragged last step and two build rounds
# per thread), few and many quads, a fused triple, an older output consumed, an external vector mid-chain
UNIFIED = [
    (3200, [3200, 640, 640], None),
    (3200, [3200], (0, 0)),
    (3200, [8640, 8640], (1, 0)),
    (8640, [3200], (2, 0)),
    (640, [1024], (0, 1)),
    (1024, [64], (4, 0)),
    (3200, [256], None),
]


@pytest.mark.parametrize("bits,mg,dev_f16", [(2, 1, False), (2, 1, True), (1, 1, False), (4, 1, True), (3, 1, False), (2, 4, False)])
def test_unified_scale_chain(tm, bits, mg, dev_f16):
    """the int32 / scale-final path (a5: tbl.cc:536-630, qgemm.py:170-174,192-206) inside the persistent chain: every op
    bit-identical to its stand-alone launch, fp16 outputs bit-identical to the oracle's fp32 results rounded once"""
    m = Model(tm, UNIFIED, bits=bits, zp=False, dev_f16=dev_f16, seed=40 + bits + mg, mg=mg)
    chain = m.record()
    assert chain.nops == len(UNIFIED)
    for rep in range(3):
        for os_ in m.outs:
            for o in os_:
                o.fill_(float(rep))
        chain.launch()
        m.check(chain, oracle_ops=None if rep == 0 else [])
    m.check_tap(chain)          # the exact per-plane totals inside k_decode_chain against the oracle
    chain.free()
    m.free()


def test_bitnet_layer_full_size(tm):
    """BitNet-b1.58-3B's decode calls at full size (python/t_mac/model_utils.py:50-54: 3200 x 3200 q/k/v/o, 8640 x 3200
    gate/up, 3200 x 8640 down; ternary weights in 2 bits, one scale per matrix), one layer and a half chained as bench.py
    --workload bitnet-3b chains them"""
    ops = [(3200, [3200, 3200, 3200], None), (3200, [3200], (0, 0)), (3200, [8640, 8640], (1, 0)), (8640, [3200], (2, 0)),
           (3200, [3200, 3200, 3200], (3, 0)), (3200, [3200], (4, 0))]
    m = Model(tm, ops, bits=2, zp=False, dev_f16=False, seed=77, mg=1, ternary=True)
    chain = m.record()
    chain.launch()
    m.check(chain)
    m.check_tap(chain)
    chain.free()
    m.free()


def _random_chain(seed):
    """a random call sequence: 3-7 calls, 1-3 matrices each, K and row counts on the grid the layouts allow, every call fed by
    an external vector or by a random earlier output (so that K = that output's rows)"""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(3, 8))
    ops = []
    for i in range(n):
        feeders = [(j, m) for j in range(i) for m in range(len(ops[j][1])) if ops[j][1][m] % 128 == 0 and ops[j][1][m] <= 6144]
        if feeders and rng.random() < 0.75:
            src = feeders[int(rng.integers(len(feeders)))]
            K = ops[src[0]][1][src[1]]
        else:
            src, K = None, int(rng.choice([128, 256, 640, 1024, 2688, 3200, 4096, 6144]))
        rows = [int(rng.choice([64, 128, 256, 640, 1024, 2688, 3200, 4224])) for _ in range(int(rng.integers(1, 4)))]
        ops.append((K, rows, src))
    return ops


@pytest.mark.parametrize("seed", range(int(os.environ.get("TMAC_FUZZ_CHAINS", "12"))))   # tools/gpu/r3_fuzz.sh runs a few hundred
def test_random_chains(tm, seed):
    """random call sequences through every flavour of the chain kernel (bits 1-4, per-group scales with / without zero points or
    unified scales, fp16 / fp32 scale storage): the same two bars as the fixed cases"""
    rng = np.random.default_rng(1000 + seed)
    bits = int(rng.integers(1, 5))
    mg = 1 if rng.random() < 0.35 else -1
    zp = bool(rng.integers(0, 2))
    m = Model(tm, _random_chain(seed), bits=bits, zp=zp, dev_f16=bool(rng.integers(0, 2)), seed=60 + seed, mg=mg, ternary=(bits == 2))
    chain = m.record()
    for rep in range(2):
        chain.launch()
        m.check(chain, oracle_ops=None if rep == 0 else [])
    m.check_tap(chain)
    chain.free()
    m.free()


def test_chain_launch_inside_a_hip_graph(tm):
    """the chain's ONE launch captured into a hipGraph (a decode step that also holds the attention kernels would be captured like this)
    and replayed: every replay is a new generation of the hand-off tags, outputs as from the eager launch"""
    import torch
    m = Model(tm, SMALL, bits=2, zp=True, dev_f16=True, seed=91)
    chain = m.record()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        chain.launch()                       # warm-up outside the capture (first use sets the kernel's LDS attribute)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        chain.launch()
    for rep in range(3):
        for os_ in m.outs:
            for o in os_:
                o.fill_(float("nan"))
        g.replay()
        m.check(chain, oracle_ops=[] if rep else None)
    del g
    chain.free()
    m.free()


def test_chain_equals_eager_sequence(tm):
    """the recorded calls issued one by one give the same final buffers as the chain (default launch configuration, so
    only to fp32 summation order: 1e-3 on fp16)"""
    import torch
    m = Model(tm, SMALL, seed=7)
    chain = m.record()
    chain.launch()
    torch.cuda.synchronize()
    assert chain.status() == 0
    got = [[o.clone() for o in os_] for os_ in m.outs]
    m.issue()
    torch.cuda.synchronize()
    for a, b in zip(got, m.outs):
        for x, y in zip(a, b):
            assert rel_err(x.float().cpu().numpy(), y.float().cpu().numpy()) <= 5e-3
    chain.free()
    m.free()


def test_chain_rejections(tm):
    """what the chain does not cover is refused at tmac_hip_chain_end with -1 (the caller keeps launching one by one)"""
    import torch
    L = tm.lib()
    wr = tm.TMACGeMMWrapper(act_group_size=AGS)
    case = orc.make_case(3, 128, 512, bits=2)
    A = orc.preprocess_weights(case["w"], 2, 128, KF)
    S = orc.preprocess_scales(case["sc"], case["zr"], 2, 128)
    w = wr.register_weights(A, S, 128, 512, 2, tm.KCfg.make(128, 512, 2, 128, KF, GS, AGS, True))
    x32 = torch.zeros(512, dtype=torch.float32, device="cuda")
    out = torch.zeros(128, dtype=torch.float16, device="cuda")
    case2 = orc.make_case(4, 128, 128, bits=2)
    w2 = wr.register_weights(orc.preprocess_weights(case2["w"], 2, 128, KF), orc.preprocess_scales(case2["sc"], case2["zr"], 2, 128), 128, 128, 2,
                             tm.KCfg.make(128, 128, 2, 128, KF, GS, AGS, True))
    out2 = torch.zeros(128, dtype=torch.float16, device="cuda")
    with pytest.raises(tm.TMACHipError) as e:
        with wr.record_chain():
            wr.fused([w], x32.half(), [out], 1)
            wr.fused([w2], out, [out2], 1, act_dtype=tm.F32)     # an earlier output read as fp32: handed over as fp16, not representable
    assert e.value.code == -1
    with pytest.raises(tm.TMACHipError):
        with wr.record_chain():
            pass                                   # nothing recorded
    # act groups of 32 (legal in the reference: qgemm.py:402-404) live in the row-block layout, which neither the chain nor the stream
    # kernel reads: refused with -1 -- formally out of the persistent paths' scope -- and served by the per-launch path
    wr32 = tm.TMACGeMMWrapper(act_group_size=32)
    case3 = orc.make_case(5, 128, 512, bits=2, gs=GS, ags=32, zero_point=True, fp16_values=True)
    A3, S3 = orc.preprocess_weights(case3["w"], 2, 128, KF), orc.preprocess_scales(case3["sc"], case3["zr"], 2, 128)
    w3 = wr32.register_weights(A3, S3, 128, 512, 2, tm.KCfg.make(128, 512, 2, 128, KF, GS, 32, True))
    x3 = torch.from_numpy(case3["B"][0].astype(np.float32)).cuda().half()
    with pytest.raises(tm.TMACHipError) as e:
        with wr32.record_chain():
            wr32.fused([w3], x3, [out], 1)
    assert e.value.code == -1
    with pytest.raises(tm.TMACHipError) as e:      # the fused entry point (LUT built inside the GEMV) reads the QUAD layout too
        wr32.fused([w3], x3, [out], 1)
    assert e.value.code == -1
    wr32.set_workspace(512, 1)                     # ... served by the reference's own call structure: preprocessor, then qgemm_lut
    wr32.llama_cpp_init(x3, 128, 512, 1, 2, act_group_size=32, act_dtype=tm.F16)
    wr32.llama_cpp_compute(w3, out, 1, out_dtype=tm.F16)
    torch.cuda.synchronize()
    q3, ls3, lb3 = orc.preprocessor(x3.float().cpu().numpy()[None, :], 32)
    want3 = orc.qgemm_float(A3, q3, S3, ls3, lb3, 128, 512, 1, 2, 128, KF, GS, 32, True)[0]
    assert rel_err(out.float().cpu().numpy(), want3) <= 1e-3
    w3.free()
    # recording state is per thread and was closed by the failures above
    with wr.record_chain() as rec:
        wr.fused([w], x32.half(), [out], 1)
    rec.chain.launch()
    torch.cuda.synchronize()
    assert rec.chain.status() == 0
    rec.chain.free()
    w.free(); w2.free()


def test_chain_hazards_are_refused(tm):
    """Workgroups of the chain are not synchronised with each other: data flow is taken from byte RANGES, and whatever no
    hand-off orders is refused with -1 instead of computed wrongly (overlapping buffers, write-after-read,
    write-after-write between independent calls)."""
    import torch
    wr = tm.TMACGeMMWrapper(act_group_size=AGS)

    def mat(Mw, K, seed):
        case = orc.make_case(seed, Mw, K, bits=2, fp16_values=True)
        c = 1.0 / np.sqrt(2.5 * K)                   # unit gain, zero-mean real weights (as Model): the chained values stay O(1)
        case["sc"] = (case["sc"] * c).astype(np.float16).astype(np.float32)
        case["zr"] = (case["zr"] * c - 0.5 * case["sc"]).astype(np.float16).astype(np.float32)
        A = orc.preprocess_weights(case["w"], 2, 128, KF)
        S = orc.preprocess_scales(case["sc"], case["zr"], 2, 128)
        return wr.register_weights(A, S, Mw, K, 2, tm.KCfg.make(Mw, K, 2, 128, KF, GS, AGS, True))

    a, b = mat(512, 512, 1), mat(512, 512, 2)
    few, tail = mat(128, 512, 3), mat(512, 128, 4)      # 128 rows = 32 quads: most workgroups own no row of `few`
    big = torch.zeros(4096, dtype=torch.float16, device="cuda")
    x = torch.randn(512, device="cuda").half()
    y = torch.randn(512, device="cuda").half()
    o1, o2 = big[0:512], big[1024:1536]

    def refused(calls):
        with pytest.raises(tm.TMACHipError) as e:
            with wr.record_chain():
                calls()
        assert e.value.code == -1, e.value
        return str(e.value)

    def partial_overlap():          # (1) activations that overlap an earlier output without being it
        wr.fused([a], x, [big[0:512]], 1)
        wr.fused([b], big[256:768], [o2], 1)

    def war_unordered():            # (2) the second call overwrites what the first reads from memory and depends on nothing
        wr.fused([a], x, [o1], 1)
        wr.fused([b], y, [x], 1)

    def war_thin_reader():          # (3) ... a hand-off path exists, but the reader does not give every workgroup rows
        wr.fused([few], x, [big[2048:2176]], 1)
        wr.fused([tail], big[2048:2176], [x], 1)

    def waw_unordered():            # (4) two independent calls write overlapping outputs
        wr.fused([a], x, [o1], 1)
        wr.fused([b], y, [big[256:768]], 1)

    assert "overlap" in refused(partial_overlap)
    assert "overwrites" in refused(war_unordered)
    assert "overwrites" in refused(war_thin_reader)
    assert "overlapping outputs" in refused(waw_unordered)
    # the legal forms of the same patterns: a decoder's "next x = last output" with every workgroup owning rows of the reader
    # is covered by test_bench_launches_full_size; a buffer rewritten by a DEPENDENT call:
    with wr.record_chain() as rec:
        wr.fused([a], x, [o1], 1)
        wr.fused([b], o1, [o2], 1)
        wr.fused([a], o2, [o1], 1)          # op 2 depends on op 1 depends on op 0: the rewrite of o1 is ordered
    rec.chain.launch()
    torch.cuda.synchronize()
    assert rec.chain.status() == 0
    r1 = torch.empty(512, dtype=torch.float16, device="cuda"); r2 = torch.empty_like(r1); r3 = torch.empty_like(r1)
    wr.fused([a], x, [r1], 1); wr.fused([b], r1, [r2], 1); wr.fused([a], r2, [r3], 1)
    torch.cuda.synchronize()
    assert rel_err(o1.float().cpu().numpy(), r3.float().cpu().numpy()) <= 5e-3
    assert rel_err(o2.float().cpu().numpy(), r2.float().cpu().numpy()) <= 5e-3
    rec.chain.free()
    for w in (a, b, few, tail):
        w.free()


def test_chain_in_flight_guard(tm):
    """a chain owns one set of hand-off buffers: a launch on a second stream while the first may still run is refused"""
    import torch
    m = Model(tm, [(4096, [4096, 4096, 4096], None), (4096, [4096], (0, 0)), (4096, [11008, 11008], (1, 0)), (11008, [4096], (2, 0))] * 1, seed=5)
    chain = m.record()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    L = tm.lib()
    for _ in range(50):
        tm.binding.check(L.tmac_hip_chain_launch(chain.handle, s1.cuda_stream))
    rc = L.tmac_hip_chain_launch(chain.handle, s2.cuda_stream)          # 50 launches (~3 ms) are still queued on s1
    assert rc == -4 and b"in flight" in L.tmac_hip_last_error()
    s1.synchronize()
    tm.binding.check(L.tmac_hip_chain_launch(chain.handle, s2.cuda_stream))
    s2.synchronize()
    assert chain.status() == 0
    chain.free()
    m.free()


LLAMA = [("qkv", 4096, [4096, 4096, 4096]), ("o", 4096, [4096]), ("gate_up", 4096, [11008, 11008]), ("down", 11008, [4096])]


def test_bench_launches_full_size(tm):
    """The launches bench.py times, at full size: one llama-2-7B layer and a half (q/k/v 3 x 4096 x 4096 and gate/up
    2 x 11008 x 4096 as fused groups, o, down 4096 x 11008, then the next layer's q/k/v and o), W2 g128 zero points, chained as
    bench.py chains them (x1 = q, x2 = o, x3 = gate, next x0 = down) -- every output against the stand-alone launch (bit for
    bit) and the oracle."""
    ops = [(4096, [4096, 4096, 4096], None), (4096, [4096], (0, 0)), (4096, [11008, 11008], (1, 0)), (11008, [4096], (2, 0)),
           (4096, [4096, 4096, 4096], (3, 0)), (4096, [4096], (4, 0))]
    m = Model(tm, ops, seed=11)
    chain = m.record()
    chain.launch()
    m.check(chain)
    m.check_tap(chain)          # one full llama-2-7B layer and a half: k_decode_chain's own integers against the oracle
    # and the same matrices through the default per-launch path (what bench.py --path fused times): against the oracle
    import torch
    x = m.x_ext[0]
    for i, (K, rows, src) in enumerate(ops[:4]):
        xi = x if src is None else m.outs[src[0]][src[1]]
        outs = [torch.empty(Mw, dtype=torch.float16, device="cuda") for Mw in rows]
        m.wr.fused(m.ws[i], xi, outs, 1, act_dtype=tm.F16)
        torch.cuda.synchronize()
        q, ls, lb = orc.preprocessor(xi.float().cpu().numpy()[None, :], AGS)
        for mi, Mw in enumerate(rows):
            A, S = m.host[i][mi]
            Cc = orc.qgemm_float(A, q, S, ls, lb, Mw, K, 1, 2, 128, KF, GS, AGS, True)
            assert rel_err(outs[mi].float().cpu().numpy(), Cc[0]) <= 1e-3
    chain.free()
    m.free()
