"""The persistent decode chain (k_decode_chain, tmac_hip_chain_*): a recorded sequence of fused N = 1 calls executed by one
launch, outputs handed from call to call inside the kernel.

Bar: every call's outputs (a) BIT-IDENTICAL to the same call launched on its own through tmac_hip_qgemm_fused_dev with
768-thread workgroups and the chain's number of waves per row quad, fed the activation vector the chain actually produced,
and (b) within 1e-3 (max-norm, fp16 outputs) of the oracle run on that activation vector (lut_ctor.cc / tbl.cc restated in
oracle/tmac_oracle.c).  (a) ties the chain to the kernel whose integer path is tapped bit for bit in test_gpu_parity.py;
(b) ties it to the reference independently of that kernel.
"""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

BITS_BM = {2: 128, 4: 256}
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
    """ops: list of (K, [Mw, ...], src) with src = None (external activations) or (op index, matrix index)"""

    def __init__(self, tm, ops, bits=2, zp=True, dev_f16=True, seed=0, out_f16=True):
        import torch
        self.tm, self.ops, self.bits, self.zp = tm, ops, bits, zp
        bm = BITS_BM[bits]
        self.wr = tm.TMACGeMMWrapper(act_group_size=AGS)
        rng = np.random.default_rng(seed)
        self.host, self.ws, self.outs, self.x_ext = [], [], [], {}
        for i, (K, rows, src) in enumerate(ops):
            hs, ws, os_ = [], [], []
            for m, Mw in enumerate(rows):
                case = orc.make_case(1000 * seed + 10 * i + m, Mw, K, bits=bits, gs=GS, ags=AGS, zero_point=zp, fp16_values=True)
                c = 1.0 / np.sqrt(2.5 * K)           # keeps the chained activations O(1)
                case["sc"] = (case["sc"] * c).astype(np.float16).astype(np.float32)
                if zp:                               # zero-mean real weights: a common component of x is not amplified op after op
                    lvl = (2 ** bits - 1) / 2.0 - 2 ** (bits - 1)
                    case["zr"] = (case["zr"] * c + lvl * case["sc"]).astype(np.float16).astype(np.float32)
                else:                                # without zero points the mean level is -1/2 scale: damp the chain instead
                    case["sc"] = (case["sc"] * (4.0 / np.sqrt(K))).astype(np.float16).astype(np.float32)
                A = orc.preprocess_weights(case["w"], bits, bm, KF)
                S = orc.preprocess_scales(case["sc"], case["zr"] if zp else None, bits, bm)
                cfg = tm.KCfg.make(Mw, K, bits, bm, KF, GS, AGS, zp)
                ws.append(self.wr.register_weights(A, S, Mw, K, bits, cfg, scales_dtype=tm.F32, dev_dtype=tm.F16 if dev_f16 else tm.F32))
                hs.append((A, S))
                os_.append(torch.zeros(Mw, dtype=torch.float16 if out_f16 else torch.float32, device="cuda"))
            self.host.append(hs); self.ws.append(ws); self.outs.append(os_)
            if src is None:
                self.x_ext[i] = torch.from_numpy(rng.standard_normal(K).astype(np.float32)).cuda().half()

    def x_of(self, i):
        src = self.ops[i][2]
        return self.x_ext[i] if src is None else self.outs[src[0]][src[1]]

    def issue(self):
        for i in range(len(self.ops)):
            self.wr.fused(self.ws[i], self.x_of(i), self.outs[i], 1, act_dtype=self.tm.F16)

    def record(self):
        with self.wr.record_chain() as rec:
            self.issue()
        return rec.chain

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
            # (a) the same call on its own, the chain's threads per workgroup and waves per quad
            L.tmac_hip_debug_quad_config(chain.threads, chain.wpq(i))
            ref = [torch.empty_like(o) for o in got[i]]
            try:
                self.wr.fused(self.ws[i], x, ref, 1, act_dtype=tm.F16)
                torch.cuda.synchronize()
            finally:
                L.tmac_hip_debug_quad_config(0, 0)
            for m in range(len(rows)):
                a, b = got[i][m].cpu().numpy(), ref[m].cpu().numpy()
                assert np.array_equal(a.view(np.uint16 if a.dtype == np.float16 else np.uint32),
                                      b.view(np.uint16 if b.dtype == np.float16 else np.uint32)), f"op {i} matrix {m}: chain != stand-alone launch"
            # (b) the oracle on the activation vector the chain produced
            if oracle_ops is None or i in oracle_ops:
                xb = x.float().cpu().numpy()[None, :]
                q, ls, lb = orc.preprocessor(xb, AGS)
                for m, Mw in enumerate(rows):
                    A, S = self.host[i][m]
                    Cc = orc.qgemm_float(A, q, S, ls, lb, Mw, K, 1, self.bits, BITS_BM[self.bits], KF, GS, AGS, self.zp)
                    assert rel_err(got[i][m].float().cpu().numpy(), Cc[0]) <= 1e-3, f"op {i} matrix {m} vs oracle"

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
    with pytest.raises(tm.TMACHipError) as e:
        with wr.record_chain():
            wr.fused([w], x32, [out], 1)          # fp32 activations
    assert e.value.code == -1
    with pytest.raises(tm.TMACHipError):
        with wr.record_chain():
            pass                                   # nothing recorded
    # recording state is per thread and was closed by the failures above
    with wr.record_chain() as rec:
        wr.fused([w], x32.half(), [out], 1)
    rec.chain.launch()
    torch.cuda.synchronize()
    assert rec.chain.status() == 0
    rec.chain.free()
    w.free()


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
