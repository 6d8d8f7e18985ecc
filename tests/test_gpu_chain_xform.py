"""Vector transforms inside the persistent decode chain (tmac_hip_chain_xform, include/tmac_hip.h): the element-wise operators that sit
between the mpGEMMs of a decoder layer -- residual add + RMSNorm in front of q/k/v and gate/up, silu(gate) * up in front of the down
projection -- applied inside the consumer's LUT build, so that a layer's calls chain up in ONE launch per segment between two operators
that stay outside (attention).

These are extensions without a reference counterpart (T-MAC has no norm operator; its call contract is one llama_cpp_init +
llama_cpp_compute per mat-mul, include/t-mac/tmac_gemm_wrapper.h:170-228).  Bar: every call's outputs within 2e-3 of max |C| of the
ORACLE (lut_ctor.cc / tbl.cc restated in oracle/tmac_oracle.c) run on the transformed vector computed with the same formulas in
numpy fp32 from the inputs the chain actually saw -- tolerance, not bits: the mean square is summed in another order and exp differs in
the last bit, which can move a LUT entry by one step; the residual stream (fp32 adds only) is compared bit for bit.
"""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
BITS, BM, KF, GS, AGS = 2, 128, 16, 128, 64


@pytest.fixture(scope="module")
def tm():
    import torch
    import tmac_amd
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return tmac_amd


@pytest.fixture(autouse=True)
def _short_spin(tm):
    tm.binding.check(tm.lib().tmac_hip_debug_chain_config(0, 1 << 17))


def rel_err(c, ref):
    return float(np.abs(c.astype(np.float64) - ref.astype(np.float64)).max() / max(np.abs(ref).max(), 1e-30))


class Mat:
    """one W2 matrix (zero points, group 128) with scales sized so that its outputs are O(1) for O(1) inputs"""

    def __init__(self, tm, wr, seed, Mw, K):
        case = orc.make_case(seed, Mw, K, bits=BITS, gs=GS, ags=AGS, zero_point=True, fp16_values=True)
        c = 1.0 / np.sqrt(2.5 * K)
        case["sc"] = (case["sc"] * c).astype(np.float16).astype(np.float32)
        lvl = (2 ** BITS - 1) / 2.0 - 2 ** (BITS - 1)
        case["zr"] = (case["zr"] * c + lvl * case["sc"]).astype(np.float16).astype(np.float32)
        self.A = orc.preprocess_weights(case["w"], BITS, BM, KF)
        self.S = orc.preprocess_scales(case["sc"], case["zr"], BITS, BM)
        self.Mw, self.K = Mw, K
        cfg = tm.KCfg.make(Mw, K, BITS, BM, KF, GS, AGS, True, -1)
        self.w = wr.register_weights(self.A, self.S, Mw, K, BITS, cfg, scales_dtype=tm.F32, dev_dtype=tm.F16)

    def oracle(self, x):
        """fp32 outputs of the oracle on the fp32 activation vector x"""
        q, ls, lb = orc.preprocessor(x[None, :].astype(np.float32), AGS)
        return orc.qgemm_float(self.A, q, self.S, ls, lb, self.Mw, self.K, 1, BITS, BM, KF, GS, AGS, True)[0]


def np_norm(t, gamma, eps):
    t = t.astype(np.float32)
    rs = np.float32(1.0) / np.sqrt(np.float32((t.astype(np.float64) ** 2).mean()) + np.float32(eps))
    return (t * rs).astype(np.float32) * gamma.astype(np.float32)


def np_glu(a, b):
    a = a.astype(np.float32); b = b.astype(np.float32)
    return (a / (np.float32(1.0) + np.exp(-a))).astype(np.float32) * b


@pytest.mark.parametrize("grid", [0, 96, 7, 1])
def test_norm_and_glu_on_external_vectors(tm, grid):
    """grid != 0: fewer workgroups than CUs (tmac_hip_debug_chain_grid: a partitioned device, a device with fewer CUs) -- residual_out is
    striped over the workgroups that exist (a fixed stripe of 256 left pairs unwritten, silently)"""
    import torch
    tm.binding.check(tm.lib().tmac_hip_debug_chain_grid(grid))
    try:
        _norm_and_glu_on_external_vectors(tm)
    finally:
        tm.binding.check(tm.lib().tmac_hip_debug_chain_grid(0))


def _norm_and_glu_on_external_vectors(tm):
    import torch
    wr = tm.TMACGeMMWrapper(act_group_size=AGS)
    rng = np.random.default_rng(11)
    K, Mw = 1024, 512
    m0, m1, m2 = Mat(tm, wr, 1, Mw, K), Mat(tm, wr, 2, Mw, K), Mat(tm, wr, 3, Mw, K)
    x = torch.from_numpy(rng.standard_normal(K).astype(np.float32)).cuda().half()
    x2 = torch.from_numpy(rng.standard_normal(K).astype(np.float32)).cuda().half()
    res = torch.from_numpy(rng.standard_normal(K).astype(np.float32)).cuda()
    gam = torch.from_numpy((1.0 + 0.1 * rng.standard_normal(K)).astype(np.float32)).cuda()
    rout = torch.zeros(K, dtype=torch.float32, device="cuda")
    o0, o1, o2 = (torch.zeros(Mw, dtype=torch.float16, device="cuda") for _ in range(3))
    with wr.record_chain() as rec:
        wr.chain_xform("norm", residual=res, gamma=gam, eps=1e-5, residual_out=rout)
        wr.fused([m0.w], x, [o0], 1, act_dtype=tm.F16)
        wr.chain_xform("glu", in2=x2)
        wr.fused([m1.w], x, [o1], 1, act_dtype=tm.F16)
        wr.chain_xform("norm", residual=res)                       # add only
        wr.fused([m2.w], x, [o2], 1, act_dtype=tm.F16)
    chain = rec.chain
    for rep in range(2):
        chain.launch()
        torch.cuda.synchronize()
        assert chain.status() == 0
        xf, x2f, rf, gf = x.float().cpu().numpy(), x2.float().cpu().numpy(), res.cpu().numpy(), gam.cpu().numpy()
        t = xf + rf
        assert np.array_equal(rout.cpu().numpy(), t)
        assert rel_err(o0.float().cpu().numpy(), m0.oracle(np_norm(t, gf, 1e-5))) <= 2e-3
        assert rel_err(o1.float().cpu().numpy(), m1.oracle(np_glu(xf, x2f))) <= 2e-3
        assert rel_err(o2.float().cpu().numpy(), m2.oracle(t)) <= 2e-3
    chain.free()


class Layer:
    def __init__(self, tm, wr, seed, H, F):
        self.q, self.k, self.v = (Mat(tm, wr, seed + i, H, H) for i in range(3))
        self.o = Mat(tm, wr, seed + 3, H, H)
        self.gate, self.up = Mat(tm, wr, seed + 4, F, H), Mat(tm, wr, seed + 5, F, H)
        self.down = Mat(tm, wr, seed + 6, H, F)
        rng = np.random.default_rng(seed)
        import torch
        self.g1 = torch.from_numpy((1.0 + 0.1 * rng.standard_normal(H)).astype(np.float32)).cuda()
        self.g2 = torch.from_numpy((1.0 + 0.1 * rng.standard_normal(H)).astype(np.float32)).cuda()


@pytest.mark.parametrize("H,F,glu_in_producer", [(1024, 2816, 1), (1024, 2816, 0), (4096, 11008, 1), (1024, 2560, 1)])
def test_decoder_layers_with_an_operator_outside(tm, H, F, glu_in_producer, monkeypatch):
    """A llama-shaped layer loop: per layer ONE launch of the segment o -> [+ residual, RMSNorm] -> gate / up -> [silu(gate) * up] ->
    down -> [+ residual, RMSNorm] -> next layer's q / k / v, then an operator that stays outside the chain (a stand-in for attention:
    any kernel on the stream) produces the next segment's input from q, k, v.  Every mpGEMM against the oracle, the residual stream bit
    for bit against fp32 adds.  silu(gate) * up is computed by the gate / up call's publishing wave (default: once per row, the down
    projection reads ONE image) or inside the down projection's LUT build (TMAC_CHAIN_GLU_EPILOGUE=0: the form that also serves vectors
    in memory): both are run.  F = 2560: 640 row pairs, not a multiple of the workgroup count (ragged ranges)."""
    import torch
    monkeypatch.setenv("TMAC_CHAIN_GLU_EPILOGUE", str(glu_in_producer))
    NL, eps = 3, 1e-5
    wr = tm.TMACGeMMWrapper(act_group_size=AGS)
    layers = [Layer(tm, wr, 100 * (li + 1), H, F) for li in range(NL)]
    rng = np.random.default_rng(5)
    h = torch.from_numpy(rng.standard_normal(H).astype(np.float32)).cuda()          # residual stream, fp32
    f16 = lambda n: torch.zeros(n, dtype=torch.float16, device="cuda")
    attn = f16(H)                                                                     # the outside operator's output
    bufs = [dict(o=f16(H), gate=f16(F), up=f16(F), down=f16(H), q=f16(H), k=f16(H), v=f16(H), h_out=torch.zeros(H, dtype=torch.float32, device="cuda"))
            for _ in range(NL)]
    # first q / k / v of the model: its own one-call chain (RMSNorm of the embedding inside)
    q0, k0, v0 = f16(H), f16(H), f16(H)
    with wr.record_chain() as rec0:
        wr.chain_xform("norm", gamma=layers[0].g1, eps=eps)
        wr.fused([layers[0].q.w, layers[0].k.w, layers[0].v.w], h.half(), [q0, k0, v0], 1, act_dtype=tm.F16)
    hx = h.half()      # (kept alive: the chain reads it)
    chains = []
    for li in range(NL - 1):
        L, Ln, b = layers[li], layers[li + 1], bufs[li]
        hin = h if li == 0 else bufs[li - 1]["h_out"]
        with wr.record_chain() as rec:
            wr.fused([L.o.w], attn, [b["o"]], 1, act_dtype=tm.F16)
            wr.chain_xform("norm", residual=hin, gamma=L.g2, eps=eps, keep=True)
            wr.fused([L.gate.w, L.up.w], b["o"], [b["gate"], b["up"]], 1, act_dtype=tm.F16)
            wr.chain_xform("glu", in2=b["up"])
            wr.fused([L.down.w], b["gate"], [b["down"]], 1, act_dtype=tm.F16)
            wr.chain_xform("norm", residual=wr.CARRY, gamma=Ln.g1, eps=eps, residual_out=b["h_out"])
            wr.fused([Ln.q.w, Ln.k.w, Ln.v.w], b["down"], [b["q"], b["k"], b["v"]], 1, act_dtype=tm.F16)
        chains.append(rec.chain)

    def outside(q, k, v):          # stand-in for attention: some kernel of the stream between two segments
        attn.copy_((torch.tanh(q.float()) * 0.5 + 0.25 * k.float() - 0.25 * v.float()).half())

    rec0.chain.launch()
    torch.cuda.synchronize()
    assert rec0.chain.status() == 0
    hn = h.cpu().numpy()
    x1 = np_norm(hx.float().cpu().numpy(), layers[0].g1.cpu().numpy(), eps)
    for m, got in ((layers[0].q, q0), (layers[0].k, k0), (layers[0].v, v0)):
        assert rel_err(got.float().cpu().numpy(), m.oracle(x1)) <= 2e-3
    q, k, v = q0, k0, v0
    for li in range(NL - 1):
        L, Ln, b = layers[li], layers[li + 1], bufs[li]
        outside(q, k, v)
        chains[li].launch()
        torch.cuda.synchronize()
        assert chains[li].status() == 0, f"layer {li}: a hand-off timed out"
        a = attn.float().cpu().numpy()
        o = b["o"].float().cpu().numpy()
        assert rel_err(o, L.o.oracle(a)) <= 2e-3
        t2 = o + hn                                                  # fp32 add: exact agreement expected downstream
        x2 = np_norm(t2, L.g2.cpu().numpy(), eps)
        g, u = b["gate"].float().cpu().numpy(), b["up"].float().cpu().numpy()
        assert rel_err(g, L.gate.oracle(x2)) <= 2e-3 and rel_err(u, L.up.oracle(x2)) <= 2e-3
        d = b["down"].float().cpu().numpy()
        m_ = np_glu(g, u)
        if glu_in_producer:
            m_ = m_.astype(np.float16).astype(np.float32)           # the hand-off image holds silu(gate) * up as fp16
        assert rel_err(d, L.down.oracle(m_)) <= 2e-3
        t3 = d + t2
        assert np.array_equal(b["h_out"].cpu().numpy(), t3), f"layer {li}: residual stream"
        x3 = np_norm(t3, Ln.g1.cpu().numpy(), eps)
        for m, got in ((Ln.q, b["q"]), (Ln.k, b["k"]), (Ln.v, b["v"])):
            assert rel_err(got.float().cpu().numpy(), m.oracle(x3)) <= 2e-3
        hn = t3
        q, k, v = b["q"], b["k"], b["v"]
    assert np.isfinite(hn).all() and np.abs(hn).max() < 1e3
    for c in chains:
        c.free()
    rec0.chain.free()


def test_transform_errors(tm):
    import torch
    wr = tm.TMACGeMMWrapper(act_group_size=AGS)
    m = Mat(tm, wr, 9, 256, 512)
    x = torch.zeros(512, dtype=torch.float16, device="cuda")
    o = torch.zeros(256, dtype=torch.float16, device="cuda")
    # a carry nobody kept
    with pytest.raises(tm.binding.TMACHipError):
        with wr.record_chain():
            wr.chain_xform("norm", residual=wr.CARRY)
            wr.fused([m.w], x, [o], 1, act_dtype=tm.F16)
    # outside a recording
    with pytest.raises(tm.binding.TMACHipError):
        wr.chain_xform("norm")
    # residual_out over the vector the same NORM reads (the workgroups that own an index range store it while the others still read)
    res = torch.zeros(512, dtype=torch.float32, device="cuda")
    with pytest.raises(tm.binding.TMACHipError):
        with wr.record_chain():
            wr.chain_xform("norm", residual=res, residual_out=res)
            wr.fused([m.w], x, [o], 1, act_dtype=tm.F16)
    # residual_out over the call's own activations, and over an output of the launch
    xf32 = torch.zeros(512, dtype=torch.float32, device="cuda")
    with pytest.raises(tm.binding.TMACHipError):
        with wr.record_chain():
            wr.chain_xform("norm", residual=res, residual_out=xf32)
            wr.fused([m.w], xf32[:256].view(torch.float16), [o], 1, act_dtype=tm.F16)
    o32 = torch.zeros(512, dtype=torch.float32, device="cuda")
    with pytest.raises(tm.binding.TMACHipError):
        with wr.record_chain():
            wr.chain_xform("norm", residual=res, residual_out=o32)
            wr.fused([m.w], x, [o32[:128].view(torch.float16)], 1, act_dtype=tm.F16)
    # an output of the launch over the norm weights of a later call that no hand-off orders behind it
    gam = torch.ones(512, dtype=torch.float32, device="cuda")
    with pytest.raises(tm.binding.TMACHipError):
        with wr.record_chain():
            wr.fused([m.w], x, [gam[:128].view(torch.float16)], 1, act_dtype=tm.F16)
            wr.chain_xform("norm", residual=res, gamma=gam)
            wr.fused([m.w], x, [o], 1, act_dtype=tm.F16)
