"""k_gemm_planes (tmac_gemm2.hip): qgemm_lut for N > 1 with the bit-planes combined inside the matrix-core operand.

Bars: (a) the LUT image it streams (k_lut_image) bit-identical to the oracle's QLUT / lut_scales / lut_biases
(lut_ctor.cc restated in oracle/tmac_oracle.c); (b) the integers it feeds into the fp32 chain, comb = sum_p 2^p PS_p,
bit-identical to the oracle's per-plane partial sums (tbl.cc:445-462) combined as integers; (c) outputs within 1e-3 of
max|C| of the oracle (fp32 chain regrouped: K split over 8 / 4 waves, zero-point term once per weight group) -- measured
~1e-6; (d) the same outputs as k_gemm_onehot (the per-plane kernel, whose integer path is tapped plane by plane in
test_gpu_parity.py).  Every case runs through both workgroup forms of the kernel (tmac_gemm2.hip, PForm: eight waves and one
workgroup per CU, four waves and two per CU; tmac_hip_debug_gemm_kernel 2 / 3 -- the default picks by the number of tiles).
"""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tm():
    import torch
    import tmac_amd
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    assert tmac_amd.lib().tmac_hip_device_count() > 0
    return tmac_amd


def rel_err(c, ref):
    return float(np.abs(c.astype(np.float64) - ref.astype(np.float64)).max() / max(np.abs(ref).max(), 1e-30))


def mrow(o, p, bits):
    return (o // 8) * 8 * bits + p * 8 + (o % 8)


def run(tm, case, Mw, K, bits, bm, gs, zp, N, scale_f16=False, act_f16=False, out_f16=False, kernel=0, want_comb=True):
    import torch
    L = tm.lib()
    tm.binding.check(L.tmac_hip_set_gemm_min_n(1))
    tm.binding.check(L.tmac_hip_debug_gemm_kernel(kernel))
    try:
        A = orc.preprocess_weights(case["w"], bits, bm, 16)
        S = orc.preprocess_scales(case["sc"], case["zr"] if zp else None, bits, bm)
        cfg = tm.KCfg.make(Mw, K, bits, bm, 16, gs, 64, zp, -1, N)
        wr = tm.TMACGeMMWrapper(act_group_size=64)
        wr.set_workspace(K, N)
        w = wr.register_weights(A, S, Mw, K, bits, cfg, scales_dtype=tm.F32, dev_dtype=tm.F16 if scale_f16 else tm.F32)
        Bt = torch.from_numpy(case["B"]).cuda()
        if act_f16:
            Bt = Bt.half()
        Ct = torch.full((N, Mw), float("nan"), dtype=torch.float16 if out_f16 else torch.float32, device="cuda")
        wr.llama_cpp_init(Bt, Mw, K, N, bits)
        wr.llama_cpp_compute(w, Ct, N)
        torch.cuda.synchronize()
        out = dict(C=Ct.float().cpu().numpy(), A=A, S=S)
        if kernel != 1:
            out["img"] = wr.workspace.read_gemm_image(K, N)
            if want_comb:
                out["comb"] = wr.comb_sums(w, N)
        w.free()
        return out
    finally:
        L.tmac_hip_debug_gemm_kernel(0)
        L.tmac_hip_set_gemm_min_n(32)


CASES = [
    # Mw,   K,    bits, bm,  gs,  zp,    N
    (128, 1024, 2, 128, 128, True, 64),
    (128, 1024, 2, 128, 128, True, 2),       # two activation rows in a 64-row tile
    (320, 3200, 2, 320, 128, True, 33),      # 25 weight groups over 8 waves: ragged K ranges; rows / tokens ragged
    (512, 2048, 4, 256, 128, True, 16),
    (256, 1024, 4, 256, 64, False, 130),     # group size 64 (one act group per weight group), three token blocks
    (704, 1024, 2, 128, 128, False, 8),      # 11 row blocks: the last XCD round is partly empty
    (64, 512, 2, 128, 256, True, 70),        # group size 256: 2 weight groups, six of the eight waves idle
    (1088, 11008, 2, 128, 128, True, 40),    # K = 11008: 86 weight groups
    (192, 4096, 4, 256, 128, True, 256),
    # 1- and 3-bit weights: a byte of the layout mixes tables, 16-byte operand-row entries (tmac_gemm2.hip, ODD)
    (128, 1024, 1, 64, 128, True, 64),
    (256, 1024, 1, 64, 64, False, 130),
    (192, 2048, 3, 192, 128, True, 40),
    (320, 3200, 3, 192, 128, False, 33),     # ragged K ranges, rows and tokens
    (1088, 11008, 3, 192, 128, True, 16),
    (64, 4096, 1, 64, 256, True, 256),
    # more than four token blocks: the chunk-major LUT image's n-tile stride (round 6), ragged last block
    (128, 1024, 2, 128, 128, True, 300),
    (192, 2048, 4, 256, 128, True, 449),
]


FORMS = [2, 3]     # tmac_hip_debug_gemm_kernel: k_gemm_planes with eight- / four-wave workgroups


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("Mw,K,bits,bm,gs,zp,N", CASES)
def test_gemm_planes(tm, Mw, K, bits, bm, gs, zp, N, form):
    case = orc.make_case(9000 + N + K, Mw, K, N=N, bits=bits, gs=gs, ags=64, zero_point=zp)
    r = run(tm, case, Mw, K, bits, bm, gs, zp, N, kernel=form)
    q, ls, lb = orc.preprocessor(case["B"], 64)
    # (a) the LUT image: entries 0..7 of every table, scales, biases bit for bit; the entry sums are what they say
    h, gls, glb, hs = r["img"]
    assert np.array_equal(h, q[:, :, :8])
    assert np.array_equal(gls.view(np.uint32), ls.view(np.uint32)) and np.array_equal(glb.view(np.uint32), lb.view(np.uint32))
    assert np.array_equal(hs, q[:, :, :8].astype(np.int32).reshape(N, K // 64, 128).sum(-1).astype(np.float32))
    # (b) comb = sum_p 2^p PS_p
    PS = np.stack([orc.partial_sums(r["A"], q[n], Mw, K, bits, bm, 16, 64) for n in range(N)])       # [N][M][G]
    rows = np.arange(Mw)
    comb = sum((PS[:, mrow(rows, p, bits), :].astype(np.int64) << p) for p in range(bits))
    assert np.array_equal(r["comb"].astype(np.int64), comb)
    # (c) outputs
    Cc = orc.qgemm_float(r["A"], q, r["S"], ls, lb, Mw, K, N, bits, bm, 16, gs, 64, zp)
    assert rel_err(r["C"], Cc) <= 1e-5
    # (d) the per-plane kernel on the same inputs
    r1 = run(tm, case, Mw, K, bits, bm, gs, zp, N, kernel=1)
    assert rel_err(r["C"], r1["C"]) <= 1e-5


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("scale_f16,act_f16,out_f16", [(True, True, True), (True, False, False), (False, True, True)])
def test_gemm_planes_dtypes(tm, scale_f16, act_f16, out_f16, form):
    Mw, K, bits, bm, gs, N = 256, 2048, 2, 128, 128, 96
    case = orc.make_case(77, Mw, K, N=N, bits=bits, gs=gs, ags=64, fp16_values=True)
    r = run(tm, case, Mw, K, bits, bm, gs, True, N, scale_f16=scale_f16, act_f16=act_f16, out_f16=out_f16, want_comb=False, kernel=form)
    q, ls, lb = orc.preprocessor(case["B"], 64)
    Cc = orc.qgemm_float(r["A"], q, r["S"], ls, lb, Mw, K, N, bits, bm, 16, gs, 64, True)
    assert rel_err(r["C"], Cc) <= (1e-3 if out_f16 else 1e-5)


@pytest.mark.parametrize("form", FORMS)
def test_gemm_planes_edge_activations(tm, form):
    """all-zero act groups (scale 0), huge and tiny magnitudes, exact .5 ties: the LUT image equals the oracle's bit for
    bit and the outputs stay finite where the oracle's are"""
    Mw, K, bits, bm, gs, N = 128, 1024, 2, 128, 128, 12
    case = orc.make_case(5, Mw, K, N=N, bits=bits, gs=gs, ags=64)
    B = case["B"]
    B[0, :64] = 0.0
    B[1, :] = 0.0
    B[2, 128:192] *= 1e20
    B[3, 192:256] *= 1e-20
    B[4, 256:320] = np.tile(np.array([0.5, 1.5, 2.5, 127.0], np.float32), 16)
    r = run(tm, case, Mw, K, bits, bm, gs, True, N, kernel=form)
    q, ls, lb = orc.preprocessor(case["B"], 64)
    h, gls, glb, hs = r["img"]
    assert np.array_equal(h, q[:, :, :8])
    assert np.array_equal(gls.view(np.uint32), ls.view(np.uint32)) and np.array_equal(glb.view(np.uint32), lb.view(np.uint32))
    Cc = orc.qgemm_float(r["A"], q, r["S"], ls, lb, Mw, K, N, bits, bm, 16, gs, 64, True)
    finite = np.isfinite(Cc)
    assert np.array_equal(np.isfinite(r["C"]), finite)
    for n in range(N):
        m = finite[n]
        if m.any():
            assert rel_err(r["C"][n][m], Cc[n][m]) <= 1e-5


US_CASES = [
    # Mw,   K,    N,  bits
    (320, 3200, 40, 2), (160, 640, 33, 2), (3200, 8640, 70, 2), (3200, 3200, 256, 2), (64, 12288, 13, 2),
    # round 4: the other widths (operand rows of k_gemm_planes; 3- / 4-bit rows leave their +7 / +15 with the row's entry sum)
    (320, 3200, 40, 4), (192, 12288, 13, 4), (256, 1024, 130, 4),
    (320, 3200, 33, 3), (192, 12288, 16, 3),
    (128, 1024, 64, 1), (320, 8640, 70, 1),
    (128, 3200, 300, 2),     # five token blocks, direct B loads of the row-wise image
]


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("Mw,K,N,bits", US_CASES)
def test_gemm_planes_unified_scale(tm, Mw, K, N, bits, form):
    """the BitNet flavour (m_groups = 1, one act group per row) through k_gemm_planes_us, 1- to 4-bit weights: the LUT image of
    the row-wise build bit-identical to the oracle's, the combined integer totals bit-identical to the oracle's per-plane totals,
    and the outputs BIT-IDENTICAL to the oracle's scale-final expression (qgemm.py:170-174) -- integer accumulation has no order"""
    import torch
    bm = {1: 64, 2: 320 if Mw % 160 == 0 else 128, 3: 192, 4: 256}[bits]
    if Mw % (bm // bits) != 0:
        bm = 32 * bits
    case = orc.make_case(400 + N + K + bits, Mw, K, N=N, bits=bits, ags=K, zero_point=False, m_groups=1)
    L = tm.lib()
    tm.binding.check(L.tmac_hip_set_gemm_min_n(1))
    tm.binding.check(L.tmac_hip_debug_gemm_kernel(form))
    try:
        A = orc.preprocess_weights(case["w"], bits, bm, 16)
        S = case["sc"]
        cfg = tm.KCfg.make(Mw, K, bits, bm, 16, 128, K, False, 1, N)
        wr = tm.TMACGeMMWrapper(act_group_size=K)
        wr.set_workspace(K, N)
        w = wr.register_weights(A, S, Mw, K, bits, cfg, scales_dtype=tm.F32, dev_dtype=tm.F32)
        Bt = torch.from_numpy(case["B"]).cuda()
        Ct = torch.full((N, Mw), float("nan"), dtype=torch.float32, device="cuda")
        wr.llama_cpp_init(Bt, Mw, K, N, bits)
        wr.llama_cpp_compute(w, Ct, N)
        torch.cuda.synchronize()
        h, gls, glb, hs = wr.workspace.read_gemm_image(K, N, act_group_size=K)
        comb = wr.comb_sums(w, N)
        C = Ct.cpu().numpy()
        # the fused entry point (one image build + one launch) gives the same bits
        C2 = torch.full((N, Mw), float("nan"), dtype=torch.float32, device="cuda")
        wr.fused([w], Bt, [C2], N)
        torch.cuda.synchronize()
        C2 = C2.cpu().numpy()
        w.free()
    finally:
        L.tmac_hip_debug_gemm_kernel(0)
        L.tmac_hip_set_gemm_min_n(32)
    q, ls, lb = orc.preprocessor(case["B"], K)
    assert np.array_equal(h, q[:, :, :8])
    assert np.array_equal(gls.view(np.uint32), ls.view(np.uint32)) and np.array_equal(glb.view(np.uint32), lb.view(np.uint32))
    assert np.array_equal(hs[:, 0], q[:, :, :8].astype(np.int32).reshape(N, -1).sum(-1).astype(np.float32))
    Cc, cb = orc.qgemm_scale_final(A, q, S, ls[:, 0], lb[:, 0], Mw, K, N, bits, bm, 16, 1)      # cb: int32 [N][M] per-plane totals
    rows = np.arange(Mw)
    want = sum((cb[:, mrow(rows, p, bits)].astype(np.int64) << p) for p in range(bits))
    assert np.array_equal(comb[:, :, 0].astype(np.int64), want)
    assert np.array_equal(C.view(np.uint32), Cc.view(np.uint32))
    assert np.array_equal(C2.view(np.uint32), Cc.view(np.uint32))


@pytest.mark.parametrize("bits,bm", [(1, 64), (3, 192), (2, 128)])
def test_fused_entry_point_takes_the_planes_kernel(tm, bits, bm):
    """tmac_hip_qgemm_fused_dev with N > 1 and two matrices that share the activations (gate / up): one LUT image build and ONE
    k_gemm_planes launch for both -- also for 1- and 3-bit weights, which ran the GEMV kernel once per activation row before
    round 3 -- outputs against the oracle and against the split entry points"""
    import torch
    K, gs, N, Mws = 2048, 128, 130, (192, 320)
    L = tm.lib()
    cases = [orc.make_case(700 + bits + i, Mw, K, N=N, bits=bits, gs=gs, ags=64, zero_point=True) for i, Mw in enumerate(Mws)]
    B = cases[0]["B"]
    wr = tm.TMACGeMMWrapper(act_group_size=64)
    wr.set_workspace(K, N)
    ws, refs = [], []
    q, ls, lb = orc.preprocessor(B, 64)
    for c, Mw in zip(cases, Mws):
        A = orc.preprocess_weights(c["w"], bits, bm, 16)
        S = orc.preprocess_scales(c["sc"], c["zr"], bits, bm)
        ws.append(wr.register_weights(A, S, Mw, K, bits, tm.KCfg.make(Mw, K, bits, bm, 16, gs, 64, True, -1, N), scales_dtype=tm.F32, dev_dtype=tm.F32))
        refs.append(orc.qgemm_float(A, q, S, ls, lb, Mw, K, N, bits, bm, 16, gs, 64, True))
    Bt = torch.from_numpy(B).cuda()
    outs = [torch.full((N, Mw), float("nan"), dtype=torch.float32, device="cuda") for Mw in Mws]
    wr.fused(ws, Bt, outs, N)
    torch.cuda.synchronize()
    for o, ref in zip(outs, refs):
        assert rel_err(o.cpu().numpy(), ref) <= 1e-5
    # the split entry points (preprocessor + qgemm per matrix) take the same kernel: bit-identical outputs
    wr.llama_cpp_init(Bt, Mws[0], K, N, bits)
    for w, o, Mw in zip(ws, outs, Mws):
        C2 = torch.full((N, Mw), float("nan"), dtype=torch.float32, device="cuda")
        wr.llama_cpp_compute(w, C2, N)
        torch.cuda.synchronize()
        assert torch.equal(C2, o)
    for w in ws:
        w.free()


def test_workspace_write_invalidates_the_lut_image(tm):
    """tmac_hip_preprocessor_dev(N = 16, X1) builds k_gemm_planes' LUT image; tmac_hip_workspace_write then replaces the LUT with
    that of X2 (the split C-ABI, tmac_gemm_wrapper.h:170-228 with a caller-built LUT).  The following qgemm must compute with X2's
    LUT -- not stream the image of X1 (round-3 advisor finding)."""
    import torch
    Mw, K, bits, bm, gs, N = 256, 1024, 2, 128, 128, 16
    c1 = orc.make_case(71, Mw, K, bits=bits, N=N, gs=gs, ags=64, zero_point=True)
    c2 = orc.make_case(72, Mw, K, bits=bits, N=N, gs=gs, ags=64, zero_point=True)
    A = orc.preprocess_weights(c1["w"], bits, bm, 16)
    S = orc.preprocess_scales(c1["sc"], c1["zr"], bits, bm)
    cfg = tm.KCfg.make(Mw, K, bits, bm, 16, gs, 64, True, -1, N)
    wr = tm.TMACGeMMWrapper(act_group_size=64)
    wr.set_workspace(K, N)
    w = wr.register_weights(A, S, Mw, K, bits, cfg, scales_dtype=tm.F32, dev_dtype=tm.F32)
    Ct = torch.full((N, Mw), float("nan"), dtype=torch.float32, device="cuda")
    wr.llama_cpp_init(torch.from_numpy(c1["B"]).cuda(), Mw, K, N, bits)       # image of X1
    q2, ls2, lb2 = orc.preprocessor(c2["B"], 64)
    wr.workspace.write(q2, ls2, lb2, 64)                                       # LUT of X2
    wr.llama_cpp_compute(w, Ct, N)
    torch.cuda.synchronize()
    want = orc.qgemm_float(A, q2, S, ls2, lb2, Mw, K, N, bits, bm, 16, gs, 64, True)
    assert rel_err(Ct.cpu().numpy(), want) <= 1e-3
    w.free()


@pytest.mark.parametrize("N", [5, 8, 11, 12, 20])
def test_split_entry_small_n_matches_oracle(tm, N):
    """the split entry points (preprocessor_dev + qgemm_dev) around the GEMM thresholds: below PLANES_MIN_N the LUT image is not
    built and the row loop runs (k_gemm_onehot only from its own crossover on), from it on k_gemm_planes -- same results"""
    import torch
    Mw, K, bits, bm, gs = 512, 1024, 2, 128, 128
    c = orc.make_case(80 + N, Mw, K, bits=bits, N=N, gs=gs, ags=64, zero_point=True)
    A = orc.preprocess_weights(c["w"], bits, bm, 16)
    S = orc.preprocess_scales(c["sc"], c["zr"], bits, bm)
    cfg = tm.KCfg.make(Mw, K, bits, bm, 16, gs, 64, True, -1, N)
    wr = tm.TMACGeMMWrapper(act_group_size=64)
    wr.set_workspace(K, N)
    w = wr.register_weights(A, S, Mw, K, bits, cfg, scales_dtype=tm.F32, dev_dtype=tm.F32)
    Ct = torch.full((N, Mw), float("nan"), dtype=torch.float32, device="cuda")
    wr.llama_cpp_init(torch.from_numpy(c["B"]).cuda(), Mw, K, N, bits)
    wr.llama_cpp_compute(w, Ct, N)
    torch.cuda.synchronize()
    q, ls, lb = orc.preprocessor(c["B"], 64)
    want = orc.qgemm_float(A, q, S, ls, lb, Mw, K, N, bits, bm, 16, gs, 64, True)
    assert rel_err(Ct.cpu().numpy(), want) <= 1e-3
    w.free()
