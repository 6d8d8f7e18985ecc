"""Parity of the HIP path against the oracle / golden vectors, through the C-ABI (libtmac_hip.so).

Bar (BASELINE.json north_star): QLUT, lut_scales, lut_biases and the integer partial sums bit-exact;
fp32 outputs within 1e-3 relative (max-abs-diff / max-abs-ref) — measured values are ~1e-6 because
only the fp32 summation ORDER differs — and fp16 outputs equal to the oracle's fp32 rounded once, to
within one fp16 ulp-class tolerance (1e-3).  The generic reference-layout kernel (variant 3) keeps the
reference's exact float order and must match the oracle BIT FOR BIT in fp32.
"""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REL_TOL = 1e-3


@pytest.fixture(scope="module")
def tm():
    import torch
    import tmac_amd
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    L = tmac_amd.lib()
    assert L.tmac_hip_device_count() > 0
    return tmac_amd


def rel_err(c, ref):
    return float(np.abs(c.astype(np.float64) - ref.astype(np.float64)).max() / max(np.abs(ref).max(), 1e-30))


def run_case(tm, case, Mw, K, bits, bm, kf, gs, ags, zp, m_groups=-1, N=1, variant=0, scale_dtype=None, out_f16=False,
             act_f16=False, want_ps=True, gemm_min_n=None, fast_aggregation=0):
    """register + preprocess + gemv on the GPU; returns dict(q, ls, lb, C, PS)"""
    import torch
    L = tm.lib()
    tm.binding.check(L.tmac_hip_set_variant(variant))
    if gemm_min_n is not None:
        tm.binding.check(L.tmac_hip_set_gemm_min_n(gemm_min_n))
    A = orc.preprocess_weights(case["w"], bits, bm, kf)
    if m_groups == -1:
        S = orc.preprocess_scales(case["sc"], case["zr"] if zp else None, bits, bm)
    else:
        S = case["sc"]
    cfg = tm.KCfg.make(Mw, K, bits, bm, kf, gs, ags, zp, m_groups, N)
    wr = tm.TMACGeMMWrapper(act_group_size=ags)
    wr.set_workspace(K, N)
    dev_dt = tm.F16 if scale_dtype == "f16" else tm.F32
    w = wr.register_weights(A, S, Mw, K, bits, cfg, scales_dtype=tm.F32, dev_dtype=dev_dt, fast_aggregation=fast_aggregation)
    Bt = torch.from_numpy(case["B"]).cuda()
    if act_f16:
        Bt = Bt.half()
    Ct = torch.empty((N, Mw), dtype=torch.float16 if out_f16 else torch.float32, device="cuda")
    wr.llama_cpp_init(Bt, Mw, K, N, bits)
    wr.llama_cpp_compute(w, Ct, N)
    torch.cuda.synchronize()
    q, ls, lb = wr.workspace.read(K, N, ags)
    out = dict(q=q, ls=ls, lb=lb, C=Ct.float().cpu().numpy(), A=A, S=S)
    if want_ps:
        out["PS"] = wr.partial_sums(w, N)
    w.free()
    L.tmac_hip_set_variant(0)
    L.tmac_hip_set_gemm_min_n(32)
    return out


def oracle_case(case, A, S, Mw, K, bits, bm, kf, gs, ags, zp, m_groups=-1, N=1):
    q, ls, lb = orc.preprocessor(case["B"], ags)
    if m_groups == -1:
        Cc = orc.qgemm_float(A, q, S, ls, lb, Mw, K, N, bits, bm, kf, gs, ags, zp)
        PS = np.stack([orc.partial_sums(A, q[n], Mw, K, bits, bm, kf, ags) for n in range(N)])
    elif ags == K:
        Cc, cb = orc.qgemm_scale_final(A, q, S, ls[:, 0], lb[:, 0], Mw, K, N, bits, bm, kf, m_groups)
        PS = cb[:, :, None]
    else:
        Cc = orc.qgemm_float(A, q, S, ls, lb, Mw, K, N, bits, bm, kf, gs, ags, False, one_scale=True)
        PS = np.stack([orc.partial_sums(A, q[n], Mw, K, bits, bm, kf, ags) for n in range(N)])
    return q, ls, lb, Cc, PS


def check_bits(a, b):
    assert np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


# -------------------------------------------------------------------------------------------------

def test_isa_models(tm):
    """v_perm_b32 / v_mqsad_pk_u16_u8 on the hardware == the host models the CPU emulation test relies on"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(0)
    n = 4096
    x = rng.integers(0, 2 ** 32, size=(n, 4), dtype=np.uint64).astype(np.uint32)
    x[: n // 2, 2] &= 0x0f0f0f0f          # half of the selectors in the 0..15 range, half arbitrary bytes
    out = np.zeros_like(x)
    tm.binding.check(tm.lib().tmac_hip_selftest(x.ctypes.data, out.ctypes.data, n))

    def perm(s0, s1, sel):
        src = (np.uint64(s0) << np.uint64(32)) | np.uint64(s1)
        r = 0
        for i in range(4):
            c = (int(sel) >> (8 * i)) & 0xff
            if c <= 7: b = (int(src) >> (8 * c)) & 0xff
            elif c == 8: b = 0xff if (int(s1) >> 15) & 1 else 0
            elif c == 9: b = 0xff if (int(s1) >> 31) & 1 else 0
            elif c == 10: b = 0xff if (int(s0) >> 15) & 1 else 0
            elif c == 11: b = 0xff if (int(s0) >> 31) & 1 else 0
            elif c == 12: b = 0
            else: b = 0xff
            r |= b << (8 * i)
        return r

    for i in range(n):
        a, b, c, d = [int(v) for v in x[i]]
        assert int(out[i, 0]) == perm(a, b, c), (i, hex(a), hex(b), hex(c), hex(int(out[i, 0])))
        acc_lo, acc_hi = b & 0x0fff0fff, d & 0x0fff0fff
        acc = [(acc_lo & 0xffff), acc_lo >> 16, acc_hi & 0xffff, acc_hi >> 16]
        exp = [acc[k] + (255 - ((a >> (8 * k)) & 0xff)) for k in range(4)]
        got = [int(out[i, 1]) & 0xffff, int(out[i, 1]) >> 16, int(out[i, 2]) & 0xffff, int(out[i, 2]) >> 16]
        assert got == exp, (i, hex(a), acc, got, exp)


CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*.npz")))


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 7])
@pytest.mark.parametrize("name", CASES)
def test_golden_vectors(tm, name, variant):
    """committed vectors produced by the reference itself (tests/golden/make_golden.py)"""
    d = dict(np.load(os.path.join(GOLD, name + ".npz")))
    Mw, K, bits, bm, kf, gs, ags, zp, mg = [int(x) for x in d["meta"]]
    case = dict(w=d["w"], sc=d["sc"], zr=d.get("zr"), B=d["B"])
    r = run_case(tm, case, Mw, K, bits, bm, kf, gs, ags, bool(zp), mg, variant=variant)
    assert np.array_equal(r["A"], d["A_ref"])
    assert np.array_equal(r["q"][0], d["qlut"])
    check_bits(r["ls"][0], d["lut_scales"])
    check_bits(r["lb"][0], d["lut_biases"])
    if mg == -1:
        assert np.array_equal(r["PS"][0], d["PS"])
        if variant == 3:
            check_bits(r["C"][0], d["C"])          # reference float order -> bit-exact
        assert rel_err(r["C"][0], d["C"]) <= REL_TOL
    else:
        assert np.array_equal(r["PS"][0, :, 0], d["cbits32"])


CFGS = [  # Mw, K, bits, bm, kf, gs, ags, zp, m_groups
    (4096, 4096, 2, 128, 16, 128, 64, True, -1),     # BASELINE config #1 (tests/test_e2e.py)
    (512, 11008, 2, 128, 16, 128, 64, True, -1),     # headline K (ragged last segment block)
    (704, 4096, 2, 128, 16, 128, 64, False, -1),     # 11 tiles, no zero points
    (256, 4096, 4, 256, 16, 128, 64, True, -1),      # W4 GPTQ-style (config #3)
    (256, 1024, 1, 128, 16, 128, 64, True, -1),
    (256, 1024, 3, 192, 16, 128, 64, False, -1),
    (256, 1024, 2, 128, 8, 128, 32, True, -1),       # act_group 32
    (256, 1024, 4, 256, 16, 64, 64, False, -1),      # group_size 64
    (320, 3200, 2, 320, 16, 128, 3200, False, 1),    # BitNet x86: unified scale, int32 aggregation (config #4)
    (128, 1024, 1, 128, 16, 128, 1024, False, 1),    # ... with 1-bit weights: one MFMA chain per step, its result read right
    (256, 576, 1, 128, 8, 64, 576, False, 1),        #     behind the loop branch (a missing wait state was found here)
    (128, 2048, 3, 192, 16, 128, 2048, False, 1), (128, 1024, 4, 256, 16, 128, 1024, False, 1),
    (320, 8640, 2, 128, 16, 128, 8640, False, 1),
    (320, 3200, 2, 128, 16, 128, 64, False, 1),      # BitNet ARM flavour: one weight scale, per-group LUT scales
]


@pytest.mark.parametrize("variant", [0, 1, 2, 4, 5, 7])
@pytest.mark.parametrize("Mw,K,bits,bm,kf,gs,ags,zp,mg", CFGS)
def test_vs_oracle(tm, Mw, K, bits, bm, kf, gs, ags, zp, mg, variant):
    case = orc.make_case(Mw + K + bits, Mw, K, bits=bits, gs=gs, ags=ags, zero_point=zp, m_groups=mg)
    r = run_case(tm, case, Mw, K, bits, bm, kf, gs, ags, zp, mg, variant=variant)
    q, ls, lb, Cc, PS = oracle_case(case, r["A"], r["S"], Mw, K, bits, bm, kf, gs, ags, zp, mg)
    assert np.array_equal(r["q"], q)
    check_bits(r["ls"], ls)
    check_bits(r["lb"], lb)
    assert np.array_equal(r["PS"], PS)
    assert rel_err(r["C"], Cc) <= REL_TOL
    assert rel_err(r["C"], Cc) <= 2e-5  # what fp32 re-association actually costs


@pytest.mark.parametrize("Mw,K,bits,bm,kf,gs,ags,zp,mg", CFGS[:4] + CFGS[8:])
def test_reference_layout_kernel_is_bit_exact(tm, Mw, K, bits, bm, kf, gs, ags, zp, mg):
    Mw = min(Mw, 640) // (bm // bits) * (bm // bits) or bm // bits
    case = orc.make_case(K + bits, Mw, K, bits=bits, gs=gs, ags=ags, zero_point=zp, m_groups=mg)
    r = run_case(tm, case, Mw, K, bits, bm, kf, gs, ags, zp, mg, variant=3)
    q, ls, lb, Cc, PS = oracle_case(case, r["A"], r["S"], Mw, K, bits, bm, kf, gs, ags, zp, mg)
    assert np.array_equal(r["PS"], PS)
    check_bits(r["C"], Cc)


def _random_configs(n, seed):
    """valid (Mw, K, bits, bm, kf, gs, ags, zp, m_groups, N) tuples drawn from the reference's knob space
    (qgemm.py:98-129: bm % 32 == 0, (bm / bits) % 8 == 0, 4 kfactor % ags == 0, gs % (4 kfactor) == 0, K % gs == 0)"""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        bits = int(rng.integers(1, 5))
        bm = int(rng.choice([192, 384] if bits == 3 else [128, 256, 512, 320, 640]))
        if bm % bits or (bm // bits) % 8:
            continue
        kf = int(rng.choice([8, 16]))
        flavour = int(rng.integers(0, 4))         # 0, 1: per-group scales (zp on/off); 2: unified scale; 3: one scale + per-group LUT
        gs = int(rng.choice([64, 128, 256]))
        if gs % (4 * kf):
            continue
        K = gs * int(rng.integers(2, 12))
        if (K // 4) % kf:
            continue
        ags = int(rng.choice([32, 64]))
        if (4 * kf) % ags:
            continue
        mg = -1
        zp = flavour == 0
        if flavour == 2:
            ags, mg, zp = K, 1, False
        elif flavour == 3:
            mg, zp = 1, False
        Mw = (bm // bits) * int(rng.integers(1, 5))
        N = int(rng.choice([1, 1, 1, 2, 3, 37]))
        out.append((Mw, K, bits, bm, kf, gs, ags, zp, mg, N))
    return out


@pytest.mark.parametrize("cfg", _random_configs(48, 20260924), ids=lambda c: "-".join(str(int(x)) for x in c))
def test_random_configurations(tm, cfg):
    """48 configurations drawn from the reference's whole knob space (1-4 bits, every legal bm / kfactor, group sizes,
    act groups 32 / 64 / K, zero points, unified and single scales, 1-37 activation rows), whatever kernel the dispatcher
    picks for them: QLUT, LUT scales / biases and integer sums bit-exact, outputs within the fp32 bound"""
    Mw, K, bits, bm, kf, gs, ags, zp, mg, N = cfg
    case = orc.make_case(sum(cfg), Mw, K, N=N, bits=bits, gs=gs, ags=ags, zero_point=zp, m_groups=mg)
    variant = int(os.environ.get("TMAC_FUZZ_VARIANT", "0"))     # tools/gpu/fuzz.sh sweeps the A/B variants too
    fa = int(os.environ.get("TMAC_FUZZ_FA", "0"))               # ... and the two fast-aggregation flavours
    if fa:
        if mg != -1:
            pytest.skip("fast aggregation is defined for per-group scales only")
        r = run_case(tm, case, Mw, K, bits, bm, kf, gs, ags, zp, mg, N=N, variant=variant, fast_aggregation=fa)
        Cc, tap = orc.qgemm_float_fa(r["A"], r["q"], r["S"], r["ls"], r["lb"], Mw, K, N, bits, bm, kf, gs, ags, zp, fa)
        assert np.array_equal(np.asarray(r["PS"]).reshape(tap.shape), tap)
        assert np.abs(r["C"] - Cc).max() <= (1e-3 if fa == 1 else 1e-4) * np.abs(Cc).max()
        return
    try:
        r = run_case(tm, case, Mw, K, bits, bm, kf, gs, ags, zp, mg, N=N, variant=variant)
    except tm.binding.TMACHipError as e:
        if variant and e.code == -1:
            tm.lib().tmac_hip_set_variant(0)
            pytest.skip("this variant has no kernel for the configuration")
        raise
    q, ls, lb, Cc, PS = oracle_case(case, r["A"], r["S"], Mw, K, bits, bm, kf, gs, ags, zp, mg, N=N)
    assert np.array_equal(r["q"], q)
    check_bits(r["ls"], ls); check_bits(r["lb"], lb)
    assert np.array_equal(np.asarray(r["PS"]).reshape(PS.shape), PS)
    assert rel_err(r["C"], Cc) <= 2e-5


def _random_fused(n, seed):
    rng = np.random.default_rng(seed)
    big = int(os.environ.get("TMAC_FUZZ_BIG", "0"))      # tools/gpu/fuzz.sh: matrices large enough for every launch configuration
    out = []
    while len(out) < n:
        bits = int(rng.integers(1, 5))
        bm = int(rng.choice([192, 384] if bits == 3 else [128, 256, 512, 320, 640]))
        if bm % bits or (bm // bits) % 8:
            continue
        gs = int(rng.choice([64, 128, 256]))
        K = gs * int(rng.integers(2, 96 if big else 40))
        if K > 24576:
            continue
        unified = bool(rng.integers(0, 4) == 0)
        zp = (not unified) and bool(rng.integers(0, 2))
        nmat = int(rng.integers(1, 4))
        Mws = [(bm // bits) * int(rng.integers(1, 160 if big else 6)) for _ in range(nmat)]
        N = int(rng.choice([1, 1, 1, 1, 2] if big else [1, 1, 2, 40, 70]))
        out.append((tuple(Mws), K, bits, bm, gs, zp, unified, N, bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))))
    return out


@pytest.mark.parametrize("cfg", _random_fused(40, 99), ids=lambda c: "-".join(str(x) for x in c).replace(" ", ""))
def test_random_fused_calls(tm, cfg):
    """40 random calls of the fused entry point (1-3 matrices sharing the activations, 1-4 bits, per-group or unified scale,
    fp16 / fp32 activations, scales and outputs, 1-70 activation rows: decode kernel, row loop or pair-wise LUT build +
    batched GEMM, whatever the dispatcher picks) against the oracle"""
    import torch
    Mws, K, bits, bm, gs, zp, unified, N, act_f16, sc_f16, out_f16 = cfg
    ags, mg, kf = (K, 1, 16) if unified else (64, -1, 16)
    wr = tm.TMACGeMMWrapper(act_group_size=ags)
    B = np.random.default_rng(sum(Mws) + K).standard_normal((N, K)).astype(np.float32)
    if act_f16:
        B = B.astype(np.float16).astype(np.float32)
    ws, cases, outs = [], [], []
    for i, Mw in enumerate(Mws):
        case = orc.make_case(1000 * i + K + bits, Mw, K, N=N, bits=bits, gs=gs, ags=ags, zero_point=zp, m_groups=mg, fp16_values=sc_f16)
        case["B"] = B
        A = orc.preprocess_weights(case["w"], bits, bm, kf)
        S = orc.preprocess_scales(case["sc"], case["zr"] if zp else None, bits, bm) if mg == -1 else case["sc"]
        cfgk = tm.KCfg.make(Mw, K, bits, bm, kf, gs, ags, zp, mg, N)
        ws.append(wr.register_weights(A, S, Mw, K, bits, cfgk, scales_dtype=tm.F32,
                                      dev_dtype=tm.F16 if (sc_f16 and mg == -1) else tm.F32))
        cases.append((case, A, S))
        outs.append(torch.empty((N, Mw), dtype=torch.float16 if out_f16 else torch.float32, device="cuda"))
    Bt = torch.from_numpy(B).cuda()
    if act_f16:
        Bt = Bt.half()
    wr.fused(ws, Bt, outs, N)
    torch.cuda.synchronize()
    for (case, A, S), o, Mw in zip(cases, outs, Mws):
        _, _, _, Cc, _ = oracle_case(case, A, S, Mw, K, bits, bm, kf, gs, ags, zp, mg, N=N)
        got = o.float().cpu().numpy()
        assert rel_err(got, Cc) <= (REL_TOL if out_f16 else 2e-5)
    tm.lib().tmac_hip_cache_clear()
    for w in ws:
        w.free()


def test_fp16_storage_path(tm):
    """fp16 activations, fp16 scales on the device, fp16 output — exact w.r.t. the oracle fed the same
    fp16-representable values, up to the final rounding to fp16."""
    Mw, K, bits, bm, kf, gs, ags = 512, 4096, 2, 128, 16, 128, 64
    case = orc.make_case(42, Mw, K, bits=bits, fp16_values=True)
    r = run_case(tm, case, Mw, K, bits, bm, kf, gs, ags, True, scale_dtype="f16", out_f16=True, act_f16=True)
    q, ls, lb, Cc, PS = oracle_case(case, r["A"], r["S"], Mw, K, bits, bm, kf, gs, ags, True)
    assert np.array_equal(r["q"], q) and np.array_equal(r["PS"], PS)
    check_bits(r["ls"], ls)
    assert rel_err(r["C"], Cc.astype(np.float16).astype(np.float32)) <= REL_TOL


def test_multi_row_activations(tm):
    """N > 1 (small prefill): each activation row is an independent GEMV (qgemm.py:183-190)"""
    Mw, K, bits, bm, kf, gs, ags, N = 256, 1024, 2, 128, 16, 128, 64, 5
    case = orc.make_case(9, Mw, K, N=N, bits=bits)
    r = run_case(tm, case, Mw, K, bits, bm, kf, gs, ags, True, N=N)
    q, ls, lb, Cc, PS = oracle_case(case, r["A"], r["S"], Mw, K, bits, bm, kf, gs, ags, True, N=N)
    assert np.array_equal(r["q"], q) and np.array_equal(r["PS"], PS)
    assert rel_err(r["C"], Cc) <= 2e-5


FA_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "fa", "*.npz")))


@pytest.mark.parametrize("variant", [0, 3])
@pytest.mark.parametrize("name", FA_CASES)
def test_fast_aggregation_golden(tm, name, variant):
    """(a9) the reference's own FastAggregation build (AVX2 flavour, tests/golden/fa/) through the GPU: mode 2"""
    d = dict(np.load(os.path.join(GOLD, name + ".npz")))
    g = dict(np.load(os.path.join(GOLD, "fa", name + ".npz")))
    Mw, K, bits, bm, kf, gs, ags, zp, mg = [int(x) for x in d["meta"]]
    case = dict(w=d["w"], sc=d["sc"], zr=d.get("zr"), B=d["B"])
    r = run_case(tm, case, Mw, K, bits, bm, kf, gs, ags, bool(zp), mg, variant=variant, fast_aggregation=2)
    assert np.array_equal(r["q"][0], d["qlut"])
    _, tap = orc.qgemm_float_fa(r["A"], r["q"], r["S"], r["ls"], r["lb"], Mw, K, 1, bits, bm, kf, gs, ags, bool(zp), 2)
    assert np.array_equal(r["PS"], tap)          # the halving-tree results, bit for bit
    if variant == 3:
        check_bits(r["C"][0], g["C_fa"])         # generic kernel: the reference's float order
    else:
        assert np.abs(r["C"][0] - g["C_fa"]).max() <= 1e-4 * np.abs(g["C_fa"]).max()


@pytest.mark.parametrize("variant", [0, 3])
@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("Mw,K,bits,bm,kf,gs,ags,zp,N", [
    (256, 4096, 2, 128, 16, 128, 64, True, 1), (128, 11008, 2, 128, 16, 128, 64, True, 2),
    (256, 4096, 4, 256, 16, 128, 64, True, 1), (128, 2048, 3, 192, 16, 128, 64, False, 1),
    (128, 2048, 1, 128, 16, 128, 64, True, 3), (128, 2048, 2, 128, 8, 128, 32, True, 1),
    (64, 1024, 2, 128, 8, 64, 32, False, 1),
])
def test_fast_aggregation_vs_oracle(tm, Mw, K, bits, bm, kf, gs, ags, zp, N, mode, variant):
    """(a9) both flavours of the halving-adder aggregation against the restatement: tree results bit-exact, outputs
    within the fp tolerance; the signed flavour also stays close to the exact path (it is the usable one)"""
    case = orc.make_case(1000 + Mw + K + bits, Mw, K, N=N, bits=bits, gs=gs, ags=ags, zero_point=zp)
    r = run_case(tm, case, Mw, K, bits, bm, kf, gs, ags, zp, N=N, variant=variant, fast_aggregation=mode)
    Cc, tap = orc.qgemm_float_fa(r["A"], r["q"], r["S"], r["ls"], r["lb"], Mw, K, N, bits, bm, kf, gs, ags, zp, mode)
    assert np.array_equal(r["PS"], tap)
    if variant == 3:
        check_bits(r["C"], Cc)
    elif mode == 1:
        assert rel_err(r["C"], Cc) < REL_TOL
        if ags == 64:
            exact = orc.qgemm_float(r["A"], r["q"], r["S"], r["ls"], r["lb"], Mw, K, N, bits, bm, kf, gs, ags, zp)
            assert np.mean((r["C"] - exact) ** 2) / np.mean(exact ** 2) < 5e-3
    else:
        assert np.abs(r["C"] - Cc).max() <= 1e-4 * np.abs(Cc).max()


def test_autotuner(tm, tmp_path):
    """(N4) tmac_hip_autotune_fused: measures the launch configurations on the q/k/v-like set, may record one, and never
    changes results; the table survives a save / clear / load cycle"""
    import torch
    L = tm.lib()
    L.tmac_hip_tune_clear()
    Mw, K, bits, bm = 1024, 4096, 2, 128
    cfg = tm.KCfg.make(Mw, K, bits, bm, 16, 128, 64, True, -1, 1)
    wr = tm.TMACGeMMWrapper(act_group_size=64)
    ws, cases = [], []
    for i in range(3):
        case = orc.make_case(40 + i, Mw, K, bits=bits, fp16_values=True)
        A = orc.preprocess_weights(case["w"], bits, bm, 16)
        S = orc.preprocess_scales(case["sc"], case["zr"], bits, bm)
        ws.append(wr.register_weights(A, S, Mw, K, bits, cfg, scales_dtype=tm.F32, dev_dtype=tm.F16))
        cases.append(case)
    Bt = torch.from_numpy(cases[0]["B"]).cuda().half()
    Cs = [torch.empty((1, Mw), dtype=torch.float16, device="cuda") for _ in range(3)]
    wr.fused(ws, Bt, Cs, 1)
    torch.cuda.synchronize()
    before = [c.clone() for c in Cs]
    r = wr.autotune(ws, tm.F16, tm.F16)
    assert r["heuristic_us"] > 0 and r["us"] > 0 and r["us"] <= r["heuristic_us"] * 1.0001
    assert (r["ft"], r["wpq"]) in [(0, 0), (512, 1), (512, 2), (768, 1), (768, 2), (768, 3), (1024, 1), (1024, 2), (1024, 4)]
    # force an entry so that the table lookup path is exercised whatever the measurement said
    p = tmp_path / "t.txt"
    p.write_text("2 4096 %d 3 13 1024 2 1.0\n" % (3 * Mw // 4))
    assert wr.tune_load(str(p)) == 1
    for c in Cs:
        c.zero_()
    wr.fused(ws, Bt, Cs, 1)
    torch.cuda.synchronize()
    for a, b in zip(before, Cs):          # same arithmetic; only the fp32 order across a row's waves may differ
        assert rel_err(b.float().cpu().numpy(), a.float().cpu().numpy()) < REL_TOL
    n = wr.tune_save(str(tmp_path / "o.txt"))
    assert n >= 1
    L.tmac_hip_tune_clear()
    assert wr.tune_load(str(tmp_path / "o.txt")) == n
    L.tmac_hip_tune_clear()
    for w in ws:
        w.free()


def test_fast_aggregation_rejections(tm):
    """what the reference does not define stays undefined: no fast aggregation on the unified-scale / int32 path, and
    fast-aggregation weights do not run through the fused entry point"""
    import torch
    L = tm.lib()
    case = orc.make_case(5, 320, 640, bits=2, m_groups=1, ags=640)
    A = orc.preprocess_weights(case["w"], 2, 320, 16)
    cfg = tm.KCfg.make(320, 640, 2, 320, 16, 0, 640, False, 1, 1)
    wr = tm.TMACGeMMWrapper(act_group_size=640)
    with pytest.raises(Exception):
        wr.register_weights(A, case["sc"], 320, 640, 2, cfg, fast_aggregation=1)
    case = orc.make_case(6, 128, 1024, bits=2)
    A = orc.preprocess_weights(case["w"], 2, 128, 16)
    S = orc.preprocess_scales(case["sc"], case["zr"], 2, 128)
    cfg = tm.KCfg.make(128, 1024, 2, 128, 16, 128, 64, True, -1, 1)
    wr = tm.TMACGeMMWrapper(act_group_size=64)
    wr.set_workspace(1024, 1)
    w = wr.register_weights(A, S, 128, 1024, 2, cfg, fast_aggregation=1)
    Bt = torch.from_numpy(case["B"]).cuda()
    Ct = torch.empty((1, 128), dtype=torch.float32, device="cuda")
    with pytest.raises(Exception):
        wr.fused([w], Bt, [Ct], 1)
    w.free()
    # and the mode does not leak into later registrations
    w2 = wr.register_weights(A, S, 128, 1024, 2, cfg)
    wr.fused([w2], Bt, [Ct], 1)
    torch.cuda.synchronize()
    w2.free()


@pytest.mark.parametrize("Mw,K,bits,bm,kf,gs,ags,zp,N", [
    (256, 1024, 2, 128, 16, 128, 64, True, 5),      # ragged n tile
    (256, 1024, 2, 128, 16, 128, 64, False, 32),
    (320, 3200, 2, 320, 16, 128, 64, True, 33),     # K/32 = 100 units, ragged everything
    (512, 2048, 4, 256, 16, 128, 64, True, 16),
    (256, 1024, 4, 256, 16, 64, 64, False, 130),    # two workgroups along n
    (704, 1024, 2, 128, 16, 128, 64, True, 8),      # Mw not a multiple of the 32-row workgroup tile
])
def test_onehot_mfma_gemm(tm, Mw, K, bits, bm, kf, gs, ags, zp, N):
    """N > 1 through k_gemm_onehot (one-hot(nibble) x QLUT on v_mfma_i32_16x16x64_i8): integer sums bit-exact,
    outputs equal to the GEMV loop and within tolerance of the oracle"""
    case = orc.make_case(77 + N, Mw, K, N=N, bits=bits, gs=gs, ags=ags)
    r = run_case(tm, case, Mw, K, bits, bm, kf, gs, ags, zp, N=N, gemm_min_n=1)
    q, ls, lb, Cc, PS = oracle_case(case, r["A"], r["S"], Mw, K, bits, bm, kf, gs, ags, zp, N=N)
    assert np.array_equal(r["q"], q) and np.array_equal(r["PS"], PS)
    assert rel_err(r["C"], Cc) <= 2e-5
    r2 = run_case(tm, case, Mw, K, bits, bm, kf, gs, ags, zp, N=N, gemm_min_n=0, want_ps=False)
    assert rel_err(r["C"], r2["C"]) <= 2e-6


@pytest.mark.parametrize("Mw,K,bits,bm,N", [(320, 3200, 2, 320, 5), (320, 3200, 2, 320, 40), (160, 640, 2, 320, 33),
                                            (128, 1024, 4, 256, 20), (3200, 8640, 2, 128, 34)])
def test_onehot_mfma_gemm_unified_scale(tm, Mw, K, bits, bm, N):
    """the BitNet flavour (m_groups = 1, act group = K) through k_gemm_onehot: int32 totals over the whole K bit-exact,
    outputs bit-identical to the GEMV loop's (same scale-final expression on the same integers) and to the oracle"""
    case = orc.make_case(300 + N + K, Mw, K, N=N, bits=bits, ags=K, zero_point=False, m_groups=1)
    r = run_case(tm, case, Mw, K, bits, bm, 16, 128, K, False, m_groups=1, N=N, gemm_min_n=1)
    q, ls, lb, Cc, PS = oracle_case(case, r["A"], r["S"], Mw, K, bits, bm, 16, 128, K, False, m_groups=1, N=N)
    assert np.array_equal(r["q"], q) and np.array_equal(r["PS"], PS)
    check_bits(r["ls"], ls); check_bits(r["lb"], lb)      # the row-wise pair build (k_preprocess_pairs_row) from N = 2 on
    check_bits(r["C"], Cc)
    r2 = run_case(tm, case, Mw, K, bits, bm, 16, 128, K, False, m_groups=1, N=N, gemm_min_n=0, want_ps=False)
    check_bits(r["C"], r2["C"])


@pytest.mark.parametrize("K,N,act_f16,edge", [(2048, 48, False, False), (11008, 40, True, False), (4096, 33, False, True),
                                              (4096, 64, True, True)])
def test_fused_entry_point_prefill(tm, K, N, act_f16, edge):
    """tmac_hip_qgemm_fused_dev with N above the GEMM threshold: library-owned LUT workspace, the pair-wise LUT build
    (k_preprocess_pairs: image only) and one one-hot GEMM per matrix.  A wrong table entry moves an output by ~1e-3 of
    max|C|, far above the 2e-5 bound, so the outputs pin the LUT as well."""
    import torch
    Mw, bits, bm, kf, gs, ags = 512, 2, 128, 16, 128, 64
    case = orc.make_case(4242 + K, Mw, K, N=N, bits=bits, gs=gs, ags=ags, fp16_values=act_f16)
    if edge:   # all-zero act groups (scale 0 -> t_scales 0), huge / tiny magnitudes, exact .5 ties, as test_edge_activations
        B = case["B"]
        B[0, :64] = 0.0
        B[1, 64:128] = 0.0
        B[2, :] = 0.0
        B[3, 128:192] *= (60000.0 / np.abs(B[3, 128:192]).max() / 4) if act_f16 else 1e20
        B[4, 192:256] *= 1e-4 if act_f16 else 1e-20
        B[5, 256:320] = np.tile(np.array([0.5, 1.5, 2.5, 127.0], np.float32), 16)
        if act_f16:
            case["B"] = B.astype(np.float16).astype(np.float32)
    A = orc.preprocess_weights(case["w"], bits, bm, kf)
    S = orc.preprocess_scales(case["sc"], case["zr"], bits, bm)
    wr = tm.TMACGeMMWrapper(act_group_size=ags)
    ws = [wr.register_weights(A, S, Mw, K, bits, tm.KCfg.make(Mw, K, bits, bm, kf, gs, ags, True, -1, N)) for _ in range(2)]
    Bt = torch.from_numpy(case["B"]).cuda()
    if act_f16:
        Bt = Bt.half()
    outs = [torch.empty((N, Mw), dtype=torch.float32, device="cuda") for _ in range(2)]
    tm.binding.check(tm.lib().tmac_hip_set_gemm_min_n(16))     # explicit threshold: these small matrices take the GEMM
    wr.fused(ws, Bt, outs, N)
    torch.cuda.synchronize()
    tm.lib().tmac_hip_set_gemm_min_n(32)
    if not edge:   # the default threshold sends an under-filled grid below 64 rows through the row loop: same results
        outs2 = [torch.empty_like(o) for o in outs]
        wr.fused(ws, Bt, outs2, N)
        torch.cuda.synchronize()
        for a, b in zip(outs, outs2):
            assert rel_err(b.cpu().numpy(), a.cpu().numpy()) <= 2e-6
    q, ls, lb, Cc, PS = oracle_case(case, A, S, Mw, K, bits, bm, kf, gs, ags, True, N=N)
    finite = np.isfinite(Cc)
    for o in outs:
        got = o.cpu().numpy()
        assert np.array_equal(np.isfinite(got), finite)
        for nrow in range(N):          # per activation row: a huge row must not hide an error in the others
            m = finite[nrow]
            if m.any():
                assert rel_err(got[nrow][m], Cc[nrow][m]) <= 2e-5
    tm.binding.check(tm.lib().tmac_hip_cache_clear())
    for w in ws:
        w.free()


def _random_prefill(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        bits = int(rng.choice([2, 4]))
        bm = 128 if bits == 2 else 256
        out.append((tuple((bm // bits) * int(rng.integers(4, 40)) for _ in range(int(rng.integers(1, 4)))), 128 * int(rng.integers(8, 70)),
                    bits, bm, bool(rng.integers(0, 2)), int(rng.integers(64, 400)), bool(rng.integers(0, 2))))
    return out


@pytest.mark.parametrize("cfg", _random_prefill(8, 5), ids=lambda c: "-".join(str(x) for x in c).replace(" ", ""))
def test_random_prefill_sampled_rows(tm, cfg):
    """random prefill calls at sizes the oracle cannot cover whole (1-3 matrices of 256-2500 rows, K up to 8960, 64-400
    activation rows, W2 / W4, zero points on / off): three sampled activation rows against the oracle, every 23rd against the
    decode kernel run on that row alone"""
    import torch
    Mws, K, bits, bm, zp, N, act_f16 = cfg
    kf, gs, ags = 16, 128, 64
    wr = tm.TMACGeMMWrapper(act_group_size=ags)
    B = np.random.default_rng(K + N).standard_normal((N, K)).astype(np.float32)
    if act_f16:
        B = B.astype(np.float16).astype(np.float32)
    ws, cases, outs = [], [], []
    for i, Mw in enumerate(Mws):
        case = orc.make_case(50 * i + K, Mw, K, N=1, bits=bits, gs=gs, ags=ags, zero_point=zp, fp16_values=True)
        A = orc.preprocess_weights(case["w"], bits, bm, kf)
        S = orc.preprocess_scales(case["sc"], case["zr"] if zp else None, bits, bm)
        ws.append(wr.register_weights(A, S, Mw, K, bits, tm.KCfg.make(Mw, K, bits, bm, kf, gs, ags, zp, -1, N), scales_dtype=tm.F32, dev_dtype=tm.F16))
        cases.append((case, A, S))
        outs.append(torch.empty((N, Mw), dtype=torch.float32, device="cuda"))
    Bt = torch.from_numpy(B).cuda()
    if act_f16:
        Bt = Bt.half()
    wr.fused(ws, Bt, outs, N)
    torch.cuda.synchronize()
    rows = [0, N // 2, N - 1]
    ones = [torch.empty((1, Mw), dtype=torch.float32, device="cuda") for Mw in Mws]
    gots = [o.cpu().numpy() for o in outs]
    for (case, A, S), got, Mw in zip(cases, gots, Mws):
        sub = dict(case, B=B[rows])
        _, _, _, Cc, _ = oracle_case(sub, A, S, Mw, K, bits, bm, kf, gs, ags, zp, N=len(rows))
        for i, r in enumerate(rows):
            assert rel_err(got[r], Cc[i]) <= 2e-5
    for r in range(0, N, 23):
        wr.fused(ws, Bt[r:r + 1].contiguous(), ones, 1)
        torch.cuda.synchronize()
        for got, one in zip(gots, ones):
            assert rel_err(got[r], one.cpu().numpy()[0]) <= 2e-5
    tm.lib().tmac_hip_cache_clear()
    for w in ws:
        w.free()


@pytest.mark.parametrize("Mw,K", [(4096, 4096), (4096, 11008)])
def test_prefill_full_size_sampled_rows(tm, Mw, K):
    """BASELINE configs[4] shape (llama-2-7B W2, N = 256) at full size through the fused entry point (pair-wise LUT build +
    one-hot GEMM).  The oracle is run on a sample of the activation rows only (it needs seconds per row); every other row
    is checked against the decode kernel run on that row alone -- an independent kernel with the same integer contract."""
    import torch
    bits, bm, kf, gs, ags, N = 2, 128, 16, 128, 64, 256
    case = orc.make_case(90 + K, Mw, K, N=N, bits=bits, gs=gs, ags=ags, fp16_values=True)
    A = orc.preprocess_weights(case["w"], bits, bm, kf)
    S = orc.preprocess_scales(case["sc"], case["zr"], bits, bm)
    wr = tm.TMACGeMMWrapper(act_group_size=ags)
    w = wr.register_weights(A, S, Mw, K, bits, tm.KCfg.make(Mw, K, bits, bm, kf, gs, ags, True, -1, N), scales_dtype=tm.F32, dev_dtype=tm.F16)
    Bt = torch.from_numpy(case["B"]).cuda().half()
    out = torch.empty((N, Mw), dtype=torch.float32, device="cuda")
    wr.fused([w], Bt, [out], N)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    rows = [0, 97, 255]
    sub = dict(case, B=case["B"][rows])
    _, _, _, Cc, _ = oracle_case(sub, A, S, Mw, K, bits, bm, kf, gs, ags, True, N=len(rows))
    for i, r in enumerate(rows):
        assert rel_err(got[r], Cc[i]) <= 2e-5
    one = torch.empty((1, Mw), dtype=torch.float32, device="cuda")
    for r in range(0, N, 15):
        wr.fused([w], Bt[r:r + 1].contiguous(), [one], 1)          # k_gemv_quad, LUT built in the kernel
        torch.cuda.synchronize()
        assert rel_err(got[r], one.cpu().numpy()[0]) <= 2e-5
    tm.binding.check(tm.lib().tmac_hip_cache_clear())
    w.free()


def test_edge_activations(tm):
    """all-zero act groups (scale 0 -> t_scales 0), huge/small magnitudes, exact .5 ties"""
    Mw, K, bits, bm, kf, gs, ags = 128, 512, 2, 128, 16, 128, 64
    case = orc.make_case(1, Mw, K, bits=bits)
    B = case["B"]
    B[0, :64] = 0
    B[0, 64:128] *= 1e20
    B[0, 128:192] *= 1e-20
    B[0, 192:256] = np.tile(np.array([0.5, 1.5, 2.5, 127.0], np.float32), 16)
    r = run_case(tm, case, Mw, K, bits, bm, kf, gs, ags, True)
    q, ls, lb, Cc, PS = oracle_case(case, r["A"], r["S"], Mw, K, bits, bm, kf, gs, ags, True)
    assert np.array_equal(r["q"], q) and np.array_equal(r["PS"], PS)
    check_bits(r["ls"], ls); check_bits(r["lb"], lb)
    assert rel_err(r["C"], Cc) <= 2e-5


@pytest.mark.parametrize("act_f16", [False, True])
def test_edge_activations_fused_lut_build(tm, act_f16):
    """the in-kernel LUT build (short exact divisions, magic-add rounding) on all-zero act groups, huge / tiny / denormal
    magnitudes, exact .5 ties and an all-ones-significand scale: scales, biases and integer sums bit-exact"""
    import torch
    Mw, K, bits, bm, kf, gs, ags = 128, 1024, 2, 128, 16, 128, 64
    case = orc.make_case(11, Mw, K, bits=bits)
    B = case["B"]
    B[0, :64] = 0
    B[0, 64:128] *= (60000.0 / np.abs(B[0, 64:128]).max() / 4) if act_f16 else 1e20
    B[0, 128:192] *= 1e-4 if act_f16 else 1e-20
    B[0, 192:256] = np.tile(np.array([0.5, 1.5, 2.5, 127.0], np.float32), 16)
    B[0, 256:320] = 0.0
    # fp16 denormal / fp32 value whose scale (abs-sum / 127) is an fp32 denormal: below the fast-division range, IEEE path.
    # (Smaller still, 1 / scale overflows and the reference itself saturates every entry to -128: outside the contract.)
    B[0, 256] = np.float32(6e-8) if act_f16 else np.float32(1e-36)
    B[0, 320:384] = 0.0
    B[0, 320:324] = np.float32(127.0 * (2.0 - 2.0 ** -23) / 4) if not act_f16 else np.float32(31.75)   # abs-sum / 127 = all-ones significand (fp32)
    if act_f16:
        B = B.astype(np.float16).astype(np.float32)
        case["B"] = B
    A = orc.preprocess_weights(case["w"], bits, bm, kf)
    S = orc.preprocess_scales(case["sc"], case["zr"], bits, bm)
    wr = tm.TMACGeMMWrapper(act_group_size=ags)
    w = wr.register_weights(A, S, Mw, K, bits, tm.KCfg.make(Mw, K, bits, bm, kf, gs, ags, True))
    Bt = torch.from_numpy(B).cuda()
    if act_f16:
        Bt = Bt.half()
    PS, Cf = wr.fused_partial_sums(w, Bt)
    q, ls, lb, Cc, PSo = oracle_case(case, A, S, Mw, K, bits, bm, kf, gs, ags, True)
    assert np.array_equal(PS, PSo)
    check_bits(wr.last_fused_lut[:, 0, :], ls)
    check_bits(wr.last_fused_lut[:, 1, :], lb)
    assert rel_err(Cf, Cc) <= 2e-5
    w.free()


def test_headline_shape_properties(tm):
    """Full BASELINE headline shape (Mw=4096, K=11008, W2, zp): oracle comparison + size-independent
    properties: run-to-run determinism and row-shard consistency (the multi-GPU partitioning)."""
    import torch
    Mw, K, bits, bm, kf, gs, ags = 4096, 11008, 2, 128, 16, 128, 64
    case = orc.make_case(2024, Mw, K, bits=bits)
    r = run_case(tm, case, Mw, K, bits, bm, kf, gs, ags, True, want_ps=False)
    r2 = run_case(tm, case, Mw, K, bits, bm, kf, gs, ags, True, want_ps=False)
    check_bits(r["C"], r2["C"])
    q, ls, lb, Cc, _ = oracle_case({**case, "w": case["w"][:64]}, r["A"][:1], r["S"][:1], 64, K, bits, bm, kf, gs, ags, True)
    assert np.array_equal(r["q"], q)
    Call = orc.qgemm_float(r["A"], q, r["S"], ls, lb, Mw, K, 1, bits, bm, kf, gs, ags, True)
    assert rel_err(r["C"], Call) <= 2e-5
    # row shards of 1024 rows (8 tiles each), as a 4-GPU split would register them
    parts = []
    for sh in range(4):
        sub = dict(w=case["w"][sh * 1024:(sh + 1) * 1024], sc=case["sc"][sh * 1024:(sh + 1) * 1024],
                   zr=case["zr"][sh * 1024:(sh + 1) * 1024], B=case["B"])
        parts.append(run_case(tm, sub, 1024, K, bits, bm, kf, gs, ags, True, want_ps=False)["C"])
    check_bits(np.concatenate(parts, axis=1), r["C"])


def test_host_pointer_cabi_matches_prebuilt_reference(tm, tmp_path):
    """the reference-named entry points (preprocessor_int8 / qgemm_lut_int8, host pointers, per-tile calls)
    against the vector produced by the reference's checked-in prebuilt kernel"""
    d = dict(np.load(os.path.join(GOLD, "prebuilt_llama2_7b_w2_k4096.npz")))
    L = tm.lib()
    ini = tmp_path / "kcfg.ini"
    ini.write_text("[qgemm_lut_t1_int8_m8192_k4096_n1_b2]\nbm = 128\nsimd_n_in = 16\nsimd_n_out = 8\nkfactor = 16\n"
                   "group_size = 128\nlut_scales_size = 64\nscales_size = 262144\nn_tile_num = 64\n")
    tm.binding.check(L.tmac_hip_load_kcfg(str(ini).encode()))
    K = 4096
    B = np.ascontiguousarray(d["B"][0]); ls = np.zeros(64, np.float32); lb = np.zeros(64, np.float32)
    q = np.zeros((K // 4, 16), np.int8)
    vp = lambda a: C.c_void_p(a.ctypes.data)
    fn = L.preprocessor_t1_int8_m8192_k4096_n1_b2
    fn.restype = C.c_int32
    assert fn(vp(B), vp(ls), vp(lb), vp(q)) == 0
    assert np.array_equal(q, d["qlut"]); check_bits(ls, d["lut_scales"]); check_bits(lb, d["lut_biases"])
    A = np.ascontiguousarray(d["A_ref"][0]); S = np.ascontiguousarray(d["S_ref"][0]); c = np.zeros(64, np.float32)
    assert L.qgemm_lut_int8(128, K, 1, 2, A.ctypes.data, q.ctypes.data, S.ctypes.data, ls.ctypes.data, lb.ctypes.data,
                            c.ctypes.data) == 0
    assert rel_err(c, d["C"]) <= 2e-5
    assert L.qgemm_lut_int8(96, K, 1, 2, A.ctypes.data, q.ctypes.data, S.ctypes.data, ls.ctypes.data, lb.ctypes.data,
                            c.ctypes.data) == -1          # unknown shape -> -1, as the reference dispatcher
    L.tmac_hip_cache_clear()


def test_host_pointer_cabi_serves_whole_runs(tm, tmp_path):
    """the per-tile host-pointer ABI driven like llama.cpp drives the reference (same LUT, tile after tile): from the second
    GEMV on, the contiguous tiles are computed as one matrix and the tile calls are served from that result -- same values
    as serving every tile on its own, for in-order and out-of-order tile calls, changing LUTs and two matrices of equal K"""
    L = tm.lib()
    L.tmac_hip_debug_host_runs.argtypes = [C.c_int]
    Mw, K, bits, bm, kf, gs, ags = 512, 1024, 2, 128, 16, 128, 64
    ntile, rpt = Mw * bits // bm, bm // bits
    ini = tmp_path / "kcfg.ini"
    ini.write_text(f"[qgemm_lut_t1_int8_m{Mw * bits}_k{K}_n1_b2]\nbm = {bm}\nsimd_n_in = 16\nsimd_n_out = 8\nkfactor = {kf}\n"
                   f"group_size = {gs}\nlut_scales_size = {K // ags}\nscales_size = {Mw * K // gs * 2}\nn_tile_num = {ntile}\n")
    tm.binding.check(L.tmac_hip_load_kcfg(str(ini).encode()))
    L.tmac_hip_cache_clear()
    vp = lambda a: C.c_void_p(a.ctypes.data)
    mats = []
    for seed in (1, 2):
        case = orc.make_case(500 + seed, Mw, K, bits=bits, gs=gs, ags=ags)
        A = np.ascontiguousarray(orc.preprocess_weights(case["w"], bits, bm, kf))       # [ntile][...]: tiles are contiguous
        S = np.ascontiguousarray(orc.preprocess_scales(case["sc"], case["zr"], bits, bm))
        mats.append((case, A, S))
    luts = []
    for seed in (11, 12, 13):
        B = np.random.default_rng(seed).standard_normal((1, K)).astype(np.float32)
        q = np.zeros((K // 4, 16), np.int8); ls = np.zeros(K // ags, np.float32); lb = np.zeros(K // ags, np.float32)
        assert L.preprocessor_int8(Mw * bits, K, 1, bits, vp(B), vp(ls), vp(lb), vp(q)) == 0
        qo, lso, lbo = orc.preprocessor(B, ags)
        assert np.array_equal(q, qo[0]); check_bits(ls, lso[0]); check_bits(lb, lbo[0])
        luts.append((q, ls, lb))

    def gemv(mat, lut, order):
        case, A, S = mat
        q, ls, lb = lut
        out = np.full(Mw, np.nan, np.float32)
        for t in order:
            c = np.zeros(rpt, np.float32)
            assert L.qgemm_lut_int8(bm, K, 1, bits, vp(A[t]), vp(q), vp(S[t]), vp(ls), vp(lb), vp(c)) == 0, L.tmac_hip_last_error()
            out[t * rpt:(t + 1) * rpt] = c
        return out

    def expect(mat, lut):
        case, A, S = mat
        q, ls, lb = lut
        return orc.qgemm_float(A, q[None], S, ls[None], lb[None], Mw, K, 1, bits, bm, kf, gs, ags, True)[0]

    fwd, rev = list(range(ntile)), [5, 0, 7, 2, 1, 6, 3, 4]
    for runs in (1, 0):
        L.tmac_hip_debug_host_runs(runs)
        L.tmac_hip_cache_clear()
        seq = [(0, 0, fwd), (1, 0, fwd), (0, 1, fwd), (1, 1, rev), (0, 2, rev), (0, 0, fwd), (1, 2, fwd[:3]), (0, 2, fwd)]
        for mi, li, order in seq:
            got = gemv(mats[mi], luts[li], order)
            want = expect(mats[mi], luts[li])
            idx = np.concatenate([np.arange(t * rpt, (t + 1) * rpt) for t in order])
            assert rel_err(got[idx], want[idx]) <= 2e-5, (runs, mi, li)
    L.tmac_hip_debug_host_runs(1)
    L.tmac_hip_cache_clear()


FUSED_CFGS = [c for c in CFGS if (c[6] == 64 and c[8] == -1) or c[6] == c[1]]


@pytest.mark.parametrize("variant", [0, 4, 5, 7])
@pytest.mark.parametrize("act_f16", [False, True])
@pytest.mark.parametrize("Mw,K,bits,bm,kf,gs,ags,zp,mg", FUSED_CFGS)
def test_fused_kernel_builds_the_same_lut(tm, Mw, K, bits, bm, kf, gs, ags, zp, mg, act_f16, variant):
    """tmac_hip_qgemm_fused_dev: LUT constructed inside the GEMV kernel.  The integer partial sums can only
    be bit-exact if the in-kernel LUT (QLUT, and through C the scales/biases) equals the oracle's."""
    import torch
    case = orc.make_case(7 * Mw + K, Mw, K, bits=bits, gs=gs, ags=ags, zero_point=zp, m_groups=mg, fp16_values=act_f16)
    A = orc.preprocess_weights(case["w"], bits, bm, kf)
    S = orc.preprocess_scales(case["sc"], case["zr"] if zp else None, bits, bm) if mg == -1 else case["sc"]
    cfg = tm.KCfg.make(Mw, K, bits, bm, kf, gs, ags, zp, mg)
    tm.binding.check(tm.lib().tmac_hip_set_variant(variant))   # 0: quad kernel, MFMA accumulate (default); 7: quad kernel, v_mqsad accumulate; 4 / 5: row-block fused kernel
    wr = tm.TMACGeMMWrapper(act_group_size=ags)
    wr.set_workspace(K, 1)
    w = wr.register_weights(A, S, Mw, K, bits, cfg)
    Bt = torch.from_numpy(case["B"]).cuda()
    if act_f16:
        Bt = Bt.half()
    PS, Cf = wr.fused_partial_sums(w, Bt)
    Ct = torch.empty((1, Mw), dtype=torch.float32, device="cuda")
    wr.fused([w], Bt, [Ct])
    torch.cuda.synchronize()
    q, ls, lb, Cc, PSo = oracle_case(case, A, S, Mw, K, bits, bm, kf, gs, ags, zp, mg)
    assert np.array_equal(PS, PSo)
    check_bits(wr.last_fused_lut[:, 0, :], ls)        # LUT scales built in LDS, bit for bit
    check_bits(wr.last_fused_lut[:, 1, :], lb)        # LUT biases (the reference's horizontal-add order)
    assert rel_err(Cf, Cc) <= 2e-5
    # the tap launch may use another (threads, waves-per-quad) configuration, i.e. another fp32 summation order
    assert rel_err(Ct.cpu().numpy(), Cf) <= 2e-6
    w.free()
    tm.lib().tmac_hip_set_variant(0)


def test_fused_multi_matrix_launch(tm):
    """q/k/v-style: three matrices, one activation vector, one launch == three separate launches, bit for bit"""
    import torch
    K, bits, bm, kf, gs, ags = 4096, 2, 128, 16, 128, 64
    rows = [256, 512, 128]
    wr = tm.TMACGeMMWrapper(act_group_size=ags)
    wr.set_workspace(K, 1)
    Bv = orc.make_case(5, 64, K)["B"]
    Bt = torch.from_numpy(Bv).cuda().half()
    ws, refs = [], []
    for i, Mw in enumerate(rows):
        case = orc.make_case(100 + i, Mw, K, bits=bits)
        A = orc.preprocess_weights(case["w"], bits, bm, kf)
        S = orc.preprocess_scales(case["sc"], case["zr"], bits, bm)
        w = wr.register_weights(A, S, Mw, K, bits, tm.KCfg.make(Mw, K, bits, bm, kf, gs, ags, True))
        ws.append(w)
        q, ls, lb = orc.preprocessor(Bt.float().cpu().numpy(), ags)
        refs.append(orc.qgemm_float(A, q, S, ls, lb, Mw, K, 1, bits, bm, kf, gs, ags, True))
    outs = [torch.empty((1, Mw), dtype=torch.float32, device="cuda") for Mw in rows]
    wr.fused(ws, Bt, outs)
    singles = [torch.empty((1, Mw), dtype=torch.float32, device="cuda") for Mw in rows]
    for w, c in zip(ws, singles):
        wr.fused([w], Bt, [c])
    torch.cuda.synchronize()
    for o, s1, ref in zip(outs, singles, refs):
        check_bits(o.cpu().numpy(), s1.cpu().numpy())
        assert rel_err(o.cpu().numpy(), ref) <= 2e-5
    for w in ws:
        w.free()


@pytest.mark.parametrize("variant", [0, 7])
@pytest.mark.parametrize("ft,wpq", [(512, 1), (512, 2), (768, 1), (768, 2), (768, 3), (1024, 1), (1024, 2), (1024, 4)])
@pytest.mark.parametrize("Mw,K,bits,bm,kf,gs,ags,zp,mg", [CFGS[0], CFGS[1], CFGS[3], CFGS[8]])
def test_quad_kernel_configurations(tm, Mw, K, bits, bm, kf, gs, ags, zp, mg, ft, wpq, variant):
    """every (threads per workgroup, waves per quad, accumulate) configuration of k_gemv_quad, LUT built
    in-kernel; the integer tap exists in the 512-thread configurations"""
    import torch
    if variant == 7 and ft != 512:
        pytest.skip("the v_mqsad accumulate is instantiated for 512-thread workgroups only")
    L = tm.lib()
    L.tmac_hip_debug_quad_config.argtypes = [C.c_int, C.c_int]
    case = orc.make_case(3 * Mw + K, Mw, K, bits=bits, gs=gs, ags=ags, zero_point=zp, m_groups=mg)
    A = orc.preprocess_weights(case["w"], bits, bm, kf)
    S = orc.preprocess_scales(case["sc"], case["zr"] if zp else None, bits, bm) if mg == -1 else case["sc"]
    tm.binding.check(L.tmac_hip_set_variant(variant))
    wr = tm.TMACGeMMWrapper(act_group_size=ags)
    wr.set_workspace(K, 1)
    w = wr.register_weights(A, S, Mw, K, bits, tm.KCfg.make(Mw, K, bits, bm, kf, gs, ags, zp, mg))
    Bt = torch.from_numpy(case["B"]).cuda()
    Ct = torch.empty((1, Mw), dtype=torch.float32, device="cuda")
    q, ls, lb, Cc, PSo = oracle_case(case, A, S, Mw, K, bits, bm, kf, gs, ags, zp, mg)
    L.tmac_hip_debug_quad_config(ft, wpq)
    try:
        wr.fused([w], Bt, [Ct])
        torch.cuda.synchronize()
        assert rel_err(Ct.cpu().numpy(), Cc) <= 2e-5
        if ft == 512:
            PS, Cf = wr.fused_partial_sums(w, Bt)
            assert np.array_equal(PS, PSo)
            check_bits(wr.last_fused_lut[:, 0, :], ls)
            check_bits(wr.last_fused_lut[:, 1, :], lb)
            check_bits(Ct.cpu().numpy(), Cf)
    finally:
        L.tmac_hip_debug_quad_config(0, 0)
        L.tmac_hip_set_variant(0)
    w.free()


@pytest.mark.parametrize("Mw,K,bits,bm,zp,mg,ags", [
    (4096, 14336, 2, 128, True, -1, 64),     # llama-3-8b down: 7 steps, 4 waves per quad, 1024-thread LUT build
    (14336, 4096, 2, 256, True, -1, 64),     # llama-3-8b gate/up: one quad per wave
    (1024, 4096, 2, 512, True, -1, 64),      # llama-3-8b k/v (GQA)
    (5120, 13824, 2, 128, True, -1, 64),     # llama-2-13b down
    (13824, 5120, 2, 128, True, -1, 64),     # llama-2-13b gate/up
    (4096, 11008, 4, 256, True, -1, 64),     # llama-2-7b W4 down: 3 waves per quad, 2-deep fragment ring
    (3200, 8640, 2, 128, False, 1, 8640),    # BitNet-3B down: unified scale, act group = K
    (8640, 3200, 2, 128, False, 1, 3200),    # BitNet-3B gate/up
    (4096, 4096, 3, 192, True, -1, 64),      # W3: three uint4 per unit, 2-deep ring
    (4096, 11008, 1, 128, True, -1, 64),     # W1
])
def test_model_shape_zoo(tm, Mw, K, bits, bm, zp, mg, ags):
    """full-size shapes of the reference's preset models through the default (fused, auto-configured) path: every
    launch-configuration branch of the quad kernel's heuristic against the oracle"""
    import torch
    gs = 128
    tm.binding.check(tm.lib().tmac_hip_set_variant(0))
    case = orc.make_case(Mw + K + bits, Mw, K, bits=bits, gs=gs, ags=ags, zero_point=zp, m_groups=mg)
    A = orc.preprocess_weights(case["w"], bits, bm, 16)
    S = orc.preprocess_scales(case["sc"], case["zr"] if zp else None, bits, bm) if mg == -1 else case["sc"]
    wr = tm.TMACGeMMWrapper(act_group_size=ags)
    wr.set_workspace(K, 1)
    w = wr.register_weights(A, S, Mw, K, bits, tm.KCfg.make(Mw, K, bits, bm, 16, gs, ags, zp, mg))
    Bt = torch.from_numpy(case["B"]).cuda()
    Ct = torch.empty((1, Mw), dtype=torch.float32, device="cuda")
    wr.fused([w], Bt, [Ct])
    torch.cuda.synchronize()
    q, ls, lb, Cc, PSo = oracle_case(case, A, S, Mw, K, bits, bm, 16, gs, ags, zp, mg)
    assert rel_err(Ct.cpu().numpy(), Cc) <= 2e-5
    C2 = torch.empty_like(Ct)
    wr.fused([w], Bt, [C2])
    torch.cuda.synchronize()
    check_bits(Ct.cpu().numpy(), C2.cpu().numpy())
    w.free()
