"""Pins the scalar oracle (oracle/tmac_oracle.c) against the reference itself.

Runs only where oracle/_ref/*.so exists (built from /root/reference by `make -C oracle ref`;
the prebuilt files travel to the GPU box).  Everything here is CPU-only.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.skipif(not orc.have_ref("intrins"), reason="oracle/_ref not built")


def _ref_python_preprocess_weights():
    if not os.path.isdir(orc.REF_ROOT):
        return None
    sys.path.insert(0, os.path.join(orc.REF_ROOT, "python"))
    try:
        from t_mac.weights import preprocess_weights  # numpy-only, imports without TVM
        return preprocess_weights
    finally:
        sys.path.pop(0)


@pytest.mark.parametrize("K,ags", [(256, 64), (4096, 64), (11008, 64), (1024, 32), (3200, 3200), (8640, 8640)])
def test_preprocessor_bit_exact(K, ags):
    rng = np.random.default_rng(K + ags)
    B = rng.standard_normal((1, K)).astype(np.float32)
    q, ls, lb = orc.preprocessor(B, ags)
    qr, lsr, lbr = orc.ref_preprocessor(B[0], ags)
    assert np.array_equal(q[0], qr)
    assert np.array_equal(ls[0].view(np.uint32), lsr.view(np.uint32))
    assert np.array_equal(lb[0].view(np.uint32), lbr.view(np.uint32))


def test_preprocessor_edge_values():
    # zero group (scale 0 -> t_scales 0), huge dynamic range, exact ties for RNE
    K, ags = 256, 64
    B = np.zeros((1, K), np.float32)
    B[0, 64:128] = np.linspace(-3, 3, 64, dtype=np.float32)
    B[0, 128:192] = 1e-30
    B[0, 192:256] = np.tile(np.array([0.5, 1.5, 2.5, 127.0], np.float32), 16)
    q, ls, lb = orc.preprocessor(B, ags)
    qr, lsr, lbr = orc.ref_preprocessor(B[0], ags)
    assert np.array_equal(q[0], qr) and np.array_equal(ls[0], lsr) and np.array_equal(lb[0], lbr)
    assert ls[0, 0] == 0 and not q[0, :16].any()
    # exact antisymmetry the GPU half-table relies on (lut_ctor.cc:152-155)
    assert np.array_equal(q[0][:, ::-1].astype(np.int16), -q[0].astype(np.int16))


@pytest.mark.parametrize("bits,bm,kfactor", [(1, 128, 16), (2, 128, 16), (2, 256, 16), (3, 192, 16), (4, 256, 16),
                                             (2, 128, 8), (2, 320, 16)])
def test_preprocess_weights_matches_reference_python(bits, bm, kfactor):
    ref_pw = _ref_python_preprocess_weights()
    if ref_pw is None:
        pytest.skip("/root/reference not present")
    Mw, K = bm // bits * 3 if bm != 320 else 480, 512
    case = orc.make_case(7, Mw, K, bits=bits)
    A = orc.preprocess_weights(case["w"], bits, bm, kfactor)
    A_ref, S_ref = ref_pw(case["w"], case["sc"], case["zr"], bits=bits, g=4, bm=bm, kfactor=kfactor)
    assert np.array_equal(A, A_ref)
    S = orc.preprocess_scales(case["sc"], case["zr"], bits, bm)
    assert np.array_equal(S, S_ref.astype(np.float32))
    S1 = orc.preprocess_scales(case["sc"], None, bits, bm)
    _, S1_ref = ref_pw(case["w"], case["sc"], None, bits=bits, g=4, bm=bm, kfactor=kfactor)
    assert np.array_equal(S1, S1_ref.astype(np.float32))


CFGS = [  # bits, bm, kfactor, gs, ags, zp
    (2, 128, 16, 128, 64, True), (2, 128, 16, 128, 64, False), (4, 256, 16, 128, 64, True),
    (1, 128, 16, 128, 64, True), (3, 192, 16, 128, 64, False), (2, 128, 8, 128, 32, True),
    (2, 128, 16, 128, 32, True), (4, 256, 8, 64, 32, True),
]


@pytest.mark.parametrize("bits,bm,kfactor,gs,ags,zp", CFGS)
def test_float_path_bit_exact_vs_reference_intrinsics(bits, bm, kfactor, gs, ags, zp):
    Mw, K = bm // bits * 2, 1024
    case = orc.make_case(11 * bits + ags, Mw, K, bits=bits, gs=gs, ags=ags, zero_point=zp)
    A = orc.preprocess_weights(case["w"], bits, bm, kfactor)
    S = orc.preprocess_scales(case["sc"], case["zr"], bits, bm)
    q, ls, lb = orc.preprocessor(case["B"], ags)
    Cor = orc.qgemm_float(A, q, S, ls, lb, Mw, K, 1, bits, bm, kfactor, gs, ags, zp)
    cbits = orc.ref_cbits_float(A, q[0], S, ls[0], lb[0], Mw, K, bits, bm, kfactor, gs, ags, zp)
    Cref = orc.combine_planes(cbits, Mw, bits)
    assert np.array_equal(Cor[0].view(np.uint32), Cref.view(np.uint32))
    # integer partial sums: scalar restatement == reference int32 intrinsic per act group
    if ags % (4 * kfactor) == 0 and (kfactor, bits) in [(16, 1), (16, 2), (16, 3), (16, 4), (8, 2)]:
        PS = orc.partial_sums(A, q[0], Mw, K, bits, bm, kfactor, ags)
        assert np.array_equal(PS, orc.ref_partial_sums(A, q[0], Mw, K, bits, bm, kfactor, ags))
    # and the statistical check of tests/test_e2e.py (NMSE <= 5e-4, qgemm.py:277-282)
    Cdq = orc.dequant_matmul(case["w"], case["sc"], case["zr"], case["B"], bits, gs)[0]
    nmse = np.mean((Cdq - Cor[0]) ** 2) / np.mean(Cdq ** 2)
    assert nmse < 5e-4


FA_CFGS = [  # bits, bm, kfactor, gs, ags, zp  (the FastAggregation = true instantiations of oracle/ref_shim.cc)
    (2, 128, 16, 128, 64, True), (2, 128, 16, 128, 64, False), (4, 256, 16, 128, 64, True), (4, 256, 16, 128, 64, False),
    (1, 128, 16, 128, 64, True), (3, 192, 16, 128, 64, True), (2, 128, 8, 128, 32, True), (2, 128, 16, 128, 32, True),
]


@pytest.mark.parametrize("bits,bm,kfactor,gs,ags,zp", FA_CFGS)
def test_fast_aggregation_bit_exact_vs_reference_intrinsics(bits, bm, kfactor, gs, ags, zp):
    """(a9) the halving-adder tree, the ActK rescale and the analytic bias (tbl.cc:201-256,301-318,474-477):
    fa_mode 2 of the restatement == the reference's own FastAggregation build on this x86 host, to the bit."""
    Mw, K = bm // bits * 2, 1024
    case = orc.make_case(5 * bits + ags + 1, Mw, K, bits=bits, gs=gs, ags=ags, zero_point=zp)
    A = orc.preprocess_weights(case["w"], bits, bm, kfactor)
    S = orc.preprocess_scales(case["sc"], case["zr"], bits, bm)
    q, ls, lb = orc.preprocessor(case["B"], ags)
    Cor, tap = orc.qgemm_float_fa(A, q, S, ls, lb, Mw, K, 1, bits, bm, kfactor, gs, ags, zp, fa_mode=2)
    cbits = orc.ref_cbits_float(A, q[0], S, ls[0], lb[0], Mw, K, bits, bm, kfactor, gs, ags, zp, fa=True)
    Cref = orc.combine_planes(cbits, Mw, bits)
    assert np.array_equal(Cor[0].view(np.uint32), Cref.view(np.uint32))
    assert tap.min() >= -128 and tap.max() <= 127


def _call_prebuilt(setname, bm, K, bits, Mw_total_bits_name, case, gs=128, ags=64):
    """Drive a checked-in prebuilt kernel set exactly as llama.cpp would (per-tile pointers)."""
    L = orc.ref_lib(setname)
    Mw = case["w"].shape[0]
    kfactor = 16
    A = orc.preprocess_weights(case["w"], bits, bm, kfactor)
    S = orc.preprocess_scales(case["sc"], case["zr"], bits, bm)
    B = np.ascontiguousarray(case["B"][0])
    G = K // ags
    ls = np.zeros(G, np.float32); lb = np.zeros(G, np.float32); q = np.zeros((K // 4, 16), np.int8)
    pre = getattr(L, f"preprocessor_t1_int8_m{Mw_total_bits_name}_k{K}_n1_b{bits}")
    assert pre(orc._p(B), orc._p(ls), orc._p(lb), orc._p(q)) == 0
    qg = getattr(L, f"qgemm_lut_t1_int8_m{bm}_k{K}_n1_b{bits}")
    Cout = np.zeros(Mw, np.float32)
    rpt = bm // bits
    for tile in range(Mw * bits // bm):
        c = np.zeros(rpt, np.float32)
        assert qg(orc._p(A[tile]), orc._p(q), orc._p(S[tile]), orc._p(ls), orc._p(lb), orc._p(c)) == 0
        Cout[tile * rpt:(tile + 1) * rpt] = c
    return A, S, q, ls, lb, Cout


@pytest.mark.parametrize("setname,bits,bm,Mw,K,mname", [
    ("aarch64-llama-2-7b-2bit", 2, 128, 4096, 4096, 8192),     # BASELINE config #1 (tests/test_e2e.py)
    ("aarch64-llama-2-7b-2bit", 2, 128, 512, 11008, 8192),     # headline shape, 8 tiles
    ("aarch64-llama-2-7b-4bit", 4, 256, 512, 4096, 44032),     # bm=256 kernel only (SURVEY §8c caveat)
    ("aarch64-llama-3-8b-2bit", 2, 128, 256, 14336, 8192),
])
def test_prebuilt_reference_kernels_bit_exact(setname, bits, bm, Mw, K, mname):
    if not orc.have_ref(setname):
        pytest.skip("prebuilt set not compiled")
    case = orc.make_case(0, Mw, K, bits=bits, zero_point=True)
    A, S, q, ls, lb, Cref = _call_prebuilt(setname, bm, K, bits, mname, case)
    qo, lso, lbo = orc.preprocessor(case["B"], 64)
    assert np.array_equal(qo[0], q) and np.array_equal(lso[0], ls) and np.array_equal(lbo[0], lb)
    Cor = orc.qgemm_float(A, qo, S, lso, lbo, Mw, K, 1, bits, bm, 16, 128, 64, True)
    assert np.array_equal(Cor[0].view(np.uint32), Cref.view(np.uint32))
    Cdq = orc.dequant_matmul(case["w"], case["sc"], case["zr"], case["B"], bits, 128)[0]
    assert np.mean((Cdq - Cref) ** 2) / np.mean(Cdq ** 2) < 5e-4


@pytest.mark.parametrize("bits,bm,Mw,K", [(2, 128, 256, 3200), (2, 320, 320, 3200), (2, 128, 128, 8640), (4, 256, 128, 1024),
                                          # the widths k_gemm_planes_us and the chain cover since rounds 3 / 4
                                          (1, 64, 128, 1024), (1, 64, 320, 8640), (3, 192, 192, 3200), (3, 192, 64, 12288), (4, 256, 192, 12288)])
def test_int32_scale_final_path(bits, bm, Mw, K):
    case = orc.make_case(5, Mw, K, bits=bits, m_groups=1, ags=K, zero_point=False)
    A = orc.preprocess_weights(case["w"], bits, bm, 16)
    q, ls, lb = orc.preprocessor(case["B"], K)
    Cor, cb = orc.qgemm_scale_final(A, q, case["sc"], ls[:, 0], lb[:, 0], Mw, K, 1, bits, bm, 16, 1)
    assert np.array_equal(cb[0], orc.ref_cbits_int32(A, q[0], Mw, K, bits, bm, 16))
    Cdq = orc.dequant_matmul(case["w"], case["sc"], None, case["B"], bits, 128, m_groups=1)[0]
    assert np.mean((Cdq - Cor[0]) ** 2) / np.mean(Cdq ** 2) < 5e-4
