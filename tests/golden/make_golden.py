#!/usr/bin/env python3
"""Generates tests/golden/*.npz and lut_ctor_kat.json FROM THE REFERENCE ITSELF.

Run in the build container (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_golden.py

Sources of truth used here (never our own restatement):
  * weight/scale permutation : python/t_mac/weights.py:preprocess_weights, imported from /root/reference
  * QLUT / lut_scales / lut_biases, CBits, integer partial sums :
        python/t_mac/intrins/{lut_ctor,tbl}.cc compiled by oracle/Makefile into oracle/_ref/
  * one case through a checked-in prebuilt kernel set (deploy/tuned/aarch64-llama-2-7b-2bit/kernels.cc)
  * the known-answer vector printed by the reference's own tests/test_lut_ctor.cc
The vectors are committed; the GPU box has no /root/reference.
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402  (only its _ref loaders + case generator are used)

REF = orc.REF_ROOT
sys.path.insert(0, os.path.join(REF, "python"))
from t_mac.weights import preprocess_weights as ref_preprocess_weights  # noqa: E402


def gen(name, seed, Mw, K, bits, bm, kfactor, gs, ags, zp, m_groups=-1):
    case = orc.make_case(seed, Mw, K, bits=bits, gs=gs, ags=ags, zero_point=zp, m_groups=m_groups)
    A, S = ref_preprocess_weights(case["w"], case["sc"], case["zr"], bits=bits, g=4, bm=bm, kfactor=kfactor)
    A = np.ascontiguousarray(A, np.uint8)
    S = np.ascontiguousarray(S, np.float32)
    q, ls, lb = orc.ref_preprocessor(case["B"][0], ags)
    out = dict(w=case["w"], sc=case["sc"], B=case["B"], A_ref=A, S_ref=S, qlut=q, lut_scales=ls, lut_biases=lb,
               meta=np.array([Mw, K, bits, bm, kfactor, gs, ags, int(zp), m_groups], np.int64))
    if case["zr"] is not None:
        out["zr"] = case["zr"]
    if m_groups == -1:
        cbits = orc.ref_cbits_float(A, q, S, ls, lb, Mw, K, bits, bm, kfactor, gs, ags, zp)
        out["cbits"] = cbits
        out["C"] = orc.combine_planes(cbits, Mw, bits)
        out["PS"] = orc.ref_partial_sums(A, q, Mw, K, bits, bm, kfactor, ags)
    else:
        out["cbits32"] = orc.ref_cbits_int32(A, q, Mw, K, bits, bm, kfactor)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: v.shape for k, v in out.items()})


def gen_fa(name):
    """(a9) the same inputs through the reference's FastAggregation = true instantiation of the tbl intrinsic (its AVX2
    flavour, the one that runs on this host): tests/golden/fa/<name>.npz holds only the outputs."""
    d = dict(np.load(os.path.join(HERE, name + ".npz")))
    Mw, K, bits, bm, kfactor, gs, ags, zp, mg = [int(x) for x in d["meta"]]
    cbits = orc.ref_cbits_float(d["A_ref"], d["qlut"], d["S_ref"], d["lut_scales"], d["lut_biases"], Mw, K, bits, bm,
                                kfactor, gs, ags, bool(zp), fa=True)
    os.makedirs(os.path.join(HERE, "fa"), exist_ok=True)
    np.savez_compressed(os.path.join(HERE, "fa", name + ".npz"), cbits_fa=cbits, C_fa=orc.combine_planes(cbits, Mw, bits),
                        meta=d["meta"])
    print("fa/" + name)


def gen_prebuilt(name, setname, seed, Mw, K, bits, bm, mname):
    """Through the checked-in prebuilt C-ABI kernels, driven tile by tile like llama.cpp."""
    L = orc.ref_lib(setname)
    case = orc.make_case(seed, Mw, K, bits=bits, zero_point=True)
    A, S = ref_preprocess_weights(case["w"], case["sc"], case["zr"], bits=bits, g=4, bm=bm, kfactor=16)
    A = np.ascontiguousarray(A, np.uint8); S = np.ascontiguousarray(S, np.float32)
    B = np.ascontiguousarray(case["B"][0])
    G = K // 64
    ls = np.zeros(G, np.float32); lb = np.zeros(G, np.float32); q = np.zeros((K // 4, 16), np.int8)
    assert getattr(L, f"preprocessor_t1_int8_m{mname}_k{K}_n1_b{bits}")(orc._p(B), orc._p(ls), orc._p(lb), orc._p(q)) == 0
    qg = getattr(L, f"qgemm_lut_t1_int8_m{bm}_k{K}_n1_b{bits}")
    rpt = bm // bits
    Cout = np.zeros(Mw, np.float32)
    for tile in range(Mw * bits // bm):
        c = np.zeros(rpt, np.float32)
        assert qg(orc._p(A[tile]), orc._p(q), orc._p(S[tile]), orc._p(ls), orc._p(lb), orc._p(c)) == 0
        Cout[tile * rpt:(tile + 1) * rpt] = c
    np.savez_compressed(os.path.join(HERE, name + ".npz"), w=case["w"], sc=case["sc"], zr=case["zr"], B=case["B"],
                        A_ref=A, S_ref=S, qlut=q, lut_scales=ls, lut_biases=lb, C=Cout,
                        PS=orc.ref_partial_sums(A, q, Mw, K, bits, bm, 16, 64),
                        meta=np.array([Mw, K, bits, bm, 16, 128, 64, 1, -1], np.int64))
    print(name, "via", setname)


def gen_kat():
    """Compile and run the reference's own tests/test_lut_ctor.cc; record what it prints."""
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "t")
        subprocess.run(["g++", "-O2", "-mavx2", "-mfma", "-ffp-contract=off", "-std=c++17", "-w", "-fpermissive",
                        "-include", "cstdio", "-I", os.path.join(REF, "python/t_mac/intrins"),
                        os.path.join(REF, "tests/test_lut_ctor.cc"), "-o", exe], check=True)
        txt = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    lines = txt.strip().splitlines()
    kat = dict(source="tests/test_lut_ctor.cc (b[i]=i, i<32; act_k=32)", raw=lines,
               lut_scales=float(lines[1].split(":")[1]), lut_biases=float(lines[2].split(":")[1]),
               qlut=[[int(x) for x in l.split()] for l in lines[3:11]])
    json.dump(kat, open(os.path.join(HERE, "lut_ctor_kat.json"), "w"), indent=1)
    print("KAT", kat["lut_scales"], kat["lut_biases"])


if __name__ == "__main__":
    gen_kat()
    gen("w2_zp_g128_a64", 1, 128, 512, 2, 128, 16, 128, 64, True)
    gen("w2_nozp_g128_a64", 2, 128, 512, 2, 128, 16, 128, 64, False)
    gen("w4_zp_g128_a64", 3, 128, 512, 4, 256, 16, 128, 64, True)
    gen("w1_zp_g128_a64", 4, 256, 256, 1, 128, 16, 128, 64, True)
    gen("w3_nozp_g128_a64", 5, 128, 256, 3, 192, 16, 128, 64, False)
    gen("w2_zp_g128_a32_kf8", 6, 64, 512, 2, 128, 8, 128, 32, True)
    gen("bitnet_w2_int32_k640", 7, 160, 640, 2, 320, 16, 128, 640, False, m_groups=1)
    for n in ("w2_zp_g128_a64", "w2_nozp_g128_a64", "w4_zp_g128_a64", "w1_zp_g128_a64", "w3_nozp_g128_a64", "w2_zp_g128_a32_kf8"):
        gen_fa(n)
    gen_prebuilt("prebuilt_llama2_7b_w2_k4096", "aarch64-llama-2-7b-2bit", 0, 64, 4096, 2, 128, 8192)
