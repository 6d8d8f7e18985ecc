"""Oracle (oracle/tmac_oracle.c) vs the committed golden vectors (tests/golden/, produced from the
reference itself by tests/golden/make_golden.py).  CPU only; runs anywhere gcc exists."""
import glob
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*.npz")))


def load(name):
    d = dict(np.load(os.path.join(GOLD, name + ".npz")))
    Mw, K, bits, bm, kfactor, gs, ags, zp, m_groups = [int(x) for x in d["meta"]]
    d["cfg"] = dict(Mw=Mw, K=K, bits=bits, bm=bm, kfactor=kfactor, gs=gs, ags=ags, zp=bool(zp), m_groups=m_groups)
    return d


def test_lut_ctor_known_answer():
    """tests/test_lut_ctor.cc of the reference: b[i] = i, one 32-activation group."""
    kat = json.load(open(os.path.join(GOLD, "lut_ctor_kat.json")))
    b = np.arange(32, dtype=np.float32)[None, :]
    q, ls, lb = orc.preprocessor(b, 32)
    assert q[0].tolist() == kat["qlut"]
    assert f"{ls[0, 0]:.6f}" == f"{kat['lut_scales']:.6f}"
    assert lb[0, 0] == kat["lut_biases"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(name):
    d = load(name); c = d["cfg"]
    A = orc.preprocess_weights(d["w"], c["bits"], c["bm"], c["kfactor"])
    assert np.array_equal(A, d["A_ref"])
    q, ls, lb = orc.preprocessor(d["B"], c["ags"])
    assert np.array_equal(q[0], d["qlut"])
    assert np.array_equal(ls[0].view(np.uint32), d["lut_scales"].view(np.uint32))
    assert np.array_equal(lb[0].view(np.uint32), d["lut_biases"].view(np.uint32))
    if c["m_groups"] == -1:
        S = orc.preprocess_scales(d["sc"], d.get("zr"), c["bits"], c["bm"])
        assert np.array_equal(S, d["S_ref"])
        PS = orc.partial_sums(A, q[0], c["Mw"], c["K"], c["bits"], c["bm"], c["kfactor"], c["ags"])
        assert np.array_equal(PS, d["PS"])
        Cc = orc.qgemm_float(A, q, S, ls, lb, c["Mw"], c["K"], 1, c["bits"], c["bm"], c["kfactor"], c["gs"],
                             c["ags"], c["zp"])
        assert np.array_equal(Cc[0].view(np.uint32), d["C"].view(np.uint32))
    else:
        _, cb = orc.qgemm_scale_final(A, q, d["sc"], ls[:, 0], lb[:, 0], c["Mw"], c["K"], 1, c["bits"], c["bm"],
                                      c["kfactor"], c["m_groups"])
        assert np.array_equal(cb[0], d["cbits32"])


FA_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "fa", "*.npz")))


@pytest.mark.parametrize("name", FA_CASES)
def test_fast_aggregation_reproduces_golden(name):
    """(a9) tests/golden/fa/*.npz: the golden inputs through the reference's FastAggregation = true intrinsic as built on
    x86 (make_golden.py gen_fa).  fa_mode 2 of the restatement must reproduce it to the bit."""
    d = load(name); c = d["cfg"]
    g = dict(np.load(os.path.join(GOLD, "fa", name + ".npz")))
    Cc, tap = orc.qgemm_float_fa(d["A_ref"], d["qlut"][None], d["S_ref"], d["lut_scales"][None], d["lut_biases"][None],
                                 c["Mw"], c["K"], 1, c["bits"], c["bm"], c["kfactor"], c["gs"], c["ags"], c["zp"], fa_mode=2)
    assert np.array_equal(Cc[0].view(np.uint32), g["C_fa"].view(np.uint32))


@pytest.mark.parametrize("bits,bm,kf,ags", [(2, 128, 16, 64), (4, 256, 16, 64), (3, 192, 16, 64), (2, 128, 8, 32)])
def test_fast_aggregation_signed_flavour_properties(bits, bm, kf, ags):
    """fa_mode 1 (vrhaddq_s8, the reference's ARM build; it cannot run here, so this flavour is pinned through the tree
    it shares with mode 2 plus what the arithmetic implies): every level rounds up by at most 1/2, so the tree result
    lies in [mean, mean + log2(ActK)/2]; with the analytic bias the output stays close to the exact path."""
    Mw, K, gs = 64 if bits != 3 else 128, 2048, 128
    case = orc.make_case(77 + bits, Mw, K, bits=bits, gs=gs, ags=ags)
    A = orc.preprocess_weights(case["w"], bits, bm, kf)
    S = orc.preprocess_scales(case["sc"], case["zr"], bits, bm)
    q, ls, lb = orc.preprocessor(case["B"], ags)
    C, tap = orc.qgemm_float_fa(A, q, S, ls, lb, Mw, K, 1, bits, bm, kf, gs, ags, True, fa_mode=1)
    PS = orc.partial_sums(A, q[0], Mw, K, bits, bm, kf, ags)
    actk = ags // 4
    d = tap[0] - PS / actk
    assert d.min() >= 0 and d.max() <= np.log2(actk) / 2
    Cdq = orc.dequant_matmul(case["w"], case["sc"], case["zr"], case["B"], bits, gs)[0]
    nmse = np.mean((Cdq - C[0]) ** 2) / np.mean(Cdq ** 2)
    # ActK = 8: mylog2<8>::value / 4 == 0 in the reference's integer arithmetic, i.e. no bias correction (tbl.cc:476)
    assert nmse < (5e-3 if actk == 16 else 0.5)
