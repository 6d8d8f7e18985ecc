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
