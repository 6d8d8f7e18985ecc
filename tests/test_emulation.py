"""CPU emulation of the tiled GEMV kernel's integer path (tmac_amd/csrc/emu.cpp) vs the oracle and the
golden vectors.  The emulation compiles the same tmac_layout.h / tmac_core.h the HIP kernels use, so this
pins the device layout, the nibble recoding and the perm/mqsad arithmetic without a GPU."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def emu():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tmac_amd", "csrc"), "emu"], check=True)
    return C.CDLL(os.path.join(ROOT, "tmac_amd", "lib", "libtmac_emu.so"))


def run_emu(emu, A, q, Mw, K, bits, bm, kf, ags, mode):
    G = 1 if ags == K else K // ags
    PS = np.zeros((Mw * bits, G), np.int32)
    assert emu.emu_partial_sums(orc._p(A), orc._p(np.ascontiguousarray(q)), Mw, K, bits, bm, kf, ags, mode, orc._p(PS)) == 0
    return PS


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("name", sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*.npz"))))
def test_emulation_matches_golden(emu, name, mode):
    d = dict(np.load(os.path.join(GOLD, name + ".npz")))
    Mw, K, bits, bm, kf, gs, ags, zp, mg = [int(x) for x in d["meta"]]
    if mode >= 2 and ags == 32:
        pytest.skip("the fused layout covers act_group 64 / unified scale only")
    PS = run_emu(emu, d["A_ref"], d["qlut"], Mw, K, bits, bm, kf, ags, mode)
    if mg == -1:
        assert np.array_equal(PS, d["PS"])
    else:
        assert np.array_equal(PS[:, 0], d["cbits32"])


@pytest.mark.parametrize("bits,bm,kf,ags,Mw,K", [
    (2, 128, 16, 64, 64, 11008),    # headline K: 172 segments -> ragged last segment block
    (2, 128, 16, 64, 72 * 0 + 192, 1024),
    (4, 256, 16, 64, 64, 4096),
    (2, 128, 16, 3200, 64, 3200),   # BitNet: 50 segments, unified scale
    (1, 128, 16, 64, 128, 1024),
    (3, 192, 16, 32 * 2, 64, 1024),
])
def test_emulation_matches_oracle(emu, bits, bm, kf, ags, Mw, K):
    case = orc.make_case(bits * 100 + K, Mw, K, bits=bits, ags=ags)
    A = orc.preprocess_weights(case["w"], bits, bm, kf)
    q, _, _ = orc.preprocessor(case["B"], ags)
    PSo = orc.partial_sums(A, q[0], Mw, K, bits, bm, kf, ags)
    for mode in (0, 1, 2, 3, 4):
        assert np.array_equal(run_emu(emu, A, q[0], Mw, K, bits, bm, kf, ags, mode), PSo)


def test_extreme_tables(emu):
    """all-(+127)/(-127) tables and all-zero tables: the biased u16 accumulators must not overflow or borrow"""
    Mw, K, bits, bm, kf, ags = 64, 256, 2, 128, 16, 64
    rng = np.random.default_rng(0)
    w = rng.integers(0, 4, (Mw, K), dtype=np.uint8)
    A = orc.preprocess_weights(w, bits, bm, kf)
    for fill in (127, -127, 0):
        q = np.full((K // 4, 16), fill, np.int8)
        q[:, 8:] = -q[:, 7::-1]   # keep the antisymmetry the kernel relies on
        PSo = orc.partial_sums(A, q, Mw, K, bits, bm, kf, ags)
        for mode in (0, 1, 2, 3, 4):
            assert np.array_equal(run_emu(emu, A, q, Mw, K, bits, bm, kf, ags, mode), PSo)


@pytest.mark.parametrize("bits,bm,kf,ags,Mw,K", [
    (2, 128, 16, 64, 64, 11008), (4, 256, 16, 64, 64, 1024), (2, 128, 8, 32, 64, 1024), (3, 192, 16, 64, 64, 512),
    (1, 128, 16, 32, 128, 512),
])
def test_emulation_fast_aggregation(emu, bits, bm, kf, ags, Mw, K):
    """(a9) the halving-adder tree of tmac_core.h (SegAcc<BITS, 2>, v_lerp_u8 modelled on the host) == the oracle's tree,
    both flavours"""
    case = orc.make_case(bits * 10 + ags, Mw, K, bits=bits, ags=ags)
    A = orc.preprocess_weights(case["w"], bits, bm, kf)
    S = orc.preprocess_scales(case["sc"], case["zr"], bits, bm)
    q, ls, lb = orc.preprocessor(case["B"], ags)
    for fa in (1, 2):
        _, tap = orc.qgemm_float_fa(A, q, S, ls, lb, Mw, K, 1, bits, bm, kf, 128, ags, True, fa)
        assert np.array_equal(run_emu(emu, A, q[0], Mw, K, bits, bm, kf, ags, 4 + fa), tap[0])
