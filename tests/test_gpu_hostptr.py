"""The reference-named HOST-pointer entry points (preprocessor_int8 / qgemm_lut_int8, include/tmac_hip.h layer 1:
`/root/reference/include/t-mac/tmac_gemm_wrapper.h:170-228`, generated `deploy/tuned/<set>/kernels.h`) under the
conditions that broke them in round 2.

Round 2's driver run failed `test_host_pointer_cabi_matches_prebuilt_reference` with the lookup term missing from the
output (all-zero half tables): `tmac_hip_workspace_create` filled the LUT image on the null stream, the host-pointer layer
built the LUT on its own NON-BLOCKING stream, and nothing ordered the two.  These tests drop the layer's workspace in a
warm process and take the first-call path again and again against the reference's own golden vector.
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
INI = ("[qgemm_lut_t1_int8_m8192_k4096_n1_b2]\nbm = 128\nsimd_n_in = 16\nsimd_n_out = 8\nkfactor = 16\n"
       "group_size = 128\nlut_scales_size = 64\nscales_size = 262144\nn_tile_num = 64\n")


@pytest.fixture(scope="module")
def tm():
    import torch
    import tmac_amd
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    assert tmac_amd.lib().tmac_hip_device_count() > 0
    return tmac_amd


def rel_err(c, ref):
    return float(np.abs(c.astype(np.float64) - ref.astype(np.float64)).max() / max(np.abs(ref).max(), 1e-30))


def first_call_errors(tm, tmp_path, rounds, fill_sync, busy=None):
    """`rounds` x {fresh library state -> preprocessor_int8 -> the FIRST qgemm_lut_int8 of a tile}; returns the relative error
    of every round against the reference's prebuilt-kernel output"""
    d = dict(np.load(os.path.join(GOLD, "prebuilt_llama2_7b_w2_k4096.npz")))
    L = tm.lib()
    ini = tmp_path / "kcfg.ini"
    ini.write_text(INI)
    K = 4096
    B = np.ascontiguousarray(d["B"][0])
    A = np.ascontiguousarray(d["A_ref"][0]); S = np.ascontiguousarray(d["S_ref"][0])
    vp = lambda a: C.c_void_p(a.ctypes.data)
    errs = []
    for it in range(rounds):
        tm.binding.check(L.tmac_hip_reset_state())            # drops the layer's workspace: the next call creates and fills it
        tm.binding.check(L.tmac_hip_debug_ws_fill_sync(fill_sync))
        tm.binding.check(L.tmac_hip_load_kcfg_ex(str(ini).encode(), 1))
        if busy is not None:
            busy()                                           # null-stream work in flight when the workspace is created
        ls = np.zeros(64, np.float32); lb = np.zeros(64, np.float32); q = np.zeros((K // 4, 16), np.int8)
        assert L.preprocessor_int8(8192, K, 1, 2, vp(B), vp(ls), vp(lb), vp(q)) == 0, L.tmac_hip_last_error()
        assert np.array_equal(q, d["qlut"].reshape(q.shape))
        assert np.array_equal(ls.view(np.uint32), d["lut_scales"].view(np.uint32).reshape(-1))
        assert np.array_equal(lb.view(np.uint32), d["lut_biases"].view(np.uint32).reshape(-1))
        c = np.zeros(64, np.float32)
        assert L.qgemm_lut_int8(128, K, 1, 2, vp(A), vp(q), vp(S), vp(ls), vp(lb), vp(c)) == 0, L.tmac_hip_last_error()
        errs.append(rel_err(c, d["C"].reshape(-1)))
    L.tmac_hip_reset_state()
    return errs


def _null_stream_load():
    import torch
    filler = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")

    def busy():
        for _ in range(4):
            filler.fill_(1)                                  # legacy null stream: what the workspace fills queue behind
    return busy


def test_first_call_after_workspace_drop(tm, tmp_path):
    """200 first calls in a warm process, with null-stream work queued in front of the workspace creation every time"""
    errs = first_call_errors(tm, tmp_path, 200, 1, _null_stream_load())
    assert max(errs) <= 2e-5, (max(errs), int(np.sum(np.array(errs) > 2e-5)))


def test_round2_race_is_the_fill_ordering(tm, tmp_path, record_property):
    """Informational: the same loop with the ordering synchronisation switched off (tmac_hip_debug_ws_fill_sync(0)) -- the
    round-2 library.  Records how many of the first calls come back without the lookup term; never fails on that count
    (a race need not fire), but the ordered run right after it must be clean."""
    busy = _null_stream_load()
    errs = first_call_errors(tm, tmp_path, 100, 0, busy)
    bad = int(np.sum(np.array(errs) > 2e-5))
    record_property("unordered_fill_wrong_results", bad)
    print(f"unordered fills: {bad} of {len(errs)} first calls wrong (max rel err {max(errs):.3g})")
    errs = first_call_errors(tm, tmp_path, 50, 1, busy)
    assert max(errs) <= 2e-5


def test_conflicting_kcfg_sections_are_refused(tm, tmp_path):
    """two loaded sections with the same (bm, k, n, b) key that disagree on the quantisation layout: the per-tile entry
    point has no way to tell which one the bytes follow -> -1 (the reference compiles ONE kernel per such name,
    deploy/compile.py:52-71), and a replacing load resolves it"""
    L = tm.lib()
    a = tmp_path / "a.ini"; a.write_text(INI)
    b = tmp_path / "b.ini"
    b.write_text("[qgemm_lut_t1_int8_m2048_k4096_n1_b2]\nbm = 128\nsimd_n_in = 16\nsimd_n_out = 8\nkfactor = 16\n"
                 "group_size = 128\nlut_scales_size = 64\nscales_size = 32768\nn_tile_num = 16\n")     # no zero points
    tm.binding.check(L.tmac_hip_load_kcfg(str(a).encode()))
    tm.binding.check(L.tmac_hip_load_kcfg(str(b).encode()))
    d = dict(np.load(os.path.join(GOLD, "prebuilt_llama2_7b_w2_k4096.npz")))
    A = np.ascontiguousarray(d["A_ref"][0]); S = np.ascontiguousarray(d["S_ref"][0])
    q = np.ascontiguousarray(d["qlut"]); ls = np.ascontiguousarray(d["lut_scales"]).reshape(-1); lb = np.ascontiguousarray(d["lut_biases"]).reshape(-1)
    c = np.zeros(64, np.float32)
    vp = lambda x: C.c_void_p(x.ctypes.data)
    assert L.qgemm_lut_int8(128, 4096, 1, 2, vp(A), vp(q), vp(S), vp(ls), vp(lb), vp(c)) == -1
    assert b"disagree" in L.tmac_hip_last_error()
    tm.binding.check(L.tmac_hip_load_kcfg_ex(str(a).encode(), 1))
    assert L.qgemm_lut_int8(128, 4096, 1, 2, vp(A), vp(q), vp(S), vp(ls), vp(lb), vp(c)) == 0, L.tmac_hip_last_error()
    assert rel_err(c, d["C"].reshape(-1)) <= 2e-5
