"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE — see oracle/tmac_oracle.c header).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  Nothing under ``tmac_amd/`` does.

Two libraries live here:

* ``liboracle.so``   — our scalar C restatement (tmac_oracle.c), always buildable (gcc only).
* ``_ref/*.so``      — the reference's own sources compiled from /root/reference
  (``make ref``; exists in the build container, travels to the GPU box as a prebuilt file).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = "/root/reference"

_lib = None
_ref_libs = {}


def build(with_ref: Optional[bool] = None) -> None:
    """Compile liboracle.so (and _ref/ when /root/reference is present)."""
    subprocess.run(["make", "-s", "-C", HERE], check=True)
    if with_ref is None:
        with_ref = os.path.isdir(REF_ROOT)
    if with_ref:
        subprocess.run(["make", "-s", "-C", HERE, "ref"], check=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build(with_ref=False)
        _lib = C.CDLL(path)
    return _lib


def have_ref(name: str = "intrins") -> bool:
    return os.path.exists(os.path.join(HERE, "_ref", f"libtmac_ref_{name}.so"))


def ref_lib(name: str = "intrins") -> C.CDLL:
    """name: 'intrins' or a prebuilt set such as 'aarch64-llama-2-7b-2bit'."""
    if name not in _ref_libs:
        path = os.path.join(HERE, "_ref", f"libtmac_ref_{name}.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path + " (run `make -C oracle ref` where /root/reference exists)")
        _ref_libs[name] = C.CDLL(path)
    return _ref_libs[name]


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


# ----------------------------------------------------------------------------------------
# scalar restatement
# ----------------------------------------------------------------------------------------

def preprocessor(B: np.ndarray, ags: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """B [N][K] float32 -> (qlut int8 [N][K/4][16], lut_scales [N][K/ags], lut_biases)."""
    B = _c(np.atleast_2d(B), np.float32)
    N, K = B.shape
    ls = np.zeros((N, K // ags), np.float32)
    lb = np.zeros((N, K // ags), np.float32)
    q = np.zeros((N, K // 4, 16), np.int8)
    rc = lib().oracle_preprocessor(_p(B), N, K, ags, _p(ls), _p(lb), _p(q))
    if rc != 0:
        raise ValueError(f"oracle_preprocessor rc={rc}")
    return q, ls, lb


def preprocess_weights(w: np.ndarray, bits: int, bm: int, kfactor: int) -> np.ndarray:
    """uint8 [Mw][K] -> reference-layout bytes [M/bm][K/4][bm/2] (weights.py:5-73)."""
    w = _c(w, np.uint8)
    Mw, K = w.shape
    M = Mw * bits
    A = np.zeros((M // bm, K // 4, bm // 2), np.uint8)
    rc = lib().oracle_preprocess_weights(_p(w), Mw, K, bits, bm, kfactor, _p(A))
    if rc != 0:
        raise ValueError(f"oracle_preprocess_weights rc={rc}")
    return A


def preprocess_scales(sc: np.ndarray, zr: Optional[np.ndarray], bits: int, bm: int) -> np.ndarray:
    """[Mw][K/gs] (+zeros) -> [M/bm][K/gs][bm/bits*(2|1)] float32 (weights.py:75-84)."""
    sc = _c(sc, np.float32)
    Mw, SG = sc.shape
    M = Mw * bits
    zp = zr is not None
    out = np.zeros((M // bm, SG, bm // bits * (2 if zp else 1)), np.float32)
    zr_c = _c(zr, np.float32) if zp else None
    # K only enters through K/gs: pass gs=1, K=SG
    rc = lib().oracle_preprocess_scales(_p(sc), _p(zr_c) if zp else None, Mw, SG, bits, bm, 1, _p(out))
    if rc != 0:
        raise ValueError(f"oracle_preprocess_scales rc={rc}")
    return out


def partial_sums(A: np.ndarray, qlut_row: np.ndarray, Mw: int, K: int, bits: int, bm: int,
                 kfactor: int, ags: int) -> np.ndarray:
    """int32 [M][K/ags] in M-space (bit-plane) row order, for ONE activation row."""
    A = _c(A, np.uint8)
    q = _c(qlut_row, np.int8)
    PS = np.zeros((Mw * bits, K // ags), np.int32)
    rc = lib().oracle_partial_sums(_p(A), _p(q), Mw, K, bits, bm, kfactor, ags, _p(PS))
    if rc != 0:
        raise ValueError(f"oracle_partial_sums rc={rc}")
    return PS


def qgemm_float(A, qlut, scales, ls, lb, Mw, K, N, bits, bm, kfactor, gs, ags, zero_point,
                one_scale=False) -> np.ndarray:
    A = _c(A, np.uint8); qlut = _c(qlut, np.int8); scales = _c(scales, np.float32)
    ls = _c(ls, np.float32); lb = _c(lb, np.float32)
    Cout = np.zeros((N, Mw), np.float32)
    rc = lib().oracle_qgemm_float(_p(A), _p(qlut), _p(scales), _p(ls), _p(lb), _p(Cout), Mw, K, N,
                                  bits, bm, kfactor, gs, ags, int(zero_point), int(one_scale))
    if rc != 0:
        raise ValueError(f"oracle_qgemm_float rc={rc}")
    return Cout


def qgemm_float_fa(A, qlut, scales, ls, lb, Mw, K, N, bits, bm, kfactor, gs, ags, zero_point, fa_mode):
    """(a9) fast-aggregation flavour of the float path.  fa_mode 1: NEON signed rounding-halving adds,
    2: what the reference's AVX2 build computes (tbl.cc:201-256).  Returns (C [N][Mw], tree results
    int32 [N][M][K/ags])."""
    A = _c(A, np.uint8); qlut = _c(qlut, np.int8); scales = _c(scales, np.float32)
    ls = _c(ls, np.float32); lb = _c(lb, np.float32)
    Cout = np.zeros((N, Mw), np.float32)
    tap = np.zeros((N, Mw * bits, K // ags), np.int32)
    rc = lib().oracle_qgemm_float_fa(_p(A), _p(qlut), _p(scales), _p(ls), _p(lb), _p(Cout), _p(tap), Mw, K, N,
                                     bits, bm, kfactor, gs, ags, int(zero_point), int(fa_mode))
    if rc != 0:
        raise ValueError(f"oracle_qgemm_float_fa rc={rc}")
    return Cout, tap


def qgemm_scale_final(A, qlut, scales, ls, lb, Mw, K, N, bits, bm, kfactor, m_groups=1):
    A = _c(A, np.uint8); qlut = _c(qlut, np.int8); scales = _c(scales, np.float32)
    ls = _c(ls, np.float32); lb = _c(lb, np.float32)
    Cout = np.zeros((N, Mw), np.float32)
    cb = np.zeros((N, Mw * bits), np.int32)
    rc = lib().oracle_qgemm_scale_final(_p(A), _p(qlut), _p(scales), _p(ls), _p(lb), _p(Cout), _p(cb),
                                        Mw, K, N, bits, bm, kfactor, m_groups)
    if rc != 0:
        raise ValueError(f"oracle_qgemm_scale_final rc={rc}")
    return Cout, cb


def dequant_matmul(w, sc, zr, B, bits, gs, m_groups=-1) -> np.ndarray:
    w = _c(w, np.uint8); sc = _c(sc, np.float32); B = _c(np.atleast_2d(B), np.float32)
    Mw, K = w.shape
    N = B.shape[0]
    zr_c = _c(zr, np.float32) if zr is not None else None
    Cout = np.zeros((N, Mw), np.float64)
    lib().oracle_dequant_matmul(_p(w), _p(sc), _p(zr_c) if zr is not None else None, _p(B), _p(Cout),
                                Mw, K, N, bits, gs, m_groups)
    return Cout


def m_to_out_perm(Mw: int, bits: int) -> np.ndarray:
    """index array r[o, p] = M-space row of (output row o, plane p)  (weights.py:65)."""
    o = np.arange(Mw)[:, None]
    p = np.arange(bits)[None, :]
    return (o // 8) * 8 * bits + p * 8 + (o % 8)


# ----------------------------------------------------------------------------------------
# the reference itself (oracle/_ref)
# ----------------------------------------------------------------------------------------

def ref_preprocessor(B_row: np.ndarray, ags: int):
    B_row = _c(B_row, np.float32).reshape(-1)
    K = B_row.size
    ls = np.zeros(K // ags, np.float32); lb = np.zeros(K // ags, np.float32)
    q = np.zeros((K // 4, 16), np.int8)
    rc = ref_lib().ref_preprocessor(K, ags, _p(B_row), _p(ls), _p(lb), _p(q))
    assert rc == 0
    return q, ls, lb


def ref_cbits_float(A, qlut_row, scales_t, ls, lb, Mw, K, bits, bm, kfactor, gs, ags, zero_point, fa=False):
    """CBits fp32 [M] from the reference's tbl intrinsic, tile by tile (fa: its FastAggregation = true
    instantiation, AVX2 flavour)."""
    A = _c(A, np.uint8); q = _c(qlut_row, np.int8); S = _c(scales_t, np.float32)
    ls = _c(ls, np.float32); lb = _c(lb, np.float32)
    M = Mw * bits
    out = np.zeros(M, np.float32)
    L = ref_lib()
    for tile in range(M // bm):
        cb = np.zeros(bm, np.float32)
        fn = L.ref_tile_cbits_float_fa if fa else L.ref_tile_cbits_float
        rc = fn(bits, kfactor, ags, int(zero_point), bm, K, gs, _p(A[tile]), _p(q), _p(S[tile]), _p(ls), _p(lb), _p(cb))
        if rc != 0:
            raise ValueError("no reference instantiation for this configuration")
        out[tile * bm:(tile + 1) * bm] = cb
    return out


def ref_partial_sums(A, qlut_row, Mw, K, bits, bm, kfactor, ags) -> np.ndarray:
    A = _c(A, np.uint8); q = _c(qlut_row, np.int8)
    M = Mw * bits
    G = K // ags
    out = np.zeros((M, G), np.int32)
    L = ref_lib()
    for tile in range(M // bm):
        ps = np.zeros((bm, G), np.int32)
        rc = L.ref_tile_partial_sums(bits, kfactor, ags, bm, K, _p(A[tile]), _p(q), _p(ps))
        if rc != 0:
            raise ValueError("no reference instantiation for this configuration")
        out[tile * bm:(tile + 1) * bm] = ps
    return out


def ref_cbits_int32(A, qlut_row, Mw, K, bits, bm, kfactor) -> np.ndarray:
    A = _c(A, np.uint8); q = _c(qlut_row, np.int8)
    M = Mw * bits
    out = np.zeros(M, np.int32)
    L = ref_lib()
    for tile in range(M // bm):
        cb = np.zeros(bm, np.int32)
        rc = L.ref_tile_cbits_int32(bits, kfactor, bm, K, _p(A[tile]), _p(q), _p(cb))
        if rc != 0:
            raise ValueError("no reference instantiation for this configuration")
        out[tile * bm:(tile + 1) * bm] = cb
    return out


def combine_planes(cbits: np.ndarray, Mw: int, bits: int) -> np.ndarray:
    """C[o] = sum_p float32(CBits[r(o,p)]) * alpha_p, fp32 left-to-right (kernels.cc:1068)."""
    alphas = np.array([0.5, 1.0, 2.0, 4.0], np.float32)
    idx = m_to_out_perm(Mw, bits)
    cb = cbits.astype(np.float32)
    acc = cb[idx[:, 0]] * alphas[0]
    for p in range(1, bits):
        acc = (acc + cb[idx[:, p]] * alphas[p]).astype(np.float32)
    return acc


# ----------------------------------------------------------------------------------------
# synthetic cases (mirrors tests/test_e2e.py:57-65 with fixed seeds)
# ----------------------------------------------------------------------------------------

def make_case(seed: int, Mw: int, K: int, N: int = 1, bits: int = 2, gs: int = 128, ags: int = 64,
              zero_point: bool = True, m_groups: int = -1, fp16_values: bool = False):
    """Returns dict(w, sc, zr, B) of numpy arrays.  With fp16_values the float tensors are
    rounded to fp16-representable values (so an fp16-storing device path sees identical data)."""
    rng = np.random.default_rng(seed)
    w = rng.integers(0, 2 ** bits, size=(Mw, K), dtype=np.uint8)
    if m_groups == -1:
        sc = np.abs(rng.standard_normal((Mw, K // gs))).astype(np.float32)
        zr = rng.standard_normal((Mw, K // gs)).astype(np.float32) if zero_point else None
    else:
        sc = np.abs(rng.standard_normal((m_groups,))).astype(np.float32)
        zr = None
    B = rng.standard_normal((N, K)).astype(np.float32)
    if fp16_values:
        sc = sc.astype(np.float16).astype(np.float32)
        if zr is not None:
            zr = zr.astype(np.float16).astype(np.float32)
        B = B.astype(np.float16).astype(np.float32)
    return dict(w=w, sc=sc, zr=zr, B=B)
