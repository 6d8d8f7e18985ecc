"""ctypes binding of libtmac_hip.so (C-ABI: include/tmac_hip.h).

The library is built IN-TREE (``tmac_amd/lib/libtmac_hip.so``, ``make -C tmac_amd/csrc``) so that
it travels with the repo snapshot to the GPU box.  There is deliberately no fallback: if the
library is missing, or no HIP device is visible when a compute entry point is called, an exception
is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
F32, F16 = 0, 1


class TMACHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"tmac_hip error {code}: {msg}")
        self.code = code


class XForm(C.Structure):
    """tmac_hip_xform (include/tmac_hip.h): a vector transform of the next recorded call's activations"""
    _fields_ = [("kind", C.c_int32), ("in2", C.c_void_p), ("residual", C.c_void_p), ("gamma", C.c_void_p), ("eps", C.c_float),
                ("residual_out", C.c_void_p), ("keep", C.c_int32)]


class KCfg(C.Structure):
    """tmac_kcfg — TMAC::TMACGeMMConfig (tmac_gemm_wrapper.h:26-35) + zero_point/act_group_size/m_groups."""
    _fields_ = [(n, C.c_int) for n in ("bm", "simd_n_in", "simd_n_out", "kfactor", "group_size", "lut_scales_size",
                                       "scales_size", "n_tile_num", "act_group_size", "zero_point", "m_groups")]

    @classmethod
    def make(cls, Mw, K, bits, bm, kfactor=16, group_size=128, act_group_size=64, zero_point=True, m_groups=-1, N=1):
        per = 2 if zero_point else 1
        scales_size = m_groups if m_groups >= 1 else Mw * (K // group_size) * per
        return cls(bm, 16, 8, kfactor, group_size, N * K // act_group_size, scales_size, Mw * bits // bm,
                   act_group_size, int(bool(zero_point)) if m_groups < 1 else 0, m_groups)


def lib_path() -> str:
    # $TMAC_HIP_LIB: another build of the same library (A/B measurements of two builds in one run; tools/gpu/)
    return os.environ.get("TMAC_HIP_LIB") or os.path.join(HERE, "lib", "libtmac_hip.so")


def build_library(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of the in-tree shared library (cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", os.path.join(HERE, "csrc"), "clean"], check=True)
    subprocess.run(["make", "-s", "-j4", "-C", os.path.join(HERE, "csrc")], check=True)
    return lib_path()


_lib = None


def load_library() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise ImportError(f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); tmac_amd has no fallback path")
    L = C.CDLL(p)
    vp, i32, sz = C.c_void_p, C.c_int32, C.c_size_t
    sigs = {
        "tmac_hip_init": ([C.c_int], i32),
        "tmac_hip_last_error": ([], C.c_char_p),
        "tmac_hip_version": ([], C.c_char_p),
        "tmac_hip_device_count": ([], i32),
        "tmac_hip_pointer_on_device": ([vp], i32),
        "tmac_hip_load_kcfg": ([C.c_char_p], i32),
        "tmac_hip_load_kcfg_ex": ([C.c_char_p, C.c_int], i32),
        "tmac_hip_clear_kcfg": ([], i32),
        "tmac_hip_reset_state": ([], i32),
        "tmac_hip_debug_ws_fill_sync": ([C.c_int], i32),
        "tmac_hip_debug_host_runs": ([C.c_int], i32),
        "tmac_hip_get_kcfg": ([C.c_int] * 4 + [C.POINTER(KCfg)], i32),
        "tmac_hip_set_kcfg": ([C.c_int] * 4 + [C.POINTER(KCfg)], i32),
        "tmac_hip_register_weights": ([C.POINTER(vp), vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(KCfg), C.c_int, C.c_int, vp], i32),
        "tmac_hip_register_weights_dev": ([C.POINTER(vp), vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(KCfg), C.c_int, C.c_int, vp], i32),
        "tmac_hip_free_weights": ([vp], i32),
        "tmac_hip_weights_bytes": ([vp], sz),
        "tmac_hip_workspace_create": ([C.POINTER(vp), C.c_int, C.c_int], i32),
        "tmac_hip_workspace_free": ([vp], i32),
        "tmac_hip_preprocessor_dev": ([vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp], i32),
        "tmac_hip_qgemm_dev": ([vp, vp, vp, C.c_int, C.c_int, vp], i32),
        "tmac_hip_qgemm_fused_dev": ([C.POINTER(vp), C.c_int, vp, C.c_int, C.POINTER(vp), C.c_int, C.c_int, vp], i32),
        "tmac_hip_qgemm_fused_partial_sums": ([vp, vp, C.c_int, vp, vp, vp, C.c_int, vp], i32),
        "tmac_hip_workspace_ptrs": ([vp, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(vp)], i32),
        "tmac_hip_workspace_read": ([vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp], i32),
        "tmac_hip_workspace_write": ([vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp], i32),
        "tmac_hip_qgemm_partial_sums": ([vp, vp, vp, C.c_int, vp], i32),
        "tmac_hip_debug_gemm_kernel": ([C.c_int], i32),
        "tmac_hip_debug_gemm_stamps": ([vp], i32),
        "tmac_hip_debug_gemm_comb_sums": ([vp, vp, vp, C.c_int, vp], i32),
        "tmac_hip_debug_gemm_image_read": ([vp, vp, vp, vp, vp, C.c_int, vp], i32),
        "tmac_hip_set_variant": ([C.c_int], i32),
        "tmac_hip_set_gemm_min_n": ([C.c_int], i32),
        "tmac_hip_set_fast_aggregation": ([C.c_int], i32),
        "tmac_hip_debug_stream_read": ([vp, sz, vp, vp], i32),
        "tmac_hip_autotune_fused": ([C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                     C.POINTER(C.c_float), C.POINTER(C.c_float)], i32),
        "tmac_hip_tune_save": ([C.c_char_p], i32),
        "tmac_hip_tune_load": ([C.c_char_p], i32),
        "tmac_hip_tune_clear": ([], i32),
        "tmac_hip_comm_unique_id": ([vp], i32),
        "tmac_hip_comm_init": ([C.POINTER(vp), vp, C.c_int, C.c_int], i32),
        "tmac_hip_comm_allgather": ([vp, vp, vp, sz, vp], i32),
        "tmac_hip_comm_destroy": ([vp], i32),
        "tmac_hip_comm_init_ipc": ([C.POINTER(vp), sz, C.c_int, C.c_int], i32),
        "tmac_hip_comm_export": ([vp, vp], i32),
        "tmac_hip_comm_connect": ([vp, vp, C.c_int], i32),
        "tmac_hip_comm_status": ([vp, C.POINTER(C.c_uint32)], i32),
        "tmac_hip_comm_last_error": ([], C.c_char_p),
        "tmac_hip_defer": ([C.c_int], i32),
        "tmac_hip_flush": ([vp], i32),
        "tmac_hip_defer_stats": ([C.POINTER(C.c_uint64)] * 4, i32),
        "tmac_hip_chain_begin": ([], i32),
        "tmac_hip_chain_end": ([C.POINTER(vp)], i32),
        "tmac_hip_chain_launch": ([vp, vp], i32),
        "tmac_hip_chain_status": ([vp, C.POINTER(C.c_uint32)], i32),
        "tmac_hip_chain_info": ([vp, C.c_int, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(sz)], i32),
        "tmac_hip_chain_xform": ([C.POINTER(XForm)], i32),
        "tmac_hip_chain_free": ([vp], i32),
        "tmac_hip_chain_set_stamps": ([vp, vp], i32),
        "tmac_hip_chain_set_tap": ([vp, vp], i32),
        "tmac_hip_chain_tap_layout": ([vp, C.c_int, C.POINTER(sz), C.POINTER(sz)], i32),
        "tmac_hip_chain_threads": ([], i32),
        "tmac_hip_chain_is_stream": ([vp], i32),
        "tmac_hip_chain_record_gather": ([vp, vp, sz, C.c_int, C.c_int], i32),
        "tmac_hip_chain_export": ([vp, vp], i32),
        "tmac_hip_chain_connect": ([vp, vp, C.c_int], i32),
        "tmac_hip_debug_chain_grid": ([C.c_int], i32),
        "tmac_hip_debug_stream_schedule": ([C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)], i32),
        "tmac_hip_debug_chain_config": ([C.c_int, C.c_uint], i32),
        "tmac_hip_debug_quad_config": ([C.c_int, C.c_int], i32),
        "tmac_hip_selftest": ([vp, vp, C.c_int], i32),
        "tmac_hip_selftest_mfma": ([vp, vp], i32),
        "tmac_hip_cache_clear": ([], i32),
        "qgemm_lut_int8": ([C.c_int] * 4 + [vp] * 6, i32),
        "preprocessor_int8": ([C.c_int] * 4 + [vp] * 4, i32),
    }
    # $TMAC_HIP_LIB may name an OLDER build for an A/B run (tools/gpu): entry points it lacks stay unbound (calling one raises)
    optional = {"tmac_hip_chain_is_stream", "tmac_hip_chain_xform", "tmac_hip_comm_init_ipc", "tmac_hip_comm_export", "tmac_hip_comm_connect", "tmac_hip_comm_status"} if os.environ.get("TMAC_HIP_LIB") else set()
    for name, (argt, rest) in sigs.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            if name in optional:
                continue
            raise
        fn.argtypes, fn.restype = argt, rest
    _lib = L
    return L


def lib() -> C.CDLL:
    return load_library()


def check(rc: int) -> None:
    if rc != 0:
        raise TMACHipError(rc, load_library().tmac_hip_last_error().decode())


def check_count(rc: int) -> int:
    """for entry points that return a count (>= 0) or an error code (< 0)"""
    if rc < 0:
        raise TMACHipError(rc, load_library().tmac_hip_last_error().decode())
    return rc
