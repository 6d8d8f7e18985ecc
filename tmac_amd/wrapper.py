"""Python mirror of ``TMAC::TMACGeMMWrapper`` (include/t-mac/tmac_gemm_wrapper.h:79-347) on the GPU.

Same method names and argument meaning as the reference's C++ wrapper — ``set_workspace``,
``get_kcfg``, ``llama_cpp_init`` (preprocessor), ``llama_cpp_compute`` (qgemm_lut) — with device
pointers instead of host pointers, and weights registered once (:class:`Weights`) instead of passed
as raw tile pointers on every call.  All compute goes through libtmac_hip.so's C-ABI.

PyTorch is used only as the owner of device memory / streams when the caller passes tensors; raw
integer device pointers work as well.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import binding as B
from .binding import KCfg, F16, F32, check, check_count


def _ptr(x) -> int:
    """device pointer of a torch tensor / raw int; host pointer of a numpy array"""
    if x is None:
        return 0
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    return x.data_ptr()


def _stream(stream=None) -> int:
    if stream is not None:
        return stream if isinstance(stream, int) else stream.cuda_stream
    try:
        import torch
        if torch.cuda.is_available():
            return torch.cuda.current_stream().cuda_stream
    except Exception:
        pass
    return 0


def _dtype_code(t) -> int:
    import torch
    if t.dtype == torch.float16:
        return F16
    if t.dtype == torch.float32:
        return F32
    raise TypeError(f"unsupported dtype {t.dtype}")


class Weights:
    """One weight matrix resident on the GPU (tmac_hip_register_weights)."""

    def __init__(self, A_ref, scales_ref, Mw: int, K: int, bits: int, cfg: KCfg, scales_dtype=F32, dev_dtype=F32,
                 on_device: bool = False, stream=None):
        self.Mw, self.K, self.bits, self.cfg = Mw, K, bits, cfg
        self._h = C.c_void_p()
        L = B.lib()
        fn = L.tmac_hip_register_weights_dev if on_device else L.tmac_hip_register_weights
        if not on_device:
            A_ref = np.ascontiguousarray(A_ref, np.uint8)
            scales_ref = np.ascontiguousarray(scales_ref, np.float16 if scales_dtype == F16 else np.float32)
        self._keep = (A_ref, scales_ref)
        check(fn(C.byref(self._h), _ptr(A_ref), _ptr(scales_ref), Mw, K, bits, C.byref(cfg), scales_dtype, dev_dtype,
                 _stream(stream)))
        self._keep = None
        self._caches = []          # fused-call caches holding this handle (TMACGeMMWrapper.fused)

    @property
    def handle(self):
        return self._h

    def algorithmic_bytes(self) -> int:
        return int(B.lib().tmac_hip_weights_bytes(self._h))

    def free(self):
        if self._h:
            hv = self._h.value
            for cache in getattr(self, "_caches", []):     # no cached call may keep the freed native handle
                for key in [k for k in cache if hv in k[0]]:
                    del cache[key]
            self._caches = []
            B.lib().tmac_hip_free_weights(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Workspace:
    """QLUT / LUT_Scales / LUT_Biases on the device (TMACGeMMWrapper::set_workspace)."""

    def __init__(self, maxK: int, maxN: int = 1):
        self.maxK, self.maxN = maxK, maxN
        self._h = C.c_void_p()
        check(B.lib().tmac_hip_workspace_create(C.byref(self._h), maxK, maxN))

    @property
    def handle(self):
        return self._h

    def ptrs(self):
        q, ls, lb, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_size_t()
        check(B.lib().tmac_hip_workspace_ptrs(self._h, C.byref(q), C.byref(n), C.byref(ls), C.byref(lb)))
        return q.value, n.value, ls.value, lb.value

    def read(self, K: int, N: int, act_group_size: int, stream=None):
        q = np.zeros((N, K // 4, 16), np.int8)
        ls = np.zeros((N, K // act_group_size), np.float32)
        lb = np.zeros((N, K // act_group_size), np.float32)
        check(B.lib().tmac_hip_workspace_read(self._h, q.ctypes.data, ls.ctypes.data, lb.ctypes.data, K, N,
                                              act_group_size, _stream(stream)))
        return q, ls, lb

    def read_gemm_image(self, K: int, N: int, act_group_size: int = 64, stream=None):
        """The LUT image k_gemm_planes streams, in plain layouts: half tables int8 [N][K/4][8], lut_scales, lut_biases,
        entry sums fp32 [N][K/act_group_size] (act_group_size 64, or K for the unified-scale flavour)."""
        G = K // act_group_size
        h = np.zeros((N, K // 4, 8), np.int8)
        ls = np.zeros((N, G), np.float32)
        lb = np.zeros((N, G), np.float32)
        hs = np.zeros((N, G), np.float32)
        check(B.lib().tmac_hip_debug_gemm_image_read(self._h, h.ctypes.data, ls.ctypes.data, lb.ctypes.data, hs.ctypes.data, N,
                                                     _stream(stream)))
        return h, ls, lb, hs

    def write(self, qlut: np.ndarray, lut_scales: np.ndarray, lut_biases: np.ndarray, act_group_size: int, stream=None):
        q = np.ascontiguousarray(qlut, np.int8)
        ls = np.ascontiguousarray(lut_scales, np.float32)
        lb = np.ascontiguousarray(lut_biases, np.float32)
        N, T, _ = q.shape
        check(B.lib().tmac_hip_workspace_write(self._h, q.ctypes.data, ls.ctypes.data, lb.ctypes.data, T * 4, N,
                                               act_group_size, _stream(stream)))

    def free(self):
        if self._h:
            B.lib().tmac_hip_workspace_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Comm:
    """The exchange step of the row-sharded multi-GPU path through the C-ABI (tmac_hip_comm_*: RCCL over xGMI, one process
    per GPU).  ``Comm.unique_id()`` on rank 0, hand the 128 bytes to every rank, ``Comm(id, rank, world)`` on each."""

    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_char * Comm.ID_BYTES)()
        rc = B.lib().tmac_hip_comm_unique_id(buf)
        if rc:
            raise B.TMACHipError(rc, B.lib().tmac_hip_comm_last_error().decode())
        return bytes(buf)

    def __init__(self, uid: bytes, rank: int, world: int):
        self._h = C.c_void_p()
        self.rank, self.world = rank, world
        buf = (C.c_char * Comm.ID_BYTES).from_buffer_copy(uid)
        rc = B.lib().tmac_hip_comm_init(C.byref(self._h), buf, rank, world)
        if rc:
            raise B.TMACHipError(rc, B.lib().tmac_hip_comm_last_error().decode())

    BLOB_BYTES = 128

    @classmethod
    def ipc(cls, max_bytes_per_rank: int, rank: int, world: int) -> "Comm":
        """the same exchange step over IPC-mapped windows instead of RCCL (tmac_hip_comm_init_ipc): ``export()`` on every rank,
        all-gather the blobs over any transport, ``connect(blobs)``"""
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        self.rank, self.world = rank, world
        rc = B.lib().tmac_hip_comm_init_ipc(C.byref(self._h), max_bytes_per_rank, rank, world)
        if rc:
            raise B.TMACHipError(rc, B.lib().tmac_hip_comm_last_error().decode())
        return self

    def export(self) -> bytes:
        buf = C.create_string_buffer(self.BLOB_BYTES)
        rc = B.lib().tmac_hip_comm_export(self._h, buf)
        if rc:
            raise B.TMACHipError(rc, B.lib().tmac_hip_comm_last_error().decode())
        return buf.raw

    def connect(self, blobs) -> None:
        raw = b"".join(blobs) if not isinstance(blobs, (bytes, bytearray)) else bytes(blobs)
        rc = B.lib().tmac_hip_comm_connect(self._h, raw, len(raw) // self.BLOB_BYTES)
        if rc:
            raise B.TMACHipError(rc, B.lib().tmac_hip_comm_last_error().decode())

    def status(self) -> int:
        w = C.c_uint32(0)
        rc = B.lib().tmac_hip_comm_status(self._h, C.byref(w))
        if rc:
            raise B.TMACHipError(rc, B.lib().tmac_hip_comm_last_error().decode())
        return w.value

    def allgather(self, send, recv, nbytes_per_rank: int, stream=None) -> None:
        rc = B.lib().tmac_hip_comm_allgather(self._h, _ptr(send), _ptr(recv), nbytes_per_rank, _stream(stream))
        if rc:
            raise B.TMACHipError(rc, B.lib().tmac_hip_comm_last_error().decode())

    def destroy(self):
        if self._h:
            B.lib().tmac_hip_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class DecodeChain:
    """A recorded sequence of ``TMACGeMMWrapper.fused`` calls (N = 1) executed by ONE persistent kernel launch
    (tmac_hip_chain_*, include/tmac_hip.h).  Built by ``TMACGeMMWrapper.record_chain``."""

    def __init__(self, handle, keep):
        self._h = handle
        self._keep = keep          # weights / tensors the chain points into
        n, g, b = C.c_int32(0), C.c_int32(0), C.c_size_t(0)
        check(B.lib().tmac_hip_chain_info(self._h, 0, C.byref(n), None, C.byref(g), C.byref(b)))
        self.nops, self.grid, self.weight_bytes = n.value, g.value, b.value
        self.threads = int(B.lib().tmac_hip_chain_threads())     # threads per workgroup of k_decode_chain (for A/B against k_gemv_quad)
        is_stream = getattr(B.lib(), "tmac_hip_chain_is_stream", None)
        # stream mode: no recorded call consumes another's output -- tables prebuilt by k_lut_images, lookups by k_gemv_stream (include/tmac_hip.h)
        mode = int(is_stream(self._h)) if is_stream is not None and getattr(is_stream, "argtypes", None) else 0
        self.stream = mode != 0
        self.quarter_walk = mode == 2      # k_gemv_stream's quarter-walk form: per-group-scale outputs to the oracle's tolerance, not bit-identical to stand-alone launches

    @property
    def handle(self):
        return self._h

    BLOB_BYTES = 128       # TMAC_HIP_CHAIN_BLOB_BYTES

    def export(self) -> bytes:
        """row-sharded chains: this rank's hand-off arena as an IPC blob; all-gather the blobs of all ranks (rank order, any
        transport) and hand them to :meth:`connect`"""
        buf = C.create_string_buffer(self.BLOB_BYTES)
        check(B.lib().tmac_hip_chain_export(self._h, buf))
        return buf.raw

    def connect(self, blobs) -> None:
        """blobs: the exported blobs of ALL ranks in rank order (list of bytes, or one bytes object)"""
        raw = b"".join(blobs) if not isinstance(blobs, (bytes, bytearray)) else bytes(blobs)
        world = len(raw) // self.BLOB_BYTES
        check(B.lib().tmac_hip_chain_connect(self._h, raw, world))

    def launch(self, stream=None) -> None:
        check(B.lib().tmac_hip_chain_launch(self._h, _stream(stream)))

    def status(self) -> int:
        """after a stream synchronisation: 0 if every in-kernel hand-off of the last launches completed"""
        w = C.c_uint32(0)
        check(B.lib().tmac_hip_chain_status(self._h, C.byref(w)))
        return w.value

    def wpq(self, op: int) -> int:
        w = C.c_int32(0)
        check(B.lib().tmac_hip_chain_info(self._h, op, None, C.byref(w), None, None))
        return w.value

    def tap_layout(self, op: int):
        """(offset, count) in ints of call `op` inside the tap buffer; op == nops: (size of the whole buffer, 0)"""
        off, cnt = C.c_size_t(0), C.c_size_t(0)
        check(B.lib().tmac_hip_chain_tap_layout(self._h, op, C.byref(off), C.byref(cnt)))
        return off.value, cnt.value

    def set_tap(self, dev_buffer) -> None:
        """parity tap (include/tmac_hip.h): an int32 device buffer of tap_layout(nops)[0] ints, or None"""
        check(B.lib().tmac_hip_chain_set_tap(self._h, _ptr(dev_buffer) if dev_buffer is not None else None))
        self._tap = dev_buffer

    def set_stamps(self, dev_buffer) -> None:
        check(B.lib().tmac_hip_chain_set_stamps(self._h, _ptr(dev_buffer)))
        self._stamps = dev_buffer

    def free(self):
        if self._h:
            B.lib().tmac_hip_chain_free(self._h)
            self._h = C.c_void_p()
        self._keep = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _ChainRecorder:
    def __init__(self, wrapper):
        self.wr = wrapper
        self.chain: Optional[DecodeChain] = None

    def __enter__(self):
        check(B.lib().tmac_hip_chain_begin())
        self.wr._recording = []
        return self

    def __exit__(self, et, ev, tb):
        keep, self.wr._recording = self.wr._recording, None
        h = C.c_void_p()
        rc = B.lib().tmac_hip_chain_end(C.byref(h))
        if et is None:
            check(rc)
            self.chain = DecodeChain(h, keep)
        elif rc == 0 and h:
            B.lib().tmac_hip_chain_free(h)      # the block raised: the chain that was built anyway is not handed out
        return False


class TMACGeMMWrapper:
    """Mirror of the reference wrapper's no-TVM branch (tmac_gemm_wrapper.h:170-228).

    >>> wr = TMACGeMMWrapper(act_group_size=64, kcfg_file="kcfg.ini")
    >>> wr.set_workspace(maxK=11008, maxN=1)
    >>> w = wr.register_weights(A_ref, scales_ref, M=4096, K=11008, bits=2)
    >>> wr.llama_cpp_init(B_dev, M=4096, K=11008, N=1, bits=2)        # preprocessor -> workspace
    >>> wr.llama_cpp_compute(w, C_dev, N=1)                           # qgemm_lut
    """

    def __init__(self, n_threads: int = 1, act_group_size: int = 32, kcfg_file: str = "", library_file: str = ""):
        self._act_group_size = act_group_size
        self._ws: Optional[Workspace] = None
        L = B.lib()
        if kcfg_file:
            check(L.tmac_hip_load_kcfg(kcfg_file.encode()))

    def set_num_threads(self, n_threads: int) -> None:  # no-op without a TVM threadpool, as in the reference
        pass

    def set_workspace(self, maxK: int, maxN: int) -> None:
        self._ws = Workspace(maxK, maxN)

    @property
    def workspace(self) -> Workspace:
        if self._ws is None:
            raise RuntimeError("set_workspace() must be called first (tmac_gemm_wrapper.h:257-270)")
        return self._ws

    def get_kcfg(self, M: int, K: int, N: int, bits: int) -> KCfg:
        cfg = KCfg()
        check(B.lib().tmac_hip_get_kcfg(M, K, N, bits, C.byref(cfg)))
        return cfg

    def set_kcfg(self, M: int, K: int, N: int, bits: int, cfg: KCfg) -> None:
        check(B.lib().tmac_hip_set_kcfg(M, K, N, bits, C.byref(cfg)))

    def register_weights(self, A_ref, scales_ref, M: int, K: int, bits: int, cfg: Optional[KCfg] = None,
                         scales_dtype=F32, dev_dtype=F32, on_device=False, fast_aggregation: int = 0) -> Weights:
        """fast_aggregation: 0 exact (default) | 1 signed halving adds (the reference's ARM ``-fa`` build) | 2 its AVX2
        build; the lossy aggregation of python/t_mac/ops/qgemm.py:30,86 — a property of the registered weights here."""
        cfg = cfg or self.get_kcfg(M, K, 1, bits)
        if not fast_aggregation:
            return Weights(A_ref, scales_ref, M, K, bits, cfg, scales_dtype, dev_dtype, on_device)
        check(B.lib().tmac_hip_set_fast_aggregation(int(fast_aggregation)))
        try:
            return Weights(A_ref, scales_ref, M, K, bits, cfg, scales_dtype, dev_dtype, on_device)
        finally:
            B.lib().tmac_hip_set_fast_aggregation(0)

    def llama_cpp_init(self, B_dev, M: int, K: int, N: int, bits: int, act_group_size: Optional[int] = None,
                       act_dtype: Optional[int] = None, stream=None) -> None:
        """preprocessor_int8(M*bits, K, N, bits, B, lut_scales, lut_biases, qlut) on the device."""
        ags = act_group_size or self._act_group_size
        if act_dtype is None:
            act_dtype = _dtype_code(B_dev)
        check(B.lib().tmac_hip_preprocessor_dev(self.workspace.handle, _ptr(B_dev), act_dtype, K, N, ags, _stream(stream)))

    def llama_cpp_compute(self, weights: Weights, C_dev, N: int = 1, out_dtype: Optional[int] = None, stream=None) -> None:
        """qgemm_lut_int8 over ALL M-tiles of the registered matrix in one launch."""
        if out_dtype is None:
            out_dtype = _dtype_code(C_dev)
        check(B.lib().tmac_hip_qgemm_dev(weights.handle, self.workspace.handle, _ptr(C_dev), out_dtype, N, _stream(stream)))

    def fused(self, weights_list, B_dev, C_list, N: int = 1, act_dtype: Optional[int] = None,
              out_dtype: Optional[int] = None, stream=None) -> None:
        """llama_cpp_init + llama_cpp_compute in ONE launch for up to 4 matrices that share the activation
        rows B_dev (q/k/v, gate/up): tmac_hip_qgemm_fused_dev.  The LUT is built inside the kernel."""
        n = len(weights_list)
        if act_dtype is None:
            act_dtype = _dtype_code(B_dev)
        if out_dtype is None:
            out_dtype = _dtype_code(C_list[0])
        # keyed by the native handle VALUES (an id() can be reused once a Weights object is collected); the arrays
        # hold raw handles, so an entry is dropped when any of its weights is freed (see Weights.free / _forget)
        key = (tuple(w.handle.value for w in weights_list), tuple(_ptr(c) for c in C_list))
        cache = self.__dict__.setdefault("_fused_cache", {})
        if key not in cache:   # the ctypes pointer arrays are reused across calls (cheap host path)
            if len(cache) > 4096:
                cache.clear()
            wa = (C.c_void_p * n)(*[w.handle.value for w in weights_list])
            ca = (C.c_void_p * n)(*[_ptr(c) for c in C_list])
            cache[key] = (wa, ca)
            for w in weights_list:
                w._caches.append(cache)
        wa, ca = cache[key]
        rec = getattr(self, "_recording", None)
        if rec is not None:
            rec.append((list(weights_list), B_dev, list(C_list)))
        check(B.lib().tmac_hip_qgemm_fused_dev(wa, n, _ptr(B_dev), act_dtype, ca, out_dtype, N, _stream(stream)))

    def record_gather(self, send_dev, recv_dev, bytes_per_rank: int, rank: int, world: int) -> None:
        """while recording a chain: the exchange step ``recv = all-gather over the ranks of send`` (what ``Comm.allgather`` records
        by itself); inside the launch it becomes part of the hand-off"""
        rec = getattr(self, "_recording", None)
        if rec is not None:
            rec.append((send_dev, recv_dev))
        check(B.lib().tmac_hip_chain_record_gather(_ptr(send_dev), _ptr(recv_dev), bytes_per_rank, rank, world))

    CARRY = "carry"

    def chain_xform(self, kind: str, in2=None, residual=None, gamma=None, eps: float = 1e-5, residual_out=None, keep: bool = False) -> None:
        """while recording a chain: a vector transform of the NEXT ``fused`` call's activations, applied inside its LUT build
        (tmac_hip_chain_xform).  kind "norm": t = in + residual (fp32 tensor, None, or ``TMACGeMMWrapper.CARRY`` = the t the latest
        kept NORM left in LDS); x = t * rsqrt(mean(t^2) + eps) * gamma (x = t without gamma); residual_out: t also goes to memory;
        keep: leave t for a later CARRY.  kind "glu": x = silu(in) * in2."""
        xf = B.XForm()
        xf.kind = {"norm": 1, "glu": 2}[kind]
        xf.in2 = _ptr(in2) if in2 is not None else None
        xf.residual = 1 if residual is TMACGeMMWrapper.CARRY else (_ptr(residual) if residual is not None else None)
        xf.gamma = _ptr(gamma) if gamma is not None else None
        xf.eps = float(eps)
        xf.residual_out = _ptr(residual_out) if residual_out is not None else None
        xf.keep = 1 if keep else 0
        rec = getattr(self, "_recording", None)
        if rec is not None:
            rec.append((in2, residual, gamma, residual_out))      # kept alive with the chain
        check(B.lib().tmac_hip_chain_xform(C.byref(xf)))

    def record_chain(self) -> "_ChainRecorder":
        """``with wr.record_chain() as rec: <the token's wr.fused(...) calls>`` — the calls are noted instead of launched;
        afterwards ``rec.chain.launch()`` executes all of them in ONE persistent kernel launch (tmac_hip_chain_*)."""
        return _ChainRecorder(self)

    def autotune(self, weights_list, act_dtype=F16, out_dtype=F16):
        """Measure the launch configurations of the fused decode kernel on these matrices (the list ``fused`` will be
        called with) and keep the fastest: ``tmac_hip_autotune_fused``.  Returns dict(ft, wpq, us, heuristic_us);
        ft == 0 means the built-in heuristic was kept."""
        n = len(weights_list)
        wa = (C.c_void_p * n)(*[w.handle.value for w in weights_list])
        ft, wpq, us, hus = C.c_int(0), C.c_int(0), C.c_float(0), C.c_float(0)
        check(B.lib().tmac_hip_autotune_fused(wa, n, act_dtype, out_dtype, C.byref(ft), C.byref(wpq), C.byref(us), C.byref(hus)))
        return dict(ft=ft.value, wpq=wpq.value, us=us.value, heuristic_us=hus.value)

    @staticmethod
    def tune_save(path: str) -> int:
        return check_count(B.lib().tmac_hip_tune_save(path.encode()))

    @staticmethod
    def tune_load(path: str) -> int:
        return check_count(B.lib().tmac_hip_tune_load(path.encode()))

    def fused_partial_sums(self, weights: Weights, B_dev, N: int = 1, act_dtype: Optional[int] = None, stream=None):
        """Parity tap of the fused kernel: (int32 PS as partial_sums(), fp32 C [N][Mw]); the in-kernel LUT
        scales/biases are left in ``self.last_fused_lut``."""
        if act_dtype is None:
            act_dtype = _dtype_code(B_dev)
        s_final = weights.cfg.m_groups >= 1 and weights.cfg.act_group_size == weights.K
        G = 1 if s_final else weights.K // weights.cfg.act_group_size
        ps = np.zeros((N, weights.Mw * weights.bits, G), np.int32)
        c = np.zeros((N, weights.Mw), np.float32)
        lut = np.zeros((N, 2, weights.K // weights.cfg.act_group_size), np.float32)
        check(B.lib().tmac_hip_qgemm_fused_partial_sums(weights.handle, _ptr(B_dev), act_dtype, ps.ctypes.data,
                                                        c.ctypes.data, lut.ctypes.data, N, _stream(stream)))
        self.last_fused_lut = lut   # [N][0] = LUT scales, [N][1] = LUT biases built inside the kernel
        return ps, c

    def comb_sums(self, weights: Weights, N: int, stream=None) -> np.ndarray:
        """Parity tap of the plane-combined GEMM (k_gemm_planes): int32 [N][Mw][K/64], sum_p 2^p PS_p."""
        G = 1 if weights.cfg.m_groups >= 1 else weights.K // 64
        out = np.zeros((N, weights.Mw, G), np.int32)
        check(B.lib().tmac_hip_debug_gemm_comb_sums(weights.handle, self.workspace.handle, out.ctypes.data, N, _stream(stream)))
        return out

    def partial_sums(self, weights: Weights, N: int = 1, stream=None) -> np.ndarray:
        """Parity tap: int32 [N][M][K/ags] (or [N][M] for the unified-scale path), M-space row order."""
        s_final = weights.cfg.m_groups >= 1 and weights.cfg.act_group_size == weights.K
        G = 1 if s_final else weights.K // weights.cfg.act_group_size
        out = np.zeros((N, weights.Mw * weights.bits, G), np.int32)
        check(B.lib().tmac_hip_qgemm_partial_sums(weights.handle, self.workspace.handle, out.ctypes.data, N, _stream(stream)))
        return out
