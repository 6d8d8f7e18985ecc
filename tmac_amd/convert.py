"""Checkpoint-side data formats of the T-MAC hot path: what sits on either side of
``tmac_hip_register_weights``.

* GPTQ (v1 / v2) packed tensors -> biased uint weights, scales, T-MAC zeros      (``python/t_mac/model_utils.py:95-129``)
* ``kcfg.ini`` emission for a set of (bits, M, K, N, m_groups) kernels           (``deploy/compile.py:153-165``; the
  file the converter and the runtime must share because bm / kfactor are baked into the weight bytes)
* the per-tensor blob a T-MAC GGUF stores: ``[permuted weight bytes][fp32 scales]`` (``model_utils.py:243-271``), and
  its inverse split for registration on the GPU

Nothing here touches a GPU: numpy in, numpy out.  Results are identical to the reference's functions (tests compare
them against the reference's Python, which imports without TVM).
"""
import configparser
import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .weights import preprocess_weights

# (bits, M = weight rows, K, N, m_groups) per preset, as the reference ships them (model_utils.py:20-89)
PRESET_KERNELS: Dict[str, List[List[int]]] = {
    "llama-2-7b-4bit": [[4, 4096, 4096, 1, -1], [4, 11008, 4096, 1, -1], [4, 4096, 11008, 1, -1]],
    "llama-2-7b-2bit": [[2, 4096, 4096, 1, -1], [2, 11008, 4096, 1, -1], [2, 4096, 11008, 1, -1]],
    "llama-2-13b-2bit": [[2, 5120, 5120, 1, -1], [2, 13824, 5120, 1, -1], [2, 5120, 13824, 1, -1]],
    "llama-3-8b-2bit": [[2, 4096, 4096, 1, -1], [2, 14336, 4096, 1, -1], [2, 4096, 14336, 1, -1], [2, 1024, 4096, 1, -1]],
    "llama-3-8b-4bit": [[4, 4096, 4096, 1, -1], [4, 14336, 4096, 1, -1], [4, 4096, 14336, 1, -1], [4, 1024, 4096, 1, -1]],
    "hf-bitnet-3b": [[2, 3200, 8640, 1, 1], [2, 8640, 3200, 1, 1], [2, 3200, 3200, 1, 1]],
    "hf-bitnet-large-intn": [[2, 1536, 4096, 1, 1], [2, 4096, 1536, 1, 1], [2, 1536, 1536, 1, 1]],
    "hf-bitnet-large-tq": [[2, 1536, 4096, 1, -1], [2, 4096, 1536, 1, -1], [2, 1536, 1536, 1, -1]],
    "ms-bitnet-3b": [[2, 3200, 800, 1, 1], [2, 3200, 3200, 1, 1], [2, 3200, 10240, 1, 1], [2, 10240, 3200, 1, 1], [2, 800, 3200, 1, 1]],
    "phi-3-mini-2bit": [[2, 3072, 3072, 1, -1], [2, 9216, 3072, 1, -1], [2, 3072, 8192, 1, -1], [2, 16384, 3072, 1, -1]],
    "trilm-3.9b": [[2, 3072, 3072, 1, -1], [2, 3072, 9216, 1, -1], [2, 9216, 3072, 1, -1], [2, 768, 3072, 1, -1]],
    "test": [],
    "gptq-auto": [],            # shapes read from the checkpoint (extract_kernel_shapes)
}


def get_preset_models() -> List[str]:
    return list(PRESET_KERNELS.keys())


def _safetensors_shapes(path: str) -> Dict[str, Tuple[str, Tuple[int, ...]]]:
    """{tensor name: (dtype, shape)} from a .safetensors file's header alone (8-byte little-endian length + JSON): a 7B checkpoint's kernel
    shapes are known after reading a few hundred KB, no tensor is loaded (the reference reads every scales / qzeros / qweight tensor)"""
    import json
    import struct
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        head = f.read(8)
        if len(head) != 8:
            raise RuntimeError("{}: not a safetensors file (shorter than its header length field)".format(path))
        (n,) = struct.unpack("<Q", head)
        if n == 0 or n > size - 8:
            raise RuntimeError("{}: corrupt safetensors header (length {} in a file of {} bytes)".format(path, n, size))
        try:
            hdr = json.loads(f.read(n))
            return {k: (v["dtype"], tuple(int(d) for d in v["shape"])) for k, v in hdr.items() if k != "__metadata__"}
        except (ValueError, KeyError, TypeError) as e:
            raise RuntimeError("{}: corrupt safetensors header ({})".format(path, e))


def extract_kernel_shapes(model_arch: Optional[str] = "gptq-auto", model_dir: Optional[str] = None) -> List[List[int]]:
    """The kernels a model needs, [bits, M, K, N, m_groups] each (``model_utils.py:206-216``): a preset's list, or -- "gptq-auto" -- the distinct
    (bits, M, K) of the GPTQ-packed linear layers found in the checkpoint's ``model*.safetensors`` parts, in order of first appearance.
    One group size per model, as the reference requires (``model_utils.py:170-198``); ``checkpoint_group_size`` returns it."""
    if model_arch not in PRESET_KERNELS:
        raise KeyError("Unsupported model_arch: {}".format(model_arch))
    if model_arch != "gptq-auto":
        return [list(k) for k in PRESET_KERNELS[model_arch]]
    return _scan_checkpoint(model_dir)[0]


def checkpoint_group_size(model_dir: str) -> int:
    return _scan_checkpoint(model_dir)[1]


def _scan_checkpoint(model_dir: Optional[str]) -> Tuple[List[List[int]], int]:
    if model_dir is None:
        raise ValueError("gptq-auto needs the checkpoint directory")
    parts = sorted(f for f in os.listdir(model_dir) if f.startswith("model") and f.endswith(".safetensors"))
    if not parts:
        raise RuntimeError("Models in {} not in GPTQ safetensors format (torch .bin checkpoints: convert them to safetensors first)".format(model_dir))
    shapes: Dict[str, Tuple[int, ...]] = {}
    order: List[str] = []
    for part in parts:
        for name, (_, shp) in _safetensors_shapes(os.path.join(model_dir, part)).items():
            shapes[name] = shp
            if name.endswith(".qweight"):
                order.append(name)
    ks: List[List[int]] = []
    gs_all = None
    for name in order:
        qw, sc, qz = shapes[name], shapes.get(name[:-8] + ".scales"), shapes.get(name[:-8] + ".qzeros")
        if sc is None or qz is None:
            raise RuntimeError("{}: scales / qzeros missing".format(name))
        if len(qw) != 2 or len(sc) != 2 or len(qz) != 2 or qz[1] <= 0 or sc[0] <= 0 or sc[1] < qz[1]:
            raise RuntimeError("{}: qweight {} / scales {} / qzeros {} are not GPTQ-packed shapes".format(name, qw, sc, qz))
        bits = 32 // (sc[1] // qz[1])                      # parse_gptq on shapes alone
        if bits not in (1, 2, 3, 4, 8):
            raise RuntimeError("{}: {} values per packed word".format(name, sc[1] // qz[1]))
        K, M = qw[0] * (32 // bits), qw[1]
        gs = K // sc[0]
        if [bits, M, K, 1, -1] not in ks:
            ks.append([bits, M, K, 1, -1])
        if gs_all is None:
            gs_all = gs
        elif gs_all != gs:
            raise RuntimeError("Different group_sizes unsupported")
    if not ks:
        raise RuntimeError("Models in {} not in GPTQ format".format(model_dir))
    return ks, int(gs_all)


def get_quantization_config(model_dir: str) -> dict:
    """what the pipeline reads from a checkpoint's config.json (``model_utils.py:219-240``): GPTQ fields of ``quantization_config`` and
    BitNet's ``weight_bits``; act-order checkpoints (desc_act) are refused as the reference refuses them"""
    import json
    with open(os.path.join(model_dir, "config.json"), "r", encoding="utf-8") as f:
        hp = json.load(f)
    qc = hp.get("quantization_config", {})
    if qc.get("desc_act", False):
        raise AssertionError("desc_act=True currently unsupported by T-MAC")
    return {"quantizer": qc.get("meta", {}).get("quantizer", ""), "group_size": qc.get("group_size", 0), "bits": qc.get("bits", 0),
            "sym": qc.get("sym", False), "quant_method": qc.get("quant_method", ""), "weight_bits": hp.get("weight_bits", 0)}


def parse_gptq(qweight: np.ndarray, scales: np.ndarray, qzeros: np.ndarray) -> Tuple[int, int, int, int]:
    """(K, M, bits, group_size) from the packed tensor shapes: qweight int32 [K*bits/32][M], scales [K/gs][M],
    qzeros int32 [K/gs][M*bits/32]."""
    bits = 32 // (scales.shape[1] // qzeros.shape[1])
    K = qweight.shape[0] * (32 // bits)
    M = qweight.shape[1]
    return K, M, bits, K // scales.shape[0]


def _unpack_fields(x: np.ndarray, bits: int) -> np.ndarray:
    """int32 [...] -> [..., 32/bits] little-end-first bit fields"""
    shifts = np.arange(0, 32, bits, dtype=np.uint32)
    return ((x.astype(np.uint32)[..., None] >> shifts) & np.uint32((1 << bits) - 1)).astype(np.uint8)


def unpack_gptq(qweight: np.ndarray, scales: np.ndarray, qzeros: np.ndarray, gptq_v2: bool = True):
    """GPTQ tensors -> (w uint8 [M][K] in [0, 2^bits), scales [M][K/gs], zeros [M][K/gs], bits, group_size) with the
    zeros already in T-MAC's convention ``(z - 2^(bits-1)) * scale`` (``weights.py:29-31``); AutoGPTQ v1 stores z - 1."""
    if qweight.dtype != np.int32 or qzeros.dtype != np.int32:
        raise TypeError("qweight and qzeros must be int32")
    K, M, bits, gs = parse_gptq(qweight, scales, qzeros)
    # qweight[kp][m] packs rows kp*(32/bits) .. +32/bits-1 of column m
    w = _unpack_fields(qweight, bits).transpose(1, 0, 2).reshape(M, K)
    sc = np.ascontiguousarray(scales.T)
    z = _unpack_fields(qzeros, bits).reshape(K // gs, M).T.astype(sc.dtype)
    if not gptq_v2:
        z = z + 1
    return np.ascontiguousarray(w), sc, (z - (2 ** (bits - 1))) * sc, bits, gs


def kernel_name(M_bits: int, K: int, N: int, bits: int) -> str:
    return f"qgemm_lut_t1_int8_m{M_bits}_k{K}_n{N}_b{bits}"


def default_bm(bits: int, Mw: int) -> int:
    """a legal M-tile for shapes without a tuned entry: the largest of the reference's knob space that tiles M
    (``qgemm.py:98-116``); the GPU kernels do not care, the value only fixes the blob layout"""
    for bm in (256, 128, 512, 1024, 320, 640) if bits != 3 else (192, 384, 576, 768):
        if (Mw * bits) % bm == 0 and bm % bits == 0 and (bm // bits) % 8 == 0:
            return bm
    raise ValueError(f"no legal bm for Mw={Mw}, bits={bits}")


def write_kcfg(path: str, kernels: Iterable[Sequence[int]], group_size: int = 128, act_group_size: int = 64,
               zero_point: bool = True, bm: Optional[Dict[Tuple[int, int, int], int]] = None, kfactor: int = 16) -> None:
    """Emit the kcfg.ini the reference's ``deploy/compile.py`` writes next to its kernels: one section per kernel,
    fields bm, simd_n_in, simd_n_out, kfactor, group_size, lut_scales_size, scales_size, n_tile_num."""
    cf = configparser.ConfigParser()
    for bits, Mw, K, N, m_groups in kernels:
        b = (bm or {}).get((bits, Mw, K), default_bm(bits, Mw))
        ags = K if act_group_size == -1 else act_group_size
        zp = zero_point and m_groups == -1
        scales_size = m_groups if m_groups != -1 else Mw * K // group_size * (2 if zp else 1)
        cf[kernel_name(Mw * bits, K, N, bits)] = {
            "bm": str(b), "simd_n_in": "16", "simd_n_out": "8", "kfactor": str(kfactor),
            "group_size": str(group_size), "lut_scales_size": str(N * K // ags),
            "scales_size": str(scales_size), "n_tile_num": str(Mw * bits // b),
        }
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        cf.write(f)


def read_kcfg_entry(path: str, Mw: int, K: int, bits: int) -> Dict[str, int]:
    """the section for (M*bits, K) as ``preprocess_for_t_mac`` looks it up (first match on the m / k name fields)"""
    cf = configparser.ConfigParser()
    cf.read(path)
    for sec in cf.sections():
        f = sec.split("_")
        if f[-4] == f"m{Mw * bits}" and f[-3] == f"k{K}":
            return {k: int(v) for k, v in cf[sec].items()}
    raise KeyError(f"GEMM of shape ({Mw}, {K}) is not in {path}")


def preprocess_for_t_mac(kcfg_file: str, w: np.ndarray, scales: np.ndarray, zeros: Optional[np.ndarray] = None,
                         bits: int = 2) -> np.ndarray:
    """uint weights [M][K] (+ scales, zeros) -> the byte blob a T-MAC GGUF tensor holds: the permuted weight bytes
    followed by the fp32 scales (``model_utils.py:243-271``)."""
    M, K = w.shape
    e = read_kcfg_entry(kcfg_file, M, K, bits)
    A, S = preprocess_weights(w, scales, zeros, bits=bits, g=4, bm=e["bm"], kfactor=e["kfactor"],
                              simd_n_in=e["simd_n_in"], simd_n_out=e["simd_n_out"])
    return np.concatenate([A.reshape(-1), np.ascontiguousarray(S.astype(np.float32)).view(np.uint8).reshape(-1)])


def split_blob(blob: np.ndarray, Mw: int, K: int, bits: int) -> Tuple[np.ndarray, np.ndarray]:
    """inverse of the concatenation: (weight bytes uint8 [Mw*K*bits/8], scales fp32 [...]) — the two pointers
    ``tmac_hip_register_weights`` takes"""
    nw = Mw * K * bits // 8
    b = np.ascontiguousarray(blob, dtype=np.uint8)
    return b[:nw], b[nw:].view(np.float32)
