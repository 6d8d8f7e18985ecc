"""Converter-side formats (tmac_amd/convert.py) against the reference's own Python (imports without TVM) and against
the kcfg parser of the C library.  CPU only."""
import os
import sys

import numpy as np
import pytest

import tmac_amd
from tmac_amd import convert

REF_PY = "/root/reference/python"
have_ref = os.path.isdir(os.path.join(REF_PY, "t_mac"))


def ref_model_utils():
    if REF_PY not in sys.path:
        sys.path.insert(0, REF_PY)
    import t_mac.model_utils as mu
    return mu


def gptq_case(seed, K, M, bits, gs, fp16=True):
    rng = np.random.default_rng(seed)
    per = 32 // bits
    w = rng.integers(0, 2 ** bits, size=(K, M), dtype=np.uint32)
    qweight = np.zeros((K // per, M), np.uint32)
    for f in range(per):
        qweight |= w[f::per] << np.uint32(bits * f)
    z = rng.integers(0, 2 ** bits, size=(K // gs, M), dtype=np.uint32)
    qzeros = np.zeros((K // gs, M // per), np.uint32)
    for f in range(per):
        qzeros |= z[:, f::per] << np.uint32(bits * f)
    scales = np.abs(rng.standard_normal((K // gs, M))).astype(np.float16 if fp16 else np.float32)
    return qweight.view(np.int32), scales, qzeros.view(np.int32), w.T.astype(np.uint8), z.T


@pytest.mark.parametrize("bits,K,M,gs,v2", [(2, 256, 64, 128, True), (4, 512, 96, 128, True), (4, 256, 32, 64, False), (2, 128, 160, 32, True)])
def test_unpack_gptq(bits, K, M, gs, v2):
    qw, sc, qz, w_true, z_true = gptq_case(bits * 100 + K, K, M, bits, gs)
    assert convert.parse_gptq(qw, sc, qz) == (K, M, bits, gs)
    w, s, z, b, g = convert.unpack_gptq(qw, sc, qz, gptq_v2=v2)
    assert b == bits and g == gs
    assert np.array_equal(w, w_true)
    assert np.array_equal(s, sc.T)
    zt = (z_true.astype(sc.dtype) + (0 if v2 else 1) - 2 ** (bits - 1)) * sc.T
    assert np.array_equal(z, zt)
    if have_ref:
        mu = ref_model_utils()
        wr, sr, zr, br, gr = mu.unpack_gptqv2(qw, sc, qz, gptq_v2=v2)
        assert (br, gr) == (bits, gs)
        assert np.array_equal(w, wr) and np.array_equal(s, sr) and np.array_equal(z, zr)


def test_kcfg_roundtrip_and_blob(tmp_path):
    path = str(tmp_path / "kcfg.ini")
    kernels = convert.PRESET_KERNELS["llama-2-7b-2bit"] + convert.PRESET_KERNELS["hf-bitnet-3b"][:1]
    convert.write_kcfg(path, kernels[:3], bm={(2, 4096, 4096): 128, (2, 11008, 4096): 128, (2, 4096, 11008): 128})
    e = convert.read_kcfg_entry(path, 4096, 11008, 2)
    assert e == dict(bm=128, simd_n_in=16, simd_n_out=8, kfactor=16, group_size=128, lut_scales_size=172,
                     scales_size=704512, n_tile_num=64)
    # byte-identical in content with the file the reference ships for this model
    if have_ref:
        import configparser
        a, b = configparser.ConfigParser(), configparser.ConfigParser()
        a.read(path)
        b.read("/root/reference/deploy/tuned/aarch64-llama-2-7b-2bit/kcfg.ini")
        for sec in a.sections():
            assert dict(a[sec]) == dict(b[sec]), sec
    # the C library parses what we wrote
    L = tmac_amd.lib()
    tmac_amd.binding.check(L.tmac_hip_load_kcfg(path.encode()))
    cfg = tmac_amd.TMACGeMMWrapper(act_group_size=64).get_kcfg(4096, 11008, 1, 2)
    assert (cfg.bm, cfg.kfactor, cfg.group_size, cfg.n_tile_num) == (128, 16, 128, 64)

    # blob: [weight bytes][fp32 scales], identical to the reference's, and split_blob inverts the concatenation
    rng = np.random.default_rng(5)
    Mw, K, bits = 4096, 4096, 2
    w = rng.integers(0, 4, size=(Mw, K), dtype=np.uint8)
    sc = np.abs(rng.standard_normal((Mw, K // 128))).astype(np.float16)
    zr = rng.standard_normal((Mw, K // 128)).astype(np.float16)
    blob = convert.preprocess_for_t_mac(path, w, sc, zr, bits=bits)
    assert blob.dtype == np.uint8 and blob.size == Mw * K * bits // 8 + Mw * (K // 128) * 2 * 4
    A, S = convert.split_blob(blob, Mw, K, bits)
    A2, S2 = tmac_amd.weights.preprocess_weights(w, sc, zr, bits=bits, bm=128, kfactor=16)
    assert np.array_equal(A, A2.reshape(-1)) and np.array_equal(S, S2.astype(np.float32).reshape(-1))
    if have_ref:
        mu = ref_model_utils()
        assert np.array_equal(blob, mu.preprocess_for_t_mac(path, w, sc, zr, bits=bits))


def test_bitnet_kcfg_entry(tmp_path):
    path = str(tmp_path / "k.ini")
    convert.write_kcfg(path, convert.PRESET_KERNELS["hf-bitnet-3b"], act_group_size=-1, zero_point=False,
                       bm={(2, 3200, 8640): 128, (2, 8640, 3200): 128, (2, 3200, 3200): 320})
    e = convert.read_kcfg_entry(path, 3200, 8640, 2)
    assert e["scales_size"] == 1 and e["lut_scales_size"] == 1 and e["bm"] == 128 and e["n_tile_num"] == 50
    if have_ref:   # the shipped (ARM) set keeps act_group_size = 64 with one weight scale (lut_scales_size = K/64)
        import configparser
        path2 = str(tmp_path / "k64.ini")
        convert.write_kcfg(path2, convert.PRESET_KERNELS["hf-bitnet-3b"], act_group_size=64, zero_point=False,
                           bm={(2, 3200, 8640): 128, (2, 8640, 3200): 128, (2, 3200, 3200): 320})
        b = configparser.ConfigParser()
        b.read("/root/reference/deploy/tuned/aarch64-hf-bitnet-3b/kcfg.ini")
        a = configparser.ConfigParser()
        a.read(path2)
        common = [sec for sec in a.sections() if sec in b]
        assert common
        for sec in common:
            assert dict(a[sec]) == dict(b[sec]), sec


def _fake_checkpoint(d, layers, desc_act=False):
    """config.json + two safetensors parts holding GPTQ-packed linear layers (qweight / scales / qzeros) and an unquantised tensor"""
    import json
    from safetensors.numpy import save_file
    parts = [{}, {}]
    for i, (bits, M, K, gs) in enumerate(layers):
        qw, sc, qz, _, _ = gptq_case(50 + i, K, M, bits, gs)
        pre = f"model.layers.{i}.proj"
        parts[i % 2].update({pre + ".qweight": qw, pre + ".scales": sc, pre + ".qzeros": qz})
    parts[0]["model.embed_tokens.weight"] = np.zeros((8, 4), np.float16)
    save_file(parts[0], os.path.join(d, "model-00001-of-00002.safetensors"))
    save_file(parts[1], os.path.join(d, "model-00002-of-00002.safetensors"))
    cfg = {"quantization_config": {"bits": layers[0][0], "group_size": layers[0][3], "sym": False, "quant_method": "gptq", "desc_act": desc_act,
                                   "meta": {"quantizer": "gptqmodel:1.0"}}, "hidden_size": 64}
    json.dump(cfg, open(os.path.join(d, "config.json"), "w"))


def test_presets_and_checkpoint_discovery(tmp_path):
    """preset shape table, kernel shapes read from a GPTQ checkpoint's safetensors HEADERS, and the quantisation config -- equal to the
    reference's model_utils (which loads the tensors to learn their shapes)"""
    d = str(tmp_path)
    layers = [(2, 64, 256, 128), (2, 160, 256, 128), (2, 64, 256, 128), (4, 96, 512, 128), (2, 64, 640, 128)]
    _fake_checkpoint(d, layers)
    ks = convert.extract_kernel_shapes("gptq-auto", d)
    assert sorted(ks) == sorted([[2, 64, 256, 1, -1], [2, 160, 256, 1, -1], [4, 96, 512, 1, -1], [2, 64, 640, 1, -1]])    # distinct (bits, M, K), each once
    assert convert.checkpoint_group_size(d) == 128
    qc = convert.get_quantization_config(d)
    assert qc["bits"] == 2 and qc["group_size"] == 128 and qc["quant_method"] == "gptq" and qc["quantizer"] == "gptqmodel:1.0"
    with pytest.raises(KeyError):
        convert.extract_kernel_shapes("no-such-model")
    if have_ref:
        from pathlib import Path
        mu = ref_model_utils()
        assert set(convert.get_preset_models()) == set(mu.get_preset_models())
        for name in mu.get_preset_models():
            if name != "gptq-auto":
                assert convert.extract_kernel_shapes(name) == mu.extract_kernel_shapes(name)
        assert sorted(ks) == sorted(mu.extract_kernel_shapes("gptq-auto", d))
        assert qc == mu.get_quantization_config(Path(d))
    # mixed group sizes and act-order checkpoints are refused, as in the reference
    d2 = os.path.join(d, "mixed"); os.makedirs(d2)
    _fake_checkpoint(d2, [(2, 64, 256, 128), (2, 64, 256, 64)])
    with pytest.raises(RuntimeError):
        convert.extract_kernel_shapes("gptq-auto", d2)
    d3 = os.path.join(d, "actorder"); os.makedirs(d3)
    _fake_checkpoint(d3, [(2, 64, 256, 128)], desc_act=True)
    with pytest.raises(AssertionError):
        convert.get_quantization_config(d3)


def test_corrupt_checkpoint_parts_are_named(tmp_path):
    """a truncated or malformed safetensors part surfaces as a RuntimeError that names the file, not as a bare struct / JSON /
    ZeroDivisionError"""
    import json
    import struct
    from tmac_amd import convert as cv
    d = tmp_path / "ckpt"; d.mkdir()
    p = d / "model.safetensors"
    p.write_bytes(b"\x10\x00\x00")                                       # shorter than the length field
    with pytest.raises(RuntimeError, match="model.safetensors"):
        cv.extract_kernel_shapes("gptq-auto", str(d))
    p.write_bytes(struct.pack("<Q", 1 << 40) + b"{}")                     # header length beyond the file
    with pytest.raises(RuntimeError, match="corrupt safetensors header"):
        cv.extract_kernel_shapes("gptq-auto", str(d))
    p.write_bytes(struct.pack("<Q", 5) + b"{oops")                        # not JSON
    with pytest.raises(RuntimeError, match="corrupt safetensors header"):
        cv.extract_kernel_shapes("gptq-auto", str(d))
    hdr = {"l.qweight": {"dtype": "I32", "shape": [256, 64], "data_offsets": [0, 0]}, "l.scales": {"dtype": "F16", "shape": [32, 64], "data_offsets": [0, 0]},
           "l.qzeros": {"dtype": "I32", "shape": [32, 0], "data_offsets": [0, 0]}}
    raw = json.dumps(hdr).encode()
    p.write_bytes(struct.pack("<Q", len(raw)) + raw)                     # a zero-width qzeros tensor
    with pytest.raises(RuntimeError, match="not GPTQ-packed shapes"):
        cv.extract_kernel_shapes("gptq-auto", str(d))
