"""Callers and data formats either side of the hot path, on the GPU (SURVEY.md 8f N1 / N2 / N3 and the host-pointer ABI):

* N2  GPTQ-packed int32 tensors -> unpack_gptq -> preprocess_for_t_mac (the GGUF blob) -> split_blob ->
      tmac_hip_register_weights -> GEMV, against a dequantise-and-multiply in fp64 (python/t_mac/model_utils.py:104-129,243-271)
* N1  the ggml op-hook glue (src/ggml_tmac_hip.cc: upload / mul_mat / free against a stand-in ggml_tensor), compiled with g++
      and run on the same blob, against the oracle
* the reference's wrapper interface with HOST pointers (include/t-mac/tmac_gemm_wrapper.h: llama_cpp_init on the main
      thread, llama_cpp_compute per tile from 8 threads), against the oracle; a reused pointer with other contents is detected
* N3  the CMake consumer computes a GEMV through the package and compares it with the oracle
"""
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "tmac_amd", "lib")


@pytest.fixture(scope="module")
def tm():
    import torch
    import tmac_amd
    assert torch.cuda.is_available()
    return tmac_amd


def gxx(out, *srcs, extra=()):
    subprocess.run(["g++", "-O2", "-std=c++17", "-w", "-I" + os.path.join(ROOT, "include"), *srcs, "-o", out, "-L" + LIBDIR, "-ltmac_hip",
                    "-Wl,-rpath," + LIBDIR, "-ldl", "-lpthread", *extra], check=True, capture_output=True, timeout=300)


def pack_fields(v, bits):
    """uint8 [..., 32/bits] -> int32 [...], little-end-first bit fields (the inverse of the GPTQ unpack)"""
    sh = np.arange(0, 32, bits, dtype=np.uint64)
    return (v.astype(np.uint64) << sh).sum(axis=-1).astype(np.uint32).view(np.int32)


def make_gptq(rng, M, K, bits, gs):
    """synthetic GPTQ v2 tensors: qweight int32 [K*bits/32][M], scales fp16 [K/gs][M], qzeros int32 [K/gs][M*bits/32]"""
    w = rng.integers(0, 2 ** bits, size=(M, K), dtype=np.uint8)
    z = rng.integers(0, 2 ** bits, size=(M, K // gs), dtype=np.uint8)
    sc = (np.abs(rng.standard_normal((M, K // gs))) * 0.02 + 0.005).astype(np.float16)
    per = 32 // bits
    qweight = np.ascontiguousarray(pack_fields(w.T.reshape(K // per, per, M).transpose(0, 2, 1), bits))
    qzeros = np.ascontiguousarray(pack_fields(z.T.reshape(K // gs, M // per, per), bits))
    return w, z, sc, qweight, np.ascontiguousarray(sc.T), qzeros


@pytest.mark.parametrize("bits,M,K", [(2, 1024, 4096), (4, 512, 2048)])
def test_gptq_checkpoint_to_gemv(tm, tmp_path, bits, M, K):
    import torch
    from tmac_amd import convert
    gs, ags = 128, 64
    rng = np.random.default_rng(bits)
    w, z, sc, qweight, scales_t, qzeros = make_gptq(rng, M, K, bits, gs)
    wu, s2, zeros, b2, g2 = convert.unpack_gptq(qweight, scales_t, qzeros, gptq_v2=True)
    assert b2 == bits and g2 == gs and np.array_equal(wu, w)
    kcfg = str(tmp_path / "kcfg.ini")
    convert.write_kcfg(kcfg, [[bits, M, K, 1, -1]], group_size=gs, act_group_size=ags, zero_point=True)
    blob = convert.preprocess_for_t_mac(kcfg, wu, s2.astype(np.float32), zeros.astype(np.float32), bits=bits)
    A, S = convert.split_blob(blob, M, K, bits)
    tm.binding.check(tm.lib().tmac_hip_load_kcfg(kcfg.encode()))
    wr = tm.TMACGeMMWrapper(act_group_size=ags)
    cfg = wr.get_kcfg(M, K, 1, bits)
    wt = wr.register_weights(A, S, M, K, bits, cfg)
    x = rng.standard_normal(K).astype(np.float32)
    out = torch.empty(M, dtype=torch.float32, device="cuda")
    wr.fused([wt], torch.from_numpy(x).cuda(), [out], 1)
    torch.cuda.synchronize()
    # dequantise-and-multiply in fp64: w_real = (w - z) * scale (GPTQ), the LUT path quantises sums of 4 activations to int8
    wreal = (w.astype(np.float64) - np.repeat(z, gs, axis=1)) * np.repeat(sc.astype(np.float64), gs, axis=1)
    ref = wreal @ x.astype(np.float64)
    got = out.cpu().numpy().astype(np.float64)
    nmse = float(((got - ref) ** 2).mean() / (ref ** 2).mean())
    assert nmse <= 5e-4, nmse          # the reference's own acceptance bound for the LUT path (qgemm.py:277-282)
    wt.free()


def fixtures(tmp_path, Mw, K, bits, bm, N=1, seed=3):
    from tmac_amd import convert
    case = orc.make_case(seed, Mw, K, N=N, bits=bits, fp16_values=False)
    A = orc.preprocess_weights(case["w"], bits, bm, 16)
    S = orc.preprocess_scales(case["sc"], case["zr"], bits, bm)
    q, ls, lb = orc.preprocessor(case["B"], 64)
    ref = orc.qgemm_float(A, q, S, ls, lb, Mw, K, N, bits, bm, 16, 128, 64, True)
    d = str(tmp_path)
    A.tofile(os.path.join(d, "A.bin")); S.astype(np.float32).tofile(os.path.join(d, "S.bin"))
    np.concatenate([A.reshape(-1), S.astype(np.float32).view(np.uint8).reshape(-1)]).tofile(os.path.join(d, "blob.bin"))
    case["B"].astype(np.float32).tofile(os.path.join(d, "x.bin")); ref.astype(np.float32).tofile(os.path.join(d, "ref.bin"))
    convert.write_kcfg(os.path.join(d, "kcfg.ini"), [[bits, Mw, K, 1, -1], [bits, Mw, K, N, -1]] if N != 1 else [[bits, Mw, K, 1, -1]],
                       bm={(bits, Mw, K): bm})
    return d


def test_reference_wrapper_host_pointers_eight_threads(tm, tmp_path):
    Mw, K, bits, bm = 4096, 4096, 2, 128
    d = fixtures(tmp_path, Mw, K, bits, bm)
    exe = os.path.join(d, "hostptr_threads")
    gxx(exe, os.path.join(ROOT, "tests", "cpp", "hostptr_threads.cc"))
    env = dict(os.environ); env.pop("TMAC_KCFG_FILE", None)
    r = subprocess.run([exe, d, str(Mw), str(K), str(bits), str(bm), "8"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [x for x in r.stdout.splitlines() if x.startswith("RESULT")][0].split()
    assert float(line[2]) <= 2e-5 and int(line[6]) > Mw // 2
    print(r.stdout)


@pytest.mark.parametrize("N,where", [(1, "host"), (3, "host"), (1, "dev"), (3, "dev")])
def test_ggml_op_hook_glue(tm, tmp_path, N, where):
    Mw, K, bits, bm = 1024, 4096, 2, 128
    d = fixtures(tmp_path, Mw, K, bits, bm, N=N)
    exe = os.path.join(d, "ggml_shim_main")
    # (the device-tensor mode calls three HIP runtime functions itself: link the runtime libtmac_hip.so was built against)
    gxx(exe, os.path.join(ROOT, "tests", "cpp", "ggml_shim_main.cc"), os.path.join(ROOT, "src", "ggml_tmac_hip.cc"),
        extra=("-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"))
    env = dict(os.environ); env.pop("TMAC_KCFG_FILE", None)
    r = subprocess.run([exe, d, str(Mw), str(K), str(bits), str(N)] + (["dev"] if where == "dev" else []), capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.skipif(shutil.which("cmake") is None, reason="cmake not installed")
def test_cmake_consumer_computes_a_gemv(tm, tmp_path):
    Mw, K, bits, bm = 512, 4096, 2, 128
    d = fixtures(tmp_path, Mw, K, bits, bm)
    build = os.path.join(d, "build")
    gen = ["-G", "Ninja"] if shutil.which("ninja") else []
    subprocess.run(["cmake", "-S", os.path.join(ROOT, "tests", "cmake_consumer"), "-B", build, *gen, f"-DTMAC_DIR={os.path.join(ROOT, 'cmake')}",
                    f"-DTMAC_KCFG={os.path.join(d, 'kcfg.ini')}", "-DCMAKE_CXX_COMPILER=g++"], check=True, capture_output=True, timeout=300)
    subprocess.run(["cmake", "--build", build], check=True, capture_output=True, timeout=300)
    env = dict(os.environ); env.pop("TMAC_KCFG_FILE", None)
    r = subprocess.run([os.path.join(build, "consumer"), d, str(Mw), str(K), str(bits), str(bm)], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "gemv max rel err" in r.stdout


@pytest.mark.parametrize("extra,path", [([], "chain"), (["--force-dist"], "chain"), (["--path", "fused"], "fused"),
                                        (["--workload", "bitnet-3b"], "chain")])
def test_bench_runs_and_verifies(tm, extra, path):
    """bench.py end to end on two layers: ONE JSON line on stdout with the contract's keys, the timed path verified against the oracle
    inside the run; --force-dist takes the multi-GPU code path (recorded exchange steps, blob exchange, trial launches) with one rank"""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("MASTER_ADDR", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["MASTER_PORT"] = "29577"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--layers", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"] + extra,
                       capture_output=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["config"]["path"] == path and d["steps"] == 3 and d["n_gpus"] == 1
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1 and d["activations_finite"]
    if "--force-dist" not in extra:
        assert d["verified"]["ok"], d["verified"]
    if path == "chain":
        assert d["roofline"]["headline_gemv"]["us"] > 0
