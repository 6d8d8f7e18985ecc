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


def test_ggml_hook_without_recording_reaches_stream_mode(tm, tmp_path):
    """a hook that is called mat-mul by mat-mul (no chain_begin / segment) issues q / k / v as three separate device-resident
    calls; in deferred mode they run as ONE stream-mode launch (k_lut_images + k_gemv_stream), the recording is cached from the second token
    on, a call that reads a queued output flushes the queue by itself, and every output equals the reference (tests/cpp/ggml_shim_main.cc)"""
    Mw, K, bits, bm = 4096, 4096, 2, 128
    d = fixtures(tmp_path, Mw, K, bits, bm, N=1)
    exe = os.path.join(d, "ggml_shim_main")
    gxx(exe, os.path.join(ROOT, "tests", "cpp", "ggml_shim_main.cc"), os.path.join(ROOT, "src", "ggml_tmac_hip.cc"),
        extra=("-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"))
    env = dict(os.environ); env.pop("TMAC_KCFG_FILE", None)
    r = subprocess.run([exe, d, str(Mw), str(K), str(bits), "1", "batch"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "stream_launches 3" in r.stdout and "cache_hits 2" in r.stdout, r.stdout


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
    if path == "chain":
        # since round 6 a decode line times SURVEY 8(d)'s measurement (independent calls, stream mode) and carries the dependent chain of the
        # same matrices -- rounds 1-5's timed workload -- beside it, measured and verified in the same run
        assert d["config"]["pattern"] == "independent" and "k_gemv_stream" in d["roofline"]["kernel"], d["config"]
        dc = d["dependent_chain"]
        assert "error" not in dc and dc["value"] > 0 and "k_decode_chain" in dc["roofline"]["kernel"], dc
        if "--force-dist" not in extra:
            assert dc["verified"]["ok"], dc["verified"]
    if "--force-dist" not in extra:
        assert d["verified"]["ok"], d["verified"]
        tw = d["prefill_twin"]          # the N = 256 twin of the same matrices rides along in a one-GPU decode line
        assert "error" not in tw and tw["value"] > 0 and tw["dense_fp16_baseline"]["ms_per_step"] > 0, tw
    if path == "chain":
        assert d["roofline"]["headline_gemv"]["us"] > 0


def test_ggml_glue_decoder_segments(tm, tmp_path):
    """A two-layer llama-shaped loop through the ggml glue's decoder segments (include/ggml-tmac-hip.h): one launch per segment
    o -> [+ residual, RMSNorm] -> gate/up -> [silu(gate) * up] -> down -> [+ residual, RMSNorm] -> next q/k/v, an operator outside the
    hook (a device copy on the glue's stream, the stand-in for attention) between q/k/v and o.  The C++ program dumps every tensor;
    each mpGEMM is recomputed with the oracle from the inputs the run actually saw (2e-3 of max |C|: the element-wise operators are
    extensions specified to a tolerance), the residual stream with fp32 adds (bit for bit)."""
    from tmac_amd import convert
    H, F, bits, bm, NL, eps = 1024, 2816, 2, 128, 2, 1e-5
    d = str(tmp_path)
    rng = np.random.default_rng(21)
    names = ["q", "k", "v", "o", "gate", "up", "down"]
    shape = {"q": (H, H), "k": (H, H), "v": (H, H), "o": (H, H), "gate": (F, H), "up": (F, H), "down": (H, F)}
    mats = {}
    for l in range(NL):
        for n in names:
            Mw, K = shape[n]
            case = orc.make_case(1000 + 10 * l + names.index(n), Mw, K, bits=bits, fp16_values=True)
            c = 1.0 / np.sqrt(2.5 * K)
            sc = (case["sc"] * c).astype(np.float16).astype(np.float32)
            zr = (case["zr"] * c + ((2 ** bits - 1) / 2.0 - 2 ** (bits - 1)) * sc).astype(np.float16).astype(np.float32)
            A = orc.preprocess_weights(case["w"], bits, bm, 16)
            S = orc.preprocess_scales(sc, zr, bits, bm)
            np.concatenate([A.reshape(-1), S.astype(np.float32).view(np.uint8).reshape(-1)]).tofile(os.path.join(d, f"blob_{l}_{n}.bin"))
            mats[(l, n)] = (A, S, Mw, K)
    convert.write_kcfg(os.path.join(d, "kcfg.ini"), [[bits, H, H, 1, -1], [bits, F, H, 1, -1], [bits, H, F, 1, -1]],
                       bm={(bits, H, H): bm, (bits, F, H): bm, (bits, H, F): bm})
    h0 = rng.standard_normal(H).astype(np.float32)
    h0.tofile(os.path.join(d, "h0.bin")); h0.astype(np.float16).tofile(os.path.join(d, "x0.bin"))
    g = {}
    for l in range(NL):
        for k in (1, 2):
            g[(l, k)] = (1.0 + 0.1 * rng.standard_normal(H)).astype(np.float32)
            g[(l, k)].tofile(os.path.join(d, f"g{l}_{k}.bin"))
    exe = os.path.join(d, "ggml_segment_main")
    gxx(exe, os.path.join(ROOT, "tests", "cpp", "ggml_segment_main.cc"), os.path.join(ROOT, "src", "ggml_tmac_hip.cc"),
        extra=("-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"))
    env = dict(os.environ); env.pop("TMAC_KCFG_FILE", None)
    r = subprocess.run([exe, d, str(H), str(F), str(bits)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr

    def out(name, dt=np.float16):
        return np.fromfile(os.path.join(d, f"out_{name}.bin"), dtype=dt).astype(np.float32)

    def oracle(l, n, x):
        A, S, Mw, K = mats[(l, n)]
        q, ls, lb = orc.preprocessor(x[None, :].astype(np.float32), 64)
        return orc.qgemm_float(A, q, S, ls, lb, Mw, K, 1, bits, bm, 16, 128, 64, True)[0]

    def norm(t, gam):
        rs = np.float32(1.0) / np.sqrt(np.float32((t.astype(np.float64) ** 2).mean()) + np.float32(eps))
        return (t * rs).astype(np.float32) * gam

    def rel(a, b):
        return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))

    x1 = norm(h0, g[(0, 1)])                      # the first q/k/v takes the fp32 embedding as it is (ggml_tmac_hip_segment_mul_mat_f32)
    for n in ("q", "k", "v"):
        assert rel(out(f"0_{n}"), oracle(0, n, x1)) <= 2e-3, n
    hcur = h0
    for l in range(NL):
        a = out(f"attn{l}")
        assert np.array_equal(a, out(f"{l}_q"))                       # the outside operator ran between the launches
        o = out(f"{l}_o")
        assert rel(o, oracle(l, "o", a)) <= 2e-3
        t2 = o + hcur
        x2 = norm(t2, g[(l, 2)])
        gt, up = out(f"{l}_gate"), out(f"{l}_up")
        assert rel(gt, oracle(l, "gate", x2)) <= 2e-3 and rel(up, oracle(l, "up", x2)) <= 2e-3
        dn = out(f"{l}_down")
        m = (gt / (np.float32(1.0) + np.exp(-gt))).astype(np.float32) * up
        assert rel(dn, oracle(l, "down", m)) <= 2e-3
        if l + 1 < NL:
            t3 = dn + t2
            assert np.array_equal(out("h1", np.float32), t3)
            x3 = norm(t3, g[(l + 1, 1)])
            for n in ("q", "k", "v"):
                assert rel(out(f"{l + 1}_{n}"), oracle(l + 1, n, x3)) <= 2e-3, n
            hcur = t3
