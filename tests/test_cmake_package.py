"""find_package(TMAC): a consumer project configured against cmake/TMACConfig.cmake builds, links libtmac_hip.so and
reads the kcfg through the TMAC_KCFG_FILE compile definition.  CPU only."""
import os
import shutil
import subprocess

import pytest

import tmac_amd
from tmac_amd import convert

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("cmake") is None, reason="cmake not installed")
def test_consumer_builds_and_runs(tmp_path):
    tmac_amd.lib()   # makes sure the library is built
    kcfg = str(tmp_path / "kcfg.ini")
    convert.write_kcfg(kcfg, convert.PRESET_KERNELS["llama-2-7b-2bit"],
                       bm={(2, 4096, 4096): 128, (2, 11008, 4096): 128, (2, 4096, 11008): 128})
    build = str(tmp_path / "build")
    gen = ["-G", "Ninja"] if shutil.which("ninja") else []
    subprocess.run(["cmake", "-S", os.path.join(ROOT, "tests", "cmake_consumer"), "-B", build, *gen,
                    f"-DTMAC_DIR={os.path.join(ROOT, 'cmake')}", f"-DTMAC_KCFG={kcfg}", "-DCMAKE_CXX_COMPILER=g++"],
                   check=True, capture_output=True, timeout=300)
    subprocess.run(["cmake", "--build", build], check=True, capture_output=True, timeout=300)
    env = dict(os.environ)
    env.pop("TMAC_KCFG_FILE", None)
    r = subprocess.run([os.path.join(build, "consumer")], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bm=128 kfactor=16 n_tile_num=64" in r.stdout
