# round 4: chain tests (+ IPC) and the default bench lines of the decode workloads; LIBS / CFGS as in r4_chain.sh
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4
if [ "${TESTS:-1}" = "1" ]; then
  timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_chain_ipc.py tests/test_gpu_gemm_planes.py -q -m gpu -x > gpurun_out/r4/tests.log 2>&1
  tail -5 gpurun_out/r4/tests.log
fi
for wl in ${WLS:-llama2-7b-w2}; do
  timeout 600 python bench.py --workload $wl $BENCH_EXTRA > gpurun_out/r4/bench_$wl.json 2> gpurun_out/r4/bench_$wl.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r4/bench_$wl.json"))
    r = d["roofline"]
    print("$wl", d["ms_per_step"], "ms frac", r["frac"], "verified", d["verified"], "| headline_gemv", (r.get("headline_gemv") or {}).get("us"), (r.get("headline_gemv") or {}).get("frac"),
          "| stream_core", (r.get("stream_core") or {}).get("us_per_gemv"), (r.get("stream_core") or {}).get("frac"), "| cpu", d.get("cpu_baseline"))
except Exception as e:
    print("$wl FAILED", e); print(open("gpurun_out/r4/bench_$wl.err").read()[-1500:])
PY
done
