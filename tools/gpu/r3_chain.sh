# round 3: the persistent chain with unified-scale / W1 / W3 ops: tests, then the three decode workloads
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_gpu_chain.py -q -m gpu -x > gpurun_out/r3/chain_tests.log 2>&1
tail -5 gpurun_out/r3/chain_tests.log
for wl in llama2-7b-w2 bitnet-3b llama2-7b-w4; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline $BENCH_EXTRA > gpurun_out/r3/bench_$wl.json 2> gpurun_out/r3/bench_$wl.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3/bench_$wl.json"))
    print("$wl", d["ms_per_step"], "ms", d["value"], d["unit"], "frac", d["roofline"]["frac"], "verified", d["verified"]["ok"], d["verified"]["max_rel_err"], "finite", d["activations_finite"])
    if "per_call_from_stamps" in d["roofline"]: print(json.dumps(d["roofline"]["per_call_from_stamps"]))
except Exception as e:
    print("$wl FAILED", e); print(open("gpurun_out/r3/bench_$wl.err").read()[-1500:])
PY
done
