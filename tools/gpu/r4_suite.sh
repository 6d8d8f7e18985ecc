# round 4: the whole GPU suite (log kept), then the default bench lines of every workload
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4/suite; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
for w in ${WLS:-llama2-7b-w2}; do
  timeout 900 python bench.py --workload $w $BENCH_EXTRA > $O/bench_$w.json 2> $O/bench_$w.err
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_$w.json")); r = d["roofline"]
    print("$w", d["ms_per_step"], d["value"], d["unit"], "frac", r["frac"], "verified", (d.get("verified") or {}).get("ok"), "| headline", (r.get("headline_gemv") or {}).get("us"),
          "| stream_core", (r.get("stream_core") or {}).get("us_per_gemv"), "| decoder", (r.get("decoder_pattern") or {}).get("ms_per_token"), "| cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("unit"))
except Exception as e:
    print("$w FAILED", e); print(open("$O/bench_$w.err").read()[-1500:])
PY
done
