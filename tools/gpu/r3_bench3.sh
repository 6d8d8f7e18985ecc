# the three decode workloads with everything the default line carries (verification, headline GEMV, cpu baseline)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
for wl in llama2-7b-w2 bitnet-3b llama2-7b-w4; do
  timeout 600 python bench.py --workload $wl > gpurun_out/r3/full_$wl.json 2> gpurun_out/r3/full_$wl.err || tail -5 gpurun_out/r3/full_$wl.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r3/full_$wl.json"))
print("$wl", d["ms_per_step"], "ms frac", d["roofline"]["frac"], "headline", d["roofline"].get("headline_gemv"), "cpu", d["cpu_baseline"].get("value"), d["cpu_baseline"].get("cores"), d["cpu_baseline"].get("kind"))
PY
done
