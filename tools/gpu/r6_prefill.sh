# prefill A/B: parity of the plane-combined GEMM, per-shape times, the three prefill lines
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6_prefill; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm_planes.py -x -q -m gpu 2>&1 | tail -4
for b in 2 4; do timeout 200 python tools/bench_gemm2.py 256 $b 2>&1 | grep -v "^N =" ; done
for w in llama2-7b-w2-prefill llama2-7b-w4-prefill; do timeout 300 python bench.py --workload $w --no-cpu-baseline > $O/$w.json 2>$O/$w.err; python - <<PY
import json
d=json.loads(open("$O/$w.json").read().strip().splitlines()[-1])
r=d.get("roofline",{})
print("$w", d["ms_per_step"], "ms", d["value"], d["unit"], "dense", (r.get("dense_fp16_baseline") or {}).get("ms_per_step"))
PY
done
