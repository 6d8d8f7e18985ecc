# round-3 prefill evidence after the second workgroup form of k_gemm_planes: bench lines, kernel trace + stats, FETCH_SIZE pass
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03pre; rm -rf $O; mkdir -p $O
for w in llama2-7b-w2-prefill llama2-7b-w4-prefill; do
  timeout 600 python $R/bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err
done
B="python $R/bench.py --no-cpu-baseline --no-verify --workload llama2-7b-w2-prefill"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_prefill -- $B > $O/trace_prefill.json 2> $O/trace_prefill.log
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_prefill -- $B --steps 4 --warmup 1 > $O/fetch_prefill.log 2>&1
cd $R
python tools/rocprof_summary.py $O > $O/summary.txt 2>&1
echo "== bench line of the trace_prefill run:" >> $O/summary.txt; cut -c1-400 $O/trace_prefill.json >> $O/summary.txt
find $O -name "*.csv" -size +2M -delete
find $O -name "*.db" -delete 2>/dev/null
cut -c1-260 $O/summary.txt
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); r=d['roofline']
print(d['ms_per_step'], d['value'], d['unit'], 'frac', r['frac'], 'verified', (d.get('verified') or {}).get('ok'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))"; done
