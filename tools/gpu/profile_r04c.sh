# evidence refresh at the final HEAD (after the issue-split change of the decode chain): GPU suite log, the decode bench lines, kernel traces
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04cprof; rm -rf $O; mkdir -p $O
cd $R; timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log | head -1
cd /tmp
for w in llama2-7b-w2 llama2-7b-w4 bitnet-3b; do timeout 600 python $R/bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; done
B="python $R/bench.py --no-cpu-baseline --no-verify --no-decoder-pattern --no-stream-core"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_chain -- $B > $O/trace_chain.json 2> $O/trace_chain.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bitnet -- $B --workload bitnet-3b > $O/trace_bitnet.json 2> $O/trace_bitnet.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_w4 -- $B --workload llama2-7b-w4 > $O/trace_w4.json 2> $O/trace_w4.log
cd $R
python tools/rocprof_summary.py $O > $O/summary.txt 2>&1
for t in trace_chain trace_bitnet trace_w4; do echo "== bench line of the $t run:" >> $O/summary.txt; cut -c1-400 $O/$t.json >> $O/summary.txt; echo >> $O/summary.txt; done
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete 2>/dev/null
grep "k_decode_chain" $O/summary.txt | cut -c1-230
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); r=d['roofline']
print(d['ms_per_step'], d['value'], d['unit'], 'frac', r['frac'], 'verified', (d.get('verified') or {}).get('ok'), 'headline', (r.get('headline_gemv') or {}).get('us'), 'stream_core', (r.get('stream_core') or {}).get('us_per_gemv'), 'decoder', (r.get('decoder_pattern') or {}).get('ms_per_token'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))"; done
