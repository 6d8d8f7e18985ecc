# round-5 evidence at HEAD: full GPU suite log, smoke, bench lines of every workload and pattern, rocprofv3 kernel trace + stats of the default
# bench command and of the stream-mode launch (with the bench line of THAT run next to it), FETCH_SIZE and SQ counters in separate --pmc passes
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05prof; rm -rf $O; mkdir -p $O
cd $R; timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
cd /tmp
for w in llama2-7b-w2 llama2-7b-w4 bitnet-3b llama2-7b-w2-prefill llama2-7b-w4-prefill bitnet-3b-prefill; do
  timeout 600 python $R/bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err
done
timeout 300 python $R/bench.py --steps 20 --warmup 5 > $O/bench_default_steps20.json 2> $O/bench_default_steps20.err
timeout 300 python $R/bench.py --pattern decoder --no-cpu-baseline > $O/bench_decoder_pattern.json 2> $O/bench_decoder_pattern.err
timeout 300 python $R/bench.py --pattern independent --no-cpu-baseline --no-stream-core > $O/bench_independent_pattern.json 2> $O/bench_independent_pattern.err
timeout 300 python $R/tools/bench_stream.py 4096x11008 4096x4096 11008x4096x2 4096x4096x3 4096x11008:4 11008x4096x2:4 4096x11008:3 4096x11008:1 > $O/bench_stream.txt 2>&1
TMAC_CHAIN_STREAM=0 timeout 300 python $R/tools/bench_stream.py 4096x11008 4096x4096 11008x4096x2 4096x4096x3 > $O/bench_stream_as_chain.txt 2>&1
B="python $R/bench.py --no-cpu-baseline --no-verify --no-decoder-pattern --no-stream-core --no-prefill-headline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_chain -- $B > $O/trace_chain.json 2> $O/trace_chain.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_stream -- $B --pattern independent --steps 200 > $O/trace_stream.json 2> $O/trace_stream.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bitnet -- $B --workload bitnet-3b > $O/trace_bitnet.json 2> $O/trace_bitnet.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_w4 -- $B --workload llama2-7b-w4 > $O/trace_w4.json 2> $O/trace_w4.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_prefill -- $B --workload llama2-7b-w2-prefill > $O/trace_prefill.json 2> $O/trace_prefill.log
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_chain -- $B --steps 5 --warmup 2 > $O/fetch_chain.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_stream -- $B --pattern independent --steps 5 --warmup 2 > $O/fetch_stream.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_bitnet -- $B --steps 5 --warmup 2 --workload bitnet-3b > $O/fetch_bitnet.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_w4 -- $B --steps 5 --warmup 2 --workload llama2-7b-w4 > $O/fetch_w4.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_prefill -- $B --steps 3 --warmup 1 --workload llama2-7b-w2-prefill > $O/fetch_prefill.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/sq1_chain -- $B --steps 5 --warmup 2 > $O/sq1_chain.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq2_chain -- $B --steps 5 --warmup 2 > $O/sq2_chain.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/sq1_stream -- $B --pattern independent --steps 5 --warmup 2 > $O/sq1_stream.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq2_stream -- $B --pattern independent --steps 5 --warmup 2 > $O/sq2_stream.log 2>&1
cd $R
python tools/rocprof_summary.py $O > $O/summary.txt 2>&1
for t in trace_chain trace_stream trace_bitnet trace_w4 trace_prefill; do echo "== bench line of the $t run:" >> $O/summary.txt; cut -c1-400 $O/$t.json >> $O/summary.txt; echo >> $O/summary.txt; done
find $O -name "*.csv" -size +2M -delete
find $O -name "*.db" -delete 2>/dev/null
cut -c1-260 $O/summary.txt
cat $O/bench_stream.txt $O/bench_stream_as_chain.txt | grep -v amdgpu
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); r=d['roofline']
print(d['ms_per_step'], d['value'], d['unit'], 'frac', r['frac'], 'verified', (d.get('verified') or {}).get('ok'), 'headline', (r.get('headline_gemv') or {}).get('us'), 'stream_core', (r.get('stream_core') or {}).get('us_per_gemv'), 'indep', (r.get('independent_pattern') or {}).get('ms_per_token'), 'decoder', (r.get('decoder_pattern') or {}).get('ms_per_token'), 'dense', (r.get('dense_fp16_baseline') or {}).get('ms_per_step'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))"; done
