# round-6 evidence at HEAD: full GPU suite log, smoke, bench lines of every workload and pattern, rocprofv3 kernel trace + stats of the bench
# command (stream mode = the line's value; the dependent chain; BitNet; W4; prefill), FETCH_SIZE and SQ counters in separate --pmc passes
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06prof; rm -rf $O; mkdir -p $O
cd $R; timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
cd /tmp
timeout 900 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 900 python $R/bench.py --steps 20 --warmup 5 > $O/bench_default_steps20.json 2> $O/bench_default_steps20.err
for w in llama2-7b-w4 bitnet-3b llama2-7b-w2-prefill llama2-7b-w4-prefill bitnet-3b-prefill; do
  timeout 600 python $R/bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err
done
timeout 300 python $R/bench.py --pattern chained --no-cpu-baseline --no-stream-core > $O/bench_chained.json 2> $O/bench_chained.err
timeout 300 python $R/bench.py --pattern decoder --no-cpu-baseline > $O/bench_decoder_pattern.json 2> $O/bench_decoder_pattern.err
SB=10 NL=96 timeout 300 python $R/tools/bench_stream.py 4096x11008 4096x4096 11008x4096x2 4096x4096x3 4096x11008:4 11008x4096x2:4 4096x11008:3 4096x11008:1 > $O/bench_stream.txt 2>&1
SB=10 NL=32 timeout 300 python $R/tools/bench_stream.py 4096x11008 4096x4096 11008x4096x2 4096x4096x3 > $O/bench_stream_32calls.txt 2>&1
TMAC_STREAM_QW=0 SB=10 NL=96 timeout 300 python $R/tools/bench_stream.py 4096x11008 4096x4096 11008x4096x2 4096x4096x3 > $O/bench_stream_quad64.txt 2>&1
TMAC_STREAM_NCLS=1 TMAC_STREAM_QW=0 SB=10 NL=96 timeout 300 python $R/tools/bench_stream.py 4096x11008 4096x4096 11008x4096x2 4096x4096x3 > $O/bench_stream_round5_form.txt 2>&1
B="python $R/bench.py --no-cpu-baseline --no-verify --no-decoder-pattern --no-stream-core --no-prefill-headline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_stream -- $B --pattern independent > $O/trace_stream.json 2> $O/trace_stream.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_chain -- $B --pattern chained > $O/trace_chain.json 2> $O/trace_chain.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bitnet -- $B --pattern independent --workload bitnet-3b > $O/trace_bitnet.json 2> $O/trace_bitnet.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_w4 -- $B --pattern independent --workload llama2-7b-w4 > $O/trace_w4.json 2> $O/trace_w4.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_prefill -- $B --workload llama2-7b-w2-prefill > $O/trace_prefill.json 2> $O/trace_prefill.log
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_stream -- $B --pattern independent --steps 5 --warmup 2 > $O/fetch_stream.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_chain -- $B --pattern chained --steps 5 --warmup 2 > $O/fetch_chain.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_bitnet -- $B --pattern independent --steps 5 --warmup 2 --workload bitnet-3b > $O/fetch_bitnet.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_bitnet_chain -- $B --pattern chained --steps 5 --warmup 2 --workload bitnet-3b > $O/fetch_bitnet_chain.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_w4 -- $B --pattern independent --steps 5 --warmup 2 --workload llama2-7b-w4 > $O/fetch_w4.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_w4_chain -- $B --pattern chained --steps 5 --warmup 2 --workload llama2-7b-w4 > $O/fetch_w4_chain.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_prefill -- $B --steps 3 --warmup 1 --workload llama2-7b-w2-prefill > $O/fetch_prefill.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/sq1_stream -- $B --pattern independent --steps 5 --warmup 2 > $O/sq1_stream.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq2_stream -- $B --pattern independent --steps 5 --warmup 2 > $O/sq2_stream.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/sq1_chain -- $B --pattern chained --steps 5 --warmup 2 > $O/sq1_chain.log 2>&1
cd $R
python tools/rocprof_summary.py $O > $O/summary.txt 2>&1
for t in trace_stream trace_chain trace_bitnet trace_w4 trace_prefill; do echo "== bench line of the $t run:" >> $O/summary.txt; cut -c1-400 $O/$t.json >> $O/summary.txt; echo >> $O/summary.txt; done
find $O -name "*.csv" -size +2M -delete
find $O -name "*.db" -delete 2>/dev/null
cut -c1-260 $O/summary.txt
cat $O/bench_stream.txt $O/bench_stream_32calls.txt $O/bench_stream_quad64.txt $O/bench_stream_round5_form.txt | grep -v amdgpu
