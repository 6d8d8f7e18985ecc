# round-2 profiles, part b (prefill on k_gemm_planes + the other workloads): bench JSON per workload, kernel trace + stats and
# SQ counters of the prefill run, one-rank run of the multi-GPU code path -> gpurun_out/r02b/
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02b; rm -rf $O; mkdir -p $O
for w in llama2-7b-w2-prefill llama2-7b-w4-prefill llama2-7b-w4 bitnet-3b; do
  timeout 300 python $R/bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err
done
timeout 300 python $R/bench.py --workload llama2-7b-w2-prefill --force-dist > $O/bench_prefill_dist1.json 2> $O/bench_prefill_dist1.err
timeout 300 python $R/bench.py --workload llama2-7b-w2-prefill --force-dist --comm lib > $O/bench_prefill_dist1_lib.json 2> $O/bench_prefill_dist1_lib.err
P="python $R/bench.py --workload llama2-7b-w2-prefill --steps 3 --warmup 1 --no-verify"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_prefill -- $P > $O/trace_prefill.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/sq1_prefill -- $P > $O/sq1_prefill.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --output-format csv -d $O/sq2_prefill -- $P > $O/sq2_prefill.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_prefill -- $P --no-graph > $O/fetch_prefill.log 2>&1
cd $R
python tools/rocprof_summary.py $O > $O/summary.txt 2>&1
find $O -name "*.csv" -size +2M -delete
cat $O/summary.txt | cut -c1-300
for f in $O/bench_*.json; do echo $f; cut -c1-260 $f; echo; done
