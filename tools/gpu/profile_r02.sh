# round-2 profiles of the timed configuration: kernel trace + stats, FETCH_SIZE, SQ counters (separate --pmc passes,
# kernel-trace only), for the persistent chain and for the per-launch path; summaries -> gpurun_out/r02prof/*.txt
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02prof; rm -rf $O; mkdir -p $O
B="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-verify"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_chain -- $B > $O/trace_chain.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fused -- $B --path fused > $O/trace_fused.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_chain -- $B > $O/fetch_chain.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_fused -- $B --path fused --no-graph > $O/fetch_fused.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/sq1_chain -- $B > $O/sq1_chain.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq2_chain -- $B > $O/sq2_chain.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_prefill -- python $R/bench.py --workload llama2-7b-w2-prefill --steps 2 --warmup 1 --no-verify > $O/trace_prefill.log 2>&1
cd $R
python tools/rocprof_summary.py $O > $O/summary.txt 2>&1
find $O -name "*.csv" -size +2M -delete      # keep the merged directory small: the summaries carry what is needed
cat $O/summary.txt | cut -c1-260
