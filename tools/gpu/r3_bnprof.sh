cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bnp; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --workload bitnet-3b-prefill --no-cpu-baseline --no-verify > $O/b.json 2> $O/b.log
cd $R; python tools/rocprof_summary.py $O 2>&1 | cut -c1-200 | head -20
