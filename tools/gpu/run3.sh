cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40 > gpurun_out/pytest_gpu3.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench3_fused.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --path split > gpurun_out/bench3_split.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof3 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $GRAFT_REPO_ROOT/gpurun_out/prof3.log 2>&1
cd $GRAFT_REPO_ROOT; tail -8 gpurun_out/pytest_gpu3.log; tail -1 gpurun_out/bench3_fused.log | cut -c1-1500; tail -1 gpurun_out/bench3_split.log | cut -c1-200
