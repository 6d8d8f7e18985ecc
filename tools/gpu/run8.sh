cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -30 > gpurun_out/pytest_gpu8.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench8_quad.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --variant 4 > gpurun_out/bench8_fused.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof8 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $GRAFT_REPO_ROOT/gpurun_out/prof8.log 2>&1
cd $GRAFT_REPO_ROOT
tail -8 gpurun_out/pytest_gpu8.log; tail -1 gpurun_out/bench8_quad.log | cut -c1-1300; tail -1 gpurun_out/bench8_fused.log | cut -c1-250; cut -c1-150 gpurun_out/prof8/*/*kernel_stats.csv | head -8
