cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -6 > gpurun_out/rocminfo.txt 2>&1
nproc >> gpurun_out/rocminfo.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -60 > gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench1.log 2>&1
timeout 300 python bench.py --steps 5 --warmup 2 --variant 2 --no-cpu-baseline > gpurun_out/bench1_sdwa.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof1.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/prof1 -name "*stats*" | head; tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/bench1.log | tail -3
