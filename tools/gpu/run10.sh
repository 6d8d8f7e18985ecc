cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=line 2>&1 | tail -12 > gpurun_out/pytest_gpu10.log
python tools/stamps.py > gpurun_out/stamps10.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench10_quad.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --variant 7 > gpurun_out/bench10_quad_mqsad.log 2>&1
tail -4 gpurun_out/pytest_gpu10.log; grep -E "==|medians|duration" gpurun_out/stamps10.log; tail -1 gpurun_out/bench10_quad.log | cut -c1-330; tail -1 gpurun_out/bench10_quad_mqsad.log | cut -c1-330
