cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu23.log 2>&1
timeout 600 python tools/bench_models.py > gpurun_out/models23.txt 2>&1
tail -4 gpurun_out/pytest_gpu23.log; tail -11 gpurun_out/models23.txt
