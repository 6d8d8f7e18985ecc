cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
tail -8 gpurun_out/pytest_gpu.log
