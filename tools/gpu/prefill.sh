cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -k "onehot or multi_row or prefill" > gpurun_out/pytest_gpu21.log 2>&1
timeout 600 python tools/bench_prefill.py 256 > gpurun_out/prefill21.txt 2>&1
tail -5 gpurun_out/pytest_gpu21.log; tail -7 gpurun_out/prefill21.txt
