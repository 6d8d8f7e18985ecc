cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python tools/bench_prefill.py 256 > gpurun_out/prefill16.txt 2>&1
cat gpurun_out/prefill16.txt | tail -8
