cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python tools/bench_models.py > gpurun_out/models23.txt 2>&1; tail -11 gpurun_out/models23.txt
