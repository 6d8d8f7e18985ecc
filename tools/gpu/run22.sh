cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python tools/bench_models.py > gpurun_out/models22.txt 2>&1
tail -14 gpurun_out/models22.txt
