cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python tools/tune_quad.py 0 4 > gpurun_out/tune_w4.txt 2>&1; tail -4 gpurun_out/tune_w4.txt
