cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python tools/tune_quad.py 0 2 > gpurun_out/tune_w2.txt 2>&1; tail -4 gpurun_out/tune_w2.txt
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | cut -c100-330
