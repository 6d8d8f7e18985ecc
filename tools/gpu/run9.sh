cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
python tools/stamps.py > gpurun_out/stamps9.log 2>&1; grep -E "==|medians|p90|duration|spread" gpurun_out/stamps9.log
