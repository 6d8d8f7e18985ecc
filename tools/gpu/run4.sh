cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
python tools/stamps.py > gpurun_out/stamps.log 2>&1; cat gpurun_out/stamps.log
