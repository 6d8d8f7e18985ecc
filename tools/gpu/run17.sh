cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python tools/stamps.py > gpurun_out/stamps17.txt 2>&1
cat gpurun_out/stamps17.txt | tail -30
