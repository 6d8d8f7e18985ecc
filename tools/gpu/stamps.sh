cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python tools/stamps.py > gpurun_out/stamps_x.txt 2>&1; grep -v "kernel span\|block start" gpurun_out/stamps_x.txt | tail -20
