cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
AMD_LOG_LEVEL=2 timeout 600 python -m pytest tests -q -m gpu -k "zoo and 14336-2-128" -x > gpurun_out/dbg27.log 2>&1
grep -i "error\|invalid\|:1:\|:2:" gpurun_out/dbg27.log | head -20
