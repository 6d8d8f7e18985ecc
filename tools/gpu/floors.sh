cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 ./tools/ubench graph 2>&1 | tee gpurun_out/floors.txt
