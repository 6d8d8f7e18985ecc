cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; cut -c1-200 gpurun_out/bench_final.json
