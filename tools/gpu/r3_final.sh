# what the driver runs at round end, at HEAD: the GPU suite (-x, as the driver), smoke(), the default bench; logs -> gpurun_out/r3final
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3final; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.json
