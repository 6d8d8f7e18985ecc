cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/divcheck.hip -o /tmp/divcheck 2>/dev/null && timeout 300 /tmp/divcheck > gpurun_out/divcheck14.txt 2>&1
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu14.log 2>&1
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench14.json 2> gpurun_out/bench14.err
timeout 300 python tools/tune_quad.py > gpurun_out/tune14.txt 2>&1
tail -4 gpurun_out/divcheck14.txt; tail -5 gpurun_out/pytest_gpu14.log; cut -c1-1200 gpurun_out/bench14.json; tail -5 gpurun_out/tune14.txt
