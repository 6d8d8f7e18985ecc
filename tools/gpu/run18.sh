cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu18.log 2>&1
timeout 300 python tools/tune_quad.py > gpurun_out/tune18.txt 2>&1
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench18.json 2> gpurun_out/bench18.err
tail -3 gpurun_out/pytest_gpu18.log; tail -4 gpurun_out/tune18.txt; cut -c1-330 gpurun_out/bench18.json
