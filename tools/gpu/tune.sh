cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu20.log 2>&1
timeout 300 python tools/tune_quad.py > gpurun_out/tune20.txt 2>&1
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench20.json 2> gpurun_out/bench20.err
timeout 600 python tools/stamps.py > gpurun_out/stamps20.txt 2>&1
tail -3 gpurun_out/pytest_gpu20.log; tail -4 gpurun_out/tune20.txt; cut -c1-330 gpurun_out/bench20.json; grep -A3 "== down" gpurun_out/stamps20.txt
