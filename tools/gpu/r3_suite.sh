# round 3: the host-pointer repro first (its own process, -s for the race count), then the WHOLE gpu suite without -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
timeout 600 python -m pytest tests/test_gpu_hostptr.py -q -m gpu -s > gpurun_out/r3/hostptr.log 2>&1
tail -6 gpurun_out/r3/hostptr.log
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r3/pytest_gpu.log 2>&1
tail -12 gpurun_out/r3/pytest_gpu.log
timeout 300 python bench.py > gpurun_out/r3/bench.json 2> gpurun_out/r3/bench.err
tail -c 1500 gpurun_out/r3/bench.json
