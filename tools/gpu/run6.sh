cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -15 > gpurun_out/pytest_gpu6.log
timeout 900 python bench.py > gpurun_out/bench6.json 2> gpurun_out/bench6.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof6 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof6.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc6 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-kernel-events > $GRAFT_REPO_ROOT/gpurun_out/pmc6.log 2>&1
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/pytest_gpu6.log; cat gpurun_out/bench6.json | cut -c1-2500; ls gpurun_out/prof6/* gpurun_out/pmc6/* | head
