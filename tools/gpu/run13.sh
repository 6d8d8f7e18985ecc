cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke13.log 2>&1
timeout 900 python bench.py > gpurun_out/bench13.json 2> gpurun_out/bench13.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof13 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof13.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc13 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-kernel-events > $GRAFT_REPO_ROOT/gpurun_out/pmc13.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/smoke13.log; cat gpurun_out/bench13.json | cut -c1-2600; cut -c1-160 gpurun_out/prof13/*/*kernel_stats.csv | head -6
