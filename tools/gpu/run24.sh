cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out; rm -rf gpurun_out/prof24 gpurun_out/pmc24
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke24.log 2>&1
timeout 900 python bench.py > gpurun_out/bench24.json 2> gpurun_out/bench24.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof24 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof24.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc24 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-kernel-events > $GRAFT_REPO_ROOT/gpurun_out/pmc24.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 300 python tools/tune_quad.py > gpurun_out/tune24.txt 2>&1
timeout 300 python tools/stamps.py > gpurun_out/stamps24.txt 2>&1
tail -2 gpurun_out/smoke24.log; cat gpurun_out/bench24.json | cut -c1-2600; cut -c1-160 gpurun_out/prof24/*/*kernel_stats.csv | head -6; ls gpurun_out/pmc24/*/
