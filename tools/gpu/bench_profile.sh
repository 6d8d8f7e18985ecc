cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out; rm -rf gpurun_out/prof_final gpurun_out/pmc_final
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_final.log 2>&1
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_final -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_final.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_final -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-kernel-events > $GRAFT_REPO_ROOT/gpurun_out/pmc_final.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 300 python tools/tune_quad.py > gpurun_out/tune_final.txt 2>&1
timeout 300 python tools/stamps.py > gpurun_out/stamps_final.txt 2>&1
tail -2 gpurun_out/smoke_final.log; cat gpurun_out/bench_final.json | cut -c1-2600; cut -c1-160 gpurun_out/prof_final/*/*kernel_stats.csv | head -6; ls gpurun_out/pmc_final/*/
