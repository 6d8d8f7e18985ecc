cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
./tools/ubench > gpurun_out/ubench.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench2_graph.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph > gpurun_out/bench2_eager.log 2>&1
cat gpurun_out/ubench.log; tail -1 gpurun_out/bench2_graph.log | cut -c1-1200; tail -1 gpurun_out/bench2_eager.log | cut -c1-300
