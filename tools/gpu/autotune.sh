cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu -k "autotuner or fast_aggregation" 2>&1 | tail -5
timeout 600 python tools/autotune_shapes.py 2>&1 | tee gpurun_out/autotune_shapes.txt
for f in "" "--autotune"; do for i in 1 2; do timeout 900 python bench.py --no-cpu-baseline $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'], d['config']['autotune'])"; done; done
