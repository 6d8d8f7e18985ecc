cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=line 2>&1 | tail -8 > gpurun_out/pytest_gpu12.log
for v in 0 7 4; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --variant $v > gpurun_out/bench12_v$v.log 2>&1; done
tail -3 gpurun_out/pytest_gpu12.log; for v in 0 7 4; do tail -1 gpurun_out/bench12_v$v.log | cut -c1-180; done
