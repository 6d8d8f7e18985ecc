cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short  2>&1 | tail -30 > gpurun_out/pytest_gpu5.log
python tools/stamps.py > gpurun_out/stamps5.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench5_fused.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --variant 5 > gpurun_out/bench5_fused_mqsad.log 2>&1
tail -6 gpurun_out/pytest_gpu5.log; grep -A7 "== " gpurun_out/stamps5.log | grep -E "==|medians|duration"; tail -1 gpurun_out/bench5_fused.log | cut -c1-1300; tail -1 gpurun_out/bench5_fused_mqsad.log | cut -c1-250
