# round 4 (second session): new coverage (unified-scale prefill for 1-4 bit, transform hazards, glue stream), the BitNet prefill line
# (the row-wise LUT build gained an output), and a poll-knob sweep of the decode chain at HEAD
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm_planes.py tests/test_gpu_chain_xform.py tests/test_gpu_integration.py -q -m gpu > $O/cover_tests.log 2>&1; tail -15 $O/cover_tests.log
timeout 300 python bench.py --workload bitnet-3b-prefill --no-cpu-baseline > $O/bench_bitnet_prefill.json 2> $O/bench_bitnet_prefill.err; cut -c1-300 $O/bench_bitnet_prefill.json; echo
: > $O/sweep.txt
for cfg in "X=0" "TMAC_CHAIN_POLL_DELAY=0" "TMAC_CHAIN_POLL_DELAY=8" "TMAC_CHAIN_POLL_SLEEP=4" "TMAC_CHAIN_POLL_SLEEP=16" "TMAC_CHAIN_POLL_DELAY=2 TMAC_CHAIN_POLL_SLEEP=4" "X=1"; do
  for wl in llama2-7b-w2 bitnet-3b; do
    r=$(env $cfg timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-verify --no-stream-core --no-decoder-pattern 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "$wl $cfg -> $r" | tee -a $O/sweep.txt
  done
done
