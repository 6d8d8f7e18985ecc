# A/B of TMAC_CHAIN_RING_BEHIND (the rest of the weight ring issued behind the first poll) inside one call: chain tests, then the decode workloads
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4b; mkdir -p $O
if [ "${TESTS:-1}" = "1" ]; then timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_chain_ipc.py tests/test_gpu_chain_xform.py -q -m gpu -x > $O/behind_tests.log 2>&1; tail -4 $O/behind_tests.log; fi
for rep in 1 2; do for wl in ${WLS:-llama2-7b-w2 llama2-7b-w4 bitnet-3b}; do for v in 0 1; do
  r=$(TMAC_CHAIN_RING_BEHIND=$v $EXTRA_ENV timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-verify --no-stream-core --no-decoder-pattern 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "$wl ring_behind=$v -> $r" | tee -a $O/behind.txt
done; done; done
