# round 4: the wave-specialised chain: chain tests (+ IPC test), then A/B of library builds on the decode workloads inside ONE call
# (boxes differ by up to 16 %).  LIBS="name=path ..." (default: the round-3 build kept under tmac_amd/lib/base against the tree's),
# WLS, TESTS=0 skips the tests, CFGS = extra environment per run ("VAR=val VAR=val" lines)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4
if [ "${TESTS:-1}" = "1" ]; then
  timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_chain_ipc.py -q -m gpu -x > gpurun_out/r4/chain_tests.log 2>&1
  tail -15 gpurun_out/r4/chain_tests.log
fi
out=gpurun_out/r4/ab.txt; : > $out
for wl in ${WLS:-llama2-7b-w2 bitnet-3b llama2-7b-w4}; do
  for lib in ${LIBS:-r3=tmac_amd/lib/base/libtmac_hip_r3.so new=tmac_amd/lib/libtmac_hip.so}; do
    name=${lib%%=*}; path=${lib#*=}
    while IFS= read -r cfg; do
      r=$(env TMAC_HIP_LIB=$PWD/$path $cfg timeout 300 python bench.py --workload $wl --no-cpu-baseline ${BENCH_EXTRA:---no-verify} 2>gpurun_out/r4/err_$name.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], (d.get('verified') or {}).get('ok'), d.get('activations_finite'))")
      echo "$wl $name $cfg -> $r" | tee -a $out
      [ -z "$r" ] && tail -5 gpurun_out/r4/err_$name.txt
    done <<< "${CFGS:-X=0}"
  done
done
