# critical-path view (tools/chain_crit.py, tools/chain_stamps.py) of the chain for the workloads given in WLS
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4
for wl in ${WLS:-llama2-7b-w2}; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-verify --no-decoder-pattern --stamps $BENCH_EXTRA > gpurun_out/r4/stamps_$wl.json 2>gpurun_out/r4/stamps_err.txt
  cp gpurun_out/chain_stamps.npy gpurun_out/r4/chain_stamps_$wl.npy
  echo "== $wl $(python -c "import json; print(json.load(open('gpurun_out/r4/stamps_$wl.json'))['ms_per_step'])") ms (with stamps)"
  python tools/chain_crit.py gpurun_out/r4/chain_stamps_$wl.npy
  python tools/chain_stamps.py gpurun_out/r4/chain_stamps_$wl.npy 2>/dev/null | tail -8
done
