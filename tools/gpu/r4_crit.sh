# critical-path / stage view (tools/chain_crit.py, chain_stages.py) of the chain for the workloads given in WLS; BENCH_EXTRA adds bench.py flags
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4
for wl in ${WLS:-llama2-7b-w2}; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-verify --stamps $BENCH_EXTRA > gpurun_out/r4/stamps_$wl.json 2>gpurun_out/r4/stamps_err.txt
  cp gpurun_out/chain_stamps.npy gpurun_out/r4/chain_stamps_$wl.npy
  echo "== $wl $(python -c "import json; print(json.load(open('gpurun_out/r4/stamps_$wl.json'))['ms_per_step'])") ms (with stamps)"
  python tools/chain_crit.py gpurun_out/r4/chain_stamps_$wl.npy; python tools/chain_stages.py gpurun_out/r4/chain_stamps_$wl.npy
done
