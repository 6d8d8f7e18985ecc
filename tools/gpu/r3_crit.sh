# critical-path view (tools/chain_crit.py) of the chain for the workloads given in WLS
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
for wl in ${WLS:-bitnet-3b llama2-7b-w2}; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-verify --stamps > gpurun_out/r3/stamps_$wl.json 2>/dev/null
  cp gpurun_out/chain_stamps.npy gpurun_out/r3/chain_stamps_$wl.npy
  echo "== $wl $(python -c "import json; print(json.load(open('gpurun_out/r3/stamps_$wl.json'))['ms_per_step'])") ms (with stamps)"
  python tools/chain_crit.py gpurun_out/r3/chain_stamps_$wl.npy
  python tools/chain_stamps.py gpurun_out/r3/chain_stamps_$wl.npy 2>/dev/null | tail -8
done
