# sweep of the chain's hand-off knobs (read once per chain from the environment)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
out=gpurun_out/r3/sweep.txt; : > $out
run() {  # workload, env...
  wl=$1; shift
  r=$(env "$@" timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "$wl $* -> $r" | tee -a $out
}
for wl in ${WLS:-llama2-7b-w2 bitnet-3b llama2-7b-w4}; do
  for big in 1000 72 40 20; do
    run $wl TMAC_CHAIN_BIG_OP_KB=$big
  done
  run $wl TMAC_CHAIN_BIG_OP_KB=72 TMAC_CHAIN_POLL_DELAY=8
  run $wl TMAC_CHAIN_BIG_OP_KB=72 TMAC_CHAIN_POLL_DELAY=0 TMAC_CHAIN_POLL_SLEEP=4
  run $wl TMAC_CHAIN_ISSUE_FIRST=4 TMAC_CHAIN_POLL_DELAY=24 TMAC_CHAIN_POLL_SLEEP=16
done
