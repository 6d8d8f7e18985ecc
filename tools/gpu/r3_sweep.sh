# sweep of chain knobs (read once per chain from the environment); CFG lines: "VAR=val VAR=val"
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
out=gpurun_out/r3/sweep.txt; : > $out
run() {  # workload, env...
  wl=$1; shift
  r=$(env "$@" timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "$wl $* -> $r" | tee -a $out
}
for wl in ${WLS:-llama2-7b-w2 bitnet-3b}; do
  while IFS= read -r cfg; do
    [ -z "$cfg" ] && continue
    run $wl $cfg
  done <<< "$CFGS"
done
