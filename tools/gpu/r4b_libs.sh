# A/B of library builds (TMAC_HIP_LIB) x env settings on bench.py decode workloads; LIBS = space-separated .so paths ("main" = the in-tree build), CFGS = newline-separated env lines
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4b; mkdir -p $O
for rep in 1 2; do for wl in ${WLS:-llama2-7b-w2}; do for lib in ${LIBS:-main}; do
  while IFS= read -r cfg; do
    L=""; [ "$lib" != "main" ] && L="TMAC_HIP_LIB=$PWD/$lib"
    r=$(env $L $cfg timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-verify --no-stream-core --no-decoder-pattern $BENCH_EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "$wl $(basename $lib) $cfg -> $r" | tee -a $O/libs.txt
  done <<< "${CFGS:-X=0}"
done; done; done
