cd $GRAFT_REPO_ROOT
for w in llama2-7b-w2-prefill llama2-7b-w4-prefill; do for gk in 0 2 3; do
  echo "$w gemm-kernel $gk: $(timeout 300 python bench.py --workload $w --no-cpu-baseline --no-verify --gemm-kernel $gk 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
done; done
VARS="new vpipe1 vpipe5" bash tools/gpu/r6_prefill_ab2.sh
