# same-box A/B of library builds on the prefill LINES (bench.py: distinct weights per layer, streamed from HBM): VARS="base new ..."
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in $VARS; do
  if [ "$v" = new ]; then unset TMAC_HIP_LIB; else export TMAC_HIP_LIB=$PWD/tmac_amd/lib/ko/libtmac_hip_$v.so; fi
  for w in ${WL:-llama2-7b-w2-prefill llama2-7b-w4-prefill}; do
    timeout 300 python bench.py --workload $w --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-8s %-22s' % ('$v', '$w'), d['ms_per_step'], 'ms  dense fp16', (r.get('dense_fp16_baseline') or {}).get('ms_per_step'))"
  done
done; done
