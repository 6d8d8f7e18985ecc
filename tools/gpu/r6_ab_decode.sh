# same-box A/B of library builds on a decode line: VARS="new x ..." (tmac_amd/lib/ko/libtmac_hip_<v>.so), WL=workload, PAT=pattern
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in $VARS; do
  if [ "$v" = new ]; then unset TMAC_HIP_LIB; else export TMAC_HIP_LIB=$PWD/tmac_amd/lib/ko/libtmac_hip_$v.so; fi
  echo "$v: $(timeout 200 python bench.py --workload ${WL:-bitnet-3b} --pattern ${PAT:-independent} --no-cpu-baseline --no-decoder-pattern --no-prefill-headline --no-stream-core 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], (d.get('verified') or {}).get('ok'), (d.get('verified') or {}).get('max_rel_err'))")"
done; done
