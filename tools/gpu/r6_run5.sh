B="python bench.py --no-cpu-baseline --no-verify --no-decoder-pattern --no-stream-core --no-prefill-headline --steps 300"
for rep in 1 2; do for v in head imgonly issonly new; do
  if [ "$v" = new ]; then unset TMAC_HIP_LIB; else export TMAC_HIP_LIB=$PWD/tmac_amd/lib/ko/libtmac_hip_$v.so; fi
  for w in llama2-7b-w2 llama2-7b-w4; do
    echo -n "== $v $w: "; timeout 300 $B --workload $w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])"
  done
done; done
unset TMAC_HIP_LIB
