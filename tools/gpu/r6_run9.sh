timeout 900 python -m pytest tests/test_gpu_gemm_planes.py -x -q 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-verify"
for rep in 1 2; do for v in vpacked new; do
  if [ "$v" = new ]; then unset TMAC_HIP_LIB; else export TMAC_HIP_LIB=$PWD/tmac_amd/lib/ko/libtmac_hip_$v.so; fi
  for w in llama2-7b-w2-prefill llama2-7b-w4-prefill bitnet-3b-prefill; do
    echo -n "== $v $w: "; timeout 300 $B --workload $w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['dense_fp16_baseline']['ms_per_step'])"
  done
done; done
unset TMAC_HIP_LIB
