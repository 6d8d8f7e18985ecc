timeout 900 python -m pytest tests/test_gpu_stream.py -x -q 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-verify --no-decoder-pattern --no-stream-core --no-prefill-headline --pattern independent --steps 300"
for rep in 1 2; do for l in 0 1; do for w in llama2-7b-w2 bitnet-3b llama2-7b-w4; do
  echo -n "== LPT=$l $w: "; TMAC_STREAM_LPT=$l timeout 300 $B --workload $w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])"
done; done; done
