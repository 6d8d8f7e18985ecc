B="python bench.py --no-cpu-baseline --no-verify --no-decoder-pattern --no-stream-core --no-prefill-headline --pattern independent"
for sw in "20 5" "20 5" "20 50" "20 200" "100 5" "300 5" "20 5"; do set -- $sw
  echo -n "steps $1 warmup $2: "; timeout 300 $B --steps $1 --warmup $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['event_ms_per_step'])"
done
B="python bench.py --no-cpu-baseline --no-verify --no-decoder-pattern --no-stream-core --no-prefill-headline --pattern chained"
for sw in "20 5" "20 200" "300 5"; do set -- $sw
  echo -n "chained steps $1 warmup $2: "; timeout 300 $B --steps $1 --warmup $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['event_ms_per_step'])"
done
