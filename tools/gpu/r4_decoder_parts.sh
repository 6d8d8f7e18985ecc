# where a decoder segment's time goes: the decoder pattern with all / no / only-NORM / only-GLU transforms, the no-transform segments through
# the kernel instance that knows transforms (TMAC_HIP_CHAIN_FORCE_XF=1), and the synthetic chain through that instance
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4/decparts; rm -rf $O; mkdir -p $O
WLS=${WLS:-llama2-7b-w2}
for wl in $WLS; do
run() {  # name, env..., -- bench args
  name=$1; shift
  env "$@" python bench.py --workload $wl --no-cpu-baseline --no-verify --no-stream-core --no-decoder-pattern --steps 50 --warmup 5 $BARGS > $O/$wl.$name.json 2> $O/$wl.$name.err
  python - "$O/$wl.$name.json" "$wl $name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "->", d["ms_per_step"], d.get("config", {}).get("pattern"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
BARGS="--pattern decoder"
run all   TMAC_BENCH_DECODER_XF=all
run none  TMAC_BENCH_DECODER_XF=none
run nonefx TMAC_BENCH_DECODER_XF=none TMAC_HIP_CHAIN_FORCE_XF=1
run norm  TMAC_BENCH_DECODER_XF=norm
run glu   TMAC_BENCH_DECODER_XF=glu
run all2  TMAC_BENCH_DECODER_XF=all
BARGS="--pattern chained"
run chained X=0
run chainedfx TMAC_HIP_CHAIN_FORCE_XF=1
done
