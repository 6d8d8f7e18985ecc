cd $GRAFT_REPO_ROOT
B="python bench.py --workload bitnet-3b --pattern independent --no-cpu-baseline --no-verify --no-decoder-pattern --no-prefill-headline --no-stream-core"
run() { echo "$1: $(env $1 timeout 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])")"; }
run X=0
run TMAC_STREAM_QW=0
run TMAC_STREAM_VISIT_ITEMS=80
run TMAC_STREAM_VISIT_ITEMS=320
run TMAC_STREAM_NCLS=8
run TMAC_STREAM_NCLS=4
run TMAC_STREAM_LPT=0
