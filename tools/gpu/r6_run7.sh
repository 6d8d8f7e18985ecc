export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_bn
rm -rf $O; mkdir -p $O
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-verify --no-decoder-pattern --no-stream-core --no-prefill-headline --pattern independent --steps 100"
for w in bitnet-3b llama2-7b-w2; do
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -- $B --workload $w > $O/$w.json 2> $O/$w.log
f=$(find $O/$w -name "*kernel_stats.csv" | head -1); echo "== $w"; python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'tmac' in r['Name'] and 'retile' not in r['Name']:
        print(r['Name'][:70], 'calls', r['Calls'], 'avg_us', float(r['AverageNs'])/1e3, 'min', float(r['MinNs'])/1e3)
PY
cut -c1-200 $O/$w.json
done
find $O -name "*kernel_trace.csv" -delete
