export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_sb
rm -rf $O; mkdir -p $O
cd /tmp
SB=10 NL=32 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/tools/bench_stream.py 4096x11008 > $O/log.txt 2>&1
t=$(find $O -name "*kernel_trace.csv" | head -1); python - "$t" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if 'k_gemv_stream' in r['Kernel_Name'] or 'k_lut_images' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
prev=None
for r in rows[-24:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print(r['Kernel_Name'][:30], 'dur', (e-s)/1e3, 'gap', None if prev is None else (s-prev)/1e3)
    prev=e
PY
find $O -name "*kernel_trace.csv" -delete
