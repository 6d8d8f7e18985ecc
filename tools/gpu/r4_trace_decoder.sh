# kernel trace of the decoder pattern: where a segment launch's time goes (kernel durations vs the gaps between them)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4/trace_decoder; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/bench.py --pattern decoder --no-cpu-baseline --no-verify --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.log
cd $R
python - <<'PY'
import csv, glob, os, collections
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r4/trace_decoder")
for f in glob.glob(O + "/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_decode_chain" in row["Name"] or "copy" in row["Name"].lower()[:60]:
            print({k: (row[k][:70] if k == "Name" else row[k]) for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs") if k in row})
for f in glob.glob(O + "/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "k_decode_chain" in r["Kernel_Name"] and "true>(" in r["Kernel_Name"].replace(" ", "")]
    if not idx:
        print("no transform-chain kernels in the trace"); continue
    seg = rows[idx[len(idx) // 2]: idx[len(idx) // 2] + 67] if len(idx) > 200 else rows[idx[0]:idx[-1] + 1]
    print("kernels in the window:", len(seg))
    t0 = int(seg[0]["Start_Timestamp"])
    for a_, b_ in list(zip(seg, seg[1:]))[:12]:
        print("  %-28s start %8.2f us dur %7.2f us | gap to next %6.2f us" % (a_["Kernel_Name"][:28], (int(a_["Start_Timestamp"]) - t0) / 1e3,
              (int(a_["End_Timestamp"]) - int(a_["Start_Timestamp"])) / 1e3, (int(b_["Start_Timestamp"]) - int(a_["End_Timestamp"])) / 1e3))
    d = collections.defaultdict(list)
    for r in seg: d[r["Kernel_Name"][:28]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in d.items(): print("  %-28s n=%d mean %.2f us" % (k, len(v), sum(v) / len(v)))
    gaps = [(int(b_["Start_Timestamp"]) - int(a_["End_Timestamp"])) / 1e3 for a_, b_ in zip(seg, seg[1:])]
    print("  mean gap %.2f us, total span %.1f us" % (sum(gaps) / len(gaps), (int(seg[-1]["End_Timestamp"]) - t0) / 1e3))
PY
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete 2>/dev/null; true
