# SQ / LDS counters of k_gemm_planes on the llama-2-7B prefill shapes (tools/bench_gemm2.py), one --pmc pass per counter group
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_gemm2; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/lds -- python $R/tools/bench_gemm2.py > $O/lds.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/sq -- python $R/tools/bench_gemm2.py > $O/sq.log 2>&1
cd $R
python - <<'PY' > $O/summary.txt 2>&1
import csv, glob, os, sys
from collections import defaultdict
csv.field_size_limit(1 << 30)
root = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/pmc_gemm2")
for d in sorted(os.listdir(root)):
    if not os.path.isdir(os.path.join(root, d)):
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_gemm_planes" in r["Kernel_Name"]:
                acc[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", d)
    for g, cs in sorted(acc.items(), key=lambda kv: int(kv[0])):
        print("  grid", g, " ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(cs.items())), " launches", max(len(v) for v in cs.values()))
PY
find $O -name "*.csv" -size +1M -delete
cat $O/summary.txt | cut -c1-400
