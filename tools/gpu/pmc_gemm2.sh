# SQ / LDS / TA / L2 counters of k_gemm_planes on the llama-2-7B prefill shapes (tools/bench_gemm2.py), one --pmc pass per counter group
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_gemm2; rm -rf $O; mkdir -p $O
BITS=${BITS:-2}
run() { tag=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$tag -- python $R/tools/bench_gemm2.py 256 $BITS > $O/$tag.log 2>&1; }
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_ADDR_CONFLICT
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES
run act SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_SALU
run mem TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE
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
for t in lds sq act mem; do tail -2 $O/$t.log | cut -c1-300; done
cat $O/summary.txt | cut -c1-600
