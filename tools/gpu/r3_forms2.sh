#!/bin/bash
# bench.py on the prefill workloads with each workgroup form of k_gemm_planes forced (2 / 3) and the default (0); same box
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/forms
for wl in ${WLS:-llama2-7b-w2-prefill llama2-7b-w4-prefill}; do for k in ${KS:-2 3 0 2 3}; do
  r=$(timeout 300 python bench.py --workload $wl --no-verify --no-cpu-baseline --gemm-kernel $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])")
  echo "$wl gemm-kernel $k -> $r" | tee -a gpurun_out/forms/bench.txt
done; done
