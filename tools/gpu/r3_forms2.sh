#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/forms
for wl in llama2-7b-w2-prefill llama2-7b-w4-prefill bitnet-3b-prefill; do for k in 2 3 0 2 3; do
  r=$(timeout 300 python bench.py --workload $wl --no-verify --no-cpu-baseline --gemm-kernel $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])")
  echo "$wl gemm-kernel $k -> $r" | tee -a gpurun_out/forms/bench.txt
done; done
