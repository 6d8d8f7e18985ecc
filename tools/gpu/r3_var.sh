#!/bin/bash
# A/B of library builds (tmac_amd/lib/ko/libtmac_hip_v*.so, built by hand with a -D variant) on bench.py workloads; same box
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in 1 2; do for f in tmac_amd/lib/ko/libtmac_hip_v*.so; do
  for wl in ${WLS:-llama2-7b-w2-prefill}; do
  r=$(TMAC_HIP_LIB=$PWD/$f timeout 300 python bench.py --workload $wl --no-verify --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "$(basename $f) $wl -> $r"
done; done; done
