#!/bin/bash
# the two workgroup forms of k_gemm_planes: parity tests, then per-shape times (same box)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/forms; rm -f gpurun_out/forms/shapes.txt
timeout 900 python -m pytest tests/test_gpu_gemm_planes.py -q -m gpu -x 2>&1 | tail -15 | tee gpurun_out/forms/tests.txt
for b in ${BITSET:-2 4}; do for k in ${KS:-2 3 0}; do
  timeout 300 python tools/bench_gemm2.py 256 $b $k 2>&1 | grep -v "^$" >> gpurun_out/forms/shapes.txt
done; done
cat gpurun_out/forms/shapes.txt
