#!/bin/bash
# flakiness soak: the GPU suite several times in a row, then its files in reverse order (round 2 lost a third of its evidence to one
# order-dependent test)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/soak; : > gpurun_out/soak/soak.txt
for i in 1 2 3; do
  timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|error" | tail -2 | tee -a gpurun_out/soak/soak.txt
done
files=$(ls tests/test_gpu_*.py | sort -r | tr '\n' ' ')
timeout 900 python -m pytest $files -q -m gpu -x 2>&1 | grep -E "passed|failed|error" | tail -2 | tee -a gpurun_out/soak/soak.txt
