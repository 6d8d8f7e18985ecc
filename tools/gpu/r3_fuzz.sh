#!/bin/bash
# wide randomised parity sweeps (one-off; other seeds / more cases than the suite runs): chain call sequences, the per-launch
# kernels' random configurations and fused launches
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/fuzz
TMAC_FUZZ_CHAINS=${CHAINS:-300} timeout 2400 python -m pytest tests/test_gpu_chain.py -q -m gpu -k "random_chains" -x 2>&1 | tail -5 | tee gpurun_out/fuzz/chains.txt
sed -i 's/_random_configs(48, 20260924)/_random_configs(400, 424242)/; s/_random_fused(40, 99)/_random_fused(300, 171717)/' tests/test_gpu_parity.py
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "random_" -x 2>&1 | tail -3 | tee gpurun_out/fuzz/parity.txt
