cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
# one-off wider sweep of the two randomised parity tests (other seeds, more cases; edits the box copy only)
sed -i 's/_random_configs(48, 20260924)/_random_configs(400, 31337)/; s/_random_fused(40, 99)/_random_fused(300, 4242)/' tests/test_gpu_parity.py
timeout 1500 python -m pytest tests -q -m gpu -k "random_" -x 2>&1 | tail -2
# and the A/B kernel variants on the first set
for v in 1 7; do echo "variant $v"; TMAC_FUZZ_VARIANT=$v timeout 900 python -m pytest tests -q -m gpu -k "random_configurations" 2>&1 | tail -1; done
for f in 1 2; do for v in 0 3; do echo "fast aggregation $f variant $v"; TMAC_FUZZ_FA=$f TMAC_FUZZ_VARIANT=$v timeout 900 python -m pytest tests -q -m gpu -k "random_configurations" 2>&1 | tail -3; done; done
