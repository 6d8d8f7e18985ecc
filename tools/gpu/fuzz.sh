cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
# one-off wider sweep of the two randomised parity tests (other seeds, more cases)
sed -i 's/_random_configs(48, 20260924)/_random_configs(400, 31337)/; s/_random_fused(40, 99)/_random_fused(300, 4242)/' tests/test_gpu_parity.py
timeout 1500 python -m pytest tests -q -m gpu -k "random_" -x 2>&1 | tail -12
