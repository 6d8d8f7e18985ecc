cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu -k "onehot or prefill or random_fused" 2>&1 | tail -2
timeout 600 python tools/bench_prefill.py 256 4 2>&1 | grep -v amdgpu | cut -c1-140
