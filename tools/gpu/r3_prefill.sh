cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_gpu_gemm_planes.py -q -m gpu -x 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "prefill or onehot" 2>&1 | tail -2
for lib in "" ${PREV:+/root/repo/tmac_amd/lib/libtmac_hip_prev.so}; do
  for wl in llama2-7b-w2-prefill llama2-7b-w4-prefill; do
    r=$(TMAC_HIP_LIB=$lib timeout 300 python bench.py --workload $wl --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])")
    echo "$wl lib=${lib:-current} -> $r"
  done
done
python tools/bench_gemm2.py 2>/dev/null | tail -8
