cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
# GEMV-loop against one-hot GEMM at small N (where tmac_hip_set_gemm_min_n's default of 32 comes from)
for n in 2 4 8 16 32; do timeout 300 python tools/bench_prefill.py $n 2>&1 | grep "int8 TOP" | sed -e 's/(.*TFLOP\/s)//' | cut -c1-130; done | tee gpurun_out/small_n.txt
