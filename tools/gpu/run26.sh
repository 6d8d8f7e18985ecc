cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for cap in 384 512 640 704 768 1024; do echo "cap $cap" >> gpurun_out/cap26.txt; TMAC_QUAD_CAP=$cap timeout 300 python tools/tune_quad.py 2>&1 | grep "qkv\|gate_up" | cut -c1-75 >> gpurun_out/cap26.txt; done
cat gpurun_out/cap26.txt
