cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
python tools/tune_quad.py 0 > gpurun_out/tune_quad_mfma.log 2>&1
python tools/tune_quad.py 7 > gpurun_out/tune_quad_mqsad.log 2>&1
python tools/tune_quad.py 4 > gpurun_out/tune_fused.log 2>&1
cat gpurun_out/tune_quad_mfma.log gpurun_out/tune_quad_mqsad.log gpurun_out/tune_fused.log | grep -v amdgpu.ids
