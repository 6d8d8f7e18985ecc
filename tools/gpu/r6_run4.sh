timeout 900 python -m pytest tests/test_gpu_stream.py -x -q 2>&1 | tail -3
for rep in 1 2; do for v in prev new; do
  if [ "$v" = new ]; then unset TMAC_HIP_LIB; else export TMAC_HIP_LIB=$PWD/tmac_amd/lib/ko/libtmac_hip_$v.so; fi
  echo "== $v"; SB=10 NL=96 timeout 300 python tools/bench_stream.py 4096x11008 11008x4096x2 4096x4096x3 2>&1 | grep -v "Warn\|amdgpu.ids"
  echo "== $v QW=1"; TMAC_STREAM_QW=1 SB=10 NL=96 timeout 300 python tools/bench_stream.py 11008x4096x2 2>&1 | grep -v "Warn\|amdgpu.ids"
done; done
unset TMAC_HIP_LIB
