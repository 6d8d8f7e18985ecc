#!/bin/bash
# ab_stream.sh "variant ..." [shapes]: tools/bench_stream.py per library build (tmac_amd/lib/ko/libtmac_hip_<variant>.so; "new" = the tree's build), twice, interleaved
vars=$1; shift
SH=${*:-"4096x11008 11008x4096x2 4096x4096x3 4096x4096"}
for rep in 1 2; do
  for v in $vars; do
    if [ "$v" = new ]; then unset TMAC_HIP_LIB; else export TMAC_HIP_LIB=$PWD/tmac_amd/lib/ko/libtmac_hip_$v.so; fi
    echo "== $v"; python tools/bench_stream.py $SH 2>&1 | grep -v "Warn\|amdgpu.ids"
  done
done
