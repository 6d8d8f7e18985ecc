# same-box A/B of library builds on the prefill shapes: VARS="base new vX ..." (tmac_amd/lib/ko/libtmac_hip_<v>.so; new = the tree's build)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in $VARS; do
  if [ "$v" = new ]; then unset TMAC_HIP_LIB; else export TMAC_HIP_LIB=$PWD/tmac_amd/lib/ko/libtmac_hip_$v.so; fi
  for b in ${BITSET:-2 4}; do echo "$v W$b: $(timeout 200 python tools/bench_gemm2.py 256 $b 2>&1 | grep -E 'gemm alone|qkv|gate_up' | sed -E 's/\(.*//; s/LUT image \+ gemm//; s/ +/ /g' | tr '\n' '|')"; done
done; done
