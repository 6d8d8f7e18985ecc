#!/bin/bash
# build_variant.sh NAME "-DTMAC_CHAIN_NLW=8 -DTMAC_CHAIN_NBW=7 ..." : a second build of the library with other chain settings, for A/B runs
# inside one gpurun call (tmac_amd/lib/ko/libtmac_hip_NAME.so; TMAC_HIP_LIB selects it).  Only the chain kernels and the chain's host
# file depend on these macros; the other objects are copied from the main build.
set -e
cd "$(dirname "$0")/../tmac_amd/csrc"
name=$1; cfg=$2
mkdir -p ../lib/ko build_$name
cp -u build/*.o build_$name/ 2>/dev/null || true
rm -f build_$name/tmac_chain_b*.o build_$name/tmac_chain_host.o build_$name/tmac_stream.o build_$name/tmac_stream_qw.o
make -s -j8 BUILD=build_$name OUT=../lib/ko/libtmac_hip_$name.so CHAIN_CFG="$cfg"
ls -la ../lib/ko/libtmac_hip_$name.so
