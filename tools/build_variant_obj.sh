#!/bin/bash
# build_variant_obj.sh NAME OBJ "-DFLAG=..." : a second build of the library in which ONE object (e.g. tmac_gemm2) is compiled with extra
# flags; tmac_amd/lib/ko/libtmac_hip_vNAME.so, selected by TMAC_HIP_LIB (tools/gpu/r3_var.sh A/Bs every v*.so inside one gpurun call)
set -e
cd "$(dirname "$0")/../tmac_amd/csrc"
name=$1; obj=$2; cfg=$3
mkdir -p ../lib/ko build_v$name
cp -u build/*.o build_v$name/
rm -f build_v$name/$obj*.o
make -s -j8 BUILD=build_v$name OUT=../lib/ko/libtmac_hip_v$name.so CHAIN_CFG="$cfg"
ls -la ../lib/ko/libtmac_hip_v$name.so
