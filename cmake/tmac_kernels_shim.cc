// TMAC_KERNELS_SOURCE of the MI355X package: the reference hands its consumer a generated kernels.cc to compile into
// itself (CMakeLists.txt:123-128 of microsoft/T-MAC); the kernels of this implementation are HIP code inside
// libtmac_hip.so, so the consumer-compiled source only pins the ABI version it was built against.
#include "tmac_hip.h"
extern "C" int tmac_consumer_abi_version(void) { return TMAC_HIP_ABI_VERSION; }
