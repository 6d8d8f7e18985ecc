// What a T-MAC consumer does with the package: include the wrapper and the kernels header, construct the wrapper
// (loads kcfg.ini through the TMAC_KCFG_FILE compile definition), look a kernel configuration up, call a dispatcher.
#include <cstdio>
#include "t-mac/kernels.h"
#include "t-mac/tmac_gemm_wrapper.h"
extern "C" int tmac_consumer_abi_version(void);
int main() {
    TMAC::TMACGeMMWrapper<float> w(1, 64, "", "");
    TMAC::TMACGeMMConfig cfg = w.get_kcfg(4096, 11008, 1, 2);
    // no kernel is generated for this shape: the dispatcher's "no match" code (-1), or "no device" (-2) on a CPU box
    float dummy[4] = {0};
    int rc = preprocessor_int8(123, 456, 1, 2, dummy, dummy, dummy, dummy);
    std::printf("abi=%d bm=%d kfactor=%d n_tile_num=%d rc=%d version=%s\n", tmac_consumer_abi_version(), cfg.bm, cfg.kfactor,
                cfg.n_tile_num, rc, tmac_hip_version());
    return (cfg.bm == 128 && cfg.kfactor == 16 && cfg.n_tile_num == 64 && rc < 0) ? 0 : 1;
}
