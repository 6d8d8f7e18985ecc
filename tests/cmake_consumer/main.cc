// What a T-MAC consumer does with the package: include the wrapper and the kernels header, construct the wrapper
// (loads kcfg.ini through the TMAC_KCFG_FILE compile definition), look a kernel configuration up, call a dispatcher.
// With arguments (a fixture directory written by tests/test_gpu_integration.py and a shape) it also computes a GEMV through
// the wrapper's host-pointer interface and compares it with the oracle's output.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>
#include "t-mac/kernels.h"
#include "t-mac/tmac_gemm_wrapper.h"
extern "C" int tmac_consumer_abi_version(void);

static std::vector<char> slurp(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
    TMAC::TMACGeMMWrapper<float> w(1, 64, "", "");
    if (argc >= 6) {
        const std::string d = argv[1];
        const int Mw = atoi(argv[2]), K = atoi(argv[3]), bits = atoi(argv[4]), bm = atoi(argv[5]);
        std::vector<char> A = slurp(d + "/A.bin"), S = slurp(d + "/S.bin"), xb = slurp(d + "/x.bin"), rb = slurp(d + "/ref.bin");
        const int ntile = Mw * bits / bm, rows = bm / bits;
        const size_t a_tile = (size_t)bm / 2 * (K / 4), s_tile = S.size() / sizeof(float) / ntile;
        std::vector<int8_t> qlut((size_t)K / 4 * 16);
        std::vector<float> ls(K / 64), lb(K / 64), C(Mw, -1.0f);
        w.llama_cpp_init(xb.data(), qlut.data(), ls.data(), lb.data(), Mw, K, 1, bits);
        for (int i = 0; i < ntile; ++i)
            w.llama_cpp_compute(A.data() + (size_t)i * a_tile, (float*)S.data() + (size_t)i * s_tile, qlut.data(), ls.data(), lb.data(),
                                C.data() + (size_t)i * rows, rows, K, 1, bits);
        const float* ref = (const float*)rb.data();
        double mx = 0, err = 0;
        for (int i = 0; i < Mw; ++i) { mx = std::fmax(mx, std::fabs(ref[i])); err = std::fmax(err, std::fabs(C[i] - ref[i])); }
        std::printf("gemv max rel err %.3g\n", err / mx);
        return err / mx <= 2e-5 ? 0 : 1;
    }
    TMAC::TMACGeMMConfig cfg = w.get_kcfg(4096, 11008, 1, 2);
    // no kernel is generated for this shape: the dispatcher's "no match" code (-1), or "no device" (-2) on a CPU box
    float dummy[4] = {0};
    int rc = preprocessor_int8(123, 456, 1, 2, dummy, dummy, dummy, dummy);
    std::printf("abi=%d bm=%d kfactor=%d n_tile_num=%d rc=%d version=%s\n", tmac_consumer_abi_version(), cfg.bm, cfg.kfactor,
                cfg.n_tile_num, rc, tmac_hip_version());
    return (cfg.bm == 128 && cfg.kfactor == 16 && cfg.n_tile_num == 64 && rc < 0) ? 0 : 1;
}
