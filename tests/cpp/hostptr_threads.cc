// Drives the reference's wrapper interface -- TMAC::TMACGeMMWrapper<float>::llama_cpp_init on the main thread,
// llama_cpp_compute tile by tile from worker threads, all buffers host memory (include/t-mac/tmac_gemm_wrapper.h:170-228)
// -- through this repository's source-compatible header and libtmac_hip.so, and compares with the oracle's output.
// usage: hostptr_threads <dir with A.bin S.bin x.bin ref.bin kcfg.ini> Mw K bits bm nthreads
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "t-mac/tmac_gemm_wrapper.h"

static std::vector<char> slurp(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
    if (argc < 7) return 2;
    const std::string d = argv[1];
    const int Mw = atoi(argv[2]), K = atoi(argv[3]), bits = atoi(argv[4]), bm = atoi(argv[5]), nth = atoi(argv[6]);
    std::vector<char> A = slurp(d + "/A.bin"), S = slurp(d + "/S.bin"), xb = slurp(d + "/x.bin"), rb = slurp(d + "/ref.bin");
    const float* ref = (const float*)rb.data();
    TMAC::TMACGeMMWrapper<float> wr(nth, 64, d + "/kcfg.ini", "");
    wr.set_workspace(K, 1);
    const TMAC::TMACGeMMConfig cfg = wr.get_kcfg(Mw, K, 1, bits);
    if (cfg.bm != bm) { fprintf(stderr, "kcfg bm %d != %d\n", cfg.bm, bm); return 3; }
    const int ntile = Mw * bits / bm, rows = bm / bits;
    const size_t a_tile = (size_t)bm / 2 * (K / 4), s_tile = S.size() / sizeof(float) / ntile;
    std::vector<int8_t> qlut((size_t)K / 4 * 16);
    std::vector<float> ls(K / 64), lb(K / 64), C(Mw);
    double worst = 0, last_us = 0;
    for (int pass = 0; pass < 4; ++pass) {      // pass 0 registers the tiles, pass 1 groups them into a run, 2 and 3 take the fast path
        std::fill(C.begin(), C.end(), -1.0f);
        auto t0 = std::chrono::steady_clock::now();
        wr.llama_cpp_init(xb.data(), qlut.data(), ls.data(), lb.data(), Mw, K, 1, bits);
        std::vector<std::thread> th;
        for (int t = 0; t < nth; ++t)
            th.emplace_back([&, t]() {
                for (int i = t; i < ntile; i += nth)     // llama.cpp deals the tiles out to its threads
                    wr.llama_cpp_compute(A.data() + (size_t)i * a_tile, (float*)S.data() + (size_t)i * s_tile, qlut.data(), ls.data(), lb.data(),
                                         C.data() + (size_t)i * rows, rows, K, 1, bits);
            });
        for (auto& x : th) x.join();
        last_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        double mx = 0, err = 0;
        for (int i = 0; i < Mw; ++i) { mx = std::fmax(mx, std::fabs(ref[i])); err = std::fmax(err, std::fabs(C[i] - ref[i])); }
        worst = std::fmax(worst, err / mx);
        printf("pass %d: %.1f us per GEMV (preprocessor + %d tile calls from %d threads), max rel err %.3g\n", pass, last_us, ntile, nth, err / mx);
    }
    {   // steady-state cost of the route from one caller thread (no thread start-up in the timing): 50 GEMVs, new activations each
        float* xf = (float*)xb.data();
        double init_us = 0;
        auto t0 = std::chrono::steady_clock::now();
        for (int rep = 0; rep < 50; ++rep) {
            xf[rep % K] += 0.25f;
            auto ti0 = std::chrono::steady_clock::now();
            wr.llama_cpp_init(xb.data(), qlut.data(), ls.data(), lb.data(), Mw, K, 1, bits);
            init_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ti0).count();
            for (int i = 0; i < ntile; ++i)
                wr.llama_cpp_compute(A.data() + (size_t)i * a_tile, (float*)S.data() + (size_t)i * s_tile, qlut.data(), ls.data(), lb.data(),
                                     C.data() + (size_t)i * rows, rows, K, 1, bits);
        }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 50;
        printf("TIMING %.1f us per GEMV from one thread (preprocessor call + %d tile calls, host pointers, PCIe both ways); preprocessor call %.1f us, tile calls %.1f us\n",
               us, ntile, init_us / 50, us - init_us / 50);
        for (int rep = 0; rep < 50; ++rep) xf[rep % K] -= 0.25f;
        wr.llama_cpp_init(xb.data(), qlut.data(), ls.data(), lb.data(), Mw, K, 1, bits);
        for (int i = 0; i < ntile; ++i)
            wr.llama_cpp_compute(A.data() + (size_t)i * a_tile, (float*)S.data() + (size_t)i * s_tile, qlut.data(), ls.data(), lb.data(),
                                 C.data() + (size_t)i * rows, rows, K, 1, bits);
    }
    // a model "reloaded" at the same addresses: other weight bytes behind the same pointers must not be served from the cache
    for (size_t i = 0; i < A.size(); i += 97) A[i] = (char)(A[i] ^ 0x5a);
    std::vector<float> C2(Mw, -1.0f);
    wr.llama_cpp_init(xb.data(), qlut.data(), ls.data(), lb.data(), Mw, K, 1, bits);
    for (int i = 0; i < ntile; ++i)
        wr.llama_cpp_compute(A.data() + (size_t)i * a_tile, (float*)S.data() + (size_t)i * s_tile, qlut.data(), ls.data(), lb.data(),
                             C2.data() + (size_t)i * rows, rows, K, 1, bits);
    int changed = 0;
    for (int i = 0; i < Mw; ++i) changed += C2[i] != C[i];
    printf("after editing the weights in place: %d of %d outputs changed\n", changed, Mw);
    printf("RESULT worst_rel_err %.3g last_us %.1f changed %d\n", worst, last_us, changed);
    return (worst <= 2e-5 && changed > Mw / 2) ? 0 : 1;
}
