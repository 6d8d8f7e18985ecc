// The ggml op-hook glue (src/ggml_tmac_hip.cc) against a minimal stand-in for ggml_tensor: upload a converted weight blob,
// mul_mat activations against it (N = 1 decode and N = 3), free.  usage: ggml_shim_main <dir with blob.bin x.bin ref.bin kcfg.ini> M K bits N [dev]
// with "dev" the activation and output tensors live in device memory (a device backend's buffers): the glue must pass them on unstaged.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "ggml-tmac-hip.h"
#include "tmac_hip.h"

// (the HIP runtime comes with libtmac_hip.so; three calls of it, declared by hand: this file is built with g++ without HIP headers)
extern "C" int hipMalloc(void**, size_t);
extern "C" int hipMemcpy(void*, const void*, size_t, int);
extern "C" int hipFree(void*);

static std::vector<char> slurp(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
    if (argc < 6) return 2;
    const std::string d = argv[1];
    const int M = atoi(argv[2]), K = atoi(argv[3]), bits = atoi(argv[4]), N = atoi(argv[5]);
    std::vector<char> blob = slurp(d + "/blob.bin"), xb = slurp(d + "/x.bin"), rb = slurp(d + "/ref.bin");
    if (ggml_tmac_hip_init((d + "/kcfg.ini").c_str(), 0)) { fprintf(stderr, "init: %s\n", ggml_tmac_hip_last_error()); return 3; }
    tmac_ggml_tensor w{{K, M, 1, 1}, blob.data(), nullptr};
    if (!ggml_tmac_hip_can_mul_mat(&w, bits)) { fprintf(stderr, "no kcfg entry\n"); return 4; }
    if (ggml_tmac_hip_upload(&w, bits)) { fprintf(stderr, "upload: %s\n", ggml_tmac_hip_last_error()); return 5; }
    std::vector<float> y((size_t)N * M, -1.0f);
    const bool dev = argc > 6 && std::string(argv[6]) == "dev";
    void *xd = nullptr, *yd = nullptr;
    if (dev) {
        if (hipMalloc(&xd, xb.size()) || hipMalloc(&yd, y.size() * sizeof(float)) || hipMemcpy(xd, xb.data(), xb.size(), 1)) return 7;
        if (tmac_hip_pointer_on_device(xd) != 1 || tmac_hip_pointer_on_device(xb.data()) != 0) { fprintf(stderr, "pointer classification\n"); return 8; }
    }
    tmac_ggml_tensor x{{K, N, 1, 1}, dev ? xd : (void*)xb.data(), nullptr}, dst{{M, N, 1, 1}, dev ? yd : (void*)y.data(), nullptr};
    double worst = 0;
    for (int rep = 0; rep < 3; ++rep) {
        if (ggml_tmac_hip_mul_mat(&w, &x, &dst)) { fprintf(stderr, "mul_mat: %s\n", ggml_tmac_hip_last_error()); return 6; }
        if (dev && hipMemcpy(y.data(), yd, y.size() * sizeof(float), 2)) return 9;
        const float* ref = (const float*)rb.data();
        double mx = 0, err = 0;
        for (size_t i = 0; i < y.size(); ++i) { mx = std::fmax(mx, std::fabs(ref[i])); err = std::fmax(err, std::fabs(y[i] - ref[i])); }
        worst = std::fmax(worst, err / mx);
    }
    ggml_tmac_hip_free(&w);
    if (dev) { hipFree(xd); hipFree(yd); }
    printf("RESULT worst_rel_err %.3g\n", worst);
    return worst <= 2e-5 ? 0 : 1;
}
