// The ggml op-hook glue (src/ggml_tmac_hip.cc) against a minimal stand-in for ggml_tensor: upload a converted weight blob,
// mul_mat activations against it (N = 1 decode and N = 3), free.  usage: ggml_shim_main <dir with blob.bin x.bin ref.bin kcfg.ini> M K bits N [dev]
// with "dev" the activation and output tensors live in device memory (a device backend's buffers): the glue must pass them on unstaged.
// with "batch" (N = 1): THREE uploads of the blob stand for q / k / v; a hook without a view of the graph issues them as three separate
// device-resident calls (ggml_tmac_hip_mul_mat_dev) in deferred mode, then synchronises: the calls must have run as ONE stream-mode launch
// (tmac_hip_defer_stats), the recording cached from the second token on, every output equal to the reference; a fourth call that READS the
// first output must flush the queue by itself.
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
    if (argc > 6 && std::string(argv[6]) == "batch") {
        if (N != 1 || M != K) { fprintf(stderr, "batch mode: N = 1, M = K\n"); return 2; }
        tmac_ggml_tensor w2{{K, M, 1, 1}, blob.data(), nullptr}, w3{{K, M, 1, 1}, blob.data(), nullptr};
        if (ggml_tmac_hip_upload(&w2, bits) || ggml_tmac_hip_upload(&w3, bits)) return 5;
        void *xd = nullptr, *yq = nullptr, *yk = nullptr, *yv = nullptr, *y4 = nullptr;
        const size_t by = (size_t)M * sizeof(float);
        if (hipMalloc(&xd, xb.size()) || hipMalloc(&yq, by) || hipMalloc(&yk, by) || hipMalloc(&yv, by) || hipMalloc(&y4, by) || hipMemcpy(xd, xb.data(), xb.size(), 1)) return 7;
        const tmac_ggml_tensor* wq[1] = {&w}; const tmac_ggml_tensor* wk[1] = {&w2}; const tmac_ggml_tensor* wv[1] = {&w3};
        void* oq[1] = {yq}; void* ok_[1] = {yk}; void* ov[1] = {yv}; void* o4[1] = {y4};
        if (ggml_tmac_hip_set_deferred(1)) return 10;
        double worst = 0;
        const float* ref = (const float*)rb.data();
        for (int tok = 0; tok < 3; ++tok) {
            if (ggml_tmac_hip_mul_mat_dev(wq, 1, xd, 1, oq, 1) || ggml_tmac_hip_mul_mat_dev(wk, 1, xd, 1, ok_, 1) || ggml_tmac_hip_mul_mat_dev(wv, 1, xd, 1, ov, 1)) {
                fprintf(stderr, "mul_mat_dev: %s\n", ggml_tmac_hip_last_error()); return 6;
            }
            if (tok == 2 && ggml_tmac_hip_mul_mat_dev(wk, 1, yq, 1, o4, 1)) return 6;        // reads q's output: the three queued calls go first
            if (ggml_tmac_hip_synchronize()) { fprintf(stderr, "synchronize: %s\n", ggml_tmac_hip_last_error()); return 6; }
            for (void* yd : {yq, yk, yv}) {
                if (hipMemcpy(y.data(), yd, by, 2)) return 9;
                double mx = 0, err = 0;
                for (size_t i = 0; i < y.size(); ++i) { mx = std::fmax(mx, std::fabs(ref[i])); err = std::fmax(err, std::fabs(y[i] - ref[i])); }
                worst = std::fmax(worst, err / mx);
            }
        }
        uint64_t fl = 0, hits = 0, st = 0, single = 0;
        tmac_hip_defer_stats(&fl, &hits, &st, &single);
        ggml_tmac_hip_set_deferred(0);
        printf("RESULT worst_rel_err %.3g flushes %llu cache_hits %llu stream_launches %llu single_calls %llu\n", worst, (unsigned long long)fl,
               (unsigned long long)hits, (unsigned long long)st, (unsigned long long)single);
        ggml_tmac_hip_free(&w); ggml_tmac_hip_free(&w2); ggml_tmac_hip_free(&w3);
        // tokens 0..2: one stream-mode launch each (the recording cached after the first), + the dependent call of token 2 on its own
        return (worst <= 2e-5 && st == 3 && hits == 2 && single == 1 && fl == 4) ? 0 : 1;
    }
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
