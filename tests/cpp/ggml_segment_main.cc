// Decoder segments through the ggml glue (include/ggml-tmac-hip.h): a two-layer llama-shaped loop -- first q/k/v, then per layer ONE
// launch of o -> [+ residual, RMSNorm] -> gate/up -> [silu(gate) * up] -> down (-> [+ residual, RMSNorm] -> next q/k/v), with an
// operator outside the hook between q/k/v and o (a device copy on the glue's stream: the stand-in for attention).  Every tensor is
// dumped; tests/test_gpu_integration.py recomputes each stage with the oracle.
// usage: ggml_segment_main <dir> H F bits      (dir: kcfg.ini, blob_<l>_<name>.bin, h0.bin (fp32 [H]), x0.bin (fp16 [H]), g<l>_{1,2}.bin (fp32 [H]))
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "ggml-tmac-hip.h"

extern "C" int hipMalloc(void**, size_t);
extern "C" int hipMemcpy(void*, const void*, size_t, int);
extern "C" int hipMemcpyAsync(void*, const void*, size_t, int, void*);
extern "C" int hipMemset(void*, int, size_t);
extern "C" int hipStreamCreateWithFlags(void**, unsigned);

static std::vector<char> slurp(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void* dev(const std::vector<char>& h) {
    void* d = nullptr;
    if (hipMalloc(&d, h.size()) || hipMemcpy(d, h.data(), h.size(), 1)) { fprintf(stderr, "device upload failed\n"); exit(7); }
    return d;
}
static void* dzero(size_t n) {
    void* d = nullptr;
    if (hipMalloc(&d, n) || hipMemset(d, 0, n)) { fprintf(stderr, "device allocation failed\n"); exit(7); }
    return d;
}
static void dump(const std::string& p, const void* d, size_t n) {
    std::vector<char> h(n);
    if (hipMemcpy(h.data(), d, n, 2)) { fprintf(stderr, "download failed\n"); exit(8); }
    std::ofstream(p, std::ios::binary).write(h.data(), (std::streamsize)n);
}
#define CK(x) do { if ((x)) { fprintf(stderr, "%s: %s\n", #x, ggml_tmac_hip_last_error()); return 6; } } while (0)

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    const std::string d = argv[1];
    const int H = atoi(argv[2]), F = atoi(argv[3]), bits = atoi(argv[4]);
    const int NL = 2;
    if (ggml_tmac_hip_init((d + "/kcfg.ini").c_str(), 0)) { fprintf(stderr, "init: %s\n", ggml_tmac_hip_last_error()); return 3; }
    const char* names[7] = {"q", "k", "v", "o", "gate", "up", "down"};
    const int rows[7] = {H, H, H, H, F, F, H}, cols[7] = {H, H, H, H, H, H, F};
    std::vector<std::vector<char>> blobs(NL * 7);
    tmac_ggml_tensor w[NL][7];
    for (int l = 0; l < NL; ++l)
        for (int m = 0; m < 7; ++m) {
            blobs[l * 7 + m] = slurp(d + "/blob_" + std::to_string(l) + "_" + names[m] + ".bin");
            w[l][m] = tmac_ggml_tensor{{cols[m], rows[m], 1, 1}, blobs[l * 7 + m].data(), nullptr};
            if (!ggml_tmac_hip_can_mul_mat(&w[l][m], bits)) { fprintf(stderr, "no kcfg entry for %s\n", names[m]); return 4; }
            CK(ggml_tmac_hip_upload(&w[l][m], bits));
        }
    float* h0 = (float*)dev(slurp(d + "/h0.bin"));
    void* x0 = dev(slurp(d + "/x0.bin"));
    float* g[NL][2];
    for (int l = 0; l < NL; ++l)
        for (int k = 0; k < 2; ++k) g[l][k] = (float*)dev(slurp(d + "/g" + std::to_string(l) + "_" + std::to_string(k + 1) + ".bin"));
    float* h1 = (float*)dzero(sizeof(float) * H);
    void* attn = dzero(2 * (size_t)H);
    void *qkv[NL][3], *o[NL], *gu[NL][2], *dn[NL];
    for (int l = 0; l < NL; ++l) {
        for (int k = 0; k < 3; ++k) qkv[l][k] = dzero(2 * (size_t)H);
        o[l] = dzero(2 * (size_t)H); gu[l][0] = dzero(2 * (size_t)F); gu[l][1] = dzero(2 * (size_t)F); dn[l] = dzero(2 * (size_t)H);
    }
    // ---- record once
    ggml_tmac_hip_segment *s0 = nullptr, *seg[NL] = {nullptr, nullptr};
    const tmac_ggml_tensor* wqkv0[3] = {&w[0][0], &w[0][1], &w[0][2]};
    CK(ggml_tmac_hip_segment_begin());
    CK(ggml_tmac_hip_segment_norm(nullptr, 0, g[0][0], 1e-5f, nullptr, 0));
    CK(ggml_tmac_hip_segment_mul_mat_f32(wqkv0, 3, h0, qkv[0]));        // the token's embedding as ggml holds it: fp32
    (void)x0;
    CK(ggml_tmac_hip_segment_end(&s0));
    for (int l = 0; l < NL; ++l) {
        const tmac_ggml_tensor *wo[1] = {&w[l][3]}, *wgu[2] = {&w[l][4], &w[l][5]}, *wd[1] = {&w[l][6]};
        CK(ggml_tmac_hip_segment_begin());
        CK(ggml_tmac_hip_segment_mul_mat(wo, 1, attn, &o[l]));
        CK(ggml_tmac_hip_segment_norm(l == 0 ? h0 : h1, 0, g[l][1], 1e-5f, nullptr, 1));
        CK(ggml_tmac_hip_segment_mul_mat(wgu, 2, o[l], gu[l]));
        CK(ggml_tmac_hip_segment_glu(gu[l][1]));
        CK(ggml_tmac_hip_segment_mul_mat(wd, 1, gu[l][0], &dn[l]));
        if (l + 1 < NL) {
            const tmac_ggml_tensor* wn[3] = {&w[l + 1][0], &w[l + 1][1], &w[l + 1][2]};
            CK(ggml_tmac_hip_segment_norm(nullptr, 1, g[l + 1][0], 1e-5f, h1, 0));
            CK(ggml_tmac_hip_segment_mul_mat(wn, 3, dn[l], qkv[l + 1]));
        }
        CK(ggml_tmac_hip_segment_end(&seg[l]));
    }
    // ---- two "tokens" (the second replays the recorded segments; tensors of the last one are dumped)
    void* backend_stream = nullptr;
    if (hipStreamCreateWithFlags(&backend_stream, 1)) return 9;
    for (int tok = 0; tok < 2; ++tok) {
        // the second token runs on a stream the "backend" hands to the glue (ggml_tmac_hip_set_stream): launches and the outside
        // operator stay ordered on it
        if (tok == 1) { CK(ggml_tmac_hip_set_stream(backend_stream)); if (ggml_tmac_hip_stream() != backend_stream) return 10; }
        CK(ggml_tmac_hip_segment_compute(s0));
        for (int l = 0; l < NL; ++l) {
            // the operator outside the hook, on the glue's stream (stand-in for attention): attn = q of this layer
            if (hipMemcpyAsync(attn, qkv[l][0], 2 * (size_t)H, 3, ggml_tmac_hip_stream())) return 9;
            CK(ggml_tmac_hip_segment_compute(seg[l]));
            if (l == 0) {   // (dump layer 0's attention input before layer 1 overwrites the buffer)
                CK(ggml_tmac_hip_segment_wait(seg[l]));
                dump(d + "/out_attn0.bin", attn, 2 * (size_t)H);
            }
        }
        CK(ggml_tmac_hip_segment_wait(seg[NL - 1]));
        CK(ggml_tmac_hip_segment_wait(s0));
    }
    dump(d + "/out_attn1.bin", attn, 2 * (size_t)H);
    dump(d + "/out_h1.bin", h1, sizeof(float) * H);
    for (int l = 0; l < NL; ++l) {
        for (int k = 0; k < 3; ++k) dump(d + "/out_" + std::to_string(l) + "_" + names[k] + ".bin", qkv[l][k], 2 * (size_t)H);
        dump(d + "/out_" + std::to_string(l) + "_o.bin", o[l], 2 * (size_t)H);
        dump(d + "/out_" + std::to_string(l) + "_gate.bin", gu[l][0], 2 * (size_t)F);
        dump(d + "/out_" + std::to_string(l) + "_up.bin", gu[l][1], 2 * (size_t)F);
        dump(d + "/out_" + std::to_string(l) + "_down.bin", dn[l], 2 * (size_t)H);
    }
    ggml_tmac_hip_segment_free(s0);
    for (int l = 0; l < NL; ++l) ggml_tmac_hip_segment_free(seg[l]);
    for (int l = 0; l < NL; ++l)
        for (int m = 0; m < 7; ++m) ggml_tmac_hip_free(&w[l][m]);
    printf("RESULT ok\n");
    return 0;
}
