/*
 * tmac_oracle.c — CPU restatement of T-MAC's LUT mpGEMM hot path (x86 / fp32 flavour).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (tmac_amd/) never
 * links, imports or calls anything in oracle/.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit against the
 * reference's own sources compiled from /root/reference (oracle/_ref, see Makefile and
 * tests/test_oracle_vs_ref.py) and against the known-answer vector of
 * tests/test_lut_ctor.cc (tests/golden/lut_ctor_kat.json).
 * The one expression with no compiled x86 exemplar in the reference is the scale-final
 * epilogue (oracle_qgemm_scale_final, spec python/t_mac/ops/qgemm.py:170-174,192-206);
 * its integer part is pinned (tbl.cc:586-628), its last three float ops are "restated".
 * Fast aggregation (oracle_qgemm_float_fa): fa_mode 2 (the reference's AVX2 build) is PINNED bit for
 * bit against the reference's FastAggregation = true instantiation; fa_mode 1 (its NEON build,
 * vrhaddq_s8) cannot be compiled on this x86 host: it shares the pinned tree/rescale/bias code and
 * differs in the one averaging expression -- "restated", parity unpinned for that expression.
 *
 * Plain scalar C; every float op is individually rounded (compile with
 * -ffp-contract=off); fmaf() appears exactly where the reference uses _mm256_fmadd_ps.
 *
 * Notation: Mw weight rows, bits b, M = Mw*b bit-plane rows, g = 4,
 * K/4 LUT groups ("tables"), ags = act_group_size, gs = weight group_size,
 * bm = M-tile (bit-plane rows), kfactor = tables per tbl call.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- helpers -------------------------------------------------------------------- */

/* _mm256_cvtps_epi32(_mm256_round_ps(x, NEAREST)) followed by packs_epi32/packs_epi16
 * (python/t_mac/intrins/lut_ctor.cc:169-177): RNE, then saturate to int8.  cvtps of a
 * NaN / out-of-range value yields INT_MIN, which saturates to -128. */
static inline int8_t rne_sat_int8(float x) {
    float r = nearbyintf(x); /* default rounding mode = round-to-nearest-even */
    int32_t i;
    if (!(r >= -2147483648.0f && r < 2147483648.0f)) i = INT32_MIN;
    else i = (int32_t)r;
    if (i > 127) i = 127;
    if (i < -128) i = -128;
    return (int8_t)i;
}

/* ---- (a1)+(a2)+(a3): preprocessor ------------------------------------------------
 * python/t_mac/intrins/lut_ctor.cc:232-266 (partial_max), :38-221 (lut_ctor, AVX2 branch
 * :120-215); glue deploy/tuned/aarch64-llama-2-7b-2bit/kernels.cc:1223-1231.
 *   B          [N][K] float
 *   lut_scales [N][K/ags], lut_biases [N][K/ags] float
 *   qlut       [N][K/4][16] int8
 * Requires ags % 32 == 0 (python/t_mac/ops/qgemm.py:402-404) and K % ags == 0. */
int oracle_preprocessor(const float* B, int N, int K, int ags,
                        float* lut_scales, float* lut_biases, int8_t* qlut) {
    if (ags <= 0 || ags % 32 != 0 || K % ags != 0) return -1;
    const int G = K / ags;
    for (int n = 0; n < N; ++n) {
        const float* b_n = B + (size_t)n * K;
        for (int kk = 0; kk < G; ++kk) {
            const float* b = b_n + (size_t)kk * ags;
            /* partial_max_reset + partial_max_g4_int8_k8 per 32 activations */
            float scales = 0.0f;
            for (int c = 0; c < ags / 32; ++c) {
                float mx = -INFINITY; /* max over the chunk's 8 abs-sums */
                for (int i = 0; i < 8; ++i) {
                    const float* x = b + c * 32 + i * 4;
                    float as = (fabsf(x[0]) + fabsf(x[1])) + (fabsf(x[2]) + fabsf(x[3]));
                    if (i == 0 || as > mx) mx = as;
                }
                float s = mx / 127; /* lut_ctor.cc:255 */
                scales = scales > s ? scales : s; /* std::max(*lut_scales, scales) */
            }
            const float t_scales = scales ? 1.0f / scales : 0.0f; /* lut_ctor.cc:124-125 */
            float biases = 0.0f;
            int8_t* q = qlut + ((size_t)n * (K / 4) + (size_t)kk * (ags / 4)) * 16;
            for (int c = 0; c < ags / 32; ++c) {
                float L[8][16];
                for (int i = 0; i < 8; ++i) {
                    const float* x = b + c * 32 + i * 4;
                    for (int g = 1; g < 16; g += 2) { /* lut_ctor.cc:134-151 */
                        float v = x[0];
                        v = (g & 2) ? v + x[1] : v - x[1];
                        v = (g & 4) ? v + x[2] : v - x[2];
                        v = (g & 8) ? v + x[3] : v - x[3];
                        L[i][g] = v;
                    }
                    for (int g = 0; g < 16; g += 2) L[i][g] = -L[i][15 - g]; /* :152-155 */
                }
                /* biases += _mm256_addv_ps(vec_lut[0])  (lut_ctor.cc:25-31,157) */
                {
                    float r0 = L[4][0] + L[0][0], r1 = L[5][0] + L[1][0];
                    float r2 = L[6][0] + L[2][0], r3 = L[7][0] + L[3][0];
                    float s0 = r0 + r2, s1 = r1 + r3;
                    biases += s0 + s1;
                }
                for (int i = 0; i < 8; ++i)
                    for (int g = 0; g < 16; ++g)
                        q[(c * 8 + i) * 16 + g] = rne_sat_int8(L[i][g] * t_scales);
            }
            lut_scales[(size_t)n * G + kk] = scales;
            lut_biases[(size_t)n * G + kk] = biases;
        }
    }
    return 0;
}

/* ---- reference weight-blob addressing (a7 / Appendix A.3) --------------------------
 * python/t_mac/weights.py:57-73.  Nibble of M-space (bit-plane) row r at table t. */
static inline int ref_nibble(const uint8_t* A, int K, int bm, int kfactor, int r, int t) {
    const int tile = r / bm, rr = r % bm;
    const size_t byte = (size_t)tile * ((size_t)(bm / 2) * (K / 4)) +
                        ((size_t)(t / kfactor) * (bm / 32) + rr / 32) * kfactor * 16 +
                        (size_t)(t % kfactor) * 16 + (rr % 16);
    return (A[byte] >> (4 * ((rr % 32) / 16))) & 15;
}

/* M-space row of (output row o, bit-plane p): weights.py:65 ([Mw/8][bits][8]). */
static inline int mrow(int o, int p, int bits) { return (o / 8) * 8 * bits + p * 8 + (o % 8); }

/* ---- integer partial sums (the bit-exact contract) --------------------------------
 * PS[r][kk] = sum over the act group's tables of QLUT[t][nibble(r,t)]
 * (tbl.cc:452-462 + SignedWideningAdder :258-280; int32 flavour :596-625).
 *   A    reference-layout weights for all M = Mw*bits rows
 *   qlut [K/4][16] for ONE activation row
 *   PS   [M][K/ags] int32, M-space row order */
int oracle_partial_sums(const uint8_t* A, const int8_t* qlut, int Mw, int K, int bits,
                        int bm, int kfactor, int ags, int32_t* PS) {
    const int M = Mw * bits;
    if (M % bm || bm % 32 || (K / 4) % kfactor || K % ags || ags % 4) return -1;
    const int G = K / ags, TG = ags / 4;
    for (int r = 0; r < M; ++r)
        for (int kk = 0; kk < G; ++kk) {
            int32_t s = 0;
            for (int tl = 0; tl < TG; ++tl) {
                int t = kk * TG + tl;
                s += qlut[(size_t)t * 16 + ref_nibble(A, K, bm, kfactor, r, t)];
            }
            PS[(size_t)r * G + kk] = s;
        }
    return 0;
}

/* ---- (a4)+(a6): GPTQ-style float path ---------------------------------------------
 * tbl_g4_int8_float_update_impl AVX2 branch (tbl.cc:435-529) called once per k_outer by
 * the generated glue (aarch64-llama-2-7b-2bit/kernels.cc:1059-1100), then the bit-plane
 * combine (kernels.cc:1065-1100, spec qgemm.py:192-206).
 *   A       reference layout, all tiles            uint8 [M/bm][K/4][bm/2]
 *   qlut    [N][K/4][16]; lut_scales/lut_biases [N][K/ags]
 *   scales  reference layout, all tiles:
 *             zero_point: [M/bm][K/gs][bm/bits/8][2][8]
 *             else      : [M/bm][K/gs][bm/bits/8][8]
 *             one_scale : [1] (m_groups == 1 with per-group LUT scales: the ARM "os=true" kernel)
 *   C       [N][Mw] float
 * one_scale follows the NEON branch (tbl.cc:417-423) — the AVX2 branch ignores OneScale
 * (tbl.cc:518-527), a known gap of the reference on x86. */
/* (a9) fast aggregation: the act group's ActK looked-up bytes are not summed exactly but folded by a
 * balanced tree of rounding-halving adds, in table order (SignedHalvingAdder<N>, tbl.cc:86-141 NEON /
 * :201-256 AVX2): H_2(v0,v1) = avg(v0,v1); H_N = avg(H_{N/2}(first half), H_{N/2}(second half)).
 *   fa_mode 1: NEON  vrhaddq_s8  = (a + b + 1) >> 1 on SIGNED bytes          (tbl.cc:101,123)
 *   fa_mode 2: AVX2  _mm256_avg_epu8 = (a + b + 1) >> 1 on the same bytes read as UNSIGNED, the result
 *              sign-extended again (tbl.cc:218,237,226-231) -- what the reference's x86 build computes.
 * The result stands for sum/ActK, so lut_s is multiplied by ActK and the expected rounding bias
 * (log2(ActK)/4 in INTEGER arithmetic, times get_bias_scale(bits) = 2^bits - 1) is taken off lut_b
 * (tbl.cc:301-318,369-372,474-477). */
static int fa_tree(const int8_t* v, int n, int fa_mode) {
    if (n == 1) return v[0];
    const int a = fa_tree(v, n / 2, fa_mode), b = fa_tree(v + n / 2, n / 2, fa_mode);
    if (fa_mode == 1) return (a + b + 1) >> 1;                           /* arithmetic shift: floor */
    return (int8_t)(uint8_t)((((unsigned)(uint8_t)a) + ((unsigned)(uint8_t)b) + 1u) >> 1);
}
static int ilog2(int k) { int l = -1; while (k) { ++l; k /= 2; } return l; }  /* mylog2, tbl.cc:287-299 */

static int qgemm_float_impl(const uint8_t* A, const int8_t* qlut, const float* scales,
                       const float* lut_scales, const float* lut_biases, float* C,
                       int Mw, int K, int N, int bits, int bm, int kfactor, int gs, int ags,
                       int zero_point, int one_scale, int fa_mode, int32_t* agg_tap) {
    const int M = Mw * bits;
    if (fa_mode < 0 || fa_mode > 2) return -1;
    if (M % bm || bm % 32 || bm % bits || (bm / bits) % 8 || (K / 4) % kfactor) return -1;
    if (K % ags || (4 * kfactor) % ags) return -1;
    if (!one_scale && (K % gs || gs % (4 * kfactor))) return -1;
    const int G = K / ags, TG = ags / 4;
    const int ActK = TG < kfactor ? TG : kfactor;   /* tbl.py: min(ags/4, kfactor) */
    if (fa_mode && (ActK > 64 || (ActK & (ActK - 1)) || ActK < 2)) return -1;
    const int groups_per_call = kfactor / ActK;
    const int ncalls = (K / 4) / kfactor;
    const int rows_per_tile = bm / bits;             /* output rows per tile */
    const int sstride = rows_per_tile * (zero_point ? 2 : 1);
    static const float alphas[4] = {0.5f, 1.0f, 2.0f, 4.0f};
    float* cbits = (float*)malloc(sizeof(float) * (size_t)M);
    if (!cbits) return -2;
    for (int n = 0; n < N; ++n) {
        const int8_t* q = qlut + (size_t)n * (K / 4) * 16;
        const float* ls = lut_scales + (size_t)n * G;
        const float* lb = lut_biases + (size_t)n * G;
        memset(cbits, 0, sizeof(float) * (size_t)M); /* tbl_float_reset */
        for (int r = 0; r < M; ++r) {
            const int tile = r / bm, rr = r % bm;
            const int plane = (rr / 8) % bits;          /* (ib % Bits), ib = row/8 */
            const int m_out = (rr / 8 / bits) * 8 + rr % 8; /* row within the tile's outputs */
            float c = 0.0f;
            for (int ko = 0; ko < ncalls; ++ko) {
                float vec_c = 0.0f;
                float partial_sum = -0.0f;
                for (int j = 0; j < groups_per_call; ++j) {
                    const int kk = ko * groups_per_call + j;
                    int32_t s = 0; /* exact int16 sum in the reference (|s| <= 16*127) */
                    int8_t looked[64];
                    for (int tl = 0; tl < ActK; ++tl) {
                        int t = ko * kfactor + j * ActK + tl;
                        looked[tl] = q[(size_t)t * 16 + ref_nibble(A, K, bm, kfactor, r, t)];
                        s += looked[tl];
                    }
                    float lut_s = ls[kk], lut_b = lb[kk];
                    partial_sum += lut_b;
                    if (fa_mode) {
                        s = fa_tree(looked, ActK, fa_mode);
                        if (agg_tap && ActK == TG) agg_tap[((size_t)n * M + r) * G + kk] = s;
                        lut_s = lut_s * (float)ActK;
                        lut_b -= lut_s * (float)(ilog2(ActK) / 4 * ((1 << bits) - 1));
                    }
                    const float v = (float)s;
                    /* lut_fma: plane 0 -> fmadd(v, lut_s, lut_b); else mul (tbl.cc:479-481) */
                    const float f = (plane == 0) ? fmaf(v, lut_s, lut_b) : v * lut_s;
                    vec_c = (j == 0) ? f : vec_c + f;
                }
                if (one_scale) {
                    c = fmaf(vec_c, scales[0], c); /* c += vec_c * scales[0] (tbl.cc:417-423) */
                } else {
                    const int sg = (ko * 4 * kfactor) / gs;
                    const float* sp = scales + ((size_t)tile * (K / gs) + sg) * sstride;
                    if (zero_point) {
                        const float sc = sp[(m_out / 8) * 16 + (m_out % 8)];
                        const float zr = sp[(m_out / 8) * 16 + 8 + (m_out % 8)];
                        c = fmaf(vec_c, sc, c);               /* tbl.cc:501-504 */
                        partial_sum *= 2;                     /* tbl.cc:509 */
                        if (plane == 0) c = fmaf(zr, partial_sum, c); /* tbl.cc:510-516 */
                    } else {
                        const float sc = sp[(m_out / 8) * 8 + (m_out % 8)];
                        c = fmaf(vec_c, sc, c);               /* tbl.cc:523-526 */
                    }
                }
            }
            cbits[r] = c;
        }
        /* bit-plane combine, fp32 left-to-right (kernels.cc:1068; qgemm.py:192-206) */
        for (int o = 0; o < Mw; ++o) {
            float acc = cbits[mrow(o, 0, bits)] * alphas[0];
            for (int p = 1; p < bits; ++p) acc = acc + cbits[mrow(o, p, bits)] * alphas[p];
            C[(size_t)n * Mw + o] = acc;
        }
    }
    free(cbits);
    return 0;
}

int oracle_qgemm_float(const uint8_t* A, const int8_t* qlut, const float* scales,
                       const float* lut_scales, const float* lut_biases, float* C,
                       int Mw, int K, int N, int bits, int bm, int kfactor, int gs, int ags,
                       int zero_point, int one_scale) {
    return qgemm_float_impl(A, qlut, scales, lut_scales, lut_biases, C, Mw, K, N, bits, bm, kfactor, gs, ags,
                            zero_point, one_scale, 0, NULL);
}

/* fast-aggregation flavour; agg_tap (optional) receives the tree result per (n, M-space row, act group) */
int oracle_qgemm_float_fa(const uint8_t* A, const int8_t* qlut, const float* scales,
                          const float* lut_scales, const float* lut_biases, float* C, int32_t* agg_tap,
                          int Mw, int K, int N, int bits, int bm, int kfactor, int gs, int ags,
                          int zero_point, int fa_mode) {
    if (fa_mode != 1 && fa_mode != 2) return -1;
    return qgemm_float_impl(A, qlut, scales, lut_scales, lut_biases, C, Mw, K, N, bits, bm, kfactor, gs, ags,
                            zero_point, 0, fa_mode, agg_tap);
}

/* ---- (a5): unified-scale (BitNet) path, int32 aggregation, scale applied last -------
 * Selected by the reference when m_groups != -1 and ags == K (qgemm.py:93-96).
 * Integer part: tbl_g4_int8_int32_update_impl (tbl.cc:586-628).
 * Epilogue (qgemm.py:170-174,192-206):
 *   C[n][o] = ((sum_p float(CBits32[r(o,p)]) * alpha_p) * LUT_Scales[n][0]
 *              + LUT_Biases[n][0] * alpha_0) * Scales[o / (Mw / m_groups)]
 * CBits32 (optional, may be NULL) receives the int32 sums [N][M]. */
int oracle_qgemm_scale_final(const uint8_t* A, const int8_t* qlut, const float* scales,
                             const float* lut_scales, const float* lut_biases, float* C,
                             int32_t* CBits32, int Mw, int K, int N, int bits, int bm,
                             int kfactor, int m_groups) {
    const int M = Mw * bits;
    if (M % bm || bm % 32 || (K / 4) % kfactor || m_groups <= 0 || Mw % m_groups) return -1;
    static const float alphas[4] = {0.5f, 1.0f, 2.0f, 4.0f};
    int32_t* cb = (int32_t*)malloc(sizeof(int32_t) * (size_t)M);
    if (!cb) return -2;
    for (int n = 0; n < N; ++n) {
        const int8_t* q = qlut + (size_t)n * (K / 4) * 16;
        for (int r = 0; r < M; ++r) {
            int32_t s = 0;
            for (int t = 0; t < K / 4; ++t)
                s += q[(size_t)t * 16 + ref_nibble(A, K, bm, kfactor, r, t)];
            cb[r] = s;
        }
        if (CBits32) memcpy(CBits32 + (size_t)n * M, cb, sizeof(int32_t) * (size_t)M);
        for (int o = 0; o < Mw; ++o) {
            float acc = (float)cb[mrow(o, 0, bits)] * alphas[0];
            for (int p = 1; p < bits; ++p) acc = acc + (float)cb[mrow(o, p, bits)] * alphas[p];
            float v = acc * lut_scales[n] + lut_biases[n] * alphas[0];
            C[(size_t)n * Mw + o] = v * scales[o / (Mw / m_groups)];
        }
    }
    free(cb);
    return 0;
}

/* ---- (a7): offline weight transform ------------------------------------------------
 * python/t_mac/weights.py:5-88 restated through the address formulas of Appendix A.3.
 *   w       [Mw][K] uint8 in [0, 2^bits)
 *   A_out   [M/bm][K/4][bm/2] uint8 (zeroed here)
 * Scales: sc/zr [Mw][K/gs] -> S_out [M/bm][K/gs][bm/bits/8][2|1][8]  (zr may be NULL). */
int oracle_preprocess_weights(const uint8_t* w, int Mw, int K, int bits, int bm, int kfactor,
                              uint8_t* A_out) {
    const int M = Mw * bits;
    if (M % bm || bm % 32 || (K / 4) % kfactor || Mw % 8) return -1;
    memset(A_out, 0, (size_t)M * (K / 4) / 2);
    for (int o = 0; o < Mw; ++o)
        for (int p = 0; p < bits; ++p) {
            const int r = mrow(o, p, bits);
            const int tile = r / bm, rr = r % bm;
            for (int t = 0; t < K / 4; ++t) {
                int nib = 0;
                for (int ig = 0; ig < 4; ++ig) /* weights.py:59-60 */
                    nib |= ((w[(size_t)o * K + 4 * t + ig] >> p) & 1) << ig;
                const size_t byte = (size_t)tile * ((size_t)(bm / 2) * (K / 4)) +
                                    ((size_t)(t / kfactor) * (bm / 32) + rr / 32) * kfactor * 16 +
                                    (size_t)(t % kfactor) * 16 + (rr % 16);
                A_out[byte] |= (uint8_t)(nib << (4 * ((rr % 32) / 16)));
            }
        }
    return 0;
}

int oracle_preprocess_scales(const float* sc, const float* zr, int Mw, int K, int bits, int bm,
                             int gs, float* S_out) {
    const int M = Mw * bits;
    if (M % bm || bm % bits || (bm / bits) % 8 || K % gs) return -1;
    const int rpt = bm / bits, SG = K / gs, zp = zr != NULL;
    for (int o = 0; o < Mw; ++o) {
        const int tile = o / rpt, m_out = o % rpt;
        for (int g = 0; g < SG; ++g) {
            float* sp = S_out + ((size_t)tile * SG + g) * rpt * (zp ? 2 : 1);
            if (zp) {
                sp[(m_out / 8) * 16 + (m_out % 8)] = sc[(size_t)o * SG + g];
                sp[(m_out / 8) * 16 + 8 + (m_out % 8)] = zr[(size_t)o * SG + g];
            } else {
                sp[(m_out / 8) * 8 + (m_out % 8)] = sc[(size_t)o * SG + g];
            }
        }
    }
    return 0;
}

/* Plain dequantise-then-multiply in fp64: the *statistical* reference of
 * tests/test_e2e.py:68-77 (NMSE check only; not part of the bit-exact contract). */
int oracle_dequant_matmul(const uint8_t* w, const float* sc, const float* zr, const float* B,
                          double* C, int Mw, int K, int N, int bits, int gs, int m_groups) {
    const double dz = (double)(1 << (bits - 1));
    for (int n = 0; n < N; ++n)
        for (int o = 0; o < Mw; ++o) {
            double acc = 0.0;
            for (int k = 0; k < K; ++k) {
                double s, z = 0.0;
                if (m_groups > 0) s = sc[o / (Mw / m_groups)];
                else { s = sc[(size_t)o * (K / gs) + k / gs]; if (zr) z = zr[(size_t)o * (K / gs) + k / gs]; }
                acc += (double)B[(size_t)n * K + k] * (((double)w[(size_t)o * K + k] - dz) * s - z);
            }
            C[(size_t)n * Mw + o] = acc;
        }
    return 0;
}

#ifdef __cplusplus
}
#endif
