// emu.cpp — CPU emulation of the tiled GEMV kernel's INTEGER path (test infrastructure).
//
// Built by g++ into tmac_amd/lib/libtmac_emu.so and used only by tests/test_emulation.py
// (-m "not gpu"): it runs the very same layout math (tmac_layout.h) and per-thread lookup /
// accumulate code (tmac_core.h) the HIP kernels compile, with v_perm_b32 / v_mqsad_pk_u16_u8
// replaced by their bit-exact host models, thread by thread, and returns the integer partial
// sums.  It is NOT a product path: libtmac_hip.so contains no CPU compute.
#include <cstdint>
#include <cstring>
#include <vector>

#include "tmac_core.h"

using namespace tmac;

static uint32_t g_fa_xor = 0;   // MODE 2 (fast aggregation): 0 signed halving adds, 0x80808080 the AVX2 flavour

template <int BITS, int MODE>
static void run(const std::vector<uint32_t>& W, const std::vector<uint32_t>& QL, const Shape& s, int32_t* PS) {
    constexpr int NJ = TS * BITS / 8;
    const int TG = s.ags == s.K ? TS : s.ags / 4;  // tables per act group inside a segment
    const int NA = TS / TG, G = s.K / s.ags;
    for (int b = 0; b < s.nb(); ++b)
        for (int sb = 0; sb < s.nsb(); ++sb)
            for (int rl = 0; rl < RL; ++rl)
                for (int kl = 0; kl < KL; ++kl) {
                    const int seg = sb * KL + kl, rq = b * RL + rl;
                    if (seg >= s.nseg()) continue;
                    uint32_t wd[TS * BITS / 2], tb[2 * TS];
                    for (int j = 0; j < NJ; ++j)
                        for (int e = 0; e < 4; ++e) wd[4 * j + e] = W[weight_u4_index(s, b, sb, j, rl, kl) * 4 + e];
                    for (int j8 = 0; j8 < 8; ++j8)
                        for (int e = 0; e < 4; ++e) tb[4 * j8 + e] = QL[(((size_t)sb * 8 + j8) * KL + kl) * 4 + e];
                    for (int a = 0; a < NA; ++a) {
                        SegAcc<BITS, MODE> acc;
                        acc.reset();
                        if constexpr (MODE == 2) acc.xr = g_fa_xor;
                        if (TG == 16) accumulate_tables<BITS, 0, 16>(wd, tb, acc);
                        else if (a == 0) accumulate_tables<BITS, 0, 8>(wd, tb, acc);
                        else accumulate_tables<BITS, 8, 8>(wd, tb, acc);
                        for (int beta = 0; beta < 4; ++beta)
                            for (int p = 0; p < BITS; ++p) {
                                const int o = 4 * rq + beta;
                                if (o >= s.Mw) continue;
                                const int32_t v = acc.ps(p, beta, TG);
                                if (s.ags == s.K) PS[mrow(o, p, BITS)] += v;  // scale-final: whole-K sum
                                else PS[(size_t)mrow(o, p, BITS) * G + seg * NA + a] = v;
                            }
                    }
                }
}

// fused-layout kernel (ts = 8): a thread owns (row quad, 8-table unit); the LUT sits in the LDS image
// [4][nu_pad+1] uint4; lanes (ul, ul^1) add their packed u16 sums to complete a 64-activation group.
template <int BITS>
static void run8(const std::vector<uint32_t>& W, const std::vector<uint32_t>& QLDS, const Shape& s, int32_t* PS) {
    constexpr int NJ = 8 * BITS / 8;
    const int nu = s.K / 32, tstride = ((nu + 15) & ~15) + 1, G = s.K / s.ags;
    for (int b = 0; b < s.nb(); ++b)
        for (int ub = 0; ub < s.nsb(); ++ub)
            for (int rl = 0; rl < RL; ++rl)
                for (int ul = 0; ul < KL; ul += 2) {
                    uint64_t pair[BITS] = {};
                    for (int half = 0; half < 2; ++half) {
                        const int u = ub * KL + ul + half;
                        if (u >= nu) continue;
                        uint32_t wd[8 * BITS / 2], tb[16];
                        for (int j = 0; j < NJ; ++j)
                            for (int e = 0; e < 4; ++e) wd[4 * j + e] = W[weight_u4_index(s, b, ub, j, rl, ul + half) * 4 + e];
                        for (int j4 = 0; j4 < 4; ++j4)
                            for (int e = 0; e < 4; ++e) tb[4 * j4 + e] = QLDS[((size_t)j4 * tstride + u) * 4 + e];
                        SegAcc<BITS, 0> acc;
                        acc.reset();
                        accumulate_tables<BITS, 0, 8>(wd, tb, acc);
                        for (int p = 0; p < BITS; ++p) pair[p] += acc.a[p];   // packed u16 add, no carries
                    }
                    const int u0 = ub * KL + ul;
                    if (u0 >= nu) continue;
                    for (int beta = 0; beta < 4; ++beta)
                        for (int p = 0; p < BITS; ++p) {
                            const int o = 4 * (b * RL + rl) + beta;
                            if (o >= s.Mw) continue;
                            const int32_t v = 127 * 16 - (int32_t)((pair[p] >> (16 * beta)) & 0xffff);
                            if (s.ags == s.K) PS[mrow(o, p, BITS)] += v;
                            else PS[(size_t)mrow(o, p, BITS) * G + u0 / 2] = v;
                        }
                }
}

// QUAD layout kernel (k_gemv_quad): the 64 lanes of a wave are 64 consecutive units of one row quad
template <int BITS>
static void run_quad(const std::vector<uint32_t>& W, const std::vector<uint32_t>& QLDS, const Shape& s, int32_t* PS) {
    constexpr int NJ = 8 * BITS / 8;
    const int nu = s.K / 32, tstride = ((nu + 15) & ~15) + 1, G = s.K / s.ags;
    for (int quad = 0; quad < s.nquads(); ++quad)
        for (int st = 0; st < s.nst64(); ++st)
            for (int lane = 0; lane < 64; lane += 2) {
                uint64_t pair[BITS] = {};
                for (int half = 0; half < 2; ++half) {
                    const int u = st * 64 + lane + half;
                    if (u >= nu) continue;
                    uint32_t wd[8 * BITS / 2], tb[16];
                    for (int j = 0; j < NJ; ++j)
                        for (int e = 0; e < 4; ++e) wd[4 * j + e] = W[quad_weight_u4_index(s, quad, st, j, lane + half) * 4 + e];
                    for (int j4 = 0; j4 < 4; ++j4)
                        for (int e = 0; e < 4; ++e) tb[4 * j4 + e] = QLDS[((size_t)j4 * tstride + u) * 4 + e];
                    SegAcc<BITS, 0> acc;
                    acc.reset();
                    accumulate_tables<BITS, 0, 8>(wd, tb, acc);
                    for (int p = 0; p < BITS; ++p) pair[p] += acc.a[p];
                }
                const int u0 = st * 64 + lane;
                if (u0 >= nu) continue;
                for (int beta = 0; beta < 4; ++beta)
                    for (int p = 0; p < BITS; ++p) {
                        const int o = 4 * quad + beta;
                        if (o >= s.Mw) continue;
                        const int32_t v = 127 * 16 - (int32_t)((pair[p] >> (16 * beta)) & 0xffff);
                        if (s.ags == s.K) PS[mrow(o, p, BITS)] += v;
                        else PS[(size_t)mrow(o, p, BITS) * G + u0 / 2] = v;
                    }
            }
}

// fused-layout kernel with the MFMA accumulate (ACC = 1): whole-wave emulation with the operand model of
// v_mfma_i32_16x16x64_i8 that tools/mfma_probe.py checks against the hardware:
//   D[i][j] = sum_g sum_{16 bytes} A[lane 16g+i] . B[lane 16g+j];  D[i][j] lives in lane j + 16*(i/4), reg i%4
template <int BITS>
static void run8_mfma(const std::vector<uint32_t>& W, const std::vector<uint32_t>& QS, const Shape& s, int32_t* PS) {
    constexpr int NJ = 8 * BITS / 8;
    const int nu = s.K / 32, tstride = ((nu + 15) & ~15) + 1, G = s.K / s.ags;
    uint32_t bsel[64][4];
    for (int lane = 0; lane < 64; ++lane) {
        const int jrel = (lane & 15) - 4 * (lane >> 4);
        const uint32_t be = (jrel >= 0 && jrel < 4) ? (0x01u << (8 * jrel)) : 0u, bo = (jrel >= 0 && jrel < 4) ? (0xffu << (8 * jrel)) : 0u;
        bsel[lane][0] = be; bsel[lane][1] = bo; bsel[lane][2] = be; bsel[lane][3] = bo;
    }
    for (int b = 0; b < s.nb(); ++b)
        for (int ub = 0; ub < s.nsb(); ++ub) {          // one wave-step: 16 units x 4 row quads
            int32_t c[BITS][64][4] = {};
            for (int tp = 0; tp < 4; ++tp)
                for (int pl = 0; pl < BITS; ++pl) {
                    uint32_t A[64][4];
                    for (int lane = 0; lane < 64; ++lane) {
                        const int rl = lane >> 4, ul = lane & 15, u = ub * KL + ul;
                        uint32_t wd[8 * BITS / 2] = {}, tb[16] = {};
                        if (u < nu) {
                            for (int j = 0; j < NJ; ++j)
                                for (int e = 0; e < 4; ++e) wd[4 * j + e] = W[weight_u4_index(s, b, ub, j, rl, ul) * 4 + e];
                            for (int j4 = 0; j4 < 4; ++j4)
                                for (int e = 0; e < 4; ++e) tb[4 * j4 + e] = QS[((size_t)j4 * tstride + u) * 4 + e];
                        }
                        const int qa = (2 * tp) * BITS + pl, qb = (2 * tp + 1) * BITS + pl;
                        if (qa & 1) lookup4_pm<1>(wd[qa >> 1], tb[4 * tp], tb[4 * tp + 1], A[lane][0], A[lane][1]);
                        else lookup4_pm<0>(wd[qa >> 1], tb[4 * tp], tb[4 * tp + 1], A[lane][0], A[lane][1]);
                        if (qb & 1) lookup4_pm<1>(wd[qb >> 1], tb[4 * tp + 2], tb[4 * tp + 3], A[lane][2], A[lane][3]);
                        else lookup4_pm<0>(wd[qb >> 1], tb[4 * tp + 2], tb[4 * tp + 3], A[lane][2], A[lane][3]);
                    }
                    for (int i = 0; i < 16; ++i)
                        for (int j = 0; j < 16; ++j) {
                            int32_t acc = 0;
                            for (int g = 0; g < 4; ++g)
                                for (int q = 0; q < 4; ++q)
                                    for (int be = 0; be < 4; ++be)
                                        acc += (int32_t)(int8_t)(A[16 * g + i][q] >> (8 * be)) * (int32_t)(int8_t)(bsel[16 * g + j][q] >> (8 * be));
                            c[pl][j + 16 * (i / 4)][i % 4] += acc;
                        }
                }
            for (int lane = 0; lane < 64; ++lane) {
                const int lg = lane >> 4, rlp = (lane & 15) >> 2, bp = lane & 3, ub4 = ub * KL + 4 * lg;
                const int o = 4 * (b * RL + rlp) + bp;
                if (o >= s.Mw) continue;
                for (int gi = 0; gi < 2; ++gi) {
                    const int ug = ub4 + 2 * gi;
                    if (ug >= nu) continue;
                    for (int pl = 0; pl < BITS; ++pl) {
                        const int32_t ps = c[pl][lane][2 * gi] + c[pl][lane][2 * gi + 1];
                        if (s.ags == s.K) PS[mrow(o, pl, BITS)] += ps;
                        else PS[(size_t)mrow(o, pl, BITS) * G + ug / 2] = ps;
                    }
                }
            }
        }
}

extern "C" int emu_partial_sums(const uint8_t* A_ref, const int8_t* qlut_ref, int Mw, int K, int bits, int bm,
                                int kfactor, int ags, int mode, int32_t* PS) {
    Shape s;
    memset(&s, 0, sizeof(s));
    s.Mw = Mw; s.K = K; s.bits = bits; s.bm = bm; s.kfactor = kfactor; s.gs = 128; s.ags = ags; s.m_groups = -1;
    const int fa = mode == 5 ? 1 : mode == 6 ? 2 : 0;   // modes 5 / 6: two-kernel layout with fast aggregation (NEON / AVX2 flavour):
    if (fa) mode = 0;                                   //   PS then holds the halving-tree results instead of sums
    s.ts = (mode >= 2) ? 8 : 16;     // mode 2 = fused-layout kernel (mqsad), 3 = fused-layout kernel (MFMA accumulate)
    s.lay = (mode == 4) ? 2 : 0;     // mode 4 = QUAD layout kernel
    if (K % 64 || (ags != 32 && ags != 64 && ags != K)) return -1;
    if (mode >= 2 && ags == 32) return -1;
    if (fa && ags == K) return -1;
    std::vector<uint32_t> W(s.weight_u4() * 4);
    for (size_t i = 0; i < W.size(); ++i) W[i] = retile_dword(A_ref, s, i >> 2, (int)(i & 3));
    std::vector<uint32_t> QL(s.qlut_dev_u4() * 4, 0x80808080u);
    for (int t = 0; t < K / 4; ++t) {
        uint32_t lo = 0, hi = 0;
        for (int i = 0; i < 4; ++i) {
            lo |= (uint32_t)(qlut_ref[t * 16 + i] + 128) << (8 * i);
            hi |= (uint32_t)(qlut_ref[t * 16 + 4 + i] + 128) << (8 * i);
        }
        const int seg = t / TS, tls = t % TS;
        const size_t u2 = qlut_dev_u4_index(seg, tls >> 1) * 2 + (tls & 1);
        QL[u2 * 2] = lo; QL[u2 * 2 + 1] = hi;
    }
    const size_t n = (size_t)Mw * bits * (ags == K ? 1 : K / ags);
    memset(PS, 0, n * sizeof(int32_t));
    if (mode == 3) {
        const int nu = K / 32, tstride = ((nu + 15) & ~15) + 1;
        std::vector<uint32_t> QS((size_t)4 * tstride * 4, 0u);   // signed half tables
        for (int t = 0; t < K / 4; ++t) {
            uint32_t lo = 0, hi = 0;
            for (int i = 0; i < 4; ++i) {
                lo |= (uint32_t)(uint8_t)qlut_ref[t * 16 + i] << (8 * i);
                hi |= (uint32_t)(uint8_t)qlut_ref[t * 16 + 4 + i] << (8 * i);
            }
            const size_t u2 = ((size_t)((t & 7) >> 1) * tstride + (t >> 3)) * 2 + (t & 1);
            QS[u2 * 2] = lo; QS[u2 * 2 + 1] = hi;
        }
        switch (bits) {
            case 1: run8_mfma<1>(W, QS, s, PS); break;
            case 2: run8_mfma<2>(W, QS, s, PS); break;
            case 3: run8_mfma<3>(W, QS, s, PS); break;
            case 4: run8_mfma<4>(W, QS, s, PS); break;
            default: return -1;
        }
        return 0;
    }
    if (mode == 2 || mode == 4) {
        const int nu = K / 32, tstride = ((nu + 15) & ~15) + 1;
        std::vector<uint32_t> QLDS((size_t)4 * tstride * 4, 0x80808080u);
        for (int t = 0; t < K / 4; ++t) {
            uint32_t lo = 0, hi = 0;
            for (int i = 0; i < 4; ++i) {
                lo |= (uint32_t)(qlut_ref[t * 16 + i] + 128) << (8 * i);
                hi |= (uint32_t)(qlut_ref[t * 16 + 4 + i] + 128) << (8 * i);
            }
            const size_t u2 = ((size_t)((t & 7) >> 1) * tstride + (t >> 3)) * 2 + (t & 1);
            QLDS[u2 * 2] = lo; QLDS[u2 * 2 + 1] = hi;
        }
        if (mode == 4) {
            switch (bits) {
                case 1: run_quad<1>(W, QLDS, s, PS); break;
                case 2: run_quad<2>(W, QLDS, s, PS); break;
                case 3: run_quad<3>(W, QLDS, s, PS); break;
                case 4: run_quad<4>(W, QLDS, s, PS); break;
                default: return -1;
            }
            return 0;
        }
        switch (bits) {
            case 1: run8<1>(W, QLDS, s, PS); break;
            case 2: run8<2>(W, QLDS, s, PS); break;
            case 3: run8<3>(W, QLDS, s, PS); break;
            case 4: run8<4>(W, QLDS, s, PS); break;
            default: return -1;
        }
        return 0;
    }
    g_fa_xor = fa == 2 ? 0x80808080u : 0u;
#define RUN(B) (fa ? run<B, 2>(W, QL, s, PS) : mode == 0 ? run<B, 0>(W, QL, s, PS) : run<B, 1>(W, QL, s, PS))
    switch (bits) {
        case 1: RUN(1); break;
        case 2: RUN(2); break;
        case 3: RUN(3); break;
        case 4: RUN(4); break;
        default: return -1;
    }
    return 0;
}
