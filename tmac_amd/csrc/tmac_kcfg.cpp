// tmac_kcfg.cpp — the kcfg.ini table: same sections / keys as deploy/compile.py:153-165, lookup as
// include/t-mac/tmac_gemm_wrapper.h:230-255, plus the (bm, k, n, b) lookup the per-tile host-pointer entry points need.
#include "tmac_host.h"

using namespace tmac_host;

static std::map<std::string, tmac_kcfg> g_kcfg;          // guarded by tmac_host::g_mu
static unsigned long long g_kcfg_gen = 0;   // bumped whenever g_kcfg changes (memoised lookups check it)

static std::string section_name(int threads, int M_bits, int K, int N, int bits) {
    char buf[128];
    snprintf(buf, sizeof(buf), "qgemm_lut_t%d_int8_m%d_k%d_n%d_b%d", threads, M_bits, K, N, bits);
    return buf;
}

static void derive_kcfg(tmac_kcfg& c, int M_bits, int K, int N, int bits) {
    const int Mw = M_bits / bits;
    if (c.lut_scales_size > 0) c.act_group_size = (int)((long long)N * K / c.lut_scales_size);
    if (c.scales_size > 0 && c.scales_size < Mw) {
        c.m_groups = c.scales_size;
        c.zero_point = 0;
    } else {
        c.m_groups = -1;
        const long long per_row = c.group_size > 0 ? K / c.group_size : 1;
        c.zero_point = (c.scales_size == 2LL * Mw * per_row) ? 1 : 0;
    }
}

extern "C" int32_t tmac_hip_load_kcfg(const char* path) { return tmac_hip_load_kcfg_ex(path, 0); }

extern "C" int32_t tmac_hip_clear_kcfg(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_kcfg.clear();
    ++g_kcfg_gen;
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_load_kcfg_ex(const char* path, int replace) {
    std::string p = path ? path : "";
    if (p.empty()) {
        const char* e = getenv("TMAC_KCFG_FILE");  // tmac_gemm_wrapper.h:40-56
        if (!e) return fail(TMAC_HIP_E_ARG, "no kcfg path given and TMAC_KCFG_FILE is not set");
        p = e;
    }
    std::ifstream f(p);
    if (!f) return fail(TMAC_HIP_E_ARG, "cannot open kcfg file %s", p.c_str());
    std::lock_guard<std::mutex> lk(g_mu);
    std::string line, sec;
    std::map<std::string, std::map<std::string, long long>> raw;
    while (std::getline(f, line)) {
        size_t a = line.find_first_not_of(" \t\r\n");
        if (a == std::string::npos) continue;
        line = line.substr(a);
        if (line[0] == '#' || line[0] == ';') continue;
        if (line[0] == '[') {
            size_t b = line.find(']');
            if (b != std::string::npos) sec = line.substr(1, b - 1);
            continue;
        }
        size_t eq = line.find('=');
        if (eq == std::string::npos || sec.empty()) continue;
        std::string k = line.substr(0, eq), v = line.substr(eq + 1);
        k.erase(k.find_last_not_of(" \t") + 1);
        raw[sec][k] = atoll(v.c_str());
    }
    if (replace) { g_kcfg.clear(); ++g_kcfg_gen; }     // the file becomes the whole table (the reference holds exactly one kcfg.ini)
    for (auto& kv : raw) {
        int t, m, k, n, b;
        if (sscanf(kv.first.c_str(), "qgemm_lut_t%d_int8_m%d_k%d_n%d_b%d", &t, &m, &k, &n, &b) != 5) continue;
        tmac_kcfg c;
        memset(&c, 0, sizeof(c));
        auto& r = kv.second;
        c.bm = (int)r["bm"]; c.simd_n_in = (int)r["simd_n_in"]; c.simd_n_out = (int)r["simd_n_out"];
        c.kfactor = (int)r["kfactor"]; c.group_size = (int)r["group_size"];
        c.lut_scales_size = (int)r["lut_scales_size"]; c.scales_size = (int)r["scales_size"];
        c.n_tile_num = (int)r["n_tile_num"];
        derive_kcfg(c, m, k, n, b);
        g_kcfg[kv.first] = c;
        ++g_kcfg_gen;
    }
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_set_kcfg(int M, int K, int N, int bits, const tmac_kcfg* cfg) {
    if (!cfg) return fail(TMAC_HIP_E_ARG, "null cfg");
    std::lock_guard<std::mutex> lk(g_mu);
    g_kcfg[section_name(1, M * bits, K, N, bits)] = *cfg;
    ++g_kcfg_gen;
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_get_kcfg(int M, int K, int N, int bits, tmac_kcfg* out) {
    if (!out) return fail(TMAC_HIP_E_ARG, "null out");
    std::lock_guard<std::mutex> lk(g_mu);
    static const int hints[] = {1, 4, 8, 16};  // tmac_gemm_wrapper.h:233
    for (int t : hints) {
        auto it = g_kcfg.find(section_name(t, M * bits, K, N, bits));
        if (it != g_kcfg.end()) {
            *out = it->second;
            return TMAC_HIP_OK;
        }
    }
    return fail(TMAC_HIP_E_NOMATCH, "no kcfg section for m=%d k=%d n=%d b=%d", M * bits, K, N, bits);
}

void tmac_host::kcfg_clear_locked() {
    g_kcfg.clear();
    ++g_kcfg_gen;
}

// first kcfg entry whose (k, n, b) match and, when bm_filter > 0, whose bm matches; looked up once per distinct key (the
// per-tile entry points come here on every call) -- the memo is dropped when the table changes.  Caller holds g_mu.
static bool same_numerics(const tmac_kcfg& a, const tmac_kcfg& b) {
    return a.bm == b.bm && a.kfactor == b.kfactor && a.group_size == b.group_size && a.act_group_size == b.act_group_size &&
           a.zero_point == b.zero_point && (a.m_groups >= 1) == (b.m_groups >= 1);
}
// 1 = found, 0 = no section matches, -1 = several sections match and disagree on what the bytes mean
int tmac_host::find_cfg(int k, int n, int b, int bm_filter, int m_filter, tmac_kcfg* out, bool act_only) {
    static std::map<std::array<int, 6>, std::pair<int, tmac_kcfg>> memo;
    static unsigned long long memo_for = ~0ull;
    if (memo_for != g_kcfg_gen) { memo.clear(); memo_for = g_kcfg_gen; }
    const std::array<int, 6> mk = {k, n, b, bm_filter, m_filter, act_only ? 1 : 0};
    auto mi = memo.find(mk);
    if (mi != memo.end()) { *out = mi->second.second; return mi->second.first; }
    int found = 0;
    tmac_kcfg first;
    memset(&first, 0, sizeof(first));
    for (auto& kv : g_kcfg) {
        int t, m, kk, nn, bb;
        if (sscanf(kv.first.c_str(), "qgemm_lut_t%d_int8_m%d_k%d_n%d_b%d", &t, &m, &kk, &nn, &bb) != 5) continue;
        if (kk != k || nn != n || bb != b) continue;
        if (bm_filter > 0 && kv.second.bm != bm_filter) continue;
        if (m_filter > 0 && m != m_filter) continue;
        if (!found) { first = kv.second; found = 1; }
        // the reference compiles ONE kernel per (bm, k, n, b) name (deploy/compile.py:52-71): sections that share the key and
        // disagree on the quantisation layout cannot both be served through the per-tile entry point
        else if (act_only ? first.act_group_size != kv.second.act_group_size : !same_numerics(first, kv.second)) { found = -1; break; }
    }
    memo[mk] = std::make_pair(found, first);
    *out = first;
    return found;
}
