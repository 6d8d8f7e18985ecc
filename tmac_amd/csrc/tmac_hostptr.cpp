// tmac_hostptr.cpp — layer (1) of include/tmac_hip.h: the reference-named entry points taking HOST pointers
// (preprocessor_int8 / qgemm_lut_int8 and the shape-named kernels of deploy/tuned/<set>/kernels.h), as the llama.cpp fork
// binds them through include/t-mac/tmac_gemm_wrapper.h:170-228.
#include "tmac_host.h"

using namespace tmac_host;

struct TileKey {
    const void* A; int bm, K, bits;
    bool operator<(const TileKey& o) const {
        if (A != o.A) return A < o.A;
        if (bm != o.bm) return bm < o.bm;
        if (K != o.K) return K < o.K;
        return bits < o.bits;
    }
};
// The reference's caller (llama.cpp) walks the M-tiles of a matrix and calls qgemm_lut once per tile with the SAME LUT
// (tmac_gemm_wrapper.h:197-228).  Served literally that is a PCIe staging and a synchronisation per 64 rows.  So the
// host-pointer layer learns the matrices behind the tiles: tiles whose weight and scale pointers are contiguous (the
// reference layout stores a matrix tile after tile) form a RUN; once a run has been seen whole, the first tile call
// that arrives with a new LUT computes the run's entire output in one launch and the following tile calls are served
// from that result as long as the LUT they pass is the one it was computed from (same three pointers and a 192-byte sample
// of the table bytes on the shared-lock fast path; scales, biases and the whole table are compared when a pointer differs).
// verbatim sample of a tile's weight and scale bytes (4 x 16 + 2 x 16 bytes): what the per-call staleness check of a grouped
// tile compares (a memcmp of 96 bytes instead of a 192-byte hash per tile call)
struct TileBytes { unsigned char b[96]; };
static void tile_bytes(const void* A, size_t a_bytes, const void* S, size_t s_bytes, TileBytes& out) {
    const size_t n = a_bytes < 16 ? a_bytes : 16, m = s_bytes < 16 ? s_bytes : 16;
    memset(out.b, 0, sizeof(out.b));
    for (int i = 0; i < 4; ++i) memcpy(out.b + 16 * i, (const char*)A + (a_bytes - n) * i / 3, n);
    if (S && s_bytes) { memcpy(out.b + 64, S, m); memcpy(out.b + 80, (const char*)S + s_bytes - m, m); }
}
static bool tile_bytes_match(const void* A, size_t a_bytes, const void* S, size_t s_bytes, const TileBytes& ref) {
    const size_t n = a_bytes < 16 ? a_bytes : 16, m = s_bytes < 16 ? s_bytes : 16;
    for (int i = 0; i < 4; ++i) if (memcmp(ref.b + 16 * i, (const char*)A + (a_bytes - n) * i / 3, n) != 0) return false;
    if (S && s_bytes) return memcmp(ref.b + 64, S, m) == 0 && memcmp(ref.b + 80, (const char*)S + s_bytes - m, m) == 0;
    return true;
}

struct HostRun {
    tmac_hip_weights* w = nullptr;   // the run registered as one matrix
    int ntile = 0, Mw_tile = 0;
    // direct addressing of the run's tiles (they are contiguous in the caller's memory): tile i = (A0 + i * a_bytes, S0 + i * s_bytes)
    const char* A0 = nullptr; const char* S0 = nullptr;
    size_t a_bytes = 0, s_bytes = 0;
    int m = 0, k = 0, b = 0;
    std::vector<TileBytes> bytes;    // per tile: the sample the fast path compares
    float* C = nullptr;              // pinned host memory: [n][ntile * Mw_tile] outputs for LUT generation `gen`
    size_t C_elems = 0;
    unsigned long long gen = 0;
    unsigned long long used = 0;     // LRU stamp
    size_t dev_bytes = 0;
};
struct TileInfo {
    tmac_hip_weights* w = nullptr;   // the tile alone (first pass; released when a run takes over)
    const void* S = nullptr;         // its scale pointer
    HostRun* run = nullptr;
    int idx = 0;                     // tile index inside the run
    uint64_t sample = 0;             // hash of sampled weight + scale bytes at registration: a reused pointer with other
                                     // contents (model reload, in-place edit) is detected instead of served stale
    size_t a_bytes = 0, s_bytes = 0;
};
// Everything the layer remembers between calls, in ONE object with an explicit lifetime:
//   created   lazily, member by member, by the first calls that need it (stream, flag, workspace, staging);
//   cache     tiles / runs: dropped by tmac_hip_cache_clear() (and LRU-evicted above cache_cap_bytes);
//   released  workspace, staging buffers, LUT memo: host_route_release(), i.e. tmac_hip_reset_state();
//   never     destroyed at process exit (no HIP calls from static destructors: the runtime may be gone by then).
// Locking: `mu` shared = a tile call served from a computed run; exclusive (+ tmac_host::g_mu) = everything else.
struct HostRoute {
    std::map<TileKey, TileInfo> tiles;
    std::vector<HostRun*> runs;
    unsigned long long run_epoch = 1;   // bumped (under the exclusive lock) whenever a run is created or freed: per-thread run memos check it
    tmac_hip_workspace* ws = nullptr;   // the LUT the caller's host buffers hold, on the device
    void* stage = nullptr;              // device staging for C / B
    size_t stage_bytes = 0;
    void* pin = nullptr;                // pinned host staging (activations in, LUT out)
    size_t pin_bytes = 0;
    hipStream_t stream = nullptr;       // the layer's own (non-blocking) stream: async copies + launches, one completion wait per entry point
    std::shared_mutex mu;
    unsigned long long use_clock = 0;
    size_t cache_dev_bytes = 0, cache_cap_bytes = 0;
    // host copy of the LUT (qlut | lut_scales | lut_biases) that `ws` currently holds, its generation, and the caller's buffers it was taken from
    std::vector<unsigned char> lut_host;
    int lut_k = 0, lut_n = 0, lut_ags = 0;
    unsigned long long lut_gen = 0;
    const void *lut_q = nullptr, *lut_ls = nullptr, *lut_lb = nullptr;
    // completion flag in pinned host memory that the last launch of a call sets and the calling thread spins on
    uint32_t* flag = nullptr;
    uint32_t flag_gen = 0;
    int zero_copy = -1;                 // $TMAC_HIP_HOST_ZERO_COPY (default 1): 0 restores copy commands + hipStreamSynchronize
};
static HostRoute& H = *new HostRoute();     // heap object on purpose: see "never" above

static uint64_t fnv64(const void* p, size_t n, uint64_t h = 1469598103934665603ull) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}
// 4 x 32 bytes of the weight tile and 2 x 32 bytes of its scales
static uint64_t tile_sample(const void* A, size_t a_bytes, const void* S, size_t s_bytes) {
    uint64_t h = 1469598103934665603ull;
    const size_t n = a_bytes < 32 ? a_bytes : 32;
    for (int i = 0; i < 4; ++i) h = fnv64((const char*)A + (a_bytes - n) * i / 3, n, h);
    if (S && s_bytes) {
        const size_t m = s_bytes < 32 ? s_bytes : 32;
        h = fnv64(S, m, h);
        h = fnv64((const char*)S + s_bytes - m, m, h);
    }
    return h;
}

static int32_t host_stream() {
    if (H.stream) return TMAC_HIP_OK;
    HIP_TRY(hipStreamCreateWithFlags(&H.stream, hipStreamNonBlocking));
    if (const char* e = getenv("TMAC_HIP_HOST_CACHE_MB")) H.cache_cap_bytes = (size_t)atoll(e) << 20;
    if (!H.cache_cap_bytes) H.cache_cap_bytes = (size_t)64 << 30;      // weights cached on the device for host-pointer callers: 64 GB by default
    return TMAC_HIP_OK;
}
// Completion of the host-pointer calls: a flag in pinned host memory that the last launch of the call sets and this thread spins
// on (a hipStreamSynchronize costs ~8 us more per call); bounded, with the stream synchronisation as the fallback.
static bool host_zero_copy() {
    if (H.zero_copy < 0) { const char* e = getenv("TMAC_HIP_HOST_ZERO_COPY"); H.zero_copy = e ? atoi(e) != 0 : 1; }
    return H.zero_copy != 0;
}
static int32_t host_flag_init() {
    if (H.flag) return TMAC_HIP_OK;
    HIP_TRY(hipHostMalloc((void**)&H.flag, 64, hipHostMallocDefault));
    *H.flag = 0;
    return TMAC_HIP_OK;
}
static int32_t host_flag_wait(uint32_t val) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        if (__atomic_load_n(H.flag, __ATOMIC_ACQUIRE) == val) return TMAC_HIP_OK;
        if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
    }
    HIP_TRY(hipStreamSynchronize(H.stream));      // slow launch (first use, contention) or an error: the stream tells
    return TMAC_HIP_OK;
}

static int32_t host_ws(int K, int N) {
    if (H.ws && H.ws->maxK >= K && H.ws->maxN >= N) return TMAC_HIP_OK;
    if (H.ws) { if (H.stream) (void)hipStreamSynchronize(H.stream); tmac_hip_workspace_free(H.ws); }
    H.ws = nullptr;
    H.lut_k = 0;                 // a new workspace holds no LUT
    return tmac_hip_workspace_create(&H.ws, K, N);
}
static int32_t host_stage(size_t bytes) {
    if (H.stage_bytes >= bytes) return TMAC_HIP_OK;
    if (H.stream) (void)hipStreamSynchronize(H.stream);
    if (H.stage) (void)hipFree(H.stage);
    H.stage = nullptr; H.stage_bytes = 0;
    HIP_TRY(hipMalloc(&H.stage, bytes));
    H.stage_bytes = bytes;
    return TMAC_HIP_OK;
}
static int32_t host_pin(size_t bytes) {
    if (H.pin_bytes >= bytes) return TMAC_HIP_OK;
    if (H.stream) (void)hipStreamSynchronize(H.stream);
    if (H.pin) (void)hipHostFree(H.pin);
    H.pin = nullptr; H.pin_bytes = 0;
    HIP_TRY(hipHostMalloc(&H.pin, bytes, hipHostMallocDefault));
    H.pin_bytes = bytes;
    return TMAC_HIP_OK;
}

// Is the LUT the caller passes the one the workspace holds?  Same buffers as at the last full comparison and a matching
// sample: yes (llama.cpp builds the LUT once per matmul and passes it to every tile call).  Otherwise compare in full.
static bool lut_sample_ok(const void* q, size_t nq) {
    const size_t n = nq < 64 ? nq : 64;
    return memcmp(H.lut_host.data(), q, n) == 0 && memcmp(H.lut_host.data() + (nq - n) / 2, (const char*)q + (nq - n) / 2, n) == 0 &&
           memcmp(H.lut_host.data() + nq - n, (const char*)q + nq - n, n) == 0;
}
static bool lut_is_current(const void* q, const void* ls, const void* lb, int k, int n, int ags) {
    if (H.lut_k != k || H.lut_n != n || H.lut_ags != ags) return false;
    const size_t nq = (size_t)n * (k / 4) * 16, ns = sizeof(float) * (size_t)n * (k / ags);
    if (H.lut_host.size() != nq + 2 * ns) return false;
    if (q == H.lut_q && ls == H.lut_ls && lb == H.lut_lb)
        return lut_sample_ok(q, nq) && memcmp(H.lut_host.data() + nq, ls, ns) == 0 && memcmp(H.lut_host.data() + nq + ns, lb, ns) == 0;
    return memcmp(H.lut_host.data(), q, nq) == 0 && memcmp(H.lut_host.data() + nq, ls, ns) == 0 &&
           memcmp(H.lut_host.data() + nq + ns, lb, ns) == 0;
}
static void lut_remember(const void* q, const void* ls, const void* lb, int k, int n, int ags) {
    const size_t nq = (size_t)n * (k / 4) * 16, ns = sizeof(float) * (size_t)n * (k / ags);
    H.lut_host.resize(nq + 2 * ns);
    memcpy(H.lut_host.data(), q, nq);
    memcpy(H.lut_host.data() + nq, ls, ns);
    memcpy(H.lut_host.data() + nq + ns, lb, ns);
    H.lut_k = k; H.lut_n = n; H.lut_ags = ags;
    H.lut_q = q; H.lut_ls = ls; H.lut_lb = lb;
    ++H.lut_gen;
}

// first kcfg entry whose (k, n, b) match and, when bm_filter > 0, whose bm matches; looked up once per distinct key (the
// per-tile entry points come here on every call) -- the memo is dropped when the table changes
static void free_run(HostRun* r) {
    ++H.run_epoch;
    if (r->w) tmac_hip_free_weights(r->w);
    if (r->C) (void)hipHostFree(r->C);
    H.cache_dev_bytes -= r->dev_bytes < H.cache_dev_bytes ? r->dev_bytes : H.cache_dev_bytes;
    delete r;
}
// drop one run (or one lone tile) and every tile entry that points into it
static void evict_run(HostRun* r) {
    for (auto it = H.tiles.begin(); it != H.tiles.end();) it = (it->second.run == r) ? H.tiles.erase(it) : std::next(it);
    for (size_t i = 0; i < H.runs.size(); ++i) if (H.runs[i] == r) { H.runs.erase(H.runs.begin() + i); break; }
    free_run(r);
}
// least recently used runs go first when the device-side cache of host-pointer weights outgrows its cap
static void evict_to_cap(const HostRun* keep) {
    while (H.cache_dev_bytes > H.cache_cap_bytes && !H.runs.empty()) {
        HostRun* lru = nullptr;
        for (HostRun* r : H.runs) if (r != keep && (!lru || r->used < lru->used)) lru = r;
        if (!lru) break;
        if (H.stream) (void)hipStreamSynchronize(H.stream);
        evict_run(lru);
    }
}

extern "C" int32_t tmac_hip_cache_clear(void) {
    (void)hipDeviceSynchronize();
    defer_release_thread();                   // the calling thread's cached recordings of deferred batches (tmac_hip_defer)
    std::unique_lock<std::shared_mutex> hl(H.mu);
    std::lock_guard<std::mutex> lk(g_mu);
    if (H.stream) (void)hipStreamSynchronize(H.stream);
    for (auto& kv : H.tiles) if (kv.second.w) tmac_hip_free_weights(kv.second.w);
    H.tiles.clear();
    for (HostRun* r : H.runs) free_run(r);
    H.runs.clear();
    H.cache_dev_bytes = 0;
    release_fused_workspaces();
    return TMAC_HIP_OK;
}

// A/B knob: 0 = serve every tile call on its own (the literal reading of the reference's ABI), 1 = whole runs (default)
extern "C" int32_t tmac_hip_debug_host_runs(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_knobs.host_runs = on ? 1 : 0;
    return TMAC_HIP_OK;
}

extern "C" int32_t preprocessor_int8(int m, int k, int n, int b, void* B, void* LUT_Scales, void* LUT_Biases, void* QLUT) {
    bind_thread_device();
    if (!B || !LUT_Scales || !LUT_Biases || !QLUT) return fail(TMAC_HIP_E_ARG, "null argument");
    std::unique_lock<std::shared_mutex> hl(H.mu);
    std::lock_guard<std::mutex> lk(g_mu);
    tmac_kcfg cfg;
    // `m` is only a dispatch key in the reference too (qgemm.py:518-519)
    int fc = find_cfg(k, n, b, 0, m, &cfg, true);      // the LUT build depends on the act group size alone
    if (fc == 0) fc = find_cfg(k, n, b, 0, 0, &cfg, true);
    if (fc <= 0)
        return fail(TMAC_HIP_E_NOMATCH, fc ? "preprocessor_int8: the loaded kcfg sections for m=%d k=%d n=%d b=%d disagree on the act group size"
                                           : "preprocessor_int8: no kcfg for m=%d k=%d n=%d b=%d", m, k, n, b);
    int32_t rc = ensure_device();
    if (rc) return rc;
    if ((rc = host_stream())) return rc;
    if ((rc = host_ws(k, n))) return rc;
    const int ags = cfg.act_group_size;
    const size_t nb = sizeof(float) * (size_t)n * k, nq = (size_t)n * (k / 4) * 16, ns = sizeof(float) * (size_t)n * (k / ags);
    if ((rc = host_stage(nb))) return rc;
    if ((rc = host_pin(nb + nq + 2 * ns))) return rc;
    H.lut_k = 0;     // the workspace is about to change
    // pinned staging, everything asynchronous on the layer's own stream, ONE synchronisation:
    // activations up, LUT build, LUT (the caller owns it: tmac_gemm_wrapper.h:170-195) back down
    char* pin = (char*)H.pin;
    memcpy(pin, B, nb);
    if (host_zero_copy() && nq <= (1u << 20) && nq % 16 == 0 && ns % 16 == 0) {
        // small LUT (decode): the build reads the activations from the pinned buffer itself, one single-workgroup launch writes
        // the LUT back into it and raises the flag
        if ((rc = host_flag_init())) return rc;
        if ((rc = tmac_hip_preprocessor_dev(H.ws, pin, TMAC_F32, k, n, ags, H.stream))) return rc;
        const uint32_t val = ++H.flag_gen;
        hipError_t e = launch_host_copy3_flag(H.ws->qlut_ref, nq, H.ws->lut_scales, H.ws->lut_biases, ns, pin + nb, H.flag, val, H.stream);
        if (e != hipSuccess) return fail(TMAC_HIP_E_RUNTIME, "LUT copy-back launch: %s", hipGetErrorString(e));
        if ((rc = host_flag_wait(val))) return rc;
    } else {
        HIP_TRY(hipMemcpyAsync(H.stage, pin, nb, hipMemcpyHostToDevice, H.stream));
        if ((rc = tmac_hip_preprocessor_dev(H.ws, H.stage, TMAC_F32, k, n, ags, H.stream))) return rc;
        HIP_TRY(hipMemcpyAsync(pin + nb, H.ws->qlut_ref, nq, hipMemcpyDeviceToHost, H.stream));
        HIP_TRY(hipMemcpyAsync(pin + nb + nq, H.ws->lut_scales, ns, hipMemcpyDeviceToHost, H.stream));
        HIP_TRY(hipMemcpyAsync(pin + nb + nq + ns, H.ws->lut_biases, ns, hipMemcpyDeviceToHost, H.stream));
        HIP_TRY(hipStreamSynchronize(H.stream));
    }
    memcpy(QLUT, pin + nb, nq);
    memcpy(LUT_Scales, pin + nb + nq, ns);
    memcpy(LUT_Biases, pin + nb + nq + ns, ns);
    // the LUT the caller now holds is the one in the workspace: the qgemm calls that follow need not upload it again
    lut_remember(QLUT, LUT_Scales, LUT_Biases, k, n, ags);
    return TMAC_HIP_OK;
}

// the maximal run of registered, not yet grouped tiles around `key` with contiguous weight and scale pointers
static HostRun* build_run(const TileKey& key, const tmac_kcfg& cfg, int Mw_tile, size_t a_bytes, size_t s_bytes) {
    auto nb = [&](const TileKey& k, const TileInfo& ti, long d) -> std::map<TileKey, TileInfo>::iterator {
        TileKey kk = k;
        kk.A = (const char*)k.A + d * (long)a_bytes;
        auto it = H.tiles.find(kk);
        if (it == H.tiles.end() || it->second.run || !it->second.w) return H.tiles.end();
        if ((const char*)it->second.S != (const char*)ti.S + d * (long)s_bytes) return H.tiles.end();
        return it;
    };
    auto first = H.tiles.find(key);
    int n = 1;
    for (auto it = nb(first->first, first->second, -1); it != H.tiles.end(); it = nb(first->first, first->second, -1)) { first = it; ++n; }
    auto last = H.tiles.find(key);
    for (auto it = nb(last->first, last->second, +1); it != H.tiles.end(); it = nb(last->first, last->second, +1)) { last = it; ++n; }
    if (n < 2) return nullptr;
    tmac_kcfg rc = cfg;
    if (rc.m_groups >= 1) rc.m_groups = 1;
    tmac_hip_weights* w = nullptr;
    if (register_impl(&w, first->first.A, first->second.S, false, n * Mw_tile, key.K, key.bits, &rc, TMAC_F32, TMAC_F32, nullptr) != TMAC_HIP_OK)
        return nullptr;     // e.g. out of device memory: the tiles keep serving themselves
    HostRun* r = new HostRun();
    r->w = w; r->ntile = n; r->Mw_tile = Mw_tile;
    r->dev_bytes = w->w_bytes + w->sc_bytes;
    r->A0 = (const char*)first->first.A; r->S0 = (const char*)first->second.S; r->a_bytes = a_bytes; r->s_bytes = s_bytes;
    r->m = key.bm; r->k = key.K; r->b = key.bits;
    r->bytes.resize(n);
    for (int i = 0; i < n; ++i) tile_bytes(r->A0 + (size_t)i * a_bytes, a_bytes, r->S0 + (size_t)i * s_bytes, s_bytes, r->bytes[i]);
    H.cache_dev_bytes += r->dev_bytes;
    H.runs.push_back(r);
    ++H.run_epoch;
    TileKey kk = first->first;
    for (int i = 0; i < n; ++i) {
        TileInfo& ti = H.tiles[kk];
        if (ti.w) { H.cache_dev_bytes -= ti.w->w_bytes + ti.w->sc_bytes; tmac_hip_free_weights(ti.w); }
        ti.w = nullptr; ti.run = r; ti.idx = i;
        kk.A = (const char*)kk.A + a_bytes;
    }
    return r;
}

// copy one tile's rows out of a run's host result
static void serve_from_run(const HostRun* r, const TileInfo& ti, int n, int Mw_tile, void* C) {
    const size_t Mw_run = (size_t)r->ntile * Mw_tile;
    for (int i = 0; i < n; ++i)   // C tile is [n][Mw_tile] (kernels.cc:1068: C + n * bm/bits)
        memcpy((float*)C + (size_t)i * Mw_tile, r->C + (size_t)i * Mw_run + (size_t)ti.idx * Mw_tile, sizeof(float) * Mw_tile);
}

extern "C" int32_t qgemm_lut_int8(int m, int k, int n, int b, void* A, void* LUT, void* Scales, void* LUT_Scales,
                                  void* LUT_Biases, void* C) {
    bind_thread_device();
    if (!A || !LUT || !Scales || !LUT_Scales || !LUT_Biases || !C) return fail(TMAC_HIP_E_ARG, "null argument");
    const int Mw_tile = m / b;
    const TileKey key{A, m, k, b};
    {
        // Fast path, shared lock: the tile belongs to a run whose output for THIS LUT is already on the host.  This is what
        // llama.cpp's worker threads hit concurrently, one tile each (tmac_gemm_wrapper.h:197-199): they copy their rows out
        // side by side instead of queueing on one mutex.
        std::shared_lock<std::shared_mutex> sl(H.mu);
        // the run this thread served last (valid while no run has been created or freed since): its tiles are addressed
        // directly, no table lookup
        static thread_local const HostRun* memo_run = nullptr;
        static thread_local unsigned long long memo_epoch = 0;
        if (memo_run && memo_epoch == H.run_epoch) {
            const HostRun* r = memo_run;
            const ptrdiff_t da = (const char*)A - r->A0;
            if (r->m == m && r->k == k && r->b == b && da >= 0 && (size_t)da < r->a_bytes * (size_t)r->ntile && (size_t)da % r->a_bytes == 0) {
                const int idx = (int)((size_t)da / r->a_bytes);
                if ((const char*)Scales == r->S0 + (size_t)idx * r->s_bytes && r->gen == H.lut_gen && r->C &&
                    r->C_elems == (size_t)n * r->ntile * Mw_tile && LUT == H.lut_q && LUT_Scales == H.lut_ls && LUT_Biases == H.lut_lb &&
                    H.lut_k == k && H.lut_n == n && lut_sample_ok(LUT, (size_t)n * (k / 4) * 16) &&
                    tile_bytes_match(A, r->a_bytes, Scales, r->s_bytes, r->bytes[idx])) {
                    const size_t Mw_run = (size_t)r->ntile * Mw_tile;
                    for (int i = 0; i < n; ++i)
                        memcpy((float*)C + (size_t)i * Mw_tile, r->C + (size_t)i * Mw_run + (size_t)idx * Mw_tile, sizeof(float) * Mw_tile);
                    return TMAC_HIP_OK;
                }
            }
        }
        auto it = H.tiles.find(key);
        if (it != H.tiles.end() && it->second.run && it->second.S == Scales) {
            const TileInfo& ti = it->second;
            const HostRun* r = ti.run;
            if (r->gen == H.lut_gen && r->C && r->C_elems == (size_t)n * r->ntile * Mw_tile && LUT == H.lut_q && LUT_Scales == H.lut_ls &&
                LUT_Biases == H.lut_lb && H.lut_k == k && H.lut_n == n && lut_sample_ok(LUT, (size_t)n * (k / 4) * 16) &&
                tile_sample(A, ti.a_bytes, Scales, ti.s_bytes) == ti.sample) {
                serve_from_run(r, ti, n, Mw_tile, C);
                memo_run = r; memo_epoch = H.run_epoch;
                return TMAC_HIP_OK;
            }
        }
    }
    std::unique_lock<std::shared_mutex> hl(H.mu);
    std::lock_guard<std::mutex> lk(g_mu);
    tmac_kcfg cfg;
    const int fc = find_cfg(k, n, b, m, 0, &cfg);
    if (fc <= 0)
        return fail(TMAC_HIP_E_NOMATCH, fc ? "qgemm_lut_int8: the loaded kcfg sections with bm=%d k=%d n=%d b=%d disagree on the quantisation layout (load ONE kcfg.ini: tmac_hip_load_kcfg_ex(path, 1))"
                                           : "qgemm_lut_int8: no kcfg with bm=%d k=%d n=%d b=%d", m, k, n, b);
    int32_t rc = ensure_device();
    if (rc) return rc;
    if ((rc = host_stream())) return rc;
    tmac_kcfg tc = cfg;
    if (tc.m_groups >= 1) tc.m_groups = 1;  // a tile sees one unified scale
    Shape tshape;
    if ((rc = make_shape(tshape, Mw_tile, k, b, &tc))) return rc;
    const size_t a_bytes = ref_weight_bytes(tshape), s_bytes = ref_scale_elems(tshape) * sizeof(float);
    const uint64_t sample = tile_sample(A, a_bytes, Scales, s_bytes);
    auto it = H.tiles.find(key);
    if (it != H.tiles.end() && (it->second.S != Scales || it->second.sample != sample)) {
        // the pointer is known but its contents (or its scales) are not what was registered: a model was reloaded at the
        // same addresses, or edited in place.  Drop what was cached for it and register afresh.
        if (H.stream) (void)hipStreamSynchronize(H.stream);
        if (it->second.run) evict_run(it->second.run);
        else {
            if (it->second.w) { H.cache_dev_bytes -= it->second.w->w_bytes + it->second.w->sc_bytes; tmac_hip_free_weights(it->second.w); }
            H.tiles.erase(it);
        }
        it = H.tiles.end();
    }
    const bool known = it != H.tiles.end();
    if (!known) {
        TileInfo ti;
        if ((rc = register_impl(&ti.w, A, Scales, false, Mw_tile, k, b, &tc, TMAC_F32, TMAC_F32, nullptr))) return rc;
        ti.S = Scales; ti.sample = sample; ti.a_bytes = a_bytes; ti.s_bytes = s_bytes;
        H.cache_dev_bytes += ti.w->w_bytes + ti.w->sc_bytes;
        it = H.tiles.insert(std::make_pair(key, ti)).first;
    }
    if ((rc = host_ws(k, n))) return rc;
    if (!lut_is_current(LUT, LUT_Scales, LUT_Biases, k, n, cfg.act_group_size)) {
        H.lut_k = 0;
        if ((rc = tmac_hip_workspace_write(H.ws, (const int8_t*)LUT, (const float*)LUT_Scales, (const float*)LUT_Biases, k, n,
                                           cfg.act_group_size, H.stream)))
            return rc;
        lut_remember(LUT, LUT_Scales, LUT_Biases, k, n, cfg.act_group_size);
    }
    TileInfo& ti = it->second;
    // a tile that comes back (second GEMV on its matrix) with registered neighbours: group the run
    if (g_knobs.host_runs && known && !ti.run && ti.w) build_run(key, cfg, Mw_tile, a_bytes, s_bytes);
    if (ti.run) {
        HostRun* r = ti.run;
        r->used = ++H.use_clock;
        const size_t Mw_run = (size_t)r->ntile * Mw_tile, elems = (size_t)n * Mw_run;
        if (r->gen != H.lut_gen || r->C_elems != elems) {
            if (r->C_elems != elems) {
                if (r->C) (void)hipHostFree(r->C);
                r->C = nullptr; r->C_elems = 0;
                HIP_TRY(hipHostMalloc((void**)&r->C, sizeof(float) * elems, hipHostMallocDefault));
                r->C_elems = elems;
            }
            const size_t bytes = sizeof(float) * elems;
            if ((rc = host_stage(bytes))) return rc;
            // the whole run in one launch, its output straight into the run's pinned host buffer, one synchronisation
            if (host_zero_copy() && bytes <= (1u << 20)) {
                // the kernel stores the run's output into the pinned host buffer itself; a one-thread launch raises the flag
                if ((rc = host_flag_init())) return rc;
                if ((rc = qgemm_impl(r->w, H.ws, r->C, TMAC_F32, n, nullptr, H.stream))) return rc;
                const uint32_t val = ++H.flag_gen;
                hipError_t e = launch_host_flag(H.flag, val, H.stream);
                if (e != hipSuccess) return fail(TMAC_HIP_E_RUNTIME, "flag launch: %s", hipGetErrorString(e));
                if ((rc = host_flag_wait(val))) return rc;
            } else {
                if ((rc = qgemm_impl(r->w, H.ws, H.stage, TMAC_F32, n, nullptr, H.stream))) return rc;
                HIP_TRY(hipMemcpyAsync(r->C, H.stage, bytes, hipMemcpyDeviceToHost, H.stream));
                HIP_TRY(hipStreamSynchronize(H.stream));
            }
            r->gen = H.lut_gen;
        }
        serve_from_run(r, ti, n, Mw_tile, C);
        evict_to_cap(r);
        return TMAC_HIP_OK;
    }
    tmac_hip_weights* w = ti.w;
    const size_t cb = sizeof(float) * (size_t)n * Mw_tile;
    if ((rc = host_stage(cb))) return rc;
    if ((rc = host_pin(cb))) return rc;
    if ((rc = qgemm_impl(w, H.ws, H.stage, TMAC_F32, n, nullptr, H.stream))) return rc;
    HIP_TRY(hipMemcpyAsync(H.pin, H.stage, cb, hipMemcpyDeviceToHost, H.stream));
    HIP_TRY(hipStreamSynchronize(H.stream));
    memcpy(C, H.pin, cb);
    return TMAC_HIP_OK;
}

#define TMAC_DEF_Q(bm, k, n, b)                                                                                   \
    extern "C" int32_t qgemm_lut_t1_int8_m##bm##_k##k##_n##n##_b##b(void* A, void* LUT, void* Scales, void* LS,   \
                                                                     void* LB, void* C) {                         \
        return qgemm_lut_int8(bm, k, n, b, A, LUT, Scales, LS, LB, C);                                            \
    }
#define TMAC_DEF_P(m, k, n, b)                                                                                     \
    extern "C" int32_t preprocessor_t1_int8_m##m##_k##k##_n##n##_b##b(void* B, void* LS, void* LB, void* QLUT) {   \
        return preprocessor_int8(m, k, n, b, B, LS, LB, QLUT);                                                     \
    }
TMAC_DEF_Q(128, 4096, 1, 2) TMAC_DEF_Q(128, 11008, 1, 2)
TMAC_DEF_P(8192, 4096, 1, 2) TMAC_DEF_P(22016, 4096, 1, 2) TMAC_DEF_P(8192, 11008, 1, 2)
TMAC_DEF_Q(1024, 4096, 1, 4) TMAC_DEF_Q(256, 4096, 1, 4) TMAC_DEF_Q(256, 11008, 1, 4)
TMAC_DEF_P(16384, 4096, 1, 4) TMAC_DEF_P(44032, 4096, 1, 4) TMAC_DEF_P(16384, 11008, 1, 4)
TMAC_DEF_Q(256, 4096, 1, 2) TMAC_DEF_Q(512, 4096, 1, 2) TMAC_DEF_Q(128, 14336, 1, 2)
TMAC_DEF_P(28672, 4096, 1, 2) TMAC_DEF_P(8192, 14336, 1, 2) TMAC_DEF_P(2048, 4096, 1, 2)
TMAC_DEF_Q(128, 8640, 1, 2) TMAC_DEF_Q(128, 3200, 1, 2) TMAC_DEF_Q(320, 3200, 1, 2)
TMAC_DEF_P(6400, 8640, 1, 2) TMAC_DEF_P(17280, 3200, 1, 2) TMAC_DEF_P(6400, 3200, 1, 2)

// tmac_hip_reset_state: what tmac_hip_cache_clear leaves in place (the next call re-creates it)
void tmac_host::host_route_release() {
    std::unique_lock<std::shared_mutex> hl(H.mu);
    std::lock_guard<std::mutex> lk(g_mu);
    if (H.stream) (void)hipStreamSynchronize(H.stream);
    if (H.ws) { tmac_hip_workspace_free(H.ws); H.ws = nullptr; }
    if (H.stage) { (void)hipFree(H.stage); H.stage = nullptr; H.stage_bytes = 0; }
    if (H.pin) { (void)hipHostFree(H.pin); H.pin = nullptr; H.pin_bytes = 0; }
    H.lut_host.clear();
    H.lut_k = H.lut_n = H.lut_ags = 0;
    H.lut_q = H.lut_ls = H.lut_lb = nullptr;
    ++H.lut_gen;
}
