// tmac_chain_host.cpp — host side of the persistent decode chain (kernel: tmac_chain.hip).
#include "tmac_host.h"
#include <cstdlib>
#include <algorithm>

using namespace tmac_host;

// ---------------------------------------------------------------------------------------------
// Persistent decode chain (tmac_chain.hip): the fused calls of one decoded token recorded once, then executed by ONE
// launch.  Recording mirrors stream capture: between tmac_hip_chain_begin() and tmac_hip_chain_end() the calling thread's
// tmac_hip_qgemm_fused_dev calls (N = 1) are noted instead of launched; data flow is inferred from pointer identity (an
// op whose activation pointer equals an earlier op's output pointer consumes that output inside the launch).
// ---------------------------------------------------------------------------------------------
struct ChainRecOp {
    std::vector<const tmac_hip_weights*> w;
    const void* B;
    std::vector<void*> C;
    tmac_dtype_t act, out;
    tmac_hip_xform xf;               // vector transform of the activations (kind 0: none)
};
// an exchange step noted while recording: recv = all-gather over the ranks of send (rank r's bytes at r * bytes)
struct ChainRecGather {
    const void* send;
    const void* recv;
    size_t bytes;
    int rank, world;
    size_t pos;                       // number of calls recorded before it
};
static thread_local std::vector<ChainRecOp>* g_chain_rec = nullptr;
static thread_local std::vector<ChainRecGather>* g_chain_gat = nullptr;
static thread_local tmac_hip_xform g_chain_xf = {};                  // applies to the next recorded call
bool tmac_host::chain_recording() { return g_chain_rec != nullptr; }
void tmac_host::chain_clear_xform() { memset(&g_chain_xf, 0, sizeof(g_chain_xf)); }

struct tmac_hip_chain {
    std::vector<ChainOp> ops;
    ChainOp* d_ops = nullptr;
    unsigned* ctl = nullptr;
    // hand-off images of all consumed outputs: ONE arena (same layout on every rank of a row-sharded chain: a peer's address of
    // a granule is its arena base plus the local offset)
    void* arena = nullptr;
    size_t arena_bytes = 0;
    unsigned long long layout_hash = 0;
    int rank = 0, world = 1;
    std::vector<void*> peers;         // the other ranks' arenas, mapped through IPC (rank order, self skipped)
    bool connected = false;
    int bits = 0, zp = 0, sc_f16 = 0, out_f16 = 0;
    int sm = 0;                       // 0 per-group scales, 2 unified scale (k_decode_chain's SM)
    int grid = 0, buf_u4 = 0;
    size_t lds_bytes = 0;
    unsigned long long* stamps = nullptr;
    int32_t* tap = nullptr;           // parity tap (tmac_hip_chain_set_tap): caller's device buffer; per-op offsets (ints) on the device
    unsigned long long* d_tap_off = nullptr;
    size_t bytes = 0;                 // algorithmic weight + scale bytes of one launch
    int xforms = 0, carry_floats = 0;    // some op carries a vector transform; LDS floats of the kept vector
    int tmp_floats = 0, gam_floats = 0, ext_floats = 0, carry_K = 0;   // LDS floats of an op's own transform vector / norm weights; K of the latest kept vector
    int poll_sleep = 8, poll_delay = 4, issue_first = -1, poll_mode = 0, poll_grid = 0;   // read from the environment once, at tmac_hip_chain_end
    hipStream_t last_stream = nullptr;   // stream of the most recent launch (in-flight guard)
    bool launched = false;
    // stream mode (tmac_stream.hip): no op consumes another's output -- k_lut_images builds every op's tables once into `images`
    // (one image per op, the layout of the LDS LUT buffer), k_gemv_stream walks the ops with the tables prebuilt
    bool stream = false;
    void* images = nullptr;
    int max_nst = 0;
    const int* roles = nullptr;       // stream mode: the lookup waves' role records (device, behind the images), then the classes' visit counts
    const int* nvis = nullptr;
    int ncls = 1, vmax = 0;           // the schedule: classes of row ranges, records per class
    bool qw = false;                  // k_gemv_stream's quarter-walk form (rows dealt in groups of four quads: q_end / q_per / q_extra of the ops count groups)
    int nsplit = 1;                   // workgroups per row range (two share a CU and take alternate ops when LDS and registers allow)
};

int32_t tmac_host::chain_record(const tmac_hip_weights* const* wl, int nmat, const void* B_dev, tmac_dtype_t act_dtype,
                            void* const* C_list, tmac_dtype_t out_dtype, int N) {
    // (a rejected call must not leave its transform pending for the next recorded call)
    if (N != 1) { memset(&g_chain_xf, 0, sizeof(g_chain_xf)); return fail(TMAC_HIP_E_NOMATCH, "a decode chain records N = 1 calls only"); }
    ChainRecOp op;
    for (int i = 0; i < nmat; ++i) {
        if (!wl[i] || !C_list[i]) { memset(&g_chain_xf, 0, sizeof(g_chain_xf)); return fail(TMAC_HIP_E_ARG, "null matrix or output"); }
        op.w.push_back(wl[i]);
        op.C.push_back(C_list[i]);
    }
    op.B = B_dev; op.act = act_dtype; op.out = out_dtype;
    op.xf = g_chain_xf;
    memset(&g_chain_xf, 0, sizeof(g_chain_xf));
    g_chain_rec->push_back(op);
    return TMAC_HIP_OK;
}

// A vector transform for the NEXT recorded call (include/tmac_hip.h): validated when the chain is built.
extern "C" int32_t tmac_hip_chain_xform(const tmac_hip_xform* xf) {
    if (!g_chain_rec) return fail(TMAC_HIP_E_ARG, "no chain is being recorded on this thread");
    if (!xf) return fail(TMAC_HIP_E_ARG, "null transform");
    if (xf->kind < 0 || xf->kind > 2) return fail(TMAC_HIP_E_ARG, "unknown transform kind %d", xf->kind);
    if (xf->kind == TMAC_XF_GLU && !xf->in2) return fail(TMAC_HIP_E_ARG, "GLU needs a second vector");
    g_chain_xf = *xf;
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_chain_begin(void) {
    if (g_chain_rec) return fail(TMAC_HIP_E_ARG, "a chain is already being recorded on this thread");
    g_chain_rec = new std::vector<ChainRecOp>();
    g_chain_gat = new std::vector<ChainRecGather>();
    memset(&g_chain_xf, 0, sizeof(g_chain_xf));
    return TMAC_HIP_OK;
}

// Ends a recording without building anything (a caller whose recording hit an error: the thread is free to launch calls again).
extern "C" int32_t tmac_hip_chain_abort(void) {
    if (!g_chain_rec) return TMAC_HIP_OK;
    delete g_chain_rec;
    delete g_chain_gat;
    g_chain_rec = nullptr;
    g_chain_gat = nullptr;
    memset(&g_chain_xf, 0, sizeof(g_chain_xf));
    return TMAC_HIP_OK;
}

// The exchange step of a row-sharded chain, noted instead of executed (tmac_hip_comm_allgather calls this while the thread
// records): the calls that read recv_dev afterwards consume, inside the launch, what every rank's producer of send_dev publishes.
extern "C" int32_t tmac_hip_chain_record_gather(const void* send_dev, void* recv_dev, size_t bytes_per_rank, int rank, int world) {
    if (!g_chain_rec) return fail(TMAC_HIP_E_ARG, "no chain is being recorded on this thread");
    if (!send_dev || !recv_dev || !bytes_per_rank || world < 1 || world > 8 || rank < 0 || rank >= world)
        return fail(TMAC_HIP_E_ARG, "bad gather (1..8 ranks)");
    g_chain_gat->push_back(ChainRecGather{send_dev, recv_dev, bytes_per_rank, rank, world, g_chain_rec->size()});
    return TMAC_HIP_OK;
}
bool tmac_host::chain_record_gather_if_recording(const void* send_dev, void* recv_dev, size_t bytes_per_rank, int rank, int world, int32_t* rc) {
    if (!g_chain_rec) return false;
    *rc = tmac_hip_chain_record_gather(send_dev, recv_dev, bytes_per_rank, rank, world);
    return true;
}

static int chain_pick_wpq(int total_q, int nst, int grid, int nwv = CHAIN_NWV) {
    int best = 1;
    long best_cost = 1L << 60;
    for (int wpq = 1; wpq <= 4; ++wpq) {          // the combinations k_gemv_quad is instantiated for with this many threads
        if (nwv % wpq || (wpq > 1 && wpq > nst)) continue;
        const long ipi = nwv / wpq;
        const long cnt = (total_q + grid - 1) / grid;               // quads of the busiest workgroup (balanced contiguous ranges)
        const long iters = (cnt + ipi - 1) / ipi;
        const long steps = (nst + wpq - 1) / wpq;
        if (iters * steps < best_cost) { best_cost = iters * steps; best = wpq; }     // ties: fewer waves per quad (no LDS combine)
    }
    return best;
}

extern "C" int32_t tmac_hip_chain_free(tmac_hip_chain* c) {
    if (!c) return TMAC_HIP_OK;
    for (void* p : c->peers) if (p) (void)hipIpcCloseMemHandle(p);
    if (c->arena) (void)hipFree(c->arena);
    if (c->images) (void)hipFree(c->images);
    if (c->d_tap_off) (void)hipFree(c->d_tap_off);
    if (c->d_ops) (void)hipFree(c->d_ops);
    if (c->ctl) (void)hipFree(c->ctl);
    delete c;
    return TMAC_HIP_OK;
}

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// half-open byte ranges
struct Range { const char* lo; const char* hi; };
static bool overlap(const Range& a, const Range& b) { return a.lo < b.hi && b.lo < a.hi; }


// ---- the stream schedule (tmac_chain.h, StreamArgs): a pure function of the calls' sizes, so that it can be tested without a device ----
// items[i]: lookup items of call i; grid row ranges in ncls classes of consecutive ranges (class c = ranges ceil(c grid / ncls) ..).  Every
// call gets an aligned block of blk_w[i] classes starting at class blk_lo[i]: a call that would give a range fewer than `target` items at
// full width is dealt to 1 / n of the ranges (n a power of two <= cap); the cap that gives the shortest modelled launch is taken (a lone
// call keeps all ranges).  Calls go to the least loaded block, widest blocks first and larger calls first (lpt), or in recorded order.
// visits[c]: the calls class c visits, in visiting order.
static void stream_schedule(const std::vector<double>& items, int grid, int ncls, int target, bool lpt, std::vector<int>& blk_lo, std::vector<int>& blk_w,
                            std::vector<std::vector<int>>& visits) {
    const int nop = (int)items.size();
    const double visit_fixed = 8.0;                                    // a visit's fixed cost in items (model only)
    auto cls_lo = [&](int cl) { return (cl * grid + ncls - 1) / ncls; };
    blk_lo.assign(nop, 0); blk_w.assign(nop, ncls);
    visits.assign(ncls, std::vector<int>());
    double best_span = 0;
    int best_cap = 0;
    for (int pass = 0; pass < 2; ++pass) {
        for (int cap = ncls; cap >= 1; cap >>= 1) {
            if (pass == 1 && cap != best_cap) continue;
            std::vector<double> load(ncls, 0.0);
            std::vector<std::vector<int>> vis(ncls);
            // widest blocks first, larger calls first (longest-processing-time order: the calls are independent, so a class may visit
            // them in any order): in recorded order the classes of a llama-2-7B token ended 3 % apart, BitNet-3B's 8 % -- the launch
            // lasts as long as its most loaded class; in this order 0 % / 2 % (the same model)
            std::vector<int> order(nop), wof(nop);
            for (int i = 0; i < nop; ++i) {
                int n = 1;
                while (n < cap && items[i] * n / grid < target) n <<= 1;
                order[i] = i; wof[i] = ncls / n;
            }
            if (lpt)
                std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
                    if (wof[x] != wof[y]) return wof[x] > wof[y];
                    return items[x] > items[y];
                });
            for (int oi = 0; oi < nop; ++oi) {
                const int i = order[oi], w = wof[i];
                int bb = 0;
                double bl = 1e300;
                for (int b0 = 0; b0 + w <= ncls; b0 += w) {
                    double m = 0;
                    for (int k = b0; k < b0 + w; ++k) m = load[k] > m ? load[k] : m;
                    if (m < bl) { bl = m; bb = b0; }
                }
                const int wg = cls_lo(bb + w) - cls_lo(bb);
                for (int k = bb; k < bb + w; ++k) { load[k] += visit_fixed + items[i] / wg; vis[k].push_back(i); }
                if (pass == 1) { blk_lo[i] = bb; blk_w[i] = w; }
            }
            double span = 0;
            for (int k = 0; k < ncls; ++k) span = load[k] > span ? load[k] : span;
            if (pass == 0 && (best_cap == 0 || span < best_span * 0.999)) { best_span = span; best_cap = cap; }
            if (pass == 1) visits.swap(vis);
        }
    }
}
// test hook (no device needed): the schedule of n calls with the given item counts; out_lo / out_w [n], out_load [ncls] = items per range of every class
extern "C" int32_t tmac_hip_debug_stream_schedule(const double* items, int n, int grid, int ncls, int target, int lpt, int32_t* out_lo, int32_t* out_w, double* out_load) {
    if (!items || n < 1 || grid < 1 || ncls < 1 || ncls > 16 || (ncls & (ncls - 1)) || ncls > grid || !out_lo || !out_w) return fail(TMAC_HIP_E_ARG, "bad schedule query");
    std::vector<double> it(items, items + n);
    std::vector<int> lo, w;
    std::vector<std::vector<int>> vis;
    stream_schedule(it, grid, ncls, target, lpt != 0, lo, w, vis);
    for (int i = 0; i < n; ++i) { out_lo[i] = lo[i]; out_w[i] = w[i]; }
    if (out_load)
        for (int k = 0; k < ncls; ++k) {
            out_load[k] = 0;
            for (int i : vis[k]) {
                const int wg = ((lo[i] + w[i]) * grid + ncls - 1) / ncls - (lo[i] * grid + ncls - 1) / ncls;
                out_load[k] += it[i] / wg;
            }
        }
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_chain_end(tmac_hip_chain** out) {
    if (!g_chain_rec) return fail(TMAC_HIP_E_ARG, "no chain is being recorded on this thread");
    std::vector<ChainRecOp> rec;
    std::vector<ChainRecGather> gat;
    rec.swap(*g_chain_rec);
    gat.swap(*g_chain_gat);
    delete g_chain_rec;
    delete g_chain_gat;
    g_chain_rec = nullptr;
    g_chain_gat = nullptr;
    if (!out) return fail(TMAC_HIP_E_ARG, "null argument");
    *out = nullptr;
    if (rec.empty()) return fail(TMAC_HIP_E_ARG, "nothing was recorded");
    int32_t rc = ensure_device();
    if (rc) return rc;
    int dev = 0, cus = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (cus < 1) return fail(TMAC_HIP_E_RUNTIME, "no compute units reported");
    auto* c = new tmac_hip_chain();
    c->grid = (g_knobs.chain_grid > 0 && g_knobs.chain_grid < cus) ? g_knobs.chain_grid : cus;   // one workgroup per CU; residency is checked below
    for (const ChainRecGather& g : gat) {
        if (g.world != gat[0].world || g.rank != gat[0].rank) { tmac_hip_chain_free(c); return fail(TMAC_HIP_E_ARG, "the exchange steps of a chain share rank and world size"); }
        c->rank = g.rank; c->world = g.world;
    }
    const tmac_hip_weights* w0 = rec[0].w[0];
    c->bits = w0->s.bits; c->zp = w0->s.zero_point; c->sc_f16 = w0->sc_dtype == F16; c->out_f16 = rec[0].out == TMAC_F16;
    c->sm = (w0->s.m_groups >= 1) ? 2 : 0;
    auto bail = [&](int32_t code) { tmac_hip_chain_free(c); return code; };
    if (c->bits < 1 || c->bits > 4) return bail(fail(TMAC_HIP_E_NOMATCH, "the decode chain is built for 1- to 4-bit weights"));
    const size_t n = rec.size();
    const size_t out_esz = c->out_f16 ? 2 : 4;

    // ---- data flow and hazards, on byte RANGES (ops run on workgroups that are not synchronised with each other; an op is
    // ordered after another only through a hand-off: its activations are, transitively, made of the other's outputs) ----
    std::vector<Range> in_r(n);
    std::vector<std::vector<Range>> out_r(n);
    for (size_t i = 0; i < n; ++i) {
        const Shape& s0 = rec[i].w[0]->s;
        in_r[i] = Range{(const char*)rec[i].B, (const char*)rec[i].B + (size_t)s0.K * (rec[i].act == TMAC_F32 ? 4 : 2)};
        for (size_t m = 0; m < rec[i].w.size(); ++m)
            out_r[i].push_back(Range{(const char*)rec[i].C[m], (const char*)rec[i].C[m] + (size_t)rec[i].w[m]->s.Mw * out_esz});
        for (size_t m = 0; m < out_r[i].size(); ++m)
            for (size_t m2 = 0; m2 < m; ++m2)
                if (overlap(out_r[i][m], out_r[i][m2])) return bail(fail(TMAC_HIP_E_ARG, "op %zu: outputs %zu and %zu overlap", i, m2, m));
    }
    // source of every op's activations: the most recent earlier output that IS the range; partial overlap with an earlier
    // output cannot be handed over (the reader would see a mixture of launches) and is refused
    struct Src { int op, mat; };
    std::vector<Src> src(n, Src{-1, -1});
    std::vector<std::vector<char>> consumed(n), gathered(n);
    for (size_t i = 0; i < n; ++i) { consumed[i].assign(rec[i].w.size(), 0); gathered[i].assign(rec[i].w.size(), 0); }
    for (size_t i = 0; i < n; ++i) {
        // activations that are the result of an exchange step recorded before this call: the latest such step decides; its
        // producer is the latest earlier call that writes the gathered buffer's send side
        const ChainRecGather* via = nullptr;
        for (const ChainRecGather& g : gat)
            if (g.pos <= i && g.recv == rec[i].B && (!via || g.pos >= via->pos)) via = &g;
        if (via) {
            for (size_t j = via->pos; j-- > 0 && src[i].op < 0;)
                for (size_t m = 0; m < rec[j].C.size(); ++m)
                    if (rec[j].C[m] == via->send) {
                        if (via->bytes != (size_t)rec[j].w[m]->s.Mw * out_esz)
                            return bail(fail(TMAC_HIP_E_ARG, "op %zu: the exchange step gathers %zu bytes per rank, output %zu of op %zu has %zu", i, via->bytes, m, j,
                                             (size_t)rec[j].w[m]->s.Mw * out_esz));
                        if ((size_t)rec[i].w[0]->s.K > (size_t)via->world * rec[j].w[m]->s.Mw)
                            return bail(fail(TMAC_HIP_E_ARG, "op %zu reads %d activations from a gather of %d x %d rows", i, rec[i].w[0]->s.K, via->world, rec[j].w[m]->s.Mw));
                        src[i] = Src{(int)j, (int)m};
                        consumed[j][m] = 1; gathered[j][m] = 1;
                        break;
                    }
            if (src[i].op < 0) return bail(fail(TMAC_HIP_E_ARG, "op %zu reads a gathered buffer whose send side no earlier call of the chain writes", i));
            continue;
        }
        for (size_t j = i; j-- > 0 && src[i].op < 0;)
            for (size_t m = 0; m < rec[j].C.size(); ++m) {
                if (!overlap(in_r[i], out_r[j][m])) continue;
                if (rec[j].C[m] != rec[i].B)
                    return bail(fail(TMAC_HIP_E_NOMATCH, "op %zu reads activations that overlap output %zu of op %zu without being that output: "
                                                         "not representable as a hand-off", i, m, j));
                src[i] = Src{(int)j, (int)m};
                consumed[j][m] = 1;
                break;
            }
    }
    // the second vector of a GLU transform: the same rules as the activations (an earlier output, whole, or external memory)
    std::vector<Src> src2(n, Src{-1, -1});
    for (size_t i = 0; i < n; ++i) {
        if (rec[i].xf.kind != TMAC_XF_GLU) continue;
        const Range r2{(const char*)rec[i].xf.in2, (const char*)rec[i].xf.in2 + (size_t)rec[i].w[0]->s.K * 2};
        for (size_t j = i; j-- > 0 && src2[i].op < 0;)
            for (size_t m = 0; m < rec[j].C.size(); ++m) {
                if (!overlap(r2, out_r[j][m])) continue;
                if (rec[j].C[m] != rec[i].xf.in2)
                    return bail(fail(TMAC_HIP_E_NOMATCH, "op %zu: the GLU's second vector overlaps output %zu of op %zu without being that output", i, m, j));
                if (gathered[j][m]) return bail(fail(TMAC_HIP_E_NOMATCH, "op %zu: a gathered output as the second vector of a GLU is not covered", i));
                src2[i] = Src{(int)j, (int)m};
                consumed[j][m] = 1;
                break;
            }
        if ((src2[i].op >= 0) != (src[i].op >= 0))
            return bail(fail(TMAC_HIP_E_NOMATCH, "op %zu: the two vectors of a GLU must both be outputs of the chain or both be external", i));
    }
    // dep[k][i]: op k runs after op i has published everything (transitive closure over the hand-offs)
    std::vector<std::vector<char>> dep(n, std::vector<char>(n, 0));
    for (size_t k = 0; k < n; ++k) {
        if (src[k].op >= 0) {
            dep[k] = dep[src[k].op];
            dep[k][src[k].op] = 1;
        }
        if (src2[k].op >= 0) {
            for (size_t q = 0; q < n; ++q) dep[k][q] |= dep[src2[k].op][q];
            dep[k][src2[k].op] = 1;
        }
    }

    // GLU in the producer: when the two vectors of a GLU are outputs 0 and 1 of ONE earlier two-matrix call (gate and up) and nothing else
    // in the chain reads the gate's hand-off image, the gate/up call publishes silu(gate) * up itself -- once per row, by the wave that
    // publishes the rows anyway -- instead of all 256 workgroups of the reader evaluating K x (exp + rcp) each and polling two images
    // (A/B: TMAC_CHAIN_GLU_EPILOGUE=0 keeps the reader's form).  Needs both quads of a row pair in one workgroup iteration: pairs are dealt.
    std::vector<char> epi_of(n, 0), glu_in_producer(n, 0);
    std::vector<std::vector<int>> readers(n);
    for (size_t j = 0; j < n; ++j) readers[j].assign(rec[j].w.size(), 0);
    for (size_t i = 0; i < n; ++i) {
        if (src[i].op >= 0) ++readers[src[i].op][src[i].mat];
        if (src2[i].op >= 0) ++readers[src2[i].op][src2[i].mat];
    }
    if (env_int("TMAC_CHAIN_GLU_EPILOGUE", 1))
        for (size_t i = 0; i < n; ++i) {
            if (rec[i].xf.kind != TMAC_XF_GLU || src[i].op < 0 || src2[i].op != src[i].op || src[i].mat != 0 || src2[i].mat != 1) continue;
            const size_t j = (size_t)src[i].op;
            if (rec[j].w.size() != 2 || rec[j].w[0]->s.Mw != rec[j].w[1]->s.Mw || gathered[j][0] || gathered[j][1] || readers[j][0] != 1 || epi_of[j]) continue;
            const Shape& sj = rec[j].w[0]->s;
            const int nqj = 2 * sj.nquads(), nstj = (sj.K / 32 + 63) / 64;
            const int wq = g_knobs.chain_wpq ? g_knobs.chain_wpq : chain_pick_wpq(nqj, nstj, c->grid);
            if (CHAIN_NWV % wq || ((CHAIN_NWV / wq) & 1)) continue;          // pairs need an even number of quads per workgroup iteration
            epi_of[j] = 1; glu_in_producer[i] = 1;
            if (readers[j][1] == 1) consumed[j][1] = 0;                      // nobody else reads the up projection through a hand-off
        }

    c->ops.resize(n);
    int maxK = 0;
    for (size_t i = 0; i < n; ++i) {
        const ChainRecOp& r = rec[i];
        ChainOp& o = c->ops[i];
        memset(&o, 0, sizeof(o));
        const Shape& s0 = r.w[0]->s;
        if (r.act != TMAC_F16 && r.act != TMAC_F32) return bail(fail(TMAC_HIP_E_NOMATCH, "op %zu: the decode chain takes fp16 or fp32 activations", i));
        if (r.act == TMAC_F32 && (src[i].op >= 0 || r.xf.kind == TMAC_XF_GLU))
            return bail(fail(TMAC_HIP_E_NOMATCH, "op %zu: fp32 activations are covered for vectors in memory (an earlier output is handed over as fp16), without a GLU transform", i));
        if ((r.out == TMAC_F16) != (c->out_f16 != 0)) return bail(fail(TMAC_HIP_E_NOMATCH, "op %zu: one output dtype per chain", i));
        if (s0.K > 8 * 3 * CHAIN_FT) return bail(fail(TMAC_HIP_E_NOMATCH, "op %zu: K = %d beyond the decode chain's %d", i, s0.K, 8 * 3 * CHAIN_FT));
        if (((s0.m_groups >= 1) ? 2 : 0) != c->sm) return bail(fail(TMAC_HIP_E_NOMATCH, "op %zu: per-group and unified scales cannot share a chain", i));
        int gu = 1;
        if (c->sm == 2) {
            if (s0.ags != s0.K || s0.K % 64 || s0.m_groups > CHAIN_US_MAX_GROUPS)
                return bail(fail(TMAC_HIP_E_NOMATCH, "op %zu: the unified-scale chain covers one act group per row (act_group_size == K) and up to %d scales per matrix",
                                 i, CHAIN_US_MAX_GROUPS));
        } else {
            gu = s0.gs / 32;
            if (s0.ags != 64 || s0.gs < 128 || (gu & (gu - 1)) || s0.K % s0.gs || s0.K % 64)
                return bail(fail(TMAC_HIP_E_NOMATCH, "op %zu: the decode chain covers per-group scales (group >= 128, power of two) with act groups of 64", i));
        }
        int nq = 0;
        for (size_t m = 0; m < r.w.size(); ++m) {
            const tmac_hip_weights* w = r.w[m];
            const Shape& a = w->s;
            if (a.lay != 2 || !w->lo_ok || w->fa) return bail(fail(TMAC_HIP_E_NOMATCH, "op %zu matrix %zu is not registered in the QUAD layout", i, m));
            if (a.K != s0.K || a.bits != c->bits || a.gs != s0.gs || a.ags != s0.ags || a.zero_point != c->zp || a.m_groups != s0.m_groups ||
                (w->sc_dtype == F16) != (c->sc_f16 != 0))
                return bail(fail(TMAC_HIP_E_ARG, "op %zu: the matrices of a chain share bits, zero points and scale dtype; those of an op also K and group size", i));
            if (c->sm == 2 && a.Mw % a.m_groups) return bail(fail(TMAC_HIP_E_NOMATCH, "op %zu matrix %zu: rows not divisible by m_groups", i, m));
            nq += a.nquads();
            o.m[m].W = (const uint4*)w->W; o.m[m].SC = w->SC; o.m[m].C = r.C[m]; o.m[m].Mw = a.Mw; o.m[m].q_end = nq;
            o.m[m].GR = nullptr;
            if (consumed[i][m]) {
                if (r.out != TMAC_F16) return bail(fail(TMAC_HIP_E_NOMATCH, "op %zu: outputs consumed inside the chain must be fp16", i));
                if (gathered[i][m] && a.Mw % 4) return bail(fail(TMAC_HIP_E_NOMATCH, "op %zu: row shards of a gathered output must be whole row quads", i));
                // image of the output as its consumers see it: the rows of ALL ranks when it goes through an exchange step (rank r's
                // quads from r * nquads on); GR holds the OFFSET for now, the arena is allocated once all images are known
                const size_t nq_img = (size_t)a.nquads() * (gathered[i][m] ? c->world : 1);
                const size_t my_off = gathered[i][m] ? (size_t)c->rank * a.nquads() * 16 : 0;
                o.m[m].GR = reinterpret_cast<uint4*>(c->arena_bytes + my_off + 1);       // (+ 1: offset 0 is a valid image; fixed up below)
                c->layout_hash = (c->layout_hash ^ (nq_img * 16 + i * 4 + m)) * 1099511628211ull;
                c->arena_bytes += (nq_img * 16 + 255) & ~(size_t)255;
            }
            c->bytes += w->w_bytes + w->sc_bytes;
        }
        o.nmat = (int)r.w.size();
        for (int m = 0; m < 4; ++m) o.q_end[m] = (m < o.nmat - 1) ? o.m[m].q_end : 0x7fffffff;
        o.K = s0.K; o.nu = s0.K / 32; o.nst = (o.nu + 63) / 64; o.tstride = o.nst * 64 + 1;
        o.G = s0.K / 64; o.GP = o.nst * 32; o.nsg = c->sm == 2 ? 1 : s0.K / s0.gs;
        o.gs_shift = 0;
        for (int g = gu; g > 1; g >>= 1) ++o.gs_shift;
        o.m_groups = c->sm == 2 ? s0.m_groups : 0;
        o.total_q = nq;
        o.wpq = g_knobs.chain_wpq ? g_knobs.chain_wpq : chain_pick_wpq(nq, o.nst, c->grid);
        if (CHAIN_NWV % o.wpq) return bail(fail(TMAC_HIP_E_ARG, "waves per quad must divide %d", CHAIN_NWV));
        o.ipi = CHAIN_NWV / o.wpq;
        o.wpq_inv = (65536 + o.wpq - 1) / o.wpq;
        o.ipi_inv = (65536 + o.ipi - 1) / o.ipi;
        if (nq / c->grid + 1 + o.ipi >= 4096) return bail(fail(TMAC_HIP_E_NOMATCH, "op %zu: too many rows per workgroup for the decode chain", i));
        o.q_per = nq / c->grid; o.q_extra = nq % c->grid;
        if (epi_of[i]) { o.epi = 1; c->xforms = 1; o.q_per = (nq / 2) / c->grid; o.q_extra = (nq / 2) % c->grid; }
        if (src[i].op >= 0) {
            const ChainOp& po = c->ops[src[i].op];
            const bool via_gather = gathered[src[i].op][src[i].mat] != 0;
            if (!via_gather && po.m[src[i].mat].Mw != o.K)
                return bail(fail(TMAC_HIP_E_ARG, "op %zu reads an output of %d rows as %d activations", i, po.m[src[i].mat].Mw, o.K));
            // the image's first quad (a gathered image starts rank * nquads before this rank's part); offsets until the arena exists
            const size_t my_off = via_gather ? (size_t)c->rank * ((po.m[src[i].mat].Mw + 3) / 4) * 16 : 0;
            o.in = reinterpret_cast<const void*>(reinterpret_cast<size_t>(po.m[src[i].mat].GR) - my_off); o.in_gran = 1;
        } else {
            o.in = r.B; o.in_gran = 0;
            if (r.act == TMAC_F32) { o.in_gran = 2; c->xforms = 1;      // (the kernel instance with the extensions)
                if (chain_xf_region_floats(s0.K) > c->ext_floats) c->ext_floats = chain_xf_region_floats(s0.K); }
        }
        // ---- vector transform (tmac_hip_chain_xform)
        {
            const tmac_hip_xform& xf = r.xf;
            o.xf_kind = xf.kind;
            if (xf.kind == TMAC_XF_NORM) {
                c->xforms = 1;
                if (o.K > 2 * 8 * CHAIN_FT) return bail(fail(TMAC_HIP_E_NOMATCH, "op %zu: a NORM transform is covered up to K = %d", i, 2 * 8 * CHAIN_FT));
                if (o.K > 8192 && (xf.keep || xf.residual == TMAC_XF_CARRY))
                    return bail(fail(TMAC_HIP_E_NOMATCH, "op %zu: a kept residual vector is covered up to K = 8192", i));
                const int rf = chain_xf_region_floats(o.K);
                if (xf.residual == TMAC_XF_CARRY) {
                    if (c->carry_K != o.K) return bail(fail(TMAC_HIP_E_ARG, "op %zu: no earlier NORM of the chain keeps a vector of %d values", i, o.K));
                    o.xf_flags |= 2;
                } else o.res = xf.residual;
                if (xf.keep) { o.xf_flags |= 4; c->carry_K = o.K; if (rf > c->carry_floats) c->carry_floats = rf; }
                else if (rf > c->tmp_floats) c->tmp_floats = rf;
                if (xf.gamma && rf > c->gam_floats) c->gam_floats = rf;
                o.gamma = xf.gamma; o.res_out = xf.residual_out;
                memcpy(&o.eps_bits, &xf.eps, 4);
            } else if (xf.kind == TMAC_XF_GLU && glu_in_producer[i]) {
                o.xf_kind = TMAC_XF_NONE;                  // `in` already holds silu(gate) * up (the producer's epilogue)
            } else if (xf.kind == TMAC_XF_GLU) {
                c->xforms = 1;
                if (o.K > 2 * 8 * CHAIN_FT) return bail(fail(TMAC_HIP_E_NOMATCH, "op %zu: a GLU transform is covered up to K = %d", i, 2 * 8 * CHAIN_FT));
                if (chain_xf_region_floats(o.K) > c->tmp_floats) c->tmp_floats = chain_xf_region_floats(o.K);
                if (src2[i].op >= 0) {
                    const ChainOp& p2 = c->ops[src2[i].op];
                    if (p2.m[src2[i].mat].Mw != o.K) return bail(fail(TMAC_HIP_E_ARG, "op %zu: the GLU's second vector has %d rows, K = %d", i, p2.m[src2[i].mat].Mw, o.K));
                    o.in2 = p2.m[src2[i].mat].GR;        // (offset + 1 until the arena exists, like `in`)
                } else o.in2 = xf.in2;
            }
        }
        // weight fragments per wave in front of the polls: ONE.  The poll then returns after a fabric round trip plus 24 KB per
        // CU, and the wait behind it does not hold the LUT build until all of the op's weights have landed.  Putting the whole
        // ring in front ("start the stream at once") measured slower even for the ops whose stream outlasts the hand-off:
        // llama-2-7B W2 0.764 -> 0.714 ms, W4 1.028 -> 0.983 ms per token (profiles/r03_chain_knobs.txt).
        o.in_gran |= 1 << 8;
        if (o.K > maxK) maxK = o.K;
    }
    // Hazards between ops that no hand-off orders.  Every workgroup reads an op's activations itself (each builds the whole
    // LUT) and walks the ops in recorded order.  "Op j has published" therefore implies "every workgroup is past op i" for
    // any i <= j only when every workgroup owns rows of op j (q_per >= 1).  A later op k may overwrite an EXTERNAL input of
    // op i (a decoder's "next x = last output") exactly when such an op j lies between them on k's hand-off path.
    // Inputs handed over inside the launch are read from the hand-off image, never from the user-visible buffer.
    auto all_past = [&](size_t i, size_t k) {
        for (size_t j = i; j < k; ++j)
            if (dep[k][j] && c->ops[j].q_per >= 1) return true;
        return false;
    };
    for (size_t k = 0; k < n; ++k)
        for (size_t m = 0; m < out_r[k].size(); ++m)
            for (size_t i = 0; i < k; ++i) {
                if (src[i].op < 0 && overlap(out_r[k][m], in_r[i]) && !all_past(i, k))
                    return bail(fail(TMAC_HIP_E_NOMATCH, "op %zu overwrites activations that op %zu reads from memory and nothing in the chain orders the two "
                                                         "(no hand-off path from an op in which every workgroup owns rows): launch these calls one by one", k, i));
                for (size_t m2 = 0; m2 < out_r[i].size(); ++m2)
                    if (overlap(out_r[k][m], out_r[i][m2]) && !dep[k][i])
                        return bail(fail(TMAC_HIP_E_NOMATCH, "ops %zu and %zu write overlapping outputs and nothing in the chain orders them", i, k));
            }
    // The vectors of the transforms take part in the same analysis: what a NORM / GLU reads from memory (residual, norm weights, an
    // external second vector) must not be written by the launch unless a hand-off orders the writer behind every reader; what a NORM
    // writes (residual_out: stored by the workgroups that own the index range, while the others may still be reading) must not be read
    // by the same or a later op of the launch (a later NORM takes it as TMAC_XF_CARRY), nor overlap any output.
    std::vector<std::vector<Range>> xf_rd(n);
    std::vector<Range> xf_wr(n, Range{nullptr, nullptr});
    for (size_t i = 0; i < n; ++i) {
        const tmac_hip_xform& xf = rec[i].xf;
        const size_t K = (size_t)rec[i].w[0]->s.K;
        if (xf.kind == TMAC_XF_NORM) {
            if (xf.residual && xf.residual != TMAC_XF_CARRY) xf_rd[i].push_back(Range{(const char*)xf.residual, (const char*)xf.residual + K * 4});
            if (xf.gamma) xf_rd[i].push_back(Range{(const char*)xf.gamma, (const char*)xf.gamma + K * 4});
            if (xf.residual_out) xf_wr[i] = Range{(const char*)xf.residual_out, (const char*)xf.residual_out + K * 4};
        } else if (xf.kind == TMAC_XF_GLU && src2[i].op < 0) {
            xf_rd[i].push_back(Range{(const char*)xf.in2, (const char*)xf.in2 + K * 2});
        }
    }
    for (size_t k = 0; k < n; ++k) {
        for (size_t i = 0; i < n; ++i)
            for (const Range& r : xf_rd[i]) {
                for (size_t m = 0; m < out_r[k].size(); ++m)
                    if (overlap(out_r[k][m], r) && (i >= k || !all_past(i, k)))
                        return bail(fail(TMAC_HIP_E_NOMATCH, "op %zu writes output %zu over a vector that the transform of op %zu reads from memory and nothing in the "
                                                             "chain orders the writer behind every reader", k, m, i));
                if (xf_wr[k].lo && overlap(xf_wr[k], r) && (i >= k || !all_past(i, k)))
                    return bail(fail(TMAC_HIP_E_NOMATCH, "op %zu: residual_out overlaps a vector that the transform of op %zu reads from memory (a later NORM of the "
                                                         "launch takes the kept vector, TMAC_XF_CARRY)", k, i));
            }
        if (!xf_wr[k].lo) continue;
        for (size_t i = 0; i < n; ++i) {
            if (src[i].op < 0 && overlap(xf_wr[k], in_r[i]) && (i >= k || !all_past(i, k)))
                return bail(fail(TMAC_HIP_E_NOMATCH, "op %zu: residual_out overlaps activations that op %zu reads from memory", k, i));
            for (size_t m = 0; m < out_r[i].size(); ++m)
                if (overlap(xf_wr[k], out_r[i][m])) return bail(fail(TMAC_HIP_E_NOMATCH, "op %zu: residual_out overlaps output %zu of op %zu", k, m, i));
            if (i != k && xf_wr[i].lo && overlap(xf_wr[k], xf_wr[i])) return bail(fail(TMAC_HIP_E_NOMATCH, "ops %zu and %zu: overlapping residual_out vectors", i, k));
        }
    }
    // ---- stream mode: nothing is handed over and nothing is transformed -- the calls are independent (SURVEY 8d's back-to-back GEMVs;
    // a caller that evaluates many vectors against many matrices).  The reference's call structure then applies as it stands: tables
    // once per activation vector (llama_cpp_init), lookups per matrix (llama_cpp_compute); see tmac_stream.hip.  The hazard analysis
    // above has already refused every write of the launch that touches a vector another op reads from memory.  TMAC_CHAIN_STREAM=0: A/B.
    {
        bool indep = (c->sm == 0 || c->sm == 2) && c->world == 1 && gat.empty() && env_int("TMAC_CHAIN_STREAM", 1) != 0;
        for (size_t i = 0; i < n && indep; ++i)
            if (src[i].op >= 0 || src2[i].op >= 0 || rec[i].xf.kind != TMAC_XF_NONE || c->ops[i].epi) indep = false;
        if (indep) {
            size_t img_bytes = 0;
            int buf = 0;
            const std::vector<ChainOp> as_chain = c->ops;                     // (restored when the recording stays with k_decode_chain after all)
            for (ChainOp& o : c->ops) {
                o.img_u4 = stream_img_u4(o.K);
                o.img = reinterpret_cast<const void*>(img_bytes + 1);       // offset + 1 until the images exist
                img_bytes += (size_t)o.img_u4 * 16;
                if (o.img_u4 > buf) buf = o.img_u4;
                if (o.nst > c->max_nst) c->max_nst = o.nst;
                o.in_gran &= 2;                                               // (no fragments-in-front-of-the-polls count: there are no polls)
            }
            // ---- the schedule (tmac_chain.h, StreamArgs): which row ranges visit which op.  The per-visit costs of k_gemv_stream (two barriers,
            // the image, the waves' op change, waves without items in a short op: ~1.5 us per op whatever its size, profiles/r05_stream_knockouts.txt)
            // are paid per (workgroup, visit): an op that gives a row range fewer than `target` items is dealt to 1 / n of the ranges (n a power
            // of two) with n times the rows each, and the other classes of ranges work on other ops meanwhile.  Ops go to the least loaded aligned
            // block of classes in recorded order; the cap on n that gives the shortest modelled launch is taken (a lone call keeps all ranges).
            // TMAC_STREAM_NCLS=1: every range visits every op (the round-5 form; A/B).
            const int nwv = STREAM_NLW;
            int ncls = env_int("TMAC_STREAM_NCLS", 16);
            if (ncls < 1) ncls = 1;
            if (ncls > 16) ncls = 16;
            while (ncls > c->grid || (ncls & (ncls - 1))) --ncls;
            const int target = env_int("TMAC_STREAM_VISIT_ITEMS", 160);      // (sweep: profiles/r06_stream_schedule_sweep.txt)
            const int nop = (int)c->ops.size();
            auto cls_lo = [&](int cl, int nc) { return (cl * c->grid + nc - 1) / nc; };
            // The quarter-walk form of the kernel (tmac_stream.hip, QW): rows dealt in groups of four quads, K walked in quarters of a 64-unit
            // step.  It saves the lookups a ragged last step wastes (K = 11008, 3200, 8640 ...) and is the faster form even without one
            // (profiles/r06_stream_qw.txt), so 1- and 2-bit recordings take it whenever every matrix has whole groups (rows % 16 == 0);
            // TMAC_STREAM_QW=0 keeps the (quad x 64 units) form, whose per-group-scale outputs are bit-identical to the stand-alone launches.
            bool qw = true;
            {
                double it64 = 0, it16 = 0;
                for (const ChainOp& o : c->ops) {
                    for (int m = 0; m < o.nmat; ++m) if (((o.m[m].Mw + 3) / 4) % 4) qw = false;
                    it64 += (double)o.total_q * o.nst; it16 += (double)(o.total_q / 4) * ((o.nu + 15) / 16);
                }
                const int force = env_int("TMAC_STREAM_QW", -1);
                // measured (profiles/r06_stream_qw.txt): 1- to 3-bit streams are bound by lookup issue -- any saved item pays, and the form is
                // 2-4 % faster even at K = 4096; 4-bit streams run at the memory system's rate, where a wave-load of four 256-byte pieces
                // instead of one KB costs ~7 %: taken there only when the ragged steps outweigh that
                // (it16 <= it64 always; 1- to 3-bit: the form whenever the rows allow it)
                // 3- and 4-bit streams run at the memory system's rate already (6.1-6.4 TB/s), where four 256-byte pieces per wave-load cost
                // more than the ragged step's lookups: W3 4096 x 11008 0.75 -> 0.65 with the form, W4 equal (profiles/r06_stream_qw.txt)
                if (force == 0 || (force < 0 && c->bits >= 3 && it16 > 0.85 * it64)) qw = false;
            }
            auto op_q = [&](const ChainOp& o) { return qw ? o.total_q / 4 : o.total_q; };               // row units dealt: groups | quads
            auto op_nst = [&](const ChainOp& o) { return qw ? (o.nu + 15) / 16 : o.nst; };              // K steps walked: quarters | 64-unit steps
            std::vector<int> blk_lo, blk_w;
            std::vector<std::vector<int>> visits;
            {
                std::vector<double> items(nop);
                for (int i = 0; i < nop; ++i) items[i] = (double)op_q(c->ops[i]) * op_nst(c->ops[i]);
                stream_schedule(items, c->grid, ncls, target, env_int("TMAC_STREAM_LPT", 1) != 0, blk_lo, blk_w, visits);
            }
            bool sched_ok = true;
            for (int i = 0; i < nop; ++i) {
                ChainOp& o = c->ops[i];
                const int wlo = cls_lo(blk_lo[i], ncls), wcnt = cls_lo(blk_lo[i] + blk_w[i], ncls) - wlo;
                o.wg_lo = wlo;
                const int tq = op_q(o), nstw = op_nst(o);
                if (!g_knobs.chain_wpq || nwv % o.wpq || o.wpq > nstw) o.wpq = chain_pick_wpq(tq, nstw, wcnt, nwv);
                o.ipi = nwv / o.wpq;
                o.wpq_inv = (65536 + o.wpq - 1) / o.wpq;
                o.ipi_inv = (65536 + o.ipi - 1) / o.ipi;
                o.q_per = tq / wcnt; o.q_extra = tq % wcnt;
                if (tq / wcnt + 1 + o.ipi >= 4096) sched_ok = false;
                if (qw) for (int m = 0; m < 4; ++m) { if (o.q_end[m] != 0x7fffffff) o.q_end[m] /= 4; o.m[m].q_end /= 4; }      // the kernel counts groups
            }
            int vmax = 1;
            for (int k = 0; k < ncls; ++k) if ((int)visits[k].size() > vmax) vmax = (int)visits[k].size();
            // two workgroups per CU, alternate visits each (k_gemv_stream's nsplit): when both fit a CU's LDS.  TMAC_STREAM_SPLIT=1: A/B
            const int want_split = env_int("TMAC_STREAM_SPLIT", 2);
            const size_t lds2 = stream_lds_bytes(buf, (vmax + 1) / 2, qw);
            size_t lds = stream_lds_bytes(buf, vmax, qw);
            // Two workgroups are co-resident on a CU only with <= 64 VGPRs and <= 80 SGPRs each (measured, profiles/r05_stream_stamps.txt): 1- to
            // 3-bit weights fit with two fragments in flight per wave (3-bit: 4.06 -> 3.55 us on 4096 x 11008); 4-bit ones only with one,
            // which loses to one workgroup with two (5.15 against 4.68 us): they keep one workgroup per CU.  TMAC_STREAM_SPLIT_BITS: A/B.
            if (want_split >= 2 && c->bits <= env_int("TMAC_STREAM_SPLIT_BITS", 3) && vmax >= 2 && 2 * lds2 + 2048 <= 160 * 1024) { c->nsplit = 2; lds = lds2; }
            for (int ns = 3; ns <= want_split && ns <= 4; ++ns) {            // (A/B builds with -DTMAC_STREAM_NLW=6: more, smaller workgroups per CU)
                const size_t ldsn = stream_lds_bytes(buf, (vmax + ns - 1) / ns, qw);
                if (c->nsplit == ns - 1 && vmax >= ns && ns * (ldsn + 1024) <= 160 * 1024) { c->nsplit = ns; lds = ldsn; }
            }
            if (sched_ok && lds <= 160 * 1024) {
                // the waves' role records (tmac_chain.h) behind the images: one per (class, visit), then the classes' visit counts
                std::vector<int32_t> roles((size_t)STREAM_ROLE_INTS * ncls * vmax + ncls, 0);
                for (int k = 0; k < ncls; ++k) {
                    roles[(size_t)STREAM_ROLE_INTS * ncls * vmax + k] = (int32_t)visits[k].size();
                    for (size_t v = 0; v < visits[k].size(); ++v) {
                        const int i = visits[k][v];
                        const ChainOp& o = c->ops[i];
                        int32_t* r = roles.data() + ((size_t)k * vmax + v) * STREAM_ROLE_INTS;
                        const int nstw = op_nst(o);
                        r[SR_NST] = nstw; r[SR_IPI] = o.ipi; r[SR_NSG] = o.nsg; r[SR_GSH] = o.gs_shift; r[SR_NU] = o.nu;
                        r[SR_QE0] = o.q_end[0]; r[SR_QE1] = o.q_end[1]; r[SR_QE2] = o.q_end[2];
                        r[SR_QPER] = o.q_per; r[SR_QEXTRA] = o.q_extra;
                        r[SR_IT_LO] = (o.q_per + o.ipi - 1) / o.ipi; r[SR_IT_HI] = (o.q_per + o.ipi) / o.ipi;
                        r[SR_TSTRIDE] = o.tstride; r[SR_OP] = i; r[SR_WPQ] = o.wpq; r[SR_WLO] = o.wg_lo;
                        for (int wl = 0; wl < STREAM_NLW; ++wl) {
                            int32_t* rw = r + SR_COMMON + SRW_INTS * wl;
                            const int qs = wl / o.wpq, h = wl - qs * o.wpq;
                            auto nq = [&](int cnt) { return qs < cnt ? (cnt - 1 - qs) / o.ipi + 1 : 0; };
                            rw[SRW_NQ] = nq(o.q_per) | (nq(o.q_per + 1) << 16);
                            rw[SRW_NSTEPS] = h < nstw ? (nstw - h + o.wpq - 1) / o.wpq : 0;
                            rw[SRW_H] = h; rw[SRW_QS] = qs;
                        }
                    }
                }
                const size_t role_bytes = roles.size() * sizeof(int32_t);
                if (hipMalloc(&c->images, img_bytes + role_bytes) != hipSuccess || hipMemset(c->images, 0, img_bytes) != hipSuccess ||
                    hipMemcpy(reinterpret_cast<char*>(c->images) + img_bytes, roles.data(), role_bytes, hipMemcpyHostToDevice) != hipSuccess)
                    return bail(fail(TMAC_HIP_E_RUNTIME, "LUT image allocation failed (%zu bytes)", img_bytes + role_bytes));
                c->roles = reinterpret_cast<const int*>(reinterpret_cast<char*>(c->images) + img_bytes);
                c->nvis = c->roles + (size_t)STREAM_ROLE_INTS * ncls * vmax;
                c->ncls = ncls; c->vmax = vmax; c->qw = qw;
                for (ChainOp& o : c->ops) o.img = reinterpret_cast<const char*>(c->images) + (reinterpret_cast<size_t>(o.img) - 1);
                c->stream = true; c->buf_u4 = buf; c->lds_bytes = lds; c->xforms = 0;
            } else {
                c->ops = as_chain; c->nsplit = 1;
            }
        }
    }
    if (!c->stream) {
    c->buf_u4 = chain_buf_u4(maxK);
    c->lds_bytes = chain_lds_bytes(c->buf_u4, (int)c->ops.size(), c->carry_floats + c->tmp_floats + c->gam_floats + c->ext_floats);
    if (c->lds_bytes > 160 * 1024)
        return bail(fail(TMAC_HIP_E_NOMATCH, "%zu calls with K up to %d need %zu bytes of LDS (LUT buffers + call descriptors): record shorter chains",
                         c->ops.size(), maxK, c->lds_bytes));
    // one workgroup per CU must be resident at once: does the kernel fit a CU at all with this much LDS?
    {
        ChainArgs probe;
        memset(&probe, 0, sizeof(probe));
        probe.nops = 1;
        int resident = 0;
        hipError_t e = launch_decode_chain(probe, c->bits, c->zp != 0, c->sc_f16 != 0, c->sm, c->grid, c->lds_bytes, nullptr, &resident);
        if (e == hipErrorInvalidValue) return bail(fail(TMAC_HIP_E_NOMATCH, "no decode-chain kernel for this configuration"));
        if (e != hipSuccess || resident < 1)
            return bail(fail(TMAC_HIP_E_NOMATCH, "the decode chain's workgroup does not fit a compute unit (%s, %zu bytes of LDS)",
                             e == hipSuccess ? "occupancy 0" : hipGetErrorString(e), c->lds_bytes));
    }
    }   // !stream
    if (c->arena_bytes) {
        // Peers write into this arena over xGMI while the local kernel polls it: fine-grained device memory (coherent across
        // devices inside a running kernel; coarse-grained allocations promise that at kernel boundaries only).  Single-GPU
        // chains keep the ordinary allocation.
        // Two halves, used by generation parity: a rank that has finished launch g may publish the first granules of launch g + 1
        // while a slower peer still polls the images of launch g (it cannot get further ahead: launch g + 1 needs the peer's rows).
        const hipError_t ea = c->world > 1 ? hipExtMallocWithFlags(&c->arena, 2 * c->arena_bytes, hipDeviceMallocFinegrained)
                                           : hipMalloc(&c->arena, 2 * c->arena_bytes);
        if (ea != hipSuccess || hipMemset(c->arena, 0, 2 * c->arena_bytes) != hipSuccess)
            return bail(fail(TMAC_HIP_E_RUNTIME, "hand-off arena allocation failed (%zu bytes)", 2 * c->arena_bytes));
        const size_t base = reinterpret_cast<size_t>(c->arena) - 1;       // (offsets were stored + 1)
        for (ChainOp& o : c->ops) {
            for (int m = 0; m < o.nmat; ++m)
                if (o.m[m].GR) o.m[m].GR = reinterpret_cast<uint4*>(reinterpret_cast<size_t>(o.m[m].GR) + base);
            if (o.in_gran & 1) o.in = reinterpret_cast<const void*>(reinterpret_cast<size_t>(o.in) + base);
            if ((o.in_gran & 1) && o.xf_kind == TMAC_XF_GLU) o.in2 = reinterpret_cast<const void*>(reinterpret_cast<size_t>(o.in2) + base);
        }
    }
    c->connected = c->world == 1;
    if (hipMalloc((void**)&c->d_ops, sizeof(ChainOp) * c->ops.size()) != hipSuccess ||
        hipMemcpy(c->d_ops, c->ops.data(), sizeof(ChainOp) * c->ops.size(), hipMemcpyHostToDevice) != hipSuccess)
        return bail(fail(TMAC_HIP_E_RUNTIME, "descriptor upload failed"));
    const unsigned ctl0[4] = {1u, 0u, 0u, 0u};
    if (hipMalloc((void**)&c->ctl, sizeof(ctl0)) != hipSuccess || hipMemcpy(c->ctl, ctl0, sizeof(ctl0), hipMemcpyHostToDevice) != hipSuccess)
        return bail(fail(TMAC_HIP_E_RUNTIME, "control word allocation failed"));
    // the granule fills above ran on the null stream; the chain is launched on the caller's (possibly non-blocking) stream
    if (hipStreamSynchronize(nullptr) != hipSuccess) return bail(fail(TMAC_HIP_E_RUNTIME, "hand-off buffer initialisation failed"));
    // A workgroup reaches the polls of an op right after publishing its own share of the previous one: the first poll cannot
    // succeed before the slowest producer's stores have crossed the fabric (~1 us), and every failed poll is 16 KB per workgroup
    // of fabric traffic that the stores compete with.  Waiting ~0.75 us before the first poll and ~0.5 us between polls:
    // 0.757 -> 0.735 ms per llama-2-7B token (profiles/r02_chain_prefetch_ab.txt, E).  Knobs are read here, once per chain.
    c->poll_sleep = env_int("TMAC_CHAIN_POLL_SLEEP", 8);
    c->poll_delay = env_int("TMAC_CHAIN_POLL_DELAY", 4);
    c->issue_first = env_int("TMAC_CHAIN_ISSUE_FIRST", -1);
    c->poll_mode = env_int("TMAC_CHAIN_POLL_MODE", 0);
    c->poll_grid = env_int("TMAC_CHAIN_POLL_GRID", 0);
    *out = c;
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_chain_launch(tmac_hip_chain* c, void* stream) {
    bind_thread_device();
    if (!c) return fail(TMAC_HIP_E_ARG, "null chain");
    hipStream_t st = (hipStream_t)stream;
    // One launch of a chain at a time (its hand-off buffers and control words are per chain): launches on ONE stream are
    // ordered by the stream; a launch on another stream is refused while the previous one may still be running.
    if (c->launched && st != c->last_stream && hipStreamQuery(c->last_stream) != hipSuccess) {
        (void)hipGetLastError();
        return fail(TMAC_HIP_E_ARG, "the chain is still in flight on another stream: synchronise it first, or record one chain per stream");
    }
    if (!c->connected) return fail(TMAC_HIP_E_ARG, "a row-sharded chain must be connected to its peers first (tmac_hip_chain_export / tmac_hip_chain_connect)");
    if (c->stream) {
        hipError_t e = launch_lut_images(c->d_ops, (int)c->ops.size(), c->max_nst, c->sm, c->sc_f16 != 0, st);
        if (e != hipSuccess) return fail(TMAC_HIP_E_RUNTIME, "LUT image launch: %s", hipGetErrorString(e));
        StreamArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.ops = c->d_ops; sa.nops = (int)c->ops.size(); sa.out_f16 = c->out_f16; sa.buf_u4 = c->buf_u4; sa.nsplit = c->nsplit; sa.roles = c->roles; sa.stamps = c->stamps;
        sa.ncls = c->ncls; sa.vmax = c->vmax; sa.nvis = c->nvis; sa.tap = c->tap; sa.tap_off = c->d_tap_off;
        e = launch_gemv_stream(sa, c->bits, c->zp != 0, c->sc_f16 != 0, c->sm, c->qw, c->grid, c->lds_bytes, st);
        if (e != hipSuccess) return fail(TMAC_HIP_E_RUNTIME, "stream launch: %s", hipGetErrorString(e));
        c->last_stream = st; c->launched = true;
        return TMAC_HIP_OK;
    }
    ChainArgs a;
    memset(&a, 0, sizeof(a));
    a.ops = c->d_ops; a.nops = (int)c->ops.size(); a.ctl = c->ctl; a.out_f16 = c->out_f16;
    a.arena_base = reinterpret_cast<unsigned long long>(c->arena); a.arena_half = c->arena_bytes;
    a.npeer = (int)c->peers.size();
    for (int p = 0; p < a.npeer; ++p) a.peer_base[p] = reinterpret_cast<unsigned long long>(c->peers[p]);
    a.spin_limit = g_knobs.chain_spin_limit; a.buf_u4 = c->buf_u4; a.stamps = c->stamps;
    a.xforms = c->xforms; a.carry_floats = c->carry_floats; a.tmp_floats = c->tmp_floats; a.gam_floats = c->gam_floats; a.ext_floats = c->ext_floats;
    {   // measurement only (tools/gpu): run a chain WITHOUT transforms through the kernel instance that knows them
        static const int force_xf = [] { const char* e = getenv("TMAC_HIP_CHAIN_FORCE_XF"); return e && e[0] == '1' ? 1 : 0; }();
        if (force_xf) a.xforms = 1;
    }
    if (c->tap) { a.tap = c->tap; a.tap_off = c->d_tap_off; a.xforms = 1; }      // the tap has an instance of its own (XF + TAP), launch_decode_chain_b*
    a.poll_sleep = c->poll_sleep; a.poll_delay = c->poll_delay; a.issue_first = c->issue_first; a.poll_mode = c->poll_mode; a.poll_grid = c->poll_grid;
    hipError_t e = launch_decode_chain(a, c->bits, c->zp != 0, c->sc_f16 != 0, c->sm, c->grid, c->lds_bytes, st);
    if (e == hipErrorInvalidValue) return fail(TMAC_HIP_E_NOMATCH, "no decode-chain kernel for this configuration");
    if (e != hipSuccess) return fail(TMAC_HIP_E_RUNTIME, "decode chain launch: %s", hipGetErrorString(e));
    c->last_stream = st; c->launched = true;
    return TMAC_HIP_OK;
}

// ---- row-sharded chains: the ranks exchange the IPC handles of their hand-off arenas (any transport: MPI, torch.distributed, a
// file), then every producer stores its granules into all of them ----
struct ChainBlob {
    hipIpcMemHandle_t handle;
    unsigned long long arena_bytes, layout_hash;
    int rank, world;
};
static_assert(sizeof(ChainBlob) <= TMAC_HIP_CHAIN_BLOB_BYTES, "blob size");

extern "C" int32_t tmac_hip_chain_export(const tmac_hip_chain* c, void* blob_out) {
    if (!c || !blob_out) return fail(TMAC_HIP_E_ARG, "null argument");
    if (!c->arena) return fail(TMAC_HIP_E_ARG, "the chain hands nothing over");
    ChainBlob b;
    memset(&b, 0, sizeof(b));
    HIP_TRY(hipIpcGetMemHandle(&b.handle, c->arena));
    b.arena_bytes = 2 * c->arena_bytes; b.layout_hash = c->layout_hash; b.rank = c->rank; b.world = c->world;
    memset(blob_out, 0, TMAC_HIP_CHAIN_BLOB_BYTES);
    memcpy(blob_out, &b, sizeof(b));
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_chain_connect(tmac_hip_chain* c, const void* blobs, int world) {
    if (!c || !blobs) return fail(TMAC_HIP_E_ARG, "null argument");
    if (world != c->world) return fail(TMAC_HIP_E_ARG, "the chain was recorded for %d ranks, %d blobs given", c->world, world);
    if (c->connected && c->world > 1) return fail(TMAC_HIP_E_ARG, "the chain is already connected");
    bind_thread_device();
    for (int r = 0; r < world; ++r) {
        ChainBlob b;
        memcpy(&b, (const char*)blobs + (size_t)r * TMAC_HIP_CHAIN_BLOB_BYTES, sizeof(b));
        if (b.rank != r || b.world != world || b.arena_bytes != 2 * c->arena_bytes || b.layout_hash != c->layout_hash)
            return fail(TMAC_HIP_E_ARG, "rank %d recorded a different chain (its blob does not match this rank's hand-off layout)", r);
        if (r == c->rank) continue;
        void* p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, b.handle, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            for (void* q : c->peers) if (q) (void)hipIpcCloseMemHandle(q);
            c->peers.clear();
            return fail(TMAC_HIP_E_RUNTIME, "hipIpcOpenMemHandle of rank %d's hand-off arena: %s", r, hipGetErrorString(e));
        }
        c->peers.push_back(p);
    }
    c->connected = true;
    return TMAC_HIP_OK;
}

// After the stream has been synchronised: 0 = every hand-off completed; otherwise the error word of the first wave that
// gave up (bit 31 | op << 8 | wave) -- the outputs are then invalid.  Clears the word and re-arms the chain.
extern "C" int32_t tmac_hip_chain_status(tmac_hip_chain* c, uint32_t* error_word) {
    if (!c || !error_word) return fail(TMAC_HIP_E_ARG, "null argument");
    unsigned ctl[4];
    HIP_TRY(hipMemcpy(ctl, c->ctl, sizeof(ctl), hipMemcpyDeviceToHost));
    *error_word = ctl[2];
    if (ctl[2] || ctl[1]) {
        // A launch that gave up still ran to its end and advanced the generation like any other (every wait is bounded, the last
        // workgroup out advances): the generation stays in step with the peers of a row-sharded chain, which count launches the same
        // way -- re-arming one rank here would leave it a generation apart from its peers for good.  Only the error word (and a partial
        // exit count, should the launch have been killed from outside) is cleared.
        // A partial exit count means the launch never advanced the generation: the next launch reuses it and the same arena half, where
        // the granules the dead launch already published carry a matching tag.  They are wiped (tag 0 is never a generation), so the
        // next launch waits for data of its own.
        if (ctl[1] && c->world > 1)
            // ... on ONE GPU.  A row-sharded launch that died part-way has also stored granules into the peers' arenas under a generation the
            // peers have meanwhile left behind: the ranks are a generation apart and no local wipe repairs that
            return fail(TMAC_HIP_E_RUNTIME, "a row-sharded chain was interrupted inside a launch (%u of its workgroups exited): the ranks' generations differ now -- "
                                            "free the chain on every rank, record and connect it again", ctl[1]);
        if (ctl[1] && c->arena) HIP_TRY(hipMemset(c->arena, 0, 2 * c->arena_bytes));
        const unsigned fresh[4] = {ctl[0], 0u, 0u, 0u};
        HIP_TRY(hipMemcpy(c->ctl, fresh, sizeof(fresh), hipMemcpyHostToDevice));
    }
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_chain_info(const tmac_hip_chain* c, int op, int32_t* nops, int32_t* wpq, int32_t* grid, size_t* bytes) {
    if (!c) return fail(TMAC_HIP_E_ARG, "null chain");
    if (nops) *nops = (int32_t)c->ops.size();
    if (grid) *grid = c->grid;
    if (bytes) *bytes = c->bytes;
    if (wpq) {
        if (op < 0 || op >= (int)c->ops.size()) return fail(TMAC_HIP_E_ARG, "op index out of range");
        *wpq = c->ops[op].wpq;
    }
    return TMAC_HIP_OK;
}

// Parity tap of the persistent kernels (include/tmac_hip.h): the integers of every recorded call as they enter the float part.
extern "C" int32_t tmac_hip_chain_tap_layout(const tmac_hip_chain* c, int op, size_t* offset_ints, size_t* count_ints) {
    if (!c) return fail(TMAC_HIP_E_ARG, "null chain");
    if (op < 0 || op > (int)c->ops.size()) return fail(TMAC_HIP_E_ARG, "op index out of range");
    std::vector<unsigned long long> off(c->ops.size() + 1, 0);
    for (size_t i = 0; i < c->ops.size(); ++i)
        off[i + 1] = off[i] + (unsigned long long)4 * c->ops[i].total_q * (c->sm == 2 ? c->bits : c->ops[i].G);
    if (offset_ints) *offset_ints = (size_t)off[op];
    if (count_ints) *count_ints = op < (int)c->ops.size() ? (size_t)(off[op + 1] - off[op]) : 0;
    return TMAC_HIP_OK;
}
extern "C" int32_t tmac_hip_chain_set_tap(tmac_hip_chain* c, int32_t* dev_buffer) {
    if (!c) return fail(TMAC_HIP_E_ARG, "null chain");
    if (!dev_buffer) { c->tap = nullptr; return TMAC_HIP_OK; }
    for (const ChainOp& o : c->ops)
        if (o.epi) return fail(TMAC_HIP_E_NOMATCH, "the tap does not cover calls whose row quads are dealt in gate / up pairs (GLU in the producer)");
    bind_thread_device();
    if (!c->d_tap_off) {
        std::vector<unsigned long long> off(c->ops.size() + 1, 0);
        for (size_t i = 0; i < c->ops.size(); ++i)
            off[i + 1] = off[i] + (unsigned long long)4 * c->ops[i].total_q * (c->sm == 2 ? c->bits : c->ops[i].G);
        if (hipMalloc((void**)&c->d_tap_off, off.size() * sizeof(unsigned long long)) != hipSuccess ||
            hipMemcpy(c->d_tap_off, off.data(), off.size() * sizeof(unsigned long long), hipMemcpyHostToDevice) != hipSuccess)
            return fail(TMAC_HIP_E_RUNTIME, "tap offsets: allocation failed");
    }
    c->tap = dev_buffer;
    return TMAC_HIP_OK;
}

// profiling aid: s_memrealtime stamps [ops][workgroups][8] of wave 0 (layout: tmac_chain.h)
extern "C" int32_t tmac_hip_chain_set_stamps(tmac_hip_chain* c, unsigned long long* dev_buffer) {
    if (!c) return fail(TMAC_HIP_E_ARG, "null chain");
#ifndef TMAC_STREAM_STAMPS
    // k_gemv_stream writes stamps in profiling builds only, and then in ITS layout ([workgroups][lookup waves][8], tmac_chain.h), not the
    // [calls][workgroups][8] documented for k_decode_chain: a buffer sized for the latter would be overrun
    if (c->stream && dev_buffer) return fail(TMAC_HIP_E_NOMATCH, "stamps of a stream-mode chain exist in -DTMAC_STREAM_STAMPS builds only (layout: StreamArgs::stamps)");
#endif
    if (!c->stream && dev_buffer && !TMAC_CHAIN_STAMPS)
        return fail(TMAC_HIP_E_NOMATCH, "stamps of k_decode_chain exist in -DTMAC_CHAIN_STAMPS=1 builds of the library only (tools/build_variant.sh)");
    c->stamps = dev_buffer;
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_chain_threads(void) { return CHAIN_FT; }

extern "C" int32_t tmac_hip_chain_is_stream(const tmac_hip_chain* c) { return c && c->stream ? (c->qw ? 2 : 1) : 0; }

extern "C" int32_t tmac_hip_debug_chain_grid(int workgroups) {
    if (workgroups < 0) return fail(TMAC_HIP_E_ARG, "negative grid");
    g_knobs.chain_grid = workgroups;
    return TMAC_HIP_OK;
}

extern "C" int32_t tmac_hip_debug_chain_config(int force_wpq, unsigned spin_limit) {
    if (force_wpq < 0 || (force_wpq && CHAIN_NWV % force_wpq)) return fail(TMAC_HIP_E_ARG, "waves per quad must divide %d", CHAIN_NWV);
    g_knobs.chain_wpq = force_wpq;
    if (spin_limit) g_knobs.chain_spin_limit = spin_limit;
    return TMAC_HIP_OK;
}

// ---------------------------------------------------------------------------------------------
// Deferred launches (include/tmac_hip.h: tmac_hip_defer / tmac_hip_flush).  A caller that does NOT record -- a backend hook called mat-mul by
// mat-mul -- still issues, between two synchronisation points, calls that do not depend on each other (q / k / v of a layer as separate
// calls; the projections of several sequences).  Launched one by one they run k_gemv_quad (0.25 of the HBM peak on the headline shape, a
// launch each); queued and flushed together they are ONE stream-mode launch (k_lut_images + k_gemv_stream).  The queue holds N = 1 calls
// whose inputs are resident: a call that reads or overwrites anything a queued call writes (or overwrites what one reads) flushes the
// queue first, so a batch never carries a dependence and never needs a hand-off.  The recording built from a batch is cached by the
// batch's signature (matrices, pointers, dtypes): a decode loop pays tmac_hip_chain_end once per distinct batch.  What the persistent
// kernels do not cover (tmac_hip_chain_end returns -1) is launched call by call at the flush, as if it had never been queued.
// ---------------------------------------------------------------------------------------------
#include <atomic>
namespace {
struct DeferKey {
    std::vector<const tmac_hip_weights*> w;
    std::vector<void*> C;
    const void* B;
    int act, out;
    bool operator==(const DeferKey& o) const { return B == o.B && act == o.act && out == o.out && w == o.w && C == o.C; }
};
struct DeferEntry {
    std::vector<DeferKey> sig;
    std::vector<tmac_hip_chain*> chains;   // one stream-mode recording per configuration of the batch (bits, zero points, scale kind and dtype, output dtype)
    std::vector<uint32_t> singles;         // calls of the batch launched one by one (no persistent form, or alone in their configuration)
    unsigned long long used;
};
void defer_free_entry(DeferEntry& e, bool sync) {
    for (tmac_hip_chain* c : e.chains) {
        if (sync) (void)hipStreamSynchronize(c->last_stream);
        tmac_hip_chain_free(c);
    }
    e.chains.clear();
}
struct DeferState {
    bool on = false;
    std::vector<ChainRecOp> pending;
    hipStream_t stream = nullptr;
    std::vector<DeferEntry> cache;
    unsigned long long epoch = 0, tick = 0;
    unsigned long long n_flush = 0, n_hit = 0, n_stream = 0, n_chain = 0, n_single = 0;
    // (no destructor: a thread_local of the main thread is destroyed at process exit, when the HIP runtime may be gone -- the cached
    // recordings are released by tmac_hip_cache_clear / tmac_hip_reset_state on the owning thread, or with the process)
};
thread_local DeferState g_defer;
std::atomic<unsigned long long> g_defer_epoch{1};
constexpr size_t DEFER_MAX_BATCH = 256, DEFER_CACHE = 32, DEFER_MIN_STREAM = 3;

Range defer_in_range(const ChainRecOp& r) {
    return Range{(const char*)r.B, (const char*)r.B + (size_t)r.w[0]->s.K * (r.act == TMAC_F32 ? 4 : 2)};
}
Range defer_out_range(const ChainRecOp& r, size_t m) {
    return Range{(const char*)r.C[m], (const char*)r.C[m] + (size_t)r.w[m]->s.Mw * (r.out == TMAC_F16 ? 2 : 4)};
}

int32_t defer_flush(hipStream_t st) {
    DeferState& D = g_defer;
    if (D.pending.empty()) return TMAC_HIP_OK;
    std::vector<ChainRecOp> batch;
    batch.swap(D.pending);
    ++D.n_flush;
    const unsigned long long ep = g_defer_epoch.load(std::memory_order_acquire);
    if (ep != D.epoch) {                      // weights were freed since: every cached recording may point at dead matrices
        for (DeferEntry& e : D.cache) defer_free_entry(e, false);
        D.cache.clear();
        D.epoch = ep;
    }
    std::vector<DeferKey> sig(batch.size());
    for (size_t i = 0; i < batch.size(); ++i) { sig[i].w = batch[i].w; sig[i].C = batch[i].C; sig[i].B = batch[i].B; sig[i].act = (int)batch[i].act; sig[i].out = (int)batch[i].out; }
    DeferEntry* hit = nullptr;
    for (DeferEntry& e : D.cache) if (e.sig == sig) { hit = &e; break; }
    if (hit) ++D.n_hit;
    else {
        // The calls of a batch are independent of each other (defer_if_on), so they may be regrouped: one recording per configuration
        // a persistent kernel is instantiated for -- a caller that mixes 2- and 4-bit matrices (qgemm.py:98-116 allows any mix) gets one
        // stream launch per width instead of a launch per call.
        DeferEntry ne;
        ne.sig = sig; ne.used = 0;
        std::vector<char> taken(batch.size(), 0);
        for (size_t i = 0; i < batch.size(); ++i) {
            if (taken[i]) continue;
            const tmac_hip_weights* wi = batch[i].w[0];
            std::vector<uint32_t> grp;
            for (size_t j = i; j < batch.size(); ++j) {
                const tmac_hip_weights* wj = batch[j].w[0];
                if (taken[j] || wj->s.bits != wi->s.bits || wj->s.zero_point != wi->s.zero_point || (wj->s.m_groups >= 1) != (wi->s.m_groups >= 1) ||
                    wj->sc_dtype != wi->sc_dtype || batch[j].out != batch[i].out) continue;
                taken[j] = 1; grp.push_back((uint32_t)j);
            }
            tmac_hip_chain* c = nullptr;
            // (a stream launch costs ~10 us before its first byte -- k_lut_images + the persistent kernel's ramp -- against ~4 us of launch and
            // ramp per stand-alone call: two calls are faster one by one, three break even, four win by a third: profiles/r06_stream_small_batches.txt)
            if (grp.size() >= DEFER_MIN_STREAM && !g_chain_rec) {
                g_chain_rec = new std::vector<ChainRecOp>();
                for (uint32_t j : grp) g_chain_rec->push_back(batch[j]);
                g_chain_gat = new std::vector<ChainRecGather>();
                memset(&g_chain_xf, 0, sizeof(g_chain_xf));
                const int32_t rc = tmac_hip_chain_end(&c);          // (ends the recording whatever comes out)
                if (rc != TMAC_HIP_OK) c = nullptr;
                if (c && !c->stream) { tmac_hip_chain_free(c); c = nullptr; }    // (a batch carries no dependence: anything but a stream is not worth a persistent launch)
            }
            if (c) ne.chains.push_back(c);
            else ne.singles.insert(ne.singles.end(), grp.begin(), grp.end());
        }
        if (D.cache.size() >= DEFER_CACHE) {                    // evict the least recently used recording
            size_t v = 0;
            for (size_t i = 1; i < D.cache.size(); ++i) if (D.cache[i].used < D.cache[v].used) v = i;
            defer_free_entry(D.cache[v], true);
            D.cache.erase(D.cache.begin() + (long)v);
        }
        D.cache.push_back(ne);
        hit = &D.cache.back();
    }
    hit->used = ++D.tick;
    int32_t rc = TMAC_HIP_OK;
    for (tmac_hip_chain* c : hit->chains) {
        ++D.n_stream;
        if ((rc = tmac_hip_chain_launch(c, st)) != TMAC_HIP_OK) return rc;
    }
    const bool was_on = D.on;
    D.on = false;                                               // call by call, as if never queued
    for (uint32_t j : hit->singles) {
        const ChainRecOp& r = batch[j];
        ++D.n_single;
        rc = fused_impl(r.w.data(), (int)r.w.size(), r.B, r.act, r.C.data(), r.out, 1, nullptr, nullptr, st);
        if (rc != TMAC_HIP_OK) break;
    }
    D.on = was_on;
    return rc;
}
}  // namespace

bool tmac_host::defer_if_on(const tmac_hip_weights* const* wl, int nmat, const void* B_dev, tmac_dtype_t act_dtype, void* const* C_list,
                            tmac_dtype_t out_dtype, int N, hipStream_t st, int32_t* rc) {
    DeferState& D = g_defer;
    if (!D.on) return false;
    *rc = TMAC_HIP_OK;
    if (N != 1) { *rc = defer_flush(D.stream); return false; }               // (ordered behind the queue; launched as usual)
    ChainRecOp op;
    for (int i = 0; i < nmat; ++i) {
        if (!wl[i] || !C_list[i]) { *rc = fail(TMAC_HIP_E_ARG, "null matrix or output"); return true; }
        op.w.push_back(wl[i]); op.C.push_back(C_list[i]);
    }
    op.B = B_dev; op.act = act_dtype; op.out = out_dtype;
    memset(&op.xf, 0, sizeof(op.xf));
    // a dependence on the queue (RAW: reads a queued output; WAR / WAW: writes what a queued call reads or writes), another stream, or a
    // full queue: the queue goes first
    bool must_flush = !D.pending.empty() && (st != D.stream || D.pending.size() >= DEFER_MAX_BATCH);
    const Range in = defer_in_range(op);
    for (size_t j = 0; j < D.pending.size() && !must_flush; ++j) {
        const ChainRecOp& p = D.pending[j];
        const Range pin = defer_in_range(p);
        for (size_t m = 0; m < p.C.size() && !must_flush; ++m) {
            const Range po = defer_out_range(p, m);
            if (overlap(in, po)) must_flush = true;
            for (size_t k = 0; k < op.C.size(); ++k) if (overlap(defer_out_range(op, k), po)) must_flush = true;
        }
        for (size_t k = 0; k < op.C.size(); ++k) if (overlap(defer_out_range(op, k), pin)) must_flush = true;
    }
    if (must_flush && (*rc = defer_flush(D.stream)) != TMAC_HIP_OK) return true;
    D.stream = st;
    D.pending.push_back(op);
    return true;
}
void tmac_host::defer_forget_all() { g_defer_epoch.fetch_add(1, std::memory_order_acq_rel); }
void tmac_host::defer_release_thread() {          // the calling thread's cached recordings (nothing of them may be in flight: the caller has synchronised)
    DeferState& D = g_defer;
    for (DeferEntry& e : D.cache) defer_free_entry(e, false);
    D.cache.clear();
}

extern "C" int32_t tmac_hip_defer(int on) {
    DeferState& D = g_defer;
    if (!on && !D.pending.empty()) {
        const int32_t rc = defer_flush(D.stream);
        if (rc != TMAC_HIP_OK) return rc;
    }
    D.on = on != 0;
    return TMAC_HIP_OK;
}
extern "C" int32_t tmac_hip_flush(void* stream) {
    (void)stream;                                               // (the queue remembers the stream its calls were issued on)
    return defer_flush(g_defer.stream);
}
extern "C" int32_t tmac_hip_defer_stats(uint64_t* flushes, uint64_t* cache_hits, uint64_t* stream_launches, uint64_t* single_calls) {
    const DeferState& D = g_defer;
    if (flushes) *flushes = D.n_flush;
    if (cache_hits) *cache_hits = D.n_hit;
    if (stream_launches) *stream_launches = D.n_stream;
    if (single_calls) *single_calls = D.n_single;
    return TMAC_HIP_OK;
}
