// C-ABI entry points of libdiart_amd.so (see include/diart_amd.h) and the launch sequences
// of the two networks.  Everything here is host code driving the kernels in k_*.hip.
#include "dz_common.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <new>

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void dz_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* dz_last_error(void) { return g_err; }
extern "C" int dz_version(void) { return DZ_VERSION; }
extern "C" int dz_abi_struct_sizes(int out[5]) {
    if (!out) return 2;
    out[0] = (int)sizeof(dz_sincnet_weights);
    out[1] = (int)sizeof(dz_seg_weights);
    out[2] = (int)sizeof(dz_emb_weights);
    out[3] = (int)sizeof(dz_ecapa_weights);
    out[4] = (int)sizeof(dz_convgemm_desc);
    return 0;
}

// ---------------------------------------------------------------------------
// context + scratch arena
// ---------------------------------------------------------------------------
struct Arena {
    char* base = nullptr;
    size_t size = 0, used = 0;
    // first pass (base == nullptr) only measures
    float* take(size_t nfloats) {
        const size_t bytes = (nfloats * sizeof(float) + 255) & ~size_t(255);
        float* p = base ? reinterpret_cast<float*>(base + used) : nullptr;
        used += bytes;
        return p;
    }
};

extern "C" int dz_ctx_create(int hip_device, dz_ctx** out) {
    DZ_REQUIRE(out != nullptr, "dz_ctx_create: out is NULL");
    int n = 0;
    DZ_HIP(hipGetDeviceCount(&n));
    DZ_REQUIRE(hip_device >= 0 && hip_device < n, "dz_ctx_create: device %d of %d", hip_device, n);
    hipDeviceProp_t prop;
    DZ_HIP(hipGetDeviceProperties(&prop, hip_device));
    DZ_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
               "dz_ctx_create: device %d is %s, this library is built for gfx950 only", hip_device,
               prop.gcnArchName);
    dz_ctx* c = new (std::nothrow) dz_ctx;
    DZ_REQUIRE(c != nullptr, "dz_ctx_create: out of memory");
    c->device = hip_device;
    c->oflag_host = c->oflag_dev = nullptr;
    c->conv0_frag = c->convp_frag = nullptr;
    c->conv0_src = c->convp_src = nullptr;
    c->conv0_user = c->convp_user = nullptr;
    c->conv0_used = c->convp_used = false;
    c->convp_cin = c->convp_kpad = 0;
    DZ_HIP(hipSetDevice(hip_device));
    DZ_HIP(hipHostMalloc((void**)&c->oflag_host, sizeof(int), hipHostMallocMapped));
    *c->oflag_host = 0;
    DZ_HIP(hipHostGetDevicePointer((void**)&c->oflag_dev, c->oflag_host, 0));
    *out = c;
    return 0;
}
extern "C" int dz_ctx_destroy(dz_ctx* ctx) {
    if (ctx && ctx->oflag_host) (void)hipHostFree(ctx->oflag_host);
    if (ctx && ctx->conv0_frag) (void)hipFree(ctx->conv0_frag);
    if (ctx && ctx->convp_frag) (void)hipFree(ctx->convp_frag);
    delete ctx;
    return 0;
}
extern "C" int dz_range_check(dz_ctx* ctx, int reset) {
    DZ_REQUIRE(ctx != nullptr, "dz_range_check: NULL context");
    const int seen = *(volatile int*)ctx->oflag_host;
    if (reset) *(volatile int*)ctx->oflag_host = 0;
    if (seen) {
        dz_set_error("an operand of a split-f16 (\"f16x3\") kernel was outside +-65504 and has been clamped: "
                     "the result differs from an f32 reference; use precision=\"f32\" for such inputs");
        return 6;
    }
#ifdef DZ_EXPERIMENTS
    DZ_HIP(hipSetDevice(ctx->device));
    if (dz_g3_error(reset)) {
        dz_set_error("k_gemm_g3.hip: a workgroup gave up waiting for the partial sums of a neighbour (the results of "
                     "that launch are wrong); DZ_GEMM_GEN=1 selects the non-persistent kernel");
        return 7;
    }
#endif
    return 0;
}

// ---------------------------------------------------------------------------
// per-kernel timing (bench.py's roofline leg).  Off by default; when on, every launch issued by
// the forward passes carries a (start, stop) event pair from a fixed pool that the runtime fills
// with the dispatch's own timestamps (DZ_LAUNCH / hipExtLaunchKernelGGL): kernel execution time on
// whatever stream it ran, no marker packets between kernels.  dz_prof_collect() synchronises and
// accumulates.
// ---------------------------------------------------------------------------
enum { PROF_POOL = 8192, PROF_TAGS = 32 };
static const char* kProfNames[PROF_TAGS] = {
    "wave_stats", "sinc_conv0", "finalize_norm", "conv1_pool", "conv2_pool", "lstm_proj",
    "lstm_rec", "seg_mlp", "seg_classifier", "tdnn1", "tdnn2", "tdnn3", "tdnn4", "tdnn5",
    "stats_pool", "emb_linear", "l2norm", "osp", "powerset", "cdist", "lstm_proj0",
    // config 3 (ecapa_api.hip; DZ_T_ECAPA_* in dz_common.h)
    "ecapa_fbank", "ecapa_block0", "ecapa_wide1x1", "ecapa_res2net", "ecapa_se", "ecapa_asp", "ecapa_fc",
    "sinc_conv0_pair", "norm_split", "", ""};
enum { T_WAVE = 0, T_CONV0, T_FIN, T_CONV1, T_CONV2, T_PROJ, T_REC, T_MLP, T_CLS, T_TDNN1, T_TDNN2,
       T_TDNN3, T_TDNN4, T_TDNN5, T_POOL, T_EMBLIN, T_L2, T_OSP, T_PSET, T_CDIST, T_PROJ0, T_CONV0_PAIR = 28, T_NSPLIT = 29 };
thread_local DzLaunchProf* dz_launch_prof = nullptr;
thread_local int* dz_cur_oflag = nullptr;
struct Prof {
    bool on = false;
    int used = 0;
    DzLaunchProf ev[PROF_POOL];
    int tag[PROF_POOL];
    int units[PROF_POOL];        // chunks the bracketed launch works on
    bool made = false;
    double ms[PROF_TAGS];
    long long n[PROF_TAGS];
    long long chunks[PROF_TAGS];
};
// One profiler per process, shared by every handle and host thread: slots are handed out under a
// lock (the forward passes of different handles may be driven from different threads); the
// "consumed by the next DZ_LAUNCH" hand-off itself is thread local.
static Prof g_prof;
static std::mutex g_prof_mu;
// `chunks`: how many 5 s chunks (ECAPA: embedding rows) this launch processes — the unit bench.py's roofline counts in
DzProfScope::DzProfScope(int tag, int chunks) {
    if (!g_prof.on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof.on && g_prof.used < PROF_POOL && tag >= 0 && tag < PROF_TAGS) {
        const int slot = g_prof.used++;
        g_prof.tag[slot] = tag;
        g_prof.units[slot] = chunks;
        dz_launch_prof = &g_prof.ev[slot];   // consumed by the next DZ_LAUNCH
    }
}
DzProfScope::~DzProfScope() { dz_launch_prof = nullptr; }
typedef DzProfScope ProfScope;
extern "C" int dz_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (on && !g_prof.made) {
        for (int i = 0; i < PROF_POOL; ++i) {
            DZ_HIP(hipEventCreate(&g_prof.ev[i].start));
            DZ_HIP(hipEventCreate(&g_prof.ev[i].stop));
        }
        g_prof.made = true;
    }
    g_prof.on = on != 0;
    g_prof.used = 0;
    for (int t = 0; t < PROF_TAGS; ++t) { g_prof.ms[t] = 0.0; g_prof.n[t] = 0; g_prof.chunks[t] = 0; }
    return 0;
}
// suspend / resume the bracketing without touching what has been accumulated: bench.py instruments
// every k-th step of its timed region only (a dispatch that carries profiling events costs the
// runtime ~15 % of throughput when every launch has one)
extern "C" int dz_prof_pause(int paused) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof.made) g_prof.on = !paused;
    return 0;
}
// drains the event pool (device must be idle or will be synchronised); returns #tags
extern "C" int dz_prof_collect(void) {
    DZ_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(g_prof_mu);
    // DZ_PROF_TIMELINE=<file>: additionally append "tag chunks start_us duration_us" of every bracketed launch
    // (start relative to the first one of this drain; the dispatches' own timestamps, whatever stream they
    // ran on) — the step's schedule without a tracer slowing the host down (tools/timeline.py)
    const char* tl_path = getenv("DZ_PROF_TIMELINE");
    FILE* tl = tl_path && tl_path[0] && g_prof.used > 0 ? fopen(tl_path, "a") : nullptr;
    if (tl) fprintf(tl, "# drain of %d launches\n", g_prof.used);
    for (int i = 0; i < g_prof.used; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_prof.ev[i].start, g_prof.ev[i].stop) == hipSuccess) {
            g_prof.ms[g_prof.tag[i]] += ms;
            g_prof.n[g_prof.tag[i]] += 1;
            g_prof.chunks[g_prof.tag[i]] += g_prof.units[i];
            float t0 = 0.f;
            if (tl && hipEventElapsedTime(&t0, g_prof.ev[0].start, g_prof.ev[i].start) == hipSuccess)
                fprintf(tl, "%s %d %.1f %.1f\n", kProfNames[g_prof.tag[i]], g_prof.units[i], t0 * 1e3, ms * 1e3);
        }
    }
    if (tl) fclose(tl);
    g_prof.used = 0;
    // an event pair whose bracket never launched (a ProfScope around a path that returned early) fails
    // in hipEventElapsedTime: it is skipped above, and the runtime's sticky "last error" must not be left
    // for the next caller's error check (torch raises on it)
    (void)hipGetLastError();
    return PROF_TAGS;
}
extern "C" int dz_prof_get(int tag, const char** name, double* total_ms, long long* launches,
                           long long* chunks) {
    if (tag < 0 || tag >= PROF_TAGS) return 2;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (name) *name = kProfNames[tag];
    if (total_ms) *total_ms = g_prof.ms[tag];
    if (launches) *launches = g_prof.n[tag];
    if (chunks) *chunks = g_prof.chunks[tag];
    return 0;
}

// ---------------------------------------------------------------------------
// geometry of the SincNet front-end for S samples
// ---------------------------------------------------------------------------
struct SincGeom {
    int S, F0, P0, T1, P1, T2, P2;  // conv frames / pooled frames per stage
    int nt0, nt1, nt2;              // tiles carrying instance-norm partials
    bool ok;
};
static SincGeom sinc_geom(int S, bool conv0_split = false) {
    SincGeom g;
    memset(&g, 0, sizeof(g));
    g.S = S;
    if (S < 251) return g;
    g.F0 = (S - 251) / 10 + 1;
    g.P0 = g.F0 / 3;
    g.T1 = g.P0 - 4;
    g.P1 = g.T1 > 0 ? g.T1 / 3 : 0;
    g.T2 = g.P1 - 4;
    g.P2 = g.T2 > 0 ? g.T2 / 3 : 0;
    g.nt0 = conv0_split ? dz_conv0_split_ntile(g.F0) : (g.F0 + 191) / 192;
    g.nt1 = g.T1 > 0 ? dz_convgemm_ntile(g.T1) : 0;
    g.nt2 = g.T2 > 0 ? dz_convgemm_ntile(g.T2) : 0;
    g.ok = g.P2 > 0;
    return g;
}
extern "C" int dz_seg_frames_for(int num_samples) { return sinc_geom(num_samples).P2; }
extern "C" int dz_emb_frames_for(int num_samples) {
    const int f = sinc_geom(num_samples).P2 - 4 - 4 - 6;
    return f > 0 ? f : 0;
}

// ---- run-time options (dz_common.h) ----------------------------------------------------------------
static int g_options[DZ_OPT_COUNT] = {1, 1, 0};
static const char* const kOptionNames[DZ_OPT_COUNT] = {"f32_gemm", "pool_fuse", "pack_cache"};
int dz_option(int id) { return id >= 0 && id < DZ_OPT_COUNT ? __atomic_load_n(&g_options[id], __ATOMIC_RELAXED) : 0; }
static int option_index(const char* name) {
    for (int i = 0; name && i < DZ_OPT_COUNT; ++i)
        if (strcmp(name, kOptionNames[i]) == 0) return i;
    return -1;
}
extern "C" int dz_set_option(const char* name, int value) {
    const int i = option_index(name);
    DZ_REQUIRE(i >= 0, "dz_set_option: unknown option '%s' (f32_gemm, pool_fuse, pack_cache)", name ? name : "(null)");
    __atomic_store_n(&g_options[i], value, __ATOMIC_RELAXED);
    return 0;
}
extern "C" int dz_get_option(const char* name, int* value) {
    const int i = option_index(name);
    DZ_REQUIRE(i >= 0 && value, "dz_get_option: unknown option '%s' (f32_gemm, pool_fuse, pack_cache)", name ? name : "(null)");
    *value = dz_option(i);
    return 0;
}
extern "C" int dz_has_experiments(void) {
#ifdef DZ_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

// DZ_FUSED_NORM=0 keeps the three finalize_norm launches of a SincNet; by default (split-f16 path)
// every consumer derives its InstanceNorm scale / shift from the producer's tile partials itself
static bool fused_norm_enabled() {
    const char* v = dz_exp_env("DZ_FUSED_NORM");
    return !(v && v[0] == '0');
}
// DZ_POOL_FUSE=0: tdnn5 writes its f32 output and stats_pool reads it back (the round-2 path)
static bool pool_fuse_enabled() {
    return dz_option(DZ_OPT_POOL_FUSE) != 0;
}
static bool conv_pool_enabled() {
    const char* v = dz_exp_env("DZ_CONV_POOL");
    return !(v && v[0] == '0');
}

// exact-f32 MFMA kernel, or the split-f16 kernel when the layer came with split planes
static int run_gemm(DzConvGemm& p, const void* split, hipStream_t st, const void* wfrag = nullptr) {
    if (split) {
        p.Wsplit = split;
        if (p.epi == DZ_EPI_POOL3 && conv_pool_enabled()) return dz_launch_conv_pool(p, st, wfrag);
        return dz_launch_gemm_split(p, st);
    }
    return dz_launch_convgemm(p, st);
}

struct SincScratch {
    float *stats, *y0, *part0, *sc0, *sh0, *y1, *part1, *sc1, *sh1, *y2, *part2, *sc2, *sh2;
    float* y2s;      // y2 normalised + split: the f16 planes [2][Bm * P2 rows][64] (kb-major) the first layer of each network reads
    void *bank_frag, *w1_frag, *w2_frag;     // the sinc bank / conv1 / conv2 weights in their kernels' fragment order (filled once, at create)
    void carve(Arena& a, const SincGeom& g, int Bm) {
        bank_frag = a.take((size_t)dz_sinc_bank_frag_bytes() / 4);
        w1_frag = a.take((size_t)dz_conv_pool_wfrag_bytes(80) / 4);
        w2_frag = a.take((size_t)dz_conv_pool_wfrag_bytes(64) / 4);
        stats = a.take((size_t)Bm * 2 * DZ_WS_G);   // slice moments of the waveform
        y0 = a.take((size_t)Bm * g.P0 * 80);
        part0 = a.take((size_t)Bm * g.nt0 * 80 * 2);
        sc0 = a.take((size_t)Bm * 80);
        sh0 = a.take((size_t)Bm * 80);
        y1 = a.take((size_t)Bm * g.P1 * 64);
        part1 = a.take((size_t)Bm * g.nt1 * 64 * 2);
        sc1 = a.take((size_t)Bm * 64);
        sh1 = a.take((size_t)Bm * 64);
        y2 = a.take((size_t)Bm * g.P2 * 64);
        part2 = a.take((size_t)Bm * g.nt2 * 64 * 2);
        sc2 = a.take((size_t)Bm * 64);
        sh2 = a.take((size_t)Bm * 64);
        y2s = a.take((size_t)Bm * g.P2 * 64);
    }
};

// the register-resident operands of the SincNet kernels in fragment order: once per handle (the weights are final then)
static int sinc_repack(const dz_sincnet_weights& w, const SincScratch& s) {
    int rc = 0;
    if (w.filt_split) rc = dz_launch_sinc_bank_frag(w.filt_split, s.bank_frag, nullptr);
    if (!rc && w.w1_split) rc = dz_launch_conv_pool_wfrag(80, w.w1_split, 416, s.w1_frag, nullptr);
    if (!rc && w.w2_split) rc = dz_launch_conv_pool_wfrag(64, w.w2_split, 320, s.w2_frag, nullptr);
    if (!rc && hipStreamSynchronize(nullptr) != hipSuccess) {
        dz_set_error("sinc_repack: hipStreamSynchronize failed");
        rc = 1;
    }
    return rc;
}

static bool sinc_fused_norm(const dz_sincnet_weights& w) {
    return w.w1_split && w.w2_split && conv_pool_enabled() && fused_norm_enabled();
}
// the consumer of y2 (first LSTM projection / tdnn1) reads its norm from part2 when fused
static void sinc_out_norm(DzConvGemm& p, const dz_sincnet_weights& w, const SincGeom& g, const SincScratch& s,
                          bool split_layer) {
    p.nld = 64;
    p.norm_on_load = 1;
    if (split_layer && sinc_fused_norm(w)) {
        p.npart = s.part2; p.npart_tiles = g.nt2; p.npart_T = g.P2; p.ngamma = w.in2_g; p.nbeta = w.in2_b;
    } else {
        p.nscale = s.sc2; p.nshift = s.sh2;
    }
}

// Round 6: y2 is normalised and split ONCE (norm_split_kernel) and its two consumers — the first LSTM projection, tdnn1
// — run on the pre-split GEMM: whenever the weights came with kb-major planes for that layer and the SincNet's norms
// are the fused ones (tile partials).  -> planes in s.y2s, plane = B * P2 * 64 elements.
static bool sinc_pre_split_ok(const dz_sincnet_weights& w, const void* first_layer_kb) {
    return first_layer_kb != nullptr && sinc_fused_norm(w);
}
static int sinc_norm_split(const dz_sincnet_weights& w, const SincGeom& g, const SincScratch& s, int B, hipStream_t st) {
    ProfScope ps(T_NSPLIT, B);
    return dz_launch_norm_split(s.y2, s.part2, g.nt2, g.P2, w.in2_g, w.in2_b, s.y2s, (long long)B * g.P2 * 64, B, st);
}

// Exact-f32 weights (no split planes anywhere in the SincNet) with the f32 MFMA GEMM on: y2 is normalised ONCE into
// s.y2s as f32 rows (norm_f32_kernel reads the tile partials itself: no third finalize_norm launch) and its two
// consumers — the first LSTM projection, tdnn1 — run flattened on k_gemm_f32.hip.  The experiments build's
// non-fused f16 configurations (DZ_FUSED_NORM=0 / DZ_CONV_POOL=0) keep finalize_norm + norm-on-load.
static bool sinc_f32_norm_pass(const dz_sincnet_weights& w) {
    return !w.filt_split && !w.w1_split && !w.w2_split && dz_option(DZ_OPT_F32_GEMM) != 0;
}

// wave -> y2 [B][P2][64] (pre-norm) + part2 (or sc2/sh2): the consumer applies InstanceNorm +
// LeakyReLU on load.  4 launches (7 with the finalize_norm launches of the exact-f32 path).
// ext_stats: slice moments of these B windows somebody already computed (dz_wave_stats: the
// segmentation and the embedding network normalise the SAME waveform, InstanceNorm1d(1) statistics
// do not depend on the network) — NULL: compute them here.
static int run_sincnet(const dz_sincnet_weights& w, const SincGeom& g, const SincScratch& s,
                       const float* wave, long long stride, int B, hipStream_t st,
                       const float* ext_stats = nullptr, bool ext_conv0 = false) {
    int rc;
    const float* stats = ext_stats ? ext_stats : s.stats;
    // ext_conv0: y0 / part0 of these B windows are already (being) written on this stream's dependencies by
    // dz_sinc_conv0_pair — the first stage of both networks in one launch
    if (!ext_stats && !ext_conv0)
    { ProfScope ps(T_WAVE, B); if ((rc = dz_launch_wave_stats(wave, stride, B, g.S, s.stats, st))) return rc; }
    if (!ext_conv0)
    { ProfScope ps(T_CONV0, B);
    rc = w.filt_split
             ? dz_launch_sinc_conv0_split(wave, stride, B, g.S, stats, 1, w.wav_gamma, w.wav_beta,
                                          w.filt_split, s.y0, g.P0, s.part0, g.nt0, st, s.bank_frag)
             : dz_launch_sinc_conv0(wave, stride, B, g.S, stats, 1, w.wav_gamma, w.wav_beta, w.filt,
                                    s.y0, g.P0, s.part0, g.nt0, st);
    if (rc) return rc; }
    const bool fused = sinc_fused_norm(w);
    if (!fused)
    { ProfScope ps(T_FIN, B);
    if ((rc = dz_launch_finalize_norm(s.part0, B, g.nt0, 80, g.P0, w.in0_g, w.in0_b, s.sc0, s.sh0,
                                      st)))
        return rc; }
    DzConvGemm p;
    memset(&p, 0, sizeof(p));
    // conv1: 80 -> 60(64), k5, + pool3
    p.X = s.y0; p.W = w.w1; p.bias = w.b1; p.nld = 80;
    if (fused) { p.npart = s.part0; p.npart_tiles = g.nt0; p.npart_T = g.P0; p.ngamma = w.in0_g; p.nbeta = w.in0_b; }
    else { p.nscale = s.sc0; p.nshift = s.sh0; }
    p.Y = s.y1; p.partials = s.part1;
    p.B = B; p.Tin = g.P0; p.Tout = g.T1; p.Cin = 80; p.taps = 5; p.dil = 1; p.K = 400;
    p.Kpad = 416; p.Npad = 64; p.Nstore = 64; p.ldx = 80; p.ldy = 64; p.Tstore = g.P1;
    p.xbs = (long long)g.P0 * 80; p.ybs = (long long)g.P1 * 64;
    p.norm_on_load = 1; p.epi = DZ_EPI_POOL3;
    { ProfScope ps(T_CONV1, B); if ((rc = run_gemm(p, w.w1_split, st, s.w1_frag))) return rc; }
    if (!fused)
    { ProfScope ps(T_FIN, B);
    if ((rc = dz_launch_finalize_norm(s.part1, B, g.nt1, 64, g.P1, w.in1_g, w.in1_b, s.sc1, s.sh1,
                                      st)))
        return rc; }
    // conv2: 60(64) -> 60(64), k5, + pool3
    p.X = s.y1; p.W = w.w2; p.bias = w.b2; p.nld = 64;
    if (fused) { p.npart = s.part1; p.npart_tiles = g.nt1; p.npart_T = g.P1; p.ngamma = w.in1_g; p.nbeta = w.in1_b; }
    else { p.nscale = s.sc1; p.nshift = s.sh1; }
    p.Y = s.y2; p.partials = s.part2;
    p.Tin = g.P1; p.Tout = g.T2; p.Cin = 64; p.K = 320; p.Kpad = 320; p.ldx = 64;
    p.Tstore = g.P2; p.xbs = (long long)g.P1 * 64; p.ybs = (long long)g.P2 * 64;
    { ProfScope ps(T_CONV2, B); if ((rc = run_gemm(p, w.w2_split, st, s.w2_frag))) return rc; }
    if (fused) return 0;
    if (sinc_f32_norm_pass(w)) {
        ProfScope ps(T_NSPLIT, B);
        return dz_launch_norm_f32(s.y2, s.part2, g.nt2, g.P2, w.in2_g, w.in2_b, s.y2s, B, st);
    }
    ProfScope ps(T_FIN, B);
    return dz_launch_finalize_norm(s.part2, B, g.nt2, 64, g.P2, w.in2_g, w.in2_b, s.sc2, s.sh2, st);
}

static int check_wave(const char* who, const float* d_wave, long long stride, int S) {
    DZ_REQUIRE(d_wave != nullptr, "%s: d_wave is NULL", who);
    DZ_REQUIRE(((uintptr_t)d_wave & 15) == 0 && (stride & 3) == 0,
               "%s: waveform rows must be 16-byte aligned (ptr %p, stride %lld)", who,
               (const void*)d_wave, stride);
    DZ_REQUIRE(stride >= 0, "%s: negative stride", who);
    (void)S;
    return 0;
}

// ---------------------------------------------------------------------------
// segmentation
// ---------------------------------------------------------------------------
struct dz_seg {
    dz_ctx* ctx;
    dz_seg_weights w;
    SincGeom g;
    int Bm;
    bool pre;    // wide layers on k_gemm_pre.hip (activations travel as f16 hi/lo planes)
    const float* ext_stats;   // dz_seg_use_wave_stats: consumed (and cleared) by the next forward
    const float* cur_stats;   // the slice moments the front half of the current forward normalised with (NaN rows, dz_ws_bad)
    int ext_conv0_B;          // dz_sinc_conv0_pair wrote y0 / part0 of this many chunks: consumed by the next forward
    char* arena;
    SincScratch ss;
    float *gx, *gx0, *h0, *h1, *m0, *m1, *logit;
    int front_B;              // dz_seg_front ran for this many chunks and dz_seg_back has not consumed it yet
    hipEvent_t ev_gx0_free;   // recorded behind the layer-0 recurrence: the next dz_seg_front may overwrite gx0
};

static void seg_carve(dz_seg* s, Arena& a) {
    const size_t rows = (size_t)s->Bm * s->g.P2;
    s->ss.carve(a, s->g, s->Bm);
    s->gx = a.take(rows * 1024);
    s->gx0 = a.take(rows * 1024);      // layer 0's x-projection: written by the front half, one step ahead
    s->h0 = a.take(rows * 256);
    s->h1 = a.take(rows * 256);
    s->m0 = a.take(rows * 128);
    s->m1 = a.take(rows * 128);
    s->logit = a.take(rows * 8);
}

extern "C" int dz_seg_create(dz_ctx* ctx, const dz_seg_weights* w, int max_batch, int num_samples,
                             dz_seg** out) {
    DZ_REQUIRE(ctx && w && out, "dz_seg_create: NULL argument");
    DZ_REQUIRE(max_batch >= 1, "dz_seg_create: max_batch %d", max_batch);
    const SincGeom g = sinc_geom(num_samples, w->sinc.filt_split != nullptr);
    DZ_REQUIRE(g.ok, "dz_seg_create: %d samples is too short for SincNet", num_samples);
    DZ_REQUIRE(w->num_classes >= 1 && w->num_classes <= 8, "dz_seg_create: num_classes %d",
               w->num_classes);
    if (w->powerset)
        DZ_REQUIRE(w->num_classes == 1 + w->num_speakers + w->num_speakers * (w->num_speakers - 1) / 2,
                   "dz_seg_create: powerset with %d classes / %d speakers", w->num_classes,
                   w->num_speakers);
    // run_sincnet leaves the last InstanceNorm to the consumer's prologue whenever conv1 / conv2 came
    // with split planes; a first projection WITHOUT planes would then read sc2 / sh2 nobody wrote
    DZ_REQUIRE((w->sinc.w1_split != nullptr) == (w->sinc.w2_split != nullptr) &&
                   (w->sinc.w1_split != nullptr) == (w->wih_split[0] != nullptr),
               "dz_seg_create: the split planes of SincNet conv1 / conv2 and of the first LSTM projection "
               "must be all present or all absent");
    DZ_HIP(hipSetDevice(ctx->device));
    dz_seg* s = new (std::nothrow) dz_seg;
    DZ_REQUIRE(s != nullptr, "dz_seg_create: out of memory");
    s->ctx = ctx; s->w = *w; s->g = g; s->Bm = max_batch; s->arena = nullptr; s->ext_stats = nullptr; s->cur_stats = nullptr;
    s->ext_conv0_B = 0;
    s->front_B = 0; s->ev_gx0_free = nullptr;
    DZ_HIP(hipEventCreateWithFlags(&s->ev_gx0_free, hipEventDisableTiming));
    s->pre = w->wih_split[1] && w->wih_split[2] && w->wih_split[3] &&
             w->lin0_split && w->lin1_split;
    Arena measure;
    seg_carve(s, measure);
    hipError_t e = hipMalloc((void**)&s->arena, measure.used);
    if (e != hipSuccess) {
        dz_set_error("dz_seg_create: hipMalloc(%zu) failed: %s", measure.used, hipGetErrorString(e));
        (void)hipEventDestroy(s->ev_gx0_free);
        delete s;
        return 1;
    }
    DZ_HIP(hipMemset(s->arena, 0, measure.used));
    Arena real;
    real.base = s->arena; real.size = measure.used;
    seg_carve(s, real);
    if (int rc = sinc_repack(w->sinc, s->ss)) {
        (void)hipFree(s->arena);
        (void)hipEventDestroy(s->ev_gx0_free);       // (the event leaked on this path)
        delete s;
        return rc;
    }
    *out = s;
    return 0;
}

extern "C" int dz_seg_destroy(dz_seg* seg) {
    if (seg) {
        if (seg->arena) (void)hipFree(seg->arena);
        if (seg->ev_gx0_free) (void)hipEventDestroy(seg->ev_gx0_free);
        delete seg;
    }
    return 0;
}

// phase 0: the whole network on one stream; 1: front half (SincNet + the first x-projection into gx0);
// 2: back half (4 recurrences, projections 1..3, MLP head) of the chunks the last front half left
static int seg_forward(dz_seg* s, const float* d_wave, long long wave_stride, int B, float* d_out,
                       float* d_osp, float gamma, float beta, int normalize, void* stream, int phase = 0);
static bool mlp_head_enabled() {
    static const bool on = [] {
        const char* e = dz_exp_env("DZ_MLP_HEAD");
        return !(e && e[0] == '0');
    }();
    return on;
}
extern "C" int dz_seg_forward(dz_seg* s, const float* d_wave, long long wave_stride, int B,
                              float* d_out, void* stream) {
    return seg_forward(s, d_wave, wave_stride, B, d_out, nullptr, 0.f, 0.f, 0, stream);
}
extern "C" int dz_seg_forward_osp(dz_seg* s, const float* d_wave, long long wave_stride, int B,
                                  float* d_out, float gamma, float beta, int normalize,
                                  float* d_weights, void* stream) {
    DZ_REQUIRE(d_weights != nullptr, "dz_seg_forward_osp: d_weights is NULL");
    return seg_forward(s, d_wave, wave_stride, B, d_out, d_weights, gamma, beta, normalize, stream);
}
// The two halves of dz_seg_forward_osp for a caller that keeps the stateless front end of the NEXT step
// off the long dependent chain of this one (StreamBatch): dz_seg_front(t + 2) — SincNet and the first
// x-projection, on a stream of its own — runs under the recurrences of dz_seg_back(t) on the same handle.
// The only buffer both halves touch is gx0; the front half waits (on the GPU) for the event the back half
// records behind the layer-0 recurrence that reads it.
extern "C" int dz_seg_front(dz_seg* s, const float* d_wave, long long wave_stride, int B, void* stream) {
    DZ_REQUIRE(s != nullptr, "dz_seg_front: NULL handle");
    return seg_forward(s, d_wave, wave_stride, B, nullptr, nullptr, 0.f, 0.f, 0, stream, 1);
}
extern "C" int dz_seg_back(dz_seg* s, int B, float* d_out, float gamma, float beta, int normalize,
                           float* d_weights, void* stream) {
    DZ_REQUIRE(s != nullptr, "dz_seg_back: NULL handle");
    DZ_REQUIRE(B == s->front_B, "dz_seg_back: %d chunks, but dz_seg_front prepared %d", B, s->front_B);
    return seg_forward(s, nullptr, 0, B, d_out, d_weights, gamma, beta, normalize, stream, 2);
}
static int seg_forward(dz_seg* s, const float* d_wave, long long wave_stride, int B, float* d_out,
                       float* d_osp, float gamma, float beta, int normalize, void* stream, int phase) {
    DZ_REQUIRE(s && (d_out || phase == 1), "dz_seg_forward: NULL argument");
    DZ_REQUIRE(B >= 1 && B <= s->Bm, "dz_seg_forward: batch %d outside [1, %d]", B, s->Bm);
    int rc;
    if (phase != 2 && (rc = check_wave("dz_seg_forward", d_wave, wave_stride, s->g.S))) return rc;
    DZ_HIP(hipSetDevice(s->ctx->device));
    DzRangeScope range_scope(s->ctx->oflag_dev);
    hipStream_t st = (hipStream_t)stream;
    const int F = s->g.P2;
    if (phase != 2) {
        if (phase == 1) DZ_HIP(hipStreamWaitEvent(st, s->ev_gx0_free, 0));   // (never recorded yet: no-op)
        const float* ext = s->ext_stats;
        s->ext_stats = nullptr;
        const int pair_B = s->ext_conv0_B;
        s->ext_conv0_B = 0;
        DZ_REQUIRE(pair_B == 0 || pair_B == B, "dz_seg_forward: dz_sinc_conv0_pair ran for %d chunks, this call has %d",
                   pair_B, B);
        if ((rc = run_sincnet(s->w.sinc, s->g, s->ss, d_wave, wave_stride, B, st, ext, pair_B > 0))) return rc;
        // (front half alone with the handle's OWN moments: the next dz_seg_front may overwrite them before this step's
        // back half reads them — the NaN rows then come from caller-owned moments only, dz_seg_use_wave_stats)
        s->cur_stats = ext ? ext : (phase == 0 ? s->ss.stats : nullptr);
    }

    // 4 x { x-projection of both directions as one GEMM (N = 1024); persistent recurrence }.
    // With s->pre the hidden states travel as f16 (hi, lo) planes (same bytes as f32, same buffers)
    // and the projections of layers 1..3 and the MLP run on k_gemm_pre.hip.
    const long long rows = (long long)B * F;
    const float* lin = nullptr;
    for (int layer = 0; layer < 4; ++layer) {
        DzConvGemm p;
        memset(&p, 0, sizeof(p));
        float* const gxl = layer == 0 ? s->gx0 : s->gx;
        p.W = s->w.wih[layer]; p.bias = s->w.bih[layer]; p.Y = gxl;
        p.taps = 1; p.dil = 1; p.Npad = 1024; p.Nstore = 1024; p.ldy = 1024; p.epi = DZ_EPI_BIAS;
        const bool do_proj = layer == 0 ? phase != 2 : phase != 1;
        const bool do_rec = phase != 1;
        if (layer == 0) {
            p.X = s->ss.y2;
            sinc_out_norm(p, s->w.sinc, s->g, s->ss, s->w.wih_split[0] != nullptr);
            p.B = B; p.Tin = p.Tout = p.Tstore = F; p.Cin = 64; p.K = 64; p.Kpad = 64; p.ldx = 64;
            p.xbs = (long long)F * 64; p.ybs = (long long)F * 1024;
        } else {
            p.X = lin;
            p.B = 1; p.Tin = p.Tout = p.Tstore = B * F; p.Cin = 256; p.K = 256; p.Kpad = 256;
            p.ldx = 256;
        }
        const bool pre0 = layer == 0 && s->pre && sinc_pre_split_ok(s->w.sinc, s->w.wih0_split_kb);
        if (layer == 0 && !pre0 && !s->w.wih_split[0] && sinc_f32_norm_pass(s->w.sinc)) {
            // exact f32: y2 was normalised into y2s by run_sincnet; one flattened GEMM without a prologue (k_gemm_f32.hip)
            p.X = s->ss.y2s; p.norm_on_load = 0; p.npart = nullptr; p.nscale = p.nshift = nullptr;
            p.B = 1; p.Tin = p.Tout = p.Tstore = B * F; p.xbs = p.ybs = 0;
        }
        if (do_proj && pre0) {       // y2 -> normalised planes, then the projection as one flattened pre-split GEMM
            if ((rc = sinc_norm_split(s->w.sinc, s->g, s->ss, B, st))) return rc;
            p.X = nullptr; p.norm_on_load = 0; p.npart = nullptr; p.nscale = p.nshift = nullptr;
            p.B = 1; p.Tin = p.Tout = p.Tstore = B * F; p.xbs = p.ybs = 0;
        }
        if (do_proj)
        { ProfScope ps(layer == 0 ? T_PROJ0 : T_PROJ, B);
          if (pre0) {
              p.Xsplit = s->ss.y2s; p.xplane = rows * 64; p.Wsplit = s->w.wih0_split_kb;
              rc = dz_launch_gemm_pre(p, st);
          } else if (layer > 0 && s->pre) {
              p.X = nullptr; p.Xsplit = lin; p.xplane = rows * 256; p.Wsplit = s->w.wih_split[layer];
              rc = dz_launch_gemm_pre(p, st);
          } else {
              rc = run_gemm(p, s->w.wih_split[layer], st);
          }
          if (rc) return rc; }
        if (!do_rec) {               // front half: SincNet + the first projection are enqueued, that is all
            s->front_B = B;
            return 0;
        }
        float* hout = (layer & 1) ? s->h1 : s->h0;
        { ProfScope ps(T_REC, B);
          // gx columns are unit-major (weights.py permutes the rows of W_ih); 16 chains per
          // workgroup on the matrix cores when the layer came with split planes of W_hh
          float* hf = s->pre ? nullptr : hout;
          void* hs = s->pre ? (void*)hout : nullptr;
          rc = s->w.whh_split[layer]
                   ? dz_launch_lstm_mfma(gxl, s->w.whh_split[layer], hf, hs, rows * 256, B, F, 1,
                                         s->w.lstm_variant, st)
                   : dz_launch_lstm(gxl, s->w.whh[layer], hf, hs, rows * 256, B, F, 1, st);
          if (rc) return rc; }
        if (layer == 0) {            // gx0 has been read: the next front half may overwrite it
            DZ_HIP(hipEventRecord(s->ev_gx0_free, st));
            s->front_B = 0;
        }
        lin = hout;
    }
    // Linear(256,128)+leaky, Linear(128,128)+leaky, classifier
    DzConvGemm p;
    memset(&p, 0, sizeof(p));
    p.B = 1; p.Tin = p.Tout = p.Tstore = B * F; p.taps = 1; p.dil = 1;
    p.W = s->w.lin0_w; p.bias = s->w.lin0_b;
    p.Cin = 256; p.K = 256; p.Kpad = 256; p.ldx = 256; p.Npad = 128; p.Nstore = 128; p.ldy = 128;
    p.epi = DZ_EPI_BIAS_LEAKY;
    // default precision, no min-max normalisation of the OSP weights (it needs whole chunks): MLP +
    // classifier + activation + OSP in ONE launch (k_mlp_head.hip); DZ_MLP_HEAD=0: three launches
    if (s->pre && mlp_head_enabled() && !(d_osp && normalize) && s->w.num_classes <= 8) {
        DzMlpHead m{};
        m.Xsplit = lin; m.xplane = rows * 256;
        m.W0split = s->w.lin0_split; m.W1split = s->w.lin1_split;
        m.b0 = s->w.lin0_b; m.b1 = s->w.lin1_b; m.cw = s->w.cls_w; m.cb = s->w.cls_b;
        m.rows = B * F; m.F = F; m.classes = s->w.num_classes; m.K = s->w.num_speakers; m.powerset = s->w.powerset;
        m.gamma = gamma; m.beta = beta; m.seg = d_out; m.wout = d_osp;
        m.wave_mom = s->cur_stats;        // (split-f16 path: its clamps turn NaN into finite values)
        ProfScope ps(T_MLP, B);
        return dz_launch_mlp_head(m, st);
    }
    if (s->pre) {
        p.Xsplit = lin; p.xplane = rows * 256; p.Wsplit = s->w.lin0_split;
        p.Ysplit = s->m0; p.yplane = rows * 128;
        { ProfScope ps(T_MLP, B); if ((rc = dz_launch_gemm_pre(p, st))) return rc; }
        p.Xsplit = s->m0; p.xplane = rows * 128; p.Wsplit = s->w.lin1_split;
        p.Ysplit = nullptr; p.Y = s->m1; p.W = s->w.lin1_w; p.bias = s->w.lin1_b;
        p.Cin = 128; p.K = 128; p.Kpad = 128; p.ldx = 128;
        { ProfScope ps(T_MLP, B); if ((rc = dz_launch_gemm_pre(p, st))) return rc; }
        p.Xsplit = nullptr;
    } else {
        p.X = lin; p.Y = s->m0;
        { ProfScope ps(T_MLP, B); if ((rc = run_gemm(p, s->w.lin0_split, st))) return rc; }
        p.X = s->m0; p.W = s->w.lin1_w; p.bias = s->w.lin1_b; p.Y = s->m1;
        p.Cin = 128; p.K = 128; p.Kpad = 128; p.ldx = 128;
        { ProfScope ps(T_MLP, B); if ((rc = run_gemm(p, s->w.lin1_split, st))) return rc; }
    }
    // classifier + sigmoid / powerset decision (+ OverlappedSpeechPenalty weights): one launch
    ProfScope ps(T_CLS, B);
    return dz_launch_seg_head(s->m1, s->w.cls_w, s->w.cls_b, B, F, s->w.num_classes, s->w.num_speakers,
                              s->w.powerset, d_out, gamma, beta, normalize, d_osp, st, s->pre ? s->cur_stats : nullptr);
}

// ---------------------------------------------------------------------------
// embedding
// ---------------------------------------------------------------------------
struct dz_emb {
    dz_ctx* ctx;
    dz_emb_weights w;
    SincGeom g;
    int Bm, T[5];
    bool pre;    // tdnn2..5 on k_gemm_pre.hip (tdnn1 writes f16 hi/lo planes)
    const float* ext_stats;   // dz_emb_use_wave_stats: consumed (and cleared) by the next forward
    const float* cur_stats;   // the slice moments dz_emb_frames normalised with (NaN rows of dz_emb_pool, dz_ws_bad)
    int ext_conv0_B;          // dz_sinc_conv0_pair: see dz_seg
    char* arena;
    SincScratch ss;
    float *a, *b, *x5, *pooled, *parts, *ppart, *ps0;
    // tdnn5 + statistics pooling in one launch (k_gemm_pre.hip, pooled epilogue): dz_emb_frames then
    // stops after tdnn4 and leaves tdnn5 to the call that brings the pooling weights
    int pending_B;            // > 0: frames of that many chunks are waiting at tdnn4's output
    const float* pending_in;  // tdnn4's planes
};
static const int kTdnnTaps[5] = {5, 3, 3, 1, 1};
static const int kTdnnDil[5] = {1, 2, 3, 1, 1};
static const int kPoolLd = 3008;
static const int kMaxSpk = 8;
static const int kEmbSplit = 16;  // split-K of Linear(3000, 512): 8 tiles -> 128 workgroups

static void emb_carve(dz_emb* e, Arena& a) {
    e->ss.carve(a, e->g, e->Bm);
    // every TDNN activation keeps the row pitch of the network input (P2 = 293 frames per chunk,
    // the first T[i] rows valid): layers 2..5 then run as ONE flattened GEMM over Bm * P2 rows
    e->a = a.take((size_t)e->Bm * e->g.P2 * 512);
    e->b = a.take((size_t)e->Bm * e->g.P2 * 512);
    e->x5 = a.take((size_t)e->Bm * e->g.P2 * 1536);
    e->pooled = a.take((size_t)e->Bm * kMaxSpk * kPoolLd);
    const int np = dz_pool_pieces(e->g.P2);
    e->ppart = a.take((size_t)e->Bm * np * 4 * 1536 * 2);      // [chunk][np pieces][<= 4 speakers][1536][2]
    e->ps0 = a.take((size_t)e->Bm * np * 4 * 2);
    e->parts = a.take((size_t)kEmbSplit * e->Bm * kMaxSpk * 512);
}

extern "C" int dz_emb_create(dz_ctx* ctx, const dz_emb_weights* w, int max_batch, int num_samples,
                             dz_emb** out) {
    DZ_REQUIRE(ctx && w && out, "dz_emb_create: NULL argument");
    DZ_REQUIRE(max_batch >= 1, "dz_emb_create: max_batch %d", max_batch);
    DZ_REQUIRE(w->dimension == 512, "dz_emb_create: dimension %d (only 512 is built)", w->dimension);
    const SincGeom g = sinc_geom(num_samples, w->sinc.filt_split != nullptr);
    DZ_REQUIRE(g.ok && g.P2 > 14, "dz_emb_create: %d samples is too short", num_samples);
    DZ_REQUIRE((w->sinc.w1_split != nullptr) == (w->sinc.w2_split != nullptr) &&
                   (w->sinc.w1_split != nullptr) == (w->tw_split[0] != nullptr),
               "dz_emb_create: the split planes of SincNet conv1 / conv2 and of tdnn1 must be all present "
               "or all absent (the consumer of the last SincNet stage finalises its InstanceNorm)");
    DZ_HIP(hipSetDevice(ctx->device));
    dz_emb* e = new (std::nothrow) dz_emb;
    DZ_REQUIRE(e != nullptr, "dz_emb_create: out of memory");
    e->ctx = ctx; e->w = *w; e->g = g; e->Bm = max_batch; e->arena = nullptr; e->ext_stats = nullptr; e->cur_stats = nullptr;
    e->ext_conv0_B = 0;
    e->pending_B = 0; e->pending_in = nullptr;
    e->pre = w->tw_split[0] && w->tw_split[1] && w->tw_split[2] &&
             w->tw_split[3] && w->tw_split[4];
    int t = g.P2;
    for (int i = 0; i < 5; ++i) {
        t -= (kTdnnTaps[i] - 1) * kTdnnDil[i];
        e->T[i] = t;
    }
    Arena measure;
    emb_carve(e, measure);
    hipError_t err = hipMalloc((void**)&e->arena, measure.used);
    if (err != hipSuccess) {
        dz_set_error("dz_emb_create: hipMalloc(%zu) failed: %s", measure.used, hipGetErrorString(err));
        delete e;
        return 1;
    }
    DZ_HIP(hipMemset(e->arena, 0, measure.used));
    Arena real;
    real.base = e->arena; real.size = measure.used;
    emb_carve(e, real);
    if (int rc = sinc_repack(w->sinc, e->ss)) { (void)hipFree(e->arena); delete e; return rc; }
    *out = e;
    return 0;
}

extern "C" int dz_emb_destroy(dz_emb* emb) {
    if (emb) {
        if (emb->arena) (void)hipFree(emb->arena);
        delete emb;
    }
    return 0;
}

// frame features: wave (B) -> x5 [B][T5][1536]
static int emb_frames(dz_emb* e, const float* d_wave, long long stride, int B, hipStream_t st) {
    int rc;
    const float* ext = e->ext_stats;
    e->ext_stats = nullptr;
    const int pair_B = e->ext_conv0_B;
    e->ext_conv0_B = 0;
    DZ_REQUIRE(pair_B == 0 || pair_B == B, "dz_emb_frames: dz_sinc_conv0_pair ran for %d chunks, this call has %d", pair_B, B);
    if ((rc = run_sincnet(e->w.sinc, e->g, e->ss, d_wave, stride, B, st, ext, pair_B > 0))) return rc;
    e->cur_stats = ext ? ext : e->ss.stats;
    const int cin[5] = {64, 512, 512, 512, 512};
    const int npad[5] = {512, 512, 512, 512, 1536};
    // Row pitch P = P2 for every activation.  tdnn1 normalises on load with per-chunk statistics, so
    // it runs per chunk; tdnn2..5 run flattened over all B * P rows: a row whose taps reach past
    // the valid frames of its chunk (t >= T[i]) computes garbage that no valid row ever reads (a
    // valid output row t < T[i] reads input rows t + tap * dil < T[i-1] of the same chunk), in
    // exchange the M tiles are 95 % full instead of 73 % (279 rows in 3 x 128).
    const float* in = e->ss.y2;
    const int P = e->g.P2;
    // e->pre: tdnn1 writes its output as f16 (hi, lo) planes (same bytes, same buffers), tdnn2..5 run
    // on k_gemm_pre.hip, tdnn5 writes the f32 features the statistics pooling reads
    const long long plane = (long long)B * P * 512;
    e->pending_B = 0;
    for (int i = 0; i < 5; ++i) {
        DzConvGemm p;
        memset(&p, 0, sizeof(p));
        float* outp = (i == 4) ? e->x5 : ((i & 1) ? e->b : e->a);
        const int span = (kTdnnTaps[i] - 1) * kTdnnDil[i];
        p.X = in; p.W = e->w.tw[i]; p.bias = e->w.tb[i]; p.e0 = e->w.ts[i]; p.e1 = e->w.th[i];
        p.Y = outp;
        p.Cin = cin[i]; p.taps = kTdnnTaps[i];
        p.dil = kTdnnDil[i]; p.K = cin[i] * kTdnnTaps[i]; p.Kpad = (p.K + 31) / 32 * 32;
        p.Npad = npad[i]; p.Nstore = npad[i]; p.ldx = cin[i]; p.ldy = npad[i];
        p.epi = DZ_EPI_TDNN;
        const bool pre0 = i == 0 && e->pre && sinc_pre_split_ok(e->w.sinc, e->w.tw0_split_kb);
        if (pre0) {                   // y2 -> normalised planes; tdnn1 flattened over B * P rows like the layers behind it
            if ((rc = sinc_norm_split(e->w.sinc, e->g, e->ss, B, st))) return rc;
            p.B = 1; p.Tin = B * P; p.Tout = p.Tstore = B * P - span;
        } else if (i == 0 && !e->pre && !e->w.tw_split[0] && sinc_f32_norm_pass(e->w.sinc)) {
            // exact f32: the normalised y2 (run_sincnet -> y2s), tdnn1 flattened over B * P rows like the layers behind it
            p.X = e->ss.y2s; p.B = 1; p.Tin = B * P; p.Tout = p.Tstore = B * P - span;
        } else if (i == 0) {
            p.B = B; p.Tin = P; p.Tout = p.Tstore = e->T[0];
            p.xbs = (long long)P * cin[i]; p.ybs = (long long)P * npad[i];
            sinc_out_norm(p, e->w.sinc, e->g, e->ss, e->w.tw_split[0] != nullptr);
            if (e->pre) { p.Y = nullptr; p.Ysplit = outp; p.yplane = plane; }
        } else {
            p.B = 1; p.Tin = B * P; p.Tout = p.Tstore = B * P - span;
        }
        // (the pooled epilogue walks at most two chunks per 128-row tile: chunk pitch >= 128 rows, i.e. windows of
        // ~2.3 s and longer; shorter windows keep the unfused tdnn5 + stats_pool)
        if (i == 4 && e->pre && pool_fuse_enabled() && e->g.P2 >= 128 && e->T[4] >= 2 && dz_gemm_pre_pool_ok(p)) {
            // tdnn5 runs with the pooling in its epilogue, i.e. when the weights are known (emb_head)
            e->pending_B = B;
            e->pending_in = in;
            return 0;
        }
        { ProfScope ps(T_TDNN1 + i, B);
          if (pre0) {
              p.X = nullptr; p.Xsplit = e->ss.y2s; p.xplane = (long long)B * P * 64; p.Wsplit = e->w.tw0_split_kb;
              p.Y = nullptr; p.Ysplit = outp; p.yplane = plane;
              rc = dz_launch_gemm_pre(p, st);
          } else if (i > 0 && e->pre) {
              p.X = nullptr; p.Xsplit = in; p.xplane = plane; p.Wsplit = e->w.tw_split[i];
              if (i < 4) { p.Y = nullptr; p.Ysplit = outp; p.yplane = plane; }
              rc = dz_launch_gemm_pre(p, st);
          } else {
              rc = run_gemm(p, e->w.tw_split[i], st);
          }
          if (rc) return rc; }
        in = outp;
    }
    return 0;
}

static int emb_head(dz_emb* e, const float* d_weights, int Fw, int rows, int rows_per_x,
                    int normalize, float* d_out, hipStream_t st) {
    int rc;
    if (e->w.pool_nearest && d_weights) Fw = -Fw;      // the internal launchers carry the resampling mode in the sign (dz_pool_weight)
    const int nx = rows / rows_per_x;
    if (e->pending_B > 0) {
        DZ_REQUIRE(nx == e->pending_B, "dz_emb_pool: %d chunks, but the frame features of %d are pending", nx,
                   e->pending_B);
        const int B = e->pending_B, P = e->g.P2;
        DzConvGemm p;
        memset(&p, 0, sizeof(p));
        p.Xsplit = e->pending_in; p.xplane = (long long)B * P * 512; p.Wsplit = e->w.tw_split[4];
        p.W = e->w.tw[4]; p.bias = e->w.tb[4]; p.e0 = e->w.ts[4]; p.e1 = e->w.th[4];
        p.B = 1; p.Tin = p.Tout = p.Tstore = B * P; p.Cin = 512; p.taps = 1; p.dil = 1; p.K = 512; p.Kpad = 512;
        p.Npad = 1536; p.Nstore = 1536; p.ldx = 512; p.ldy = 1536; p.epi = DZ_EPI_TDNN;
        if (rows_per_x <= 4) {
            DzPoolFuse q;
            q.w = d_weights; q.Fw = Fw; q.K = rows_per_x; q.P = P; q.T = e->T[4]; q.np = dz_pool_pieces(P);
            q.part = e->ppart; q.s0 = e->ps0;
            { ProfScope ps(T_TDNN5, B); if ((rc = dz_launch_gemm_pre_pool(p, q, st))) return rc; }
            // the frames stay pending: tdnn4's planes are intact until the next dz_emb_frames, so a second
            // dz_emb_pool on the same frames (other weights) runs the pooled tdnn5 again
            { ProfScope ps(T_POOL, B);
              if ((rc = dz_launch_pool_combine(e->ppart, e->ps0, B, rows_per_x, dz_pool_pieces(P), P, e->T[4], 1500, 1536, e->pooled,
                                               kPoolLd, st)))
                  return rc; }
        } else {            // more than 4 speakers per chunk: plain tdnn5, then the stand-alone pooling below
            p.Y = e->x5;
            { ProfScope ps(T_TDNN5, B); if ((rc = dz_launch_gemm_pre(p, st))) return rc; }
            e->pending_B = 0;
            ProfScope ps(T_POOL, nx);
            if ((rc = dz_launch_stats_pool(e->x5, (long long)P * 1536, e->T[4], 1500, 1536, d_weights, Fw, rows,
                                           rows_per_x, e->pooled, kPoolLd, st)))
                return rc;
        }
    } else
    { ProfScope ps(T_POOL, nx);
    if ((rc = dz_launch_stats_pool(e->x5, (long long)e->g.P2 * 1536, e->T[4], 1500, 1536, d_weights, Fw,
                                   rows, rows_per_x, e->pooled, kPoolLd, st)))
        return rc; }
    DzConvGemm p;
    memset(&p, 0, sizeof(p));
    // M = rows is tiny (3 per chunk): split K 16 ways so 128 workgroups share the 3008-deep
    // contraction, then reduce the partials in fixed order (+ L2 normalisation) in one pass
    p.X = e->pooled; p.W = e->w.emb_w; p.bias = e->w.emb_b; p.Y = e->parts;
    p.B = 1; p.Tin = p.Tout = p.Tstore = rows; p.Cin = kPoolLd; p.taps = 1; p.dil = 1;
    p.K = kPoolLd; p.Kpad = kPoolLd; p.Npad = 512; p.Nstore = 512; p.ldx = kPoolLd; p.ldy = 512;
    p.epi = DZ_EPI_BIAS; p.ksplit = kEmbSplit; p.ysplit = (long long)rows * 512;
    { ProfScope ps(T_EMBLIN, rows / rows_per_x); if ((rc = dz_launch_convgemm(p, st))) return rc; }
    ProfScope ps(T_L2, rows / rows_per_x);
    return dz_launch_splitk_finish(e->parts, kEmbSplit, p.ysplit, rows, 512, normalize, d_out, st, e->pre ? e->cur_stats : nullptr,
                                   rows_per_x);
}

extern "C" int dz_emb_forward(dz_emb* e, const float* d_wave, long long wave_stride,
                              const float* d_weights, int n_rows, int weight_frames, float* d_out,
                              void* stream) {
    DZ_REQUIRE(e && d_out, "dz_emb_forward: NULL argument");
    DZ_REQUIRE(n_rows >= 1 && n_rows <= e->Bm, "dz_emb_forward: %d rows outside [1, %d]", n_rows,
               e->Bm);
    DZ_REQUIRE(d_weights == nullptr || weight_frames >= 2, "dz_emb_forward: weight_frames %d",
               weight_frames);
    int rc;
    if ((rc = check_wave("dz_emb_forward", d_wave, wave_stride, e->g.S))) return rc;
    DZ_HIP(hipSetDevice(e->ctx->device));
    DzRangeScope range_scope(e->ctx->oflag_dev);
    hipStream_t st = (hipStream_t)stream;
    if ((rc = emb_frames(e, d_wave, wave_stride, n_rows, st))) return rc;
    return emb_head(e, d_weights, d_weights ? weight_frames : e->T[4], n_rows, 1, 0, d_out, st);
}

extern "C" int dz_emb_forward_multi(dz_emb* e, const float* d_wave, long long wave_stride,
                                    const float* d_weights, int batch, int num_speakers,
                                    int weight_frames, int normalize, float* d_out, void* stream) {
    DZ_REQUIRE(e && d_out && d_weights, "dz_emb_forward_multi: NULL argument");
    DZ_REQUIRE(batch >= 1 && batch <= e->Bm, "dz_emb_forward_multi: batch %d outside [1, %d]",
               batch, e->Bm);
    DZ_REQUIRE(num_speakers >= 1 && num_speakers <= kMaxSpk,
               "dz_emb_forward_multi: %d speakers outside [1, %d]", num_speakers, kMaxSpk);
    DZ_REQUIRE(weight_frames >= 2, "dz_emb_forward_multi: weight_frames %d", weight_frames);
    int rc;
    if ((rc = check_wave("dz_emb_forward_multi", d_wave, wave_stride, e->g.S))) return rc;
    DZ_HIP(hipSetDevice(e->ctx->device));
    DzRangeScope range_scope(e->ctx->oflag_dev);
    hipStream_t st = (hipStream_t)stream;
    if ((rc = emb_frames(e, d_wave, wave_stride, batch, st))) return rc;
    return emb_head(e, d_weights, weight_frames, batch * num_speakers, num_speakers, normalize,
                    d_out, st);
}

// frame features only (SincNet + 5 TDNN) -> internal buffer; independent of the segmentation,
// so a caller can run it on a second stream beside dz_seg_forward
extern "C" int dz_emb_frames(dz_emb* e, const float* d_wave, long long wave_stride, int batch,
                             void* stream) {
    DZ_REQUIRE(e, "dz_emb_frames: NULL argument");
    DZ_REQUIRE(batch >= 1 && batch <= e->Bm, "dz_emb_frames: batch %d outside [1, %d]", batch, e->Bm);
    int rc;
    if ((rc = check_wave("dz_emb_frames", d_wave, wave_stride, e->g.S))) return rc;
    DZ_HIP(hipSetDevice(e->ctx->device));
    DzRangeScope range_scope(e->ctx->oflag_dev);
    return emb_frames(e, d_wave, wave_stride, batch, (hipStream_t)stream);
}
// pooling + Linear (+ normalisation) of the frame features left by the last dz_emb_frames
extern "C" int dz_emb_pool(dz_emb* e, const float* d_weights, int batch, int num_speakers,
                           int weight_frames, int normalize, float* d_out, void* stream) {
    DZ_REQUIRE(e && d_out && d_weights, "dz_emb_pool: NULL argument");
    DZ_REQUIRE(batch >= 1 && batch <= e->Bm, "dz_emb_pool: batch %d outside [1, %d]", batch, e->Bm);
    DZ_REQUIRE(num_speakers >= 1 && num_speakers <= kMaxSpk, "dz_emb_pool: %d speakers", num_speakers);
    DZ_REQUIRE(weight_frames >= 2, "dz_emb_pool: weight_frames %d", weight_frames);
    DZ_HIP(hipSetDevice(e->ctx->device));
    DzRangeScope range_scope(e->ctx->oflag_dev);
    return emb_head(e, d_weights, weight_frames, batch * num_speakers, num_speakers, normalize,
                    d_out, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// InstanceNorm1d(1) statistics of the windows, shared by both networks
// ---------------------------------------------------------------------------
extern "C" int dz_wave_stats_floats(void) { return 2 * DZ_WS_G; }
extern "C" int dz_wave_stats(dz_ctx* ctx, const float* d_wave, long long wave_stride, int batch,
                             int num_samples, float* d_moments, void* stream) {
    DZ_REQUIRE(ctx && d_moments, "dz_wave_stats: NULL argument");
    DZ_REQUIRE(batch >= 1 && num_samples >= 1, "dz_wave_stats: empty input");
    int rc;
    if ((rc = check_wave("dz_wave_stats", d_wave, wave_stride, num_samples))) return rc;
    DZ_HIP(hipSetDevice(ctx->device));
    ProfScope ps(T_WAVE, batch);
    return dz_launch_wave_stats(d_wave, wave_stride, batch, num_samples, d_moments, (hipStream_t)stream);
}
#ifdef DZ_EXPERIMENTS
// The first SincNet stage of BOTH networks in one launch (k_front.hip sinc_conv0_pair_kernel): writes y0 / part0
// of the two handles; the next dz_seg_forward* / dz_emb_frames of each handle (same B, enqueued behind this launch:
// the same stream, or one that waits for an event recorded after it) then starts at conv1.
extern "C" int dz_sinc_conv0_pair(dz_seg* seg, dz_emb* emb, const float* d_wave, long long wave_stride, int batch,
                                  const float* d_moments, const void* d_pair_planes, const float* d_pair_bsum,
                                  void* stream) {
    DZ_REQUIRE(seg && emb && d_moments && d_pair_planes && d_pair_bsum, "dz_sinc_conv0_pair: NULL argument");
    DZ_REQUIRE(seg->ctx == emb->ctx, "dz_sinc_conv0_pair: the two handles belong to different contexts");
    DZ_REQUIRE(batch >= 1 && batch <= seg->Bm && batch <= emb->Bm, "dz_sinc_conv0_pair: batch %d outside [1, %d]", batch,
               seg->Bm < emb->Bm ? seg->Bm : emb->Bm);
    DZ_REQUIRE(seg->g.S == emb->g.S && seg->g.nt0 == emb->g.nt0 && seg->g.P0 == emb->g.P0,
               "dz_sinc_conv0_pair: the handles were created for different window lengths");
    DZ_REQUIRE(seg->w.sinc.filt_split && emb->w.sinc.filt_split,
               "dz_sinc_conv0_pair: both networks must be in the split-f16 precision (the exact-f32 path keeps one "
               "launch per network)");
    int rc;
    if ((rc = check_wave("dz_sinc_conv0_pair", d_wave, wave_stride, seg->g.S))) return rc;
    DZ_HIP(hipSetDevice(seg->ctx->device));
    DzRangeScope range_scope(seg->ctx->oflag_dev);
    { ProfScope ps(T_CONV0_PAIR, batch);
      if ((rc = dz_launch_sinc_conv0_pair(d_wave, wave_stride, batch, seg->g.S, d_moments, d_pair_planes, d_pair_bsum,
                                          seg->w.sinc.wav_gamma, emb->w.sinc.wav_gamma, seg->ss.y0, emb->ss.y0, seg->g.P0,
                                          seg->ss.part0, emb->ss.part0, seg->g.nt0, (hipStream_t)stream)))
          return rc; }
    seg->ext_conv0_B = emb->ext_conv0_B = batch;
    return 0;
}
#endif  // DZ_EXPERIMENTS
extern "C" int dz_seg_use_wave_stats(dz_seg* seg, const float* d_moments) {
    DZ_REQUIRE(seg != nullptr, "dz_seg_use_wave_stats: NULL handle");
    seg->ext_stats = d_moments;
    return 0;
}
extern "C" int dz_emb_use_wave_stats(dz_emb* emb, const float* d_moments) {
    DZ_REQUIRE(emb != nullptr, "dz_emb_use_wave_stats: NULL handle");
    emb->ext_stats = d_moments;
    return 0;
}

// ---------------------------------------------------------------------------
// small ops
// ---------------------------------------------------------------------------
extern "C" int dz_osp(dz_ctx* ctx, const float* d_seg, int batch, int frames, int speakers,
                      float gamma, float beta, int normalize, int speaker_major, float* d_out,
                      void* stream) {
    DZ_REQUIRE(ctx && d_seg && d_out, "dz_osp: NULL argument");
    DZ_REQUIRE(batch >= 1 && frames >= 1, "dz_osp: empty input");
    DZ_HIP(hipSetDevice(ctx->device));
    ProfScope ps(T_OSP, batch);
    return dz_launch_osp(d_seg, batch, frames, speakers, gamma, beta, normalize, speaker_major,
                         d_out, (hipStream_t)stream);
}

extern "C" int dz_l2_normalize(dz_ctx* ctx, float* d_emb, int rows, int dim, float norm,
                               void* stream) {
    DZ_REQUIRE(ctx && d_emb, "dz_l2_normalize: NULL argument");
    DZ_REQUIRE(rows >= 1 && dim >= 1, "dz_l2_normalize: empty input");
    DZ_HIP(hipSetDevice(ctx->device));
    return dz_launch_l2norm(d_emb, rows, dim, norm, (hipStream_t)stream);
}

extern "C" int dz_cdist_cosine(dz_ctx* ctx, const float* d_emb, const double* d_centers,
                               int n_streams, int k_local, int g_global, int dim, double* d_out,
                               void* stream) {
    DZ_REQUIRE(ctx && d_emb && d_centers && d_out, "dz_cdist_cosine: NULL argument");
    DZ_REQUIRE(n_streams >= 1 && k_local >= 1 && g_global >= 1 && dim >= 1,
               "dz_cdist_cosine: empty input");
    DZ_HIP(hipSetDevice(ctx->device));
    return dz_launch_cdist(d_emb, d_centers, n_streams, k_local, g_global, dim, d_out,
                           (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// kernel-level entry points (parity tests)
// ---------------------------------------------------------------------------
extern "C" int dz_k_convgemm(dz_ctx* ctx, const dz_convgemm_desc* d, void* stream) {
    DZ_REQUIRE(ctx && d, "dz_k_convgemm: NULL argument");
    DZ_HIP(hipSetDevice(ctx->device));
    DzRangeScope range_scope(ctx->oflag_dev);
    return dz_launch_convgemm(*d, (hipStream_t)stream);
}
extern "C" int dz_k_gemm_f32(dz_ctx* ctx, const dz_convgemm_desc* d, void* stream) {
    DZ_REQUIRE(ctx && d, "dz_k_gemm_f32: NULL argument");
    DZ_HIP(hipSetDevice(ctx->device));
    return dz_launch_gemm_f32(*d, (hipStream_t)stream);
}
extern "C" int dz_k_gemm_split(dz_ctx* ctx, const dz_convgemm_desc* d, void* stream) {
    DZ_REQUIRE(ctx && d, "dz_k_gemm_split: NULL argument");
    DZ_HIP(hipSetDevice(ctx->device));
    DzRangeScope range_scope(ctx->oflag_dev);
    return dz_launch_gemm_split(*d, (hipStream_t)stream);
}
extern "C" int dz_k_gemm_pre(dz_ctx* ctx, const dz_convgemm_desc* d, void* stream) {
    DZ_REQUIRE(ctx && d, "dz_k_gemm_pre: NULL argument");
    DZ_HIP(hipSetDevice(ctx->device));
    DzRangeScope range_scope(ctx->oflag_dev);
    return dz_launch_gemm_pre(*d, (hipStream_t)stream);
}
#ifdef DZ_EXPERIMENTS
extern "C" int dz_k_gemm_g2(dz_ctx* ctx, const dz_convgemm_desc* d, int row_fragments, void* stream) {
    DZ_REQUIRE(ctx && d, "dz_k_gemm_g2: NULL argument");
    DZ_REQUIRE(d->Wsplit && d->Xsplit && (d->Y || d->Ysplit) && d->B == 1 && d->K == d->Kpad && d->K == d->taps * d->Cin &&
                   d->Cin % 32 == 0 && d->Npad % 128 == 0 && d->pad == 0 && !d->X2 && !d->rowbias && d->ksplit <= 1 &&
                   !d->norm_on_load && d->Tout > 0 && d->Tout == d->Tin - (d->taps - 1) * d->dil && d->ldx % 32 == 0 &&
                   d->xplane % d->ldx == 0 && d->xplane / d->ldx >= d->Tin,
               "dz_k_gemm_g2: the requirements of dz_k_gemm_pre apply");
    DZ_HIP(hipSetDevice(ctx->device));
    DzRangeScope range_scope(ctx->oflag_dev);
    DzConvGemm p = *d;
    if (!p.oflag) p.oflag = ctx->oflag_dev;
    return dz_launch_gemm_g2(p, row_fragments, (hipStream_t)stream);
}
extern "C" int dz_k_gemm_g3(dz_ctx* ctx, const dz_convgemm_desc* d, int row_fragments, void* stream) {
    DZ_REQUIRE(ctx && d, "dz_k_gemm_g3: NULL argument");
    DZ_REQUIRE(d->Wsplit && d->Xsplit && (d->Y || d->Ysplit) && d->B == 1 && d->K == d->Kpad && d->K == d->taps * d->Cin &&
                   d->Cin % 32 == 0 && d->Npad % 128 == 0 && d->pad == 0 && !d->X2 && !d->rowbias && d->ksplit <= 1 &&
                   !d->norm_on_load && d->Tout > 0 && d->Tout == d->Tin - (d->taps - 1) * d->dil && d->ldx % 32 == 0 &&
                   d->xplane % d->ldx == 0 && d->xplane / d->ldx >= d->Tin,
               "dz_k_gemm_g3: the requirements of dz_k_gemm_pre apply");
    DZ_HIP(hipSetDevice(ctx->device));
    DzRangeScope range_scope(ctx->oflag_dev);
    DzConvGemm p = *d;
    if (!p.oflag) p.oflag = ctx->oflag_dev;
    return dz_launch_gemm_g3(p, row_fragments, (hipStream_t)stream);
}
#endif  // DZ_EXPERIMENTS
extern "C" int dz_k_mlp_head(dz_ctx* ctx, const void* xsplit, long long xplane, const void* w0split,
                             const void* w1split, const float* b0, const float* b1, const float* cw,
                             const float* cb, int rows, int frames, int classes, int speakers, int powerset,
                             float gamma, float beta, float* d_seg, float* d_weights, void* stream) {
    DZ_REQUIRE(ctx, "dz_k_mlp_head: NULL argument");
    DZ_HIP(hipSetDevice(ctx->device));
    DzRangeScope range_scope(ctx->oflag_dev);
    DzMlpHead m{};
    m.Xsplit = xsplit; m.xplane = xplane; m.W0split = w0split; m.W1split = w1split;
    m.b0 = b0; m.b1 = b1; m.cw = cw; m.cb = cb;
    m.rows = rows; m.F = frames; m.classes = classes; m.K = speakers; m.powerset = powerset;
    m.gamma = gamma; m.beta = beta; m.seg = d_seg; m.wout = d_weights;
    return dz_launch_mlp_head(m, (hipStream_t)stream);
}
extern "C" int dz_k_seg_head(dz_ctx* ctx, const float* m1, const float* cw, const float* cb, int batch,
                             int frames, int classes, int speakers, int powerset, float* d_seg, float gamma,
                             float beta, int normalize, float* d_weights, void* stream) {
    DZ_REQUIRE(ctx && m1 && cw && cb && d_seg, "dz_k_seg_head: NULL argument");
    DZ_HIP(hipSetDevice(ctx->device));
    return dz_launch_seg_head(m1, cw, cb, batch, frames, classes, speakers, powerset, d_seg, gamma, beta,
                              normalize, d_weights, (hipStream_t)stream);
}
extern "C" int dz_k_conv_pool(dz_ctx* ctx, const dz_convgemm_desc* d, void* stream) {
    DZ_REQUIRE(ctx && d, "dz_k_conv_pool: NULL argument");
    DZ_HIP(hipSetDevice(ctx->device));
    DzRangeScope range_scope(ctx->oflag_dev);
    // kernel-level entry: the weights go into fragment order on every call (the handles do it once, at create)
    DZ_REQUIRE(d->Wsplit && (d->Cin == 80 || d->Cin == 64), "dz_k_conv_pool: Wsplit is NULL or Cin is not 80 / 64");
    std::lock_guard<std::mutex> frag_lock(ctx->frag_mu);
    if (!ctx->convp_frag) DZ_HIP(hipMalloc(&ctx->convp_frag, (size_t)dz_conv_pool_wfrag_bytes(80)));
    if (ctx->convp_used && ctx->convp_user != (hipStream_t)stream) DZ_HIP(hipStreamSynchronize(ctx->convp_user));
    ctx->convp_user = (hipStream_t)stream; ctx->convp_used = true;
    int rc;
    if (!(dz_option(DZ_OPT_PACK_CACHE) && ctx->convp_src == d->Wsplit && ctx->convp_cin == d->Cin && ctx->convp_kpad == d->Kpad)) {
        if ((rc = dz_launch_conv_pool_wfrag(d->Cin, d->Wsplit, d->Kpad, ctx->convp_frag, (hipStream_t)stream))) return rc;
        ctx->convp_src = d->Wsplit; ctx->convp_cin = d->Cin; ctx->convp_kpad = d->Kpad;
    }
    return dz_launch_conv_pool(*d, (hipStream_t)stream, ctx->convp_frag);
}
#ifdef DZ_EXPERIMENTS
// phase time stamps of conv_pool_h (tools/kbench.py): 2 x 64 shader-clock stamps per workgroup
extern "C" int dz_k_conv_pool_debug(long long* d_stamps) {
    dz_conv_pool_dbg = d_stamps;
    return 0;
}
#endif
extern "C" int dz_k_convgemm_ntile(int t_out) { return dz_convgemm_ntile(t_out); }
extern "C" int dz_k_wave_stats(dz_ctx* ctx, const float* d_wave, long long stride, int batch,
                               int samples, float* d_stats, void* stream) {
    DZ_REQUIRE(ctx && d_stats, "dz_k_wave_stats: NULL argument");
    int rc;
    if ((rc = check_wave("dz_k_wave_stats", d_wave, stride, samples))) return rc;
    DZ_HIP(hipSetDevice(ctx->device));
    // kernel-level entry: slice moments into a throw-away buffer, then the (mean, rstd) contract
    float* mom = nullptr;
    DZ_HIP(hipMalloc((void**)&mom, (size_t)batch * 2 * DZ_WS_G * sizeof(float)));
    rc = dz_launch_wave_stats(d_wave, stride, batch, samples, mom, (hipStream_t)stream);
    if (!rc) rc = dz_launch_wave_stats_combine(mom, batch, samples, d_stats, (hipStream_t)stream);
    (void)hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(mom);
    return rc;
}
extern "C" int dz_k_sinc_conv0(dz_ctx* ctx, const float* d_wave, long long stride, int batch,
                               int samples, const float* d_stats, float gamma, float beta,
                               const float* d_filt, float* d_y0, float* d_partials, void* stream) {
    DZ_REQUIRE(ctx && d_stats && d_filt && d_y0 && d_partials, "dz_k_sinc_conv0: NULL argument");
    int rc;
    if ((rc = check_wave("dz_k_sinc_conv0", d_wave, stride, samples))) return rc;
    const SincGeom g = sinc_geom(samples);
    DZ_REQUIRE(g.P0 > 0, "dz_k_sinc_conv0: %d samples is too short", samples);
    DZ_HIP(hipSetDevice(ctx->device));
    return dz_launch_sinc_conv0(d_wave, stride, batch, samples, d_stats, 0, gamma, beta, d_filt, d_y0,
                                g.P0, d_partials, g.nt0, (hipStream_t)stream);
}
extern "C" int dz_k_sinc_conv0_split(dz_ctx* ctx, const float* d_wave, long long stride, int batch,
                                     int samples, const float* d_stats, float gamma, float beta,
                                     const void* d_filt_split, float* d_y0, float* d_partials,
                                     void* stream) {
    DZ_REQUIRE(ctx && d_stats && d_filt_split && d_y0 && d_partials, "dz_k_sinc_conv0_split: NULL argument");
    int rc;
    if ((rc = check_wave("dz_k_sinc_conv0_split", d_wave, stride, samples))) return rc;
    const SincGeom g = sinc_geom(samples, true);
    DZ_REQUIRE(g.P0 > 0, "dz_k_sinc_conv0_split: %d samples is too short", samples);
    DZ_HIP(hipSetDevice(ctx->device));
    DzRangeScope range_scope(ctx->oflag_dev);
    // kernel-level entry: the bank goes into fragment order on every call (the handles do it once, at create)
    // (option "pack_cache", off by default: skip the repack when the bank pointer is the one of the previous call —
    // for the timing tools, whose weights do not change; a framework's allocator may hand the same address out again)
    std::lock_guard<std::mutex> frag_lock(ctx->frag_mu);
    if (!ctx->conv0_frag) DZ_HIP(hipMalloc(&ctx->conv0_frag, (size_t)dz_sinc_bank_frag_bytes()));
    if (ctx->conv0_used && ctx->conv0_user != (hipStream_t)stream) DZ_HIP(hipStreamSynchronize(ctx->conv0_user));
    ctx->conv0_user = (hipStream_t)stream; ctx->conv0_used = true;
    if (!(dz_option(DZ_OPT_PACK_CACHE) && ctx->conv0_src == d_filt_split)) {
        if ((rc = dz_launch_sinc_bank_frag(d_filt_split, ctx->conv0_frag, (hipStream_t)stream))) return rc;
        ctx->conv0_src = d_filt_split;
    }
    return dz_launch_sinc_conv0_split(d_wave, stride, batch, samples, d_stats, 0, gamma, beta,
                                      d_filt_split, d_y0, g.P0, d_partials, g.nt0, (hipStream_t)stream, ctx->conv0_frag);
}
extern "C" int dz_k_conv0_split_ntile(int samples) { return sinc_geom(samples, true).nt0; }
#ifdef DZ_EXPERIMENTS
extern "C" int dz_k_sinc_conv0_pair(dz_ctx* ctx, const float* d_wave, long long stride, int batch, int samples,
                                    const float* d_moments, const void* d_pair_planes, const float* d_pair_bsum,
                                    float gamma_seg, float gamma_emb, float* d_y0_seg, float* d_y0_emb,
                                    float* d_part_seg, float* d_part_emb, void* stream) {
    DZ_REQUIRE(ctx && d_moments && d_pair_planes && d_pair_bsum && d_y0_seg && d_y0_emb && d_part_seg && d_part_emb,
               "dz_k_sinc_conv0_pair: NULL argument");
    const SincGeom g = sinc_geom(samples, true);
    DZ_REQUIRE(batch >= 1 && g.F0 >= 3, "dz_k_sinc_conv0_pair: empty input");
    int rc;
    if ((rc = check_wave("dz_k_sinc_conv0_pair", d_wave, stride, samples))) return rc;
    DZ_HIP(hipSetDevice(ctx->device));
    DzRangeScope range_scope(ctx->oflag_dev);
    return dz_launch_sinc_conv0_pair(d_wave, stride, batch, samples, d_moments, d_pair_planes, d_pair_bsum, gamma_seg,
                                     gamma_emb, d_y0_seg, d_y0_emb, g.P0, d_part_seg, d_part_emb, g.nt0,
                                     (hipStream_t)stream);
}
#endif  // DZ_EXPERIMENTS
extern "C" int dz_k_finalize_norm(dz_ctx* ctx, const float* d_partials, int batch, int ntile,
                                  int channels, int frames, const float* d_gamma,
                                  const float* d_beta, float* d_scale, float* d_shift,
                                  void* stream) {
    DZ_REQUIRE(ctx && d_partials && d_gamma && d_beta && d_scale && d_shift,
               "dz_k_finalize_norm: NULL argument");
    DZ_HIP(hipSetDevice(ctx->device));
    return dz_launch_finalize_norm(d_partials, batch, ntile, channels, frames, d_gamma, d_beta,
                                   d_scale, d_shift, (hipStream_t)stream);
}
extern "C" int dz_k_lstm(dz_ctx* ctx, const float* d_gx, const float* d_whh, float* d_hout,
                         int batch, int frames, void* stream) {
    DZ_REQUIRE(ctx && d_gx && d_whh && d_hout, "dz_k_lstm: NULL argument");
    DZ_REQUIRE(batch >= 1 && frames >= 1, "dz_k_lstm: empty input");
    DZ_HIP(hipSetDevice(ctx->device));
    return dz_launch_lstm(d_gx, d_whh, d_hout, nullptr, 0, batch, frames, 0, (hipStream_t)stream);
}
extern "C" int dz_k_lstm_planes(dz_ctx* ctx, const float* d_gx, const float* d_whh,
                                const void* d_whh_split, int variant, void* d_hsplit, long long hplane,
                                int batch, int frames, void* stream) {
    DZ_REQUIRE(ctx && d_gx && d_hsplit && (d_whh || d_whh_split), "dz_k_lstm_planes: NULL argument");
    DZ_REQUIRE(batch >= 1 && frames >= 1, "dz_k_lstm_planes: empty input");
    DZ_HIP(hipSetDevice(ctx->device));
    if (d_whh_split)
        return dz_launch_lstm_mfma(d_gx, d_whh_split, nullptr, d_hsplit, hplane, batch, frames, variant >= 3 ? 1 : 0,
                                   variant, (hipStream_t)stream);     // (variants 3 / 4 exist for unit-major gx only)
    return dz_launch_lstm(d_gx, d_whh, nullptr, d_hsplit, hplane, batch, frames, 0, (hipStream_t)stream);
}
extern "C" int dz_k_lstm_mfma(dz_ctx* ctx, const float* d_gx, const void* d_whh_split, float* d_hout,
                              int batch, int frames, int unit_major, int variant, void* stream) {
    DZ_REQUIRE(ctx && d_gx && d_whh_split && d_hout, "dz_k_lstm_mfma: NULL argument");
    DZ_REQUIRE(batch >= 1 && frames >= 1, "dz_k_lstm_mfma: empty input");
    DZ_HIP(hipSetDevice(ctx->device));
    return dz_launch_lstm_mfma(d_gx, d_whh_split, d_hout, nullptr, 0, batch, frames, unit_major,
                               variant, (hipStream_t)stream);
}
extern "C" int dz_k_stats_pool(dz_ctx* ctx, const float* d_x, int frames, int channels, int ldx,
                               const float* d_weights, int weight_frames, int rows,
                               int rows_per_x, float* d_out, int ldo, void* stream) {
    DZ_REQUIRE(ctx && d_x && d_out, "dz_k_stats_pool: NULL argument");
    DZ_REQUIRE(rows >= 1 && rows_per_x >= 1 && frames >= 2, "dz_k_stats_pool: empty input");
    DZ_HIP(hipSetDevice(ctx->device));
    DZ_REQUIRE(!d_weights || weight_frames >= 2 || weight_frames <= -2, "dz_k_stats_pool: weight_frames %d", weight_frames);
    return dz_launch_stats_pool(d_x, (long long)frames * ldx, frames, channels, ldx, d_weights,
                                d_weights ? weight_frames : frames, rows, rows_per_x, d_out, ldo,
                                (hipStream_t)stream);
}
extern "C" int dz_k_powerset(dz_ctx* ctx, const float* d_logits, int rows, int classes,
                             int speakers, float* d_out, void* stream) {
    DZ_REQUIRE(ctx && d_logits && d_out, "dz_k_powerset: NULL argument");
    DZ_HIP(hipSetDevice(ctx->device));
    return dz_launch_powerset(d_logits, rows, classes, speakers, d_out, (hipStream_t)stream);
}
