// dz_ecapa_*: launch sequence of the ECAPA-TDNN embedding (include/diart_amd.h).  Host code.
#include "dz_common.h"

#include <math.h>
#include <string.h>
#include <new>
#include <vector>

namespace {

enum { MIN_NUM_SAMPLES = 640, HOP = 160, NFFT = 400, C1 = 1024, C3 = 3072, EMB = 192, FC_SPLIT = 16, SE_SPLIT = 8,
       PLANE_SLACK = 16384 };

struct Carve {
    char* base = nullptr;
    size_t used = 0;
    template <typename T>
    T* take(size_t n) {
        const size_t bytes = (n * sizeof(T) + 255) & ~size_t(255);
        T* p = base ? reinterpret_cast<T*>(base + used) : nullptr;
        used += bytes;
        return p;
    }
};

}  // namespace

struct dz_ecapa {
    dz_ctx* ctx;
    dz_ecapa_weights w;
    int Nm, S, Tc;
    long long lstride;
    char* arena;
    int* h_pin;  // pinned host: lens[Nm] | nvalid[Nm] | nmask[Nm] | tooshort[Nm]
    // device buffers
    float *sig, *spec, *pw, *melp, *feats, *b0, *t1, *res, *t2, *cat, *mfa, *a1;
    float *smean, *sfc1, *gate, *gstat, *rb, *pooled, *parts;
    // split-f16 precision: the inputs of the wide 1 x 1 layers as kb-major f16 planes (k_gemm_pre.hip), [2][C / 32][N T][32]
    unsigned short *b0s, *ress, *cats;
    int *lens, *nvalid, *nmask, *tooshort;
    int lastN, lastT;
};

static void ecapa_carve(dz_ecapa* e, Carve& a) {
    const size_t N = e->Nm, NT = N * e->Tc;
    e->sig = a.take<float>(N * e->lstride);
    e->spec = a.take<float>(NT * 404);
    e->pw = a.take<float>(NT * 204);
    e->melp = a.take<float>(NT * 80);
    e->feats = a.take<float>(NT * 80);
    e->b0 = a.take<float>(NT * C1);
    e->t1 = a.take<float>(NT * C1);
    e->res = a.take<float>(NT * C1);
    e->t2 = a.take<float>(NT * C1);
    e->cat = a.take<float>(NT * C3);   // after the MFA convolution it is reused for the logits
    e->mfa = a.take<float>(NT * C3);
    e->a1 = a.take<float>(NT * 128);
    e->smean = a.take<float>(N * C1);
    e->sfc1 = a.take<float>(N * 128);
    e->gate = a.take<float>(N * C1);
    e->gstat = a.take<float>(N * 2 * C3);
    e->rb = a.take<float>(N * 128);
    e->pooled = a.take<float>(N * 2 * C3);
    e->parts = a.take<float>((size_t)FC_SPLIT * N * EMB);
    e->b0s = e->ress = e->cats = nullptr;
    if (e->w.mfa.wsplit) {
        // (+ PLANE_SLACK: a 128-row tile that starts inside the last rows of a plane's last k-block reads up to 127
        // rows x 64 bytes past it — zeros through the buffer bounds check when the resource ends there, but a
        // consumer that reads a COLUMN SLICE of a plane (tdnn1 of blocks 1 and 2: k-blocks [32 (i - 1), 32 i) of
        // the concatenation) has a resource that ends further on, so the bytes must exist)
        e->b0s = a.take<unsigned short>(2 * NT * C1 + PLANE_SLACK);
        e->ress = a.take<unsigned short>(2 * NT * C1 + PLANE_SLACK);
        e->cats = a.take<unsigned short>(2 * NT * C3 + PLANE_SLACK);
    }
    e->lens = a.take<int>(N);
    e->nvalid = a.take<int>(N);
    e->nmask = a.take<int>(N);
    e->tooshort = a.take<int>(N);
}

extern "C" int dz_ecapa_frames_for(int num_samples) { return num_samples > 0 ? 1 + num_samples / HOP : 0; }

extern "C" int dz_ecapa_create(dz_ctx* ctx, const dz_ecapa_weights* w, int max_rows, int num_samples,
                               dz_ecapa** out) {
    DZ_REQUIRE(ctx && w && out, "dz_ecapa_create: NULL argument");
    DZ_REQUIRE(max_rows >= 1 && num_samples >= MIN_NUM_SAMPLES, "dz_ecapa_create: max_rows %d, %d samples",
               max_rows, num_samples);
    DZ_HIP(hipSetDevice(ctx->device));
    dz_ecapa* e = new (std::nothrow) dz_ecapa;
    DZ_REQUIRE(e != nullptr, "dz_ecapa_create: out of memory");
    memset(e, 0, sizeof(*e));
    e->ctx = ctx; e->w = *w; e->Nm = max_rows; e->S = num_samples;
    e->Tc = 1 + num_samples / HOP;
    e->lstride = ((long long)num_samples + NFFT + 3) / 4 * 4;
    Carve measure;
    ecapa_carve(e, measure);
    hipError_t err = hipMalloc((void**)&e->arena, measure.used);
    if (err != hipSuccess) {
        dz_set_error("dz_ecapa_create: hipMalloc(%zu) failed: %s", measure.used, hipGetErrorString(err));
        delete e;
        return 1;
    }
    err = hipHostMalloc((void**)&e->h_pin, sizeof(int) * 4 * max_rows, hipHostMallocDefault);
    if (err != hipSuccess) {
        dz_set_error("dz_ecapa_create: hipHostMalloc failed: %s", hipGetErrorString(err));
        (void)hipFree(e->arena);
        delete e;
        return 1;
    }
    Carve real;
    real.base = e->arena;
    ecapa_carve(e, real);
    *out = e;
    return 0;
}

extern "C" int dz_ecapa_destroy(dz_ecapa* e) {
    if (e) {
        if (e->arena) (void)hipFree(e->arena);
        if (e->h_pin) (void)hipHostFree(e->h_pin);
        delete e;
    }
    return 0;
}

// one convgemm launch; X is [B][Tin][ldx] with Cin channels used, Y [B][Tin or flat][ldy]
static int gemm(int tag, int rows_n, hipStream_t st, const float* X, int ldx, long long xbs, int B, int T, int Cin, int taps,
                int dil, int pad, const dz_layer& L, const float* bias, int Kpad, int Npad, int Nstore,
                float* Y, int ldy, long long ybs, int epi, const float* X2 = nullptr,
                const float* rowbias = nullptr, int ksplit = 0, long long ysplit = 0, void* Yplanes = nullptr,
                long long yplane = 0) {
    DzConvGemm p;
    memset(&p, 0, sizeof(p));
    p.X = X; p.W = L.w; p.bias = bias ? bias : L.b; p.e0 = L.s; p.e1 = L.h; p.Y = Y;
    p.B = B; p.Tin = T; p.Tout = pad ? T : T - (taps - 1) * dil; p.Tstore = p.Tout;
    p.Cin = Cin; p.taps = taps; p.dil = dil; p.pad = pad; p.K = taps * Cin; p.Kpad = Kpad;
    p.Npad = Npad; p.Nstore = Nstore; p.ldx = ldx; p.ldy = ldy; p.xbs = xbs; p.ybs = ybs;
    p.epi = epi; p.X2 = X2; p.rowbias = rowbias; p.ksplit = ksplit; p.ysplit = ysplit;
    // every layer that comes with split-f16 planes (default precision: block 0, the wide 1x1 layers, the
    // Res2Net convolutions with their reflect padding and second input, the attention's output
    // convolution, the DFT) runs the same contraction on the f16 matrix cores (k_gemm_split.hip)
    if (L.wsplit && (epi == DZ_EPI_RELU_BN || epi == DZ_EPI_BIAS || epi == DZ_EPI_RELU_BN_TANH) && ksplit <= 1 && Cin % 8 == 0) {
        p.Wsplit = L.wsplit;
        p.Npad = (Npad + 127) / 128 * 128;       // (the DFT's planes are packed with 512 rows)
        p.Ysplit = Yplanes;                      // the next wide layer's input, written by this epilogue
        p.yplane = yplane;
        DzProfScope ps(tag, rows_n);
        return dz_launch_gemm_split(p, st);
    }
    DZ_REQUIRE(Yplanes == nullptr, "ecapa: plane output asked of a layer that is not on the split-f16 path");
    DzProfScope ps(tag, rows_n);
    return dz_launch_convgemm(p, st);
}

// a wide 1 x 1 layer with BOTH operands pre-split (k_gemm_pre.hip): rows x Cin -> rows x Npad, ReLU -> BN, f32 out.
// Xplanes = the k-block of the layer's first input column inside planes of `pcols` columns (hi | lo, xplane apart)
static int gemm_pre(int tag, int rows_n, hipStream_t st, const void* Xplanes, long long xplane, int pcols, long long rows,
                    int Cin, const dz_layer& L, int Npad, float* Y) {
    DzConvGemm p;
    memset(&p, 0, sizeof(p));
    p.Xsplit = Xplanes; p.xplane = xplane; p.ldx = pcols; p.Wsplit = L.wsplit; p.bias = L.b; p.e0 = L.s; p.e1 = L.h;
    p.Y = Y; p.B = 1; p.Tin = p.Tout = p.Tstore = (int)rows; p.Cin = Cin; p.taps = 1; p.dil = 1; p.K = p.Kpad = Cin;
    p.Npad = p.Nstore = Npad; p.ldy = Npad; p.epi = DZ_EPI_RELU_BN;
    DzProfScope ps(tag, rows_n);
    return dz_launch_gemm_pre(p, st);
}

extern "C" int dz_ecapa_forward(dz_ecapa* e, const float* d_wave, long long wave_stride,
                                const float* d_masks, int N, int mask_frames, float* d_out,
                                void* stream) {
    DZ_REQUIRE(e && d_wave && d_out, "dz_ecapa_forward: NULL argument");
    DZ_REQUIRE(N >= 1 && N <= e->Nm, "dz_ecapa_forward: %d rows outside [1, %d]", N, e->Nm);
    DZ_REQUIRE(d_masks == nullptr || mask_frames >= 1, "dz_ecapa_forward: mask_frames %d", mask_frames);
    DZ_REQUIRE(wave_stride >= 0, "dz_ecapa_forward: negative stride");
    DZ_HIP(hipSetDevice(e->ctx->device));
    DzRangeScope range_scope(e->ctx->oflag_dev);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    const dz_ecapa_weights& w = e->w;

    // ---- 1. mask -> kept samples, zero padded rows (200 leading zeros = centred STFT) ---------
    DZ_HIP(hipMemsetAsync(e->sig, 0, sizeof(float) * (size_t)N * e->lstride, st));
    { DzProfScope ps(DZ_T_ECAPA_FBANK, N);
      if ((rc = dz_launch_mask_compact(d_wave, wave_stride, e->S, d_masks, mask_frames, N, e->sig,
                                       e->lstride, e->lens, st)))
          return rc; }
    int* h_lens = e->h_pin;
    int* h_nvalid = h_lens + e->Nm;
    int* h_nmask = h_nvalid + e->Nm;
    int* h_short = h_nmask + e->Nm;
    DZ_HIP(hipMemcpyAsync(h_lens, e->lens, sizeof(int) * N, hipMemcpyDeviceToHost, st));
    DZ_HIP(hipStreamSynchronize(st));   // the batch geometry (frames) depends on the longest row
    // a row whose kept samples hold a NaN / Inf comes back as -(len + 1): it keeps its place in the batch geometry
    // (speechbrain pads and normalises by the longest row whatever its values) and its embedding is NaN — what the
    // reference computes for it, and what the split-f16 layers' clamps would otherwise turn into a finite vector
    std::vector<char> bad((size_t)N, 0);
    for (int i = 0; i < N; ++i)
        if (h_lens[i] < 0) { h_lens[i] = -h_lens[i] - 1; bad[i] = 1; }
    int lmax = 0;
    for (int i = 0; i < N; ++i) lmax = h_lens[i] > lmax ? h_lens[i] : lmax;
    e->lastN = N;
    if (lmax < MIN_NUM_SAMPLES) {       // "every signal is too short": all NaN
        for (int i = 0; i < N; ++i) h_short[i] = 1;
        DZ_HIP(hipMemcpyAsync(e->tooshort, h_short, sizeof(int) * N, hipMemcpyHostToDevice, st));
        e->lastT = 0;
        return dz_launch_nan_rows(d_out, N, EMB, e->tooshort, st);
    }
    const int T = 1 + lmax / HOP;
    e->lastT = T;
    for (int i = 0; i < N; ++i) {
        const bool too_short = h_lens[i] < MIN_NUM_SAMPLES;
        h_short[i] = too_short || bad[i];       // (only dz_launch_nan_rows reads it)
        // float32 arithmetic of torch: wav_lens / max_len, then * T
        const float rel = too_short ? 1.0f : (float)h_lens[i] / (float)lmax;
        const float v = rel * (float)T;
        int nv = (int)nearbyintf(v);            // torch.round: half to even
        nv = nv < 1 ? 1 : (nv > T ? T : nv);
        h_nvalid[i] = nv;
        int nm = (int)ceilf(v);                 // #{t : (float)t < v}
        nm = nm < 1 ? 1 : (nm > T ? T : nm);
        h_nmask[i] = nm;
    }
    DZ_HIP(hipMemcpyAsync(e->nvalid, h_nvalid, sizeof(int) * N, hipMemcpyHostToDevice, st));
    DZ_HIP(hipMemcpyAsync(e->nmask, h_nmask, sizeof(int) * N, hipMemcpyHostToDevice, st));
    DZ_HIP(hipMemcpyAsync(e->tooshort, h_short, sizeof(int) * N, hipMemcpyHostToDevice, st));
    const long long NT = (long long)N * T;

    // ---- 2. Fbank: STFT as one GEMM over overlapping rows (hop 160 < window 400) ---------------
    dz_layer dft = {w.dft, w.zeros, nullptr, nullptr, w.dft_split};
    if ((rc = gemm(DZ_T_ECAPA_FBANK, N, st, e->sig, HOP, e->lstride, N, T, NFFT, 1, 1, 0, dft, nullptr, 416, 448, 402, e->spec,
                   404, (long long)T * 404, DZ_EPI_BIAS)))
        return rc;
    { DzProfScope ps(DZ_T_ECAPA_FBANK, N); if ((rc = dz_launch_power(e->spec, 404, NT, e->pw, st))) return rc; }
    dz_layer mel = {w.mel, w.zeros, nullptr, nullptr, nullptr};
    if ((rc = gemm(DZ_T_ECAPA_FBANK, N, st, e->pw, 204, 0, 1, (int)NT, 204, 1, 1, 0, mel, nullptr, 224, 128, 80, e->melp, 80, 0,
                   DZ_EPI_BIAS)))
        return rc;
    { DzProfScope ps(DZ_T_ECAPA_FBANK, N); if ((rc = dz_launch_fbank_post(e->melp, T, N, e->nvalid, e->feats, st))) return rc; }

    // ---- 3. ECAPA-TDNN -------------------------------------------------------------------------
    // Split-f16 precision: the seven wide 1 x 1 layers (tdnn1 / tdnn2 of the three blocks, the MFA convolution: 84 % of
    // the network's MACs) run on k_gemm_pre.hip — both operands as ready f16 planes moved by LDS-DMA — so whatever
    // produces their input (block 0, the Res2Net convolutions, the squeeze-excitation's gate + residual pass) also
    // writes it as kb-major planes of N T rows; the f32 copies stay for the consumers that are not GEMMs.
    const bool pre = e->b0s != nullptr;
    const long long p1 = NT * C1, p3 = NT * C3;          // elements between the hi and lo planes
    // block 0: Conv1d(80 -> 1024, k5) -> ReLU -> BN
    if ((rc = gemm(DZ_T_ECAPA_BLOCK0, N, st, e->feats, 80, (long long)T * 80, N, T, 80, 5, 1, 2, w.block0, nullptr, 416, C1, C1,
                   e->b0, C1, (long long)T * C1, DZ_EPI_RELU_BN, nullptr, nullptr, 0, 0, pre ? e->b0s : nullptr, p1)))
        return rc;
    const int dil[3] = {2, 3, 4};
    for (int i = 0; i < 3; ++i) {
        const dz_seres2net& b = w.ser[i];
        const float* xin = i == 0 ? e->b0 : e->cat + (size_t)(i - 1) * C1;
        const int ldin = i == 0 ? C1 : C3;
        // tdnn1 (1x1): block 0 reads block0's planes, blocks 1 / 2 columns [1024 (i - 1), 1024 i) of the concatenation's
        if (pre)
            rc = i == 0 ? gemm_pre(DZ_T_ECAPA_WIDE, N, st, e->b0s, p1, C1, NT, C1, b.tdnn1, C1, e->t1)
                        : gemm_pre(DZ_T_ECAPA_WIDE, N, st, e->cats + (size_t)(i - 1) * (C1 / 32) * NT * 32, p3, C3, NT, C1, b.tdnn1,
                                   C1, e->t1);
        else
            rc = gemm(DZ_T_ECAPA_WIDE, N, st, xin, ldin, 0, 1, (int)NT, C1, 1, 1, 0, b.tdnn1, nullptr, C1, C1, C1, e->t1, C1, 0,
                      DZ_EPI_RELU_BN);
        if (rc) return rc;
        // Res2Net: y0 = x0; y1 = f1(x1); yi = fi(xi + y(i-1))
        if (pre) {      // (y0 is only read by tdnn2: planes alone)
            DzProfScope ps(DZ_T_ECAPA_RES2, N);
            if ((rc = dz_launch_se_apply_planes(e->t1, C1, nullptr, nullptr, 0, nullptr, 0, e->ress, p1, N, T, 128, st))) return rc;
        } else {
            DZ_HIP(hipMemcpy2DAsync(e->res, sizeof(float) * C1, e->t1, sizeof(float) * C1, sizeof(float) * 128,
                                    (size_t)NT, hipMemcpyDeviceToDevice, st));
        }
        for (int j = 1; j < 8; ++j) {
            const float* x2 = j >= 2 ? e->res + (j - 1) * 128 : nullptr;
            // (y7 has no f32 reader when tdnn2 takes the planes)
            if ((rc = gemm(DZ_T_ECAPA_RES2, N, st, e->t1 + j * 128, C1, (long long)T * C1, N, T, 128, 3, dil[i], dil[i], b.res[j - 1],
                           nullptr, 384, 128, 128, pre && j == 7 ? nullptr : e->res + j * 128, C1, (long long)T * C1, DZ_EPI_RELU_BN,
                           x2, nullptr, 0, 0, pre ? e->ress + (size_t)j * 4 * NT * 32 : nullptr, p1)))
                return rc;
        }
        // tdnn2 (1x1)
        if (pre)
            rc = gemm_pre(DZ_T_ECAPA_WIDE, N, st, e->ress, p1, C1, NT, C1, b.tdnn2, C1, e->t2);
        else
            rc = gemm(DZ_T_ECAPA_WIDE, N, st, e->res, C1, 0, 1, (int)NT, C1, 1, 1, 0, b.tdnn2, nullptr, C1, C1, C1, e->t2, C1, 0,
                      DZ_EPI_RELU_BN);
        if (rc) return rc;
        // squeeze-excitation + residual, written straight into its slice of the concatenation
        { DzProfScope ps(DZ_T_ECAPA_SE, N); if ((rc = dz_launch_se_mean(e->t2, T, C1, C1, N, e->nmask, e->smean, st))) return rc; }
        // squeeze (N rows x 1024 -> 128): one output tile, so the K loop is split 8 ways (a lone workgroup
        // walking 32 k-tiles took 90 us); the ReLU follows the fixed-order reduce
        if ((rc = gemm(DZ_T_ECAPA_SE, N, st, e->smean, C1, 0, 1, N, C1, 1, 1, 0, b.se1, nullptr, C1, 128, 128, e->parts, 128, 0,
                       DZ_EPI_BIAS, nullptr, nullptr, SE_SPLIT, (long long)N * 128)))
            return rc;
        { DzProfScope ps(DZ_T_ECAPA_SE, N); if ((rc = dz_launch_splitk_finish(e->parts, SE_SPLIT, (long long)N * 128, N, 128, 2, e->sfc1, st))) return rc; }
        if ((rc = gemm(DZ_T_ECAPA_SE, N, st, e->sfc1, 128, 0, 1, N, 128, 1, 1, 0, b.se2, nullptr, 128, C1, C1, e->gate, C1, 0,
                       DZ_EPI_BIAS_SIGMOID)))
            return rc;
        { DzProfScope ps(DZ_T_ECAPA_SE, N);
          if (pre)      // f32 for the next block's residual (the last block has none), planes for its tdnn1 and the MFA convolution
              rc = dz_launch_se_apply_planes(e->t2, C1, e->gate, xin, ldin, i < 2 ? e->cat + (size_t)i * C1 : nullptr, C3,
                                             e->cats + (size_t)i * (C1 / 32) * NT * 32, p3, N, T, C1, st);
          else
              rc = dz_launch_se_apply(e->t2, C1, e->gate, xin, ldin, e->cat + (size_t)i * C1, C3, N, T, C1, st);
          if (rc) return rc; }
    }
    // multi-layer feature aggregation
    if (pre)
        rc = gemm_pre(DZ_T_ECAPA_WIDE, N, st, e->cats, p3, C3, NT, C3, w.mfa, C3, e->mfa);
    else
        rc = gemm(DZ_T_ECAPA_WIDE, N, st, e->cat, C3, 0, 1, (int)NT, C3, 1, 1, 0, w.mfa, nullptr, C3, C3, C3, e->mfa, C3, 0,
                  DZ_EPI_RELU_BN);
    if (rc) return rc;
    // attentive statistics pooling with global context: W [x; mean; std] = Wx x + Wms [mean; std]
    { DzProfScope ps(DZ_T_ECAPA_ASP, N); if ((rc = dz_launch_asp_gstats(e->mfa, T, C3, N, e->nmask, e->gstat, st))) return rc; }
    dz_layer wms = {w.asp_wms, w.zeros, nullptr, nullptr};
    // (N rows x 6144 -> 128: one output tile and 192 k-tiles — 0.5 ms for a lone workgroup; split-K like fc)
    if ((rc = gemm(DZ_T_ECAPA_ASP, N, st, e->gstat, 2 * C3, 0, 1, N, 2 * C3, 1, 1, 0, wms, nullptr, 2 * C3, 128, 128, e->parts, 128, 0,
                   DZ_EPI_BIAS, nullptr, nullptr, FC_SPLIT, (long long)N * 128)))
        return rc;
    { DzProfScope ps(DZ_T_ECAPA_ASP, N); if ((rc = dz_launch_splitk_finish(e->parts, FC_SPLIT, (long long)N * 128, N, 128, 0, e->rb, st))) return rc; }
    if ((rc = gemm(DZ_T_ECAPA_ASP, N, st, e->mfa, C3, (long long)T * C3, N, T, C3, 1, 1, 0, w.asp_tdnn, nullptr, C3, 128, 128, e->a1,
                   128, (long long)T * 128, DZ_EPI_RELU_BN_TANH, nullptr, e->rb)))
        return rc;
    float* logits = e->cat;   // the concatenation is dead once the MFA layer has consumed it
    if ((rc = gemm(DZ_T_ECAPA_ASP, N, st, e->a1, 128, 0, 1, (int)NT, 128, 1, 1, 0, w.asp_conv, nullptr, 128, C3, C3, logits, C3, 0,
                   DZ_EPI_BIAS)))
        return rc;
    { DzProfScope ps(DZ_T_ECAPA_ASP, N); if ((rc = dz_launch_asp_pool(e->mfa, logits, T, C3, N, e->nmask, e->pooled, st))) return rc; }
    // asp_bn (folded) + fc, split-K with a fixed-order reduce
    const long long ysplit = (long long)N * EMB;
    if ((rc = gemm(DZ_T_ECAPA_FC, N, st, e->pooled, 2 * C3, 0, 1, N, 2 * C3, 1, 1, 0, w.fc, nullptr, 2 * C3, EMB, EMB, e->parts, EMB,
                   0, DZ_EPI_BIAS, nullptr, nullptr, FC_SPLIT, ysplit)))
        return rc;
    { DzProfScope ps(DZ_T_ECAPA_FC, N); if ((rc = dz_launch_splitk_finish(e->parts, FC_SPLIT, ysplit, N, EMB, 0, d_out, st))) return rc; }
    return dz_launch_nan_rows(d_out, N, EMB, e->tooshort, st);
}

extern "C" int dz_ecapa_peek(dz_ecapa* e, int which, const void** d_ptr, long long* count, int* frames) {
    DZ_REQUIRE(e && d_ptr && count, "dz_ecapa_peek: NULL argument");
    const long long N = e->lastN, T = e->lastT;
    if (frames) *frames = (int)T;
    switch (which) {
        case 0: *d_ptr = e->feats; *count = N * T * 80; return 0;
        case 1: *d_ptr = e->b0; *count = N * T * C1; return 0;
        case 2: *d_ptr = e->cat; *count = N * T * C3; return 0;   // holds the logits after a forward
        case 3: *d_ptr = e->mfa; *count = N * T * C3; return 0;
        case 4: *d_ptr = e->pooled; *count = N * 2 * C3; return 0;
        case 5: *d_ptr = e->lens; *count = N; return 0;
    }
    dz_set_error("dz_ecapa_peek: unknown buffer %d", which);
    return 2;
}
