// Implicit-GEMM 1-D convolution / linear layer on the exact-f32 matrix cores.
//
//   Y[b][t][n] = epi( sum_{tap,c} pro(X[b][t + tap*dil][c]) * W[n][tap*Cin + c] + bias[n] )
//
// channels-last activations, so every tap of the im2col row is a contiguous Cin-vector.
// One kernel serves SincNet conv1/conv2 (+MaxPool3 + instance-norm partials), the five
// x-vector TDNN layers (bias -> LeakyReLU -> folded BatchNorm), the LSTM input
// projections, the segmentation MLP / classifier and the embedding Linear(3000, 512)
// (third-party graphs called from /root/reference/src/diart/models.py:133 and :262;
// SURVEY.md Appendix A).
//
// Tile: 96 rows x BN cols x 32 k per step, 4 waves (2 x 2), wave tile 48 x BN/2 made of
// 16x16x4 f32 MFMA fragments.  LDS image per operand is [k/4][row ^ ((k/4)&3)][4]: a lane
// (i = l&15, q = l>>4) takes the 4 consecutive k of its quarter with ONE ds_read_b128
// (k is permuted consistently for both operands, the sum is order-free), and the
// ds_read_b128 lane groups see 16 distinct 16-B slots (row index is a bijection, the XOR
// only touches its low 2 bits) -> conflict-free reads, 2-way writes.
// Global -> LDS goes through registers and is issued one k-tile ahead of the MFMAs.
#include "dz_common.h"

namespace {

constexpr int BM = 96;
constexpr int KT = 32;

template <int BN>
struct Cfg {
    static constexpr int NT = BN / 32;           // 16-col fragments per wave
    static constexpr int A_F4 = BM * 8 / 256;    // float4 per thread per k-tile (A)
    static constexpr int B_F4 = BN * 8 / 256;    // (B)
    static constexpr int TILE = (BM + BN) * KT;  // floats per stage
    static constexpr int OLD = BN + 1;           // pooled epilogue staging row pitch
    static constexpr size_t LDS =
        sizeof(float) * (2 * TILE > BM * OLD ? 2 * TILE : BM * OLD);
};

__device__ __forceinline__ float leaky(float v) { return v > 0.f ? v : v * DZ_LEAKY_SLOPE; }

template <int BN, bool PRO, int EPI>
__global__ __launch_bounds__(256) void convgemm_kernel(DzConvGemm p) {
    using C = Cfg<BN>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    int bx, by, bz;
    dz_tile_map(p.agroup, bx, by, bz);
    const int nsplit = p.ksplit > 1 ? p.ksplit : 1;
    const int b = bz / nsplit, ks = bz - b * nsplit;
    const int t0 = bx * BM;
    const int n0 = by * BN;

    // ---- global -> register staging coordinates (fixed per thread) -----------------
    const int lrow = tid >> 3;   // 0..31 (+32 per pass)
    const int lkq = tid & 7;     // which float4 of the 32-wide k slice
    const float* Xb = p.X + (long long)b * p.xbs;
    const float* nsc = PRO ? p.nscale + (long long)b * p.nld : nullptr;
    const float* nsh = PRO ? p.nshift + (long long)b * p.nld : nullptr;
    const float* X2b = p.X2 ? p.X2 + (long long)b * p.xbs : nullptr;
    int trow[C::A_F4];
#pragma unroll
    for (int a = 0; a < C::A_F4; ++a) {
        const int t = t0 + lrow + 32 * a;
        trow[a] = t < p.Tout ? t : p.Tout - 1;
    }
    const float* Wt = p.W + (long long)(n0 + lrow) * p.Kpad + lkq * 4;

    f32x4 ra[C::A_F4], rb[C::B_F4];
    // Wide multi-tap layers (x-vector tdnn2 - 4: Cin = 512) walk K with the taps of one 32-channel block
    // consecutive — loop tile kt = (channel block kt / taps, tap kt % taps) — so that the three reads of
    // (almost) the same activation lines follow each other; in the tap-major order of W's K axis they were a
    // third of the loop apart and the L2 (16+ row tiles in flight per XCD beside the weights) had dropped
    // them in between: FETCH_SIZE 2x the algorithmic bytes.  Same products, different summation order.
    const bool tap_minor = p.taps > 1 && p.Cin % KT == 0 && p.K == p.Kpad;
    auto load_tile = [&](int kt) {
        int kw = kt * KT;                     // first column of this tile in W's K axis
        int tap = 0, c = kw + lkq * 4;
        if (tap_minor) {
            const int cblk = kt / p.taps;
            tap = kt - cblk * p.taps;
            c = cblk * KT + lkq * 4;
            kw = tap * p.Cin + cblk * KT;
        } else if (p.taps > 1) {
            tap = c / p.Cin;
            c -= tap * p.Cin;
        }
        const int k = kw + lkq * 4;
        const bool kvalid = k < p.K;
        f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (PRO && kvalid) {
            sc = *reinterpret_cast<const f32x4*>(nsc + c);
            sh = *reinterpret_cast<const f32x4*>(nsh + c);
        }
        const int toff = tap * p.dil - p.pad;
#pragma unroll
        for (int a = 0; a < C::A_F4; ++a) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (kvalid) {
                int tt = trow[a] + toff;
                if (p.pad) {  // "same" convolution with reflect padding (ECAPA TDNN blocks)
                    tt = tt < 0 ? -tt : tt;
                    tt = tt >= p.Tin ? 2 * (p.Tin - 1) - tt : tt;
                }
                const long long off = (long long)tt * p.ldx + c;
                v = *reinterpret_cast<const f32x4*>(Xb + off);
                if (X2b) {    // Res2Net: the convolution input is x_i + y_{i-1}
                    const f32x4 v2 = *reinterpret_cast<const f32x4*>(X2b + off);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += v2[e];
                }
                if (PRO) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = leaky(v[e] * sc[e] + sh[e]);
                }
            }
            ra[a] = v;
        }
#pragma unroll
        for (int a = 0; a < C::B_F4; ++a)
            rb[a] = *reinterpret_cast<const f32x4*>(Wt + (long long)(32 * a) * p.Kpad + kw);
    };
    auto store_tile = [&](int buf) {
        float* As = smem + buf * C::TILE;
        float* Bs = As + BM * KT;
        const int sw = lkq & 3;
#pragma unroll
        for (int a = 0; a < C::A_F4; ++a)
            *reinterpret_cast<f32x4*>(As + (lkq * BM + ((lrow + 32 * a) ^ sw)) * 4) = ra[a];
#pragma unroll
        for (int a = 0; a < C::B_F4; ++a)
            *reinterpret_cast<f32x4*>(Bs + (lkq * BN + ((lrow + 32 * a) ^ sw)) * 4) = rb[a];
    };

    // ---- MFMA coordinates ----------------------------------------------------------
    const int w = tid >> 6, l = tid & 63, li = l & 15, q = l >> 4;
    const int wm = w >> 1, wn = w & 1;
    f32x4 acc[3][C::NT];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk_all = p.Kpad / KT;
    const int kt0 = (int)((long long)nk_all * ks / nsplit);
    const int nk = (int)((long long)nk_all * (ks + 1) / nsplit);
    load_tile(kt0);
    store_tile(kt0 & 1);
    __syncthreads();
    for (int kt = kt0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const float* As = smem + buf * C::TILE;
        const float* Bs = As + BM * KT;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int kq = cc * 4 + q;  // (kq & 3) == q
            f32x4 af[3], bf[C::NT];
#pragma unroll
            for (int mt = 0; mt < 3; ++mt)
                af[mt] = *reinterpret_cast<const f32x4*>(
                    As + (kq * BM + ((wm * 48 + mt * 16 + li) ^ q)) * 4);
#pragma unroll
            for (int nt = 0; nt < C::NT; ++nt)
                bf[nt] = *reinterpret_cast<const f32x4*>(
                    Bs + (kq * BN + ((wn * (BN / 2) + nt * 16 + li) ^ q)) * 4);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mt = 0; mt < 3; ++mt)
#pragma unroll
                    for (int nt = 0; nt < C::NT; ++nt)
                        acc[mt][nt] = DZ_MFMA(af[mt][s], bf[nt][s], acc[mt][nt]);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ------------------------------------------------------------------
    if (EPI == DZ_EPI_POOL3) {
        // conv (+bias) -> LDS tile -> MaxPool1d(3,3) over time -> pooled rows + stats partials
        float* out_s = smem;  // all waves are past the last barrier: tiles are dead
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) {
            const int n = wn * (BN / 2) + nt * 16 + li;
            const float bv = p.bias[n0 + n];
#pragma unroll
            for (int mt = 0; mt < 3; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    out_s[(wm * 48 + mt * 16 + 4 * q + r) * C::OLD + n] = acc[mt][nt][r] + bv;
        }
        __syncthreads();
        const int p0 = t0 / 3;
        float* Yb = p.Y + (long long)b * p.ybs;
        for (int idx = tid; idx < 32 * BN; idx += 256) {
            const int pr = idx / BN, n = idx - pr * BN;
            const float* o = out_s + (3 * pr) * C::OLD + n;
            const float v = fmaxf(fmaxf(o[0], o[C::OLD]), o[2 * C::OLD]);
            const bool valid = (p0 + pr) < p.Tstore;
            if (valid && (n0 + n) < p.Nstore) Yb[(long long)(p0 + pr) * p.ldy + n0 + n] = v;
            out_s[(3 * pr) * C::OLD + n] = valid ? v : 0.f;
        }
        __syncthreads();
        if (tid < BN) {
            float s = 0.f, ss = 0.f;
            for (int pr = 0; pr < 32; ++pr) {
                const float v = out_s[(3 * pr) * C::OLD + tid];
                s += v;
                ss += v * v;
            }
            float* pp = p.partials +
                        (((long long)b * gridDim.x + bx) * p.Npad + n0 + tid) * 2;
            pp[0] = s;
            pp[1] = ss;
        }
        return;
    }

    float* Yb = p.Y + (long long)b * p.ybs + (long long)ks * p.ysplit;
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt) {
        const int n = n0 + wn * (BN / 2) + nt * 16 + li;
        float bv = ks == 0 ? p.bias[n] : 0.f;
        if (p.rowbias) bv += p.rowbias[(long long)b * p.Npad + n];
        float e0 = 1.f, e1 = 0.f;
        if (EPI == DZ_EPI_TDNN || EPI == DZ_EPI_RELU_BN || EPI == DZ_EPI_RELU_BN_TANH) {
            e0 = p.e0[n];
            e1 = p.e1[n];
        }
        if (n < p.Nstore) {
#pragma unroll
            for (int mt = 0; mt < 3; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = t0 + wm * 48 + mt * 16 + 4 * q + r;
                    if (t < p.Tout) {
                        float v = acc[mt][nt][r] + bv;
                        if (EPI == DZ_EPI_BIAS_LEAKY) v = leaky(v);
                        if (EPI == DZ_EPI_BIAS_SIGMOID) v = 1.f / (1.f + expf(-v));
                        if (EPI == DZ_EPI_TDNN) v = leaky(v) * e0 + e1;
                        if (EPI == DZ_EPI_BIAS_RELU) v = fmaxf(v, 0.f);
                        if (EPI == DZ_EPI_RELU_BN) v = fmaxf(v, 0.f) * e0 + e1;
                        if (EPI == DZ_EPI_RELU_BN_TANH) v = tanhf(fmaxf(v, 0.f) * e0 + e1);
                        Yb[(long long)t * p.ldy + n] = v;
                    }
                }
        }
    }
}

template <int BN, bool PRO, int EPI>
int launch(const DzConvGemm& p, hipStream_t st) {
    using C = Cfg<BN>;
    static DzAttrOnce attr_once;
    DZ_HIP(attr_once.raise((const void*)convgemm_kernel<BN, PRO, EPI>, (int)C::LDS));
    dim3 grid((p.Tout + BM - 1) / BM, p.Npad / BN, p.B * (p.ksplit > 1 ? p.ksplit : 1));
    DZ_LAUNCH((convgemm_kernel<BN, PRO, EPI>), grid, dim3(256), C::LDS, st, p);
    DZ_HIP(hipGetLastError());
    return 0;
}

}  // namespace

int dz_convgemm_ntile(int Tout) { return (Tout + BM - 1) / BM; }

int dz_launch_convgemm(const DzConvGemm& p, hipStream_t st) {
    if (dz_gemm_f32_ok(p)) return dz_launch_gemm_f32(p, st);      // wide layers without a prologue: k_gemm_f32.hip
    DZ_REQUIRE(p.Kpad % KT == 0 && p.Cin % 4 == 0 && p.ldx % 4 == 0, "convgemm: bad K/Cin/ldx");
    DZ_REQUIRE(p.K <= p.Kpad && p.K == p.taps * p.Cin, "convgemm: K mismatch");
    DZ_REQUIRE(p.pad >= 0 && p.Tout > 0 &&
                   p.Tout == (p.pad ? p.Tin : p.Tin - (p.taps - 1) * p.dil),
               "convgemm: Tout mismatch");
    DZ_REQUIRE(p.pad == 0 || (2 * p.pad == (p.taps - 1) * p.dil && p.pad < p.Tin),
               "convgemm: reflect 'same' padding needs 2*pad == (taps-1)*dil and pad < Tin");
    DZ_REQUIRE(p.ksplit <= 1 || (p.epi == DZ_EPI_BIAS && p.ksplit <= p.Kpad / KT),
               "convgemm: split-K needs the plain bias epilogue and ksplit <= k-tiles");
    const bool wide = (p.Npad % 128 == 0);
    DZ_REQUIRE(p.Npad % 64 == 0, "convgemm: Npad must be a multiple of 64");
    const bool pro = p.norm_on_load != 0;
#define DZ_CG(BN, PRO, EPI) return launch<BN, PRO, EPI>(p, st)
    switch (p.epi) {
        case DZ_EPI_POOL3:
            DZ_REQUIRE(pro && !wide, "convgemm: POOL3 is instantiated for norm-on-load, BN=64");
            DZ_CG(64, true, DZ_EPI_POOL3);
        case DZ_EPI_TDNN:
            DZ_REQUIRE(wide, "convgemm: TDNN needs Npad %% 128 == 0");
            if (pro) DZ_CG(128, true, DZ_EPI_TDNN);
            DZ_CG(128, false, DZ_EPI_TDNN);
        case DZ_EPI_BIAS:
            if (wide) {
                if (pro) DZ_CG(128, true, DZ_EPI_BIAS);
                DZ_CG(128, false, DZ_EPI_BIAS);
            }
            DZ_REQUIRE(!pro, "convgemm: BIAS/BN=64 has no norm-on-load instance");
            DZ_CG(64, false, DZ_EPI_BIAS);
        case DZ_EPI_BIAS_LEAKY:
            DZ_REQUIRE(wide && !pro, "convgemm: BIAS_LEAKY is BN=128, no norm-on-load");
            DZ_CG(128, false, DZ_EPI_BIAS_LEAKY);
        case DZ_EPI_BIAS_SIGMOID:
            DZ_REQUIRE(!pro, "convgemm: BIAS_SIGMOID has no norm-on-load instance");
            if (wide) DZ_CG(128, false, DZ_EPI_BIAS_SIGMOID);
            DZ_CG(64, false, DZ_EPI_BIAS_SIGMOID);
        case DZ_EPI_BIAS_RELU:
            DZ_REQUIRE(wide && !pro, "convgemm: BIAS_RELU is BN=128, no norm-on-load");
            DZ_CG(128, false, DZ_EPI_BIAS_RELU);
        case DZ_EPI_RELU_BN:
            DZ_REQUIRE(wide && !pro, "convgemm: RELU_BN is BN=128, no norm-on-load");
            DZ_CG(128, false, DZ_EPI_RELU_BN);
        case DZ_EPI_RELU_BN_TANH:
            DZ_REQUIRE(wide && !pro, "convgemm: RELU_BN_TANH is BN=128, no norm-on-load");
            DZ_CG(128, false, DZ_EPI_RELU_BN_TANH);
    }
#undef DZ_CG
    dz_set_error("convgemm: unknown epilogue %d", p.epi);
    return 2;
}
