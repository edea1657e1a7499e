// Implicit-GEMM 1-D convolution / linear layer on the f16 matrix cores with split operands.
//
//   Y[b][t][n] = epi( sum_{tap,c} pro(X[b][t + tap*dil][c]) * W[n][tap*Cin + c] + bias[n] )
//
// Same contraction, descriptor and epilogues as k_convgemm.hip (exact-f32 MFMA, 157 TFLOP/s peak)
// on the 16x faster half-precision matrix pipe, without giving up f32-grade operands: every f32
// operand is split into two f16 numbers
//     x = hi + lo * 2^-11,   hi = f16(x),   lo = f16((x - hi) * 2^11)       (22 mantissa bits kept;
//                                                        x - hi is exact, the scale is a power of 2)
// and the product is accumulated in f32 from three f16 MFMAs into two accumulators
//     main  += hi_x * hi_w
//     cross += hi_x * lo_w + lo_x * hi_w        (lo*lo ~ 2^-22 relative, dropped)
//     x*w   ~= main + cross * 2^-11
// on v_mfma_f32_32x32x16_f16 (2.5 PFLOP/s dense -> 833 TFLOP/s of split products).  The scaled low
// part keeps lo a NORMAL f16 for every |x| >= 2^-14 (an unscaled lo would be subnormal below
// |x| = 0.125).  Operand error <= 2^-22 relative: the same order as f32's own rounding (2^-24) and
// far below the summation-order noise of any f32 GEMM; inputs are clamped to the f16 range
// (+-65504; the networks' GEMM inputs are normalised activations, |x| = O(1..10)).
//
// Tile: (32 WM) rows x (64 NB) cols x 32 k per step, WM x 2 waves, wave tile 32 x 32 NB; the wide
// layers use WM = 4, NB = 2 (128 x 128, 8 waves: per k-step of 16 a wave reads 2 A and 4 B
// fragments (hi / lo, one ds_read_b128 each) and issues 6 MFMAs), the SincNet convolutions
// (60 -> 64 output channels, MaxPool1d(3) fused) use WM = 3, NB = 1 (96 x 64, 6 waves).  Two workgroups per CU = 4 waves per SIMD: the MFMAs of three waves cover
// the global-load latency of the fourth (with 4 waves per workgroup the kernel sat in s_waitcnt
// for 39 % of its wave cycles; rocprofv3 SQ_WAIT_ANY).  LDS holds four f16 planes per stage (A hi,
// A lo, B hi, B lo; [row][32 k] = 64 B rows, 16-byte chunks XOR-swizzled with (row >> 2) & 3 so the
// 16 lanes of a ds_read_b128 phase hit 16 distinct 16-byte slots: SQ_LDS_BANK_CONFLICT = 0),
// double buffered: 64 KiB.  Activations are f32 in HBM: they are split on the way into LDS
// (v_cvt_pk_f16_f32, ~4 VALU ops per element, hidden under the MFMAs of the other waves); weights
// are split once on the host (weights.py split_f16) and arrive as two f16 planes.
// The (lane -> k) assignment inside a fragment is the same for A and B (8 consecutive k per lane,
// lanes 32..63 take the upper 8 of a 16-wide k-step), so the contraction is correct for any
// hardware k-ordering; the C/D map is cdna_hip_programming.md "Fragment layout".
#include "dz_common.h"
#include <stdlib.h>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int KT = 32;

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ float leaky(float v) { return v > 0.f ? v : v * DZ_LEAKY_SLOPE; }
__device__ __forceinline__ int chunk_off(int row, int cidx) {
    return row * 64 + ((cidx ^ ((row >> 2) & 3)) << 4);
}
constexpr float LO_SCALE = 2048.f, LO_UNSCALE = 1.f / 2048.f;
constexpr float F16_MAX = 65504.f;
// 8 floats -> 8 f16 hi + 8 f16 lo (lo scaled by 2^11)
__device__ __forceinline__ void split8(const float* v, u32x4& hi, u32x4& lo, float& amax) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        amax = fmaxf(amax, fmaxf(fabsf(v[2 * e]), fabsf(v[2 * e + 1])));
        const f32x2 x = {__builtin_amdgcn_fmed3f(v[2 * e], -F16_MAX, F16_MAX),
                         __builtin_amdgcn_fmed3f(v[2 * e + 1], -F16_MAX, F16_MAX)};
        const f16x2 h = __builtin_convertvector(x, f16x2);
        const f32x2 r = (x - __builtin_convertvector(h, f32x2)) * LO_SCALE;
        hi[e] = __builtin_bit_cast(unsigned, h);
        lo[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
    }
}

template <int WM, int NB>
struct Cfg {
    static constexpr int BM = 32 * WM, BN = 64 * NB, T = 128 * WM;
    static constexpr int APLANE = BM * 64, BPLANE = BN * 64;      // bytes of one f16 plane
    static constexpr int STAGE = 2 * APLANE + 2 * BPLANE;         // A hi | A lo | B hi | B lo
    static constexpr int OLD = BN + 1;                            // pooled epilogue staging pitch
    static constexpr size_t LDS =
        2 * STAGE > 4 * BM * OLD ? 2 * STAGE : 4 * BM * OLD;
    static_assert(BN * 4 <= T, "one B chunk per thread at most");
};

template <int WM, int NB, bool PRO, int EPI>
__global__ __launch_bounds__(128 * WM) void gemm_split_kernel(DzConvGemm p) {
    using C = Cfg<WM, NB>;
    constexpr int BM = C::BM, BN = C::BN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    int bx, by, b;
    dz_tile_map(p.agroup, bx, by, b);
    const int t0 = bx * BM, n0 = by * BN;

    // ---- staging coordinates: thread -> row tid >> 2, 8-wide k chunk tid & 3 ------------------
    const int crow = tid >> 2, cidx = tid & 3;
    const bool has_b = tid < BN * 4;
    const float* Xb = p.X + (long long)b * p.xbs;
    const float* X2b = !PRO && p.X2 ? p.X2 + (long long)b * p.xbs : nullptr;
    const float* nsc = PRO ? p.nscale + (long long)b * p.nld : nullptr;
    const float* nsh = PRO ? p.nshift + (long long)b * p.nld : nullptr;
    __shared__ __attribute__((aligned(16))) float nrm_s[PRO ? 256 : 4];   // scale[nld] | shift[nld], nld <= 128
    if (PRO && p.npart) {   // derive them from the producer's tile partials: no finalize launch
        dz_norm_from_partials(p.npart, b, p.npart_tiles, p.nld, p.npart_T, p.ngamma, p.nbeta, nrm_s, tid, C::T,
                              reinterpret_cast<double*>(smem));   // (the stages are not in use yet)
        nsc = nrm_s;
        nsh = nrm_s + p.nld;
        __syncthreads();
    }
    const int trow = (t0 + crow) < p.Tout ? (t0 + crow) : p.Tout - 1;
    const unsigned short* Whi = reinterpret_cast<const unsigned short*>(p.Wsplit);
    const unsigned short* Wlo = Whi + (long long)p.Npad * p.Kpad;
    const long long wofs = (long long)(n0 + (has_b ? crow : 0)) * p.Kpad + cidx * 8;

    float amax = 0.f;          // largest |operand| this thread split (reported beyond +-65504)
    struct Regs {
        f32x4 ra[2];
        u32x4 rbh, rbl;
    };
    Regs R0;
    auto load_tile = [&](Regs& R, int kt) {
        const int k = kt * KT + cidx * 8;
        const bool kvalid = k < p.K;
        int tap = 0, c = k;
        if (p.taps > 1) {
            tap = k / p.Cin;
            c = k - tap * p.Cin;
        }
        const int toff = tap * p.dil - (PRO ? 0 : p.pad);
        {
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
            if (kvalid) {
                int tt = trow + toff;
                if (!PRO && p.pad) {   // "same" convolution with reflect padding (ECAPA TDNN blocks; k_convgemm.hip)
                    tt = tt < 0 ? -tt : tt;
                    tt = tt >= p.Tin ? 2 * (p.Tin - 1) - tt : tt;
                }
                const long long xo = (long long)tt * p.ldx + c;
                v0 = *reinterpret_cast<const f32x4*>(Xb + xo);
                v1 = *reinterpret_cast<const f32x4*>(Xb + xo + 4);
                if (!PRO && X2b) {     // Res2Net: the convolution input is x_i + y_{i-1}
                    const f32x4 u0 = *reinterpret_cast<const f32x4*>(X2b + xo);
                    const f32x4 u1 = *reinterpret_cast<const f32x4*>(X2b + xo + 4);
                    v0 += u0;
                    v1 += u1;
                }
                if (PRO) {
                    const f32x4 s0 = *reinterpret_cast<const f32x4*>(nsc + c);
                    const f32x4 s1 = *reinterpret_cast<const f32x4*>(nsc + c + 4);
                    const f32x4 h0 = *reinterpret_cast<const f32x4*>(nsh + c);
                    const f32x4 h1 = *reinterpret_cast<const f32x4*>(nsh + c + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v0[e] = leaky(v0[e] * s0[e] + h0[e]);
                        v1[e] = leaky(v1[e] * s1[e] + h1[e]);
                    }
                }
            }
            R.ra[0] = v0;
            R.ra[1] = v1;
        }
        if (has_b) {
            const long long o = wofs + kt * KT;
            R.rbh = *reinterpret_cast<const u32x4*>(Whi + o);
            R.rbl = *reinterpret_cast<const u32x4*>(Wlo + o);
        }
    };
    auto store_tile = [&](const Regs& R, int buf) {
        char* st = smem + buf * C::STAGE;
        const int off = chunk_off(crow, cidx);
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = R.ra[0][e];
            v[4 + e] = R.ra[1][e];
        }
        u32x4 hi, lo;
        split8(v, hi, lo, amax);
        *reinterpret_cast<u32x4*>(st + off) = hi;
        *reinterpret_cast<u32x4*>(st + C::APLANE + off) = lo;
        if (has_b) {
            *reinterpret_cast<u32x4*>(st + 2 * C::APLANE + off) = R.rbh;
            *reinterpret_cast<u32x4*>(st + 2 * C::APLANE + C::BPLANE + off) = R.rbl;
        }
    };

    // ---- MFMA coordinates: WM x 2 waves, wave tile 32 x (32 NB) ---------------------------------
    const int w = tid >> 6, l = tid & 63, li = l & 31, g = l >> 5;
    const int wm = w >> 1, wn = w & 1;
    f32x16 accm[NB], accx[NB];                 // hi*hi | cross terms (scaled by 2^11)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) accm[nb][r] = accx[nb][r] = 0.f;

    const int nk = p.Kpad / KT;
    auto compute = [&](int buf) {
        const char* st = smem + buf * C::STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 ah, al, bh[NB], bl[NB];
            {
                const int off = chunk_off(wm * 32 + li, 2 * ks + g);
                ah = *reinterpret_cast<const f16x8*>(st + off);
                al = *reinterpret_cast<const f16x8*>(st + C::APLANE + off);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int off = chunk_off(wn * 32 * NB + nb * 32 + li, 2 * ks + g);
                bh[nb] = *reinterpret_cast<const f16x8*>(st + 2 * C::APLANE + off);
                bl[nb] = *reinterpret_cast<const f16x8*>(st + 2 * C::APLANE + C::BPLANE + off);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                accx[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[nb], accx[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                accm[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[nb], accm[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                accx[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[nb], accx[nb], 0, 0, 0);
        }
    };
    // one k-tile per step: the global loads of tile kt + 1 are issued before the MFMAs of tile kt
    // and parked in LDS after them.  The step barrier only has to publish LDS (s_waitcnt lgkmcnt +
    // s_barrier): __syncthreads() would also drain vmcnt.  Measured and rejected: global loads two
    // tiles ahead through a second register set (spills inside a 128-register budget, 140
    // registers halve the occupancy) and register-double-buffered fragments on a 96 x 128 / 6-wave
    // tile (160 registers, 3 waves per SIMD: 20-70 % slower per layer).
    load_tile(R0, 0);
    store_tile(R0, 0);
    lds_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(R0, kt + 1);
        compute(buf);
        if (kt + 1 < nk) store_tile(R0, buf ^ 1);
        lds_barrier();
    }

    dz_flag_range(p.oflag, amax);
    amax = 0.f;
    // ---- epilogue: C/D map col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -----
    if (EPI == DZ_EPI_POOL3) {
        // conv (+bias) -> LDS tile -> MaxPool1d(3,3) over time -> pooled rows + stats partials
        float* out_s = reinterpret_cast<float*>(smem);  // all waves are past the last barrier
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int n = wn * 32 * NB + nb * 32 + li;
            const float bv = p.bias[n0 + n];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                out_s[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * g) * C::OLD + n] =
                    (accm[nb][r] + accx[nb][r] * LO_UNSCALE) + bv;
        }
        __syncthreads();
        constexpr int PR = BM / 3;                     // pooled rows per tile
        const int p0 = t0 / 3;
        float* Yb = p.Y + (long long)b * p.ybs;
        for (int idx = tid; idx < PR * BN; idx += C::T) {
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
            for (int pr = 0; pr < PR; ++pr) {
                const float v = out_s[(3 * pr) * C::OLD + tid];
                s += v;
                ss += v * v;
            }
            float* pp = p.partials + (((long long)b * gridDim.x + bx) * p.Npad + n0 + tid) * 2;
            pp[0] = s;
            pp[1] = ss;
        }
        return;
    }
    float* Yb = p.Y ? p.Y + (long long)b * p.ybs : nullptr;
    // optional (hi, lo) f16 plane output for a k_gemm_pre.hip consumer (dz_store_split)
    // in the kb-major order of dz_kb(): batch item b owns rows b * (ybs / ldy) .. of the planes
    unsigned short* Ypl = p.Ysplit ? dz_split_base(p.Ysplit, p.yplane, li & 1) : nullptr;
    const long long yrows = Ypl ? p.yplane / p.ldy : 0, yrow0 = Ypl ? (long long)b * (p.ybs / p.ldy) : 0;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = n0 + wn * 32 * NB + nb * 32 + li;
        float bv = p.bias[n];
        if (p.rowbias) bv += p.rowbias[(long long)b * p.Npad + n];       // per-batch-item bias (ECAPA's attention TDNN)
        float e0 = 1.f, e1 = 0.f;
        if (EPI == DZ_EPI_TDNN || EPI == DZ_EPI_RELU_BN || EPI == DZ_EPI_RELU_BN_TANH) {
            e0 = p.e0[n];
            e1 = p.e1[n];
        }
        const bool nok = n < p.Nstore;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = t0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
            const bool ok = t < p.Tout;
            float v = (accm[nb][r] + accx[nb][r] * LO_UNSCALE) + bv;
            if (EPI == DZ_EPI_BIAS_LEAKY) v = leaky(v);
            if (EPI == DZ_EPI_TDNN) v = leaky(v) * e0 + e1;
            if (EPI == DZ_EPI_RELU_BN) v = fmaxf(v, 0.f) * e0 + e1;
            if (EPI == DZ_EPI_RELU_BN_TANH) v = tanhf(fmaxf(v, 0.f) * e0 + e1);     // (k_convgemm.hip's expression)
            if (Yb && ok && nok) Yb[(long long)t * p.ldy + n] = v;
            if (Ypl) dz_store_split(Ypl, dz_kb(yrow0 + t, n, yrows), nok ? v : 0.f, ok, li & 1, amax);
        }
    }
    dz_flag_range(p.oflag, amax);
}

template <int WM, int NB, bool PRO, int EPI>
int launch(const DzConvGemm& p, hipStream_t st) {
    using C = Cfg<WM, NB>;
    static DzAttrOnce attr_once;
    DZ_HIP(attr_once.raise((const void*)gemm_split_kernel<WM, NB, PRO, EPI>, (int)C::LDS));
    dim3 grid((p.Tout + C::BM - 1) / C::BM, p.Npad / C::BN, p.B);
    DZ_LAUNCH((gemm_split_kernel<WM, NB, PRO, EPI>), grid, dim3(C::T), C::LDS, st, p);
    DZ_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// InstanceNorm + LeakyReLU + split of the last SincNet stage's output, ONCE (round 6): y2 [B][P][64] f32 (raw pooled
// values) + the producer's tile partials -> the two f16 planes (hi, lo * 2^11) of [B * P rows][64] in the kb-major
// order k_gemm_pre.hip reads.  The two layers that consume y2 — the first LSTM projection and tdnn1 — ran on
// gemm_split_kernel with this arithmetic in their operand-staging prologue (norm-on-load): per 32-wide k-tile every
// workgroup normalised and split its rows again (tdnn1: each row five times, once per tap; proj0: once per 128-column
// tile, eight times), on the vector units that the MFMAs of the same SIMD then wait for — 52 / 36 us
// for 5.7 / 2.3 GFLOP.  After this 4.8 MB pass both run on the pre-split GEMM like every other wide layer.
// grid (ceil(P / 32), B), 256 threads: 32 rows x 64 channels per workgroup, 8 channels per thread.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void norm_split_kernel(const float* __restrict__ y, const float* __restrict__ part,
                                                         int ntile, int P, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, unsigned short* __restrict__ planes,
                                                         long long plane, int* oflag) {
    __shared__ __attribute__((aligned(16))) float nrm_s[128];          // scale[64] | shift[64]
    __shared__ double scratch[2 * 4 * 64];                              // dz_norm_from_partials: 2 G C doubles, G = 256 / 64
    const int tid = threadIdx.x, b = blockIdx.y;
    dz_norm_from_partials(part, b, ntile, 64, P, gamma, beta, nrm_s, tid, 256, scratch);
    __syncthreads();
    const int row = blockIdx.x * 32 + (tid >> 3), c = (tid & 7) * 8;
    if (row >= P) return;
    const long long R = plane >> 6, r = (long long)b * P + row;         // rows of a plane, this row
    const float* x = y + r * 64 + c;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(x), v1 = *reinterpret_cast<const f32x4*>(x + 4);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(nrm_s + c), s1 = *reinterpret_cast<const f32x4*>(nrm_s + c + 4);
    const f32x4 h0 = *reinterpret_cast<const f32x4*>(nrm_s + 64 + c), h1 = *reinterpret_cast<const f32x4*>(nrm_s + 64 + c + 4);
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[e] = leaky(v0[e] * s0[e] + h0[e]);
        v[4 + e] = leaky(v1[e] * s1[e] + h1[e]);
    }
    float amax = 0.f;
    u32x4 hi, lo;
    split8(v, hi, lo, amax);
    const long long idx = dz_kb(r, c, R);                               // eight columns of one k-block row: 16 bytes per plane
    *reinterpret_cast<u32x4*>(planes + idx) = hi;
    *reinterpret_cast<u32x4*>(planes + plane + idx) = lo;
    dz_flag_range(oflag, amax);
}

// The same pass with f32 output (round 6, exact-f32 path): y2 normalised once, [B * P rows][64] f32, so that the first
// LSTM projection and tdnn1 run as flattened GEMMs without a prologue on k_gemm_f32.hip (they ran per chunk on the
// round-1 kernel with the norm on load: 61 + 96 us a step) and the third finalize_norm launch of each SincNet goes.
__global__ __launch_bounds__(256) void norm_f32_kernel(const float* __restrict__ y, const float* __restrict__ part,
                                                       int ntile, int P, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float nrm_s[128];          // scale[64] | shift[64]
    __shared__ double scratch[2 * 4 * 64];
    const int tid = threadIdx.x, b = blockIdx.y;
    dz_norm_from_partials(part, b, ntile, 64, P, gamma, beta, nrm_s, tid, 256, scratch);
    __syncthreads();
    const int row = blockIdx.x * 32 + (tid >> 3), c = (tid & 7) * 8;
    if (row >= P) return;
    const long long r = (long long)b * P + row;
    const float* x = y + r * 64 + c;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(x), v1 = *reinterpret_cast<const f32x4*>(x + 4);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(nrm_s + c), s1 = *reinterpret_cast<const f32x4*>(nrm_s + c + 4);
    const f32x4 h0 = *reinterpret_cast<const f32x4*>(nrm_s + 64 + c), h1 = *reinterpret_cast<const f32x4*>(nrm_s + 64 + c + 4);
    f32x4 o0, o1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o0[e] = leaky(v0[e] * s0[e] + h0[e]);
        o1[e] = leaky(v1[e] * s1[e] + h1[e]);
    }
    *reinterpret_cast<f32x4*>(out + r * 64 + c) = o0;
    *reinterpret_cast<f32x4*>(out + r * 64 + c + 4) = o1;
}

}  // namespace

int dz_launch_gemm_split(const DzConvGemm& p_in, hipStream_t st) {
    DzConvGemm p = p_in;
    if (!p.oflag) p.oflag = dz_cur_oflag;
    DZ_REQUIRE(p.Wsplit != nullptr, "gemm_split: Wsplit (f16 hi/lo planes of W) is NULL");
    DZ_REQUIRE(p.Kpad % KT == 0 && p.Cin % 8 == 0 && p.ldx % 4 == 0 && p.K % 8 == 0,
               "gemm_split: bad K/Cin/ldx (Cin and K must be multiples of 8)");
    DZ_REQUIRE(p.K <= p.Kpad && p.K == p.taps * p.Cin, "gemm_split: K mismatch");
    DZ_REQUIRE(p.ksplit <= 1, "gemm_split: split-K is an f32-path feature");
    DZ_REQUIRE(p.rowbias == nullptr || p.epi == DZ_EPI_RELU_BN_TANH || p.epi == DZ_EPI_RELU_BN || p.epi == DZ_EPI_BIAS,
               "gemm_split: a row bias is built with the BIAS / RELU_BN / RELU_BN_TANH epilogues");
    DZ_REQUIRE((p.pad == 0 && p.X2 == nullptr) || !p.norm_on_load,
               "gemm_split: padding / a second input are not built together with norm-on-load");
    DZ_REQUIRE(p.pad >= 0 && p.Tout > 0 && p.Tout == (p.pad ? p.Tin : p.Tin - (p.taps - 1) * p.dil),
               "gemm_split: Tout mismatch");
    DZ_REQUIRE(p.pad == 0 || (2 * p.pad == (p.taps - 1) * p.dil && p.pad < p.Tin),
               "gemm_split: reflect 'same' padding needs 2*pad == (taps-1)*dil and pad < Tin");
    DZ_REQUIRE(p.Y != nullptr || p.Ysplit != nullptr, "gemm_split: no output");
    DZ_REQUIRE(!p.norm_on_load || (p.nscale && p.nshift) ||
                   (p.npart && p.ngamma && p.nbeta && p.npart_tiles > 0 && p.npart_T > 0 && p.nld <= 128),
               "gemm_split: norm-on-load needs nscale / nshift or the producer's partials + affine (nld <= 128)");
    DZ_REQUIRE(p.Ysplit == nullptr || (p.ldy > 0 && p.ldy % 32 == 0 && p.yplane % p.ldy == 0 && p.ybs % p.ldy == 0),
               "gemm_split: kb-major plane output needs ldy %% 32 == 0 and yplane / ybs multiples of ldy");
    DZ_REQUIRE(p.Ysplit == nullptr || (p.epi != DZ_EPI_POOL3 && p.ldy % 2 == 0 && p.yplane % 2 == 0 &&
                                       p.ybs % 2 == 0 && p.Npad <= p.ldy),
               "gemm_split: plane output needs even ldy / yplane / ybs, Npad <= ldy and no pooling");
    const bool pro = p.norm_on_load != 0;
#define DZ_SP(WM, NB, PRO, EPI) return launch<WM, NB, PRO, EPI>(p, st)
    if (p.epi == DZ_EPI_POOL3) {
        // the pooled tile is 96 conv rows = 32 pooled rows, like the f32 kernel (same partials grid)
        DZ_REQUIRE(pro && p.Npad == 64, "gemm_split: POOL3 is built for norm-on-load, Npad = 64");
        DZ_SP(3, 1, true, DZ_EPI_POOL3);
    }
    DZ_REQUIRE(p.Npad % 128 == 0, "gemm_split: Npad must be a multiple of 128");
    // DZ_SPLIT_WM=2: the two norm-on-load launches of the default path (LSTM projection 0, tdnn1) as
    // 64 x 64 tiles with 4 waves (54 VGPRs, one wave per SIMD), whose workgroups fit on CUs that
    // already hold a recurrence workgroup or two k_gemm_pre.hip workgroups.  Their launches get
    // shorter in the pipeline (projection 0: 276 -> 164 us, tdnn1 219 -> 161 us; alone 28 -> 31.5 and
    // 46 -> 53 us), the step time does not change beyond noise (4 same-visit pairs: 2 wins, 2 losses),
    // so the 128 x 128 / 8-wave tiles stay the default.
    static const bool small_wg = [] {
        const char* e = dz_exp_env("DZ_SPLIT_WM");
        return e && e[0] == '2';
    }();
    switch (p.epi) {
        case DZ_EPI_TDNN:
            if (pro && small_wg) DZ_SP(2, 1, true, DZ_EPI_TDNN);
            if (pro) DZ_SP(4, 2, true, DZ_EPI_TDNN);
            DZ_SP(4, 2, false, DZ_EPI_TDNN);
        case DZ_EPI_BIAS:
            if (pro && small_wg) DZ_SP(2, 1, true, DZ_EPI_BIAS);
            if (pro) DZ_SP(4, 2, true, DZ_EPI_BIAS);
            DZ_SP(4, 2, false, DZ_EPI_BIAS);
        case DZ_EPI_BIAS_LEAKY:
            DZ_REQUIRE(!pro, "gemm_split: BIAS_LEAKY has no norm-on-load instance");
            DZ_SP(4, 2, false, DZ_EPI_BIAS_LEAKY);
        case DZ_EPI_RELU_BN:   // ECAPA-TDNN's 1x1 layers: conv -> ReLU -> folded BatchNorm
            DZ_REQUIRE(!pro, "gemm_split: RELU_BN has no norm-on-load instance");
            DZ_SP(4, 2, false, DZ_EPI_RELU_BN);
        case DZ_EPI_RELU_BN_TANH:   // ... and the attention TDNN of its pooling: -> tanh
            DZ_REQUIRE(!pro, "gemm_split: RELU_BN_TANH has no norm-on-load instance");
            DZ_SP(4, 2, false, DZ_EPI_RELU_BN_TANH);
    }
#undef DZ_SP
    dz_set_error("gemm_split: epilogue %d is not built on the split-f16 path", p.epi);
    return 2;
}

// y2 [B][P][64] (raw) + part [B][ntile][64][2] + InstanceNorm affine -> planes [2][B * P rows][64] kb-major (see norm_split_kernel)
int dz_launch_norm_split(const float* y, const float* part, int ntile, int P, const float* gamma, const float* beta,
                         void* planes, long long plane, int B, hipStream_t st) {
    DZ_REQUIRE(y && part && gamma && beta && planes, "norm_split: NULL argument");
    DZ_REQUIRE(B >= 1 && P >= 1 && ntile >= 1 && plane >= (long long)B * P * 64 && plane % 64 == 0,
               "norm_split: bad geometry (B %d, P %d, plane %lld)", B, P, plane);
    DZ_LAUNCH(norm_split_kernel, dim3((P + 31) / 32, B), dim3(256), 0, st, y, part, ntile, P, gamma, beta,
              reinterpret_cast<unsigned short*>(planes), plane, dz_cur_oflag);
    DZ_HIP(hipGetLastError());
    return 0;
}

int dz_launch_norm_f32(const float* y, const float* part, int ntile, int P, const float* gamma, const float* beta,
                       float* out, int B, hipStream_t st) {
    DZ_REQUIRE(y && part && gamma && beta && out, "norm_f32: NULL argument");
    DZ_REQUIRE(B >= 1 && P >= 1 && ntile >= 1, "norm_f32: bad geometry (B %d, P %d)", B, P);
    DZ_LAUNCH(norm_f32_kernel, dim3((P + 31) / 32, B), dim3(256), 0, st, y, part, ntile, P, gamma, beta, out);
    DZ_HIP(hipGetLastError());
    return 0;
}
