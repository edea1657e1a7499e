// SincNet stages 1 and 2 on the f16 matrix cores:  InstanceNorm + LeakyReLU of the input (on load)
// -> Conv1d(Cin -> 60, k = 5) + bias -> MaxPool1d(3) -> pooled rows + the (sum, sumsq) partials of
// the next InstanceNorm.  (pyannote.audio SincNet.forward, third party, called from
// /root/reference/src/diart/models.py:133 and :262; SURVEY.md Appendix A.1, kernels K3 / K4.)
//
// The general split-f16 GEMM kernel (k_gemm_split.hip, POOL3 epilogue) treats these layers as an
// implicit GEMM with K = 5 taps x Cin: every k-tile re-loads its 32-wide slice of the input rows from
// L2, re-applies the normalisation and re-splits it — five times per input element, with only 64
// output columns to amortise it over (13 % of the split-f16 matrix peak, 80 + 28 us per network).
// Here the convolution keeps its structure:
//   * the tile's input rows (96 conv frames + 4 = 100 rows x Cin channels) are normalised, split
//     into (hi, lo * 2^11) f16 planes and parked in LDS ONCE; the five taps are five row offsets
//     into that image (row pitch Cin * 2 + 16 bytes: an odd multiple of 16, which makes the 16
//     lanes of every fragment read hit 16 distinct bank groups);
//   * the weights (64 x 5 Cin, two planes) stay in REGISTERS for the whole persistent workgroup:
//     wave (nt, kh) owns output channels 32 nt .. +31 and one half of the k-steps;
//   * MaxPool1d(3) is an element-wise maximum: the 96 frames are computed as three 32-row MFMA
//     blocks, block b = frames {3 m + b}, so pooled row m sits in the same lane and register of
//     all three (the trick of sinc_conv0_h_kernel, k_front.hip);
//   * the two k-halves meet once per tile through LDS (12 KiB per 32-channel block).
// Workgroup = 4 waves, two workgroups per CU; partials layout and tile size (96 conv frames) are
// those of the POOL3 GEMM epilogue, so finalize_norm and every consumer are unchanged.
#include "dz_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FR = 96;              // conv frames per tile (= 32 pooled rows)
constexpr int ROWS = FR + 4;        // input rows a tile touches (k = 5)

template <int CIN>
struct Geo {
    static constexpr int KS = 5 * CIN / 16;            // 16-wide k-steps (k = tap * CIN + c)
    static constexpr int KS0 = (KS + 1) / 2;           // k-steps of half 0 (half 1: KS - KS0)
    static constexpr int PITCH = CIN * 2 + 16;         // bytes per LDS row of one plane
    static constexpr int PLANE = ROWS * PITCH;
    static constexpr int XCH = 2 * 3 * 16 * 64 * 4;    // exchange: [nt][block][reg][lane] f32
    static constexpr int NORM = 2 * CIN * 4;           // scale | shift of the current chunk
    static constexpr int LDS = 2 * PLANE + XCH + NORM;
    static_assert(CIN % 16 == 0 && (PITCH / 16) % 2 == 1, "row pitch must be an odd multiple of 16 bytes");
};

__device__ __forceinline__ float leaky(float v) { return v > 0.f ? v : v * DZ_LEAKY_SLOPE; }

template <int CIN>
__global__ __launch_bounds__(256, 2) void conv_pool_h_kernel(
    const float* __restrict__ X, int Tin, int Tout, int Tstore, const float* __restrict__ nscale,
    const float* __restrict__ nshift, const float* __restrict__ npart, int npart_tiles, int npart_T,
    const float* __restrict__ ngamma, const float* __restrict__ nbeta,
    const unsigned short* __restrict__ wsp, int Kpad,
    const float* __restrict__ bias, float* __restrict__ Y, float* __restrict__ partials, int ntile,
    int total, int* __restrict__ oflag, long long* __restrict__ dbg) {
    using G = Geo<CIN>;
    // dbg (kbench only): shader-clock stamps of the phases of every tile, wave 0 / wave 3 lane 0 of each workgroup
#ifdef DZ_EXPERIMENTS       // phase stamps for tools/conv_pool_phases.py: experiments build only
    long long* dq = nullptr;
    if (dbg && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == 3))
        dq = dbg + ((long long)blockIdx.x * 2 + (threadIdx.x >> 7)) * 64;
    int dn = 0;
#define DZ_STAMP() do { if (dq && dn < 64) dq[dn++] = __builtin_readcyclecounter(); } while (0)
#else
#define DZ_STAMP() do { } while (0)
#endif
    extern __shared__ __attribute__((aligned(256))) char lds[];
    char* xs = lds;                                                  // [2 planes][ROWS][PITCH]
    float* xch = reinterpret_cast<float*>(lds + 2 * G::PLANE);       // [2][3][16][64]
    float* nrm = reinterpret_cast<float*>(lds + 2 * G::PLANE + G::XCH);   // scale[CIN] | shift[CIN]
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, li = l & 31, g = l >> 5;
    const int nt = w & 1, kh = w >> 1;                                // channel block, k-half
    const int ks_lo = kh ? G::KS0 : 0, ks_n = kh ? G::KS - G::KS0 : G::KS0;
    const int wg = dz_xcd_contiguous(blockIdx.x, gridDim.x);         // neighbouring tile ranges on ONE XCD
    const int t_begin = (int)((long long)wg * total / gridDim.x);
    const int t_end = (int)((long long)(wg + 1) * total / gridDim.x);
    if (t_begin >= t_end) return;

    // ---- weight fragments of this wave: channels 32 nt + li, k = 16 ks + 8 g .. +7 -------------
    f16x8 bh[G::KS0], bl[G::KS0];
    {
        const unsigned short* row = wsp + (long long)(32 * nt + li) * Kpad + 8 * g;
#pragma unroll
        for (int i = 0; i < G::KS0; ++i) {
            const int ks = ks_lo + (i < ks_n ? i : 0);
            bh[i] = *reinterpret_cast<const f16x8*>(row + 16 * ks);
            bl[i] = *reinterpret_cast<const f16x8*>(row + 64 * (long long)Kpad + 16 * ks);
        }
    }
    // A fragment offsets: block b, k-step ks -> tap = ks / (CIN/16), j = ks % (CIN/16):
    // byte = (3 li + b + tap) * PITCH + (16 j + 8 g) * 2
    const int abase = 3 * li * G::PITCH + 16 * g;
    const float bv = bias[32 * nt + li];
    float amax = 0.f;
    int cur_b = -1;

    // The tile's raw input rows travel global -> registers as ONE batch of loads and are then consumed
    // (normalised, split, parked in LDS).  Round 2 loaded, waited and parked one float4 per lane at a time — 8 dependent L2 / HBM
    // round trips per tile with nothing else for the wave to do: 30 k cycles per tile for 3.7 k of MFMAs.
    constexpr int C4 = CIN / 4;
    constexpr int NPK = (ROWS * C4 + 255) / 256;
    f32x4 pv[NPK];
    auto fetch = [&](int t) {
        const int b = t / ntile, t0 = (t - b * ntile) * FR;
        const float* Xb = X + (long long)b * Tin * CIN;
        // the (row, column) of every piece is loop invariant: left to itself the compiler hoists all
        // 2 x NPK of them out of the tile loop and keeps them live through the MFMA phase (spills)
        int tid_l = tid;
        asm volatile("" : "+v"(tid_l));
#pragma unroll
        for (int i = 0; i < NPK; ++i) {
            const int idx = tid_l + 256 * i;
            const int r = idx / C4, c4 = idx - r * C4;
            pv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (idx < ROWS * C4 && t0 + r < Tin)
                pv[i] = *reinterpret_cast<const f32x4*>(Xb + (long long)(t0 + r) * CIN + 4 * c4);
        }
    };

    for (int t = t_begin; t < t_end; ++t) {
        // (all of the tile's loads in flight at once; issued here, ahead of the norm update and the
        // barrier.  Issued a phase earlier — before or after the previous tile's MFMAs — the 32 registers
        // would be live next to the accumulators / the pooled blocks: the kernel is at 240 - 256 VGPRs
        // and spills)
        DZ_STAMP();
        fetch(t);
        const int b = t / ntile, tile = t - b * ntile;
        const int t0 = tile * FR;
        if (b != cur_b) {                       // scale / shift of this chunk's InstanceNorm
            __syncthreads();                    // (nobody still reads the previous chunk's)
            if (npart)       // straight from the producer's tile partials: no finalize launch
                dz_norm_from_partials(npart, b, npart_tiles, CIN, npart_T, ngamma, nbeta, nrm, tid, 256,
                                      reinterpret_cast<double*>(xch));   // (the exchange area is idle between tiles)
            else
                for (int i = tid; i < 2 * CIN; i += 256)
                    nrm[i] = i < CIN ? nscale[(long long)b * CIN + i] : nshift[(long long)b * CIN + i - CIN];
            cur_b = b;
        }
        __syncthreads();                        // previous tile's fragment / exchange reads are done
        DZ_STAMP();
        // ---- park the tile: normalise + LeakyReLU, split, two planes --------------------------
        int tid_p = tid;
        asm volatile("" : "+v"(tid_p));
#pragma unroll
        for (int i = 0; i < NPK; ++i) {
            const int idx = tid_p + 256 * i;
            if (idx < ROWS * C4) {
                const int r = idx / C4, c4 = idx - r * C4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (t0 + r < Tin) {
                    v = pv[i];
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(nrm + 4 * c4);
                    const f32x4 sh = *reinterpret_cast<const f32x4*>(nrm + CIN + 4 * c4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = leaky(v[e] * sc[e] + sh[e]);
                        amax = fmaxf(amax, fabsf(v[e]));
                        v[e] = __builtin_amdgcn_fmed3f(v[e], -65504.f, 65504.f);
                    }
                }
                const f16x4 hi = __builtin_convertvector(v, f16x4);
                const f16x4 lo = __builtin_convertvector((v - __builtin_convertvector(hi, f32x4)) * 2048.f, f16x4);
                char* d = xs + r * G::PITCH + 8 * c4;
                *reinterpret_cast<f16x4*>(d) = hi;
                *reinterpret_cast<f16x4*>(d + G::PLANE) = lo;
            }
        }
        DZ_STAMP();
        __syncthreads();
        DZ_STAMP();
        // ---- three blocks (frames 3 m + bk), this wave's half of the k-steps -------------------
        f32x16 part[3];
#pragma unroll
        for (int bk = 0; bk < 3; ++bk) {
            f32x16 accm, accx;
#pragma unroll
            for (int r = 0; r < 16; ++r) accm[r] = accx[r] = 0.f;
#pragma unroll
            for (int i = 0; i < G::KS0; ++i) {
                if (i < ks_n) {
                    const int ks = ks_lo + i;                            // wave-uniform
                    const int tap = ks / (CIN / 16), j = ks - tap * (CIN / 16);
                    const char* ap = xs + abase + (bk + tap) * G::PITCH + 32 * j;
                    const f16x8 ah = *reinterpret_cast<const f16x8*>(ap);
                    const f16x8 al = *reinterpret_cast<const f16x8*>(ap + G::PLANE);
                    accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[i], accx, 0, 0, 0);
                    accm = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[i], accm, 0, 0, 0);
                    accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[i], accx, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) part[bk][r] = accm[r] + accx[r] * (1.f / 2048.f);
        }
        DZ_STAMP();
        // ---- the k-halves meet: half 1 hands its partial sums over ------------------------------
        if (kh == 1) {
#pragma unroll
            for (int bk = 0; bk < 3; ++bk)
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[((nt * 3 + bk) * 16 + r) * 64 + l] = part[bk][r];
        }
        __syncthreads();
        DZ_STAMP();
        if (kh == 0) {
            f32x16 pmax;
#pragma unroll
            for (int bk = 0; bk < 3; ++bk)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = (part[bk][r] + xch[((nt * 3 + bk) * 16 + r) * 64 + l]) + bv;
                    pmax[r] = bk == 0 ? v : fmaxf(pmax[r], v);
                }
            // pooled rows + partials: C/D column = lane & 31 = channel, row rho = pooled row
            const int ch = 32 * nt + li;
            float sum = 0.f, ssq = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pr = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (pr < Tstore) {
                    const float v = pmax[r];
                    Y[((long long)b * Tstore + pr) * 64 + ch] = v;
                    sum += v;
                    ssq += v * v;
                }
            }
            sum += __shfl_xor(sum, 32, 64);
            ssq += __shfl_xor(ssq, 32, 64);
            if (g == 0) {
                float* pp = partials + (((long long)b * ntile + tile) * 64 + ch) * 2;
                pp[0] = sum;
                pp[1] = ssq;
            }
        }
        DZ_STAMP();
    }
#undef DZ_STAMP
    dz_flag_range(oflag, amax);
}

}  // namespace
#ifdef DZ_EXPERIMENTS
long long* dz_conv_pool_dbg = nullptr;      // set by dz_k_conv_pool_debug (phase stamps, kbench only)
#else
static long long* const dz_conv_pool_dbg = nullptr;
#endif
namespace {

template <int CIN>
int launch(const DzConvGemm& p, hipStream_t st) {
    using G = Geo<CIN>;
    static DzAttrOnce attr_once;
    DZ_HIP(attr_once.raise((const void*)conv_pool_h_kernel<CIN>, (int)G::LDS));
    const int ntile = (p.Tout + FR - 1) / FR;
    const int total = ntile * p.B;
    const int grid = total < 512 ? total : 512;          // two resident workgroups per CU
    DZ_LAUNCH((conv_pool_h_kernel<CIN>), dim3(grid), dim3(256), G::LDS, st, p.X, p.Tin, p.Tout, p.Tstore,
              p.nscale, p.nshift, p.npart, p.npart_tiles, p.npart_T, p.ngamma, p.nbeta,
              reinterpret_cast<const unsigned short*>(p.Wsplit), p.Kpad, p.bias, p.Y,
              p.partials, ntile, total, p.oflag ? p.oflag : dz_cur_oflag, dz_conv_pool_dbg);
    DZ_HIP(hipGetLastError());
    return 0;
}

}  // namespace

// same descriptor as the POOL3 call of dz_launch_gemm_split (k = 5, dil = 1, Npad = 64, norm-on-load,
// X / Y dense: xbs = Tin * Cin, ybs = Tstore * 64, ldx = Cin, ldy = 64)
int dz_launch_conv_pool(const DzConvGemm& p, hipStream_t st) {
    DZ_REQUIRE(p.Wsplit && p.X && p.Y && p.partials && p.bias, "conv_pool: NULL operand");
    DZ_REQUIRE((p.nscale && p.nshift) || (p.npart && p.ngamma && p.nbeta && p.npart_tiles > 0 && p.npart_T > 0),
               "conv_pool: needs nscale / nshift or the producer's partials + affine");
    DZ_REQUIRE(p.epi == DZ_EPI_POOL3 && p.taps == 5 && p.dil == 1 && p.norm_on_load && p.Npad == 64 &&
                   p.Nstore == 64 && p.ldy == 64,
               "conv_pool: built for k = 5, dilation 1, 64 output columns, norm-on-load, MaxPool1d(3)");
    DZ_REQUIRE((p.Cin == 80 || p.Cin == 64) && p.ldx == p.Cin && p.nld == p.Cin && p.K == 5 * p.Cin &&
                   p.Kpad >= p.K,
               "conv_pool: Cin must be 80 or 64 (dense rows), K = 5 Cin");
    DZ_REQUIRE(p.Tout == p.Tin - 4 && p.Tstore == p.Tout / 3 && p.xbs == (long long)p.Tin * p.Cin &&
                   p.ybs == (long long)p.Tstore * 64,
               "conv_pool: geometry mismatch");
    return p.Cin == 80 ? launch<80>(p, st) : launch<64>(p, st);
}
