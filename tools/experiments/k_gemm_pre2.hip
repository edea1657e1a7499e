// Pre-split implicit GEMM, second structure: same contract, descriptor, arithmetic and bit-exact
// results as k_gemm_pre.hip (both operands as f16 hi / lo planes, three v_mfma_f32_32x32x16_f16 per
// product into two f32 accumulators, operand tiles by LDS-DMA), different schedule.
//
// k_gemm_pre.hip: 128 x 128 tile, 4 waves, two LDS stages, ONE tile in flight: the DMA of tile kt+1
// has a single compute phase (~0.3 us) to land, an L2 hit under load takes longer — the timing
// ablation (tools/kbench.py KB_ABLATE) puts 30 % of a K = 1536 layer on that wait, and the
// MFMA-only loop at 58 % of what the 1.15-round grid allows (barrier skew with 24 MFMAs per wave
// between barriers).  Here:
//   * FOUR stages of 32 KiB (one workgroup per CU), three tiles in flight, counted vmcnt: a wave
//     waits only for its own pieces of the tile it is about to read (s_waitcnt vmcnt(8)), never for
//     the queue to drain;
//   * 8 waves = 2 (M) x 2 (N) x 2 (k-halves): the two halves of a 32-wide k-tile go to different
//     waves, so a wave keeps the 64 x 64 wave tile (8 fragment reads per 12 MFMAs) while the 128 x
//     128 workgroup tile keeps the finer grid quantisation (588 tiles / 256 CUs = 2.3 -> 3 rounds,
//     against 4 round-equivalents for 256 x 128 tiles); two waves per SIMD cover each other's
//     barrier skew;
//   * the k-halves meet once, after the loop, through LDS: each wave hands over the 32 x 64 half of
//     its partial sums that the other one finishes (balanced epilogue).
#include "dz_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, KT = 32, NST = 4;
constexpr int PLANE = 128 * 64;            // bytes of one f16 plane of a stage
constexpr int STAGE = 4 * PLANE;           // A hi | A lo | B hi | B lo
constexpr size_t LDS_BYTES = NST * STAGE;  // 128 KiB (the 64 KiB exchange area aliases it after the loop)
constexpr float LO_UNSCALE = 1.f / 2048.f;

__device__ __forceinline__ float leaky(float v) { return v > 0.f ? v : v * DZ_LEAKY_SLOPE; }

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_pre2_kernel(DzConvGemm p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    int bx, by, bz;
    dz_tile_map(p.agroup, bx, by, bz);
    const int t0 = bx * BM, n0 = by * BN;

    // ---- staging role: plane w & 3 (0 A hi, 1 A lo, 2 B hi, 3 B lo), rows 64 (w >> 2) .. +63 ----
    const int pl = w & 3, hf = w >> 2;
    const bool isB = pl >= 2;
    const int lo = pl & 1;
    const unsigned short* A = reinterpret_cast<const unsigned short*>(p.Xsplit);
    const unsigned short* W = reinterpret_cast<const unsigned short*>(p.Wsplit);
    const unsigned short* src = isB ? W + (long long)lo * p.Npad * p.Kpad : A + (long long)lo * p.xplane;
    const int ld = isB ? p.Kpad : p.ldx;
    const unsigned nbytes = (unsigned)((isB ? (long long)p.Npad : (long long)p.Tin) * ld * 2);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    const int row0 = (isB ? n0 : t0) + 64 * hf + (l >> 2);
    const int voff0 = row0 * ld * 2 + (((l & 3) ^ ((l >> 4) & 3)) << 4);   // source chunk = slot ^ ((row >> 2) & 3)
    const int vstep = 16 * ld * 2;
    auto issue = [&](int kt) {
        int soff;
        if (isB) {
            soff = kt * (KT * 2);
        } else {
            const int k = kt * KT;
            int tap = 0, c = k;
            if (p.taps > 1) {
                tap = k / p.Cin;
                c = k - tap * p.Cin;
            }
            soff = (tap * p.dil * p.ldx + c) * 2;
        }
        char* dst = smem + (kt & (NST - 1)) * STAGE + pl * PLANE + hf * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsrc, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, voff0 + i * vstep, soff, 0, 0);
    };

    // ---- MFMA coordinates: (k-half, M, N) = (w >> 2, (w >> 1) & 1, w & 1), wave tile 64 x 64 ------
    const int li = l & 31, g = l >> 5;
    const int kh = w >> 2, wm = (w >> 1) & 1, wn = w & 1;
    const int foff = li * 64 + (((2 * kh + g) ^ ((li >> 2) & 3)) << 4);
    f32x16 accm[2][2], accx[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) accm[mt][nt][r] = accx[mt][nt][r] = 0.f;

    const int nk = p.Kpad / KT;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nk) issue(s);
    for (int kt = 0; kt < nk; ++kt) {
        // this wave's pieces of tile kt have landed (the pieces of up to two later tiles may still fly)
        const int ahead = nk - 1 - kt;
        if (ahead >= 2)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 1)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... everybody's have, and everybody has finished the fragment reads of tile kt-1
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + NST - 1 < nk) issue(kt + NST - 1);          // into the stage tile kt-1 occupied
        const char* st = smem + (kt & (NST - 1)) * STAGE;
        const char* sa = st + (wm * 64) * 64;
        const char* sb = st + 2 * PLANE + (wn * 64) * 64;
        f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            ah[t] = *reinterpret_cast<const f16x8*>(sa + t * 2048 + foff);
            al[t] = *reinterpret_cast<const f16x8*>(sa + PLANE + t * 2048 + foff);
            bh[t] = *reinterpret_cast<const f16x8*>(sb + t * 2048 + foff);
            bl[t] = *reinterpret_cast<const f16x8*>(sb + PLANE + t * 2048 + foff);
        }
        // transposed product (weights = row operand): a lane ends with one output row and groups of
        // four consecutive output columns, see k_gemm_pre.hip
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                accx[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[nt], al[mt], accx[mt][nt], 0, 0, 0);
                accm[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[nt], ah[mt], accm[mt][nt], 0, 0, 0);
                accx[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[nt], ah[mt], accx[mt][nt], 0, 0, 0);
            }
    }

    // ---- the k-halves meet: wave (kh, q) finishes rows mt = kh of its 64 x 64 tile and hands the
    // other 32 rows' partial sums to wave (1 - kh, q).  xch[q][from kh][nt * 16 + r][lane] f32.
    __syncthreads();                                       // every fragment read is done: LDS is free
    float* xch = reinterpret_cast<float*>(smem);
    const int q = w & 3;
    f32x16 mine[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v0 = accm[0][nt][r] + accx[0][nt][r] * LO_UNSCALE;
            const float v1 = accm[1][nt][r] + accx[1][nt][r] * LO_UNSCALE;
            mine[nt][r] = kh ? v1 : v0;
            xch[((q * 2 + kh) * 32 + nt * 16 + r) * 64 + l] = kh ? v0 : v1;
        }
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[nt][r] += xch[((q * 2 + (1 - kh)) * 32 + nt * 16 + r) * 64 + l];

    // ---- epilogue of rows mt = kh (C/D map of the transposed product: column = lane & 31 = output row,
    // registers 4k .. 4k+3 = four consecutive output columns) -------------------------------------------
    unsigned short* Yhi = reinterpret_cast<unsigned short*>(p.Ysplit);
    float amax = 0.f;
    const int t = t0 + wm * 64 + kh * 32 + li;
    const bool ok = t < p.Tout;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int n = n0 + wn * 64 + nt * 32 + 8 * k + 4 * g;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
            f32x4 e0 = {1.f, 1.f, 1.f, 1.f}, e1 = {0.f, 0.f, 0.f, 0.f};
            if (EPI == DZ_EPI_TDNN || EPI == DZ_EPI_RELU_BN) {
                e0 = *reinterpret_cast<const f32x4*>(p.e0 + n);
                e1 = *reinterpret_cast<const f32x4*>(p.e1 + n);
            }
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = mine[nt][4 * k + e] + bv[e];
                if (EPI == DZ_EPI_BIAS_LEAKY) x = leaky(x);
                if (EPI == DZ_EPI_TDNN) x = leaky(x) * e0[e] + e1[e];
                if (EPI == DZ_EPI_RELU_BN) x = fmaxf(x, 0.f) * e0[e] + e1[e];
                v[e] = x;
            }
            const long long idx = (long long)t * p.ldy + n;
            if (p.Y && ok) {
                if (n + 3 < p.Nstore) {
                    *reinterpret_cast<f32x4*>(p.Y + idx) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.Nstore) p.Y[idx + e] = v[e];
                }
            }
            if (Yhi) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e >= p.Nstore) v[e] = 0.f;
                    amax = fmaxf(amax, fabsf(v[e]));
                    v[e] = __builtin_amdgcn_fmed3f(v[e], -65504.f, 65504.f);
                }
                const f16x4 hi = __builtin_convertvector(v, f16x4);
                const f16x4 lo4 = __builtin_convertvector((v - __builtin_convertvector(hi, f32x4)) * 2048.f, f16x4);
                if (ok) {
                    *reinterpret_cast<f16x4*>(Yhi + idx) = hi;
                    *reinterpret_cast<f16x4*>(Yhi + p.yplane + idx) = lo4;
                }
            }
        }
    dz_flag_range(p.oflag, amax);
}

template <int EPI>
int launch(const DzConvGemm& p, hipStream_t st) {
    static DzAttrOnce attr_once;
    DZ_HIP(attr_once.raise((const void*)gemm_pre2_kernel<EPI>, (int)LDS_BYTES));
    dim3 grid((p.Tout + BM - 1) / BM, p.Npad / BN, 1);
    DZ_LAUNCH((gemm_pre2_kernel<EPI>), grid, dim3(512), LDS_BYTES, st, p);
    DZ_HIP(hipGetLastError());
    return 0;
}

}  // namespace

// requirements are checked by dz_launch_gemm_pre (k_gemm_pre.hip), which dispatches here
int dz_launch_gemm_pre2(const DzConvGemm& p, hipStream_t st) {
    switch (p.epi) {
        case DZ_EPI_BIAS: return launch<DZ_EPI_BIAS>(p, st);
        case DZ_EPI_BIAS_LEAKY: return launch<DZ_EPI_BIAS_LEAKY>(p, st);
        case DZ_EPI_TDNN: return launch<DZ_EPI_TDNN>(p, st);
        case DZ_EPI_RELU_BN: return launch<DZ_EPI_RELU_BN>(p, st);
    }
    dz_set_error("gemm_pre2: epilogue %d is not built on the pre-split path", p.epi);
    return 2;
}
