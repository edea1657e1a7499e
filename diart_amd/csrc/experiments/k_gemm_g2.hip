// Pre-split implicit GEMM, generation 2 (round 4): same contract, operand planes, descriptor and epilogues as
// k_gemm_pre.hip (third-party TDNN / LSTM-projection / MLP layers reached from
// /root/reference/src/diart/models.py:133, :262; SURVEY.md kernels K5 / K6 / K8), different loop.
//
//   Y[t][n] = epi( sum_{tap,c} X[t + tap*dil][c] * W[n][tap*Cin + c] + bias[n] )
//
// What generation 1 is bound by: one k-tile in flight per workgroup (vmcnt(0) + barrier
// every 24 MFMAs), so a wave that sits ALONE on its SIMD — every workgroup that shares a CU with an LSTM
// recurrence workgroup, i.e. half the chip in the 64-stream pipeline — stalls for the LDS-DMA round trip and
// for the fragment reads behind each barrier; two accumulators per fragment (main / cross) leave no
// registers to pipeline with.  Here:
//
//   * ONE accumulator per fragment.  x = hi + lo * 2^-11 for both operands (the plane format is unchanged);
//     the weight fragment's hi half is also used scaled by 2^11 (v_pk_mul_f16 on the fragment registers,
//     exact: weights.py refuses |w| >= 32), so
//         acc += (2^11 hi_w) hi_x  +  hi_w lo_x'  +  lo_w' hi_x          (lo' = lo * 2^11, as stored)
//     is the whole product at scale 2^11 and the epilogue multiplies by 2^-11 once.  64 (MT = 2) accumulator
//     registers instead of 128.
//   * THREE LDS stages of one 32-wide k-tile each and counted vmcnt: the pieces of tile kt + 2 are issued
//     while tile kt is computed and waited for 1.5 iterations later (s_waitcnt vmcnt(H): only the pieces
//     issued since may still fly); the queue never drains inside the loop.
//   * the barrier of a k-tile sits BETWEEN its two 16-wide k-steps, and the fragments of the next tile's
//     first k-step are read behind it into the registers the first k-step has just released (two fragment
//     sets): after the barrier the wave goes on with MFMAs whose operands are already in registers, the
//     LDS latency of the next reads runs under them.
//   * LDS-DMA pieces interleaved with the MFMAs (one piece per 3 - 4 MFMAs), every wave loads its share of
//     both operands (no per-wave roles), parameters by LDS-DMA as well (no VGPR-destination load inside the
//     pipeline: hipcc would wait vmcnt(0) for it).
//   * workgroup tile (64 MT) x 128 with MT = 2, 3, 4 row fragments per wave (wave tile 32 MT x 64): 0.67 /
//     0.56 / 0.5 KB of fragment reads and 341 / 284 / 256 operand bytes from L2 per MFMA.  MT = 2 stays below
//     176 VGPRs, so that a workgroup fits beside a recurrence workgroup (2 x 168 of a SIMD's 512 registers).
//
// LDS: 3 stages x (A hi | A lo | B hi | B lo) + parameters = 97.5 / 121.5 / 145.5 KiB: one workgroup per CU.
#include "dz_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BN = 128, KT = 32, NST = 3;
constexpr int PLANE_B = 128 * 64;                       // bytes of one f16 plane of the weight tile of a stage
constexpr int plane_a(int MT) { return 64 * MT * 64; }  // activation rows of a tile: 2 wave rows x 32 MT
constexpr int stage_bytes(int MT) { return 2 * plane_a(MT) + 2 * PLANE_B; }
constexpr int PAR_BYTES = 3 * BN * 4;                   // bias | e0 | e1
constexpr size_t lds_bytes(int MT) { return (size_t)NST * stage_bytes(MT) + PAR_BYTES; }
constexpr float UNSCALE = 1.f / 2048.f;

__device__ __forceinline__ float leaky(float v) { return v > 0.f ? v : v * DZ_LEAKY_SLOPE; }

#define G2_PIN() __builtin_amdgcn_sched_barrier(0)

// DBG (timing experiments, results are wrong): 1 = no LDS-DMA inside the loop, 2 = no MFMAs, 4 = no fragment
// reads inside the loop, 8 = no barrier (waits only)
template <int EPI, int MT, int DBG = 0>
__device__ __forceinline__ void g2_tile(const DzConvGemm& p, const int t0, const int n0, char* smem) {
    constexpr int PLANE_A = plane_a(MT), STAGE = stage_bytes(MT);
    constexpr int NPA = 2 * MT, NPB = 4, PPW = NPA + NPB;     // LDS-DMA pieces per wave and k-tile
    constexpr int H = PPW / 2;                                // pieces issued before the mid-tile barrier
    constexpr int NM = 6 * MT;                                // MFMAs per wave and k-step
    constexpr bool AFF = EPI == DZ_EPI_TDNN || EPI == DZ_EPI_RELU_BN;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;

    // ---- operand streams ---------------------------------------------------------------------------------
    // A piece = 16 rows x 64 B of one plane's k-block (1 KiB contiguous in the kb-major plane): lane -> row
    // l >> 2, LDS slot l & 3 holding source chunk slot ^ ((row >> 2) & 3) (the LDS-DMA destination is
    // lane-linear: the swizzle is applied to the source chunk and again on the fragment read).  Wave w takes
    // 16-row blocks w, w + 4, ... of every plane of both operands.
    const unsigned short* Xs = reinterpret_cast<const unsigned short*>(p.Xsplit);
    const unsigned short* Ws = reinterpret_cast<const unsigned short*>(p.Wsplit);
    const int arows = (int)((unsigned)p.xplane / (unsigned)p.ldx);
    const unsigned abytes = (unsigned)((long long)arows * p.ldx * 2), bbytes = (unsigned)((long long)p.Npad * p.Kpad * 2);
    const __amdgpu_buffer_rsrc_t rsA_hi = __builtin_amdgcn_make_buffer_rsrc((void*)Xs, 0, abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(Xs + p.xplane), 0, abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB_hi = __builtin_amdgcn_make_buffer_rsrc((void*)Ws, 0, bbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB_lo =
        __builtin_amdgcn_make_buffer_rsrc((void*)(Ws + (long long)p.Npad * p.Kpad), 0, bbytes, 0x00020000);
    const int chunk = ((l & 3) ^ ((l >> 4) & 3)) << 4;
    const int voffA = (t0 + 16 * w + (l >> 2)) * 64 + chunk;
    const int voffB = (n0 + 16 * w + (l >> 2)) * 64 + chunk;
    const int kbA = arows * 64, kbB = p.Npad * 64;              // bytes of one k-block of a plane
    const int tapA = p.dil * 64, tapB = (p.Cin >> 5) * kbB;     // one tap further
    // fetch pointer: tile order = (channel block, tap) with the tap innermost (k_gemm_pre.hip, issue())
    int f_cblk = 0, f_tap = 0, f_soffA = 0, f_soffB = 0, f_stage = 0;
    auto f_advance = [&]() {
        // everything here is wave-uniform; say so, or hipcc keeps the channel-block counter in a VGPR and wraps
        // every LDS-DMA instruction in a waterfall loop over its scalar offset (cdna_hip_programming.md T20)
        const bool wrap = f_tap + 1 == p.taps;
        f_tap = __builtin_amdgcn_readfirstlane(wrap ? 0 : f_tap + 1);
        f_cblk = __builtin_amdgcn_readfirstlane(f_cblk + (wrap ? 1 : 0));
        f_soffA = __builtin_amdgcn_readfirstlane(f_cblk * kbA + f_tap * tapA);
        f_soffB = __builtin_amdgcn_readfirstlane(f_cblk * kbB + f_tap * tapB);
        f_stage = __builtin_amdgcn_readfirstlane(f_stage == NST - 1 ? 0 : f_stage + 1);
    };
    char* const dA = smem + w * 1024;
    char* const dB = smem + 2 * PLANE_A + w * 1024;
    bool in_loop = false;
    auto piece = [&](const int j) {          // j = 0 .. PPW-1 (compile-time after unrolling)
        if ((DBG & 1) && in_loop) return;
        char* const st = (j < NPA ? dA : dB) + f_stage * STAGE;
        if (j < NPA) {
            const int lo = j >= MT, i = j - lo * MT;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(lo ? rsA_lo : rsA_hi,
                (__attribute__((address_space(3))) void*)(st + lo * PLANE_A + i * 4096), 16, voffA + i * 4096,
                f_soffA, 0, 0);
        } else {
            const int jb = j - NPA, lo = jb >= 2, i = jb - lo * 2;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(lo ? rsB_lo : rsB_hi,
                (__attribute__((address_space(3))) void*)(st + lo * PLANE_B + i * 4096), 16, voffB + i * 4096,
                f_soffB, 0, 0);
        }
    };

    // ---- epilogue parameters -> LDS by LDS-DMA (4 bytes per lane: 128 floats = 2 instructions per array) ----
    float* par = reinterpret_cast<float*>(smem + NST * STAGE);
    if (w < (AFF ? 3 : 1)) {
        const float* src = w == 0 ? p.bias : w == 1 ? p.e0 : p.e1;
        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (unsigned)p.Npad * 4u, 0x00020000);
        char* d = reinterpret_cast<char*>(par) + w * (BN * 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (__attribute__((address_space(3))) void*)d, 4, (n0 + l) * 4, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (__attribute__((address_space(3))) void*)(d + 256), 4,
                                                 (n0 + 64 + l) * 4, 0, 0, 0);
    }

    // ---- MFMA coordinates: 2 x 2 waves, wave tile (32 MT) x 64 ------------------------------------------------
    const int li = l & 31, g = l >> 5;
    const int wm = w >> 1, wn = w & 1;
    const int sw = (li >> 2) & 3;
    const int foff0 = li * 64 + (((0 + g) ^ sw) << 4), foff1 = li * 64 + (((2 + g) ^ sw) << 4);
    const char* const fa = smem + (wm * 32 * MT) * 64;
    const char* const fb = smem + 2 * PLANE_A + (wn * 64) * 64;
    f32x16 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    struct Frags {
        f16x8 ah[MT], al[MT], bh[2], bl[2];
    };
    Frags S0, S1;
    // fragment r of a set (compile-time r): the order in which a k-step's MFMAs first need them
    //   r = 0, 1: bh[0], bh[1]   2 .. MT+1: ah[.]   MT+2 .. 2MT+1: al[.]   2MT+2, 2MT+3: bl[0], bl[1]
    constexpr int NR = 4 + 2 * MT;
    auto read_one = [&](Frags& S, const int stage, const int foff, const int r) {
        if ((DBG & 4) && in_loop) return;
        const char* a = fa + stage * STAGE + foff;
        const char* b = fb + stage * STAGE + foff;
        if (r < 2)
            S.bh[r] = *reinterpret_cast<const f16x8*>(b + r * 2048);
        else if (r < 2 + MT)
            S.ah[r - 2] = *reinterpret_cast<const f16x8*>(a + (r - 2) * 2048);
        else if (r < 2 + 2 * MT)
            S.al[r - 2 - MT] = *reinterpret_cast<const f16x8*>(a + PLANE_A + (r - 2 - MT) * 2048);
        else
            S.bl[r - 2 - 2 * MT] = *reinterpret_cast<const f16x8*>(b + PLANE_B + (r - 2 - 2 * MT) * 2048);
    };
    auto read_frags = [&](Frags& S, const int stage, const int foff) {
#pragma unroll
        for (int r = 0; r < NR; ++r) read_one(S, stage, foff, r);
    };
    // One 16-wide k-step: 6 MT MFMAs ordered (product, mt, nt) so that two MFMAs on one accumulator are 2 MT
    // instructions apart.  Between them, one at a time: the NR fragment reads of the NEXT k-step into the other
    // register set (all at once they are a burst of 4 waves x 8 KB that the LDS serves at its full rate while no
    // wave issues an MFMA — measured: +12 us on a 30 us tile, tools/g2ablate.py) and one LDS-DMA piece of the
    // tile being fetched after every GAP-th MFMA (pieces [p0, p0 + np)).  TRANSPOSED product as in
    // k_gemm_pre.hip: the weight fragment is the MFMA's row operand, so a lane ends with one output row (lane &
    // 31) and groups of four consecutive output columns.
    auto kstep = [&](const Frags& S, Frags& Sn, const bool rd, const int rstage, const int rfoff, const int p0,
                     const int np) {
        f16x8 b2[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) b2[t] = S.bh[t] * (_Float16)2048.f;
        const int gap = np > 0 ? NM / np : NM + 1;
        constexpr int rgap = NM / NR;
        int issued = 0, nread = 0;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int ty = m / (2 * MT), r = m - ty * 2 * MT, mt = r >> 1, nt = r & 1;
            const f16x8 bo = ty == 0 ? b2[nt] : ty == 1 ? S.bh[nt] : S.bl[nt];
            const f16x8 ao = ty == 1 ? S.al[mt] : S.ah[mt];
            if (DBG & 2) {
                asm volatile("" ::"v"(bo), "v"(ao));
            } else {
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bo, ao, acc[mt][nt], 0, 0, 0);
            }
            if (rd && nread < NR && (m + 1) % rgap == 0) {
                G2_PIN();
                read_one(Sn, rstage, rfoff, nread);
                G2_PIN();
                ++nread;
            }
            if (issued < np && (m + 1) % gap == 0) {
                G2_PIN();
                piece(p0 + issued);
                G2_PIN();
                ++issued;
            }
        }
#pragma unroll
        for (; issued < np; ++issued) piece(p0 + issued);
        if (rd) {
#pragma unroll
            for (; nread < NR; ++nread) read_one(Sn, rstage, rfoff, nread);
        }
    };

    const int nk = p.Kpad / KT;
    // ---- prologue: tiles 0 and 1 in flight, then the first fragments --------------------------------------
#pragma unroll
    for (int j = 0; j < PPW; ++j) piece(j);
    if (nk > 1) {
        f_advance();
#pragma unroll
        for (int j = 0; j < PPW; ++j) piece(j);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");
    read_frags(S0, 0, foff0);

    // ---- main loop -------------------------------------------------------------------------------------------
    // iteration kt (stage s = kt % 3): S0 = fragments of (kt, k-step 0), read during iteration kt - 1
    //   top   : S1 <- (kt, k-step 1); MFMAs of k-step 0 with the first H pieces of tile kt + 2
    //           (stage (kt + 2) % 3 = (kt - 1) % 3: every wave has passed the barrier of iteration kt - 1,
    //           before which it had completed its reads of tile kt - 1)
    //   middle: wait for the own pieces of tile kt + 1 (issued during iteration kt - 1; only the H pieces of
    //           tile kt + 2 issued above may still fly), barrier -> tile kt + 1 is visible to everybody
    //   bottom: S0 <- (kt + 1, k-step 0); MFMAs of k-step 1 with the other pieces of tile kt + 2
    int s = 0;
    in_loop = true;
    auto body = [&](auto more_c, auto next_c) {
        constexpr bool more = decltype(more_c)::value;      // tile kt + 2 exists
        constexpr bool next = decltype(next_c)::value;      // tile kt + 1 exists
        if (more) f_advance();
        kstep(S0, S1, true, s, foff1, 0, more ? H : 0);
        const int s1 = s == NST - 1 ? 0 : s + 1;
        if (next) {
            if (DBG & 8) {
                if (more)
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(H) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            } else if (more)
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(H) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            G2_PIN();
        }
        kstep(S1, S0, next, s1, foff0, H, more ? PPW - H : 0);
        s = s1;
    };
    int kt = 0;
    for (; kt + 2 < nk; ++kt) body(std::true_type{}, std::true_type{});
    if (kt + 1 < nk) body(std::false_type{}, std::true_type{});
    body(std::false_type{}, std::false_type{});

    // ---- epilogue.  C/D map of the transposed product: column = lane & 31 = output ROW t, register r =
    // output column n0' + (r & 3) + 8 (r >> 2) + 4 (lane >> 5): registers 4k .. 4k+3 are four consecutive
    // columns -> one 16-byte f32 store, or one 8-byte store per f16 plane (hi = f16(v), lo = f16((v - hi) *
    // 2^11), clamped to +-65504 and flagged beyond).  Columns >= Nstore of a padded layer are written as zeros
    // to the planes (K padding of the consumer) and not at all to the f32 output.
    unsigned short* Yhi = reinterpret_cast<unsigned short*>(p.Ysplit);
    const long long yrows = Yhi ? (long long)((unsigned)p.yplane / (unsigned)p.ldy) : 0;
    const bool fulln = n0 + BN <= p.Nstore;                 // no column of this tile is padding
    float amax = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int t = t0 + wm * 32 * MT + mt * 32 + li;
        const bool ok = t < p.Tout;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                asm volatile("" ::: "memory");      // keep one column group's parameter reads next to their use
                const int nc = wn * 64 + nt * 32 + 8 * k + 4 * g, n = n0 + nc;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(par + nc);
                f32x4 e0 = {1.f, 1.f, 1.f, 1.f}, e1 = {0.f, 0.f, 0.f, 0.f};
                if (AFF) {
                    e0 = *reinterpret_cast<const f32x4*>(par + BN + nc);
                    e1 = *reinterpret_cast<const f32x4*>(par + 2 * BN + nc);
                }
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = acc[mt][nt][4 * k + e] * UNSCALE + bv[e];
                    if (EPI == DZ_EPI_BIAS_LEAKY) x = leaky(x);
                    if (EPI == DZ_EPI_TDNN) x = leaky(x) * e0[e] + e1[e];
                    if (EPI == DZ_EPI_RELU_BN) x = fmaxf(x, 0.f) * e0[e] + e1[e];
                    v[e] = x;
                }
                if (!fulln) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e >= p.Nstore) v[e] = 0.f;
                }
                if (p.Y && ok) {
                    float* y = p.Y + (long long)t * p.ldy + n;
                    if (fulln || n + 3 < p.Nstore) {
                        *reinterpret_cast<f32x4*>(y) = v;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + e < p.Nstore) y[e] = v[e];
                    }
                }
                if (Yhi) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (ok) amax = fmaxf(amax, fabsf(v[e]));
                        v[e] = __builtin_amdgcn_fmed3f(v[e], -65504.f, 65504.f);
                    }
                    const f16x4 hi = __builtin_convertvector(v, f16x4);
                    const f16x4 lo = __builtin_convertvector((v - __builtin_convertvector(hi, f32x4)) * 2048.f, f16x4);
                    if (ok) {
                        const long long kidx = dz_kb(t, n, yrows);
                        *reinterpret_cast<f16x4*>(Yhi + kidx) = hi;
                        *reinterpret_cast<f16x4*>(Yhi + p.yplane + kidx) = lo;
                    }
                }
            }
    }
    dz_flag_range(p.oflag, amax);
}

template <int EPI, int MT, int DBG = 0>
__global__ __launch_bounds__(256) void gemm_g2_kernel(DzConvGemm p, int gx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int gy = p.Npad / BN;
    int bx, by, bz;
    dz_tile_map_lin(blockIdx.x, gx, gy, 1, p.agroup, bx, by, bz);
    g2_tile<EPI, MT, DBG>(p, bx * 64 * MT, by * BN, smem);
}

template <int EPI, int MT>
int launch_mt(const DzConvGemm& p, hipStream_t st) {
    static DzAttrOnce attr_once;
    DZ_HIP(attr_once.raise((const void*)gemm_g2_kernel<EPI, MT>, (int)lds_bytes(MT)));
    const int gx = (p.Tout + 64 * MT - 1) / (64 * MT), gy = p.Npad / BN;
    DZ_LAUNCH((gemm_g2_kernel<EPI, MT>), dim3(gx * gy), dim3(256), lds_bytes(MT), st, p, gx);
    DZ_HIP(hipGetLastError());
    return 0;
}

// timing experiments (dz_k_gemm_g2 with row_fragments = 2 + 16 * DBG, TDNN epilogue only)
template <int DBG>
int launch_dbg(const DzConvGemm& p, hipStream_t st) {
    static DzAttrOnce attr_once;
    DZ_HIP(attr_once.raise((const void*)gemm_g2_kernel<DZ_EPI_TDNN, 2, DBG>, (int)lds_bytes(2)));
    const int gx = (p.Tout + 127) / 128, gy = p.Npad / BN;
    DZ_LAUNCH((gemm_g2_kernel<DZ_EPI_TDNN, 2, DBG>), dim3(gx * gy), dim3(256), lds_bytes(2), st, p, gx);
    DZ_HIP(hipGetLastError());
    return 0;
}

template <int EPI>
int launch(const DzConvGemm& p, int mt, hipStream_t st) {
    switch (mt) {
        case 2: return launch_mt<EPI, 2>(p, st);
        case 3: return launch_mt<EPI, 3>(p, st);
        case 4: return launch_mt<EPI, 4>(p, st);
    }
    dz_set_error("gemm_g2: row fragments per wave must be 2, 3 or 4 (got %d)", mt);
    return 2;
}

}  // namespace

// DZ_G2_MT: row fragments per wave of the generation-2 kernel (2: 128 x 128 tiles, <= 176 VGPRs; 3: 192 x 128;
// 4: 256 x 128)
int dz_g2_default_mt() {
    static const int mt = [] {
        const char* e = getenv("DZ_G2_MT");
        const int v = e ? atoi(e) : 2;
        return v >= 2 && v <= 4 ? v : 2;
    }();
    return mt;
}

// requirements are those of dz_launch_gemm_pre (k_gemm_pre.hip), which checks them and dispatches here
int dz_launch_gemm_g2(const DzConvGemm& p, int mt, hipStream_t st) {
    if (mt <= 0) mt = dz_g2_default_mt();
    if (mt >= 16) {
        switch (mt >> 4) {
            case 1: return launch_dbg<1>(p, st);
            case 2: return launch_dbg<2>(p, st);
            case 3: return launch_dbg<3>(p, st);
            case 5: return launch_dbg<5>(p, st);
            case 6: return launch_dbg<6>(p, st);
            case 7: return launch_dbg<7>(p, st);
            case 13: return launch_dbg<13>(p, st);
            case 15: return launch_dbg<15>(p, st);
        }
        dz_set_error("gemm_g2: timing experiment %d is not instantiated", mt >> 4);
        return 2;
    }
    switch (p.epi) {
        case DZ_EPI_BIAS: return launch<DZ_EPI_BIAS>(p, mt, st);
        case DZ_EPI_BIAS_LEAKY: return launch<DZ_EPI_BIAS_LEAKY>(p, mt, st);
        case DZ_EPI_TDNN: return launch<DZ_EPI_TDNN>(p, mt, st);
        case DZ_EPI_RELU_BN: return launch<DZ_EPI_RELU_BN>(p, mt, st);
    }
    dz_set_error("gemm_g2: epilogue %d is not built on the pre-split path", p.epi);
    return 2;
}
