// Pre-split implicit GEMM, generation 3 (round 4): the loop of k_gemm_g2.hip (one accumulator per fragment, three
// LDS stages, counted vmcnt, mid-tile barrier, fragment reads and LDS-DMA pieces interleaved with the MFMAs) as a
// PERSISTENT kernel over a balanced split of the whole (tile, k-tile) iteration space — same contract, operand
// planes, descriptor and epilogues as k_gemm_pre.hip (third-party TDNN / LSTM-projection / MLP layers reached from
// /root/reference/src/diart/models.py:133, :262; SURVEY.md kernels K5 / K6 / K8).
//
//   Y[t][n] = epi( sum_{tap,c} X[t + tap*dil][c] * W[n][tap*Cin + c] + bias[n] )
//
// Why.  Measured (tools/g2ablate.py, tools/g2bench.py, profiles/r04_*): a 128 x 128 tile needs 42 B/clk/CU of
// operands from the L2 at the full matrix rate and the CU gets ~47 — its loop runs at 64 % of the MFMA rate
// however it is scheduled; a 256 x 128 tile needs 31 B/clk and runs at 83 %.  But the layers of this path are
// SMALL: tdnn2 is 280 such tiles on 256 CUs, i.e. two rounds of 58 us for 1.1 rounds of work, and every tile pays
// ~11 us of prologue (first operand round trip) + epilogue.  Here every CU gets one workgroup and every workgroup
// the same number of k-tile iterations (Stream-K): a workgroup walks a contiguous range of the iteration space
// [tile][k-tile] of its XCD group, the operand prefetch runs across tile boundaries, and a tile whose k-range is
// shared by two workgroups is finished by the one that holds its END: the other one parks its accumulators in a
// workspace (same-XCD L2 in practice) and raises a flag (agent-scope release / acquire, cdna_hip_programming.md
// Guideline 16).  The schedule is a pure function of (shape, grid), so results are reproducible.
//
// A workgroup only ever waits for a LOWER-numbered workgroup of its group (the one that holds the START of the
// tile), which never waits for it: with workgroups dispatched in ascending order this cannot deadlock; the wait is
// bounded anyway (a timeout raises the context's error flag instead of hanging the device).
#include "dz_common.h"
#include <stdlib.h>
#include <mutex>
#include <unordered_map>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BN = 128, KT = 32, NST = 3, NGRP = 8;
constexpr int PLANE_B = 128 * 64;
constexpr int plane_a(int MT) { return 64 * MT * 64; }
constexpr int stage_bytes(int MT) { return 2 * plane_a(MT) + 2 * PLANE_B; }
constexpr int PAR_BYTES = 3 * BN * 4;
constexpr size_t lds_bytes(int MT) { return (size_t)NST * stage_bytes(MT) + PAR_BYTES; }
constexpr float UNSCALE = 1.f / 2048.f;
constexpr size_t ws_bytes_per_wg(int MT) { return (size_t)256 * MT * 2 * 16 * 4; }   // the accumulators of a workgroup

__device__ __forceinline__ float leaky(float v) { return v > 0.f ? v : v * DZ_LEAKY_SLOPE; }

#define G3_PIN() __builtin_amdgcn_sched_barrier(0)

struct G3Args {
    float* ws;          // [grid][MT * 2 * 4][256] f32x4: parked accumulators
    int* flags;         // [grid]: epoch of the launch whose partial sums ws[b] holds
    int* err;           // set to 1 when a wait timed out
    int epoch;
};

template <int EPI, int MT>
__global__ __launch_bounds__(256) void gemm_g3_kernel(DzConvGemm p, G3Args ga) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 64 * MT, PLANE_A = plane_a(MT), STAGE = stage_bytes(MT);
    constexpr int NPA = 2 * MT, NPB = 4, PPW = NPA + NPB, H = PPW / 2, NM = 6 * MT, NR = 4 + 2 * MT;
    constexpr bool AFF = EPI == DZ_EPI_TDNN || EPI == DZ_EPI_RELU_BN;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;

    // ---- this workgroup's share: iterations [it0, it1) of its group's [tile][k-tile] space ---------------------
    // group = blockIdx % 8 (the XCD, when workgroups are dealt round-robin); it owns the row tiles grp, grp + 8, ...
    // and walks their column tiles innermost, so the workgroups of an XCD share activation tiles in its L2
    const int nk = p.Kpad / KT, gy = p.Npad / BN, gx = (p.Tout + BM - 1) / BM;
    const int b = blockIdx.x, G = gridDim.x;
    const int grp = b & (NGRP - 1), idx = b >> 3;
    const int nwg = (G - grp + NGRP - 1) / NGRP;                    // workgroups of this group
    const int ntile = grp < gx ? ((gx - grp + NGRP - 1) / NGRP) * gy : 0;
    const long long I = (long long)ntile * nk;
    const int it0 = (int)(I * idx / nwg), it1 = (int)(I * (idx + 1) / nwg);
    if (it0 >= it1) return;

    // ---- operand streams (k_gemm_g2.hip) ---------------------------------------------------------------------------
    const unsigned short* Xs = reinterpret_cast<const unsigned short*>(p.Xsplit);
    const unsigned short* Ws = reinterpret_cast<const unsigned short*>(p.Wsplit);
    const int arows = (int)((unsigned)p.xplane / (unsigned)p.ldx);
    const unsigned abytes = (unsigned)((long long)arows * p.ldx * 2), bbytes = (unsigned)((long long)p.Npad * p.Kpad * 2);
    // ONE descriptor per operand, spanning both planes (the lo plane is reached through the scalar offset): rows
    // beyond the hi plane read finite values of the lo plane, beyond the lo plane zeros — such rows only feed outputs
    // that are never stored (k_gemm_pre.hip).  Scalar registers are the scarce resource of this kernel.
    const int loA = (int)(p.xplane * 2), loB = (int)((long long)p.Npad * p.Kpad * 2);     // bytes from hi to lo
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Xs, 0, (unsigned)loA + abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Ws, 0, (unsigned)loB + bbytes, 0x00020000);
    const int chunk = ((l & 3) ^ ((l >> 4) & 3)) << 4;
    const int lrow = (16 * w + (l >> 2)) * 64 + chunk;              // this lane's row / chunk inside a piece
    const int kbA = arows * 64, kbB = p.Npad * 64;
    const int taps = p.taps, tapA = p.dil * 64, tapB = (p.Cin >> 5) * kbB;
    // k-tile order = (channel block, tap), tap innermost: the next k-tile is one tap further, or the first tap of
    // the next channel block
    const int wrapA = kbA - (taps - 1) * tapA, wrapB = kbB - (taps - 1) * tapB;
    // fetch cursor: iteration f_it = (tile f_tile of the group, k-tile f_k); tile -> (row tile, column tile)
    // A workgroup walks its range from the TOP DOWN.  Then the part of a shared tile that a workgroup does not
    // finish (the tile's top belongs to the NEXT workgroup) is the FIRST thing it computes, and the workgroup that
    // finishes a tile (it holds the tile's last k-tile) does so at the very END of its own range: by then its
    // lower-numbered neighbours parked their sums long ago.  (Walking upwards, a workgroup needs its neighbour's
    // sums after its first few iterations and the neighbour delivers them after its last: every launch became a
    // chain — measured 4x slower than generation 1.)
    int f_tile = (it1 - 1) / nk, f_k = (it1 - 1) - f_tile * nk, f_stage = 0;
    int f_tap = 0, f_soffA = 0, f_soffB = 0, voffA = 0, voffB = 0;
    auto f_set_tile = [&]() {
        const int a = f_tile / gy, by = f_tile - a * gy;
        voffA = (grp + NGRP * a) * BM * 64 + lrow;
        voffB = by * BN * 64 + lrow;
    };
    const int cblk_last = taps > 1 ? (nk - 1) / taps : nk - 1, tap_last = nk - 1 - cblk_last * taps;
    const int soffA_last = cblk_last * kbA + tap_last * tapA, soffB_last = cblk_last * kbB + tap_last * tapB;
    auto f_set_k = [&]() {
        const int cblk = taps > 1 ? f_k / taps : f_k;
        f_tap = __builtin_amdgcn_readfirstlane(f_k - cblk * taps);
        f_soffA = __builtin_amdgcn_readfirstlane(cblk * kbA + f_tap * tapA);
        f_soffB = __builtin_amdgcn_readfirstlane(cblk * kbB + f_tap * tapB);
    };
    auto f_advance = [&]() {                   // everything wave-uniform: say so (no waterfall loops, T20)
        f_stage = __builtin_amdgcn_readfirstlane(f_stage == NST - 1 ? 0 : f_stage + 1);
        f_k = __builtin_amdgcn_readfirstlane(f_k - 1);
        if (f_k < 0) {                         // the last k-tile of the tile below
            f_k = nk - 1;
            f_tile = __builtin_amdgcn_readfirstlane(f_tile - 1);
            f_set_tile();
            f_tap = tap_last;
            f_soffA = soffA_last;
            f_soffB = soffB_last;
            return;
        }
        const bool wrap = f_tap == 0;
        f_tap = __builtin_amdgcn_readfirstlane(wrap ? taps - 1 : f_tap - 1);
        f_soffA = __builtin_amdgcn_readfirstlane(f_soffA - (wrap ? wrapA : tapA));
        f_soffB = __builtin_amdgcn_readfirstlane(f_soffB - (wrap ? wrapB : tapB));
    };
    char* const dA = smem + w * 1024;
    char* const dB = smem + 2 * PLANE_A + w * 1024;
    // The loop body is BRANCH-FREE: hipcc's wait-count pass gives up at every join behind a conditional LDS-DMA or
    // ds_read and waits lgkmcnt(0) there — the whole LDS latency in front of every MFMA pair (measured: 2x).  The
    // pieces of iterations beyond the workgroup's range are issued anyway, with an offset beyond the buffer: the
    // bounds check drops the fetch (zeros land in a stage nobody reads), and every iteration issues the same
    // number of pieces, so the counted vmcnt waits need no cases either.
    constexpr int OOB = 0x7f000000;
    int vA = 0, vB = 0;                        // voffA / voffB, or OOB
    auto piece = [&](const int j) {
        char* const st = (j < NPA ? dA : dB) + f_stage * STAGE;
        if (j < NPA) {
            const int lo = j >= MT, i = j - lo * MT;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA,
                (__attribute__((address_space(3))) void*)(st + lo * PLANE_A + i * 4096), 16, vA + i * 4096,
                lo ? f_soffA + loA : f_soffA, 0, 0);
        } else {
            const int jb = j - NPA, lo = jb >= 2, i = jb - lo * 2;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB,
                (__attribute__((address_space(3))) void*)(st + lo * PLANE_B + i * 4096), 16, vB + i * 4096,
                lo ? f_soffB + loB : f_soffB, 0, 0);
        }
    };

    // ---- MFMA coordinates: 2 x 2 waves, wave tile (32 MT) x 64 ------------------------------------------------------
    const int li = l & 31, g = l >> 5;
    const int wm = w >> 1, wn = w & 1;
    const int sw = (li >> 2) & 3;
    const int foff0 = li * 64 + (((0 + g) ^ sw) << 4), foff1 = li * 64 + (((2 + g) ^ sw) << 4);
    const char* const fa = smem + (wm * 32 * MT) * 64;
    const char* const fb = smem + 2 * PLANE_A + (wn * 64) * 64;
    f32x16 acc[MT][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    };
    struct Frags {
        f16x8 ah[MT], al[MT], bh[2], bl[2];
    };
    Frags S0, S1;
    auto read_one = [&](Frags& S, const int stage, const int foff, const int r) {
        const char* a = fa + stage * STAGE + foff;
        const char* bb = fb + stage * STAGE + foff;
        if (r < 2)
            S.bh[r] = *reinterpret_cast<const f16x8*>(bb + r * 2048);
        else if (r < 2 + MT)
            S.ah[r - 2] = *reinterpret_cast<const f16x8*>(a + (r - 2) * 2048);
        else if (r < 2 + 2 * MT)
            S.al[r - 2 - MT] = *reinterpret_cast<const f16x8*>(a + PLANE_A + (r - 2 - MT) * 2048);
        else
            S.bl[r - 2 - 2 * MT] = *reinterpret_cast<const f16x8*>(bb + PLANE_B + (r - 2 - 2 * MT) * 2048);
    };
    // one 16-wide k-step (k_gemm_g2.hip): 6 MT MFMAs; between them the fragment reads of the next k-step (`rd`)
    // and, after every `gap`-th MFMA, one LDS-DMA piece of the iteration being fetched (`dma`: pieces p0 ..)
    auto kstep = [&](const Frags& S, Frags& Sn, const int rstage, const int rfoff, const int p0, const int np) {
        f16x8 b2[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) b2[t] = S.bh[t] * (_Float16)2048.f;
        const int gap = NM / np;
        constexpr int rgap = NM / NR;
        int issued = 0, nread = 0;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int ty = m / (2 * MT), r = m - ty * 2 * MT, mt = r >> 1, nt = r & 1;
            const f16x8 bo = ty == 0 ? b2[nt] : ty == 1 ? S.bh[nt] : S.bl[nt];
            const f16x8 ao = ty == 1 ? S.al[mt] : S.ah[mt];
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bo, ao, acc[mt][nt], 0, 0, 0);
            if (nread < NR && (m + 1) % rgap == 0) {
                // UNCONDITIONAL (in the last iteration they fetch a stage nobody will use): hipcc counts the reads
                // of a conditional block as "maybe not issued" and then waits lgkmcnt(1), (0) for an OLDER fragment
                // right after issuing them — the whole LDS latency in front of every MFMA pair (measured: 2x)
                G3_PIN();
                read_one(Sn, rstage, rfoff, nread);
                G3_PIN();
                ++nread;
            }
            if (issued < np && (m + 1) % gap == 0) {
                G3_PIN();
                piece(p0 + issued);
                G3_PIN();
                ++issued;
            }
        }
#pragma unroll
        for (; issued < np; ++issued) piece(p0 + issued);
#pragma unroll
        for (; nread < NR; ++nread) read_one(Sn, rstage, rfoff, nread);
    };

    // ---- prologue: iterations it1 - 1 and it1 - 2 in flight, then the first fragments ---------------------------------
    f_set_tile();
    f_set_k();
    vA = voffA;
    vB = voffB;
#pragma unroll
    for (int j = 0; j < PPW; ++j) piece(j);
    f_advance();
    vA = it0 + 1 < it1 ? voffA : OOB;
    vB = it0 + 1 < it1 ? voffB : OOB;
#pragma unroll
    for (int j = 0; j < PPW; ++j) piece(j);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PPW) : "memory");
#pragma unroll
    for (int r = 0; r < NR; ++r) read_one(S0, 0, foff0, r);

    float* par = reinterpret_cast<float*>(smem + NST * STAGE);
    float amax = 0.f;
    // What only a segment's end needs (output pointers and strides, epilogue parameters, the workspace) is read
    // from the kernel-argument segment THERE, through a pointer the compiler cannot see through: loaded once at
    // the top they occupied ~30 scalar registers across the main loop, which then spilled its own (and wrapped
    // every LDS-DMA instruction in a waterfall loop over a scalar offset that had become a vector register).
    typedef const __attribute__((address_space(4))) DzConvGemm* KArg;
    typedef const __attribute__((address_space(4))) G3Args* KArg2;
    const KArg pk0 = (KArg)__builtin_amdgcn_kernarg_segment_ptr();

    // ---- main loop over this workgroup's iterations -----------------------------------------------------------------
    // (a loop over the workgroup's SEGMENTS — the k-tiles it holds of one tile — around the loop over their
    // k-tiles: the accumulators are defined at a segment's start and dead after its end, which the register
    // allocator handles; one flat loop with the epilogue inside it spilled them)
    int c_tile = (it1 - 1) / nk, s = 0, it = it1 - 1;
    bool landed = false;       // the iteration after an epilogue: its operands have landed already (drained there)
    while (it >= it0) {
    const int seg_lo = max(it0, c_tile * nk);                         // the segment's lowest iteration
    const int seg_hi = it;
    zero_acc();
    for (; it >= seg_lo; --it) {
        const bool more = it - 2 >= it0;                               // wave-uniform
        f_advance();
        vA = more ? voffA : OOB;
        vB = more ? voffB : OOB;
        kstep(S0, S1, s, foff1, 0, H);
        const int s1 = s == NST - 1 ? 0 : s + 1;
        // own pieces of the next iteration have landed (only the H pieces issued above may still fly; after an
        // epilogue its stores are in the queue as well: everything was drained there instead)
        if (landed)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(H) : "memory");
        G3_PIN();
        landed = false;
        kstep(S1, S0, s1, foff0, H, PPW - H);
        s = s1;
    }
        {
            // ---- end of this workgroup's segment of tile c_tile: iterations seg_hi down to seg_lo ---------------------
            const bool has_top = seg_hi == (c_tile + 1) * nk - 1;      // holds the tile's LAST k-tile: finishes the tile
            const bool has_bottom = seg_lo == c_tile * nk;
            KArg pk = pk0;
            asm volatile("" : "+s"(pk));
            const KArg2 gk = (KArg2)(pk + 1);                          // the second kernel argument follows the first
            if (!has_top) {
                // the tile's top belongs to a higher-numbered workgroup of the group: park the accumulators
                // Hand-over: plain 16-byte stores, agent-scope release, relaxed flag; the finishing workgroup polls the
                // flag (relaxed), then one agent-scope acquire and plain loads (cdna_hip_programming.md Guideline 16).
                // The cheaper forms the guide lists — sc1 stores + sc1 loads, or 8-byte relaxed agent-scope atomics,
                // with no fence — were built and returned WRONG sums here (the finishing workgroup read stale lines:
                // a workgroup's group is its XCD only if workgroups are dealt round-robin, and a coarse-grained
                // allocation is not coherent across XCD L2s without the write-back / invalidate the fences carry).
                f32x4* dst = reinterpret_cast<f32x4*>(gk->ws) + (size_t)b * (MT * 2 * 4 * 256) + tid;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            f32x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc[mt][nt][4 * k + e];
                            *dst = v;
                            dst += 256;
                            asm volatile("" : "+v"(dst));              // one running address, not 8 MT of them
                        }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(gk->flags + b, gk->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // operands of it + 1, it + 2: see `landed`
                landed = true;
                const int a = c_tile / gy, by = c_tile - a * gy;
                const int t0 = (grp + NGRP * a) * BM, n0 = by * BN;
                // epilogue parameters of this tile's columns -> LDS
                if (tid < BN) {
                    par[tid] = pk->bias[n0 + tid];
                    if (AFF) par[2 * BN + tid] = pk->e1[n0 + tid];
                } else if (AFF) {
                    par[tid] = pk->e0[n0 + tid - BN];
                }
                if (!has_bottom) {
                    // the tile's lower k-tiles were done by lower-numbered workgroup(s) of this group (first thing
                    // in their walk): add what they parked
                    const long long tile_it = (long long)c_tile * nk;          // first iteration of the tile
                    for (int j = idx - 1; j >= 0; --j) {
                        const long long r0 = I * j / nwg, r1 = I * (j + 1) / nwg;
                        if (r1 <= tile_it) break;
                        if (r0 < r1) {
                            const int bj = grp + NGRP * j;
                            if (tid == 0) {
                                int* const flags = gk->flags;
                                const int epoch = gk->epoch;
                                int spins = 0;
                                while (__hip_atomic_load(flags + bj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                                    __builtin_amdgcn_s_sleep(4);
                                    if (++spins > (1 << 22)) {              // ~1 s: never hang the device
                                        if (gk->err) *gk->err = 1;
                                        break;
                                    }
                                }
                                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                            }
                            __syncthreads();
                            const f32x4* src = reinterpret_cast<const f32x4*>(gk->ws) + (size_t)bj * (MT * 2 * 4 * 256) + tid;
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                                for (int nt = 0; nt < 2; ++nt) {
                                    // four loads in flight, not all 8 MT of them (128 registers at MT = 4)
                                    asm volatile("" : "+v"(src)::"memory");
#pragma unroll
                                    for (int k = 0; k < 4; ++k) {
                                        const f32x4 v = src[k * 256];
#pragma unroll
                                        for (int e = 0; e < 4; ++e) acc[mt][nt][4 * k + e] += v[e];
                                    }
                                    src += 4 * 256;
                                }
                        }
                        if (r0 <= tile_it) break;
                    }
                }
                __syncthreads();                                       // par is complete
                // C/D map of the transposed product: column = lane & 31 = output ROW, register r = output column
                // n0' + (r & 3) + 8 (r >> 2) + 4 (lane >> 5): k_gemm_pre.hip
                float* const Yf = pk->Y;
                unsigned short* const Yhi = reinterpret_cast<unsigned short*>(pk->Ysplit);
                const long long yplane = pk->yplane;
                const int ldy = pk->ldy, Nstore = pk->Nstore, Tout = pk->Tout;
                const long long yrows = Yhi ? (long long)((unsigned)yplane / (unsigned)ldy) : 0;
                const bool fulln = n0 + BN <= Nstore;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int t = t0 + wm * 32 * MT + mt * 32 + li;
                    const bool ok = t < Tout;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            asm volatile("" ::: "memory");
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
                                    if (n + e >= Nstore) v[e] = 0.f;
                            }
                            if (Yf && ok) {
                                float* y = Yf + (long long)t * ldy + n;
                                if (fulln || n + 3 < Nstore) {
                                    *reinterpret_cast<f32x4*>(y) = v;
                                } else {
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        if (n + e < Nstore) y[e] = v[e];
                                }
                            }
                            if (Yhi) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    if (ok) amax = fmaxf(amax, fabsf(v[e]));
                                    v[e] = __builtin_amdgcn_fmed3f(v[e], -65504.f, 65504.f);
                                }
                                const f16x4 hi = __builtin_convertvector(v, f16x4);
                                const f16x4 lo =
                                    __builtin_convertvector((v - __builtin_convertvector(hi, f32x4)) * 2048.f, f16x4);
                                if (ok) {
                                    const long long kidx = dz_kb(t, n, yrows);
                                    *reinterpret_cast<f16x4*>(Yhi + kidx) = hi;
                                    *reinterpret_cast<f16x4*>(Yhi + yplane + kidx) = lo;
                                }
                            }
                        }
                }
                __syncthreads();                                       // par may be overwritten by the next tile
            }
        }
        --c_tile;
    }
    {
        KArg pk = pk0;
        asm volatile("" : "+s"(pk));
        dz_flag_range(pk->oflag, amax);
    }
}

// ---- workspaces: one per (device, stream) — launches on one stream are ordered, so they may share it ------------
struct G3Workspace {
    float* ws = nullptr;
    int* flags = nullptr;
    int epoch = 0;
    size_t bytes = 0;
    int grid = 0;
};
std::mutex g3_mu;
std::unordered_map<unsigned long long, G3Workspace> g3_ws;
int* g3_err_dev[64] = {};
int* g3_err_host[64] = {};

int cu_count() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus > 0 ? cus : 256;
}

template <int EPI, int MT>
int launch_mt(const DzConvGemm& p, hipStream_t st) {
    static DzAttrOnce attr_once;
    DZ_HIP(attr_once.raise((const void*)gemm_g3_kernel<EPI, MT>, (int)lds_bytes(MT)));
    int dev = 0;
    DZ_HIP(hipGetDevice(&dev));
    const int grid = cu_count();
    G3Args ga;
    {
        std::lock_guard<std::mutex> lk(g3_mu);
        if (dev >= 0 && dev < 64 && !g3_err_dev[dev]) {
            DZ_HIP(hipHostMalloc((void**)&g3_err_host[dev], sizeof(int), hipHostMallocMapped));
            *g3_err_host[dev] = 0;
            DZ_HIP(hipHostGetDevicePointer((void**)&g3_err_dev[dev], g3_err_host[dev], 0));
        }
        G3Workspace& W = g3_ws[((unsigned long long)dev << 56) ^ (unsigned long long)(uintptr_t)st];
        const size_t need = (size_t)grid * ws_bytes_per_wg(4);           // sized for the largest tile
        if (W.bytes < need || W.grid < grid) {
            // (first launch on this stream: StreamBatch's warm-up steps take this allocation, not a live step)
            if (W.ws) (void)hipFree(W.ws);
            if (W.flags) (void)hipFree(W.flags);
            DZ_HIP(hipMalloc((void**)&W.ws, need));
            DZ_HIP(hipMalloc((void**)&W.flags, grid * sizeof(int)));
            DZ_HIP(hipMemsetAsync(W.flags, 0, grid * sizeof(int), st));
            W.bytes = need;
            W.grid = grid;
            W.epoch = 0;
        }
        ga.ws = W.ws;
        ga.flags = W.flags;
        ga.epoch = ++W.epoch;
        ga.err = dev >= 0 && dev < 64 ? g3_err_dev[dev] : nullptr;
    }
    DZ_LAUNCH((gemm_g3_kernel<EPI, MT>), dim3(grid), dim3(256), lds_bytes(MT), st, p, ga);
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
    dz_set_error("gemm_g3: row fragments per wave must be 2, 3 or 4 (got %d)", mt);
    return 2;
}

}  // namespace

// 1 when a workgroup of a generation-3 launch on this device gave up waiting for its neighbour's partial sums
// (results of that launch are wrong); reset != 0 clears it
int dz_g3_error(int reset) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || !g3_err_host[dev]) return 0;
    const int e = *(volatile int*)g3_err_host[dev];
    if (reset) *g3_err_host[dev] = 0;
    return e;
}

// DZ_G3_MT: row fragments per wave (2, 3, 4 -> 128 / 192 / 256 x 128 tiles); default 4
int dz_g3_default_mt() {
    static const int mt = [] {
        const char* e = getenv("DZ_G3_MT");
        const int v = e ? atoi(e) : 4;
        return v >= 2 && v <= 4 ? v : 4;
    }();
    return mt;
}

// requirements are those of dz_launch_gemm_pre (k_gemm_pre.hip), which checks them and dispatches here
int dz_launch_gemm_g3(const DzConvGemm& p, int mt, hipStream_t st) {
    if (mt <= 0) mt = dz_g3_default_mt();
    switch (p.epi) {
        case DZ_EPI_BIAS: return launch<DZ_EPI_BIAS>(p, mt, st);
        case DZ_EPI_BIAS_LEAKY: return launch<DZ_EPI_BIAS_LEAKY>(p, mt, st);
        case DZ_EPI_TDNN: return launch<DZ_EPI_TDNN>(p, mt, st);
        case DZ_EPI_RELU_BN: return launch<DZ_EPI_RELU_BN>(p, mt, st);
    }
    dz_set_error("gemm_g3: epilogue %d is not built on the pre-split path", p.epi);
    return 2;
}
