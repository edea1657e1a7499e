// Implicit-GEMM 1-D convolution / linear layer on the f16 matrix cores, BOTH operands pre-split.
//
//   Y[t][n] = epi( sum_{tap,c} X[t + tap*dil][c] * W[n][tap*Cin + c] + bias[n] )
//
// Same contraction and arithmetic as k_gemm_split.hip (x = hi + lo * 2^-11 with hi, lo f16; three
// v_mfma_f32_32x32x16_f16 per product into two f32 accumulators), for the wide layers in the
// middle of the networks (x-vector tdnn2..5, the LSTM input projections of layers 1..3, the
// segmentation MLP; third-party graphs called from /root/reference/src/diart/models.py:133, :262;
// SURVEY.md Appendix A, kernels K5 / K6 / K8).  What changed is where the split happens:
//
//   * k_gemm_split.hip reads f32 activations and splits them on the way into LDS: ~4 VALU
//     instructions per element, once per N-tile that re-reads the row (4..12 times), plus four
//     ds_write_b128 per thread per k-tile — at 29 % matrix-core busy the loop was bound by exactly
//     this traffic (profiles/r01_p_*).
//   * here the PRODUCER's epilogue writes the two f16 planes (4 bytes per element, the same HBM
//     bytes as the f32 it replaces; split once per element) in the "kb-major" order of dz_kb()
//     (dz_common.h: [K / 32][rows][32] — the weights are packed the same way by weights.py), so a
//     k-tile of either operand is plain, CONTIGUOUS bytes: it goes global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 16 B per lane, no
//     VGPR staging, no ds_write, no VALU; the k-tile / tap offset is the instruction's scalar
//     offset, out-of-range rows read as zeros through the buffer bounds check).
//
// Tile 128 x 128 x 32, 4 waves (2 x 2), wave tile 64 x 64 = 2 x 2 fragments of 32 x 32: per 16-wide
// k-step a wave issues 8 ds_read_b128 for 12 MFMAs (k_gemm_split: 6 for 6).  LDS stage = four
// planes [128 rows][64 B] (A hi, A lo, B hi, B lo), 16-byte chunks XOR-swizzled with (row >> 2) & 3
// (the LDS-DMA destination is lane-linear, so the swizzle is applied to the per-lane SOURCE chunk
// and again on the fragment read: cdna_hip_programming.md rule 21); two stages = 64 KiB, two
// workgroups per CU.  One barrier per k-tile: wait own DMA, barrier, issue the DMA of tile kt+1
// into the stage everybody has just finished reading, compute tile kt.
//
// Small tiles.  The same tile code also runs with one 32 x 32 fragment per wave (64 x 64 tiles: a
// quarter of the work per workgroup at twice the operand bytes per MFMA).  A launch whose 128 x 128
// tiles would occupy less than a quarter of the chip's 512 workgroup slots (single-chunk latency
// runs: 12 tiles for one 5 s chunk) uses them throughout.  DZ_GEMM_TAIL=1 additionally uses them for
// the rows of a mostly empty LAST round (config 2: tdnn2 has 580 tiles = 1.13 rounds, the LSTM
// projection 1172 = 2.29).  Measured (gpurun_out r02 visit "tail"): alone the kernels get 13 - 16 %
// faster (tdnn2 128 -> 111 us, tdnn4 50 -> 42 us), but the 64-stream pipeline gets ~1 % SLOWER — there
// the empty slots of a last round are filled by the kernels of the other HIP streams, and the small
// tiles only add operand traffic.  Hence off by default.
#include "dz_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, KT = 32;
constexpr int PLANE_B = 128 * 64;          // bytes of one f16 plane of the weight tile of a stage
// MW = waves along M (wave tile 64 x 64, two wave columns): the activation planes of a stage hold
// 64 MW rows.  Stage = A hi | A lo | B hi | B lo; after the two stages: bias | e0 | e1, [3][128] f32.
constexpr int plane_a(int MW) { return 64 * MW * 64; }
constexpr int stage_bytes(int MW) { return 2 * plane_a(MW) + 2 * PLANE_B; }
constexpr size_t lds_bytes(int MW) { return 2 * (size_t)stage_bytes(MW) + 3 * BN * sizeof(float); }
constexpr int MW_BIG = 6;                  // 384 x 128 tiles, 12 waves, one workgroup per CU
constexpr float LO_UNSCALE = 1.f / 2048.f;

// (slope < 1: max(v, slope v) is v for v > 0 and slope v otherwise — the same values as the select, one instruction less)
__device__ __forceinline__ float leaky(float v) { return fmaxf(v, v * DZ_LEAKY_SLOPE); }

// One output tile of (32 MT MW) x (64 NT): MW x 2 waves, MT x NT fragments of 32 x 32 per wave.
// LDS of the pooled epilogue (after the k-loop): the f32 output tile [128][YT_PITCH] over the two stages,
// then the epilogue parameters, the tile rows' pooling weights and the reduction scratch
constexpr int YT_PITCH = 132;                                   // floats; = 4 mod 32: conflict-free 16-byte row writes
constexpr int POOL_PAR = BM * YT_PITCH * 4;                     // bias | e0 | e1
constexpr int POOL_WT = POOL_PAR + 3 * BN * 4;                  // [128 rows][4 speakers] weights of the tile's rows
constexpr int POOL_RED = POOL_WT + 4 * BM * 4;                  // [4 waves][4 speakers][128] partial sums of one pass
constexpr int POOL_S0 = POOL_RED + 2 * 2 * 4 * BN * 4;          // [8 (part, k)][8 sub-ranges][2], then [8][2]
constexpr size_t POOL_LDS = POOL_S0 + 8 * 8 * 2 * 4 + 8 * 2 * 4;

template <int EPI, int MW, int MT, int NT, bool ILV = false, bool POOL = false>
__device__ __forceinline__ void gemm_pre_tile(const DzConvGemm& p, const int t0, const int n0, char* smem,
                                              const int flags = 0, const DzPoolFuse* q = nullptr) {
    constexpr int PLANE_A = plane_a(MW), STAGE = stage_bytes(MW), PAR_OFF = POOL ? POOL_PAR : 2 * STAGE;
    static_assert(!POOL || (MW == 2 && MT == 2 && NT == 2), "pooled epilogue: 128 x 128 tiles only");
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;

    // ---- staging: 16-row pieces of the four planes (A hi, A lo, B hi, B lo) by LDS-DMA ----------
    // lane -> (row l>>2 of the piece, LDS slot l&3); the slot holds source chunk slot ^ ((row >> 2) & 3),
    // and (row >> 2) & 3 = (l >> 4) & 3 for every piece.  The first NWA waves share the pieces of the
    // two activation planes, the others those of the two weight planes (wave of rank r in a group of
    // G takes pieces r, r + G, ... of "hi pieces, then lo pieces"); with 4 waves and a full tile that
    // is one plane per wave.
    constexpr int PA = 2 * MT * MW, PB = 4 * NT, NWAVE = 2 * MW;
    constexpr int NWA = (NWAVE * PA + (PA + PB) / 2) / (PA + PB);          // waves on the activation planes
    constexpr int JMAX_A = (2 * PA + NWA - 1) / NWA, JMAX_B = (2 * PB + (NWAVE - NWA) - 1) / (NWAVE - NWA);
    constexpr int JMAX = JMAX_A > JMAX_B ? JMAX_A : JMAX_B;
    const bool isB = w >= NWA;
    const int rank = isB ? w - NWA : w, group = isB ? NWAVE - NWA : NWA, npc = isB ? PB : PA;
    const unsigned short* src = isB ? reinterpret_cast<const unsigned short*>(p.Wsplit)
                                    : reinterpret_cast<const unsigned short*>(p.Xsplit);
    // kb-major planes (dz_common.h, dz_kb): plane = [K / 32][rows][32], so the 32-wide k-tile of 16
    // consecutive rows — one LDS-DMA piece — is 1 KiB of contiguous memory (8 full cache lines; row-major
    // planes gave 16 half lines per piece and cost the loop 8 - 16 %, round 3).  A row beyond the plane's
    // rows reads the start of the next k-block (finite values of other rows) or, in the last block, zeros
    // through the buffer bounds check: such rows only feed outputs that are never stored.
    const int prows = isB ? p.Npad : (int)((unsigned)p.xplane / (unsigned)p.ldx);        // rows of a plane
    const long long plane_el = isB ? (long long)p.Npad * p.Kpad : p.xplane;     // f16 elements between hi and lo
    const unsigned nbytes = (unsigned)((long long)prows * (isB ? p.Kpad : p.ldx) * 2);
    const __amdgpu_buffer_rsrc_t rs_hi = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(src + plane_el), 0, nbytes, 0x00020000);
    const int voff0 = ((isB ? n0 : t0) + (l >> 2)) * 64 + (((l & 3) ^ ((l >> 4) & 3)) << 4);
    constexpr int vstep = 16 * 64;
    const int kb_bytes = prows * 64;                                            // one k-block of the plane
    const int dst0 = isB ? 2 * PLANE_A : 0, dplane = isB ? PLANE_B : PLANE_A;
    auto issue = [&](int kt, int stage) {
        // k-tile kt of the loop = (channel block kt / taps, tap kt % taps): the taps of one channel block are
        // consecutive, so the three reads of (almost) the same activation lines — rows t, t + dil, t + 2 dil of
        // one k-block — follow each other while the lines are still in the L2.  In the tap-major order of the
        // weight matrix's K axis they were a third of the loop apart, and with 16 row tiles in flight per XCD
        // (4.3 MB of activations beside 3 MB of weights in a 4 MB L2) each sweep fetched them from HBM again:
        // FETCH_SIZE of tdnn2 - 4 was 2.05x the algorithmic bytes.  (The weights keep their layout: k-block
        // tap * Cin / 32 + channel block.)
        int cblk = kt, tap = 0;
        if (p.taps > 1) {
            cblk = kt / p.taps;
            tap = kt - cblk * p.taps;
        }
        const int soff = isB ? (tap * (p.Cin >> 5) + cblk) * kb_bytes : cblk * kb_bytes + tap * p.dil * 64;
        char* dst = smem + stage * STAGE + dst0;
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            const int q = rank + j * group;                    // wave-uniform
            if (q >= 2 * npc) break;
            const int lo = q >= npc, i = q - lo * npc;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(lo ? rs_lo : rs_hi,
                (__attribute__((address_space(3))) void*)(dst + lo * dplane + i * 1024), 16,
                voff0 + i * vstep, soff, 0, 0);
        }
    };

    // ---- MFMA coordinates: MW x 2 waves, wave tile (32 MT) x (32 NT) ----------------------------
    const int li = l & 31, g = l >> 5;
    const int wm = w >> 1, wn = w & 1;
    const int sw = (li >> 2) & 3;
    int foff[2];                               // fragment offset of this lane inside a 32-row block
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = li * 64 + (((2 * ks + g) ^ sw) << 4);
    f32x16 accm[MT][NT], accx[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) accm[mt][nt][r] = accx[mt][nt][r] = 0.f;

    auto compute = [&](int stage) {
        const char* st = smem + stage * STAGE;
        const char* sa = st + (wm * 32 * MT) * 64;
        const char* sb = st + 2 * PLANE_A + (wn * 32 * NT) * 64;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                ah[t] = *reinterpret_cast<const f16x8*>(sa + t * 2048 + foff[ks]);
                al[t] = *reinterpret_cast<const f16x8*>(sa + PLANE_A + t * 2048 + foff[ks]);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                bh[t] = *reinterpret_cast<const f16x8*>(sb + t * 2048 + foff[ks]);
                bl[t] = *reinterpret_cast<const f16x8*>(sb + PLANE_B + t * 2048 + foff[ks]);
            }
            // TRANSPOSED product: the weight fragment is the MFMA's row operand, the activation
            // fragment its column operand, so a lane ends up with ONE output row t (= lane & 31)
            // and, per accumulator register group, FOUR CONSECUTIVE output columns: the epilogue
            // stores 16 bytes (f32) / 8 bytes (f16 planes) per instruction instead of 4
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    accx[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[nt], al[mt], accx[mt][nt], 0, 0, 0);
                    accm[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[nt], ah[mt], accm[mt][nt], 0, 0, 0);
                    accx[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[nt], ah[mt], accx[mt][nt], 0, 0, 0);
                }
        }
    };

    const int nk = p.Kpad / KT;
    issue(0, 0);
    constexpr bool FULL = MW == 2 && MT == 2 && NT == 2;     // 128 x 128 tile, 4 waves: the interleaved loop below
    // The epilogue's per-column parameters go to LDS now: fetched from global memory inside the
    // epilogue they were 16 dependent round trips per lane (three 16-byte loads per column group,
    // no registers left to hoist them into) — ~20 % of a tile's time with nothing else to run.
    float* par = reinterpret_cast<float*>(smem + PAR_OFF);
    {
        constexpr bool AFF = EPI == DZ_EPI_TDNN || EPI == DZ_EPI_RELU_BN;
        const int which = tid >> 5, c4 = (tid & 31) * 4;
        if (which < (AFF ? 3 : 1)) {
            const float* src = which == 0 ? p.bias : which == 1 ? p.e0 : p.e1;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (n0 + c4 + 3 < p.Npad) v = *reinterpret_cast<const f32x4*>(src + n0 + c4);
            *reinterpret_cast<f32x4*>(par + which * BN + c4) = v;
        }
    }
    if (FULL && ILV) {
        // ---- interleaved k-loop (round 3) -------------------------------------------------------------
        // The plain loop below issues the 8 LDS-DMA pieces of tile kt+1 in one block at the top of
        // tile kt: an LDS-DMA instruction occupies its wave's issue port for 60 - 185 cycles
        // (MI355X_MICROARCH.md, cycle constants), i.e. ~1000 cycles per k-tile during which this wave
        // issues no MFMA — more than the 768 cycles its 24 MFMAs take.  Here one piece follows every
        // second MFMA of the first 16 (the last 8 MFMAs are the landing slack before the next
        // top-of-loop wait), and a fragment is re-read for the next 16-wide k-step as soon as its last
        // MFMA has been issued (same 32 fragment registers, rotated).  The order is pinned with
        // sched_barrier; no branches inside the loop body (piece -> plane / row block is a
        // compile-time function of j for 2 + 2 waves; the tap / channel of the activation offset
        // advances incrementally instead of by a division per tile).
        const int r2 = w & 1;                                       // rank inside the pair of waves of a side
        char* const dbase = smem + dst0 + r2 * 1024;
        const int vofs = voff0 + r2 * vstep;
        int cblk = 0, tap = 0;                                      // (channel block, tap) of tile kt + 1, see issue()
        const int tap_step = isB ? (p.Cin >> 5) * kb_bytes : p.dil * 64;
        auto advance = [&]() -> int {                               // -> soffset (bytes) of the NEXT tile
            if (++tap == p.taps) { tap = 0; ++cblk; }
            return cblk * kb_bytes + tap * tap_step;
        };
        int soff_next = 0, stage_next = 0;
        const bool no_dma = flags & 2, no_rd = flags & 4;           // TIMING EXPERIMENTS (DZ_GP_DBG, wrong results)
        const bool skip_a = (flags & 32) && !isB;                   // ... 32: the activation pieces of taps > 0 are not fetched (what re-using one fetch per channel block would save)
        auto piece = [&](int j) {                                   // j = 0..7: pieces r2 + 2j of "8 hi, then 8 lo"
            const int lo = j >= 4, i = r2 + 2 * (j & 3);            // i-th 16-row block of the plane (r2 folded into bases)
            (void)i;
            if (no_dma || (skip_a && tap != 0 && cblk > 0)) return;       // (the first channel block fills both stages with finite data)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(lo ? rs_lo : rs_hi,
                (__attribute__((address_space(3))) void*)(dbase + stage_next * STAGE + lo * dplane + (j & 3) * 2048), 16,
                vofs + (j & 3) * 2 * vstep, soff_next, 0, 0);
        };
        const char* sa0 = smem + (wm * 64) * 64;
        const char* sb0 = smem + 2 * PLANE_A + (wn * 64) * 64;
#define DZ_RD(base, plane, t, ks) (*reinterpret_cast<const f16x8*>((no_rd ? smem : (base) + (plane) + (t) * 2048) + foff[ks]))
#define DZ_MM(mt, nt)                                                                                     \
    accx[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[nt], al[mt], accx[mt][nt], 0, 0, 0);          \
    accm[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[nt], ah[mt], accm[mt][nt], 0, 0, 0);          \
    accx[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[nt], ah[mt], accx[mt][nt], 0, 0, 0)
#define DZ_PIN() __builtin_amdgcn_sched_barrier(0)
        // `more` is a compile-time constant: the last tile (nothing left to fetch) is peeled off, so the
        // body has no branches around the pieces
        auto body = [&](const int kt, auto more_c) {
            constexpr bool more = decltype(more_c)::value;
            // own DMA of tile kt has landed + every wave has finished the fragment reads of tile kt-1
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            soff_next = advance();
            stage_next = (kt + 1) & 1;
            const char* sa = sa0 + (kt & 1) * STAGE;
            const char* sb = sb0 + (kt & 1) * STAGE;
            f16x8 ah[2], al[2], bh[2], bl[2];
            bh[0] = DZ_RD(sb, 0, 0, 0); al[0] = DZ_RD(sa, PLANE_A, 0, 0); ah[0] = DZ_RD(sa, 0, 0, 0);
            bl[0] = DZ_RD(sb, PLANE_B, 0, 0);
            bh[1] = DZ_RD(sb, 0, 1, 0); bl[1] = DZ_RD(sb, PLANE_B, 1, 0);
            ah[1] = DZ_RD(sa, 0, 1, 0); al[1] = DZ_RD(sa, PLANE_A, 1, 0);
            DZ_PIN();
            // ---- k-step 0: (0,0) (0,1) (1,1) (1,0) ----
            accx[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[0], al[0], accx[0][0], 0, 0, 0);
            accm[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[0], ah[0], accm[0][0], 0, 0, 0);
            DZ_PIN();
            if (more) piece(0);
            DZ_PIN();
            accx[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[0], ah[0], accx[0][0], 0, 0, 0);
            accx[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[1], al[0], accx[0][1], 0, 0, 0);
            DZ_PIN();
            if (more) piece(1);
            DZ_PIN();
            accm[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[1], ah[0], accm[0][1], 0, 0, 0);
            accx[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[1], ah[0], accx[0][1], 0, 0, 0);
            DZ_PIN();
            if (more) piece(2);
            ah[0] = DZ_RD(sa, 0, 0, 1); al[0] = DZ_RD(sa, PLANE_A, 0, 1);      // row block 0 is done with k-step 0
            DZ_PIN();
            accx[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[1], al[1], accx[1][1], 0, 0, 0);
            accm[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[1], ah[1], accm[1][1], 0, 0, 0);
            DZ_PIN();
            if (more) piece(3);
            DZ_PIN();
            accx[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[1], ah[1], accx[1][1], 0, 0, 0);
            accx[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[0], al[1], accx[1][0], 0, 0, 0);
            DZ_PIN();
            if (more) piece(4);
            bh[1] = DZ_RD(sb, 0, 1, 1); bl[1] = DZ_RD(sb, PLANE_B, 1, 1);      // column block 1 is done
            DZ_PIN();
            accm[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[0], ah[1], accm[1][0], 0, 0, 0);
            accx[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[0], ah[1], accx[1][0], 0, 0, 0);
            DZ_PIN();
            if (more) piece(5);
            bh[0] = DZ_RD(sb, 0, 0, 1); bl[0] = DZ_RD(sb, PLANE_B, 0, 1);
            ah[1] = DZ_RD(sa, 0, 1, 1); al[1] = DZ_RD(sa, PLANE_A, 1, 1);
            DZ_PIN();
            // ---- k-step 1: (0,1) (0,0) (1,0) (1,1): operands in the order they were re-read ----
            accx[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[1], al[0], accx[0][1], 0, 0, 0);
            accm[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[1], ah[0], accm[0][1], 0, 0, 0);
            DZ_PIN();
            if (more) piece(6);
            DZ_PIN();
            accx[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[1], ah[0], accx[0][1], 0, 0, 0);
            accx[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[0], al[0], accx[0][0], 0, 0, 0);
            DZ_PIN();
            if (more) piece(7);
            DZ_PIN();
            accm[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[0], ah[0], accm[0][0], 0, 0, 0);
            accx[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[0], ah[0], accx[0][0], 0, 0, 0);
            DZ_MM(1, 0);
            DZ_MM(1, 1);
        };
        for (int kt = 0; kt + 1 < nk; ++kt) body(kt, std::true_type{});
        body(nk - 1, std::false_type{});
#undef DZ_RD
#undef DZ_MM
#undef DZ_PIN
    } else
    for (int kt = 0; kt < nk; ++kt) {
        // own DMA of tile kt has landed + every wave has finished the fragment reads of tile kt-1
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        compute(kt & 1);
    }

    // ---- epilogue.  C/D map of the transposed product: column = lane & 31 = output ROW t, register r
    // = output column n0' + (r & 3) + 8 (r >> 2) + 4 (lane >> 5): registers 4k .. 4k+3 are four
    // consecutive columns -> one 16-byte f32 store, or one 8-byte store per f16 plane (hi = f16(v),
    // lo = f16((v - hi) * 2^11), clamped to +-65504 and flagged beyond, like dz_store_split).
    // Columns >= Nstore of a padded layer are written as zeros to the planes (K padding of the
    // consumer) and not at all to the f32 output.
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    unsigned short* Yhi = reinterpret_cast<unsigned short*>(p.Ysplit);
    const long long yrows = Yhi ? (long long)((unsigned)p.yplane / (unsigned)p.ldy) : 0;   // rows of an output plane
    float amax = 0.f;
    float* yt = reinterpret_cast<float*>(smem);
    if (POOL) __syncthreads();        // every wave is done with the last k-tile: the tile buffer reuses the stages
    // Round 6: the common cases — the whole tile inside the stored columns, ONE kind of output — without a single
    // per-element test.  The general loop below carries eight compare / select pairs, four exec-masked regions and
    // two 64-bit address computations per group of four columns (2 000 instructions for the 64 values of a lane,
    // run beside the other workgroup's MFMAs on the same issue ports: plane output cost 10 % of a tdnn layer over
    // f32 output).  Here the stores are buffer stores: the k-block of a column group is a scalar offset, the four
    // groups of a 32-column block immediates, a row beyond Tout an out-of-range offset the hardware drops.
    if constexpr (!POOL) {
        const bool whole = n0 + 64 * NT <= p.Nstore;
        const long long ybytes = Yhi ? 2 * (long long)yrows * p.ldy : 4 * (long long)p.Tout * p.ldy;
        if (whole && (Yhi != nullptr) != (p.Y != nullptr) && ybytes < (1ll << 31) && !(flags & 64)) {
            const unsigned nrec = __builtin_amdgcn_readfirstlane((unsigned)ybytes);
            const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(Yhi ? (void*)Yhi : (void*)p.Y, 0, nrec, 0x00020000);
            const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(Yhi ? (void*)(Yhi + p.yplane) : (void*)p.Y, 0, nrec, 0x00020000);
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const float* parl = par + wn * 32 * NT + 4 * g;
            // KIND 0: f32 output; 1: planes, one 8-byte store per group and plane; 2 (default): planes, the lanes of a
            // row (g = 0, 1) exchange every second group through v_permlane32_swap so that each holds EIGHT consecutive
            // columns: one 16-byte store per pair of groups and plane, 32 contiguous bytes per row and instruction
            auto sweep = [&](auto kind_c) {
                constexpr int KIND = decltype(kind_c)::value;
                constexpr bool PL = KIND != 0;
                u32x2 ph = {0u, 0u}, pw = {0u, 0u};
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int t = t0 + wm * 32 * MT + mt * 32 + li;
                const bool ok = t < p.Tout;
                float am = 0.f;
                // planes: ((k-block) * rows + t) * 64 bytes + (column & 31) * 2; f32: (t * ldy + column) * 4
                const int vo = !ok ? (int)0x80000000 : KIND == 2 ? t * 64 + 16 * g : KIND == 1 ? t * 64 + 8 * g : (t * p.ldy + 4 * g) * 4;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int cb = n0 + wn * 32 * NT + nt * 32;                  // first column of the 32-column block
                    const int so = __builtin_amdgcn_readfirstlane(PL ? (cb >> 5) * (int)yrows * 64 : cb * 4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        asm volatile("" ::: "memory");
                        const float* pg = parl + nt * 32 + 8 * k;        // one base register, immediate offsets
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(pg);
                        f32x4 e0 = {1.f, 1.f, 1.f, 1.f}, e1 = {0.f, 0.f, 0.f, 0.f};
                        if (EPI == DZ_EPI_TDNN || EPI == DZ_EPI_RELU_BN) {
                            e0 = *reinterpret_cast<const f32x4*>(pg + BN);
                            e1 = *reinterpret_cast<const f32x4*>(pg + 2 * BN);
                        }
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float x = (accm[mt][nt][4 * k + e] + accx[mt][nt][4 * k + e] * LO_UNSCALE) + bv[e];
                            if (EPI == DZ_EPI_BIAS_LEAKY) x = leaky(x);
                            if (EPI == DZ_EPI_TDNN) x = leaky(x) * e0[e] + e1[e];
                            if (EPI == DZ_EPI_RELU_BN) x = fmaxf(x, 0.f) * e0[e] + e1[e];
                            v[e] = x;
                        }
                        if constexpr (!PL) {
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r0, vo + 32 * k, so, 0);
                        } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            am = fmaxf(am, fabsf(v[e]));
                            v[e] = __builtin_amdgcn_fmed3f(v[e], -65504.f, 65504.f);
                        }
                        asm volatile("" : "+v"(am));        // (or the maxima sink into `if (ok)` below and every group's values stay live)
                        const f16x4 hi = __builtin_convertvector(v, f16x4);
                        const f16x4 lo = __builtin_convertvector((v - __builtin_convertvector(hi, f32x4)) * 2048.f, f16x4);
                        if constexpr (KIND == 1) {
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), r0, vo + 16 * k, so, 0);
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), r1, vo + 16 * k, so, 0);
                        } else if ((k & 1) == 0) {
                            ph = __builtin_bit_cast(u32x2, hi);
                            pw = __builtin_bit_cast(u32x2, lo);
                        } else {
                            // even group in ph / pw, odd group here: the upper half-wave's even group <-> the lower's odd
                            const u32x2 ch = __builtin_bit_cast(u32x2, hi), cw = __builtin_bit_cast(u32x2, lo);
                            u32x4 oh, ow;
#pragma unroll
                            for (int d = 0; d < 2; ++d) {
                                const auto sh = __builtin_amdgcn_permlane32_swap(ph[d], ch[d], false, false);
                                const auto sw2 = __builtin_amdgcn_permlane32_swap(pw[d], cw[d], false, false);
                                oh[d] = sh[0]; oh[2 + d] = sh[1];
                                ow[d] = sw2[0]; ow[2 + d] = sw2[1];
                            }
                            __builtin_amdgcn_raw_buffer_store_b128(oh, r0, vo + 32 * (k >> 1), so, 0);
                            __builtin_amdgcn_raw_buffer_store_b128(ow, r1, vo + 32 * (k >> 1), so, 0);
                        }
                        }
                    }
                }
                if (ok) amax = fmaxf(amax, am);
            }
            };
            if (!Yhi) sweep(std::integral_constant<int, 0>{});
            else if (flags & 128) sweep(std::integral_constant<int, 1>{});
            else sweep(std::integral_constant<int, 2>{});
            dz_flag_range(p.oflag, amax);
            return;
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int t = t0 + wm * 32 * MT + mt * 32 + li;
        const bool ok = t < p.Tout;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // keep the parameter reads of one column group next to their use: hoisted over all 16
                // groups they cost 45 registers (209 instead of 164), and at more than 176 a workgroup
                // of this kernel can no longer share a CU with a recurrence workgroup (2 x 168 per SIMD)
                asm volatile("" ::: "memory");
                const int n = n0 + wn * 32 * NT + nt * 32 + 8 * k + 4 * g;
                const int nc = n - n0;                       // column inside the tile
                const f32x4 bv = *reinterpret_cast<const f32x4*>(par + nc);
                f32x4 e0 = {1.f, 1.f, 1.f, 1.f}, e1 = {0.f, 0.f, 0.f, 0.f};
                if (EPI == DZ_EPI_TDNN || EPI == DZ_EPI_RELU_BN) {
                    e0 = *reinterpret_cast<const f32x4*>(par + BN + nc);
                    e1 = *reinterpret_cast<const f32x4*>(par + 2 * BN + nc);
                }
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = (accm[mt][nt][4 * k + e] + accx[mt][nt][4 * k + e] * LO_UNSCALE) + bv[e];
                    if (EPI == DZ_EPI_BIAS_LEAKY) x = leaky(x);
                    if (EPI == DZ_EPI_TDNN) x = leaky(x) * e0[e] + e1[e];
                    if (EPI == DZ_EPI_RELU_BN) x = fmaxf(x, 0.f) * e0[e] + e1[e];
                    v[e] = x;
                }
                if (POOL) {
                    if (!(flags & 16)) *reinterpret_cast<f32x4*>(yt + (t - t0) * YT_PITCH + nc) = v;
                    continue;
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
                        if (ok) amax = fmaxf(amax, fabsf(v[e]));
                        v[e] = __builtin_amdgcn_fmed3f(v[e], -65504.f, 65504.f);
                    }
                    const f16x4 hi = __builtin_convertvector(v, f16x4);
                    const f16x4 lo = __builtin_convertvector((v - __builtin_convertvector(hi, f32x4)) * 2048.f, f16x4);
                    if (ok) {
                        const long long kidx = dz_kb(t, n, yrows);     // four columns of one k-block row
                        *reinterpret_cast<f16x4*>(Yhi + kidx) = hi;
                        *reinterpret_cast<f16x4*>(Yhi + p.yplane + kidx) = lo;
                    }
                }
            }
    }
    dz_flag_range(p.oflag, amax);
    if constexpr (POOL) {
        if (flags & 24) return;
        // ---- weighted statistics pooling of the tile (DzPoolFuse) ---------------------------------------
        // Round 6: the same two passes (sum w x -> mean, then sum w (x - mean)^2: no cancellation whatever the
        // weights select), laid out for throughput.  The form of round 3 — one column and 64 rows per thread, four
        // dependent LDS reads per row, twice — cost as much as the GEMM of the tile: 80 of the layer's 162 us
        // (timing-only builds, DZ_GP_DBG=8 / 16).  Now wave v sweeps rows 32 v .. 32 v + 31, a lane owns two columns
        // (one 8-byte read per row; the row's <= 4 weights are one broadcast 16-byte read, four rows in flight), the
        // four waves' partial sums meet in LDS, every lane derives the means of its two columns from them and sweeps
        // again.  A tile touches at most two chunks ("parts"), handled one after the other.  (A single pass on
        // shifted moments was tried first: 19 us instead of 30, but OverlappedSpeechPenalty weights select frames
        // whose features sit thousands of their own deviations from any cheap shift — 1e-2 in the embeddings.)
        float* wt = reinterpret_cast<float*>(smem + POOL_WT);          // [128 rows][4 speakers]
        float* red = reinterpret_cast<float*>(smem + POOL_RED);        // [4 waves][4 speakers][128]
        float* s0t = reinterpret_cast<float*>(smem + POOL_S0);
        float* s0f = s0t + 8 * 8 * 2;
        const int K = q->K, P = q->P, T = q->T;
        // pooling weights of the tile's rows (the interpolation of stats_pool_reg_kernel, k_pool.hip)
        for (int i = tid; i < 4 * BM; i += 256) {
            const int rl = i >> 2, k = i & 3, r = t0 + rl;
            const int b = r / P, t = r - b * P;
            float wv = 0.f;
            if (k < K && t < T && r < p.Tout) {
                wv = 1.f;
                if (q->w) wv = dz_pool_weight(q->w + (long long)(b * K + k) * (q->Fw < 0 ? -q->Fw : q->Fw), q->Fw, T, t);
            }
            wt[i] = wv;
        }
        __syncthreads();                                   // tile + weights are in LDS
        const int b0 = t0 / P;
        int rb = (b0 + 1) * P - t0;                        // first local row of the next chunk
        if (rb > BM) rb = BM;
        // (sum w, sum w^2) of each (part, speaker): 8 sub-ranges of 16 rows, then summed in fixed order
        if (tid < 64) {
            const int pk = tid >> 3, sub = tid & 7, part = pk >> 2, k = pk & 3;
            float a = 0.f, a2 = 0.f;
            if (k < K) {
                const int lo = max(part ? rb : 0, 16 * sub), hi = min(part ? BM : rb, 16 * sub + 16);
                for (int r = lo; r < hi; ++r) {
                    const float wv = wt[4 * r + k];
                    a += wv;
                    a2 += wv * wv;
                }
            }
            s0t[(pk * 8 + sub) * 2] = a;
            s0t[(pk * 8 + sub) * 2 + 1] = a2;
        }
        __syncthreads();
        if (tid < 8) {
            float a = 0.f, a2 = 0.f;
            for (int sub = 0; sub < 8; ++sub) {
                a += s0t[(tid * 8 + sub) * 2];
                a2 += s0t[(tid * 8 + sub) * 2 + 1];
            }
            s0f[tid * 2] = a;
            s0f[tid * 2 + 1] = a2;
        }
        const int wv_ = tid >> 6, c2 = 2 * (tid & 63);
        const int nx = p.Tout / P;                         // chunks in this launch
        auto pool_part = [&](auto kk_tag, const int part) {
            constexpr int KK = decltype(kk_tag)::value;
            const int lo = max(part ? rb : 0, 32 * wv_), hi = min(part ? BM : rb, 32 * wv_ + 32);   // wave-uniform
            float S[KK][2];
#pragma unroll
            for (int k = 0; k < KK; ++k) S[k][0] = S[k][1] = 0.f;
#pragma unroll 4
            for (int r = lo; r < hi; ++r) {
                const f32x2 x = *reinterpret_cast<const f32x2*>(yt + r * YT_PITCH + c2);
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(wt + 4 * r);
#pragma unroll
                for (int k = 0; k < KK; ++k) {
                    S[k][0] = __builtin_fmaf(w4[k], x[0], S[k][0]);
                    S[k][1] = __builtin_fmaf(w4[k], x[1], S[k][1]);
                }
            }
#pragma unroll
            for (int k = 0; k < KK; ++k)
                *reinterpret_cast<f32x2*>(red + (wv_ * 4 + k) * BN + c2) = (f32x2){S[k][0], S[k][1]};
            __syncthreads();                               // pass-1 sums of the four waves (and s0f) are in LDS
            float mean[KK][2];
#pragma unroll
            for (int k = 0; k < KK; ++k) {
                f32x2 a = {0.f, 0.f};
#pragma unroll
                for (int v = 0; v < 4; ++v) a += *reinterpret_cast<const f32x2*>(red + (v * 4 + k) * BN + c2);   // fixed order
                const float v1 = s0f[(part * 4 + k) * 2];
                mean[k][0] = v1 > 0.f ? a[0] / v1 : 0.f;
                mean[k][1] = v1 > 0.f ? a[1] / v1 : 0.f;
            }
            __syncthreads();                               // every lane has read the sums: the scratch takes pass 2's
#pragma unroll
            for (int k = 0; k < KK; ++k) S[k][0] = S[k][1] = 0.f;
#pragma unroll 4
            for (int r = lo; r < hi; ++r) {
                const f32x2 x = *reinterpret_cast<const f32x2*>(yt + r * YT_PITCH + c2);
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(wt + 4 * r);
#pragma unroll
                for (int k = 0; k < KK; ++k) {
                    const float d0 = x[0] - mean[k][0], d1 = x[1] - mean[k][1];
                    S[k][0] = __builtin_fmaf(w4[k] * d0, d0, S[k][0]);
                    S[k][1] = __builtin_fmaf(w4[k] * d1, d1, S[k][1]);
                }
            }
#pragma unroll
            for (int k = 0; k < KK; ++k)
                *reinterpret_cast<f32x2*>(red + (wv_ * 4 + k) * BN + c2) = (f32x2){S[k][0], S[k][1]};
            __syncthreads();
            const int b = b0 + part;
            if (wv_ == 0 && b < nx) {
                const int piece = t0 / BM - (b * P) / BM;    // 0 .. np - 1
#pragma unroll
                for (int k = 0; k < KK; ++k) {
                    f32x2 m2 = {0.f, 0.f};
#pragma unroll
                    for (int v = 0; v < 4; ++v) m2 += *reinterpret_cast<const f32x2*>(red + (v * 4 + k) * BN + c2);
                    float* o = q->part + ((((long long)b * q->np + piece) * K + k) * p.Npad + n0 + c2) * 2;
                    *reinterpret_cast<f32x4*>(o) = (f32x4){mean[k][0], m2[0], mean[k][1], m2[1]};
                    if (n0 == 0 && c2 == 0) {
                        float* so = q->s0 + (((long long)b * q->np + piece) * K + k) * 2;
                        so[0] = s0f[(part * 4 + k) * 2];
                        so[1] = s0f[(part * 4 + k) * 2 + 1];
                    }
                }
            }
        };
        for (int part = 0; part < (rb < BM ? 2 : 1); ++part) {
            if (part) __syncthreads();                     // the output lanes of part 0 are done with the scratch
            switch (K) {
                case 1: pool_part(std::integral_constant<int, 1>{}, part); break;
                case 2: pool_part(std::integral_constant<int, 2>{}, part); break;
                case 3: pool_part(std::integral_constant<int, 3>{}, part); break;
                default: pool_part(std::integral_constant<int, 4>{}, part); break;
            }
        }
    }
}

// Workgroups [0, mbig * gy): 128 x 128 tiles over the first mbig * 128 rows (XCD-aware order of
// dz_tile_map); the rest: 64 x 64 tiles over the remaining rows — XCD r owns row tiles r, r+8, ...
// and sweeps the N tiles of one row tile back to back (activation rows stay in that XCD's L2).
template <int EPI, bool ILV>
__global__ __launch_bounds__(256, 2) void gemm_pre_kernel(DzConvGemm p, int mbig, int msmall, int flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifndef DZ_EXPERIMENTS
    flags = 0;                   // the timing-only variants (wrong results) exist in the experiments build only
#endif
    const int gy = p.Npad / BN, L = blockIdx.x;
    if (L < mbig * gy) {
        int bx, by, bz;
        dz_tile_map_lin(L, mbig, gy, 1, p.agroup, bx, by, bz);
        gemm_pre_tile<EPI, 2, 2, 2, ILV>(p, bx * BM, by * BN, smem, flags);
    } else {
        const int Ls = L - mbig * gy, gys = 2 * gy;
        const int xcd = Ls & 7, j = Ls >> 3;
        const int ms = (j / gys) * 8 + xcd, by = j % gys;
        if (ms >= msmall) return;
        gemm_pre_tile<EPI, 2, 1, 1>(p, mbig * BM + ms * 64, by * 64, smem);
    }
}

// tdnn5 with the statistics pooling in its epilogue: 128 x 128 tiles only (the launcher falls back to the
// unfused path when a launch would use small tiles)
__global__ __launch_bounds__(256, 2) void gemm_pre_pool_kernel(DzConvGemm p, DzPoolFuse q, int gx, int flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int gy = p.Npad / BN;
    int bx, by, bz;
    dz_tile_map_lin(blockIdx.x, gx, gy, 1, p.agroup, bx, by, bz);
#ifndef DZ_EXPERIMENTS
    flags = 0;                   // (timing-only variants: experiments build, DZ_GP_DBG 8 = no pooling of the tile, 16 = no tile either)
#endif
    gemm_pre_tile<DZ_EPI_TDNN, 2, 2, 2, true, true>(p, bx * BM, by * BN, smem, flags, &q);
}

// 384 x 128 tiles, 12 waves (3 per SIMD), one workgroup per CU: a third fewer operand bytes per MFMA
// than two 128 x 128 workgroups per CU, and the config-2 grids fit ONE round of the 256 CUs (tdnn2:
// 49 x 4 = 196 tiles) instead of 1.13 rounds.  EXPERIMENT (DZ_GEMM_BIG=1), not the default: measured
// alone tdnn2 122 -> 107 us but tdnn5 112 -> 152 us, projection unchanged, the 64-stream pipeline 3 %
// slower.  A k-step of the single resident workgroup takes 2.2 us for 1.1 us of matrix work: the
// 64 KB it has in flight per CU move at ~30 GB/s per CU (two 128 x 128 workgroups: 47 GB/s), i.e. the
// LDS-DMA stream is latency-, not byte-bound, and neither this nor the deeper pipeline of
// tools/experiments/k_gemm_pre2.hip buys what the byte count promised.
template <int EPI>
__global__ __launch_bounds__(64 * 2 * MW_BIG) void gemm_pre_big_kernel(DzConvGemm p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int gy = p.Npad / BN, gx = (p.Tout + 64 * MW_BIG - 1) / (64 * MW_BIG);
    int bx, by, bz;
    dz_tile_map_lin(blockIdx.x, gx, gy, 1, p.agroup, bx, by, bz);
    gemm_pre_tile<EPI, MW_BIG, 2, 2>(p, bx * 64 * MW_BIG, by * BN, smem);
}
// DZ_GEMM_BIG=1: 384 x 128 tiles for every launch with at least 8 such row tiles
int big_tiles_mode() {
    static const int mode = [] {
        const char* e = dz_exp_env("DZ_GEMM_BIG");
        return e ? atoi(e) : 0;
    }();
    return mode;
}

// DZ_GP_LOOP=0: the plain k-loop (all LDS-DMA pieces of the next tile issued in one block) instead of the
// interleaved one
bool interleaved_loop() {
    static const bool on = [] {
        const char* e = dz_exp_env("DZ_GP_LOOP");
        return !(e && e[0] == '0');
    }();
    return on;
}

int dbg_flags() {      // DZ_GP_DBG: timing experiments of the interleaved loop (results are wrong)
    static const int f = [] {
        const char* e = dz_exp_env("DZ_GP_DBG");
        return e ? atoi(e) : 0;
    }();
    return f;
}

// DZ_GEMM_TAIL=1: 64 x 64 tiles for the rows of a mostly empty last round as well (see the header)
bool tail_tiles_enabled() {
    static const bool on = [] {
        const char* e = dz_exp_env("DZ_GEMM_TAIL");
        return e && e[0] == '1';
    }();
    return on;
}
int wg_slots() {   // resident workgroups of this kernel on the chip: 2 per CU (64 KiB LDS, <= 256 VGPRs)
    static const int slots = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return 2 * (cus > 0 ? cus : 256);
    }();
    return slots;
}

template <int EPI>
int launch(const DzConvGemm& p, hipStream_t st) {
    static DzAttrOnce attr_ilv;
    DZ_HIP(attr_ilv.raise((const void*)gemm_pre_kernel<EPI, true>, (int)lds_bytes(2)));
    const int gx = (p.Tout + BM - 1) / BM, gy = p.Npad / BN, slots = wg_slots();
#ifdef DZ_EXPERIMENTS
    static DzAttrOnce attr_once, attr_big;
    DZ_HIP(attr_once.raise((const void*)gemm_pre_kernel<EPI, false>, (int)lds_bytes(2)));
    {
        const int gxb = (p.Tout + 64 * MW_BIG - 1) / (64 * MW_BIG);
        const int mode = big_tiles_mode();
        if (mode == 1 && gxb >= 8) {
            DZ_HIP(attr_big.raise((const void*)gemm_pre_big_kernel<EPI>, (int)lds_bytes(MW_BIG)));
            DZ_LAUNCH((gemm_pre_big_kernel<EPI>), dim3(gxb * gy), dim3(64 * 2 * MW_BIG), lds_bytes(MW_BIG), st, p);
            DZ_HIP(hipGetLastError());
            return 0;
        }
    }
#endif
    int mbig = gx;
    const int tiles = gx * gy, rem = tiles % slots;
    if (4 * tiles < slots)
        mbig = 0;                                   // latency regime: small tiles throughout
    else if (tail_tiles_enabled() && rem != 0 && 4 * rem < 3 * slots)
        mbig = (tiles / slots) * slots / gy;        // a last round that is at least 3/4 full is left alone
    const int rows_left = p.Tout - mbig * BM;
    const int msmall = rows_left > 0 ? (rows_left + 63) / 64 : 0;
    const int nwg = mbig * gy + ((msmall + 7) / 8) * 8 * (2 * gy);
#ifdef DZ_EXPERIMENTS
    if (!interleaved_loop())
        DZ_LAUNCH((gemm_pre_kernel<EPI, false>), dim3(nwg), dim3(256), lds_bytes(2), st, p, mbig, msmall, 0);
    else
#endif
        DZ_LAUNCH((gemm_pre_kernel<EPI, true>), dim3(nwg), dim3(256), lds_bytes(2), st, p, mbig, msmall, dbg_flags());
    DZ_HIP(hipGetLastError());
    return 0;
}

}  // namespace

// DZ_GEMM_GEN: 1 = k_gemm_pre.hip's loop, 2 = k_gemm_g2.hip, 3 = k_gemm_g3.hip (persistent, Stream-K)
int dz_gemm_gen() {
    static const int gen = [] {
        const char* e = dz_exp_env("DZ_GEMM_GEN");
        const int v = e ? atoi(e) : 1;
        return v == 2 || v == 3 ? v : 1;
    }();
    return gen;
}

bool dz_gemm_pre_pool_ok(const DzConvGemm& p) {
    const int gx = (p.Tout + BM - 1) / BM, gy = p.Npad / BN;
    return 4 * gx * gy >= wg_slots();                  // the latency regime uses 64 x 64 tiles: not built pooled
}

int dz_launch_gemm_pre_pool(const DzConvGemm& p_in, const DzPoolFuse& q, hipStream_t st) {
    DzConvGemm p = p_in;
    if (!p.oflag) p.oflag = dz_cur_oflag;
    DZ_REQUIRE(p.Wsplit && p.Xsplit && q.part && q.s0, "gemm_pre_pool: NULL operand");
    DZ_REQUIRE(p.epi == DZ_EPI_TDNN && p.B == 1 && p.taps == 1 && p.K == p.Kpad && p.Cin % KT == 0 &&
                   p.Npad % BN == 0 && p.Tout == p.Tin,
               "gemm_pre_pool: built for the flattened 1 x 1 TDNN layer");
    DZ_REQUIRE(q.np == dz_pool_pieces(q.P), "gemm_pre_pool: np must be dz_pool_pieces(P)");
    DZ_REQUIRE(q.K >= 1 && q.K <= 4 && q.P >= BM && q.T >= 2 && q.T <= q.P && p.Tout % q.P == 0 && (q.Fw >= 2 || q.Fw <= -2),
               "gemm_pre_pool: 1..4 speakers, chunk pitch >= 128 rows (got K %d, P %d, T %d)", q.K, q.P, q.T);
    DZ_REQUIRE(p.ldx > 0 && p.ldx % KT == 0 && p.xplane % p.ldx == 0 && p.xplane / p.ldx >= p.Tin,
               "gemm_pre_pool: kb-major input planes need ldx %% 32 == 0 and xplane = rows * ldx with rows >= Tin");
    DZ_REQUIRE(p.xplane * 2 < (1ll << 31) && (long long)p.Npad * p.Kpad * 2 < (1ll << 31),
               "gemm_pre_pool: operand plane exceeds the 2 GiB buffer-offset range");
    // Row tiles swept together per XCD (dz_tile_map_lin).  ONE: the 12 column tiles of a row tile are then
    // consecutive workgroups of their XCD and share the activation tile while it is in the L2 (live PMC,
    // per launch: 157 MB with groups of 4, 145 with 2, 135 with 1; the weights — 3 MB — stay resident
    // either way).  DZ_POOL_AG overrides.
    static const int pool_ag = [] {
        const char* e = dz_exp_env("DZ_POOL_AG");
        return e && atoi(e) > 0 ? atoi(e) : 1;
    }();
    if (!p.agroup) p.agroup = pool_ag;
    static DzAttrOnce attr_once;
    DZ_HIP(attr_once.raise((const void*)gemm_pre_pool_kernel, (int)POOL_LDS));
    const int gx = (p.Tout + BM - 1) / BM, gy = p.Npad / BN;
    DZ_LAUNCH(gemm_pre_pool_kernel, dim3(gx * gy), dim3(256), POOL_LDS, st, p, q, gx, dbg_flags());
    DZ_HIP(hipGetLastError());
    return 0;
}

int dz_launch_gemm_pre(const DzConvGemm& p_in, hipStream_t st) {
    DzConvGemm p = p_in;
    if (!p.oflag) p.oflag = dz_cur_oflag;
    DZ_REQUIRE(p.Wsplit != nullptr && p.Xsplit != nullptr,
               "gemm_pre: Wsplit / Xsplit (f16 hi/lo planes of W and of the input) are NULL");
    DZ_REQUIRE(p.Y != nullptr || p.Ysplit != nullptr, "gemm_pre: no output");
    DZ_REQUIRE(p.B == 1, "gemm_pre: flattened layers only (B = 1)");
    DZ_REQUIRE(p.K == p.Kpad && p.K == p.taps * p.Cin && p.Cin % KT == 0,
               "gemm_pre: K = taps * Cin without padding, Cin a multiple of 32");
    DZ_REQUIRE(p.Npad % BN == 0, "gemm_pre: Npad must be a multiple of 128");
    DZ_REQUIRE(p.Tout > 0 && p.Tout == p.Tin - (p.taps - 1) * p.dil, "gemm_pre: Tout mismatch");
    DZ_REQUIRE(p.pad == 0 && p.X2 == nullptr && p.rowbias == nullptr && p.ksplit <= 1 && !p.norm_on_load,
               "gemm_pre: padding / second input / row bias / split-K / norm-on-load are not built here");
    DZ_REQUIRE(p.ldx > 0 && p.ldx % KT == 0 && p.xplane % p.ldx == 0 && p.xplane / p.ldx >= p.Tin,
               "gemm_pre: kb-major input planes need ldx %% 32 == 0 and xplane = rows * ldx with rows >= Tin");
    DZ_REQUIRE(p.xplane * 2 < (1ll << 31) && (long long)p.Npad * p.Kpad * 2 < (1ll << 31),
               "gemm_pre: operand plane exceeds the 2 GiB buffer-offset range");
    DZ_REQUIRE(p.Ysplit == nullptr || (p.ldy > 0 && p.ldy % KT == 0 && p.yplane % p.ldy == 0 && p.yplane / p.ldy >= p.Tout &&
                                       p.Npad <= p.ldy && p.yplane * 2 < (1ll << 31)),
               "gemm_pre: kb-major output planes need ldy %% 32 == 0, yplane = rows * ldy with rows >= Tout, Npad <= ldy");
    DZ_REQUIRE(p.Y == nullptr || (p.ldy % 4 == 0 && ((uintptr_t)p.Y & 15) == 0),
               "gemm_pre: f32 output needs ldy a multiple of 4 and a 16-byte aligned base");
#ifdef DZ_EXPERIMENTS
    // generation 2 (k_gemm_g2.hip) for every launch outside the latency regime (there: 64 x 64 tiles below)
    if (dz_gemm_gen() >= 2 && 4 * ((p.Tout + BM - 1) / BM) * (p.Npad / BN) >= wg_slots()) {
        // generation 3 pays a hand-over of the accumulators per workgroup: only worth it for long k-loops
        // (DZ_G3_MINK, default 1024: tdnn2 / tdnn3 of the x-vector network; the others stay on generation 1)
        static const int g3_min_k = [] {
            const char* e = dz_exp_env("DZ_G3_MINK");
            return e ? atoi(e) : 1024;
        }();
        if (dz_gemm_gen() == 3) {
            if (p.K >= g3_min_k) return dz_launch_gemm_g3(p, 0, st);
        } else {
            return dz_launch_gemm_g2(p, 0, st);
        }
    }
#endif
    switch (p.epi) {
        case DZ_EPI_BIAS: return launch<DZ_EPI_BIAS>(p, st);
        case DZ_EPI_BIAS_LEAKY: return launch<DZ_EPI_BIAS_LEAKY>(p, st);
        case DZ_EPI_TDNN: return launch<DZ_EPI_TDNN>(p, st);
        case DZ_EPI_RELU_BN: return launch<DZ_EPI_RELU_BN>(p, st);
    }
    dz_set_error("gemm_pre: epilogue %d is not built on the pre-split path", p.epi);
    return 2;
}
