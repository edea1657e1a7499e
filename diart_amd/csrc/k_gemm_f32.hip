// Implicit-GEMM 1-D convolution / linear layer on the EXACT-f32 matrix cores, generation 2 (round 4).
//
//   Y[b][t][n] = epi( sum_{tap,c} X[b][t + tap*dil][c] * W[n][tap*Cin + c] + bias[n] )
//
// The wide layers of the precision="f32" path — x-vector tdnn2..5, the LSTM input projections of layers
// 1..3, the segmentation MLP, ECAPA's 1 x 1 layers (third-party graphs called from
// /root/reference/src/diart/models.py:133, :262, :59; SURVEY.md Appendix A, kernels K5 / K6 / K8) — i.e. the
// number bench.py reports as `exact_f32`: the reference's own arithmetic.  k_convgemm.hip (round 1) serves
// them at 53 - 57 % of the 157 TFLOP/s f32 matrix peak: its operands go global -> registers -> LDS (7 loads, 7
// ds_write_b128 and the index arithmetic of an implicit GEMM per thread per k-tile, on the same issue ports
// as the MFMAs), its 96-row tile leaves 1.5 rounds of 772 tiles on 512 slots, and the fragment loop has no
// software pipeline.  This kernel is k_gemm_pre.hip's loop with f32 data:
//
//   * tile 128 x 128 x 16, 4 waves (2 x 2), wave tile 64 x 64 = 2 x 2 fragments of v_mfma_f32_32x32x2_f32
//     (64 cycles each; 32 per wave per k-tile = 2048 cycles between two barriers);
//   * both operands are plain row-major f32 ([rows][Cin] activations, [Npad][Kpad] weights): a 16-wide k-tile
//     of 16 rows is 16 x 64 contiguous bytes and goes global -> LDS by ONE LDS-DMA instruction per wave
//     (buffer_load_dwordx4 ... lds, 16 B per lane, no VGPR staging, no ds_write, no VALU); rows beyond the
//     operand read as zeros through the buffer bounds check (the row — tap shift included — is in the
//     per-lane offset, which the check always covers; the scalar offset only moves inside a row);
//   * LDS stage = A [128][64 B] | B [128][64 B]; the 16-byte chunk c of row r sits in slot c ^ ((r >> 2) & 3),
//     so the 16 lanes of a ds_read_b128 group (16 consecutive rows, one chunk) cover all 64 banks once; the
//     LDS-DMA destination is lane-linear, so the swizzle is applied to the per-lane SOURCE chunk;
//   * a lane's ds_read_b128 = 4 consecutive k of its row = the operand of 4 MFMAs: lane (row, h) of MFMA u of
//     sub-step s carries k = 8 s + 4 h + u for BOTH operands (the k order inside a k-tile is permuted
//     consistently; each product is still one exact f32 FMA into the f32 accumulator);
//   * TRANSPOSED product (weights = the MFMA's row operand): a lane ends with one output row and, per
//     register group, four consecutive columns: 16-byte stores;
//   * FOUR stages (64 KiB, two workgroups per CU), counted vmcnt: the 4 pieces a wave fetches of tile kt + 3 are
//     issued between the MFMAs of tile kt (one every 8 MFMAs), so a piece has three k-tiles (~6 000 cycles) to land
//     before the top-of-loop wait needs it.  (The first version had two 32-wide stages and fetched tile kt + 1
//     during tile kt: its last pieces were ~500 cycles old at the wait — 96 - 103 TFLOP/s on the TDNN layers.)
#include "dz_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, KT = 16;
constexpr int NST = 4;                      // LDS stages: the pieces of tile kt + 3 are issued while tile kt is computed
constexpr int OPER = 128 * KT * 4;          // bytes of one operand tile of a stage (8 KiB)
constexpr int STAGE = 2 * OPER;             // A | B
constexpr size_t LDS_BYTES = NST * (size_t)STAGE + 3 * BN * sizeof(float);

__device__ __forceinline__ float leaky(float v) { return v > 0.f ? v : v * DZ_LEAKY_SLOPE; }

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(DzConvGemm p) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    int bx, by, bz;
    dz_tile_map(p.agroup, bx, by, bz);
    const int t0 = bx * BM, n0 = by * BN, b = bz;

    // ---- staging: waves 0, 1 fetch the activation tile, waves 2, 3 the weight tile; wave of rank r takes the
    // 16-row pieces r, r + 2, ... (4 per k-tile).  Lane -> (row l >> 2 of the piece, LDS slot l & 3); the slot
    // holds source chunk slot ^ ((row >> 2) & 3) = slot ^ ((l >> 4) & 3) for every piece.
    const bool isB = w >= 2;
    const int rank = w & 1;
    const float* base = isB ? p.W : p.X + (long long)b * p.xbs;
    const int ld = isB ? p.Kpad : p.ldx;                         // floats per operand row
    const int nrows = isB ? p.Npad : p.Tin;
    const unsigned nbytes = (unsigned)((long long)nrows * ld * 4);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, nbytes, 0x00020000);
    const int voff0 = (((isB ? n0 : t0) + 16 * rank + (l >> 2)) * ld + (((l & 3) ^ ((l >> 4) & 3)) << 2)) * 4;   // bytes
    const int vstep = 32 * ld * 4;                               // pieces i and i + 2 are 32 rows apart
    const int tap_v = isB ? 0 : p.dil * ld * 4;                  // per-lane byte offset of one tap (activations)
    const int tap_s = isB ? p.Cin * 4 : 0;                       // scalar byte offset of one tap (weights' K axis)
    char* const dbase = smem + (isB ? OPER : 0) + rank * 1024;

    // ---- epilogue parameters -> LDS (visible behind the first barrier of the loop) ----------------------
    float* par = reinterpret_cast<float*>(smem + NST * STAGE);
    {
        constexpr bool AFF = EPI == DZ_EPI_TDNN || EPI == DZ_EPI_RELU_BN || EPI == DZ_EPI_RELU_BN_TANH;
        const int which = tid >> 5, c4 = (tid & 31) * 4;
        if (which < (AFF ? 3 : 1)) {
            const float* src = which == 0 ? p.bias : which == 1 ? p.e0 : p.e1;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (n0 + c4 + 3 < p.Npad) v = *reinterpret_cast<const f32x4*>(src + n0 + c4);
            *reinterpret_cast<f32x4*>(par + which * BN + c4) = v;
        }
    }

    // ---- MFMA coordinates ---------------------------------------------------------------------------------
    const int li = l & 31, g = l >> 5;
    const int wm = w >> 1, wn = w & 1;
    int foff[2];                                  // this lane's chunk of sub-step s inside a 32-row block
#pragma unroll
    for (int s = 0; s < 2; ++s) foff[s] = li * 64 + (((2 * s + g) ^ ((li >> 2) & 3)) << 4);
    f32x16 acc[2][2];
#pragma unroll
    for (int xb = 0; xb < 2; ++xb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[xb][nb][r] = 0.f;

    // k-tile kt of the loop = (channel half-block kt / taps, tap kt % taps): the taps of one 16-channel block follow
    // each other (the reads of almost the same activation lines stay in the L2, see k_gemm_pre.hip)
    const int nk = p.Kpad / KT;
    int cblk = 0, tap = 0, fetched = 0;                   // (block, tap) and index of the NEXT tile to fetch
    int stage_f = 0;                                      // its stage
    // Tiles beyond the last one are "fetched" with an out-of-range offset (zeros into a stage nobody reads): the
    // loop body has no branch around its LDS-DMA instructions and its vmcnt is the same in every iteration
    auto fetch_piece = [&](int j) {
        const int in = fetched < nk;
        const int voff = in ? voff0 + tap * tap_v + j * vstep : 0x7f000000;
        const int soff = __builtin_amdgcn_readfirstlane(cblk * (KT * 4) + tap * tap_s);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rsrc, (__attribute__((address_space(3))) void*)(dbase + stage_f * STAGE + j * 2048), 16, voff, soff, 0, 0);
    };
    auto fetched_one = [&]() {
        ++fetched;
        stage_f = (stage_f + 1) & (NST - 1);
        if (++tap == p.taps) { tap = 0; ++cblk; }
    };
#pragma unroll
    for (int t = 0; t < NST - 1; ++t) {                   // tiles 0 .. 2 -> stages 0 .. 2
#pragma unroll
        for (int j = 0; j < 4; ++j) fetch_piece(j);
        fetched_one();
    }

    const char* const sa0 = smem + (wm * 64) * 64;        // activation rows of this wave
    const char* const sb0 = smem + OPER + (wn * 64) * 64;
#define DZ_RD(base, blk, s) (*reinterpret_cast<const f32x4*>((base) + (blk) * 2048 + foff[s]))
#define DZ_PIN() __builtin_amdgcn_sched_barrier(0)
#define DZ_MM4(buf, u)                                                                                          \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[buf][0][u], xf[buf][0][u], acc[0][0], 0, 0, 0);         \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[buf][1][u], xf[buf][0][u], acc[0][1], 0, 0, 0);         \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[buf][1][u], xf[buf][1][u], acc[1][1], 0, 0, 0);         \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[buf][0][u], xf[buf][1][u], acc[1][0], 0, 0, 0)
    // The loop is software-pipelined ACROSS the k-tile boundary: the barrier of tile kt + 1 and the first fragment
    // reads behind it sit in front of the LAST EIGHT MFMAs of tile kt (whose operands are in registers), so the
    // matrix pipe has 512 cycles of work while the barrier settles and the LDS answers.  With the barrier at the top
    // of a tile the pipe was 74 % busy (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE, tools/visits/gpu_r4w.sh): two
    // workgroups per CU that started together stall together.
    f32x4 wf[2][2], xf[2][2];                              // [buffer][32-row block]; buffer 0 = sub-step 0
    // tile 0 has landed (8 younger pieces may be in flight)
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    wf[0][0] = DZ_RD(sb0, 0, 0); xf[0][0] = DZ_RD(sa0, 0, 0);
    wf[0][1] = DZ_RD(sb0, 1, 0); xf[0][1] = DZ_RD(sa0, 1, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const char* sa = sa0 + (kt & (NST - 1)) * STAGE;
        const char* sb = sb0 + (kt & (NST - 1)) * STAGE;
        const char* san = sa0 + ((kt + 1) & (NST - 1)) * STAGE;
        const char* sbn = sb0 + ((kt + 1) & (NST - 1)) * STAGE;
        DZ_PIN();
        DZ_MM4(0, 0);
        DZ_PIN();
        wf[1][0] = DZ_RD(sb, 0, 1); xf[1][0] = DZ_RD(sa, 0, 1);     // sub-step 1 of this tile
        wf[1][1] = DZ_RD(sb, 1, 1); xf[1][1] = DZ_RD(sa, 1, 1);
        DZ_PIN();
        DZ_MM4(0, 1);
        DZ_PIN();
        fetch_piece(0);                                    // tile kt + 3 -> the stage tile kt - 1 was read from
        DZ_PIN();
        DZ_MM4(0, 2);
        DZ_MM4(0, 3);
        DZ_PIN();
        fetch_piece(1);
        DZ_PIN();
        DZ_MM4(1, 0);
        DZ_MM4(1, 1);
        DZ_PIN();
        fetch_piece(2);
        DZ_PIN();
        // every fragment read of tile kt is complete (lgkmcnt) in every wave (barrier), and this wave's pieces of tile
        // kt + 1 have landed: younger than those are the 4 of tile kt + 2 and the 3 of tile kt + 3 issued above
        asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        wf[0][0] = DZ_RD(sbn, 0, 0); xf[0][0] = DZ_RD(san, 0, 0);   // sub-step 0 of tile kt + 1 (after the last tile: unused)
        wf[0][1] = DZ_RD(sbn, 1, 0); xf[0][1] = DZ_RD(san, 1, 0);
        DZ_PIN();
        DZ_MM4(1, 2);
        DZ_MM4(1, 3);
        DZ_PIN();
        fetch_piece(3);
        fetched_one();
    }
#undef DZ_MM4
#undef DZ_RD
#undef DZ_PIN

    // ---- epilogue.  C/D map of the transposed product: column = lane & 31 = output ROW t, register r = output
    // column (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the 32-column block: registers 4 k .. 4 k + 3 are four
    // consecutive columns -> one 16-byte store
    float* Yb = p.Y + (long long)b * p.ybs;
#pragma unroll
    for (int xb = 0; xb < 2; ++xb) {
        const int t = t0 + wm * 64 + xb * 32 + li;
        if (t >= p.Tout) continue;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int nl = wn * 64 + nb * 32 + 8 * k + 4 * g, n = n0 + nl;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(par + nl);
                f32x4 e0 = {1.f, 1.f, 1.f, 1.f}, e1 = {0.f, 0.f, 0.f, 0.f};
                if (EPI == DZ_EPI_TDNN || EPI == DZ_EPI_RELU_BN || EPI == DZ_EPI_RELU_BN_TANH) {
                    e0 = *reinterpret_cast<const f32x4*>(par + BN + nl);
                    e1 = *reinterpret_cast<const f32x4*>(par + 2 * BN + nl);
                }
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = acc[xb][nb][4 * k + e] + bv[e];
                    if (EPI == DZ_EPI_BIAS_LEAKY) x = leaky(x);
                    if (EPI == DZ_EPI_BIAS_SIGMOID) x = 1.f / (1.f + expf(-x));
                    if (EPI == DZ_EPI_TDNN) x = leaky(x) * e0[e] + e1[e];
                    if (EPI == DZ_EPI_BIAS_RELU) x = fmaxf(x, 0.f);
                    if (EPI == DZ_EPI_RELU_BN) x = fmaxf(x, 0.f) * e0[e] + e1[e];
                    if (EPI == DZ_EPI_RELU_BN_TANH) x = tanhf(fmaxf(x, 0.f) * e0[e] + e1[e]);
                    v[e] = x;
                }
                float* dst = Yb + (long long)t * p.ldy + n;
                if (n + 3 < p.Nstore) {
                    *reinterpret_cast<f32x4*>(dst) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.Nstore) dst[e] = v[e];
                }
            }
        }
    }
}

template <int EPI>
int launch(const DzConvGemm& p, hipStream_t st) {
    static DzAttrOnce attr_once;
    DZ_HIP(attr_once.raise((const void*)gemm_f32_kernel<EPI>, (int)LDS_BYTES));
    dim3 grid((p.Tout + BM - 1) / BM, p.Npad / BN, p.B);
    DZ_LAUNCH((gemm_f32_kernel<EPI>), grid, dim3(256), LDS_BYTES, st, p);
    DZ_HIP(hipGetLastError());
    return 0;
}

}  // namespace

// Layers this kernel serves: f32 operands without a prologue, K = taps * Cin with Cin a multiple of 32 (no K
// padding), full 128-column tiles, f32 output.  DZ_F32_GEMM=0 keeps every layer on k_convgemm.hip.
bool dz_gemm_f32_ok(const DzConvGemm& p) {
    const bool on = dz_option(DZ_OPT_F32_GEMM) != 0;      // (read per launch: the tests switch it in-process)
    // 16-byte vector loads / stores: every base the kernel touches, and the per-batch strides, must keep that alignment
    const auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    const bool aligned = al16(p.X) && al16(p.W) && al16(p.Y) && al16(p.bias) && al16(p.e0) && al16(p.e1) &&
                         p.xbs % 4 == 0 && p.ybs % 4 == 0;
    // the buffer descriptor of a batch spans Tin * ldx floats (rows may not overlap: ldx >= Cin), and the
    // out-of-range sentinel offset 0x7f000000 must lie beyond both operands
    const long long xbytes = (long long)p.Tin * p.ldx * 4, wbytes = (long long)p.Npad * p.Kpad * 4;
    return on && aligned && p.X && p.W && p.Y && !p.Ysplit && !p.norm_on_load && p.pad == 0 && !p.X2 && !p.rowbias &&
           p.ksplit <= 1 && !p.partials && p.Npad % BN == 0 && p.K == p.Kpad && p.Cin % KT == 0 && p.Kpad >= KT && p.K == p.taps * p.Cin &&
           p.ldx >= p.Cin && p.ldx % 4 == 0 && p.ldy % 4 == 0 && p.Tout > 0 && p.Tout == p.Tin - (p.taps - 1) * p.dil &&
           p.epi != DZ_EPI_POOL3 && xbytes < 0x7f000000ll && wbytes < 0x7f000000ll;
}

int dz_launch_gemm_f32(const DzConvGemm& p, hipStream_t st) {
    DZ_REQUIRE(dz_gemm_f32_ok(p), "gemm_f32: layer outside the kernel's domain (see dz_gemm_f32_ok)");
    DZ_REQUIRE(p.bias != nullptr, "gemm_f32: bias is NULL");
    switch (p.epi) {
        case DZ_EPI_BIAS: return launch<DZ_EPI_BIAS>(p, st);
        case DZ_EPI_BIAS_LEAKY: return launch<DZ_EPI_BIAS_LEAKY>(p, st);
        case DZ_EPI_BIAS_SIGMOID: return launch<DZ_EPI_BIAS_SIGMOID>(p, st);
        case DZ_EPI_TDNN: return launch<DZ_EPI_TDNN>(p, st);
        case DZ_EPI_BIAS_RELU: return launch<DZ_EPI_BIAS_RELU>(p, st);
        case DZ_EPI_RELU_BN: return launch<DZ_EPI_RELU_BN>(p, st);
        case DZ_EPI_RELU_BN_TANH: return launch<DZ_EPI_RELU_BN_TANH>(p, st);
    }
    dz_set_error("gemm_f32: unknown epilogue %d", p.epi);
    return 2;
}
