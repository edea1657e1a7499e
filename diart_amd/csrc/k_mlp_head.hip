// The tail of the segmentation network in ONE launch (PyanNet: linear[0], linear[1], classifier,
// activation; third-party graph called from /root/reference/src/diart/models.py:133, powerset
// adapter models.py:29-39) followed by the OverlappedSpeechPenalty
// weights of the result (functional.py:6-13):
//
//   h (rows x 256, f16 hi/lo planes written by the last recurrence)
//     -> lin0: 256 -> 128, LeakyReLU      f16x3 MFMA, operands by LDS-DMA (as k_gemm_pre.hip)
//     -> lin1: 128 -> 128, LeakyReLU      f16x3 MFMA, A = lin0's tile re-split INTO LDS, B by LDS-DMA
//     -> classifier 128 -> classes        f32 FMA chain per frame (the order of seg_head_kernel)
//     -> sigmoid | powerset decision -> seg; OSP weights (not normalised) -> wout
//
// It replaces three launches (two k_gemm_pre.hip GEMMs + seg_head_kernel) and the 2 x 9.6 MB round
// trip of the hidden activations; the arithmetic per output is the same as theirs, statement by
// statement (same k order, same split, same FMA chains), so the results are bit-identical
// (tests/test_gpu_kernels.py::test_mlp_head).  One workgroup = 128 consecutive rows (frames of the
// flattened batch), 4 waves (2 x 2, wave tile 64 x 64).  LDS (one object, 137 KB, one workgroup/CU):
//
//   R0 [0, 64K)      lin0: two stages of A hi | A lo | B hi | B lo ([128][64 B] each, 16-byte chunks
//                    XOR-swizzled with (row >> 2) & 3); then lin1's whole B operand: 4 k-tiles x
//                    (B hi | B lo), fetched while lin0's epilogue runs
//   R1 [64K, 128K)   lin1's A operand: 4 k-tiles x (A hi | A lo) written by lin0's epilogue; then the
//                    f32 output tile of lin1, [128][129] (66 KB: runs 2 KB into PAR's slack)
//   PAR              b0 | b1 | classifier rows | classifier bias
#include "dz_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PLANE = 128 * 64;                 // one f16 plane of a 128-row, 32-wide k-tile
constexpr int R0 = 0, R1 = 8 * PLANE;           // 64 KB each
constexpr int M1_PITCH = 129;
constexpr int PAR = R1 + 128 * M1_PITCH * 4 + 64;             // after the f32 tile (66 048 B)
constexpr int PAR_B0 = 0, PAR_B1 = 128, PAR_CW = 256, PAR_CB = 256 + 8 * 128;   // float offsets
constexpr size_t LDS_BYTES = PAR + (PAR_CB + 8) * sizeof(float);
constexpr float LO_UNSCALE = 1.f / 2048.f;

__device__ __forceinline__ float leaky(float v) { return v > 0.f ? v : v * DZ_LEAKY_SLOPE; }

__global__ __launch_bounds__(256) void mlp_head_kernel(DzMlpHead p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int t0 = blockIdx.x * 128;

    // ---- parameters -> LDS (read by the epilogues / the head long after) --------------------------
    float* par = reinterpret_cast<float*>(smem + PAR);
    if (tid < 128) {
        par[PAR_B0 + tid] = p.b0[tid];
        par[PAR_B1 + tid] = p.b1[tid];
    }
    for (int i = tid; i < p.classes * 128; i += 256) par[PAR_CW + i] = p.cw[i];
    if (tid < p.classes) par[PAR_CB + tid] = p.cb[tid];

    // ---- staging role of this wave in lin0: plane w of every stage (A hi, A lo, B hi, B lo) -------
    const bool isB = w >= 2;
    const int lo = w & 1;
    const unsigned short* A = reinterpret_cast<const unsigned short*>(p.Xsplit);
    const unsigned short* W0 = reinterpret_cast<const unsigned short*>(p.W0split);
    const unsigned short* W1 = reinterpret_cast<const unsigned short*>(p.W1split);
    const unsigned short* src = isB ? W0 + (long long)lo * 128 * 256 : A + (long long)lo * p.xplane;
    const unsigned nbytes = (unsigned)((isB ? 128ll : (long long)p.rows) * 256 * 2);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    const int chunk = ((l & 3) ^ ((l >> 4) & 3)) << 4;
    // kb-major planes (dz_common.h dz_kb): [256 / 32][rows][32] — a 16-row piece of a k-tile is 1 KiB contiguous
    const int voff0 = ((isB ? 0 : t0) + (l >> 2)) * 64 + chunk;
    const int kb_bytes = (isB ? 128 : p.rows) * 64;
    auto issue0 = [&](int kt, int stage) {
        char* dst = smem + R0 + stage * 4 * PLANE + w * PLANE;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsrc, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, voff0 + i * 1024, kt * kb_bytes, 0, 0);
    };
    // lin1's B operand: 4 k-tiles x 2 planes x 8 pieces = 64 pieces, 16 per wave (k-tile w)
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)W1, 0, 2u * 128 * 128 * 2, 0x00020000);
    auto issue1 = [&]() {
        const int voff1 = (l >> 2) * 64 + chunk;                            // [128 / 32][128][32] per plane
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    rs1, (__attribute__((address_space(3))) void*)(smem + R0 + w * 2 * PLANE + pl * PLANE + i * 1024), 16,
                    voff1 + i * 1024 + pl * (128 * 128 * 2), w * (128 * 64), 0, 0);
    };

    // ---- MFMA coordinates (k_gemm_pre.hip: transposed product, a lane owns one row) ---------------
    const int li = l & 31, g = l >> 5;
    const int wm = w >> 1, wn = w & 1;
    const int sw = (li >> 2) & 3;
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = li * 64 + (((2 * ks + g) ^ sw) << 4);
    f32x16 accm[2][2], accx[2][2];
    auto zero = [&]() {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) accm[mt][nt][r] = accx[mt][nt][r] = 0.f;
    };
    // one 32-wide k-tile: A planes at a_hi / a_hi + PLANE, B planes at b_hi / b_hi + PLANE
    auto compute = [&](const char* a_hi, const char* b_hi) {
        const char* sa = a_hi + (wm * 64) * 64;
        const char* sb = b_hi + (wn * 64) * 64;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ah[t] = *reinterpret_cast<const f16x8*>(sa + t * 2048 + foff[ks]);
                al[t] = *reinterpret_cast<const f16x8*>(sa + PLANE + t * 2048 + foff[ks]);
                bh[t] = *reinterpret_cast<const f16x8*>(sb + t * 2048 + foff[ks]);
                bl[t] = *reinterpret_cast<const f16x8*>(sb + PLANE + t * 2048 + foff[ks]);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    accx[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[nt], al[mt], accx[mt][nt], 0, 0, 0);
                    accm[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[nt], ah[mt], accm[mt][nt], 0, 0, 0);
                    accx[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[nt], ah[mt], accx[mt][nt], 0, 0, 0);
                }
        }
    };

    // ---- lin0: K = 256 = 8 k-tiles -----------------------------------------------------------------
    zero();
    issue0(0, 0);
    for (int kt = 0; kt < 8; ++kt) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 1 < 8) issue0(kt + 1, (kt + 1) & 1);
        const char* st = smem + R0 + (kt & 1) * 4 * PLANE;
        compute(st, st + 2 * PLANE);
    }
    // every wave is done with R0 -> lin1's weights can land there while the epilogue below runs
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    issue1();

    // ---- epilogue of lin0: bias, LeakyReLU, split -> lin1's A planes in R1 -------------------------
    // lane: row t = 64 wm + 32 mt + li; register group k: columns n = 64 wn + 32 nt + 8 k + 4 g + {0..3}
    // = k-tile n / 32 of lin1, 16-byte chunk (n % 32) / 8 = k, byte 8 g inside it
    float amax = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int row = wm * 64 + mt * 32 + li;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int n = wn * 64 + nt * 32 + 8 * k + 4 * g;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(par + PAR_B0 + n);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = leaky((accm[mt][nt][4 * k + e] + accx[mt][nt][4 * k + e] * LO_UNSCALE) + bv[e]);
                    amax = fmaxf(amax, fabsf(x));
                    v[e] = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
                }
                const f16x4 hi = __builtin_convertvector(v, f16x4);
                const f16x4 lo4 = __builtin_convertvector((v - __builtin_convertvector(hi, f32x4)) * 2048.f, f16x4);
                char* d = smem + R1 + (wn * 2 + nt) * 2 * PLANE + row * 64 + ((k ^ ((row >> 2) & 3)) << 4) + 8 * g;
                *reinterpret_cast<f16x4*>(d) = hi;
                *reinterpret_cast<f16x4*>(d + PLANE) = lo4;
            }
    }
    dz_flag_range(p.oflag, amax);

    // ---- lin1: K = 128 = 4 k-tiles, both operands resident ------------------------------------------
    zero();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) compute(smem + R1 + kt * 2 * PLANE, smem + R0 + kt * 2 * PLANE);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // R1 is free again

    // ---- epilogue of lin1: bias, LeakyReLU -> f32 tile [128][129] in R1 ------------------------------
    float* m1 = reinterpret_cast<float*>(smem + R1);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int row = wm * 64 + mt * 32 + li;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int n = wn * 64 + nt * 32 + 8 * k + 4 * g;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(par + PAR_B1 + n);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    m1[row * M1_PITCH + n + e] =
                        leaky((accm[mt][nt][4 * k + e] + accx[mt][nt][4 * k + e] * LO_UNSCALE) + bv[e]);
            }
    }
    __syncthreads();

    // ---- head: thread = frame (seg_head_kernel's arithmetic) ------------------------------------------
    const int t = t0 + tid;
    if (tid < 128 && t < p.rows) {
        const float* x = m1 + tid * M1_PITCH;
        float lg[8];
        for (int c = 0; c < p.classes; ++c) {
            float acc = 0.f;
            const float* wc = par + PAR_CW + c * 128;
#pragma unroll 8
            for (int k = 0; k < 128; ++k) acc = fmaf(x[k], wc[k], acc);
            lg[c] = acc + par[PAR_CB + c];
        }
        float s[8];
        dz_seg_decide(lg, p.classes, p.K, p.powerset, s);
        const int b = t / p.F, f = t - b * p.F;
        if (p.wave_mom && dz_ws_bad(p.wave_mom, b))          // a window with NaN / Inf samples: NaN rows, like the reference
            for (int k = 0; k < p.K; ++k) s[k] = __builtin_nanf("");
        for (int k = 0; k < p.K; ++k) p.seg[(long long)t * p.K + k] = s[k];
        if (p.wout) {
            float wv[8];
            dz_osp_frame(s, p.K, p.gamma, p.beta, wv);
            for (int k = 0; k < p.K; ++k) p.wout[((long long)b * p.K + k) * p.F + f] = wv[k];
        }
    }
}

}  // namespace

int dz_launch_mlp_head(const DzMlpHead& p_in, hipStream_t st) {
    DzMlpHead p = p_in;
    if (!p.oflag) p.oflag = dz_cur_oflag;
    DZ_REQUIRE(p.Xsplit && p.W0split && p.W1split && p.b0 && p.b1 && p.cw && p.cb && p.seg, "mlp_head: NULL argument");
    DZ_REQUIRE(p.rows > 0 && p.F > 0 && p.rows % p.F == 0, "mlp_head: %d rows are not whole chunks of %d frames", p.rows, p.F);
    DZ_REQUIRE(p.classes >= 1 && p.classes <= 8 && p.K >= 1 && p.K <= 8 && (p.powerset || p.K == p.classes),
               "mlp_head: classes %d / speakers %d", p.classes, p.K);
    DZ_REQUIRE((long long)p.rows * 256 * 2 < (1ll << 31), "mlp_head: plane exceeds the buffer range");
    DZ_REQUIRE(p.xplane == (long long)p.rows * 256, "mlp_head: kb-major input planes of exactly `rows` rows (xplane = rows * 256)");
    static DzAttrOnce attr_once;
    DZ_HIP(attr_once.raise((const void*)mlp_head_kernel, (int)LDS_BYTES));
    DZ_LAUNCH(mlp_head_kernel, dim3((p.rows + 127) / 128), dim3(256), LDS_BYTES, st, p);
    DZ_HIP(hipGetLastError());
    return 0;
}
