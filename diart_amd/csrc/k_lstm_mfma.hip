// Bidirectional-LSTM recurrence (hidden 128) on the f16 matrix cores, 16 chains per workgroup.
// Same contract as k_lstm.hip (pyannote PyanNet's nn.LSTM(60,128,4,bidirectional), called from
// /root/reference/src/diart/models.py:133; SURVEY.md Appendix A.1 step 2, kernel K5); this is the
// default-precision ("f16x3") form, k_lstm.hip stays the exact-f32 form.
//
// k_lstm.hip runs ONE chain per CU (W_hh in VGPRs, f32 VALU): the shortest step the chip can do
// (~0.6 us) but 128 CUs for 64 chunks x 2 directions, whose matrix pipes idle for the whole
// recurrence.  Here one workgroup steps 16 chains of one direction at once:
//
//     gates[512 x 16] = W_hh[512 x 128] . H[128 x 16]          per step, on v_mfma_f32_16x16x32_f16
//
// with both operands split into (hi, lo) f16 pairs (x = hi + lo * 2^-11, 22 mantissa bits; three
// MFMAs per product into two f32 accumulators, exactly the arithmetic of k_gemm_split.hip).  64
// chunks x 2 directions occupy 8 CUs instead of 128; a step takes about as long as before (384 MFMAs
// = 1536 matrix-pipe cycles per SIMD plus the gate activations), so the recurrence is still the
// latency of the segmentation chain, but 120 CUs are returned to the GEMMs of the other streams.
//
//   workgroup = 512 threads = 8 waves; wave w owns hidden units 16w .. 16w+15 (64 of the 512 gate
//   rows), its slice of W_hh lives in VGPRs for all T steps: 4 row tiles x 4 k-steps x (hi, lo)
//   fragments = 128 registers.  Row tile j of wave w is ordered so that row 4q + r is gate r
//   (i, f, g, o) of unit 16w + 4q + j: after the MFMAs lane (q, n) holds, for each of its 4 tiles,
//   the four gate pre-activations of ONE (unit, chain) cell — the cell update needs no cross-lane
//   traffic, and the lane's four cells are units 16w+4q .. +3: consecutive, so h_t goes to LDS as one
//   8-byte store per f16 plane and to HBM as one 16-byte store.
//   H_t lives in LDS as two f16 planes [16 chains][128 units] (256-byte rows, 16-byte chunks XOR
//   swizzled with the chain index: the B-fragment reads of a 16-lane group hit 16 distinct slots),
//   double buffered, one LDS-only barrier per step.  The x-projection of step s+2 is prefetched
//   into registers while step s runs.
#include "dz_common.h"
#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
    return 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.88539008177792681f * x)) - 1.f;
}
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

constexpr int CH = 16;                 // chains per workgroup (the N of the MFMA)
constexpr int PLANE = CH * 256;        // bytes of one f16 plane of H
constexpr float LO_SCALE = 2048.f, LO_UNSCALE = 1.f / 2048.f;

// UM: gx columns are unit-major (dir*512 + unit*4 + gate, what dz_seg_forward's projection GEMM
// writes) instead of PyTorch's gate-major (dir*512 + gate*128 + unit)
template <bool UM>
__global__ __launch_bounds__(512) void lstm_mfma_kernel(const float* __restrict__ gx,
                                                        const unsigned short* __restrict__ whs,
                                                        float* __restrict__ hout,
                                                        unsigned short* __restrict__ hsp,
                                                        long long hplane, int B, int T) {
    __shared__ __attribute__((aligned(16))) char hs[2 * 2 * PLANE];   // [buf][plane][chain][128]
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, n = l & 15, q = l >> 4;
    const int dir = blockIdx.y;
    const int b = blockIdx.x * CH + n;
    const bool valid = b < B;
    const int bb = valid ? b : B - 1;

    // ---- W_hh slice of this wave -> registers (A fragments: lane = row n of the tile, k-group q)
    f16x8 wh[4][4], wl[4][4];
    {
        const unsigned short* Wd = whs + (long long)dir * 2 * 512 * 128;
        const int row_base = (n & 3) * 128 + 16 * w + 4 * (n >> 2);   // gate (n&3), unit 16w + 4(n>>2) + j
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const long long o = (long long)(row_base + j) * 128 + 32 * ks + 8 * q;
                wh[j][ks] = *reinterpret_cast<const f16x8*>(Wd + o);
                wl[j][ks] = *reinterpret_cast<const f16x8*>(Wd + 512 * 128 + o);
            }
    }
    // h_{-1} = 0 in buffer 0
    for (int i = tid; i < 2 * PLANE / 16; i += 512)
        reinterpret_cast<f32x4*>(hs)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int rd_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rd_off[ks] = n * 256 + (((4 * ks + q) ^ n) << 4);
    const int wr_off = n * 256 + (((2 * w + (q >> 1)) ^ n) << 4) + 8 * (q & 1);

    // step s works on frame tt(s) = s (forward) or T-1-s (backward)
    const long long tstep = dir ? -1024 : 1024;
    const float* gptr = gx + ((long long)bb * T + (dir ? T - 1 : 0)) * 1024 + dir * 512 +
                        (UM ? 64 * w + 16 * q : 16 * w + 4 * q);
    const long long hbase = ((long long)bb * T + (dir ? T - 1 : 0)) * 256 + dir * 128 + 16 * w + 4 * q;
    const long long hstep = dir ? -256 : 256;

    struct GX { f32x4 v[4]; };
    auto gload = [&](GX& g, int s) {
        const float* p = gptr + (long long)(s < T ? s : T - 1) * tstep;
#pragma unroll
        for (int i = 0; i < 4; ++i) g.v[i] = *reinterpret_cast<const f32x4*>(p + (UM ? 4 * i : 128 * i));
    };

    float c[4] = {0.f, 0.f, 0.f, 0.f};
    auto step = [&](int s, const GX& g) {
        const char* hb = hs + (s & 1) * 2 * PLANE;
        f16x8 bh[4], bl[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bh[ks] = *reinterpret_cast<const f16x8*>(hb + rd_off[ks]);
            bl[ks] = *reinterpret_cast<const f16x8*>(hb + PLANE + rd_off[ks]);
        }
        f32x4 am[4], ax[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) am[j] = ax[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                am[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j][ks], bh[ks], am[j], 0, 0, 0);
                ax[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j][ks], bl[ks], ax[j], 0, 0, 0);
                ax[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j][ks], bh[ks], ax[j], 0, 0, 0);
            }
        // cell update: tile j = unit 16w + 4q + j of chain n, accumulator row r = gate r
        f32x4 hv;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pre[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                pre[r] = (am[j][r] + ax[j][r] * LO_UNSCALE) + (UM ? g.v[j][r] : g.v[r][j]);
            const float ig = fast_sigmoid(pre[0]), fg = fast_sigmoid(pre[1]), gg = fast_tanh(pre[2]),
                        og = fast_sigmoid(pre[3]);
            c[j] = fg * c[j] + ig * gg;
            hv[j] = og * fast_tanh(c[j]);
        }
        const f16x4 hhi = __builtin_convertvector(hv, f16x4);
        const f16x4 hlo = __builtin_convertvector((hv - __builtin_convertvector(hhi, f32x4)) * LO_SCALE, f16x4);
        char* hn = hs + ((s + 1) & 1) * 2 * PLANE;
        *reinterpret_cast<f16x4*>(hn + wr_off) = hhi;
        *reinterpret_cast<f16x4*>(hn + PLANE + wr_off) = hlo;
        if (valid) {
            // f32 and / or the (hi, lo * 2^11) planes of a k_gemm_pre.hip consumer: the LDS image of
            // h_t already is that representation
            const long long o = hbase + (long long)s * hstep;
            if (hout) *reinterpret_cast<f32x4*>(hout + o) = hv;
            if (hsp) {                                   // kb-major planes (dz_kb): row o / 256, column o % 256
                const long long ok = dz_kb(o >> 8, (int)(o & 255), hplane >> 8);
                *reinterpret_cast<f16x4*>(hsp + ok) = hhi;
                *reinterpret_cast<f16x4*>(hsp + hplane + ok) = hlo;
            }
        }
        lds_barrier();
    };

    GX g0, g1;
    gload(g0, 0);
    gload(g1, 1);
    __syncthreads();
    int s = 0;
    for (; s + 2 <= T; s += 2) {
        step(s, g0);
        gload(g0, s + 2);
        step(s + 1, g1);
        gload(g1, s + 3);
    }
    if (s < T) step(s, g0);
}


#ifdef DZ_EXPERIMENTS   // variants 1 / 2: measured, parity-tested, never the default
// ---------------------------------------------------------------------------------------------
// Variant with ONE accumulator per tile and the activation scales folded into the weights:
//   W' = W_hh * s_row * 2^SH,  s_row = -log2(e) (i, f, o rows) or -2 log2(e) (g rows), split as
//   hi = f16(W'), lo = f16(W' - hi)  (NO 2^11 scale on lo),
//   H' = h * 2^SH split the same way;  acc = gx * s_row * 4^SH + W'hi.H'hi + W'hi.H'lo + W'lo.H'hi
//   so that exp2(acc * 4^-SH) is exp(-pre) (exp(-2 pre) for g): sigmoid = rcp(1 + exp2(..)).
// SH = 0: the low parts are (mostly) f16 SUBNORMALS (|lo| <= 2^-12 |x|): exact as long as the matrix
// pipe does not flush f16 denormals; SH = 8 keeps every low part a normal f16 for |x| >= 2^-10 at
// the price of one multiply per gate.  Per cell 25 (SH = 0) / 30 VALU instructions instead of 36,
// and 16 registers fewer.
// ---------------------------------------------------------------------------------------------
template <bool UM, int SH>
__global__ __launch_bounds__(512) void lstm_mfma1_kernel(const float* __restrict__ gx,
                                                         const unsigned short* __restrict__ whs,
                                                         float* __restrict__ hout,
                                                         unsigned short* __restrict__ hsp,
                                                         long long hplane, int B, int T) {
    __shared__ __attribute__((aligned(16))) char hs[2 * 2 * PLANE];   // [buf][plane][chain][128]
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, n = l & 15, q = l >> 4;
    const int dir = blockIdx.y;
    const int b = blockIdx.x * CH + n;
    const bool valid = b < B;
    const int bb = valid ? b : B - 1;
    constexpr float S2 = (float)(1 << SH) * (float)(1 << SH);     // scale of the accumulator
    constexpr float HS = (float)(1 << SH);                         // scale of H in LDS
    constexpr float LOG2E = 1.44269504088896341f;

    f16x8 wh[4][4], wl[4][4];
    {
        const unsigned short* Wd = whs + (long long)dir * 2 * 512 * 128;
        const int row_base = (n & 3) * 128 + 16 * w + 4 * (n >> 2);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const long long o = (long long)(row_base + j) * 128 + 32 * ks + 8 * q;
                wh[j][ks] = *reinterpret_cast<const f16x8*>(Wd + o);
                wl[j][ks] = *reinterpret_cast<const f16x8*>(Wd + 512 * 128 + o);
            }
    }
    for (int i = tid; i < 2 * PLANE / 16; i += 512)
        reinterpret_cast<f32x4*>(hs)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int rd_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rd_off[ks] = n * 256 + (((4 * ks + q) ^ n) << 4);
    const int wr_off = n * 256 + (((2 * w + (q >> 1)) ^ n) << 4) + 8 * (q & 1);
    const long long tstep = dir ? -1024 : 1024;
    const float* gptr = gx + ((long long)bb * T + (dir ? T - 1 : 0)) * 1024 + dir * 512 +
                        (UM ? 64 * w + 16 * q : 16 * w + 4 * q);
    const long long hbase = ((long long)bb * T + (dir ? T - 1 : 0)) * 256 + dir * 128 + 16 * w + 4 * q;
    const long long hstep = dir ? -256 : 256;

    struct GX { f32x4 v[4]; };
    auto gload = [&](GX& g, int s) {
        const float* p = gptr + (long long)(s < T ? s : T - 1) * tstep;
#pragma unroll
        for (int i = 0; i < 4; ++i) g.v[i] = *reinterpret_cast<const f32x4*>(p + (UM ? 4 * i : 128 * i));
    };
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    auto step = [&](int s, const GX& g) {
        const char* hb = hs + (s & 1) * 2 * PLANE;
        f16x8 bh[4], bl[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bh[ks] = *reinterpret_cast<const f16x8*>(hb + rd_off[ks]);
            bl[ks] = *reinterpret_cast<const f16x8*>(hb + PLANE + rd_off[ks]);
        }
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[j][r] = (UM ? g.v[j][r] : g.v[r][j]) * (r == 2 ? -2.f * LOG2E * S2 : -LOG2E * S2);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j][ks], bh[ks], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j][ks], bl[ks], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j][ks], bh[ks], acc[j], 0, 0, 0);
            }
        f32x4 hv;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float e[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                e[r] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(SH ? acc[j][r] * (1.f / S2) : acc[j][r]));
            const float gg = 2.f * e[2] - 1.f;
            c[j] = e[1] * c[j] + e[0] * gg;
            const float sc = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.f * LOG2E * c[j]));
            hv[j] = e[3] * (2.f * sc - 1.f);
        }
        const f32x4 hsc = SH ? hv * HS : hv;
        const f16x4 hhi = __builtin_convertvector(hsc, f16x4);
        const f16x4 hlo = __builtin_convertvector(hsc - __builtin_convertvector(hhi, f32x4), f16x4);
        char* hn = hs + ((s + 1) & 1) * 2 * PLANE;
        *reinterpret_cast<f16x4*>(hn + wr_off) = hhi;
        *reinterpret_cast<f16x4*>(hn + PLANE + wr_off) = hlo;
        if (valid) {
            const long long o = hbase + (long long)s * hstep;
            if (hout) *reinterpret_cast<f32x4*>(hout + o) = hv;
            if (hsp) {
                const f16x4 ohi = __builtin_convertvector(hv, f16x4);
                const long long ok = dz_kb(o >> 8, (int)(o & 255), hplane >> 8);
                *reinterpret_cast<f16x4*>(hsp + ok) = ohi;
                *reinterpret_cast<f16x4*>(hsp + hplane + ok) =
                    __builtin_convertvector((hv - __builtin_convertvector(ohi, f32x4)) * LO_SCALE, f16x4);
            }
        }
        lds_barrier();
    };
    GX g0, g1;
    gload(g0, 0);
    gload(g1, 1);
    __syncthreads();
    int s = 0;
    for (; s + 2 <= T; s += 2) {
        step(s, g0);
        gload(g0, s + 2);
        step(s + 1, g1);
        gload(g1, s + 3);
    }
    if (s < T) step(s, g0);
}

#endif  // DZ_EXPERIMENTS

}  // namespace

// ---------------------------------------------------------------------------------------------
// Variant 3 = variant 0 (two accumulators, lo planes scaled by 2^11) with the x-projection fetched
// by LDS-DMA instead of through registers.  In the pipeline the register-prefetched kernel runs at
// 2.1 us per step against 1.3 us alone: its loads are issued two steps (~2.6 us) ahead, which a
// loaded memory system does not always cover, and more look-ahead has no registers left (238 of
// 256).  LDS-DMA needs none: a 4-slot ring of [16 chains][2 KiB] rows (unit-major gx: the 512 floats
// of one direction are contiguous) is filled three steps ahead, wave w fetching chains 2w, 2w+1
// (4 x 1 KiB pieces per step); the end-of-step barrier is preceded by a counted vmcnt so that the
// pieces of the NEXT step have landed.  Chain rows are 2064 bytes apart in LDS (16 B of padding:
// the 16 chains of a read hit different bank groups).  gx must be unit-major.
// ---------------------------------------------------------------------------------------------
constexpr int GX_ROW = 2048 + 16;           // bytes per chain row of a ring slot
constexpr int GX_SLOT = CH * GX_ROW;
constexpr int GX_NSLOT = 4;

__global__ __launch_bounds__(512) void lstm_mfma_dma_kernel(const float* __restrict__ gx,
                                                            const unsigned short* __restrict__ whs,
                                                            float* __restrict__ hout,
                                                            unsigned short* __restrict__ hsp,
                                                            long long hplane, int B, int T) {
    // one LDS object: H planes [2 buf][2 plane][16][256 B] | gx ring [4][16][2064 B]
    __shared__ __attribute__((aligned(16))) char lds[2 * 2 * PLANE + GX_NSLOT * GX_SLOT];
    char* hs = lds;
    char* gxr = lds + 2 * 2 * PLANE;
    const int tid = threadIdx.x, l = tid & 63, n = l & 15, q = l >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y;
    const int b = blockIdx.x * CH + n;
    const bool valid = b < B;
    const int bb = valid ? b : B - 1;

    f16x8 wh[4][4], wl[4][4];
    {
        const unsigned short* Wd = whs + (long long)dir * 2 * 512 * 128;
        const int row_base = (n & 3) * 128 + 16 * w + 4 * (n >> 2);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const long long o = (long long)(row_base + j) * 128 + 32 * ks + 8 * q;
                wh[j][ks] = *reinterpret_cast<const f16x8*>(Wd + o);
                wl[j][ks] = *reinterpret_cast<const f16x8*>(Wd + 512 * 128 + o);
            }
    }
    for (int i = tid; i < 2 * PLANE / 16; i += 512)
        reinterpret_cast<f32x4*>(hs)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int rd_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rd_off[ks] = n * 256 + (((4 * ks + q) ^ n) << 4);
    const int wr_off = n * 256 + (((2 * w + (q >> 1)) ^ n) << 4) + 8 * (q & 1);
    const long long hbase = ((long long)bb * T + (dir ? T - 1 : 0)) * 256 + dir * 128 + 16 * w + 4 * q;
    const long long hstep = dir ? -256 : 256;

    // ---- LDS-DMA of the x-projection: this wave fetches chains 2w and 2w+1 ----------------------
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)gx, 0, (unsigned)((long long)B * T * 4096 < 0xffffffffLL ? (long long)B * T * 4096 : 0xffffffffLL),
        0x00020000);
    int voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int chain = blockIdx.x * CH + 2 * w + (i >> 1);
        const int cb = chain < B ? chain : B - 1;
        voff[i] = cb * T * 4096 + dir * 2048 + (i & 1) * 1024 + l * 16;
    }
    auto fetch = [&](int s) {                 // frame of step s: s (forward) or T-1-s (backward)
        const int tt = dir ? T - 1 - s : s;
        char* slot = gxr + (s & (GX_NSLOT - 1)) * GX_SLOT;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsrc, (__attribute__((address_space(3))) void*)(slot + (2 * w + (i >> 1)) * GX_ROW + (i & 1) * 1024),
                16, voff[i], tt * 4096, 0, 0);
    };
    // this lane's 16 floats (4 cells x 4 gates) of chain n: floats (16w + 4q) * 4 .. +15 of the row
    const int g_off = n * GX_ROW + (64 * w + 16 * q) * 4;

    float c[4] = {0.f, 0.f, 0.f, 0.f};
    fetch(0);
    if (T > 1) fetch(1);
    if (T > 2) fetch(2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // step 0's four pieces
    __syncthreads();
    for (int s = 0; s < T; ++s) {
        if (s + 3 < T) fetch(s + 3);
        const char* hb = hs + (s & 1) * 2 * PLANE;
        const char* gs = gxr + (s & (GX_NSLOT - 1)) * GX_SLOT + g_off;
        f16x8 bh[4], bl[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bh[ks] = *reinterpret_cast<const f16x8*>(hb + rd_off[ks]);
            bl[ks] = *reinterpret_cast<const f16x8*>(hb + PLANE + rd_off[ks]);
        }
        f32x4 gv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) gv[j] = *reinterpret_cast<const f32x4*>(gs + 16 * j);
        f32x4 am[4], ax[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) am[j] = ax[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                am[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j][ks], bh[ks], am[j], 0, 0, 0);
                ax[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j][ks], bl[ks], ax[j], 0, 0, 0);
                ax[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j][ks], bh[ks], ax[j], 0, 0, 0);
            }
        f32x4 hv;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pre[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) pre[r] = (am[j][r] + ax[j][r] * LO_UNSCALE) + gv[j][r];
            const float ig = fast_sigmoid(pre[0]), fg = fast_sigmoid(pre[1]), gg = fast_tanh(pre[2]),
                        og = fast_sigmoid(pre[3]);
            c[j] = fg * c[j] + ig * gg;
            hv[j] = og * fast_tanh(c[j]);
        }
        const f16x4 hhi = __builtin_convertvector(hv, f16x4);
        const f16x4 hlo = __builtin_convertvector((hv - __builtin_convertvector(hhi, f32x4)) * LO_SCALE, f16x4);
        char* hn = hs + ((s + 1) & 1) * 2 * PLANE;
        *reinterpret_cast<f16x4*>(hn + wr_off) = hhi;
        *reinterpret_cast<f16x4*>(hn + PLANE + wr_off) = hlo;
        if (valid) {
            const long long o = hbase + (long long)s * hstep;
            if (hout) *reinterpret_cast<f32x4*>(hout + o) = hv;
            if (hsp) {                                   // kb-major planes (dz_kb): row o / 256, column o % 256
                const long long ok = dz_kb(o >> 8, (int)(o & 255), hplane >> 8);
                *reinterpret_cast<f16x4*>(hsp + ok) = hhi;
                *reinterpret_cast<f16x4*>(hsp + hplane + ok) = hlo;
            }
        }
        // the pieces of step s+1 (issued >= 2 steps ago) must have landed before anybody passes the
        // barrier: at most the 8 pieces of steps s+2, s+3 may still be in flight (this step's stores
        // are younger and only make the wait stricter)
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

// ---------------------------------------------------------------------------------------------
// Variant 4 = the software-pipelined form (round 6).  What a step of variant 3 costs was taken apart with
// timing-only builds (tools/rec_modes.py, profiles/r06a_rec_modes_*.json): 96 MFMAs per SIMD are ~1 990 cycles
// (v_mfma_f32_16x16x32_f16 issues every ~20 cycles, not 16), and every other part of the step ran AFTER them:
// the cell update of the last tiles (~75 vector instructions, 22 transcendental) behind the last MFMA, then
// the LDS round trip of h and the barrier.  Here the step is cut in two along K so that half of its MFMAs
// never depend on the cells that are still being computed:
//
//   * k-permutation.  Lane (wave w, q, n) owns the cells of units 16w + 4q + j, j = 0..3 (tile j).  The
//     contraction index is re-ordered so that k' = 64 (j >> 1) + 2 (4w + q) + (j & 1): k-half 0 is exactly the
//     cells j = 0, 1 of every lane, k-half 1 the cells j = 2, 3 (weights.py permutes the columns of W_hh to
//     match; h_t in LDS lives in k' order, a lane's pair of cells is one 4-byte store per plane).
//   * schedule of step s, M(j, kh) = the 6 MFMAs of tile j over k-half kh:
//       barrier P_s   (h_{s-1} half 0 visible; x-projection of step s landed)
//         M(3,1) of step s-1 || exps of cell 2' | M(0,0) || exps of cell 3' | M(1,0) || pair (2,3)' -> h_{s-1} half 1 -> LDS
//       barrier Q_s   (h_{s-1} half 1 visible)
//         M(2,0) | M(0,1) | M(1,1) || exps of cell 0 | M(3,0) || exps of cell 1 | M(2,1) || pair (0,1) -> h_s half 0 -> LDS
//     Every cell update runs under MFMAs that do not need it, and the LDS read after each barrier is covered by
//     six MFMAs whose operands were fetched before it.
//   * fewer vector instructions per cell (a SIMD issues MFMAs and vector instructions through one port,
//     DESIGN.md 5.4): the x-projection is the C operand of the first MFMA of a tile (no add, no zero fill); the
//     activation scales are folded into the weights (gx and W_hh rows carry -log2 e, the g rows -2 log2 e:
//     weights.py, `lstm_variant` 4), so an accumulator IS the exp2 argument; and with E_x = exp(-x)
//         c' = c / (1 + E_f) + (1 - E_g) / ((1 + E_i)(1 + E_g)),   h = (1 - E_c) / ((1 + E_o)(1 + E_c))
//     needs 5 exp2 + 3 rcp per cell instead of 5 + 5 (E_g, E_c clamped at 2^64: the quotients are then exact
//     limits, never inf * 0).  Plain f32 instructions only: packed-f32 ones beside MFMAs cost more than the
//     two they replace (MI355X_MICROARCH.md price list; this file is compiled with -fno-slp-vectorize).
//   * h_t -> HBM through LDS.  A vector-memory instruction costs its wave 40 - 60 issue cycles whatever it
//     moves (six 4- / 8-byte stores per wave and step: +1 900 cycles per step): the planes of h_{s-1} are
//     complete in LDS behind Q_s, each wave reads 16 bytes per lane of them (one plane, four chains, 8 units per
//     lane: two ds_read_b64, k' -> unit order is a dword shuffle) and writes them with ONE 16-byte store —
//     64 contiguous bytes per (chain, k-block), the kb-major layout of the consumer.  h is double buffered in
//     LDS for that.  The four LDS-DMA pieces of the x-projection are spread over the step for the same reason.
// gx: unit-major AND pre-scaled; whh_split: lstm_whh_planes(whh, 4).  F32OUT (kernel-level entry only): h also as
// f32 rows (8-byte stores per pair of cells, one barrier late).  MODE (experiments build, timing only, WRONG
// results): 4 = no stores of h, 16 = no cell arithmetic, 32 = no LDS-DMA, 128 = no barriers.
// ---------------------------------------------------------------------------------------------
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr float K2 = -2.88539008177792681f;

// stage A (one cell): the exp2 arguments of its four gates (main + cross * 2^-11) -> E = exp(-x) (i, f, o), exp(-2x) (g)
__device__ __forceinline__ f32x4 cell_exps(const f32x4& am, const f32x4& ax) {
    const float pi = __builtin_fmaf(ax[0], LO_UNSCALE, am[0]);
    const float pf = __builtin_fmaf(ax[1], LO_UNSCALE, am[1]);
    const float pg = __builtin_fminf(__builtin_fmaf(ax[2], LO_UNSCALE, am[2]), 64.f);
    const float po = __builtin_fmaf(ax[3], LO_UNSCALE, am[3]);
    return (f32x4){__builtin_amdgcn_exp2f(pi), __builtin_amdgcn_exp2f(pf), __builtin_amdgcn_exp2f(pg), __builtin_amdgcn_exp2f(po)};
}
// stage B (one cell): cs = -2 log2(e) c is the running cell state
__device__ __forceinline__ float cell_finish(const f32x4& e, float& cs) {
    const float u = 1.f + e[2];
    const float ra = __builtin_amdgcn_rcpf(1.f + e[1]);
    const float rb = __builtin_amdgcn_rcpf(__builtin_fmaf(e[0], u, u));
    cs = __builtin_fmaf(cs, ra, __builtin_fmaf(e[2], -K2, K2) * rb);
    const float ec = __builtin_amdgcn_exp2f(__builtin_fminf(cs, 64.f));
    const float v = 1.f + ec;
    const float rd = __builtin_amdgcn_rcpf(__builtin_fmaf(e[3], v, v));
    return __builtin_fmaf(-ec, rd, rd);
}

template <bool F32OUT, int MODE>
__global__ __launch_bounds__(512) void lstm_mfma_pipe_kernel(const float* __restrict__ gx,
                                                             const unsigned short* __restrict__ whs,
                                                             float* __restrict__ hout,
                                                             unsigned short* __restrict__ hsp,
                                                             long long hplane, int B, int T) {
    // one LDS object: H [2 buf][2 plane][16 chains][256 B] (k' order) | gx ring [4][16][2064 B]
    __shared__ __attribute__((aligned(16))) char lds[4 * PLANE + GX_NSLOT * GX_SLOT];
    char* hs = lds;
    char* gxr = lds + 4 * PLANE;
    const int tid = threadIdx.x, l = tid & 63, n = l & 15, q = l >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y;
    const int b = blockIdx.x * CH + n;
    const bool valid = b < B;
    const int bb = valid ? b : B - 1;

    f16x8 wh[4][4], wl[4][4];
    {
        const unsigned short* Wd = whs + (long long)dir * 2 * 512 * 128;
        const int row_base = (n & 3) * 128 + 16 * w + 4 * (n >> 2);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const long long o = (long long)(row_base + j) * 128 + 32 * ks + 8 * q;
                wh[j][ks] = *reinterpret_cast<const f16x8*>(Wd + o);
                wl[j][ks] = *reinterpret_cast<const f16x8*>(Wd + 512 * 128 + o);
            }
    }
    for (int i = tid; i < 4 * PLANE / 16; i += 512)        // h_{-1} = 0 (both buffers)
        reinterpret_cast<f32x4*>(hs)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int rd_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rd_off[ks] = n * 256 + (((4 * ks + q) ^ n) << 4);
    // cells (0, 1) -> k' = 2 (4w + q) + {0, 1}: 16-byte chunk w; cells (2, 3) -> 64 + the same: chunk 8 + w
    const int wr_off0 = n * 256 + ((w ^ n) << 4) + 4 * q;
    const int wr_off1 = n * 256 + (((8 + w) ^ n) << 4) + 4 * q;

    // ---- export of h_{s-1} (complete in LDS behind Q_s): wave w -> plane w >> 2, chains 4 (w & 3) .. + 3; lane -> chain
    // + (l >> 4), units 8 g .. 8 g + 7 with g = l & 15 = k' 4g .. 4g + 3 (units 8g, 8g+1, 8g+4, 8g+5) and 64 + the same
    // (8g+2, 8g+3, 8g+6, 8g+7).  Destination: kb-major plane, row = chunk * T + frame, 64 contiguous bytes per 4 lanes.
    const long long R = hplane >> 8;                                   // rows of a plane
    const int ep = w >> 2, en = 4 * (w & 3) + (l >> 4), eg = l & 15;
    const int e_rda = ep * PLANE + en * 256 + (((eg >> 1) ^ en) << 4) + 8 * (eg & 1);
    const int e_rdb = ep * PLANE + en * 256 + (((8 + (eg >> 1)) ^ en) << 4) + 8 * (eg & 1);
    const int e_chunk = blockIdx.x * CH + en;
    const unsigned dead = 0x80000000u;
    const unsigned e_vo = e_chunk < B ? (unsigned)((((long long)(dir * 4 + (eg >> 2)) * R + (long long)e_chunk * T) * 64) + 16 * (eg & 3) +
                                                   (long long)ep * hplane * 2) : dead;
    // (the size words through readfirstlane: hipcc computes them on the vector unit, and a descriptor it believes to be
    // divergent puts every access in a waterfall loop)
    const unsigned pl_bytes = __builtin_amdgcn_readfirstlane(hsp ? (unsigned)(hplane * 4) : 0u);
    const unsigned f_bytes = __builtin_amdgcn_readfirstlane(hout ? (unsigned)((long long)B * T * 1024) : 0u);
    const unsigned gx_bytes = __builtin_amdgcn_readfirstlane(
        (unsigned)((long long)B * T * 4096 < 0xffffffffLL ? (long long)B * T * 4096 : 0xffffffffLL));
    const __amdgpu_buffer_rsrc_t r_pl = __builtin_amdgcn_make_buffer_rsrc((void*)hsp, 0, pl_bytes, 0x00020000);
    // f32 rows (F32OUT): voffset = the lane's chain and column, soffset = the frame; absent output: empty descriptor
    const __amdgpu_buffer_rsrc_t r_f = __builtin_amdgcn_make_buffer_rsrc((void*)hout, 0, f_bytes, 0x00020000);
    const unsigned vo_f = valid ? (unsigned)((long long)bb * T * 1024 + (dir * 128 + 16 * w + 4 * q) * 4) : dead;

    // ---- LDS-DMA of the x-projection: this wave fetches chains 2w and 2w+1 (as variant 3), one piece at a time
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)gx, 0, gx_bytes, 0x00020000);
    int voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int chain = blockIdx.x * CH + 2 * w + (i >> 1);
        const int cb = chain < B ? chain : B - 1;
        voff[i] = cb * T * 4096 + dir * 2048 + (i & 1) * 1024 + l * 16;
    }
    auto frame = [&](int s) { return dir ? T - 1 - s : s; };
    auto fetch_piece = [&](int s, int slot_no, int i) {     // piece i of the frame of step s (clamped to the last step) -> ring slot
        if constexpr (MODE & 32) return;
        const int tt = frame(s < T ? s : T - 1);
        char* slot = gxr + slot_no * GX_SLOT;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rsrc, (__attribute__((address_space(3))) void*)(slot + (2 * w + (i >> 1)) * GX_ROW + (i & 1) * 1024),
            16, voff[i], tt * 4096, 0, 0);
    };
    const int g_off = n * GX_ROW + (64 * w + 16 * q) * 4;

#define DZ_M3(AM, AX, J, KS)                                                                  \
    AM = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[J][KS], bh[KS], AM, 0, 0, 0);              \
    AX = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[J][KS], bl[KS], AX, 0, 0, 0);              \
    AX = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[J][KS], bh[KS], AX, 0, 0, 0);
#define DZ_SB() __builtin_amdgcn_sched_barrier(0)
#define DZ_BARRIER(WAIT) do { if constexpr (MODE & 128) asm volatile("s_waitcnt " WAIT ::: "memory");      \
                              else asm volatile("s_waitcnt " WAIT "\n\ts_barrier" ::: "memory"); } while (0)

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 am[4], ax[4];
    f16x8 bh[4], bl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) am[j] = ax[j] = zero4;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bh[ks] = bl[ks] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
    float cs[4] = {0.f, 0.f, 0.f, 0.f};                      // -2 log2(e) x the cell state of the lane's four cells
    f32x2 p01_f = {0.f, 0.f}, p23_f = {0.f, 0.f};            // F32OUT: h of the last two pairs, stored one barrier late

    // pair of cells -> (hi, lo * 2^11) f16 pairs -> one 4-byte LDS store per plane
    auto put_pair = [&](float h0, float h1, char* at) {
        const f16x2 hi = {(_Float16)h0, (_Float16)h1};
        const f16x2 lo = {(_Float16)((h0 - (float)hi[0]) * LO_SCALE), (_Float16)((h1 - (float)hi[1]) * LO_SCALE)};
        *reinterpret_cast<unsigned*>(at) = __builtin_bit_cast(unsigned, hi);
        *reinterpret_cast<unsigned*>(at + PLANE) = __builtin_bit_cast(unsigned, lo);
    };
    auto exps = [&](const f32x4& m, const f32x4& x) -> f32x4 {
        if constexpr (MODE & 16) return m + x; else return cell_exps(m, x);
    };
    auto finish = [&](const f32x4& e, float& c) -> float {
        if constexpr (MODE & 16) { c += e[0]; return c; } else return cell_finish(e, c);
    };

    // behind barrier P_s: read k-half 0 of h_{s-1} and the x-projection of step s; finish step s-1 (tile 3, cells 2
    // and 3) under M(3,1)', M(0,0), M(1,0).  PHASE 0: s = 0 (there is no step s-1), 1: s = 1 (no step s-2), 2: s >= 2.
    // S4 = s & 3 at compile time (the main loop is unrolled four times): ring slots and h buffers are then immediate
    // offsets — every scalar or address instruction in the loop costs the SIMD an issue slot the MFMAs do not hide.
    f32x4 n0m, n0x, n1m, n1x, n2m, n2x, n3m, n3x;
    auto first_half = [&](auto phase_tag, auto s4_tag, int s) {
        constexpr int PHASE = decltype(phase_tag)::value;
        constexpr int S4 = decltype(s4_tag)::value;
        const char* hprev = hs + ((S4 + 1) & 1) * 2 * PLANE;                 // buffer of h_{s-1}
        const char* gs = gxr + S4 * GX_SLOT + g_off;
        if constexpr (F32OUT && !(MODE & 4)) {
            if constexpr (PHASE >= 1) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, p01_f), r_f, vo_f, frame(s - 1) * 1024, 0);
            if constexpr (PHASE >= 2) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, p23_f), r_f, vo_f + 8, frame(s - 2) * 1024, 0);
        }
        bh[0] = *reinterpret_cast<const f16x8*>(hprev + rd_off[0]);
        bl[0] = *reinterpret_cast<const f16x8*>(hprev + PLANE + rd_off[0]);
        bh[1] = *reinterpret_cast<const f16x8*>(hprev + rd_off[1]);
        bl[1] = *reinterpret_cast<const f16x8*>(hprev + PLANE + rd_off[1]);
        n0m = *reinterpret_cast<const f32x4*>(gs);
        n1m = *reinterpret_cast<const f32x4*>(gs + 16);
        n2m = *reinterpret_cast<const f32x4*>(gs + 32);
        n3m = *reinterpret_cast<const f32x4*>(gs + 48);
        n0x = n1x = n2x = n3x = zero4;
        fetch_piece(s + 3, (S4 + 3) & 3, 0);
        if constexpr (PHASE >= 1) {
            DZ_SB();
            // M(3,1) of step s-1 (operands fetched behind Q_{s-1}: covers the reads above) || exps of cell 2
            DZ_M3(am[3], ax[3], 3, 2) DZ_M3(am[3], ax[3], 3, 3)
            const f32x4 e2 = exps(am[2], ax[2]);
            DZ_SB();
            // M(0,0) || exps of cell 3, cell 2
            DZ_M3(n0m, n0x, 0, 0) DZ_M3(n0m, n0x, 0, 1)
            const f32x4 e3 = exps(am[3], ax[3]);
            const float h2 = finish(e2, cs[2]);
            DZ_SB();
            fetch_piece(s + 3, (S4 + 3) & 3, 1);
            // M(1,0) || cell 3; the pair -> LDS (k-half 1 of h_{s-1})
            DZ_M3(n1m, n1x, 1, 0) DZ_M3(n1m, n1x, 1, 1)
            const float h3 = finish(e3, cs[3]);
            put_pair(h2, h3, const_cast<char*>(hprev) + wr_off1);
            if constexpr (F32OUT) p23_f = (f32x2){h2, h3};
        } else {
            DZ_M3(n0m, n0x, 0, 0) DZ_M3(n0m, n0x, 0, 1)
            fetch_piece(s + 3, (S4 + 3) & 3, 1);
            DZ_M3(n1m, n1x, 1, 0) DZ_M3(n1m, n1x, 1, 1)
        }
        DZ_SB();
    };
    // barrier Q_s (k-half 1 of h_{s-1}); M(2,0) covers its reads; the second halves of tiles 0, 1; cells 0 and 1 of
    // step s under M(1,1), M(3,0), M(2,1); h_{s-1} -> HBM; barrier P_{s+1}
    auto second_half = [&](auto phase_tag, auto s4_tag, int s) {
        constexpr int PHASE = decltype(phase_tag)::value;
        constexpr int S4 = decltype(s4_tag)::value;
        const char* hprev = hs + ((S4 + 1) & 1) * 2 * PLANE;
        char* hcur = hs + (S4 & 1) * 2 * PLANE;
        DZ_BARRIER("lgkmcnt(0)");
        bh[2] = *reinterpret_cast<const f16x8*>(hprev + rd_off[2]);
        bl[2] = *reinterpret_cast<const f16x8*>(hprev + PLANE + rd_off[2]);
        bh[3] = *reinterpret_cast<const f16x8*>(hprev + rd_off[3]);
        bl[3] = *reinterpret_cast<const f16x8*>(hprev + PLANE + rd_off[3]);
        u32x2 ea = {0, 0}, eb = {0, 0};
        if constexpr (PHASE >= 1) {
            ea = *reinterpret_cast<const u32x2*>(hprev + e_rda);
            eb = *reinterpret_cast<const u32x2*>(hprev + e_rdb);
        }
        fetch_piece(s + 3, (S4 + 3) & 3, 2);
        DZ_SB();
        DZ_M3(n2m, n2x, 2, 0) DZ_M3(n2m, n2x, 2, 1)
        DZ_SB();
        DZ_M3(n0m, n0x, 0, 2) DZ_M3(n0m, n0x, 0, 3)
        DZ_SB();
        if constexpr (PHASE >= 1 && !(MODE & 4))
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){ea[0], eb[0], ea[1], eb[1]}, r_pl, e_vo, frame(s - 1) * 64, 0);
        fetch_piece(s + 3, (S4 + 3) & 3, 3);
        // M(1,1) || exps of cell 0
        DZ_M3(n1m, n1x, 1, 2) DZ_M3(n1m, n1x, 1, 3)
        const f32x4 e0 = exps(n0m, n0x);
        DZ_SB();
        // M(3,0) || exps of cell 1, cell 0
        DZ_M3(n3m, n3x, 3, 0) DZ_M3(n3m, n3x, 3, 1)
        const f32x4 e1 = exps(n1m, n1x);
        const float h0 = finish(e0, cs[0]);
        DZ_SB();
        // M(2,1) || cell 1; the pair -> LDS (k-half 0 of h_s)
        DZ_M3(n2m, n2x, 2, 2) DZ_M3(n2m, n2x, 2, 3)
        const float h1 = finish(e1, cs[1]);
        put_pair(h0, h1, hcur + wr_off0);
        if constexpr (F32OUT) p01_f = (f32x2){h0, h1};
        DZ_SB();
        am[2] = n2m; ax[2] = n2x; am[3] = n3m; ax[3] = n3x;
        // P_{s+1}: the pieces of step s+1 have landed (loads retire in order: with at most 8 operations in flight
        // every piece older than those of steps s+2, s+3 is in LDS), this wave's halves of h are in LDS
        DZ_BARRIER("vmcnt(8) lgkmcnt(0)");
    };

#pragma unroll
    for (int i = 0; i < 4; ++i) fetch_piece(0, 0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) fetch_piece(1, 1, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) fetch_piece(2, 2, i);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // step 0's four pieces
    __syncthreads();
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    first_half(I0{}, I0{}, 0);
    second_half(I0{}, I0{}, 0);
    if (T > 1) {
        first_half(I1{}, I1{}, 1);
        second_half(I1{}, I1{}, 1);
    }
    for (int s = 2; s < T; s += 4) {         // four steps per trip: s & 3 = 2, 3, 0, 1
        first_half(I2{}, I2{}, s);     second_half(I2{}, I2{}, s);
        if (s + 1 >= T) break;
        first_half(I2{}, I3{}, s + 1); second_half(I2{}, I3{}, s + 1);
        if (s + 2 >= T) break;
        first_half(I2{}, I0{}, s + 2); second_half(I2{}, I0{}, s + 2);
        if (s + 3 >= T) break;
        first_half(I2{}, I1{}, s + 3); second_half(I2{}, I1{}, s + 3);
    }
    {   // step T-1: tile 3, cells 2 and 3 -> LDS; h_{T-1} -> HBM
        char* hlast = hs + ((T - 1) & 1) * 2 * PLANE;
        if constexpr (F32OUT && !(MODE & 4)) {
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, p01_f), r_f, vo_f, frame(T - 1) * 1024, 0);
            if (T > 1) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, p23_f), r_f, vo_f + 8, frame(T - 2) * 1024, 0);
        }
        DZ_M3(am[3], ax[3], 3, 2) DZ_M3(am[3], ax[3], 3, 3)
        const f32x4 e2 = exps(am[2], ax[2]), e3 = exps(am[3], ax[3]);
        const float h2 = finish(e2, cs[2]), h3 = finish(e3, cs[3]);
        put_pair(h2, h3, hlast + wr_off1);
        if constexpr (F32OUT && !(MODE & 4))
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, (f32x2){h2, h3}), r_f, vo_f + 8, frame(T - 1) * 1024, 0);
        __syncthreads();
        const u32x2 ea = *reinterpret_cast<const u32x2*>(hlast + e_rda);
        const u32x2 eb = *reinterpret_cast<const u32x2*>(hlast + e_rdb);
        if constexpr (!(MODE & 4))
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){ea[0], eb[0], ea[1], eb[1]}, r_pl, e_vo, frame(T - 1) * 64, 0);
    }
#undef DZ_M3
#undef DZ_SB
#undef DZ_BARRIER
}

// variant 0: two accumulators, lo planes scaled by 2^11 (whh_split = split_f16 of W_hh);
// variant 1 / 2: one accumulator, activation scales folded into the planes, H scaled by 2^0 / 2^8
// (whh_split from weights.py lstm_whh_planes(whh, variant)); variant 3: variant 0's arithmetic and
// planes, gx by LDS-DMA (unit-major only); variant 4: the software-pipelined form above (unit-major,
// pre-scaled gx; planes with permuted columns and the activation scales folded in)
int dz_launch_lstm_mfma(const float* gx, const void* whh_split, float* hout, void* hsplit,
                        long long hplane, int B, int T, int unit_major, int variant, hipStream_t st) {
    dim3 grid((B + CH - 1) / CH, 2);
    const unsigned short* whs = reinterpret_cast<const unsigned short*>(whh_split);
    unsigned short* hsp = reinterpret_cast<unsigned short*>(hsplit);
    DZ_REQUIRE(variant >= 0 && variant <= 300, "lstm_mfma: variant %d", variant);
    DZ_REQUIRE(hout || hsp, "lstm_mfma: no output");
    DZ_REQUIRE(variant < 3 || (unit_major && (long long)B * T * 4096 < (1ll << 31)),
               "lstm_mfma: variants 3 / 4 (LDS-DMA of gx) need unit-major gx below 2 GiB");
    DZ_REQUIRE(variant < 4 || hplane * 4 < (1ll << 31), "lstm_mfma: variant 4 addresses planes below 2 GiB");
    DZ_REQUIRE(hplane % 256 == 0 && (!hsplit || hplane >= (long long)B * T * 256),
               "lstm_mfma: the kb-major planes need hplane = rows * 256 with rows >= B * T");
#define DZ_L(K) DZ_LAUNCH(K, grid, dim3(512), 0, st, gx, whs, hout, hsp, hplane, B, T)
    if (variant == 0) { if (unit_major) DZ_L(lstm_mfma_kernel<true>); else DZ_L(lstm_mfma_kernel<false>); }
#ifdef DZ_EXPERIMENTS
    if (variant == 1) { if (unit_major) DZ_L((lstm_mfma1_kernel<true, 0>)); else DZ_L((lstm_mfma1_kernel<false, 0>)); }
    if (variant == 2) { if (unit_major) DZ_L((lstm_mfma1_kernel<true, 8>)); else DZ_L((lstm_mfma1_kernel<false, 8>)); }
#else
    DZ_REQUIRE(variant == 0 || variant >= 3, "lstm_mfma: variants 1 / 2 exist in the experiments build only (variant %d)", variant);
#endif
    if (variant == 3) DZ_L(lstm_mfma_dma_kernel);
    if (variant == 4) { if (hout) DZ_L((lstm_mfma_pipe_kernel<true, 0>)); else DZ_L((lstm_mfma_pipe_kernel<false, 0>)); }
#ifdef DZ_EXPERIMENTS
    if (variant == 4 + 4) DZ_L((lstm_mfma_pipe_kernel<false, 4>));
    if (variant == 4 + 16) DZ_L((lstm_mfma_pipe_kernel<false, 16>));
    if (variant == 4 + 20) DZ_L((lstm_mfma_pipe_kernel<false, 20>));
    if (variant == 4 + 52) DZ_L((lstm_mfma_pipe_kernel<false, 52>));
    if (variant == 4 + 32) DZ_L((lstm_mfma_pipe_kernel<false, 32>));
    if (variant == 4 + 128) DZ_L((lstm_mfma_pipe_kernel<false, 128>));
    if (variant == 4 + 180) DZ_L((lstm_mfma_pipe_kernel<false, 180>));
#endif
#undef DZ_L
    DZ_HIP(hipGetLastError());
    return 0;
}
