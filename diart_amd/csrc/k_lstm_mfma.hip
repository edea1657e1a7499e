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

// variant 0: two accumulators, lo planes scaled by 2^11 (whh_split = split_f16 of W_hh);
// variant 1 / 2: one accumulator, activation scales folded into the planes, H scaled by 2^0 / 2^8
// (whh_split from weights.py lstm_whh_planes(whh, variant)); variant 3: variant 0's arithmetic and
// planes, gx by LDS-DMA (unit-major only)
int dz_launch_lstm_mfma(const float* gx, const void* whh_split, float* hout, void* hsplit,
                        long long hplane, int B, int T, int unit_major, int variant, hipStream_t st) {
    dim3 grid((B + CH - 1) / CH, 2);
    const unsigned short* whs = reinterpret_cast<const unsigned short*>(whh_split);
    unsigned short* hsp = reinterpret_cast<unsigned short*>(hsplit);
    DZ_REQUIRE(variant >= 0 && variant <= 3, "lstm_mfma: variant %d", variant);
    DZ_REQUIRE(hout || hsp, "lstm_mfma: no output");
    DZ_REQUIRE(variant != 3 || (unit_major && (long long)B * T * 4096 < (1ll << 31)),
               "lstm_mfma: variant 3 (LDS-DMA of gx) needs unit-major gx below 2 GiB");
    DZ_REQUIRE(hplane % 256 == 0 && (!hsplit || hplane >= (long long)B * T * 256),
               "lstm_mfma: the kb-major planes need hplane = rows * 256 with rows >= B * T");
#define DZ_L(K) DZ_LAUNCH(K, grid, dim3(512), 0, st, gx, whs, hout, hsp, hplane, B, T)
    if (variant == 0) { if (unit_major) DZ_L(lstm_mfma_kernel<true>); else DZ_L(lstm_mfma_kernel<false>); }
#ifdef DZ_EXPERIMENTS
    if (variant == 1) { if (unit_major) DZ_L((lstm_mfma1_kernel<true, 0>)); else DZ_L((lstm_mfma1_kernel<false, 0>)); }
    if (variant == 2) { if (unit_major) DZ_L((lstm_mfma1_kernel<true, 8>)); else DZ_L((lstm_mfma1_kernel<false, 8>)); }
#else
    DZ_REQUIRE(variant == 0 || variant == 3, "lstm_mfma: variants 1 / 2 exist in the experiments build only (variant %d)", variant);
#endif
    if (variant == 3) DZ_L(lstm_mfma_dma_kernel);
#undef DZ_L
    DZ_HIP(hipGetLastError());
    return 0;
}
