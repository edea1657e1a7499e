// SincNet front-end kernels (shared by the segmentation and embedding networks):
//   wave_stats     per-chunk mean / rstd of the raw 5 s window   (InstanceNorm1d(1))
//   sinc_conv0     normalise-on-load + 80x251 sinc FIR bank (stride 10, symmetric fold) on fp32 MFMA
//                  + abs + MaxPool1d(3) + per-tile (sum, sumsq) partials
//   finalize_norm  partials -> per (chunk, channel) scale / shift of InstanceNorm1d(C, affine)
// Restates pyannote.audio SincNet.forward (third party; called from
// /root/reference/src/diart/models.py:133 and :262; graph in SURVEY.md Appendix A.1).
#include "dz_common.h"

// ---------------------------------------------------------------------------
// wave_stats: WS_G workgroups per chunk, each over one slice of the window.  A slice is read from
// HBM ONCE (16 B coalesced loads, kept in registers) and reduced in two passes (slice mean, then
// centred second moment: no cancellation) to (mean_i, M2_i).  The consumer (sinc_conv0's
// prologue, or wave_stats_combine for the kernel-level entry point) merges the WS_G slice moments
// with Chan's formula in f64, in fixed slice order — the kernel boundary is the only
// synchronisation.  (One workgroup per chunk, the first version, streamed 320 KB through 256
// threads twice: 22 us alone, 65-90 us beside other kernels, at the head of both networks'
// critical paths.  A "last workgroup combines" variant needs a device-scope release per
// workgroup, i.e. an L2 write-back across the 8 XCDs: measured slower.)
// moments layout: [B][WS_G][2] floats.
// ---------------------------------------------------------------------------
#define WS_NV 10   /* float4 kept per thread: slices up to 256 * 10 * 4 = 10240 samples stay in registers */

__device__ __forceinline__ float dz_block_sum_f(float v, float* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__host__ __device__ __forceinline__ int dz_ws_slice(int S) { return ((S + DZ_WS_G - 1) / DZ_WS_G + 3) & ~3; }

// (mean, rstd) of chunk b from its slice moments — fixed order, f64
__device__ __forceinline__ void dz_ws_combine(const float* __restrict__ mom, int b, int S, float* mean,
                                              float* rstd) {
    const int L = dz_ws_slice(S);
    const float* m = mom + (long long)b * 2 * DZ_WS_G;
    double tot = 0.0, M2 = 0.0, mu[DZ_WS_G], cnt[DZ_WS_G];
#pragma unroll
    for (int i = 0; i < DZ_WS_G; ++i) {
        const int n = min(S, (i + 1) * L) - i * L;
        cnt[i] = (double)(n > 0 ? n : 0);
        mu[i] = (double)m[2 * i];
        M2 += (double)m[2 * i + 1];
        tot += cnt[i] * mu[i];
    }
    const double mm = tot / (double)S;
#pragma unroll
    for (int i = 0; i < DZ_WS_G; ++i) M2 += cnt[i] * (mu[i] - mm) * (mu[i] - mm);
    *mean = (float)mm;
    *rstd = (float)(1.0 / sqrt(M2 / (double)S + 1e-5));   // biased variance, like InstanceNorm
}

__global__ __launch_bounds__(256) void wave_stats_kernel(const float* __restrict__ wave,
                                                         long long stride, int S,
                                                         float* __restrict__ mom) {
    __shared__ float red[4];
    const int b = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    const int L = dz_ws_slice(S);
    const int s0 = g * L, s1 = min(S, s0 + L), n = max(0, s1 - s0);
    const float* x = wave + (long long)b * stride + s0;
    const int n4 = n >> 2;
    float4 v[WS_NV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < WS_NV; ++j) {
        const int i = tid + 256 * j;
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n4) {
            v[j] = reinterpret_cast<const float4*>(x)[i];
            s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
    }
    for (int i = tid + 256 * WS_NV; i < n4; i += 256) {                  // longer windows: not cached
        const float4 u = reinterpret_cast<const float4*>(x)[i];
        s += (u.x + u.y) + (u.z + u.w);
    }
    for (int i = (n4 << 2) + tid; i < n; i += 256) s += x[i];          // tail (< 4 samples)
    const float mean = n > 0 ? dz_block_sum_f(s, red) / (float)n : 0.f;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < WS_NV; ++j)
        if (tid + 256 * j < n4) {
            const float a = v[j].x - mean, c = v[j].y - mean, d = v[j].z - mean, e = v[j].w - mean;
            ss += (a * a + c * c) + (d * d + e * e);
        }
    for (int i = tid + 256 * WS_NV; i < n4; i += 256) {                  // re-read what was not cached
        const float4 u = reinterpret_cast<const float4*>(x)[i];
        const float a = u.x - mean, c = u.y - mean, d = u.z - mean, e = u.w - mean;
        ss += (a * a + c * c) + (d * d + e * e);
    }
    for (int i = (n4 << 2) + tid; i < n; i += 256) {
        const float a = x[i] - mean;
        ss += a * a;
    }
    const float m2 = dz_block_sum_f(ss, red);
    if (tid == 0) {
        float* out = mom + ((long long)b * DZ_WS_G + g) * 2;
        out[0] = mean;
        out[1] = m2;
    }
}

__global__ void wave_stats_combine_kernel(const float* __restrict__ mom, int B, int S,
                                          float* __restrict__ stats) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) dz_ws_combine(mom, b, S, stats + 2 * b, stats + 2 * b + 1);
}

// mom: [B][DZ_WS_G][2] floats
int dz_launch_wave_stats(const float* wave, long long stride, int B, int S, float* mom,
                         hipStream_t st) {
    DZ_LAUNCH(wave_stats_kernel, dim3(DZ_WS_G, B), dim3(256), 0, st, wave, stride, S, mom);
    DZ_HIP(hipGetLastError());
    return 0;
}
// (mean, rstd) per chunk from the slice moments: the kernel-level entry point's contract
int dz_launch_wave_stats_combine(const float* mom, int B, int S, float* stats, hipStream_t st) {
    DZ_LAUNCH(wave_stats_combine_kernel, dim3((B + 63) / 64), dim3(64), 0, st, mom, B, S, stats);
    DZ_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// sinc_conv0.  As a GEMM per chunk: out[t][c] = sum_k xn[10 t + k] * filt[c][k],
// M = 7975 frames, N = 80 filters, K = 251.  ParamSincFB's filters are symmetric by
// construction (cos filters right = flip(left), sin filters right = -flip(left)),
// so with the centre tap at 10 t + 125 and j the distance from it
//     cos:  out = sum_{j=0..125} hc[j] * (x[c + j] + x[c - j])      (hc[0] = centre / 2)
//     sin:  out = sum_{j=1..125} hs[j] * (x[c + j] - x[c - j])
// i.e. two GEMMs with K = 126 instead of one with K = 251: half the MFMA work, the price being
// one add + one sub per A element (VALU, hidden under the 32-cycle f32 MFMAs).
// One workgroup = 192 conv frames (= 64 pooled) x (48 cos | 48 sin) padded filter columns;
// 4 waves x (3 M-frags x 6 N-frags) of 16x16x4 f32 MFMA, K padded to 128 (taps 126, 127 = 0).
// LDS: folded bank [128][96] k-major, column XOR 16*(k&1) (B reads: bank = i + 16*((nt^k)&1),
// conflict-free per half-wave) + the 2165-sample slice the tile touches (A reads at stride 10
// dwords: 10 i +- q hits 32 distinct banks per half-wave).  58 KiB -> two workgroups per CU, so
// one's prologue / pooling epilogue runs under the other's MFMAs.  Every sample is read from
// HBM once per tile and reused ~25x from LDS.
// ---------------------------------------------------------------------------
#define C0_FR 192
#define C0_LP 2                       /* samples kept left of the tile start (taps 126/127)  */
#define C0_XS (C0_FR * 10 + 256)      /* >= 10*191 + 127 + 127 + 1                            */
#define C0_KS 32                      /* k-steps of 4                                         */
#define C0_NP 96                      /* padded filter columns: 48 cos | 48 sin              */
#define C0_OLD 81

__global__ __launch_bounds__(256) void sinc_conv0_kernel(
    const float* __restrict__ wave, long long stride, int S, const float* __restrict__ stats,
    int stats_are_moments, float gamma, float beta, const float* __restrict__ filt,
    float* __restrict__ y0, int P0, float* __restrict__ partials, int ntile) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* filt_s = smem;                    // [128][96], swizzled
    float* xs = smem + 4 * C0_KS * C0_NP;    // [C0_XS]
    float* out_s = smem;                     // [192][81], aliases both after the K loop
    const int b = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;

    for (int i4 = tid; i4 < 4 * C0_KS * C0_NP / 4; i4 += 256) {
        const int k = i4 / (C0_NP / 4), c4 = i4 - k * (C0_NP / 4);
        reinterpret_cast<float4*>(filt_s)[k * (C0_NP / 4) + (c4 ^ ((k & 1) << 2))] =
            reinterpret_cast<const float4*>(filt)[i4];
    }
    {
        float mean, rstd;
        if (stats_are_moments)   // wave_stats' slice moments: merged here (fixed order, f64)
            dz_ws_combine(stats, b, S, &mean, &rstd);
        else
            mean = stats[2 * b], rstd = stats[2 * b + 1];
        const float* wb = wave + (long long)b * stride;
        const int s0 = tile * (C0_FR * 10) - C0_LP;
        for (int i = tid; i < C0_XS; i += 256) {
            const int s = s0 + i;
            xs[i] = (s >= 0 && s < S) ? ((wb[s] - mean) * rstd) * gamma + beta : 0.f;
        }
    }
    __syncthreads();

    const int w = tid >> 6, l = tid & 63, i = l & 15, q = l >> 4;
    f32x4 acc[3][6];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // centre tap of frame t sits at xs[10 t + 125 + C0_LP]; this lane's k is 4 ks + q
    const float* xp = xs + 10 * (48 * w + i) + 125 + C0_LP + q;   // x[c + j]
    const float* xm = xs + 10 * (48 * w + i) + 125 + C0_LP - q;   // x[c - j]
    // column block nt of row k lives at block nt ^ (k & 1); k & 1 == q & 1 for this lane
    const float* fb = filt_s + q * C0_NP + i;
    int fo[6];
#pragma unroll
    for (int nt = 0; nt < 6; ++nt) fo[nt] = 16 * (nt ^ (q & 1));
#pragma unroll 8
    for (int ks = 0; ks < C0_KS; ++ks) {
        float ac[3], as[3], bb[6];
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
            const float hi = xp[4 * ks + 160 * mt], lo = xm[160 * mt - 4 * ks];
            ac[mt] = hi + lo;
            as[mt] = hi - lo;
        }
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) bb[nt] = fb[4 * C0_NP * ks + fo[nt]];
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
            for (int nt = 0; nt < 6; ++nt)
                acc[mt][nt] = DZ_MFMA(nt < 3 ? ac[mt] : as[mt], bb[nt], acc[mt][nt]);
    }
    __syncthreads();  // filt_s / xs are dead from here: reuse as the output tile

#pragma unroll
    for (int nt = 0; nt < 6; ++nt) {
        const int cl = 16 * (nt % 3) + i;            // column inside the cos / sin half
        if (cl < 40) {
            const int ch = (nt < 3 ? 0 : 40) + cl;   // output channel: 40 cos then 40 sin
#pragma unroll
            for (int mt = 0; mt < 3; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    out_s[(48 * w + 16 * mt + 4 * q + r) * C0_OLD + ch] = fabsf(acc[mt][nt][r]);
        }
    }
    __syncthreads();

    // MaxPool1d(3) over time + the tile's InstanceNorm partials in ONE pass: thread -> fixed channel
    // n = tid % 80 and row group tid / 80 (3 groups; threads 240..255 idle), so every thread keeps
    // its channel's running (sum, sumsq) in registers; the three group partials meet in LDS in fixed
    // order.  (A separate pass in which 80 threads walked 64 pooled rows each cost ~6k cycles per
    // tile with the matrix pipe idle.)
    const int p0 = tile * 64;
    float s = 0.f, ss = 0.f;
    const int n = tid % 80, rg = tid / 80;
    if (rg < 3) {
        for (int pr = rg; pr < 64; pr += 3) {
            const float* o = out_s + (3 * pr) * C0_OLD + n;
            const float v = fmaxf(fmaxf(o[0], o[C0_OLD]), o[2 * C0_OLD]);
            if ((p0 + pr) < P0) {
                y0[((long long)b * P0 + p0 + pr) * 80 + n] = v;
                s += v;
                ss += v * v;
            }
        }
    }
    __syncthreads();                       // every read of out_s is done: reuse its head
    if (rg < 3) {
        out_s[(rg * 80 + n) * 2] = s;
        out_s[(rg * 80 + n) * 2 + 1] = ss;
    }
    __syncthreads();
    if (tid < 80) {
        float* pp = partials + (((long long)b * ntile + tile) * 80 + tid) * 2;
        pp[0] = (out_s[tid * 2] + out_s[(80 + tid) * 2]) + out_s[(160 + tid) * 2];
        pp[1] = (out_s[tid * 2 + 1] + out_s[(80 + tid) * 2 + 1]) + out_s[(160 + tid) * 2 + 1];
    }
}

int dz_launch_sinc_conv0(const float* wave, long long stride, int B, int S, const float* stats,
                         int stats_are_moments, float gamma, float beta, const float* filt,
                         float* y0, int P0, float* partials, int ntile, hipStream_t st) {
    const size_t k_loop = (4 * C0_KS * C0_NP + C0_XS) * sizeof(float);
    const size_t epi = (size_t)C0_FR * C0_OLD * sizeof(float);
    const size_t lds = k_loop > epi ? k_loop : epi;
    static DzAttrOnce attr_once;
    DZ_HIP(attr_once.raise((const void*)sinc_conv0_kernel, (int)lds));
    DZ_LAUNCH(sinc_conv0_kernel, dim3(ntile, B), dim3(256), lds, st, wave, stride, S,
                       stats, stats_are_moments, gamma, beta, filt, y0, P0, partials, ntile);
    DZ_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// sinc_conv0 on the f16 matrix cores (default precision).  The same contraction, unfolded:
//     out[t][c] = sum_{k < 251} xn[10 t + k] * filt[c][k]          M = frames, N = 80 -> 96, K -> 256
// with both operands split into (hi, lo * 2^11) f16 pairs, three v_mfma_f32_32x32x16_f16 per
// product into two f32 accumulators — the arithmetic of k_gemm_split.hip.  The exact-f32 kernel
// above needs half the multiply-adds (symmetric fold) but runs them on the f32 matrix pipe,
// 16x slower per multiply-add than the f16 pipe; and the fold costs an add + a sub per A element
// that an f16 version would have to split again.  Here the A operand needs NO arithmetic at all:
//   * A is a Toeplitz view of the tile's samples.  Row t, k-chunk (8 consecutive k) = 8
//     consecutive samples starting at 10 t + k0: a 16-byte LDS read if the f16 sample array is
//     16-byte aligned there.  10 t is only even, so the tile keeps FOUR copies of its samples,
//     copy c shifted by 2c samples (copy_c[i] = x[i + 2c]); frame t reads copy t & 3, where
//     10 t - 2 (t & 3) is a multiple of 8.  4 copies x 2 planes x 1216 samples = 20 KiB.
//   * B (the filter bank, 96 x 256 x 2 planes) lives in REGISTERS for the whole kernel: wave w of a
//     workgroup owns filters 32w .. 32w+31 = 16 k-steps x (hi, lo) fragments = 128 VGPRs.
//   * MaxPool1d(3) needs no exchange either: the workgroup's 96 frames are processed as three
//     32-row MFMA blocks, block b = frames {3 m + b}; pooled row m is then the element-wise
//     maximum of the three blocks' |accumulators| in the SAME lane and register.
// Workgroup = 3 waves (one per 32-filter column block), tile = 96 frames = 32 pooled rows, two
// workgroups per CU; per k-step a wave reads 2 fragments (2 KiB) for 3 MFMAs.  MFMA row i of a
// block carries pooled row pi(i) = bits (1,0) and (3,2) of i swapped, and copy c starts 16 * E[c]
// bytes into its slot: with that the 16 lanes of every ds_read_b128 group hit at most 2-way bank
// conflicts (brute-forced; the straight order is 4-way).
// ---------------------------------------------------------------------------
#define CH_FR 96                       /* frames per tile                                        */
#define CH_NS 1216                     /* f16 samples per copy plane: >= 10*95 + 255 + 1         */
#define CH_PL 2560                     /* bytes per copy plane, padded to a multiple of 256      */
#define CH_CP (2 * CH_PL + 256)        /* bytes per copy (hi | lo) + room for the start offset   */
#define CH_LDS (4 * CH_CP)

typedef _Float16 ch_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ch_f16x2 __attribute__((ext_vector_type(2)));
typedef float ch_f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int ch_copy_base(int c) {
    // 16-byte start offsets (0, 12, 8, 5) of the four copies inside their slots
    return c * CH_CP + 16 * ((0x58c0 >> (4 * c)) & 15);
}
__device__ __forceinline__ int ch_pi(int i) { return ((i & 3) << 2) | ((i >> 2) & 3) | (i & 16); }

// sinc_conv0_h: the three-wave kernel of rounds 2 - 4, since round 5 built into the EXPERIMENTS library only
// (DZ_CONV0_V2=0 selects it there: the A/B arm of profiles/r05j_conv0_v2_ab.json and the subject of the phase profile
// that led to sinc_conv0_v2 below).  What round 5 measured on it (profiles/r05e_pmc_conv0.json, isolated, 64
// chunks): matrix pipes busy 28 % of the launch, LDS 18 % (a third of that bank conflicts), waves in s_waitcnt / the
// tile barrier 59 % of their cycles.  Three single-cause rebuilds each measured no gain — four waves in two roles
// without read pipelining (100 - 105 us against 99.6), one launch for both networks (sinc_conv0_pair below), and
// this kernel with its fragment reads issued one step ahead (99.4 against 99.4 us) — because the tile's time has
// three comparable parts (profiles/r05j_conv0_phases_old_kernel.json) and each rebuild removed one while the tile
// barrier kept the others; sinc_conv0_v2 removes them together.
// Persistent: the grid is at most two workgroups per CU, each walks a contiguous range of (chunk,
// tile) pairs with its B fragments resident; the samples of tile t+1 are fetched into registers
// before the MFMA loop of tile t and parked in the other LDS buffer after it (one barrier per tile).
// DBG (timing experiments, DZ_CONV0_DBG; results are wrong): 1 = no parking of the next tile's samples, 2 = no
// MFMAs, 4 = no fragment reads inside the tile loop, 8 = no result stores / partials
#ifdef DZ_EXPERIMENTS
template <int DBG = 0>
__global__ __launch_bounds__(192, 2) void sinc_conv0_h_kernel(
    const float* __restrict__ wave, long long stride, int S, const float* __restrict__ stats,
    int stats_are_moments, float gamma, float beta, const unsigned short* __restrict__ fsp,
    float* __restrict__ y0, int P0, float* __restrict__ partials, int ntile, int total,
    int* __restrict__ oflag, long long* __restrict__ dbg) {
    // dbg (experiments build, tools/conv0_phases.py): per wave 5 shader-clock stamps per tile + HW_ID in slot 63
    long long* dq = nullptr;
    int dn = 0;
#ifdef DZ_EXPERIMENTS
    if (dbg && (threadIdx.x & 63) == 0) {
        dq = dbg + ((long long)blockIdx.x * 3 + (threadIdx.x >> 6)) * 64;
        dq[63] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID: wave / SIMD / CU / SE ids
    }
#define C0_STAMP() do { if (dq && dn < 60) dq[dn++] = __builtin_readcyclecounter(); } while (0)
#else
#define C0_STAMP() do { } while (0)
#endif
    // one LDS object (a second one makes hipcc drain vmcnt before every ds_read while an LDS-DMA is
    // in flight): [2][CH_LDS] sample copies | raw[1536] f32 landing zone | (mean, rstd) of 2 chunks
    __shared__ __attribute__((aligned(256))) char lds_all[2 * CH_LDS + 1536 * 4 + 16];
    char (*xs2)[CH_LDS] = reinterpret_cast<char (*)[CH_LDS]>(lds_all);
    float* raw = reinterpret_cast<float*>(lds_all + 2 * CH_LDS);
    float (*stat_s)[2] = reinterpret_cast<float (*)[2]>(lds_all + 2 * CH_LDS + 1536 * 4);
    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63, li = l & 31, g = l >> 5;
    const int wg = dz_xcd_contiguous(blockIdx.x, gridDim.x);         // neighbouring tile ranges on ONE XCD
    const int t_begin = (int)((long long)wg * total / gridDim.x);
    const int t_end = (int)((long long)(wg + 1) * total / gridDim.x);
    if (t_begin >= t_end) return;
    const int b_first = t_begin / ntile;

    // (mean, rstd) of the (at most two) chunks this range touches
    if (tid < 2) {
        const int bb = b_first + tid;
        float mean = 0.f, rstd = 1.f;
        if (bb * ntile < t_end) {
            if (stats_are_moments)
                dz_ws_combine(stats, bb, S, &mean, &rstd);
            else
                mean = stats[2 * bb], rstd = stats[2 * bb + 1];
        }
        stat_s[tid][0] = mean;
        stat_s[tid][1] = rstd;
    }
    // ---- B fragments: filters 32w + li, k = 16 ks + 8 g .. +7, hi and lo planes ------------------
    ch_f16x8 bh[16], bl[16];
    {
        const unsigned short* row = fsp + (long long)(32 * w + li) * 256 + 8 * g;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            bh[ks] = *reinterpret_cast<const ch_f16x8*>(row + 16 * ks);
            bl[ks] = *reinterpret_cast<const ch_f16x8*>(row + 96 * 256 + 16 * ks);
        }
    }
    __syncthreads();

    // Raw samples of the NEXT tile travel global -> LDS by LDS-DMA (no registers: the resident
    // filter fragments leave none): wave w fetches floats [512 w, 512 w + 512) of the tile's 1536-float
    // window as two 1 KiB pieces and later parks exactly those (wave-local dependency: its own vmcnt).
    // Samples past the end of the chunk read as zeros through the buffer bounds check.
    auto fetch = [&](int t) {
        const int bb = t / ntile, tile = t - bb * ntile;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(wave + (long long)bb * stride), 0, (unsigned)S * 4u, 0x00020000);
        const int voff = (tile * (CH_FR * 10) + 512 * w + 4 * l) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsrc, (__attribute__((address_space(3))) void*)(raw + 512 * w + 256 * i), 16, voff + 1024 * i,
                0, 0, 0);
    };
    float amax = 0.f;
    // normalise (InstanceNorm1d(1)), split, write the four shifted copies
    auto park = [&](int t, char* xs) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's two pieces have landed
        const int bb = t / ntile, tile = t - bb * ntile;
        const float mean = stat_s[bb - b_first][0], rstd = stat_s[bb - b_first][1];
        const int s0 = tile * (CH_FR * 10);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = 512 * w + 2 * (l + 64 * q), sidx = s0 + j;
            if (j < CH_NS + 6) {
                const float2 pvq = *reinterpret_cast<const float2*>(raw + j);
                f32x2 x = {sidx < S ? ((pvq.x - mean) * rstd) * gamma + beta : 0.f,
                           sidx + 1 < S ? ((pvq.y - mean) * rstd) * gamma + beta : 0.f};
                amax = fmaxf(amax, fmaxf(fabsf(x[0]), fabsf(x[1])));
                x[0] = __builtin_amdgcn_fmed3f(x[0], -65504.f, 65504.f);
                x[1] = __builtin_amdgcn_fmed3f(x[1], -65504.f, 65504.f);
                const ch_f16x2 hi = __builtin_convertvector(x, ch_f16x2);
                const ch_f16x2 lo =
                    __builtin_convertvector((x - __builtin_convertvector(hi, f32x2)) * 2048.f, ch_f16x2);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int idx = j - 2 * c;            // copy_c[idx] = x[idx + 2c]
                    if (idx >= 0 && idx < CH_NS) {
                        char* d = xs + ch_copy_base(c) + 2 * idx;
                        *reinterpret_cast<ch_f16x2*>(d) = hi;
                        *reinterpret_cast<ch_f16x2*>(d + CH_PL) = lo;
                    }
                }
            }
        }
    };

    // MFMA row li <-> pooled row m = pi(li); block bk = frames 3 m + bk
    const int m_a = ch_pi(li);
    int aoff[3];
#pragma unroll
    for (int bk = 0; bk < 3; ++bk) {
        const int f = 3 * m_a + bk, c = f & 3;
        aoff[bk] = ch_copy_base(c) + 2 * (10 * f - 2 * c + 8 * g);
    }
    const int ch = 32 * w + li;

    fetch(t_begin);
    park(t_begin, xs2[0]);
    __syncthreads();
    for (int t = t_begin; t < t_end; ++t) {
        const char* xs = xs2[(t - t_begin) & 1];
        C0_STAMP();
        if (t + 1 < t_end) fetch(t + 1);
        ch_f32x16 pmax;
#pragma unroll
        for (int bk = 0; bk < 3; ++bk) {
            const char* ap = xs + aoff[bk];
            ch_f32x16 accm, accx;
#pragma unroll
            for (int r = 0; r < 16; ++r) accm[r] = accx[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const ch_f16x8 ah = *reinterpret_cast<const ch_f16x8*>((DBG & 4) ? xs + 16 * l : ap + 32 * ks);
                const ch_f16x8 al = *reinterpret_cast<const ch_f16x8*>((DBG & 4) ? xs + 16 * l + 1024 : ap + CH_PL + 32 * ks);
                if (DBG & 2) {
                    asm volatile("" ::"v"(ah), "v"(al), "v"(bh[ks]), "v"(bl[ks]));
                } else {
                    accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ks], accx, 0, 0, 0);
                    accm = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ks], accm, 0, 0, 0);
                    accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ks], accx, 0, 0, 0);
                }
                // keep the fragment reads at most four k-steps ahead of their MFMAs: hoisting all 32
                // reads of a block (128 registers) spills the resident filter fragments
                if ((ks & 3) == 3) asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = fabsf(accm[r] + accx[r] * (1.f / 2048.f));
                pmax[r] = bk == 0 ? v : fmaxf(pmax[r], v);
            }
        }
        C0_STAMP();
        // the next tile's samples first: parking waits for this wave's LDS-DMA (vmcnt), which must
        // not also wait for the result stores below
        if (t + 1 < t_end && !(DBG & 1)) park(t + 1, xs2[(t + 1 - t_begin) & 1]);
        if (DBG & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        C0_STAMP();
        // pooled rows + InstanceNorm partials: C/D column = lane & 31 = filter, row rho <-> m = pi(rho)
        const int bb = t / ntile, tile = t - bb * ntile;
        float sum = 0.f, ssq = 0.f;
        if (DBG & 8) {
            asm volatile("" ::"v"(pmax));
        } else if (ch < 80) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rho = (r & 3) + 8 * (r >> 2) + 4 * g;
                const int p = tile * 32 + ch_pi(rho);
                if (p < P0) {
                    const float v = pmax[r];
                    y0[((long long)bb * P0 + p) * 80 + ch] = v;
                    sum += v;
                    ssq += v * v;
                }
            }
        }
        sum += __shfl_xor(sum, 32, 64);
        ssq += __shfl_xor(ssq, 32, 64);
        if (ch < 80 && g == 0 && !(DBG & 8)) {
            float* pp = partials + (((long long)bb * ntile + tile) * 80 + ch) * 2;
            pp[0] = sum;
            pp[1] = ssq;
        }
        C0_STAMP();
        // LDS-only barrier (the parked copies become visible; the stores above drain on their own)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        C0_STAMP();
    }
#undef C0_STAMP
    dz_flag_range(oflag, amax);
}
#endif  // DZ_EXPERIMENTS (sinc_conv0_h)

// ---------------------------------------------------------------------------
// sinc_conv0_v2 (round 5): the same stage, rebuilt from a per-wave phase profile of sinc_conv0_h
// (tools/conv0_phases.py: of a tile's 17.2 k cycles a wave spends 10.0 k in the MFMA phase — 144 MFMAs = 4.6 k busy
// cycles — 3.4 k parking the next tile's samples, 3.1 k ISSUING 16 dword stores, 0.7 k at the barrier).
//   * FOUR waves, roles by the SIMD a wave landed on (HW_ID): two HEAVY waves own a 32-filter block each
//     (v_mfma_f32_32x32x16_f16, bank in 128 registers, exactly sinc_conv0_h's arithmetic), two LIGHT waves own the
//     last 16 filters on v_mfma_f32_16x16x32_f16 for one pooled-row half of the tile each (a quarter of a heavy
//     wave's matrix time, no zero filter columns) AND do all the fetching and parking of the next tile's samples.
//     The second workgroup of a CU takes the complementary SIMDs for its heavy waves: every SIMD carries one heavy
//     and one light wave, matrix work and VALU / LDS-write work side by side.
//   * fragment reads one step ahead through inline-asm ds_read_b128 with counted lgkmcnt waits (hipcc funnels
//     them through one register quad: read, wait, MFMA, read ...);
//   * results staged through a per-wave LDS tile and stored as dwordx4 rows (4 store instructions per wave and
//     tile instead of 16: the store phase is issue-bound).
// Same LDS sample copies, outputs and partials as sinc_conv0_h (the light waves' channels 64..79 accumulate in
// another order: last-bit differences there).
// ---------------------------------------------------------------------------
typedef float cq_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned cq_u32x4 __attribute__((ext_vector_type(4)));

// The bank in the order sinc_conv0_v2's waves load it (16-byte vectors): heavy block h in {0, 1}:
// [h][plane][k-step 16][lane 64] (lane = filter li + 32 g: taps 16 ks + 8 g .. + 7 of filter 32 h + li), then the
// light waves' 16 filters: [plane][k-step 8][lane 64] (lane = n + 16 q: taps 32 ks + 8 q .. + 7 of filter 64 + n).
// With the row-major planes every lane reads 16 bytes of its own 512-byte row: 32 cache lines per load instruction,
// 1 024 line requests per heavy wave — 7 - 10 k cycles of a workgroup's prologue went into issuing them
// (tools/conv0_phases.py, "own prologue loads issued").  Here an instruction reads 1 KiB of contiguous memory.
#define DZ_BANK_FRAG_HEAVY (2 * 2 * 16 * 64)                    /* vectors of the two heavy blocks */
#define DZ_BANK_FRAG_VECS (DZ_BANK_FRAG_HEAVY + 2 * 8 * 64)     /* 5 120 vectors = 80 KiB          */
__global__ void sinc_bank_frag_kernel(const unsigned short* __restrict__ fsp, cq_u32x4* __restrict__ out) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= DZ_BANK_FRAG_VECS) return;
    int plane, row, tap;
    if (v < DZ_BANK_FRAG_HEAVY) {
        const int lane = v & 63, ks = (v >> 6) & 15, pl = (v >> 10) & 1, h = v >> 11;
        plane = pl; row = 32 * h + (lane & 31); tap = 16 * ks + 8 * (lane >> 5);
    } else {
        const int u = v - DZ_BANK_FRAG_HEAVY, lane = u & 63, ks = (u >> 6) & 7, pl = u >> 9;
        plane = pl; row = 64 + (lane & 15); tap = 32 * ks + 8 * (lane >> 4);
    }
    out[v] = *reinterpret_cast<const cq_u32x4*>(fsp + ((long long)plane * 96 + row) * 256 + tap);
}
#define CQ_STAGE 4608                      /* bytes of result staging per wave: 32 rows x 36 floats            */
#define CQ_LDS (2 * CH_LDS + 1536 * 4 + 16 + 2 * 2 * 16 * 2 * 4 + 4 * CQ_STAGE + 16)
__device__ __forceinline__ char* stage_all_end(char* lds) { return lds + CQ_LDS - 16; }
__global__ __launch_bounds__(256, 2) void sinc_conv0_v2_kernel(
    const float* __restrict__ wave, long long stride, int S, const float* __restrict__ stats,
    int stats_are_moments, float gamma, float beta, const unsigned short* __restrict__ fsp,
    float* __restrict__ y0, int P0, float* __restrict__ partials, int ntile, int total,
    int* __restrict__ oflag, int rot_period, int rot_mode, long long* __restrict__ dbg,
    const unsigned short* __restrict__ ffrag) {
    // ffrag: the bank in FRAGMENT order (sinc_bank_frag_kernel below), or NULL: fsp's row-major planes.
    // dbg (experiments build, tools/conv0_phases.py): per wave shader-clock stamps per tile + (HW_ID | role << 32) in slot 63
#ifdef DZ_EXPERIMENTS
    long long* dq = nullptr;
    int dn = 0;
    if (dbg && (threadIdx.x & 63) == 0) dq = dbg + ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (dq) dq[62] = __builtin_readcyclecounter();          // kernel entry (slot 61: exit)
#define CQ_STAMP() do { if (dq && dn < 60) dq[dn++] = __builtin_readcyclecounter(); } while (0)
#else
#define CQ_STAMP() do { } while (0)
#endif
    // [2][CH_LDS] sample copies | raw[1536] landing zone | (mean, rstd) of 2 chunks | light sums [2][2][16][2] | staging
    __shared__ __attribute__((aligned(256))) char lds_all[CQ_LDS];
    char (*xs2)[CH_LDS] = reinterpret_cast<char (*)[CH_LDS]>(lds_all);
    float* raw = reinterpret_cast<float*>(lds_all + 2 * CH_LDS);
    float (*stat_s)[2] = reinterpret_cast<float (*)[2]>(lds_all + 2 * CH_LDS + 1536 * 4);
    float* lsum = reinterpret_cast<float*>(lds_all + 2 * CH_LDS + 1536 * 4 + 16);     // [t & 1][half][16][2]
    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63;
    float* stage = reinterpret_cast<float*>(lds_all + 2 * CH_LDS + 1536 * 4 + 16 + 512 + w * CQ_STAGE);
    const int wg = dz_xcd_contiguous(blockIdx.x, gridDim.x);
    const int t_begin = (int)((long long)wg * total / gridDim.x);
    const int t_end = (int)((long long)(wg + 1) * total / gridDim.x);
    if (t_begin >= t_end) return;
    const int b_first = t_begin / ntile;
    // a workgroup's four waves sit on the four SIMDs of its CU (cyclic placement from a varying start): the role
    // follows the SIMD.  First workgroup of a CU: heavy on SIMDs 0 / 1, light on 2 / 3; second: the other way round.
    // (Correctness never depends on that placement: the waves publish their SIMD ids, and unless the four are
    // distinct the roles fall back to the wave index.)
    const int simd = (__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 4) & 3;
    int* simd_of = reinterpret_cast<int*>(stage_all_end(lds_all));
    if (l == 0) simd_of[w] = simd;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef DZ_EXPERIMENTS
    if (dq) dq[60] = __builtin_readcyclecounter();          // roles known
#endif
    // (rot_mode, experiments build: 1 = no complementary roles, 2 = inverted, 3 = roles by wave index — the fall-back, so
    // that a test can run it on hardware that always places the four waves on four SIMDs)
    const bool by_simd = rot_mode != 3 && ((1 << simd_of[0]) | (1 << simd_of[1]) | (1 << simd_of[2]) | (1 << simd_of[3])) == 15;
    const int rot = rot_mode == 1 ? 0 : (((int)blockIdx.x / rot_period) & 1) ^ (rot_mode == 2);
    const int slot = by_simd ? simd : w;
    const bool heavy = ((slot >> 1) & 1) == rot;
    const int half = slot & 1;             // heavy: filter block 0 / 1; light: pooled-row half 0 / 1 of the tile
#ifdef DZ_EXPERIMENTS
    if (dq) dq[63] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)(heavy ? 1 : 0) << 32) | ((long long)half << 33);
#endif

    // (mean, rstd) of the one or two chunks this workgroup touches.  Called by every wave AFTER it has issued its own
    // prologue loads (filter bank to registers, first tile by LDS-DMA): the moments' round trip to L2 and the f64
    // combine run under those instead of in front of them, and the barriers of the prologue wait on LDS only
    // (__syncthreads() would drain vmcnt, i.e. serialise the bank loads with everything): prologue 18 k -> see
    // tools/conv0_phases.py ("prologue").
    auto chunk_stats = [&]() {
        if (tid < 2) {
            const int bb = b_first + tid;
            float mean = 0.f, rstd = 1.f;
            if (bb * ntile < t_end) {
                if (stats_are_moments)
                    dz_ws_combine(stats, bb, S, &mean, &rstd);
                else
                    mean = stats[2 * bb], rstd = stats[2 * bb + 1];
            }
            stat_s[tid][0] = mean;
            stat_s[tid][1] = rstd;
        }
    };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    float amax = 0.f;

    if (heavy) {
        // ================= a 32-filter block: sinc_conv0_h's arithmetic, no parking =================
        const int li = l & 31, g = l >> 5;
        ch_f16x8 bh[16], bl[16];
        if (ffrag) {       // fragment order: every load instruction of the wave reads 1 KiB of contiguous memory
            const ch_f16x8* fr = reinterpret_cast<const ch_f16x8*>(ffrag) + (half * 2) * 16 * 64 + l;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                bh[ks] = fr[ks * 64];
                bl[ks] = fr[(16 + ks) * 64];
            }
        } else {           // row-major planes: lane (filter, k-half) reads 16 bytes of ITS row — 32 lines per instruction
            const unsigned short* row = fsp + (long long)(32 * half + li) * 256 + 8 * g;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                bh[ks] = *reinterpret_cast<const ch_f16x8*>(row + 16 * ks);
                bl[ks] = *reinterpret_cast<const ch_f16x8*>(row + 96 * 256 + 16 * ks);
            }
        }
        const int m_a = ch_pi(li);
        int aoff[3];
#pragma unroll
        for (int bk = 0; bk < 3; ++bk) {
            const int f = 3 * m_a + bk, c = f & 3;
            aoff[bk] = ch_copy_base(c) + 2 * (10 * f - 2 * c + 8 * g);
        }
        const int ch = 32 * half + li;
#ifdef DZ_EXPERIMENTS
        if (dq) dq[59] = __builtin_readcyclecounter();      // own loads issued
#endif
        chunk_stats();
#ifdef DZ_EXPERIMENTS
        if (dq) dq[58] = __builtin_readcyclecounter();      // statistics done (wave 0) / nothing
#endif
        lds_barrier();                     // the statistics are published
#ifdef DZ_EXPERIMENTS
        if (dq) dq[57] = __builtin_readcyclecounter();
#endif
        lds_barrier();                     // the light waves have parked the first tile
        for (int t = t_begin; t < t_end; ++t) {
            const char* xs = xs2[(t - t_begin) & 1];
            CQ_STAMP();
            unsigned abase[3];
#pragma unroll
            for (int bk = 0; bk < 3; ++bk)
                abase[bk] = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)(xs + aoff[bk]);
#define CQ_LD(it, H, L)                                                                                        \
    do {                                                                                                       \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(H) : "v"(abase[(it) >> 4]), "n"(32 * ((it) & 15)));          \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(L) : "v"(abase[(it) >> 4]), "n"(32 * ((it) & 15) + CH_PL)); \
    } while (0)
            ch_f32x16 pmax, accm, accx;
            ch_f16x8 fh[2], fl[2];
            CQ_LD(0, fh[0], fl[0]);
#define CQ_STEP(it)                                                                                        \
    {                                                                                                      \
        constexpr int bk_ = (it) >> 4, ks_ = (it) & 15, cur_ = (it) & 1;                                   \
        if (ks_ == 0) {                                                                                    \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) accm[r] = accx[r] = 0.f;                        \
        }                                                                                                  \
        if ((it) + 1 < 48) {                                                                               \
            CQ_LD(((it) + 1 < 48 ? (it) + 1 : 47), fh[((it) + 1) & 1], fl[((it) + 1) & 1]);                \
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fh[cur_]), "+v"(fl[cur_]));                         \
        } else {                                                                                           \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fh[cur_]), "+v"(fl[cur_]));                         \
        }                                                                                                  \
        accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[cur_], bh[ks_], accx, 0, 0, 0);                   \
        accm = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[cur_], bh[ks_], accm, 0, 0, 0);                   \
        accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[cur_], bl[ks_], accx, 0, 0, 0);                   \
        if (ks_ == 15) {                                                                                   \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                               \
                const float v = fabsf(accm[r] + accx[r] * (1.f / 2048.f));                                 \
                pmax[r] = bk_ == 0 ? v : fmaxf(pmax[r], v);                                                \
            }                                                                                              \
        }                                                                                                  \
    }
#define CQ_STEP4(b) CQ_STEP((b)) CQ_STEP((b) + 1) CQ_STEP((b) + 2) CQ_STEP((b) + 3)
#define CQ_STEP16(b) CQ_STEP4((b)) CQ_STEP4((b) + 4) CQ_STEP4((b) + 8) CQ_STEP4((b) + 12)
            CQ_STEP16(0) CQ_STEP16(16) CQ_STEP16(32)
#undef CQ_STEP16
#undef CQ_STEP4
#undef CQ_STEP
#undef CQ_LD
            CQ_STAMP();
            CQ_STAMP();          // (no parking on a heavy wave: the phase is empty, the slot keeps the five-stamp layout)
            // results: stage the 32 x 32 tile (row = pooled row, pitch 36 floats), store rows as dwordx4
            const int bb = t / ntile, tile = t - bb * ntile;
            float sum = 0.f, ssq = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = ch_pi((r & 3) + 8 * (r >> 2) + 4 * g);
                stage[m * 36 + li] = pmax[r];
                if (tile * 32 + m < P0) {
                    sum += pmax[r];
                    ssq += pmax[r] * pmax[r];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's own writes (LDS ops of a wave retire in order)
            {
                // the chunk's y0 as a buffer of P0 rows: a row >= P0 of the ragged last tile lands beyond num_records
                // and is dropped by the hardware — no per-row branch, 32-bit offsets
                const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(y0 + (long long)bb * P0 * 80), 0, (unsigned)P0 * 320u, 0x00020000);
                f32x4 v[4];
#pragma unroll
                for (int pass = 0; pass < 4; ++pass)
                    v[pass] = *reinterpret_cast<const f32x4*>(stage + (8 * pass + (l >> 3)) * 36 + 4 * (l & 7));
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int p = tile * 32 + 8 * pass + (l >> 3);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cq_u32x4, v[pass]), yr,
                                                           (p * 80 + 32 * half + 4 * (l & 7)) * 4, 0, 0);
                }
            }
            sum += __shfl_xor(sum, 32, 64);
            ssq += __shfl_xor(ssq, 32, 64);
            if (g == 0) {
                float* pp = partials + (((long long)bb * ntile + tile) * 80 + ch) * 2;
                pp[0] = sum;
                pp[1] = ssq;
            }
            CQ_STAMP();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            CQ_STAMP();
        }
    } else {
        // ====== the last 16 filters for one pooled-row half of the tile + ALL fetching / parking of samples ======
        const int n = l & 15, q = l >> 4;
        ch_f16x8 bh[8], bl[8];
        if (ffrag) {
            const ch_f16x8* fr = reinterpret_cast<const ch_f16x8*>(ffrag) + DZ_BANK_FRAG_HEAVY + l;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                bh[ks] = fr[ks * 64];
                bl[ks] = fr[(8 + ks) * 64];
            }
        } else {
            const unsigned short* row = fsp + (long long)(64 + n) * 256 + 8 * q;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                bh[ks] = *reinterpret_cast<const ch_f16x8*>(row + 32 * ks);
                bl[ks] = *reinterpret_cast<const ch_f16x8*>(row + 96 * 256 + 32 * ks);
            }
        }
        // raw samples of the NEXT tile by LDS-DMA, one dword per lane: light wave `half` owns floats
        // [640 half, 640 half + 640) of the tile's window and later parks exactly those (its own vmcnt)
        auto fetch = [&](int t) {
            const int bb = t / ntile, tile = t - bb * ntile;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(wave + (long long)bb * stride), 0, (unsigned)S * 4u, 0x00020000);
            const int voff = (tile * (CH_FR * 10) + 640 * half + l) * 4;
#pragma unroll
            for (int i = 0; i < 10; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    rsrc, (__attribute__((address_space(3))) void*)(raw + 640 * half + 64 * i), 4, voff + 256 * i, 0, 0, 0);
        };
        // normalise (InstanceNorm1d(1)), split, write the four shifted copies: 4 samples per lane and pass, 8-byte
        // writes, no per-copy bounds checks — copy c takes x[j] at index j - 2 c, and the few indices that fall
        // outside [0, CH_NS) (j < 6 at the head, j >= CH_NS at the tail) land in the padding between the planes /
        // copy slots (CH_PL - 2 CH_NS = 128 bytes behind every plane, >= 192 bytes in front of copies 1 - 3; copy 0
        // never goes negative), where nobody reads
        auto park = [&](int t, char* xs) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int bb = t / ntile, tile = t - bb * ntile;
            const float mean = stat_s[bb - b_first][0], rstd = stat_s[bb - b_first][1];
            const int s0 = tile * (CH_FR * 10);
            typedef _Float16 cq_f16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int j = 640 * half + 4 * (l + 64 * it), sidx = s0 + j;        // 3 x 256 samples cover the wave's 640
                if (j < 640 * half + 640 && j < CH_NS + 8) {
                    const f32x4 pv = *reinterpret_cast<const f32x4*>(raw + j);
                    f32x4 x;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        x[e] = sidx + e < S ? ((pv[e] - mean) * rstd) * gamma + beta : 0.f;
                        amax = fmaxf(amax, fabsf(x[e]));
                        x[e] = __builtin_amdgcn_fmed3f(x[e], -65504.f, 65504.f);
                    }
                    const cq_f16x4 hi = __builtin_convertvector(x, cq_f16x4);
                    const cq_f16x4 lo = __builtin_convertvector((x - __builtin_convertvector(hi, f32x4)) * 2048.f, cq_f16x4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        char* d = xs + ch_copy_base(c) + 2 * (j - 2 * c);               // 4-byte aligned (j % 4 == 0)
                        if (c & 1) {                                                      // j - 2 c = 2 mod 4: two 4-byte halves
                            *reinterpret_cast<ch_f16x2*>(d) = ch_f16x2{hi[0], hi[1]};
                            *reinterpret_cast<ch_f16x2*>(d + 4) = ch_f16x2{hi[2], hi[3]};
                            *reinterpret_cast<ch_f16x2*>(d + CH_PL) = ch_f16x2{lo[0], lo[1]};
                            *reinterpret_cast<ch_f16x2*>(d + CH_PL + 4) = ch_f16x2{lo[2], lo[3]};
                        } else {
                            *reinterpret_cast<cq_f16x4*>(d) = hi;
                            *reinterpret_cast<cq_f16x4*>(d + CH_PL) = lo;
                        }
                    }
                }
            }
        };
        // the partial of channels 64..79 of a tile = the two light waves' halves, added in a fixed order by the
        // half-0 wave one barrier after both wrote them
        auto light_partial = [&](int t) {
            if (half == 0 && l < 32) {
                const int bb = t / ntile, tile = t - bb * ntile;
                const float* ls = lsum + ((t - t_begin) & 1) * 64;
                partials[(((long long)bb * ntile + tile) * 80 + 64) * 2 + l] = ls[l] + ls[32 + l];
            }
        };
        int aoff[3];
#pragma unroll
        for (int bk = 0; bk < 3; ++bk) {
            const int f = 3 * (16 * half + (((n & 3) << 2) | (n >> 2))) + bk, c = f & 3;
            aoff[bk] = ch_copy_base(c) + 2 * (10 * f - 2 * c + 8 * q);
        }
        fetch(t_begin);
#ifdef DZ_EXPERIMENTS
        if (dq) dq[59] = __builtin_readcyclecounter();
#endif
        chunk_stats();
#ifdef DZ_EXPERIMENTS
        if (dq) dq[58] = __builtin_readcyclecounter();
#endif
        lds_barrier();
#ifdef DZ_EXPERIMENTS
        if (dq) dq[57] = __builtin_readcyclecounter();
#endif
        park(t_begin, xs2[0]);
        lds_barrier();
        for (int t = t_begin; t < t_end; ++t) {
            const char* xs = xs2[(t - t_begin) & 1];
            CQ_STAMP();
            if (t + 1 < t_end) fetch(t + 1);
            if (t > t_begin) light_partial(t - 1);           // both halves of tile t - 1 are in LDS since the barrier
            unsigned abase[3];
#pragma unroll
            for (int bk = 0; bk < 3; ++bk)
                abase[bk] = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)(xs + aoff[bk]);
#define CQ_LD(it, H, L)                                                                                        \
    do {                                                                                                       \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(H) : "v"(abase[(it) >> 3]), "n"(64 * ((it) & 7)));          \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(L) : "v"(abase[(it) >> 3]), "n"(64 * ((it) & 7) + CH_PL)); \
    } while (0)
            cq_f32x4 pm, accm, accx;
            ch_f16x8 fh[2], fl[2];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the compiler's own LDS traffic above is not counted below)
            CQ_LD(0, fh[0], fl[0]);
#define CQ_STEP(it)                                                                                        \
    {                                                                                                      \
        constexpr int bk_ = (it) >> 3, ks_ = (it) & 7, cur_ = (it) & 1;                                    \
        if (ks_ == 0) accm = accx = cq_f32x4{0.f, 0.f, 0.f, 0.f};                                          \
        if ((it) + 1 < 24) {                                                                               \
            CQ_LD(((it) + 1 < 24 ? (it) + 1 : 23), fh[((it) + 1) & 1], fl[((it) + 1) & 1]);                \
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fh[cur_]), "+v"(fl[cur_]));                         \
        } else {                                                                                           \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fh[cur_]), "+v"(fl[cur_]));                         \
        }                                                                                                  \
        accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[cur_], bh[ks_], accx, 0, 0, 0);                   \
        accm = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[cur_], bh[ks_], accm, 0, 0, 0);                   \
        accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[cur_], bl[ks_], accx, 0, 0, 0);                   \
        if (ks_ == 7) {                                                                                    \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                \
                const float v = fabsf(accm[i] + accx[i] * (1.f / 2048.f));                                 \
                pm[i] = bk_ == 0 ? v : fmaxf(pm[i], v);                                                    \
            }                                                                                              \
        }                                                                                                  \
    }
#define CQ_STEP8(b) CQ_STEP((b)) CQ_STEP((b) + 1) CQ_STEP((b) + 2) CQ_STEP((b) + 3) CQ_STEP((b) + 4) CQ_STEP((b) + 5) CQ_STEP((b) + 6) CQ_STEP((b) + 7)
            CQ_STEP8(0) CQ_STEP8(8) CQ_STEP8(16)
#undef CQ_STEP8
#undef CQ_STEP
#undef CQ_LD
            CQ_STAMP();
            if (t + 1 < t_end) park(t + 1, xs2[(t + 1 - t_begin) & 1]);
            CQ_STAMP();
            const int bb = t / ntile, tile = t - bb * ntile;
            float sum = 0.f, ssq = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {                         // D row 4 q + i <-> pooled row pi16 = 4 i + q of this half
                stage[(4 * i + q) * 20 + n] = pm[i];
                if (tile * 32 + 16 * half + 4 * i + q < P0) {
                    sum += pm[i];
                    ssq += pm[i] * pm[i];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            {
                const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(y0 + (long long)bb * P0 * 80), 0, (unsigned)P0 * 320u, 0x00020000);
                const int m = l >> 2, quad = l & 3, p = tile * 32 + 16 * half + m;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stage + m * 20 + 4 * quad);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cq_u32x4, v), yr, (p * 80 + 64 + 4 * quad) * 4, 0, 0);
            }
            sum += __shfl_xor(sum, 16, 64);
            ssq += __shfl_xor(ssq, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            ssq += __shfl_xor(ssq, 32, 64);
            if (q == 0) {
                float* ls = lsum + ((t - t_begin) & 1) * 64 + half * 32 + 2 * n;
                ls[0] = sum;
                ls[1] = ssq;
            }
            CQ_STAMP();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            CQ_STAMP();
        }
        light_partial(t_end - 1);
    }
#ifdef DZ_EXPERIMENTS
    if (dq) dq[61] = __builtin_readcyclecounter();
#endif
#undef CQ_STAMP
    dz_flag_range(oflag, amax);
}

int dz_conv0_split_ntile(int F0) { return (F0 + CH_FR - 1) / CH_FR; }

// fsp: the UNFOLDED bank as f16 planes [2][96][256] (weights.py split_f16 of the zero-padded
// [96][256] filter matrix); partials [B][ntile][80][2] with ntile = dz_conv0_split_ntile(F0)
int dz_sinc_bank_frag_bytes() { return DZ_BANK_FRAG_VECS * 16; }
int dz_launch_sinc_bank_frag(const void* fsp, void* frag, hipStream_t st) {
    DZ_LAUNCH(sinc_bank_frag_kernel, dim3((DZ_BANK_FRAG_VECS + 255) / 256), dim3(256), 0, st,
              reinterpret_cast<const unsigned short*>(fsp), reinterpret_cast<cq_u32x4*>(frag));
    DZ_HIP(hipGetLastError());
    return 0;
}

int dz_launch_sinc_conv0_split(const float* wave, long long stride, int B, int S, const float* stats,
                               int stats_are_moments, float gamma, float beta, const void* fsp,
                               float* y0, int P0, float* partials, int ntile, hipStream_t st, const void* ffrag) {
    const int total = ntile * B;
    // two resident workgroups per CU — and never a tile range that touches three chunks: a workgroup keeps the
    // (mean, rstd) of its first two chunks only (stat_s), so its range holds at most ntile + 1 tiles (beyond
    // ~520 chunks of 5 s the 512-workgroup grid gave ranges of ntile + 2 tiles and the third chunk read past stat_s)
    const int need = (total + ntile) / (ntile + 1);
    const int grid = total < 512 ? total : (need > 512 ? need : 512);
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            n = 256;
        return n > 0 ? n : 256;
    }();
    int rot_mode = 0;
    long long* dbg_ptr = nullptr;
#ifdef DZ_EXPERIMENTS
    dbg_ptr = dz_conv_pool_dbg;                    // the stamp buffer of dz_k_conv_pool_debug (tools/conv0_phases.py)
    // DZ_CONV0_ROT: 1 = no complementary roles, 2 = inverted, 3 = roles by wave index; DZ_CONV0_V2=0: the three-wave kernel of rounds 2 - 4
    const char* e_rot = dz_exp_env("DZ_CONV0_ROT");
    if (e_rot) rot_mode = atoi(e_rot);
    const char* e_v2 = dz_exp_env("DZ_CONV0_V2");
    if (e_v2 && e_v2[0] == '0') {
#define DZ_C0(D)                                                                                          \
    DZ_LAUNCH(sinc_conv0_h_kernel<D>, dim3(grid), dim3(192), 0, st, wave, stride, S, stats,                \
              stats_are_moments, gamma, beta, reinterpret_cast<const unsigned short*>(fsp), y0, P0,        \
              partials, ntile, total, dz_cur_oflag, dbg_ptr)
        const char* e_dbg = dz_exp_env("DZ_CONV0_DBG");     // timing-only instantiations: results are wrong
        switch (e_dbg ? atoi(e_dbg) : 0) {
            case 1: DZ_C0(1); break;
            case 2: DZ_C0(2); break;
            case 4: DZ_C0(4); break;
            case 8: DZ_C0(8); break;
            case 9: DZ_C0(9); break;
            case 13: DZ_C0(13); break;
            case 15: DZ_C0(15); break;
            case 6: DZ_C0(6); break;
            default: DZ_C0(0); break;
        }
#undef DZ_C0
        DZ_HIP(hipGetLastError());
        return 0;
    }
#endif
    DZ_LAUNCH(sinc_conv0_v2_kernel, dim3(grid), dim3(256), 0, st, wave, stride, S, stats, stats_are_moments, gamma, beta,
              reinterpret_cast<const unsigned short*>(fsp), y0, P0, partials, ntile, total, dz_cur_oflag, cus, rot_mode,
              dbg_ptr, reinterpret_cast<const unsigned short*>(ffrag));
    DZ_HIP(hipGetLastError());
    return 0;
}

#ifdef DZ_EXPERIMENTS
// ---------------------------------------------------------------------------
// sinc_conv0_pair: the first SincNet stage of BOTH networks in one launch (round 5).
//
// The segmentation and the embedding network read the same window, and InstanceNorm1d(1) differs
// between them only by its affine pair: with xh = (x - mean) * rstd,
//     conv(gamma xh + beta)[c] = gamma * conv(xh)[c] + beta * sum_k filt[c][k]
// so ONE split of xh serves both banks and (gamma, beta * sum) become an epilogue in front of the
// |.|.  160 filters = 4 waves x 40: a wave owns three 16-filter blocks of v_mfma_f32_16x16x32_f16
// (the third half empty: 48 slots, 83 % of the matrix work useful — the two separate launches padded
// 80 to 96 as well, but ran three waves per workgroup, i.e. 2, 2, 1, 1 waves on the four SIMDs).
// One workgroup of four waves per CU, ONE WAVE PER SIMD, up to 512 registers each: the bank
// (48 slots x 256 taps x 2 planes) is 192 registers per lane, 18 independent accumulator chains
// (3 frame blocks x 3 filter blocks x (main, cross)) keep a single wave's matrix pipe full.
// Samples: the tile's four shifted f16 copies as in sinc_conv0_h (parked once for BOTH networks).
// MFMA row r of a 16-row block carries pooled row pi(r) = 4 (r & 3) + (r >> 2), block bk = frames
// 3 m + bk: MaxPool1d(3) is a maximum over the same register of three accumulators; copies start
// 16 * (0, 0, 1, 2) bytes into their slots (at most 2-way conflicts in every ds_read_b128 lane group,
// brute-forced against the group lists of MI355X_MICROARCH.md).
// Outputs: y0 / partials of each network, in the layout sinc_conv0_h writes (conv_pool_h consumes).
//
// EXPERIMENTS BUILD ONLY (DZ_CONV0_PAIR=1).  Measured on MI355X, 64 chunks (profiles/r05c_*): results equal to the
// two one-network launches (tests/test_gpu_kernels.py::test_sinc_conv0_pair), but 200.9 us alone against 2 x 101.4 us,
// and in the 64-stream pipeline 1.227 - 1.235 ms per step against 1.150 (same-visit A/B, 200 steps, twice): a
// 420-register wave per SIMD shares its CU with nothing, so the launch serialises with every other stream, and
// hipcc's schedule for ONE wave per SIMD leaves the matrix pipe idle behind accumulator read-backs, LDS waits and
// the per-block epilogue.  DESIGN.md "measured and left out" has the arithmetic of why 160 filters do not balance
// on four SIMDs at two waves each.
// ---------------------------------------------------------------------------
typedef float cp_f32x4 __attribute__((ext_vector_type(4)));
#define CP_SLOTS 192
__device__ __forceinline__ int cp_copy_base(int c) { return c * CH_CP + 16 * ((0x2100 >> (4 * c)) & 15); }

__global__ __launch_bounds__(256, 1) void sinc_conv0_pair_kernel(
    const float* __restrict__ wave, long long stride, int S, const float* __restrict__ stats,
    const unsigned short* __restrict__ fsp, const float* __restrict__ bsum, float gamma_seg, float gamma_emb,
    float* __restrict__ y0_seg, float* __restrict__ y0_emb, int P0, float* __restrict__ part_seg,
    float* __restrict__ part_emb, int ntile, int total, int* __restrict__ oflag) {
    __shared__ __attribute__((aligned(256))) char lds_all[2 * CH_LDS + 1536 * 4 + 16];
    char (*xs2)[CH_LDS] = reinterpret_cast<char (*)[CH_LDS]>(lds_all);
    float* raw = reinterpret_cast<float*>(lds_all + 2 * CH_LDS);
    float (*stat_s)[2] = reinterpret_cast<float (*)[2]>(lds_all + 2 * CH_LDS + 1536 * 4);
    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63, n = l & 15, q = l >> 4;
    const int wg = dz_xcd_contiguous(blockIdx.x, gridDim.x);
    const int t_begin = (int)((long long)wg * total / gridDim.x);
    const int t_end = (int)((long long)(wg + 1) * total / gridDim.x);
    if (t_begin >= t_end) return;
    const int b_first = t_begin / ntile;
    if (tid < 2) {
        const int bb = b_first + tid;
        float mean = 0.f, rstd = 1.f;
        if (bb * ntile < t_end) dz_ws_combine(stats, bb, S, &mean, &rstd);
        stat_s[tid][0] = mean;
        stat_s[tid][1] = rstd;
    }
    // ---- B fragments: slot 48 w + 16 j + n, taps 32 ks + 8 q .. + 7, hi and lo planes -----------------
    ch_f16x8 bh[3][8], bl[3][8];
    float bs[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int slot = 48 * w + 16 * j + n;
        const unsigned short* row = fsp + (long long)slot * 256 + 8 * q;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            bh[j][ks] = *reinterpret_cast<const ch_f16x8*>(row + 32 * ks);
            bl[j][ks] = *reinterpret_cast<const ch_f16x8*>(row + CP_SLOTS * 256 + 32 * ks);
        }
        bs[j] = bsum[slot];
    }
    const float gam = w < 2 ? gamma_seg : gamma_emb;
    float* __restrict__ y0 = w < 2 ? y0_seg : y0_emb;
    float* __restrict__ partials = w < 2 ? part_seg : part_emb;
    const int chb = 40 * (w & 1);
    __syncthreads();

    // raw samples of the NEXT tile by LDS-DMA, one dword per lane: wave w owns floats [384 w, 384 w + 384)
    // of the tile's 1536-float window and later parks exactly those (its own vmcnt, no barrier);
    // samples past the end of the chunk read as zeros (buffer bounds check)
    auto fetch = [&](int t) {
        const int bb = t / ntile, tile = t - bb * ntile;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(wave + (long long)bb * stride), 0, (unsigned)S * 4u, 0x00020000);
        const int voff = (tile * (CH_FR * 10) + 384 * w + l) * 4;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsrc, (__attribute__((address_space(3))) void*)(raw + 384 * w + 64 * i), 4, voff + 256 * i, 0, 0, 0);
    };
    float amax = 0.f;
    auto park = [&](int t, char* xs) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's six pieces have landed
        const int bb = t / ntile, tile = t - bb * ntile;
        const float mean = stat_s[bb - b_first][0], rstd = stat_s[bb - b_first][1];
        const int s0 = tile * (CH_FR * 10);
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int j = 384 * w + 2 * (l + 64 * it), sidx = s0 + j;
            if (j < CH_NS + 6) {
                const float2 pvq = *reinterpret_cast<const float2*>(raw + j);
                f32x2 x = {sidx < S ? (pvq.x - mean) * rstd : 0.f, sidx + 1 < S ? (pvq.y - mean) * rstd : 0.f};
                amax = fmaxf(amax, fmaxf(fabsf(x[0]), fabsf(x[1])));
                x[0] = __builtin_amdgcn_fmed3f(x[0], -65504.f, 65504.f);
                x[1] = __builtin_amdgcn_fmed3f(x[1], -65504.f, 65504.f);
                const ch_f16x2 hi = __builtin_convertvector(x, ch_f16x2);
                const ch_f16x2 lo =
                    __builtin_convertvector((x - __builtin_convertvector(hi, f32x2)) * 2048.f, ch_f16x2);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int idx = j - 2 * c;            // copy_c[idx] = x[idx + 2c]
                    if (idx >= 0 && idx < CH_NS) {
                        char* d = xs + cp_copy_base(c) + 2 * idx;
                        *reinterpret_cast<ch_f16x2*>(d) = hi;
                        *reinterpret_cast<ch_f16x2*>(d + CH_PL) = lo;
                    }
                }
            }
        }
    };

    // A rows: lane (n, q) supplies frame 3 (16 gm + pi(n)) + bk, taps 8 q .. + 7 of every k-step
    int aoff[3];
#pragma unroll
    for (int bk = 0; bk < 3; ++bk) {
        const int f = 3 * (((n & 3) << 2) | (n >> 2)) + bk, c = f & 3;
        aoff[bk] = cp_copy_base(c) + 2 * (10 * f - 2 * c + 8 * q);
    }

    fetch(t_begin);
    park(t_begin, xs2[0]);
    __syncthreads();
    for (int t = t_begin; t < t_end; ++t) {
        const char* xs = xs2[(t - t_begin) & 1];
        if (t + 1 < t_end) fetch(t + 1);
        const int bb = t / ntile, tile = t - bb * ntile;
        float sum[3] = {0.f, 0.f, 0.f}, ssq[3] = {0.f, 0.f, 0.f};
#pragma unroll 1
        for (int gm = 0; gm < 2; ++gm) {
            cp_f32x4 pm[3];
            // the three frame blocks of a pooled-row group one after the other: six accumulator chains (3 filter
            // blocks x (main, cross)) are live at a time, the maximum over bk accumulates in pm (more live
            // accumulators made the allocator shuffle results through AGPRs behind s_nop stalls).  The A
            // fragments of step it + 1 are requested before the nine MFMAs of step it (one wave per SIMD: nobody
            // else hides the LDS latency).
            const char* const ag = xs + 960 * gm;
            ch_f16x8 ah = *reinterpret_cast<const ch_f16x8*>(ag + aoff[0]);
            ch_f16x8 al = *reinterpret_cast<const ch_f16x8*>(ag + aoff[0] + CH_PL);
            cp_f32x4 accm[3], accx[3];
#pragma unroll
            for (int it = 0; it < 24; ++it) {
                const int bk = it >> 3, ks = it & 7;
                if (ks == 0) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) accm[j] = accx[j] = cp_f32x4{0.f, 0.f, 0.f, 0.f};
                }
                ch_f16x8 nh = ah, nl = al;
                if (it + 1 < 24) {
                    const char* np = ag + aoff[(it + 1) >> 3] + 64 * ((it + 1) & 7);
                    nh = *reinterpret_cast<const ch_f16x8*>(np);
                    nl = *reinterpret_cast<const ch_f16x8*>(np + CH_PL);
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) accx[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[j][ks], accx[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 3; ++j) accm[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[j][ks], accm[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 3; ++j) accx[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[j][ks], accx[j], 0, 0, 0);
                ah = nh;
                al = nl;
                if (ks == 7) {       // |gamma conv + beta sum|, MaxPool1d(3) over bk
#pragma unroll
                    for (int j = 0; j < 3; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float v = fabsf(gam * (accm[j][i] + accx[j][i] * (1.f / 2048.f)) + bs[j]);
                            pm[j][i] = bk == 0 ? v : fmaxf(pm[j][i], v);
                        }
                }
            }
            // The next tile's samples (this wave's share of them) BEFORE the last result stores: parking waits for
            // the wave's LDS-DMA with vmcnt(0), which must not also wait for stores that were just issued (one wave
            // per SIMD: nobody hides it).  Here the only stores in flight are those of gm = 0, a block of MFMAs old.
            if (gm == 1 && t + 1 < t_end) park(t + 1, xs2[(t + 1 - t_begin) & 1]);
            // pooled rows 16 gm + 4 i + q, channel chb + 16 j + n
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int chl = 16 * j + n;
                if (chl < 40) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int p = tile * 32 + 16 * gm + 4 * i + q;
                        if (p < P0) {
                            y0[((long long)bb * P0 + p) * 80 + chb + chl] = pm[j][i];
                            sum[j] += pm[j][i];
                            ssq[j] += pm[j][i] * pm[j][i];
                        }
                    }
                }
            }
        }
        // InstanceNorm partials of (chunk, tile, channel): the four lane groups q hold 8 pooled rows each
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float s1 = sum[j], s2 = ssq[j];
            s1 += __shfl_xor(s1, 16, 64);
            s2 += __shfl_xor(s2, 16, 64);
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            const int chl = 16 * j + n;
            if (q == 0 && chl < 40) {
                float* pp = partials + (((long long)bb * ntile + tile) * 80 + chb + chl) * 2;
                pp[0] = s1;
                pp[1] = s2;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    dz_flag_range(oflag, amax);
}

// fsp: f16 planes [2][192][256] of the PAIR bank (slot 48 w + i = filter 40 w + i of seg | emb, i < 40; the other
// slots zero; weights.py pack_conv0_pair); bsum[192] = beta_net * sum_k filt[slot][k]; ntile = dz_conv0_split_ntile(F0)
int dz_launch_sinc_conv0_pair(const float* wave, long long stride, int B, int S, const float* moments,
                              const void* fsp, const float* bsum, float gamma_seg, float gamma_emb, float* y0_seg,
                              float* y0_emb, int P0, float* part_seg, float* part_emb, int ntile, hipStream_t st) {
    const int total = ntile * B;
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    const int need = (total + ntile) / (ntile + 1);    // (a range touches at most two chunks: see dz_launch_sinc_conv0_split)
    const int grid = total < cus ? total : (need > cus ? need : cus);        // one workgroup (one wave per SIMD) per CU
    DZ_LAUNCH(sinc_conv0_pair_kernel, dim3(grid), dim3(256), 0, st, wave, stride, S, moments,
              reinterpret_cast<const unsigned short*>(fsp), bsum, gamma_seg, gamma_emb, y0_seg, y0_emb, P0, part_seg,
              part_emb, ntile, total, dz_cur_oflag);
    DZ_HIP(hipGetLastError());
    return 0;
}

#endif  // DZ_EXPERIMENTS

// ---------------------------------------------------------------------------
// finalize_norm: fixed-order (deterministic) reduction of the tile partials in fp64.
// scale = gamma * rstd, shift = beta - mean * scale  (InstanceNorm1d, eps 1e-5, biased var)
// ---------------------------------------------------------------------------
__global__ void finalize_norm_kernel(const float* __restrict__ partials, int B, int ntile, int C,
                                     int T, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, float* __restrict__ scale,
                                     float* __restrict__ shift) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * C) return;
    const int b = idx / C, c = idx - b * C;
    double s = 0.0, ss = 0.0;
    for (int t = 0; t < ntile; ++t) {
        const float* pp = partials + (((long long)b * ntile + t) * C + c) * 2;
        s += (double)pp[0];
        ss += (double)pp[1];
    }
    const double mean = s / T;
    double var = ss / T - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + 1e-5);
    const double sc = (double)gamma[c] * rstd;
    scale[idx] = (float)sc;
    shift[idx] = (float)((double)beta[c] - mean * sc);
}

int dz_launch_finalize_norm(const float* partials, int B, int ntile, int C, int T,
                            const float* gamma, const float* beta, float* scale, float* shift,
                            hipStream_t st) {
    const int n = B * C;
    DZ_LAUNCH(finalize_norm_kernel, dim3((n + 255) / 256), dim3(256), 0, st, partials, B,
                       ntile, C, T, gamma, beta, scale, shift);
    DZ_HIP(hipGetLastError());
    return 0;
}
