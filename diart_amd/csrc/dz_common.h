// Shared declarations for libdiart_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stddef.h>
#include <mutex>
#include "../../include/diart_amd.h"

// ---- build flavours ---------------------------------------------------------------------------------
// The shipped library has ONE configuration of every layer.  Kernels and switches that only exist to measure
// alternatives — k_gemm_g2.hip / k_gemm_g3.hip, the two-chunk and plain-FMA recurrences, lstm_mfma variants
// 1 / 2, the old multi-launch paths, and the timing-only instantiations whose RESULTS ARE WRONG — are compiled
// only with -DDZ_EXPERIMENTS (`python -m diart_amd.build --experiments` -> libdiart_amd_exp.so, loaded with
// DZ_EXPERIMENTS=1), where DZ_* environment variables select them through dz_exp_env().  In the shipped
// build dz_exp_env() is a null constant: the branches fold away and the variables are never read.
#ifdef DZ_EXPERIMENTS
#include <stdlib.h>
static inline const char* dz_exp_env(const char* name) { return getenv(name); }
#else
#define dz_exp_env(name) ((const char*)nullptr)
#endif
// Run-time options of the shipped library (api.hip: dz_set_option / dz_get_option, by name).
//   f32_gemm  (1): exact-f32 wide layers on k_gemm_f32.hip; 0 keeps them on k_convgemm.hip (the kernel the
//                  prologue layers use anyway; tests compare the two)
//   pool_fuse (1): statistics pooling inside tdnn5's epilogue; 0 = tdnn5 + stats_pool (the exact-f32 form)
enum { DZ_OPT_F32_GEMM = 0, DZ_OPT_POOL_FUSE, DZ_OPT_PACK_CACHE, DZ_OPT_COUNT };
int dz_option(int id);

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// v_mfma_f32_16x16x4_f32: exact-f32 matrix FMA, 32 cycles/SIMD issue (MI355X_MICROARCH.md).
// Lane l supplies A[i=l&15][k=l>>4] and B[k=l>>4][j=l&15]; D: col j=l&15, rows 4*(l>>4)+r.
#define DZ_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

#define DZ_LEAKY_SLOPE 0.01f

// hipFuncSetAttribute is a per-device setting and a process may drive several GPUs (one dz_ctx
// each) from several host threads: raise a kernel's dynamic-LDS limit once per device, under a
// lock that is held until the attribute is set (a second thread launching the same kernel for the
// first time waits for it instead of launching with the old limit).
struct DzAttrOnce {
    std::mutex mu;
    bool done[64] = {};
    hipError_t raise(const void* fn, int bytes) {
        int d = 0;
        (void)hipGetDevice(&d);
        std::lock_guard<std::mutex> lk(mu);
        if (d >= 0 && d < 64 && done[d]) return hipSuccess;
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess && d >= 0 && d < 64) done[d] = true;
        return e;
    }
};

// Every kernel of the library is launched through DZ_LAUNCH.  When the per-kernel profiler of
// api.hip is on, the launch carries a (start, stop) event pair that the runtime fills with the
// dispatch's own begin / end timestamps (hipExtLaunchKernelGGL) — the same numbers rocprofv3's
// kernel trace reports, with no extra marker packets in the stream.  Otherwise a plain launch.
struct DzLaunchProf {
    hipEvent_t start, stop;
};
// Bracket for bench.py's per-kernel table (api.hip): the NEXT DZ_LAUNCH on this thread carries the event pair.
struct DzProfScope {
    DzProfScope(int tag, int units);
    ~DzProfScope();
};
enum { DZ_T_ECAPA_FBANK = 21, DZ_T_ECAPA_BLOCK0, DZ_T_ECAPA_WIDE, DZ_T_ECAPA_RES2, DZ_T_ECAPA_SE, DZ_T_ECAPA_ASP,
       DZ_T_ECAPA_FC };
// range flag (dz_ctx::oflag_dev) of the context whose forward pass is being enqueued on this host
// thread: picked up by the split-f16 launchers when the descriptor does not name one
extern thread_local int* dz_cur_oflag;
struct DzRangeScope {
    int* prev;
    explicit DzRangeScope(int* f) : prev(dz_cur_oflag) { dz_cur_oflag = f; }
    ~DzRangeScope() { dz_cur_oflag = prev; }
};
extern thread_local DzLaunchProf* dz_launch_prof;
#define DZ_LAUNCH(kernel, grid, block, lds, st, ...)                                          \
    do {                                                                                      \
        if (dz_launch_prof) {                                                                 \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, st, dz_launch_prof->start,        \
                                  dz_launch_prof->stop, 0, __VA_ARGS__);                      \
            dz_launch_prof = nullptr; /* one launch per bracket */                            \
        } else {                                                                              \
            hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                    \
        }                                                                                     \
    } while (0)

// ---------------------------------------------------------------------------
// XCD-aware tile order shared by the GEMM kernels (grid = (M-tiles, N-tiles, batch)).
// Workgroups are dealt round-robin to the 8 XCDs, each with a private 4 MiB L2.  With the
// plain (x, y, z) order the N-tiles that share one activation tile land on 8 different
// XCDs and every L2 fetches it again (rocprofv3 FETCH_SIZE of the TDNN layers was 4.7x the
// algorithmic bytes).  Re-order so that XCD r owns activation tiles a = r, r+8, ...; inside
// an XCD, groups of AG activation tiles sweep the N-tiles together (weight slice reuse).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void dz_tile_map_lin(int L, int gx, int gy, int gz, int agroup, int& bx, int& by,
                                                int& bz) {
    const int NA = gx * gz, NA8 = NA & ~7;
    int a;
    if (L < NA8 * gy) {
        const int xcd = L & 7, j = L >> 3;          // j-th workgroup of this XCD
        const int per = NA8 >> 3;                   // activation tiles per XCD
        const int AG = agroup > 0 ? agroup : 4;
        const int g = j / (AG * gy), r = j - g * (AG * gy);
        const int gsz = per - g * AG < AG ? per - g * AG : AG;  // last group may be short
        by = r / gsz;
        a = (g * AG + (r - by * gsz)) * 8 + xcd;
    } else {
        const int r = L - NA8 * gy;
        a = NA8 + r / gy;
        by = r - (r / gy) * gy;
    }
    bz = a / gx;
    bx = a - bz * gx;
}
__device__ __forceinline__ void dz_tile_map(int agroup, int& bx, int& by, int& bz) {
    const int gx = gridDim.x, gy = gridDim.y;
    dz_tile_map_lin(blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z), gx, gy, gridDim.z, agroup, bx, by, bz);
}

// Persistent kernels that give workgroup L the contiguous range [L * total / grid, (L + 1) * total / grid) of a
// (chunk, tile) list: hardware workgroup b runs on XCD b % 8, so with L = b eight NEIGHBOURING ranges — the tiles of
// one chunk — sit on eight different XCDs and everything they share (the chunk's normalisation partials, the halo
// rows between two ranges) is fetched into eight L2s (TCC counters, tools/tcc_probe.sh: conv_pool_h<80> reads
// exactly its input alone and 29 MB more in the pipeline = 64 chunks x 8 XCDs x 54 KB of partials).  With this map
// XCD x owns the contiguous eighth [x * grid / 8, (x + 1) * grid / 8) of the ranges.
__device__ __forceinline__ int dz_xcd_contiguous(int b, int grid) {
    return (grid & 7) == 0 ? (b & 7) * (grid >> 3) + (b >> 3) : b;
}

// ---------------------------------------------------------------------------
// "kb-major" f16 planes: the operand format of k_gemm_pre.hip and k_mlp_head.hip (activations written by
// the producing kernel's epilogue, weights packed by weights.py kb_major()).  A plane of R rows x K columns
// (K % 32 == 0) is stored as K / 32 blocks of [R][32]: the 32-wide k-tile of 16 consecutive rows — what
// one LDS-DMA instruction moves — is 1 KiB of contiguous memory.  The lo plane follows the hi plane
// R * K elements further, as before.  Element (row, col) of a plane of R rows:
// ---------------------------------------------------------------------------
__host__ __device__ __forceinline__ long long dz_kb(long long row, int col, long long R) {
    return ((long long)(col >> 5) * R + row) * 32 + (col & 31);
}

// ---------------------------------------------------------------------------
// Epilogue helper of the split-f16 GEMMs: write an f32 result as the two f16 planes the next
// layer's k_gemm_pre.hip reads (hi = f16(v), lo = f16((v - hi) * 2^11), lo plane `yplane` elements
// after the hi plane).  Lanes 2i and 2i+1 of a quad hold columns n and n+1 of the same row (MFMA C/D
// layout: column = lane & 31): they exchange their (hi, lo) pair through one DPP move, the even
// lane stores (hi[n], hi[n+1]) to the hi plane and the odd lane (lo[n-1], lo[n]) to the lo plane —
// one 4-byte store per lane, like an f32 store.  Every lane of the quad must call it; `ypl` is
// dz_split_base(Yhi, yplane, odd); `idx` = dz_kb(row, column, rows of the plane) (an even / odd column pair
// shares a k-block row); inputs are clamped to +-65504.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned short* dz_split_base(void* ysplit, long long yplane, bool odd) {
    unsigned short* y = reinterpret_cast<unsigned short*>(ysplit);
    return odd ? y + (yplane - 1) : y;
}
// `amax` accumulates max |v| of what this lane wrote (dz_flag_range reports values beyond +-65504).
__device__ __forceinline__ void dz_store_split(unsigned short* ypl, long long idx, float v, bool store,
                                               bool odd, float& amax) {
    amax = fmaxf(amax, fabsf(v));
    const float x = __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
    const _Float16 h = (_Float16)x;
    const _Float16 lw = (_Float16)((x - (float)h) * 2048.f);
    const unsigned P = (unsigned)__builtin_bit_cast(unsigned short, h) |
                       ((unsigned)__builtin_bit_cast(unsigned short, lw) << 16);
    const unsigned Q = (unsigned)__builtin_amdgcn_update_dpp(0, (int)P, 0xB1, 0xF, 0xF, true);
    const unsigned word = odd ? ((Q >> 16) | (P & 0xffff0000u)) : ((P & 0xffffu) | (Q << 16));
    if (store) *reinterpret_cast<unsigned*>(ypl + idx) = word;
}

// InstanceNorm1d(C, affine) scale / shift of chunk b from the producer's tile partials — the
// arithmetic of finalize_norm_kernel (k_front.hip: f64, tiles in fixed order, biased variance,
// eps 1e-5), done by the CONSUMER's first C threads so that no launch sits between producer and
// consumer.  out: scale[C] | shift[C].
__device__ __forceinline__ void dz_norm_from_partials(const float* __restrict__ partials, int b, int ntile,
                                                      int C, int T, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* out, int tid,
                                                      int nthreads, double* scratch) {
    // Round 2 let thread c walk the ntile partials of channel c one load at a time: up to 84 dependent
    // L2 round trips (conv1 after conv0: ~25 us in front of a workgroup's first tile).  Now the threads
    // form G = nthreads / C groups, group g takes tiles g, g + G, ... with four loads in flight, and
    // the group sums (f64: exact for these magnitudes, so the order does not show) meet in `scratch`
    // (LDS, >= 2 G C doubles).  Contains a __syncthreads(): call from ALL threads of the workgroup.
    const int G = nthreads / C > 0 ? nthreads / C : 1;
    const int c = tid % C, g = tid / C;
    if (g < G) {
        double s = 0.0, ss = 0.0;
        const float2* pp = reinterpret_cast<const float2*>(partials) + ((long long)b * ntile * C + c);
        int t = g;
        for (; t + 3 * G < ntile; t += 4 * G) {
            const float2 a0 = pp[(long long)t * C], a1 = pp[(long long)(t + G) * C];
            const float2 a2 = pp[(long long)(t + 2 * G) * C], a3 = pp[(long long)(t + 3 * G) * C];
            s += (double)a0.x; ss += (double)a0.y;
            s += (double)a1.x; ss += (double)a1.y;
            s += (double)a2.x; ss += (double)a2.y;
            s += (double)a3.x; ss += (double)a3.y;
        }
        for (; t < ntile; t += G) {
            const float2 a = pp[(long long)t * C];
            s += (double)a.x; ss += (double)a.y;
        }
        scratch[(g * C + c) * 2] = s;
        scratch[(g * C + c) * 2 + 1] = ss;
    }
    __syncthreads();
    if (tid < C) {
        double s = 0.0, ss = 0.0;
        for (int k = 0; k < G; ++k) {
            s += scratch[(k * C + tid) * 2];
            ss += scratch[(k * C + tid) * 2 + 1];
        }
        const double mean = s / T;
        double var = ss / T - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + 1e-5);
        const double sc = (double)gamma[tid] * rstd;
        out[tid] = (float)sc;
        out[C + tid] = (float)((double)beta[tid] - mean * sc);
    }
}

// The split-f16 representation holds |x| <= 65504; larger operands are clamped — and REPORTED: a lane
// that saw one stores 1 into the context's flag (rare store; dz_range_check turns it into an error).
__device__ __forceinline__ void dz_flag_range(int* oflag, float amax) {
    if (oflag && amax > 65504.f) *oflag = 1;
}

// ---------------------------------------------------------------------------
// error plumbing (api.hip)
// ---------------------------------------------------------------------------
void dz_set_error(const char* fmt, ...);
#define DZ_HIP(expr)                                                                     \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            dz_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                         __LINE__);                                                      \
            return 1;                                                                    \
        }                                                                                \
    } while (0)
#define DZ_REQUIRE(cond, ...)           \
    do {                                \
        if (!(cond)) {                  \
            dz_set_error(__VA_ARGS__);  \
            return 2;                   \
        }                               \
    } while (0)

// ---------------------------------------------------------------------------
// kernel launch wrappers (one per .hip file)
// ---------------------------------------------------------------------------
// k_front.hip ---------------------------------------------------------------
// wave statistics in two steps: slice moments, then (mean, rstd) merged by the consumer
#define DZ_WS_G 8   /* slices per chunk in wave_stats */
// slice moments of the raw waveform -> mom[B][DZ_WS_G][2] (mean_i, M2_i)
// A window with a NaN / Inf sample (or slice sums beyond f32) has a non-finite slice mean.  PyTorch's InstanceNorm1d then
// makes the whole window NaN and every output of that row is NaN (the reference drops such a chunk's speakers,
// /root/reference/src/diart/blocks/clustering.py:137-145).  The exact-f32 kernels propagate NaN by themselves; the
// split-f16 kernels clamp their operands to +-65504 (`v_med3_f32` turns NaN into a finite value: that is what keeps one
// row's garbage out of its neighbours' tiles), so their LAST kernels ask this and write NaN rows themselves.
__device__ __forceinline__ bool dz_ws_bad(const float* __restrict__ mom, int b) {
    const unsigned* m = reinterpret_cast<const unsigned*>(mom) + (long long)b * 2 * DZ_WS_G;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < DZ_WS_G; ++i) bad |= (m[2 * i] & 0x7f800000u) == 0x7f800000u;      // mean_i is Inf or NaN
    return bad;
}
#ifdef DZ_EXPERIMENTS
extern long long* dz_conv_pool_dbg;
#endif
int dz_launch_wave_stats(const float* wave, long long stride, int B, int S, float* mom,
                         hipStream_t st);
// slice moments -> stats[B][2] = (mean, rstd)
int dz_launch_wave_stats_combine(const float* mom, int B, int S, float* stats, hipStream_t st);
// InstanceNorm(1)+sinc conv(80x251, stride 10)+abs+maxpool3 -> y0[B][P0][80], partials;
// stats = (mean, rstd) per chunk, or wave_stats' slice moments when stats_are_moments
int dz_launch_sinc_conv0(const float* wave, long long stride, int B, int S, const float* stats,
                         int stats_are_moments, float gamma, float beta, const float* filt,
                         float* y0, int P0, float* partials, int ntile, hipStream_t st);
// the same on the f16 matrix cores with split operands; filt_split = f16 planes [2][96][256] of the
// unfolded bank; partials tiles are 96 frames: ntile = dz_conv0_split_ntile(F0)
int dz_launch_sinc_conv0_split(const float* wave, long long stride, int B, int S, const float* stats,
                               int stats_are_moments, float gamma, float beta, const void* filt_split,
                               float* y0, int P0, float* partials, int ntile, hipStream_t st, const void* ffrag = nullptr);
// the bank in sinc_conv0_v2's fragment order (dz_sinc_bank_frag_bytes() bytes): one tiny launch per weight set
int dz_sinc_bank_frag_bytes();
int dz_launch_sinc_bank_frag(const void* fsp, void* frag, hipStream_t st);
int dz_conv0_split_ntile(int F0);
// the same stage of BOTH networks in one launch (160 filters, one split of the normalised samples; k_front.hip;
// experiments build only)
int dz_launch_sinc_conv0_pair(const float* wave, long long stride, int B, int S, const float* moments,
                              const void* pair_planes, const float* pair_bsum, float gamma_seg, float gamma_emb,
                              float* y0_seg, float* y0_emb, int P0, float* part_seg, float* part_emb, int ntile,
                              hipStream_t st);
// partial (sum,sumsq) -> per (b,c) scale/shift of InstanceNorm1d(C, affine)
int dz_launch_finalize_norm(const float* partials, int B, int ntile, int C, int T,
                            const float* gamma, const float* beta, float* scale, float* shift,
                            hipStream_t st);

// k_convgemm.hip ------------------------------------------------------------
typedef dz_convgemm_desc DzConvGemm;
int dz_launch_convgemm(const DzConvGemm& p, hipStream_t st);
// k_gemm_f32.hip: the exact-f32 kernel of the wide layers (dz_launch_convgemm routes to it when dz_gemm_f32_ok)
bool dz_gemm_f32_ok(const DzConvGemm& p);
int dz_launch_gemm_f32(const DzConvGemm& p, hipStream_t st);
// k_gemm_split.hip: the same contraction on the f16 matrix cores with both operands split into
// (hi, lo) f16 pairs — 3 MFMAs per product, f32 accumulation
int dz_launch_gemm_split(const DzConvGemm& p, hipStream_t st);
int dz_launch_norm_f32(const float* y, const float* part, int ntile, int P, const float* gamma, const float* beta,
                       float* out, int B, hipStream_t st);
int dz_launch_norm_split(const float* y, const float* part, int ntile, int P, const float* gamma, const float* beta,
                         void* planes, long long plane, int B, hipStream_t st);
// k_conv_pool.hip: SincNet stages 1 / 2 (k = 5 conv + MaxPool1d(3) + partials) with the input tile
// resident in LDS and the weights in registers; descriptor as the POOL3 call of dz_launch_gemm_split
// wfrag: p.Wsplit in conv_pool_h's fragment order (dz_launch_conv_pool_wfrag, dz_conv_pool_wfrag_bytes(Cin) bytes), or NULL
int dz_launch_conv_pool(const DzConvGemm& p, hipStream_t st, const void* wfrag = nullptr);
int dz_conv_pool_wfrag_bytes(int Cin);
int dz_launch_conv_pool_wfrag(int Cin, const void* wsplit, int Kpad, void* out, hipStream_t st);
// k_gemm_pre.hip: both operands pre-split into f16 planes, tiles loaded by LDS-DMA
int dz_launch_gemm_pre(const DzConvGemm& p, hipStream_t st);
// k_gemm_g2.hip: generation 2 of the same layer (single accumulator, three LDS stages, counted vmcnt);
// mt = row fragments per wave (2, 3, 4 -> 128 / 192 / 256 x 128 tiles), 0 = DZ_G2_MT / default
#ifdef DZ_EXPERIMENTS
int dz_launch_gemm_g2(const DzConvGemm& p, int mt, hipStream_t st);
int dz_g2_default_mt();
#endif
int dz_gemm_gen();       // 1 in the shipped build
// k_gemm_g3.hip: generation 2's loop as a persistent kernel over a balanced (Stream-K) split of the iteration
// space, one workgroup per CU; mt as above (0 = DZ_G3_MT / default 4).  dz_g3_error: a wait timed out.
#ifdef DZ_EXPERIMENTS
int dz_launch_gemm_g3(const DzConvGemm& p, int mt, hipStream_t st);
int dz_g3_error(int reset);
#endif
// The same launch with the weighted statistics pooling (paper Eq. 1) fused into the epilogue of the LAST
// x-vector layer (tdnn5): the 128 x 128 output tile is parked in LDS instead of HBM and reduced there
// to per-(tile, chunk, speaker, channel) weighted means and centred second moments — exact two-pass
// statistics of the tile's rows; dz_launch_pool_combine merges the tile pieces of a chunk
// (Chan et al.).  Rows are the flattened frames of the batch: row r = chunk r / P, frame r % P, frames
// >= T of a chunk carry no weight.
// Pooling weight of output frame t of T from a row of |Fw| weights — pyannote's StatsPool resamples the (N, Fw) weights
// to the T frames of the features: Fw > 0: F.interpolate(mode="linear", align_corners=False), the form
// of pyannote.audio 2.x .. 3.0; Fw < 0 (the sign is how the internal launchers carry the mode): mode="nearest",
// pyannote.audio >= 3.1 — source index floor(t * (|Fw| / T)) computed in f32 like PyTorch's nearest kernel.
__device__ __forceinline__ float dz_pool_weight(const float* wr, int Fw, int T, int t) {
    const int F = Fw < 0 ? -Fw : Fw;
    if (F == T) return wr[t];
    const float scale = (float)F / (float)T;
    if (Fw < 0) {
        int i = (int)floorf((float)t * scale);
        if (i > F - 1) i = F - 1;
        return wr[i];
    }
    float src = scale * ((float)t + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    const int i0 = (int)src;
    const int i1 = i0 + (i0 < F - 1 ? 1 : 0);
    const float l1 = src - (float)i0;
    return (1.f - l1) * wr[i0] + l1 * wr[i1];
}

struct DzPoolFuse {
    const float* w;     // [nx * K][Fw] pooling weights (speaker-major per chunk) or NULL (all ones)
    int Fw, K;          // weight frames per row (resampled to T when |Fw| != T; negative: nearest, see dz_pool_weight); speakers per chunk, <= 4
    int P, T;           // row pitch per chunk / valid frames per chunk
    int np;             // piece slots per chunk = dz_pool_pieces(P): 128-row tiles a chunk can touch
    float* part;        // [nx][np][K][Npad][2] = (mean, M2) of the piece
    float* s0;          // [nx][np][K][2] = (sum w, sum w^2) of the piece
};
inline int dz_pool_pieces(int P) { return (P + 2 * 128 - 2) / 128; }
bool dz_gemm_pre_pool_ok(const DzConvGemm& p);     // big tiles throughout? (else: unfused path)
int dz_launch_gemm_pre_pool(const DzConvGemm& p, const DzPoolFuse& q, hipStream_t st);
// part / s0 as above -> out [nx * K][ldo] = mean (columns 0 .. C-1) | std (columns C .. 2C-1)
int dz_launch_pool_combine(const float* part, const float* s0, int nx, int K, int np, int P, int T, int C,
                           int Npad, float* out, int ldo, hipStream_t st);
int dz_convgemm_ntile(int Tout);

// k_lstm.hip ----------------------------------------------------------------
// gx [B*T][1024] (biases included; columns dir*512+gate*128+unit, or dir*512+unit*4+gate when
// unit_major), whh [2][512][128] -> hout [B][T][256] (fwd | bwd).  One chain per workgroup, f32 VALU.
// Output: hout (f32) and / or hsplit = two f16 planes [B][T][256] (hi, lo * 2^11; lo plane hplane
// elements after hi) for a k_gemm_pre.hip consumer; either may be NULL.
int dz_launch_lstm(const float* gx, const float* whh, float* hout, void* hsplit, long long hplane,
                   int B, int T, int unit_major, hipStream_t st);
// k_lstm_mfma.hip: the same recurrence, 16 chains per workgroup on the f16 matrix cores with split
// operands; whh_split = [2 dir][2 planes (hi, lo * 2^11)][512][128] f16 (weights.py split_f16 per direction)
int dz_launch_lstm_mfma(const float* gx, const void* whh_split, float* hout, void* hsplit,
                        long long hplane, int B, int T, int unit_major, int variant, hipStream_t st);

// k_pool.hip ----------------------------------------------------------------
// weighted statistics pooling; X: nx chunks `xstride` floats apart, each [T][ldx] (C valid
// channels), weights [rows][Fw] or null, row r pools chunk r / rows_per_x;
// out [rows][ldo] = mean | std (std at column C)
int dz_launch_stats_pool(const float* X, long long xstride, int T, int C, int ldx,
                         const float* weights, int Fw, int rows, int rows_per_x, float* out, int ldo,
                         hipStream_t st);
int dz_launch_osp(const float* seg, int B, int F, int K, float gamma, float beta, int normalize,
                  int speaker_major, float* out, hipStream_t st);
// ---------------------------------------------------------------------------
// Per-frame tail of the segmentation network, shared by seg_head_kernel (k_pool.hip) and the fused
// MLP + head kernel (k_mlp_head.hip) so that both produce the same bits.
//   dz_seg_decide: class logits -> per-speaker activity: sigmoid (multilabel models) or the hard
//     powerset decision argmax -> multilabel (models.py:29-39; log_softmax is monotone, so the argmax
//     of the logits is the argmax of the log-probabilities; subsets ordered by size, then
//     lexicographically, at most two speakers per frame).
//   dz_osp_frame: OverlappedSpeechPenalty of one frame (functional.py:6-13), before the optional
//     min-max normalisation.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float dz_powg(float x, float gamma) {
    // torch.pow(tensor, scalar) special-cases 2 and 3 as repeated products
    if (gamma == 3.f) return (x * x) * x;
    if (gamma == 2.f) return x * x;
    if (gamma == 1.f) return x;
    return powf(x, gamma);
}
__device__ __forceinline__ void dz_seg_decide(const float* lg, int classes, int K, int powerset, float* s) {
    if (powerset) {
        int best = 0;
        float bv = lg[0];
        for (int c = 1; c < classes; ++c)
            if (lg[c] > bv) {
                bv = lg[c];
                best = c;
            }
        int a = -1, b2 = -1;
        if (best >= 1 && best <= K) {
            a = best - 1;
        } else if (best > K) {
            int idx = best - K - 1;
            for (int i = 0; i < K && a < 0; ++i) {
                const int cnt = K - 1 - i;
                if (idx < cnt) {
                    a = i;
                    b2 = i + 1 + idx;
                } else {
                    idx -= cnt;
                }
            }
        }
        for (int k = 0; k < K; ++k) s[k] = (k == a || k == b2) ? 1.f : 0.f;
    } else {
        for (int k = 0; k < K; ++k) s[k] = 1.f / (1.f + expf(-lg[k]));
    }
}
__device__ __forceinline__ void dz_osp_frame(const float* s, int K, float gamma, float beta, float* w) {
    float e[8], m = -INFINITY;
    for (int k = 0; k < K; ++k) m = fmaxf(m, beta * s[k]);
    float sum = 0.f;
    for (int k = 0; k < K; ++k) {
        e[k] = expf(beta * s[k] - m);
        sum += e[k];
    }
    for (int k = 0; k < K; ++k) {
        const float pr = e[k] / sum;
        float wv = dz_powg(s[k], gamma) * dz_powg(pr, gamma);
        if (wv < 1e-8f) wv = 1e-8f;
        w[k] = wv;
    }
}

int dz_launch_seg_head(const float* m1, const float* cw, const float* cb, int B, int F, int classes, int K,
                       int powerset, float* seg, float gamma, float beta, int normalize, float* wout,
                       hipStream_t st, const float* wave_mom = nullptr);
// lin0 (256 -> 128, LeakyReLU) -> lin1 (128 -> 128, LeakyReLU) -> classifier -> activation (-> OSP
// weights, not normalised) in one launch: k_mlp_head.hip
struct DzMlpHead {
    const void* Xsplit;        // [2][rows][256] f16 planes of the last LSTM layer's output
    long long xplane;
    const void *W0split, *W1split;       // [2][128][256], [2][128][128] f16 planes
    const float *b0, *b1, *cw, *cb;      // biases; classifier [>= classes][128], [classes]
    int rows, F, classes, K, powerset;
    float gamma, beta;
    float* seg;                // [rows][K]
    float* wout;               // [rows / F][K][F] or NULL
    int* oflag;
    const float* wave_mom;     // wave_stats moments of the rows' chunks, or NULL: chunks with non-finite ones get NaN rows (dz_ws_bad)
};
int dz_launch_mlp_head(const DzMlpHead& p, hipStream_t st);
int dz_launch_l2norm(float* x, int rows, int dim, float norm, hipStream_t st);
// wave_mom (or NULL): wave_stats moments of the rows' chunks (row r belongs to chunk r / rows_per_x): NaN rows for dz_ws_bad chunks
int dz_launch_splitk_finish(const float* parts, int nsplit, long long stride, int rows, int dim,
                            int normalize, float* out, hipStream_t st, const float* wave_mom = nullptr,
                            int rows_per_x = 1);
int dz_launch_powerset(const float* logp, int rows, int classes, int speakers, float* out,
                       hipStream_t st);
int dz_launch_cdist(const float* emb, const double* centers, int n, int k, int g, int dim,
                    double* out, hipStream_t st);

// k_ecapa.hip ---------------------------------------------------------------
int dz_launch_mask_compact(const float* wave, long long stride, int S, const float* masks, int Fw,
                           int rows, float* sig, long long sig_stride, int* lens, hipStream_t st);
int dz_launch_power(const float* spec, int lds, long long rows, float* pw, hipStream_t st);
int dz_launch_fbank_post(const float* melp, int T, int rows, const int* nvalid, float* feats,
                         hipStream_t st);
int dz_launch_se_mean(const float* x, int T, int C, int ldx, int rows, const int* nmask, float* s,
                      hipStream_t st);
int dz_launch_se_apply(const float* x, int ldx, const float* gate, const float* resid, int ldr,
                       float* out, int ldo, int rows, int T, int C, hipStream_t st);
int dz_launch_se_apply_planes(const float* x, int ldx, const float* gate, const float* resid, int ldr, float* out,
                              int ldo, void* planes, long long plane, int rows, int T, int C, hipStream_t st);
int dz_launch_asp_gstats(const float* x, int T, int C, int rows, const int* nmask, float* g,
                         hipStream_t st);
int dz_launch_asp_pool(const float* x, const float* logit, int T, int C, int rows, const int* nmask,
                       float* pooled, hipStream_t st);
int dz_launch_nan_rows(float* out, int rows, int dim, const int* flags, hipStream_t st);

struct dz_ctx {
    int device;
    // "an operand left the f16 range" flag of the split-f16 kernels: one int in pinned, device-mapped
    // host memory (kernels store 1 into it; the host reads it without a copy: dz_range_check)
    int* oflag_host;
    int* oflag_dev;
    void* conv0_frag;    // dz_k_sinc_conv0_split: scratch for the bank in fragment order (lazily allocated)
    void* convp_frag;    // dz_k_conv_pool: the same for its weights
    const void* conv0_src;                 // option "pack_cache": what the two scratch buffers were packed from
    const void* convp_src;
    int convp_cin, convp_kpad;
    // the two scratch buffers are shared by every caller of the context: one lock around "repack + launch", and a
    // repack first waits for the stream whose kernel may still be reading the scratch
    std::mutex frag_mu;
    hipStream_t conv0_user, convp_user;
    bool conv0_used, convp_used;
};
