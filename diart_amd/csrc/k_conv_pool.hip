// SincNet stages 1 and 2 on the f16 matrix cores:  InstanceNorm + LeakyReLU of the input (on load)
// -> Conv1d(Cin -> 60, k = 5) + bias -> MaxPool1d(3) -> pooled rows + the (sum, sumsq) partials of
// the next InstanceNorm.  (pyannote.audio SincNet.forward, third party, called from
// /root/reference/src/diart/models.py:133 and :262; SURVEY.md Appendix A.1, kernels K3 / K4.)
//
// The general split-f16 GEMM kernel (k_gemm_split.hip, POOL3 epilogue) treats these layers as an
// implicit GEMM with K = 5 taps x Cin: every k-tile re-loads its 32-wide slice of the input rows from
// L2, re-applies the normalisation and re-splits it — five times per input element, with only 64
// output columns to amortise it over (13 % of the split-f16 matrix peak, 80 + 28 us per network).
// Here the convolution keeps its structure:
//   * the tile's input rows (96 conv frames + 4 = 100 rows x Cin channels) are normalised, split
//     into (hi, lo * 2^11) f16 planes and parked in LDS ONCE; the five taps are five row offsets
//     into that image (row pitch Cin * 2 + 16 bytes: an odd multiple of 16, which makes the 16
//     lanes of every fragment read hit 16 distinct bank groups);
//   * the weights (64 x 5 Cin, two planes) stay in REGISTERS for the whole persistent workgroup:
//     wave (nt, kh) owns output channels 32 nt .. +31 and one half of the k-steps;
//   * MaxPool1d(3) is an element-wise maximum: the 96 frames are computed as three 32-row MFMA
//     blocks, block b = frames {3 m + b}, so pooled row m sits in the same lane and register of
//     all three (the trick of the sinc_conv0 kernels, k_front.hip);
//   * the two k-halves meet once per tile through LDS: each hands the other the half of its accumulator rows it
//     does not finish, so all four waves add, pool, store and sum (round 5; before, half 1 handed over everything
//     and idled through the finishing pass).
// Workgroup = 4 waves, two workgroups per CU (the size that fits beside a recurrence workgroup); partials layout and
// tile size (96 conv frames) are those of the POOL3 GEMM epilogue, so finalize_norm and every consumer are unchanged.
// Round 5 also trimmed the vector work of the tile loop — buffer loads (no address arithmetic), packed-f32 parking —
// after conv_pool_v2 (below, experiments build) showed that vector instructions are not hidden under MFMAs.
#include "dz_common.h"
#include <type_traits>

#ifdef DZ_EXPERIMENTS
extern long long* dz_conv_pool_dbg;      // set by dz_k_conv_pool_debug (phase stamps, tools/conv_pool_phases.py)
#endif

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned cp_u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int FR = 96;              // conv frames per tile (= 32 pooled rows)
constexpr int ROWS = FR + 4;        // input rows a tile touches (k = 5)

template <int CIN>
struct Geo {
    static constexpr int KS = 5 * CIN / 16;            // 16-wide k-steps (k = tap * CIN + c)
    static constexpr int KS0 = (KS + 1) / 2;           // k-steps of half 0 (half 1: KS - KS0)
    static constexpr int PITCH = CIN * 2 + 16;         // bytes per LDS row of one plane
    static constexpr int PLANE = ROWS * PITCH;
    static constexpr int XCH = 2 * 3 * 16 * 64 * 4;    // exchange: [nt][block][reg][lane] f32
    static constexpr int NORM = 2 * CIN * 4;           // scale | shift of the current chunk
    static constexpr int LDS = 2 * PLANE + XCH + NORM + 2 * 64 * 2 * 4;   // + the k-halves' (sum, sumsq) of a tile
    static_assert(CIN % 16 == 0 && (PITCH / 16) % 2 == 1, "row pitch must be an odd multiple of 16 bytes");
};

__device__ __forceinline__ float leaky(float v) { return v > 0.f ? v : v * DZ_LEAKY_SLOPE; }

template <int CIN>
__global__ __launch_bounds__(256, 2) void conv_pool_h_kernel(
    const float* __restrict__ X, int Tin, int Tout, int Tstore, const float* __restrict__ nscale,
    const float* __restrict__ nshift, const float* __restrict__ npart, int npart_tiles, int npart_T,
    const float* __restrict__ ngamma, const float* __restrict__ nbeta,
    const unsigned short* __restrict__ wsp, int Kpad,
    const float* __restrict__ bias, float* __restrict__ Y, float* __restrict__ partials, int ntile,
    int total, int* __restrict__ oflag, long long* __restrict__ dbg, const unsigned short* __restrict__ wfrag) {
    using G = Geo<CIN>;
    // dbg (kbench only): shader-clock stamps of the phases of every tile, wave 0 / wave 3 lane 0 of each workgroup
#ifdef DZ_EXPERIMENTS       // phase stamps for tools/conv_pool_phases.py: experiments build only
    long long* dq = nullptr;
    if (dbg && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == 3))
        dq = dbg + ((long long)blockIdx.x * 2 + (threadIdx.x >> 7)) * 64;
    int dn = 0;
    if (dq) dq[62] = __builtin_readcyclecounter();            // kernel entry
#define DZ_STAMP() do { if (dq && dn < 62) dq[dn++] = __builtin_readcyclecounter(); } while (0)
#else
#define DZ_STAMP() do { } while (0)
#endif
    extern __shared__ __attribute__((aligned(256))) char lds[];
    char* xs = lds;                                                  // [2 planes][ROWS][PITCH]
    float* xch = reinterpret_cast<float*>(lds + 2 * G::PLANE);       // [2][3][16][64]
    float* nrm = reinterpret_cast<float*>(lds + 2 * G::PLANE + G::XCH);   // scale[CIN] | shift[CIN]
    float* psum = reinterpret_cast<float*>(lds + 2 * G::PLANE + G::XCH + G::NORM);   // [k-half][64 ch][sum, sumsq] of a tile
    long long ptile = -1;                                             // tile whose sums are in psum
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, li = l & 31, g = l >> 5;
    const int nt = w & 1, kh = w >> 1;                                // channel block, k-half
    const int ks_lo = kh ? G::KS0 : 0, ks_n = kh ? G::KS - G::KS0 : G::KS0;
    const int wg = dz_xcd_contiguous(blockIdx.x, gridDim.x);         // neighbouring tile ranges on ONE XCD
    const int t_begin = (int)((long long)wg * total / gridDim.x);
    const int t_end = (int)((long long)(wg + 1) * total / gridDim.x);
    if (t_begin >= t_end) return;

    // ---- weight fragments of this wave: channels 32 nt + li, k = 16 ks + 8 g .. +7 -------------
    f16x8 bh[G::KS0], bl[G::KS0];
    if (wfrag) {
        // the weights in FRAGMENT order (conv_pool_wfrag_kernel below; repacked once per handle): every load
        // instruction of the wave reads 1 KiB of contiguous memory.  With the row-major planes each lane reads 16
        // bytes of its own row — 32 cache lines per instruction — and a workgroup that lives for 1 - 4 tiles spent
        // 7 - 9 k cycles (16 - 28 % of its life, tools/conv_pool_phases.py) getting its weights.
        const f16x8* fr = reinterpret_cast<const f16x8*>(wfrag) + (long long)((nt * 2 + kh) * 2) * G::KS0 * 64 + l;
#pragma unroll
        for (int i = 0; i < G::KS0; ++i) {
            bh[i] = fr[i * 64];
            bl[i] = fr[(G::KS0 + i) * 64];
        }
    } else {
        const unsigned short* row = wsp + (long long)(32 * nt + li) * Kpad + 8 * g;
#pragma unroll
        for (int i = 0; i < G::KS0; ++i) {
            const int ks = ks_lo + (i < ks_n ? i : 0);
            bh[i] = *reinterpret_cast<const f16x8*>(row + 16 * ks);
            bl[i] = *reinterpret_cast<const f16x8*>(row + 64 * (long long)Kpad + 16 * ks);
        }
    }
    // A fragment offsets: block b, k-step ks -> tap = ks / (CIN/16), j = ks % (CIN/16):
    // byte = (3 li + b + tap) * PITCH + (16 j + 8 g) * 2
    const int abase = 3 * li * G::PITCH + 16 * g;
    const float bv = bias[32 * nt + li];
    float amax = 0.f;
    int cur_b = -1;

    // The tile's raw input rows travel global -> registers as ONE batch of loads and are then consumed
    // (normalised, split, parked in LDS).  Round 2 loaded, waited and parked one float4 per lane at a time — 8 dependent L2 / HBM
    // round trips per tile with nothing else for the wave to do: 30 k cycles per tile for 3.7 k of MFMAs.
    constexpr int C4 = CIN / 4;
    constexpr int NPK = (ROWS * C4 + 255) / 256;
    f32x4 pv[NPK];
    // Buffer loads: rows are dense (ldx = CIN), so the tile is ONE span of memory and piece idx = tid + 256 i sits
    // at byte 16 idx of it — no address arithmetic, no branches; what lies past the chunk's end reads as zeros
    // through the bounds check (pieces past the tile's own rows are loaded and ignored).
    auto fetch = [&](int t) {
        const int b = t / ntile, t0 = (t - b * ntile) * FR;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(X + ((long long)b * Tin + t0) * CIN), 0, (unsigned)(Tin - t0) * (CIN * 4u), 0x00020000);
#pragma unroll
        for (int i = 0; i < NPK; ++i)
            pv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, 16 * tid + 4096 * i, 0, 0));
    };

    for (int t = t_begin; t < t_end; ++t) {
        // (all of the tile's loads in flight at once; issued here, ahead of the norm update and the
        // barrier.  Issued a phase earlier — before or after the previous tile's MFMAs — the 32 registers
        // would be live next to the accumulators / the pooled blocks: the kernel is at 240 - 256 VGPRs
        // and spills)
        DZ_STAMP();
        fetch(t);
        const int b = t / ntile, tile = t - b * ntile;
        const int t0 = tile * FR;
        if (b != cur_b) {                       // scale / shift of this chunk's InstanceNorm
            __syncthreads();                    // (nobody still reads the previous chunk's)
            if (npart)       // straight from the producer's tile partials: no finalize launch
                dz_norm_from_partials(npart, b, npart_tiles, CIN, npart_T, ngamma, nbeta, nrm, tid, 256,
                                      reinterpret_cast<double*>(xch));   // (the exchange area is idle between tiles)
            else
                for (int i = tid; i < 2 * CIN; i += 256)
                    nrm[i] = i < CIN ? nscale[(long long)b * CIN + i] : nshift[(long long)b * CIN + i - CIN];
            cur_b = b;
        }
        __syncthreads();                        // previous tile's fragment / exchange reads are done
        // (and both k-halves' sums of the previous tile are in LDS: k-half 0's + k-half 1's, tid = 2 ch + {sum, sumsq})
        if (ptile >= 0 && tid < 128) partials[ptile * 128 + tid] = psum[tid] + psum[128 + tid];
        DZ_STAMP();
        // ---- park the tile: normalise + LeakyReLU, split, two planes --------------------------
        int tid_p = tid;
        asm volatile("" : "+v"(tid_p));
#pragma unroll
        for (int i = 0; i < NPK; ++i) {
            const int idx = tid_p + 256 * i;
            if (idx < ROWS * C4) {
                const int r = idx / C4, c4 = idx - r * C4;
                // Packed f32 arithmetic where the ISA has it (v_pk_fma / v_pk_mul: two elements per instruction),
                // LeakyReLU as max(x, slope x) (slope in (0, 1)), rows past the chunk's end zeroed by a packed
                // multiply: ~33 vector instructions per 4 elements instead of ~45.  A SIMD does not overlap these
                // with the MFMAs of the other resident workgroup's wave (conv_pool_v2 below: measured), so every
                // vector instruction of the tile loop is tile time.
                const f32x4 sc = *reinterpret_cast<const f32x4*>(nrm + 4 * c4);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(nrm + CIN + 4 * c4);
                const f32x2 in01 = t0 + r < Tin ? (f32x2){1.f, 1.f} : (f32x2){0.f, 0.f};
                f32x2 va = {pv[i][0], pv[i][1]}, vb = {pv[i][2], pv[i][3]};
                va = __builtin_elementwise_fma(va, (f32x2){sc[0], sc[1]}, (f32x2){sh[0], sh[1]});
                vb = __builtin_elementwise_fma(vb, (f32x2){sc[2], sc[3]}, (f32x2){sh[2], sh[3]});
                const f32x2 sa = va * DZ_LEAKY_SLOPE, sb = vb * DZ_LEAKY_SLOPE;
                f32x4 v = {fmaxf(va[0], sa[0]), fmaxf(va[1], sa[1]), fmaxf(vb[0], sb[0]), fmaxf(vb[1], sb[1])};
                va = (f32x2){v[0], v[1]} * in01;
                vb = (f32x2){v[2], v[3]} * in01;
                amax = fmaxf(amax, fmaxf(fmaxf(fabsf(va[0]), fabsf(va[1])), fmaxf(fabsf(vb[0]), fabsf(vb[1]))));
                v = (f32x4){__builtin_amdgcn_fmed3f(va[0], -65504.f, 65504.f), __builtin_amdgcn_fmed3f(va[1], -65504.f, 65504.f),
                            __builtin_amdgcn_fmed3f(vb[0], -65504.f, 65504.f), __builtin_amdgcn_fmed3f(vb[1], -65504.f, 65504.f)};
                const f16x4 hi = __builtin_convertvector(v, f16x4);
                const f16x4 lo = __builtin_convertvector((v - __builtin_convertvector(hi, f32x4)) * 2048.f, f16x4);
                char* d = xs + r * G::PITCH + 8 * c4;
                *reinterpret_cast<f16x4*>(d) = hi;
                *reinterpret_cast<f16x4*>(d + G::PLANE) = lo;
            }
        }
        DZ_STAMP();
        __syncthreads();
        DZ_STAMP();
        // ---- three blocks (frames 3 m + bk), this wave's half of the k-steps -------------------
        f32x16 part[3];
#pragma unroll
        for (int bk = 0; bk < 3; ++bk) {
            f32x16 accm, accx;
#pragma unroll
            for (int r = 0; r < 16; ++r) accm[r] = accx[r] = 0.f;
#pragma unroll
            for (int i = 0; i < G::KS0; ++i) {
                if (i < ks_n) {
                    const int ks = ks_lo + i;                            // wave-uniform
                    const int tap = ks / (CIN / 16), j = ks - tap * (CIN / 16);
                    const char* ap = xs + abase + (bk + tap) * G::PITCH + 32 * j;
                    const f16x8 ah = *reinterpret_cast<const f16x8*>(ap);
                    const f16x8 al = *reinterpret_cast<const f16x8*>(ap + G::PLANE);
                    accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[i], accx, 0, 0, 0);
                    accm = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[i], accm, 0, 0, 0);
                    accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[i], accx, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) part[bk][r] = accm[r] + accx[r] * (1.f / 2048.f);
        }
        DZ_STAMP();
        // ---- the k-halves meet.  Each half finishes EIGHT of the sixteen accumulator rows (k-half 0: registers 0..7,
        // k-half 1: 8..15) and hands the other eight of its partial sums over: all four waves add, pool, store and
        // sum (rounds 2 - 4: half 1 handed over everything and idled while half 0 finished the tile — 3.1 k of a
        // tile's 16.6 k cycles) -----------------------------------------------------------------------------------
        // (kh is wave-uniform: two straight-line instantiations, compile-time register indices)
        auto hand_over = [&](auto KH) {
            constexpr int give = decltype(KH)::value ? 0 : 8;      // first register of the rows the OTHER half finishes
#pragma unroll
            for (int bk = 0; bk < 3; ++bk)
#pragma unroll
                for (int q = 0; q < 8; ++q) xch[((nt * 3 + bk) * 16 + give + q) * 64 + l] = part[bk][give + q];
        };
        if (kh) hand_over(std::integral_constant<int, 1>{});
        else hand_over(std::integral_constant<int, 0>{});
        __syncthreads();
        DZ_STAMP();
        auto finish = [&](auto KH) {
            constexpr int own = decltype(KH)::value ? 8 : 0;       // first register of the rows THIS half finishes
            const int ch = 32 * nt + li;
            float sum = 0.f, ssq = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float pm = 0.f;
#pragma unroll
                for (int bk = 0; bk < 3; ++bk) {
                    // (k-half 0's sum + k-half 1's sum) + bias, whichever wave adds them
                    const float v = (part[bk][own + q] + xch[((nt * 3 + bk) * 16 + own + q) * 64 + l]) + bv;
                    pm = bk == 0 ? v : fmaxf(pm, v);
                }
                // pooled rows + partials: C/D column = lane & 31 = channel, register rr -> pooled row (rr & 3) + 8 (rr >> 2) + 4 g
                constexpr int rr0 = own;
                const int pr = tile * 32 + ((rr0 + q) & 3) + 8 * ((rr0 + q) >> 2) + 4 * g;
                if (pr < Tstore) {
                    Y[((long long)b * Tstore + pr) * 64 + ch] = pm;
                    sum += pm;
                    ssq += pm * pm;
                }
            }
            sum += __shfl_xor(sum, 32, 64);
            ssq += __shfl_xor(ssq, 32, 64);
            if (g == 0) {                                  // this half's 16 rows of channel ch; the halves meet at the next barrier
                psum[(decltype(KH)::value * 64 + ch) * 2] = sum;
                psum[(decltype(KH)::value * 64 + ch) * 2 + 1] = ssq;
            }
        };
        if (kh) finish(std::integral_constant<int, 1>{});
        else finish(std::integral_constant<int, 0>{});
        ptile = (long long)b * ntile + tile;
        DZ_STAMP();
    }
    // the last tile's partials
    __syncthreads();
    if (tid < 128) partials[ptile * 128 + tid] = psum[tid] + psum[128 + tid];
#undef DZ_STAMP
    dz_flag_range(oflag, amax);
}


// [nt 2][k-half 2][plane 2][fragment i < KS0][lane 64] vectors of 8 f16: lane (li, g) of wave (nt, kh) holds the taps
// 16 ks + 8 g .. + 7 of output channel 32 nt + li, ks = ks_lo(kh) + i (fragments past a half's last k-step repeat its
// first one, like the loads they replace)
template <int CIN>
__global__ void conv_pool_wfrag_kernel(const unsigned short* __restrict__ wsp, int Kpad, f32x4* __restrict__ out) {
    using G = Geo<CIN>;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= 2 * 2 * 2 * G::KS0 * 64) return;
    const int lane = v & 63;
    int rest = v >> 6;
    const int i = rest % G::KS0;
    rest /= G::KS0;
    const int plane = rest & 1, kh = (rest >> 1) & 1, nt = rest >> 2;
    const int ks_lo = kh ? G::KS0 : 0, ks_n = kh ? G::KS - G::KS0 : G::KS0;
    const int ks = ks_lo + (i < ks_n ? i : 0);
    out[v] = *reinterpret_cast<const f32x4*>(wsp + ((long long)plane * 64 + 32 * nt + (lane & 31)) * Kpad + 16 * ks + 8 * (lane >> 5));
}

#ifdef DZ_EXPERIMENTS
// ---------------------------------------------------------------------------
// conv_pool_v2 (round 5): the same layer with the two kinds of work of a tile on DIFFERENT waves.
//
// conv_pool_h walks fetch -> park -> MFMA -> exchange -> epilogue in every wave; its phase profile
// (tools/conv_pool_phases.py, conv1, cycles per tile of 16.6 k): fetch + norm + barrier 2.4 k, park 4.0 k, MFMA
// phase 6.3 k (3.7 k of MFMA issue), k-half exchange 0.6 k, epilogue 3.1 k on two of the four waves — the matrix
// pipes of a CU are busy 45 % of the tile loop with two workgroups resident.  Here a workgroup is EIGHT waves,
// one workgroup per CU, two waves per SIMD:
//   * four MATRIX waves (channel block nt, k-half kh — the weights of conv_pool_h, still in registers) do nothing
//     but the 3 x 12 / 13 k-steps of a tile, fragment reads issued one step ahead through inline-asm ds_read_b128
//     with counted lgkmcnt waits (as in sinc_conv0_v2, k_front.hip), and drop each finished 32 x 32 block into the
//     exchange area;
//   * four SERVICE waves fetch (buffer loads: no address arithmetic, no branches), normalise, split and park the
//     NEXT tile into the other input buffer, and finish the PREVIOUS tile: k-halves added, bias, MaxPool1d(3) over
//     the three blocks, pooled rows stored as 16-byte pieces of full 256-byte rows (conv_pool_h: 4-byte stores), the
//     tile's (sum, sumsq) partials through LDS in fixed order.
// Roles go by the SIMD a wave landed on (HW_ID), one matrix wave per SIMD, so every SIMD has matrix work and
// VALU / LDS-write / memory work side by side.  One barrier per tile; the exchange area has one buffer, a service
// wave counts its reads done in LDS and a matrix wave looks at that count before its first block of the next tile
// lands there (in practice it is there long before).
// The same products as conv_pool_h, added in another (fixed) order.
//
// OUTCOME (profiles/r05l_conv_pool_v2.json): EXPERIMENTS BUILD ONLY (DZ_CONV_POOL_V2=1).  Alone on the GPU it is
// faster, 41.0 -> 37.9 us (conv1) and 21.2 -> 17.3 us (conv2) at 64 chunks; in the 64-stream pipeline the step gets
// SLOWER (same-visit A/B: 28 296 / 28 313 xRT with conv_pool_h, 27 914 / 25 444 with this kernel), for two reasons
// the phase stamps and two timing-only builds made visible:
//   * a SIMD does not overlap one wave's vector instructions with another wave's MFMAs: the service wave's 60
//     vector instructions of the finishing pass take 0.47 k cycles with the matrix wave's MFMAs compiled out and
//     3.4 k cycles beside them (s_setprio changes nothing).  A tile costs a SIMD its MFMA cycles PLUS its vector
//     cycles whichever wave issues them, so separating the roles buys only the memory latency it hides (what is
//     left of the gain above comes from the fragment sharing and the cheaper parking arithmetic);
//   * eight waves of 248 registers and 136 KB of LDS own a CU: the workgroup cannot sit beside a recurrence
//     workgroup (k_lstm.hip: 8 waves per CU on up to 128 - 256 CUs for most of a step), conv_pool_h (4 waves, 60 KB)
//     can — the rule every kernel of this pipeline is sized by.
// ---------------------------------------------------------------------------
template <int CIN>
struct Geo2 {
    using G = Geo<CIN>;
    static constexpr int IN = 2 * G::PLANE;                  // one input buffer: hi | lo planes of a tile
    static constexpr int XCH = 2 * 2 * 3 * 16 * 64 * 4;      // [kh][nt][block][reg][lane] f32
    static constexpr int RED = 2 * 16 * 64 * 2 * 4;          // [tile parity][row group][channel][sum, sumsq]
    static constexpr int OFF_XCH = 2 * IN, OFF_RED = OFF_XCH + XCH, OFF_NRM = OFF_RED + RED;
    static constexpr int OFF_MISC = OFF_NRM + G::NORM;       // simd[8] | exchange reads done
    static constexpr int LDS = OFF_MISC + 64;
    static_assert(IN % 16 == 0 && IN >= 2 * 8 * CIN * 2 * 8, "input buffer doubles as the norm scratch");
};

// The k-steps a matrix wave owns, and the order it walks them in.  Block bk of a tile is the frames {3 m + bk}, so
// the fragment of (block bk, tap, channel sub-block j) is the 32 rows {3 m + bk + tap}: every (bk, tap) with the same
// s = bk + tap reads THE SAME bytes.  conv_pool_h read them once per (bk, tap) — 15 fragment pairs per sub-block,
// 7 distinct — and with eight waves on a CU the LDS pipe, not the matrix pipe, set the pace (a ds_read_b128 of a
// wave occupies it for 8 cycles: 4 waves x 2 reads per 96 cycles of MFMA = 67 % before anything else).  Here the K
// axis is split between the two k-halves by SUB-BLOCK (n = 5 j + tap; k-half 0 owns n < KS0), a wave keeps all
// three blocks' accumulators live and walks (j, s): one fragment pair, then the MFMAs of every block it feeds.
template <int CIN, int KH>
struct Cp2Sched {
    using G = Geo<CIN>;
    static constexpr int NJ = CIN / 16;
    static constexpr int n0 = KH ? G::KS0 : 0, n1 = KH ? G::KS : G::KS0;      // owned n = 5 j + tap
    static constexpr int NW = n1 - n0;                                         // weight fragments held (<= KS0)
    static constexpr int j_lo = n0 / 5, j_hi = (n1 - 1) / 5;
    static constexpr int ta(int j) { return n0 - 5 * j > 0 ? n0 - 5 * j : 0; }
    static constexpr int tb(int j) { return n1 - 1 - 5 * j < 4 ? n1 - 1 - 5 * j : 4; }
    static constexpr int nloads() {
        int c = 0;
        for (int j = j_lo; j <= j_hi; ++j) c += tb(j) - ta(j) + 3;
        return c;
    }
    static constexpr int load_j(int idx) {
        for (int j = j_lo; j <= j_hi; ++j) {
            const int c = tb(j) - ta(j) + 3;
            if (idx < c) return j;
            idx -= c;
        }
        return j_hi;
    }
    static constexpr int load_s(int idx) {
        for (int j = j_lo; j <= j_hi; ++j) {
            const int c = tb(j) - ta(j) + 3;
            if (idx < c) return ta(j) + idx;
            idx -= c;
        }
        return 0;
    }
    static constexpr int off(int idx) { return load_s(idx) * G::PITCH + 32 * load_j(idx); }
};

template <int CIN, int KH, int LI>
__device__ __forceinline__ void cp2_steps(const unsigned abase, f16x8 (&fh)[2], f16x8 (&fl)[2],
                                          const f16x8 (&bh)[Geo<CIN>::KS0], const f16x8 (&bl)[Geo<CIN>::KS0],
                                          f32x16 (&accm)[3], f32x16 (&accx)[3]) {
    using G = Geo<CIN>;
    using S = Cp2Sched<CIN, KH>;
    constexpr int NL = S::nloads(), cur = LI & 1, nxt = cur ^ 1;
    constexpr int j = S::load_j(LI), sv = S::load_s(LI);
    if constexpr (LI + 1 < NL) {
        constexpr int o = S::off(LI + 1);
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fh[nxt]) : "v"(abase), "n"(o));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fl[nxt]) : "v"(abase), "n"(o + G::PLANE));
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fh[cur]), "+v"(fl[cur]));
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fh[cur]), "+v"(fl[cur]));
    }
#pragma unroll
    for (int bk = 0; bk < 3; ++bk) {
        const int tap = sv - bk;
        if (tap >= S::ta(j) && tap <= S::tb(j)) {
            const int i = 5 * j + tap - S::n0;
            accx[bk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[cur], bh[i], accx[bk], 0, 0, 0);
            accm[bk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[cur], bh[i], accm[bk], 0, 0, 0);
            accx[bk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[cur], bl[i], accx[bk], 0, 0, 0);
        }
    }
    if constexpr (LI + 1 < NL) cp2_steps<CIN, KH, LI + 1>(abase, fh, fl, bh, bl, accm, accx);
}

template <int CIN, int KH>
__device__ __forceinline__ void cp2_tile(const unsigned abase, const f16x8 (&bh)[Geo<CIN>::KS0],
                                         const f16x8 (&bl)[Geo<CIN>::KS0], float* __restrict__ xw, volatile int* xdone,
                                         const int need) {
    using G = Geo<CIN>;
    using S = Cp2Sched<CIN, KH>;
    f16x8 fh[2], fl[2];
    f32x16 accm[3], accx[3];
#pragma unroll
    for (int bk = 0; bk < 3; ++bk)
#pragma unroll
        for (int r = 0; r < 16; ++r) accm[bk][r] = accx[bk][r] = 0.f;
    constexpr int o = S::off(0);
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fh[0]) : "v"(abase), "n"(o));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fl[0]) : "v"(abase), "n"(o + G::PLANE));
    cp2_steps<CIN, KH, 0>(abase, fh, fl, bh, bl, accm, accx);
    // the previous tile's blocks have been read by every service wave?  (long ago: they start with those reads)
    while (*xdone < need) __builtin_amdgcn_s_sleep(1);
#pragma unroll
    for (int bk = 0; bk < 3; ++bk)
#pragma unroll
        for (int r = 0; r < 16; ++r) xw[(bk * 16 + r) * 64] = accm[bk][r] + accx[bk][r] * (1.f / 2048.f);
}

template <int CIN>
__global__ __launch_bounds__(512, 1) void conv_pool_v2_kernel(
    const float* __restrict__ X, int Tin, int Tout, int Tstore, const float* __restrict__ nscale,
    const float* __restrict__ nshift, const float* __restrict__ npart, int npart_tiles, int npart_T,
    const float* __restrict__ ngamma, const float* __restrict__ nbeta,
    const unsigned short* __restrict__ wsp, int Kpad,
    const float* __restrict__ bias, float* __restrict__ Y, float* __restrict__ partials, int ntile,
    int total, int* __restrict__ oflag, long long* __restrict__ dbg) {
    using G = Geo<CIN>;
    using G2 = Geo2<CIN>;
#ifdef DZ_EXPERIMENTS       // phase stamps for tools/conv_pool_phases.py: [workgroup][wave][64], slot 63 = HW_ID | role << 32 | index << 33
    long long* dq = nullptr;
    int dn = 0;
    if (dbg && (threadIdx.x & 63) == 0) dq = dbg + ((long long)blockIdx.x * 8 + (threadIdx.x >> 6)) * 64;
#define CP2_STAMP() do { if (dq && dn < 60) dq[dn++] = __builtin_readcyclecounter(); } while (0)
#else
#define CP2_STAMP() do { } while (0)
#endif
    extern __shared__ __attribute__((aligned(256))) char lds[];
    float* xch = reinterpret_cast<float*>(lds + G2::OFF_XCH);
    float* red = reinterpret_cast<float*>(lds + G2::OFF_RED);
    float* nrm = reinterpret_cast<float*>(lds + G2::OFF_NRM);          // scale[CIN] | shift[CIN]
    int* simd_s = reinterpret_cast<int*>(lds + G2::OFF_MISC);           // [8]
    volatile int* xdone = reinterpret_cast<volatile int*>(lds + G2::OFF_MISC + 32);
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int wg = dz_xcd_contiguous(blockIdx.x, gridDim.x);
    const int t_begin = (int)((long long)wg * total / gridDim.x);
    const int t_end = (int)((long long)(wg + 1) * total / gridDim.x);
    if (t_begin >= t_end) return;

    // ---- roles: one matrix wave per SIMD when the eight waves sit two per SIMD (else the first four) ----------
    if (l == 0) simd_s[w] = (__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 4) & 3;       // HW_ID[5:4]
    if (tid == 0) *xdone = 0;
    __syncthreads();
    int cnt[4] = {0, 0, 0, 0}, my_simd = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int sk = __builtin_amdgcn_readfirstlane(simd_s[k]);
#pragma unroll
        for (int q = 0; q < 4; ++q) cnt[q] += sk == q;
        if (k == w) my_simd = sk;
    }
    const bool by_simd = cnt[0] == 2 && cnt[1] == 2 && cnt[2] == 2 && cnt[3] == 2;
    int rank = 0, idx = 0;             // rank among the waves of my SIMD; idx = index among the waves of my role
    bool matrix;
    {
        bool role[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int sk = __builtin_amdgcn_readfirstlane(simd_s[k]);
            int rk = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q < k) rk += __builtin_amdgcn_readfirstlane(simd_s[q]) == sk;
            role[k] = by_simd ? rk == 0 : k < 4;
        }
        (void)rank;
        (void)my_simd;
        matrix = false;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k == w) matrix = role[k];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < w) idx += role[k] == matrix;
    }
    idx = __builtin_amdgcn_readfirstlane(idx);
#ifdef DZ_EXPERIMENTS
    if (dq) dq[63] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)(matrix ? 1 : 0) << 32) | ((long long)idx << 33);
#endif

    float amax = 0.f;
    int cur_b = -1;
    if (matrix) {
        // =================== matrix wave (nt, kh): weights in registers, MFMAs only ===================
        const int li = l & 31, g = l >> 5;
        const int nt = idx & 1, kh = idx >> 1;
        f16x8 bh[G::KS0], bl[G::KS0];
        {   // fragment i of this wave = (sub-block j, tap) with 5 j + tap = n0 + i; the K axis of W is tap * CIN + c
            const int n0 = kh ? G::KS0 : 0, nw = kh ? G::KS - G::KS0 : G::KS0;
            const unsigned short* row = wsp + (long long)(32 * nt + li) * Kpad + 8 * g;
#pragma unroll
            for (int i = 0; i < G::KS0; ++i) {
                const int n = n0 + (i < nw ? i : 0), j = n / 5, tap = n - 5 * j;
                const int ks = tap * (CIN / 16) + j;
                bh[i] = *reinterpret_cast<const f16x8*>(row + 16 * ks);
                bl[i] = *reinterpret_cast<const f16x8*>(row + 64 * (long long)Kpad + 16 * ks);
            }
        }
        const unsigned a0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)lds +
                            3 * li * G::PITCH + 16 * g;
        float* xw = xch + ((kh * 2 + nt) * 3) * 16 * 64 + l;
        for (int t = t_begin - 1; t <= t_end + 1; ++t) {
            {   // (the norm update of the tile the service waves park in this iteration: every thread takes part)
                const int u = t + 1;
                if (u < t_end && u / ntile != cur_b) {
                    cur_b = u / ntile;
                    if (npart)
                        dz_norm_from_partials(npart, cur_b, npart_tiles, CIN, npart_T, ngamma, nbeta, nrm, tid, 512,
                                              reinterpret_cast<double*>(lds + ((u - t_begin) & 1) * G2::IN));
                    else
                        for (int i = tid; i < 2 * CIN; i += 512)
                            nrm[i] = i < CIN ? nscale[(long long)cur_b * CIN + i] : nshift[(long long)cur_b * CIN + i - CIN];
                    __syncthreads();
                }
            }
            CP2_STAMP();
            if (t >= t_begin && t < t_end) {
                const unsigned abase = a0 + ((t - t_begin) & 1) * G2::IN;
                const int need = 4 * (t - t_begin);
                if (kh == 0) cp2_tile<CIN, 0>(abase, bh, bl, xw, xdone, need);
                else cp2_tile<CIN, 1>(abase, bh, bl, xw, xdone, need);
            }
            CP2_STAMP();
            CP2_STAMP();
            CP2_STAMP();
            __syncthreads();
            CP2_STAMP();
        }
    } else {
        // =================== service wave: park tile t + 1, finish tile t - 1 ===========================
        const int ltid = idx * 64 + l;                      // 0 .. 255
        // parking: thread = (column quad c4, row r0), rows r0 + RPP i — the quad never changes, so its scale / shift
        // live in registers (reloaded when the chunk changes) instead of two LDS reads per piece
        constexpr int C4 = CIN / 4;
        constexpr int RPP = 256 / C4;                       // rows per pass: 12 (conv1, 240 threads) / 16 (conv2)
        constexpr int NPK = (ROWS + RPP - 1) / RPP;         // 9 / 7 passes
        const int pc4 = ltid % C4, pr0 = ltid / C4;
        const bool parker = ltid < RPP * C4;
        // byte offset of this lane's first piece inside a tile (idle lanes / rows >= ROWS: beyond any buffer)
        const int pvo = parker ? (pr0 * CIN + 4 * pc4) * 4 : 0x40000000;
        static_assert(ROWS <= (NPK - 1) * RPP + RPP, "passes cover the tile");
        f32x4 pv[NPK], psc = {0.f, 0.f, 0.f, 0.f}, psh = {0.f, 0.f, 0.f, 0.f};
        int pb = -1;                                        // chunk whose scale / shift is in psc / psh
        // epilogue coordinates: 4 channels of one pooled row per thread and pass
        const int ch4 = ltid & 15, rg = ltid >> 4;
        const int ent = ch4 >> 3, eli = (4 * ch4) & 31;
        f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 4 * ch4);
        // (consumed here, before the loop: left to its first use INSIDE the loop the compiler waits there with
        // vmcnt(0) in every iteration — behind the tile loads just issued, i.e. the full memory latency per tile)
        asm volatile("" : "+v"(bv));
        for (int t = t_begin - 1; t <= t_end + 1; ++t) {
            const int u = t + 1;
            {
                if (u < t_end && u / ntile != cur_b) {
                    cur_b = u / ntile;
                    if (npart)
                        dz_norm_from_partials(npart, cur_b, npart_tiles, CIN, npart_T, ngamma, nbeta, nrm, tid, 512,
                                              reinterpret_cast<double*>(lds + ((u - t_begin) & 1) * G2::IN));
                    else
                        for (int i = tid; i < 2 * CIN; i += 512)
                            nrm[i] = i < CIN ? nscale[(long long)cur_b * CIN + i] : nshift[(long long)cur_b * CIN + i - CIN];
                    __syncthreads();
                }
            }
            CP2_STAMP();
            // ---- tile e = t - 1, first half: its blocks out of the exchange area, which is then free for tile t ----
            const bool fin = t - 1 >= t_begin && t - 1 < t_end;
            f32x4 xv[2][3][2];
            if (fin) {
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int prl = rg + 16 * pass;                               // pooled row inside the tile
                    const int r = (prl & 3) + 4 * (prl >> 3), lane = eli + 32 * ((prl >> 2) & 1);
#pragma unroll
                    for (int bk = 0; bk < 3; ++bk)
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk)
                            xv[pass][bk][kk] = *reinterpret_cast<const f32x4*>(xch + (((kk * 2 + ent) * 3 + bk) * 16 + r) * 64 + lane);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xv[0][0][0]), "+v"(xv[1][2][1]) : : "memory");
                if (l == 0) __hip_atomic_fetch_add(const_cast<int*>(xdone), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // ---- tile u = t + 1: all of its rows in flight at once.  Buffer loads: the tile's part of the chunk is the
            // buffer, rows past the chunk's end read as zeros through the bounds check, and every lane's byte offsets are
            // constants of the kernel — nine load instructions, no address arithmetic, no branches --------------------
            const int ub = u / ntile, u0 = (u - ub * ntile) * FR;
            if (u < t_end) {
                const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(X + ((long long)ub * Tin + u0) * CIN), 0, (unsigned)(Tin - u0) * (CIN * 4u), 0x00020000);
#pragma unroll
                for (int i = 0; i < NPK; ++i)
                    pv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, pvo + i * (RPP * CIN * 4), 0, 0));
            }
            CP2_STAMP();
            // ---- tile e, second half: k-halves + bias, max over the three blocks, rows, this thread's sums ----
            if (fin) {
                const int e = t - 1, b = e / ntile, tile = e - b * ntile;
                const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(Y + (long long)b * Tstore * 64), 0, (unsigned)Tstore * 256u, 0x00020000);
                f32x4 sum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    f32x4 pm;
#pragma unroll
                    for (int bk = 0; bk < 3; ++bk) {
                        const f32x4 v = (xv[pass][bk][0] + xv[pass][bk][1]) + bv;
#pragma unroll
                        for (int k = 0; k < 4; ++k) pm[k] = bk == 0 ? v[k] : fmaxf(pm[k], v[k]);
                    }
                    // (the chunk's Y as a buffer of Tstore rows: a row of the ragged last tile beyond it is dropped)
                    const int pr = tile * 32 + rg + 16 * pass;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cp_u32x4, pm), yr, (pr * 64 + 4 * ch4) * 4, 0, 0);
                    if (pr < Tstore) {
                        sum += pm;
                        ssq += pm * pm;
                    }
                }
                float* rp = red + ((e - t_begin) & 1) * (16 * 64 * 2) + rg * 128 + 8 * ch4;      // [row group][ch][2]
                *reinterpret_cast<f32x4*>(rp) = (f32x4){sum[0], ssq[0], sum[1], ssq[1]};
                *reinterpret_cast<f32x4*>(rp + 4) = (f32x4){sum[2], ssq[2], sum[3], ssq[3]};
            }
            // ---- the partials of tile t - 2: the sixteen row groups' sums in fixed order --------------------
            if (t - 2 >= t_begin && t - 2 < t_end && ltid < 128) {
                const int e = t - 2, b = e / ntile, tile = e - b * ntile;
                const float* rp = red + ((e - t_begin) & 1) * (16 * 64 * 2) + ltid;      // ltid = 2 ch + {0, 1}
                float acc = rp[0];
#pragma unroll
                for (int k = 1; k < 16; ++k) acc += rp[128 * k];
                partials[((long long)b * ntile + tile) * 128 + ltid] = acc;
            }
            CP2_STAMP();
            // ---- park tile u: normalise + LeakyReLU, split, two planes ------------------------------------
            if (u < t_end) {
                if (pb != ub) {
                    pb = ub;
                    psc = *reinterpret_cast<const f32x4*>(nrm + 4 * pc4);
                    psh = *reinterpret_cast<const f32x4*>(nrm + CIN + 4 * pc4);
                }
                char* xs = lds + ((u - t_begin) & 1) * G2::IN + 8 * pc4;
#pragma unroll
                for (int i = 0; i < NPK; ++i) {
                    const int r = pr0 + RPP * i;
                    if (parker && r < ROWS) {
                        // packed f32 arithmetic where the ISA has it (v_pk_fma / v_pk_mul / v_pk_add: two elements per
                        // instruction), LeakyReLU as max(x, slope x) (slope in (0, 1)), rows past the chunk's end zeroed by
                        // a packed multiply: ~29 VALU instructions per 4 elements instead of ~45.  On this chip a SIMD's
                        // vector instructions do not overlap the MFMAs of the matrix wave next to them (measured: the same
                        // 60 instructions take 0.47 k cycles with the MFMAs compiled out and 3.4 k beside them), so every
                        // instruction here is tile time.
                        const f32x2 in01 = u0 + r < Tin ? (f32x2){1.f, 1.f} : (f32x2){0.f, 0.f};
                        f32x2 va = {pv[i][0], pv[i][1]}, vb = {pv[i][2], pv[i][3]};
                        va = __builtin_elementwise_fma(va, (f32x2){psc[0], psc[1]}, (f32x2){psh[0], psh[1]});
                        vb = __builtin_elementwise_fma(vb, (f32x2){psc[2], psc[3]}, (f32x2){psh[2], psh[3]});
                        const f32x2 sa = va * DZ_LEAKY_SLOPE, sb = vb * DZ_LEAKY_SLOPE;
                        f32x4 v = {fmaxf(va[0], sa[0]), fmaxf(va[1], sa[1]), fmaxf(vb[0], sb[0]), fmaxf(vb[1], sb[1])};
                        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = __builtin_amdgcn_fmed3f(v[k], -65504.f, 65504.f);
                        va = (f32x2){v[0], v[1]} * in01;
                        vb = (f32x2){v[2], v[3]} * in01;
                        v = (f32x4){va[0], va[1], vb[0], vb[1]};
                        const f16x4 hi = __builtin_convertvector(v, f16x4);
                        const f32x4 hf = __builtin_convertvector(hi, f32x4);
                        const f32x2 la = (va - (f32x2){hf[0], hf[1]}) * 2048.f, lb = (vb - (f32x2){hf[2], hf[3]}) * 2048.f;
                        const f16x4 lo = __builtin_convertvector((f32x4){la[0], la[1], lb[0], lb[1]}, f16x4);
                        char* d = xs + r * G::PITCH;
                        *reinterpret_cast<f16x4*>(d) = hi;
                        *reinterpret_cast<f16x4*>(d + G::PLANE) = lo;
                    }
                }
            }
            CP2_STAMP();
            __syncthreads();
            CP2_STAMP();
        }
    }
#undef CP2_STAMP
    dz_flag_range(oflag, amax);
}

#endif  // DZ_EXPERIMENTS (conv_pool_v2)

}  // namespace
#ifdef DZ_EXPERIMENTS
long long* dz_conv_pool_dbg = nullptr;      // set by dz_k_conv_pool_debug (phase stamps, kbench only)
#else
static long long* const dz_conv_pool_dbg = nullptr;
#endif
namespace {

#ifdef DZ_EXPERIMENTS
template <int CIN>
int launch_v2(const DzConvGemm& p, hipStream_t st) {
    using G2 = Geo2<CIN>;
    static DzAttrOnce attr_once;
    DZ_HIP(attr_once.raise((const void*)conv_pool_v2_kernel<CIN>, (int)G2::LDS));
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            n = 256;
        return n > 0 ? n : 256;
    }();
    const int ntile = (p.Tout + FR - 1) / FR;
    const int total = ntile * p.B;
    const int grid = total < cus ? total : cus;           // one workgroup of eight waves per CU
    DZ_LAUNCH((conv_pool_v2_kernel<CIN>), dim3(grid), dim3(512), G2::LDS, st, p.X, p.Tin, p.Tout, p.Tstore,
              p.nscale, p.nshift, p.npart, p.npart_tiles, p.npart_T, p.ngamma, p.nbeta,
              reinterpret_cast<const unsigned short*>(p.Wsplit), p.Kpad, p.bias, p.Y,
              p.partials, ntile, total, p.oflag ? p.oflag : dz_cur_oflag, dz_conv_pool_dbg);
    DZ_HIP(hipGetLastError());
    return 0;
}
#endif

template <int CIN>
int launch(const DzConvGemm& p, hipStream_t st, const void* wfrag) {
    using G = Geo<CIN>;
    static DzAttrOnce attr_once;
    DZ_HIP(attr_once.raise((const void*)conv_pool_h_kernel<CIN>, (int)G::LDS));
    const int ntile = (p.Tout + FR - 1) / FR;
    const int total = ntile * p.B;
    const int grid = total < 512 ? total : 512;          // two resident workgroups per CU
    DZ_LAUNCH((conv_pool_h_kernel<CIN>), dim3(grid), dim3(256), G::LDS, st, p.X, p.Tin, p.Tout, p.Tstore,
              p.nscale, p.nshift, p.npart, p.npart_tiles, p.npart_T, p.ngamma, p.nbeta,
              reinterpret_cast<const unsigned short*>(p.Wsplit), p.Kpad, p.bias, p.Y,
              p.partials, ntile, total, p.oflag ? p.oflag : dz_cur_oflag, dz_conv_pool_dbg,
              reinterpret_cast<const unsigned short*>(wfrag));
    DZ_HIP(hipGetLastError());
    return 0;
}

}  // namespace

int dz_conv_pool_wfrag_bytes(int Cin) { return 2 * 2 * 2 * (Cin == 80 ? Geo<80>::KS0 : Geo<64>::KS0) * 64 * 16; }
int dz_launch_conv_pool_wfrag(int Cin, const void* wsplit, int Kpad, void* out, hipStream_t st) {
    DZ_REQUIRE((Cin == 80 || Cin == 64) && wsplit && out && Kpad >= 5 * Cin, "conv_pool_wfrag: bad operands");
    const int n = dz_conv_pool_wfrag_bytes(Cin) / 16;
    if (Cin == 80)
        DZ_LAUNCH(conv_pool_wfrag_kernel<80>, dim3((n + 255) / 256), dim3(256), 0, st, reinterpret_cast<const unsigned short*>(wsplit),
                  Kpad, reinterpret_cast<f32x4*>(out));
    else
        DZ_LAUNCH(conv_pool_wfrag_kernel<64>, dim3((n + 255) / 256), dim3(256), 0, st, reinterpret_cast<const unsigned short*>(wsplit),
                  Kpad, reinterpret_cast<f32x4*>(out));
    DZ_HIP(hipGetLastError());
    return 0;
}

// same descriptor as the POOL3 call of dz_launch_gemm_split (k = 5, dil = 1, Npad = 64, norm-on-load,
// X / Y dense: xbs = Tin * Cin, ybs = Tstore * 64, ldx = Cin, ldy = 64)
int dz_launch_conv_pool(const DzConvGemm& p, hipStream_t st, const void* wfrag) {
    DZ_REQUIRE(p.Wsplit && p.X && p.Y && p.partials && p.bias, "conv_pool: NULL operand");
    DZ_REQUIRE((p.nscale && p.nshift) || (p.npart && p.ngamma && p.nbeta && p.npart_tiles > 0 && p.npart_T > 0),
               "conv_pool: needs nscale / nshift or the producer's partials + affine");
    DZ_REQUIRE(p.epi == DZ_EPI_POOL3 && p.taps == 5 && p.dil == 1 && p.norm_on_load && p.Npad == 64 &&
                   p.Nstore == 64 && p.ldy == 64,
               "conv_pool: built for k = 5, dilation 1, 64 output columns, norm-on-load, MaxPool1d(3)");
    DZ_REQUIRE((p.Cin == 80 || p.Cin == 64) && p.ldx == p.Cin && p.nld == p.Cin && p.K == 5 * p.Cin &&
                   p.Kpad >= p.K,
               "conv_pool: Cin must be 80 or 64 (dense rows), K = 5 Cin");
    DZ_REQUIRE(p.Tout == p.Tin - 4 && p.Tstore == p.Tout / 3 && p.xbs == (long long)p.Tin * p.Cin &&
                   p.ybs == (long long)p.Tstore * 64,
               "conv_pool: geometry mismatch");
#ifdef DZ_EXPERIMENTS
    // DZ_CONV_POOL_V2=1: conv_pool_v2 (faster alone, slower in the pipeline: see its header)
    const char* e_v2 = dz_exp_env("DZ_CONV_POOL_V2");
    if (e_v2 && e_v2[0] == '1') return p.Cin == 80 ? launch_v2<80>(p, st) : launch_v2<64>(p, st);
#endif
    return p.Cin == 80 ? launch<80>(p, st, wfrag) : launch<64>(p, st, wfrag);
}
