// Kernels around the ECAPA-TDNN embedding (BASELINE.json config 3): everything that is not a
// convolution / linear layer (those run on convgemm, k_convgemm.hip).
//   mask_compact   PretrainedSpeakerEmbedding.__call__: nearest-resampled mask > 0.5 -> kept samples
//   power          |STFT|^2 from the (re | im) GEMM output
//   fbank_post     10 log10 -> top_db clip against the row maximum -> sentence mean normalisation
//   se_mean / se_apply     squeeze-excitation: masked time mean, gate * x + residual
//   asp_gstats / asp_pool  attentive statistics pooling: global context stats, masked softmax stats
//   nan_rows       rows with fewer than min_num_samples kept samples -> NaN
// Third-party graph (speechbrain ECAPA_TDNN via pyannote's PretrainedSpeakerEmbedding) reached
// from /root/reference/src/diart/models.py:59 and :262; SURVEY.md Appendix A.3.
#include "dz_common.h"

namespace {

// ---------------------------------------------------------------------------
// mask_compact: one workgroup per row.  Sample s is kept iff
// masks[min(floor(s * Fw / S), Fw - 1)] > 0.5  (F.interpolate(mode="nearest")); kept samples are
// packed in order at sig[row][200 ...] (200 = the n_fft/2 zero padding of the centred STFT;
// the buffer is zero-filled before, which also provides pad_sequence's zeros).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void mask_compact_kernel(const float* __restrict__ wave,
                                                           long long stride, int S,
                                                           const float* __restrict__ masks, int Fw,
                                                           float* __restrict__ sig,
                                                           long long sig_stride,
                                                           int* __restrict__ lens) {
    __shared__ int cnt[256];
    __shared__ int bad_s;                     // a kept sample is NaN / Inf: the row's length is reported as -(len + 1)
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* w = wave + (long long)row * stride;
    float* o = sig + (long long)row * sig_stride + 200;
    if (tid == 0) bad_s = 0;
    __syncthreads();
    bool bad = false;
    auto finite = [](float v) { return (__float_as_uint(v) & 0x7f800000u) != 0x7f800000u; };
    if (masks == nullptr) {
        for (int s = tid; s < S; s += blockDim.x) {
            const float v = w[s];
            bad |= !finite(v);
            o[s] = finite(v) ? v : 0.f;       // (the row's embedding is NaN whatever is computed from this)
        }
        if (bad) bad_s = 1;
        __syncthreads();
        if (tid == 0) lens[row] = bad_s ? -(S + 1) : S;
        return;
    }
    const float* m = masks + (long long)row * Fw;
    const float scale = (float)Fw / (float)S;
    // wave w of 16 owns the samples [w * seg, (w + 1) * seg); 64 consecutive samples per iteration, kept ones packed
    // with a ballot (coalesced reads, near-coalesced writes; the first version gave each THREAD a contiguous
    // run of 313 samples: strided lanes, 200 us for 96 rows)
    const int w4 = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    const int seg = ((S + nw - 1) / nw + 63) & ~63;
    const int s0 = min(S, w4 * seg), s1 = min(S, s0 + seg);
    auto kept = [&](int s) -> bool {
        if (s >= s1) return false;
        const int f = min((int)floorf((float)s * scale), Fw - 1);
        return m[f] > 0.5f;
    };
    int c = 0;
    for (int s = s0; s < s1; s += 64) c += __popcll(__ballot(kept(s + lane)));
    if (lane == 0) cnt[w4] = c;
    __syncthreads();
    int pos = 0;
    for (int i = 0; i < w4; ++i) pos += cnt[i];
    const int total = pos + c;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int s = s0; s < s1; s += 64) {
        const bool k = kept(s + lane);
        const unsigned long long b = __ballot(k);
        if (k) {
            const float v = w[s + lane];
            bad |= !finite(v);
            o[pos + __popcll(b & below)] = finite(v) ? v : 0.f;
        }
        pos += __popcll(b);
    }
    if (bad) bad_s = 1;
    __syncthreads();
    if (tid == (int)blockDim.x - 1) lens[row] = bad_s ? -(total + 1) : total;
}

// spec [rows][lds] = (re[0..200] | im[0..200]) -> pw [rows][204] (cols 201..203 = 0)
__global__ void power_kernel(const float* __restrict__ spec, int lds, long long rows,
                             float* __restrict__ pw) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * 204) return;
    const long long r = idx / 204;
    const int j = (int)(idx - r * 204);
    float v = 0.f;
    if (j < 201) {
        const float re = spec[r * lds + j], im = spec[r * lds + 201 + j];
        v = re * re + im * im;
    }
    pw[idx] = v;
}

__device__ __forceinline__ float block_max(float v, float* red) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// mel power [row][T][80] -> features [row][T][80]:
//   x_db = 10 log10(max(x, 1e-10)); x_db = max(x_db, rowmax - 80); x_db -= mean over the first
//   nvalid[row] frames (speechbrain Filterbank top_db + InputNormalization("sentence")).
__global__ __launch_bounds__(256) void fbank_post_kernel(const float* __restrict__ melp, int T,
                                                         const int* __restrict__ nvalid,
                                                         float* __restrict__ feats) {
    __shared__ float red[4];
    __shared__ float msum[3][80];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* x = melp + (long long)row * T * 80;
    float* y = feats + (long long)row * T * 80;
    float mx = -INFINITY;
    for (int i = tid; i < T * 80; i += 256) mx = fmaxf(mx, 10.f * log10f(fmaxf(x[i], 1e-10f)));
    const float floor_db = block_max(mx, red) - 80.f;
    // per-mel mean over the valid frames: thread (m = tid % 80, part = tid / 80) for tid < 240
    const int nv = nvalid[row];
    if (tid < 240) {
        const int m = tid % 80, part = tid / 80;
        float s = 0.f;
        for (int t = part; t < nv; t += 3) s += fmaxf(10.f * log10f(fmaxf(x[t * 80 + m], 1e-10f)), floor_db);
        msum[part][m] = s;
    }
    __syncthreads();
    for (int i = tid; i < T * 80; i += 256) {
        const int m = i % 80;
        const float mean = ((msum[0][m] + msum[1][m]) + msum[2][m]) / (float)nv;
        y[i] = fmaxf(10.f * log10f(fmaxf(x[i], 1e-10f)), floor_db) - mean;
    }
}

// The three reductions over time below share one shape: a workgroup = 256 channels (64 lanes x float4) x 4
// waves, wave w reduces the frames [w n / 4, (w + 1) n / 4) of the row's n valid ones, the four partial
// results are combined in LDS in the fixed order ((0 + 1) + 2) + 3.  (The first versions walked all n
// frames in ONE thread per channel with 4-byte loads: 124 / 253 / 583 us per launch at 96 rows.)
__device__ __forceinline__ f32x4 dz_sum4(f32x4 (*part)[64], int lane) {
    return ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}

// s[row][c] = mean over t < nmask[row] of x[row][t][c]
__global__ __launch_bounds__(256) void se_mean_kernel(const float* __restrict__ x, int T, int C,
                                                      int ldx, const int* __restrict__ nmask,
                                                      float* __restrict__ s) {
    __shared__ f32x4 part[4][64];
    const int row = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + lane * 4;
    const bool ok = c < C;
    const float* xr = x + (long long)row * T * ldx + c;
    const int n = nmask[row];
    const int t0 = (int)((long long)w * n / 4), t1 = (int)((long long)(w + 1) * n / 4);
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (ok)
        for (int t = t0; t < t1; ++t) a += *reinterpret_cast<const f32x4*>(xr + (long long)t * ldx);
    part[w][lane] = a;
    __syncthreads();
    if (w == 0 && ok) *reinterpret_cast<f32x4*>(s + (long long)row * C + c) = dz_sum4(part, lane) / (float)n;
}

// out = gate[row][c] * x + resid   (float4 over channels)
__global__ void se_apply_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gate,
                                const float* __restrict__ resid, int ldr, float* __restrict__ out,
                                int ldo, int T, int C, long long total4) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total4) return;
    const int c4 = C / 4;
    const long long rt = idx / c4;          // row * T + t
    const int c = (int)(idx - rt * c4) * 4;
    const long long row = rt / T;
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + rt * ldx + c);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gate + row * C + c);
    const f32x4 r = *reinterpret_cast<const f32x4*>(resid + rt * ldr + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = g[e] * xv[e] + r[e];
    *reinterpret_cast<f32x4*>(out + rt * ldo + c) = o;
}

// The same element-wise pass (GATE) or a plain copy (!GATE) of a 128-column-aligned block of columns that ALSO
// writes the result as the two f16 planes a k_gemm_pre.hip consumer reads (kb-major, dz_kb in dz_common.h: a
// plane of R rows is [C / 32][R][32]).  A wave owns 8 rows x 32 columns (lane = 8 * row + column quad): its f32
// accesses are eight full 128-byte lines and its plane writes 512 contiguous bytes per plane (eight 64-byte
// rows of one k-block) — the row-major lane order of se_apply_kernel would scatter 8-byte pieces over eight
// k-blocks.  `out` may be NULL (planes only).
typedef _Float16 se_f16x4 __attribute__((ext_vector_type(4)));
template <bool GATE>
__global__ __launch_bounds__(256) void se_apply_planes_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ gate, const float* __restrict__ resid, int ldr,
    float* __restrict__ out, int ldo, unsigned short* __restrict__ planes, long long plane, long long R, int T, int C,
    int* __restrict__ oflag) {
    const int tid = threadIdx.x;
    const int c = blockIdx.x * 128 + (tid >> 6) * 32 + (tid & 7) * 4;
    const long long rt = (long long)blockIdx.y * 8 + ((tid >> 3) & 7);
    if (rt >= R || c >= C) return;
    f32x4 o = *reinterpret_cast<const f32x4*>(x + rt * ldx + c);
    if (GATE) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gate + (rt / T) * C + c);
        const f32x4 r = *reinterpret_cast<const f32x4*>(resid + rt * ldr + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = g[e] * o[e] + r[e];     // (the arithmetic of se_apply_kernel)
    }
    if (out) *reinterpret_cast<f32x4*>(out + rt * ldo + c) = o;
    float amax = 0.f;
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        amax = fmaxf(amax, fabsf(o[e]));
        v[e] = __builtin_amdgcn_fmed3f(o[e], -65504.f, 65504.f);
    }
    const se_f16x4 hi = __builtin_convertvector(v, se_f16x4);
    const se_f16x4 lo = __builtin_convertvector((v - __builtin_convertvector(hi, f32x4)) * 2048.f, se_f16x4);
    const long long idx = dz_kb(rt, c, R);
    *reinterpret_cast<se_f16x4*>(planes + idx) = hi;
    *reinterpret_cast<se_f16x4*>(planes + plane + idx) = lo;
    dz_flag_range(oflag, amax);
}

// global-context statistics: g[row][c] = mean, g[row][C + c] = sqrt(max(var, 1e-12)) over the
// nmask[row] valid frames with weights 1 / nmask
__global__ __launch_bounds__(256) void asp_gstats_kernel(const float* __restrict__ x, int T, int C,
                                                         const int* __restrict__ nmask,
                                                         float* __restrict__ g) {
    __shared__ f32x4 part[4][64];
    const int row = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + lane * 4;
    const bool ok = c < C;
    const float* xr = x + (long long)row * T * C + c;
    const int n = nmask[row];
    const int t0 = (int)((long long)w * n / 4), t1 = (int)((long long)(w + 1) * n / 4);
    const float wt = 1.f / (float)n;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (ok)
        for (int t = t0; t < t1; ++t) a += wt * *reinterpret_cast<const f32x4*>(xr + (long long)t * C);
    part[w][lane] = a;
    __syncthreads();
    const f32x4 mean = dz_sum4(part, lane);
    __syncthreads();
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok)
        for (int t = t0; t < t1; ++t) {
            const f32x4 d = *reinterpret_cast<const f32x4*>(xr + (long long)t * C) - mean;
            v += wt * (d * d);
        }
    part[w][lane] = v;
    __syncthreads();
    if (w == 0 && ok) {
        const f32x4 var = dz_sum4(part, lane);
        f32x4 sd;
#pragma unroll
        for (int e = 0; e < 4; ++e) sd[e] = sqrtf(fmaxf(var[e], 1e-12f));
        *reinterpret_cast<f32x4*>(g + (long long)row * 2 * C + c) = mean;
        *reinterpret_cast<f32x4*>(g + (long long)row * 2 * C + C + c) = sd;
    }
}

// attentive statistics: a = softmax_t(logit) over the valid frames;
// pooled[row][c] = sum a x, pooled[row][C + c] = sqrt(max(sum a (x - mean)^2, 1e-12)).
// Three passes over the logits and two over x (max | sum e and sum e x | sum e (x - mean)^2), the softmax
// denominator applied once per sum instead of once per frame.
__global__ __launch_bounds__(256) void asp_pool_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ logit, int T, int C,
                                                       const int* __restrict__ nmask,
                                                       float* __restrict__ pooled) {
    __shared__ f32x4 part[4][64];
    const int row = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + lane * 4;
    const bool ok = c < C;
    const float* xr = x + (long long)row * T * C + c;
    const float* lr = logit + (long long)row * T * C + c;
    const int n = nmask[row];
    const int t0 = (int)((long long)w * n / 4), t1 = (int)((long long)(w + 1) * n / 4);
    f32x4 mx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (ok)
        for (int t = t0; t < t1; ++t) {
            const f32x4 l = *reinterpret_cast<const f32x4*>(lr + (long long)t * C);
#pragma unroll
            for (int e = 0; e < 4; ++e) mx[e] = fmaxf(mx[e], l[e]);
        }
    part[w][lane] = mx;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e)
        mx[e] = fmaxf(fmaxf(part[0][lane][e], part[1][lane][e]), fmaxf(part[2][lane][e], part[3][lane][e]));
    __syncthreads();
    f32x4 den = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    if (ok)
        for (int t = t0; t < t1; ++t) {
            const f32x4 l = *reinterpret_cast<const f32x4*>(lr + (long long)t * C);
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + (long long)t * C);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ex = expf(l[e] - mx[e]);
                den[e] += ex;
                s1[e] += ex * xv[e];
            }
        }
    part[w][lane] = den;
    __syncthreads();
    den = dz_sum4(part, lane);
    __syncthreads();
    part[w][lane] = s1;
    __syncthreads();
    const f32x4 mean = dz_sum4(part, lane) / den;
    __syncthreads();
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok)
        for (int t = t0; t < t1; ++t) {
            const f32x4 l = *reinterpret_cast<const f32x4*>(lr + (long long)t * C);
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + (long long)t * C);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = xv[e] - mean[e];
                v[e] += expf(l[e] - mx[e]) * (d * d);
            }
        }
    part[w][lane] = v;
    __syncthreads();
    if (w == 0 && ok) {
        const f32x4 var = dz_sum4(part, lane) / den;
        f32x4 sd;
#pragma unroll
        for (int e = 0; e < 4; ++e) sd[e] = sqrtf(fmaxf(var[e], 1e-12f));
        *reinterpret_cast<f32x4*>(pooled + (long long)row * 2 * C + c) = mean;
        *reinterpret_cast<f32x4*>(pooled + (long long)row * 2 * C + C + c) = sd;
    }
}

__global__ void nan_rows_kernel(float* __restrict__ out, int rows, int dim,
                                const int* __restrict__ flags) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * dim) return;
    if (flags[idx / dim]) out[idx] = __int_as_float(0x7fc00000);
}

}  // namespace

int dz_launch_mask_compact(const float* wave, long long stride, int S, const float* masks, int Fw,
                           int rows, float* sig, long long sig_stride, int* lens, hipStream_t st) {
    DZ_LAUNCH(mask_compact_kernel, dim3(rows), dim3(1024), 0, st, wave, stride, S, masks, Fw,
                       sig, sig_stride, lens);
    DZ_HIP(hipGetLastError());
    return 0;
}
int dz_launch_power(const float* spec, int lds, long long rows, float* pw, hipStream_t st) {
    const long long n = rows * 204;
    DZ_LAUNCH(power_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, spec, lds,
                       rows, pw);
    DZ_HIP(hipGetLastError());
    return 0;
}
int dz_launch_fbank_post(const float* melp, int T, int rows, const int* nvalid, float* feats,
                         hipStream_t st) {
    DZ_LAUNCH(fbank_post_kernel, dim3(rows), dim3(256), 0, st, melp, T, nvalid, feats);
    DZ_HIP(hipGetLastError());
    return 0;
}
int dz_launch_se_mean(const float* x, int T, int C, int ldx, int rows, const int* nmask, float* s,
                      hipStream_t st) {
    DZ_REQUIRE(C % 4 == 0 && ldx % 4 == 0, "se_mean: channels must be a multiple of 4");
    DZ_LAUNCH(se_mean_kernel, dim3((C + 255) / 256, rows), dim3(256), 0, st, x, T, C, ldx,
                       nmask, s);
    DZ_HIP(hipGetLastError());
    return 0;
}
int dz_launch_se_apply(const float* x, int ldx, const float* gate, const float* resid, int ldr,
                       float* out, int ldo, int rows, int T, int C, hipStream_t st) {
    const long long total4 = (long long)rows * T * (C / 4);
    DZ_LAUNCH(se_apply_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, x,
                       ldx, gate, resid, ldr, out, ldo, T, C, total4);
    DZ_HIP(hipGetLastError());
    return 0;
}
// gate == NULL: planes (and out, if given) = x.  x / out / resid point at the block's first column; `planes` at the
// k-block of that column (plane base + (column / 32) * R * 32), `plane` = elements between the hi and lo planes,
// R = rows * T = rows of a plane.
int dz_launch_se_apply_planes(const float* x, int ldx, const float* gate, const float* resid, int ldr, float* out,
                              int ldo, void* planes, long long plane, int rows, int T, int C, hipStream_t st) {
    DZ_REQUIRE(x && planes && C % 32 == 0 && ldx % 4 == 0 && (out == nullptr || ldo % 4 == 0) && plane % 4 == 0,
               "se_apply_planes: bad operands");
    DZ_REQUIRE(gate == nullptr || (resid != nullptr && ldr % 4 == 0), "se_apply_planes: gate without residual");
    const long long R = (long long)rows * T;
    const dim3 grid((C + 127) / 128, (unsigned)((R + 7) / 8));
    if (gate)
        DZ_LAUNCH(se_apply_planes_kernel<true>, grid, dim3(256), 0, st, x, ldx, gate, resid, ldr, out, ldo,
                  reinterpret_cast<unsigned short*>(planes), plane, R, T, C, dz_cur_oflag);
    else
        DZ_LAUNCH(se_apply_planes_kernel<false>, grid, dim3(256), 0, st, x, ldx, gate, resid, ldr, out, ldo,
                  reinterpret_cast<unsigned short*>(planes), plane, R, T, C, dz_cur_oflag);
    DZ_HIP(hipGetLastError());
    return 0;
}
int dz_launch_asp_gstats(const float* x, int T, int C, int rows, const int* nmask, float* g,
                         hipStream_t st) {
    DZ_REQUIRE(C % 4 == 0, "asp_gstats: channels must be a multiple of 4");
    DZ_LAUNCH(asp_gstats_kernel, dim3((C + 255) / 256, rows), dim3(256), 0, st, x, T, C,
                       nmask, g);
    DZ_HIP(hipGetLastError());
    return 0;
}
int dz_launch_asp_pool(const float* x, const float* logit, int T, int C, int rows, const int* nmask,
                       float* pooled, hipStream_t st) {
    DZ_REQUIRE(C % 4 == 0, "asp_pool: channels must be a multiple of 4");
    DZ_LAUNCH(asp_pool_kernel, dim3((C + 255) / 256, rows), dim3(256), 0, st, x, logit, T, C,
                       nmask, pooled);
    DZ_HIP(hipGetLastError());
    return 0;
}
int dz_launch_nan_rows(float* out, int rows, int dim, const int* flags, hipStream_t st) {
    DZ_LAUNCH(nan_rows_kernel, dim3((rows * dim + 255) / 256), dim3(256), 0, st, out, rows,
                       dim, flags);
    DZ_HIP(hipGetLastError());
    return 0;
}
