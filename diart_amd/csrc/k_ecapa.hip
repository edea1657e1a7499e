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
__global__ __launch_bounds__(256) void mask_compact_kernel(const float* __restrict__ wave,
                                                           long long stride, int S,
                                                           const float* __restrict__ masks, int Fw,
                                                           float* __restrict__ sig,
                                                           long long sig_stride,
                                                           int* __restrict__ lens) {
    __shared__ int cnt[256];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* w = wave + (long long)row * stride;
    float* o = sig + (long long)row * sig_stride + 200;
    if (masks == nullptr) {
        for (int s = tid; s < S; s += 256) o[s] = w[s];
        if (tid == 0) lens[row] = S;
        return;
    }
    const float* m = masks + (long long)row * Fw;
    const float scale = (float)Fw / (float)S;
    const int per = (S + 255) / 256;
    const int s0 = tid * per, s1 = min(S, s0 + per);
    int c = 0;
    for (int s = s0; s < s1; ++s) {
        const int f = min((int)floorf((float)s * scale), Fw - 1);
        c += m[f] > 0.5f;
    }
    cnt[tid] = c;
    __syncthreads();
    // exclusive scan (256 entries, done by every thread redundantly would be 256^2: use one wave)
    if (tid < 64) {
        int a0 = cnt[4 * tid], a1 = cnt[4 * tid + 1], a2 = cnt[4 * tid + 2], a3 = cnt[4 * tid + 3];
        int sum = a0 + a1 + a2 + a3, inc = sum;
        for (int o2 = 1; o2 < 64; o2 <<= 1) {
            const int v = __shfl_up(inc, o2, 64);
            if (tid >= o2) inc += v;
        }
        const int base = inc - sum;
        cnt[4 * tid] = base;
        cnt[4 * tid + 1] = base + a0;
        cnt[4 * tid + 2] = base + a0 + a1;
        cnt[4 * tid + 3] = base + a0 + a1 + a2;
        if (tid == 63) lens[row] = inc;
    }
    __syncthreads();
    int pos = cnt[tid];
    for (int s = s0; s < s1; ++s) {
        const int f = min((int)floorf((float)s * scale), Fw - 1);
        if (m[f] > 0.5f) o[pos++] = w[s];
    }
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

// s[row][c] = mean over t < nmask[row] of x[row][t][c]
__global__ __launch_bounds__(256) void se_mean_kernel(const float* __restrict__ x, int T, int C,
                                                      int ldx, const int* __restrict__ nmask,
                                                      float* __restrict__ s) {
    const int row = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float* xr = x + (long long)row * T * ldx + c;
    const int n = nmask[row];
    float a = 0.f;
    for (int t = 0; t < n; ++t) a += xr[(long long)t * ldx];
    s[(long long)row * C + c] = a / (float)n;
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

// global-context statistics: g[row][c] = mean, g[row][C + c] = sqrt(max(var, 1e-12)) over the
// nmask[row] valid frames with weights 1 / nmask
__global__ __launch_bounds__(256) void asp_gstats_kernel(const float* __restrict__ x, int T, int C,
                                                         const int* __restrict__ nmask,
                                                         float* __restrict__ g) {
    const int row = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float* xr = x + (long long)row * T * C + c;
    const int n = nmask[row];
    const float w = 1.f / (float)n;
    float mean = 0.f;
    for (int t = 0; t < n; ++t) mean += w * xr[(long long)t * C];
    float var = 0.f;
    for (int t = 0; t < n; ++t) {
        const float d = xr[(long long)t * C] - mean;
        var += w * (d * d);
    }
    g[(long long)row * 2 * C + c] = mean;
    g[(long long)row * 2 * C + C + c] = sqrtf(fmaxf(var, 1e-12f));
}

// attentive statistics: a = softmax_t(logit) over the valid frames;
// pooled[row][c] = sum a x, pooled[row][C + c] = sqrt(max(sum a (x - mean)^2, 1e-12))
__global__ __launch_bounds__(256) void asp_pool_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ logit, int T, int C,
                                                       const int* __restrict__ nmask,
                                                       float* __restrict__ pooled) {
    const int row = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float* xr = x + (long long)row * T * C + c;
    const float* lr = logit + (long long)row * T * C + c;
    const int n = nmask[row];
    float mx = -INFINITY;
    for (int t = 0; t < n; ++t) mx = fmaxf(mx, lr[(long long)t * C]);
    float den = 0.f;
    for (int t = 0; t < n; ++t) den += expf(lr[(long long)t * C] - mx);
    float mean = 0.f;
    for (int t = 0; t < n; ++t) mean += (expf(lr[(long long)t * C] - mx) / den) * xr[(long long)t * C];
    float var = 0.f;
    for (int t = 0; t < n; ++t) {
        const float d = xr[(long long)t * C] - mean;
        var += (expf(lr[(long long)t * C] - mx) / den) * (d * d);
    }
    pooled[(long long)row * 2 * C + c] = mean;
    pooled[(long long)row * 2 * C + C + c] = sqrtf(fmaxf(var, 1e-12f));
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
    DZ_LAUNCH(mask_compact_kernel, dim3(rows), dim3(256), 0, st, wave, stride, S, masks, Fw,
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
int dz_launch_asp_gstats(const float* x, int T, int C, int rows, const int* nmask, float* g,
                         hipStream_t st) {
    DZ_LAUNCH(asp_gstats_kernel, dim3((C + 255) / 256, rows), dim3(256), 0, st, x, T, C,
                       nmask, g);
    DZ_HIP(hipGetLastError());
    return 0;
}
int dz_launch_asp_pool(const float* x, const float* logit, int T, int C, int rows, const int* nmask,
                       float* pooled, hipStream_t st) {
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
