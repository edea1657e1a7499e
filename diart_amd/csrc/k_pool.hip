// Small bandwidth-bound kernels around the two networks:
//   stats_pool   weighted statistics pooling (paper Eq. 1; pyannote StatsPool), all K
//                speakers of a chunk in one pass over the frame features
//   osp          OverlappedSpeechPenalty        /root/reference/src/diart/functional.py:6-13
//                (+ min-max option              /root/reference/src/diart/blocks/embedding.py:102-106)
//   l2norm       EmbeddingNormalization         /root/reference/src/diart/functional.py:16-27
//   powerset     Powerset.to_multilabel (hard)  /root/reference/src/diart/models.py:29-39
//   cdist        cosine distances, fp64         /root/reference/src/diart/mapping.py:171-176
#include "dz_common.h"

namespace {

// ---------------------------------------------------------------------------
// stats_pool: one workgroup = (64-channel slab, one chunk).  The [T][64] slab of frame
// features is read from HBM once (256 B coalesced rows) and used for both passes (weighted
// mean, then centred second moment) of all K speakers.  Lanes = channels, the 4 waves split
// the frames; partial sums meet in LDS.
// The frames of a lane are held in REGISTERS (lane = channel, wave ph owns frames ph, ph+4, ...,
// at most NR of them), not in LDS: the round-1 version parked the [T][64] slab in LDS (73 KB), needed
// half a CU's LDS to itself and could not start while the GEMM workgroups of the other HIP streams
// (2 x 64 KB per CU) were resident — in the pipeline its launches took 0.2 - 5 ms instead of the 64 us
// it takes alone (profiles/r02_a_kernel_stats.md).  Same arithmetic and summation order as that
// version; 6 KB of LDS.
// ---------------------------------------------------------------------------
template <int K, int NR>
__global__ __launch_bounds__(256) void stats_pool_reg_kernel(
    const float* __restrict__ X, long long xstride, int T, int C, int ldx,
    const float* __restrict__ weights, int Fw, int ktot, int kofs, float* __restrict__ out, int ldo) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wk = smem;                 // [K][T]
    float* red = wk + K * T;          // [4][K][64]
    const int c0 = blockIdx.x * 64, xi = blockIdx.y, tid = threadIdx.x;
    const int c = tid & 63, ph = tid >> 6;
    const bool live = c0 + c < C;

    float xr[NR];
    {
        const float* Xb = X + (long long)xi * xstride + c0 + c;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int t = ph + 4 * i;
            xr[i] = (live && t < T) ? Xb[(long long)t * ldx] : 0.f;
        }
    }
    const int Fabs = Fw < 0 ? -Fw : Fw;
    for (int idx = tid; idx < K * T; idx += 256) {
        const int k = idx / T, t = idx - k * T;
        float wv = 1.f;
        if (weights) wv = dz_pool_weight(weights + (long long)(xi * ktot + kofs + k) * Fabs, Fw, T, t);
        wk[idx] = wv;
    }
    __syncthreads();

    float v1[K], v2[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float a = 0.f, b2 = 0.f;
        for (int t = (tid & 63); t < T; t += 64) {
            const float wv = wk[k * T + t];
            a += wv;
            b2 += wv * wv;
        }
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_xor(a, o, 64);
            b2 += __shfl_xor(b2, o, 64);
        }
        v1[k] = a;
        v2[k] = b2;
    }

    float mean[K], acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int t = ph + 4 * i;
        if (t < T) {
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k] += xr[i] * wk[k * T + t];
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) red[(ph * K + k) * 64 + c] = acc[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float s = (red[(0 * K + k) * 64 + c] + red[(1 * K + k) * 64 + c]) +
                        (red[(2 * K + k) * 64 + c] + red[(3 * K + k) * 64 + c]);
        mean[k] = s / v1[k];
        acc[k] = 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int t = ph + 4 * i;
        if (t < T) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float d = xr[i] - mean[k];
                acc[k] += (d * d) * wk[k * T + t];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) red[(ph * K + k) * 64 + c] = acc[k];
    __syncthreads();
    if (ph == 0 && live) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float s = (red[(0 * K + k) * 64 + c] + red[(1 * K + k) * 64 + c]) +
                            (red[(2 * K + k) * 64 + c] + red[(3 * K + k) * 64 + c]);
            const float var = s / (v1[k] - v2[k] / v1[k]);
            float* o = out + (long long)(xi * ktot + kofs + k) * ldo;
            o[c0 + c] = mean[k];
            o[C + c0 + c] = sqrtf(var);
        }
    }
}

template <int K>
int launch_pool(const float* X, long long xstride, int T, int C, int ldx, const float* weights, int Fw, int nx,
                int ktot, int kofs, float* out, int ldo, hipStream_t st) {
    const dim3 grid((C + 63) / 64, nx);
    DZ_REQUIRE(T <= 4 * 160, "stats_pool: %d frames per chunk (at most 640: an 11 s chunk)", T);
    const size_t lds = sizeof(float) * ((size_t)K * T + 4 * K * 64);
    if (T <= 4 * 72)              // 5 s chunks (279 frames): 172 VGPRs, i.e. a workgroup fits beside a
                                  // recurrence workgroup on its CU (2 x 168 of the 512 registers per SIMD)
        DZ_LAUNCH((stats_pool_reg_kernel<K, 72>), grid, dim3(256), lds, st, X, xstride, T, C, ldx, weights, Fw,
                  ktot, kofs, out, ldo);
    else if (T <= 4 * 80)
        DZ_LAUNCH((stats_pool_reg_kernel<K, 80>), grid, dim3(256), lds, st, X, xstride, T, C, ldx, weights, Fw,
                  ktot, kofs, out, ldo);
    else
        DZ_LAUNCH((stats_pool_reg_kernel<K, 160>), grid, dim3(256), lds, st, X, xstride, T, C, ldx, weights, Fw,
                  ktot, kofs, out, ldo);
    DZ_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// osp: one workgroup per chunk, thread = frame.  K <= 8 speakers.
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void osp_kernel(const float* __restrict__ seg, int F, int K,
                                                  float gamma, float beta, int normalize,
                                                  int speaker_major, float* __restrict__ out) {
    extern __shared__ float wbuf[];  // [F][K] then [2][K] min/max
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* sb = seg + (long long)b * F * K;
    for (int f = tid; f < F; f += 256) {
        float s[8], e[8];
        float m = -INFINITY;
        for (int k = 0; k < K; ++k) {
            s[k] = sb[f * K + k];
            m = fmaxf(m, beta * s[k]);
        }
        float sum = 0.f;
        for (int k = 0; k < K; ++k) {
            e[k] = expf(beta * s[k] - m);
            sum += e[k];
        }
        for (int k = 0; k < K; ++k) {
            const float pr = e[k] / sum;
            float wv = dz_powg(s[k], gamma) * dz_powg(pr, gamma);
            if (wv < 1e-8f) wv = 1e-8f;
            wbuf[f * K + k] = wv;
        }
    }
    __syncthreads();
    float* mm = wbuf + F * K;
    if (normalize) {
        if (tid < K) {
            float lo = INFINITY, hi = -INFINITY;
            bool nan = false;
            for (int f = 0; f < F; ++f) {
                const float v = wbuf[f * K + tid];
                nan |= (v != v);
                lo = fminf(lo, v);
                hi = fmaxf(hi, v);
            }
            if (nan) lo = hi = NAN;  // torch min/max propagate NaN
            mm[tid] = lo;
            mm[K + tid] = hi;
        }
        __syncthreads();
    }
    float* ob = out + (long long)b * F * K;
    for (int idx = tid; idx < F * K; idx += 256) {
        const int f = idx / K, k = idx - f * K;
        float v = wbuf[idx];
        if (normalize) {
            v = (v - mm[k]) / (mm[K + k] - mm[k]);
            if (v != v) v = 1e-8f;  // nan_to_num_(1e-8)
            else if (v == INFINITY) v = 3.4028234663852886e38f;
            else if (v == -INFINITY) v = -3.4028234663852886e38f;
        }
        if (speaker_major) ob[k * F + f] = v;
        else ob[idx] = v;
    }
}

// ---------------------------------------------------------------------------
// seg_head: the last three launches of the segmentation chain in one — Linear(128 -> classes) + bias,
// then sigmoid (multilabel models) or the hard powerset decision (argmax -> multilabel,
// models.py:29-39), then, when `wout` is given, the OverlappedSpeechPenalty weights of the frames
// (functional.py:6-13 + the optional min-max of blocks/embedding.py:102-106) in the speaker-major
// layout the pooling kernel reads.  One workgroup per chunk, thread = frame: the chunk's rows of the
// MLP output (F x 128 f32) are staged through LDS in slices (coalesced loads, conflict-free row
// reads at a 129-float pitch), the 128 x classes weights are wave-uniform scalar loads.  The OSP part
// is the arithmetic of osp_kernel above, statement by statement.
// ---------------------------------------------------------------------------
constexpr int SH_ROWS = 64;          // frames per LDS slice
constexpr int SH_PITCH = 129;

__global__ __launch_bounds__(256) void seg_head_kernel(const float* __restrict__ m1, const float* __restrict__ cw,
                                                       const float* __restrict__ cb, int F, int classes, int K,
                                                       int powerset, float* __restrict__ seg, float gamma,
                                                       float beta, int normalize, float* __restrict__ wout,
                                                       const float* __restrict__ wave_mom) {
    extern __shared__ float hbuf[];   // [SH_ROWS][SH_PITCH] slice | [F][K] weights | [2][K] min / max
    float* xs = hbuf;
    float* wbuf = hbuf + SH_ROWS * SH_PITCH;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* mb = m1 + (long long)b * F * 128;
    float* sb = seg + (long long)b * F * K;
    for (int f0 = 0; f0 < F; f0 += SH_ROWS) {
        const int nf = min(SH_ROWS, F - f0);
        __syncthreads();
        for (int i = tid; i < nf * 32; i += 256) {            // float4 per thread, coalesced
            const int r = i >> 5, c4 = i & 31;
            const f32x4 v = *reinterpret_cast<const f32x4*>(mb + (long long)(f0 + r) * 128 + 4 * c4);
            float* d = xs + r * SH_PITCH + 4 * c4;
            d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        }
        __syncthreads();
        if (tid < nf) {
            const float* x = xs + tid * SH_PITCH;
            float lg[8];
            for (int c = 0; c < classes; ++c) {
                float acc = 0.f;
                const float* wc = cw + c * 128;              // wave-uniform: scalar loads
#pragma unroll 8
                for (int k = 0; k < 128; ++k) acc = fmaf(x[k], wc[k], acc);
                lg[c] = acc + cb[c];
            }
            float s[8];
            dz_seg_decide(lg, classes, K, powerset, s);
            if (wave_mom && dz_ws_bad(wave_mom, b))            // a window with NaN / Inf samples: NaN rows, like the reference
                for (int k = 0; k < K; ++k) s[k] = __builtin_nanf("");
            const int f = f0 + tid;
            for (int k = 0; k < K; ++k) sb[f * K + k] = s[k];
            if (wout) {
                float wv[8];
                dz_osp_frame(s, K, gamma, beta, wv);
                for (int k = 0; k < K; ++k) wbuf[f * K + k] = wv[k];
            }
        }
    }
    if (!wout) return;
    __syncthreads();
    float* mm = wbuf + F * K;
    if (normalize) {
        if (tid < K) {
            float lo = INFINITY, hi = -INFINITY;
            bool nan = false;
            for (int f = 0; f < F; ++f) {
                const float v = wbuf[f * K + tid];
                nan |= (v != v);
                lo = fminf(lo, v);
                hi = fmaxf(hi, v);
            }
            if (nan) lo = hi = NAN;
            mm[tid] = lo;
            mm[K + tid] = hi;
        }
        __syncthreads();
    }
    float* ob = wout + (long long)b * F * K;
    for (int idx = tid; idx < F * K; idx += 256) {
        const int f = idx / K, k = idx - f * K;
        float v = wbuf[idx];
        if (normalize) {
            v = (v - mm[k]) / (mm[K + k] - mm[k]);
            if (v != v) v = 1e-8f;
            else if (v == INFINITY) v = 3.4028234663852886e38f;
            else if (v == -INFINITY) v = -3.4028234663852886e38f;
        }
        ob[k * F + f] = v;                                   // speaker-major (B, K, F)
    }
}

// ---------------------------------------------------------------------------
// l2norm: one wave per row, in place:  x <- (norm * x) / ||x||_2
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_kernel(float* __restrict__ x, int rows, int dim,
                                                     float norm) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (row >= rows) return;
    float* r = x + (long long)row * dim;
    float ss = 0.f;
    for (int i = l; i < dim; i += 64) ss += r[i] * r[i];
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float n = sqrtf(ss);
    for (int i = l; i < dim; i += 64) r[i] = (norm * r[i]) / n;
}

// ---------------------------------------------------------------------------
// splitk_finish: out[r][:] = sum_z parts[z][r][:] in fixed order (deterministic, unlike
// atomics), optionally followed by x <- x / ||x||_2 (normalize = 1) or x <- max(x, 0) (normalize = 2: the
// ReLU of a split-K layer, which cannot be applied to partial sums).  One workgroup per row, thread = column
// (dim <= 512); the partials of a column are fetched four at a time (the one-wave-per-row version
// walked 8 x nsplit dependent L2 round trips per lane: 45 us for 192 rows).  The sum of squares is
// formed in the order of that version (columns l, l+64, ... per lane, then the butterfly).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(512) void splitk_finish_kernel(const float* __restrict__ parts,
                                                            int nsplit, long long stride, int rows,
                                                            int dim, int normalize,
                                                            float* __restrict__ out,
                                                            const float* __restrict__ wave_mom, int rows_per_x) {
    __shared__ float sq[8][64];
    __shared__ float nrm;
    const int row = blockIdx.x, i = threadIdx.x, l = i & 63, w = i >> 6;
    float a = 0.f;
    if (i < dim) {
        const float* p = parts + (long long)row * dim + i;
        int z = 0;
        for (; z + 4 <= nsplit; z += 4) {
            const float x0 = p[(long long)z * stride], x1 = p[(long long)(z + 1) * stride],
                        x2 = p[(long long)(z + 2) * stride], x3 = p[(long long)(z + 3) * stride];
            a += x0;
            a += x1;
            a += x2;
            a += x3;
        }
        for (; z < nsplit; ++z) a += p[(long long)z * stride];
    }
    if (normalize == 2) a = fmaxf(a, 0.f);
    if (normalize == 1) {
        sq[w][l] = a * a;
        __syncthreads();
        if (w == 0) {
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += sq[j][l];
            for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
            if (l == 0) nrm = sqrtf(ss);
        }
        __syncthreads();
        a = a / nrm;
    }
    if (wave_mom && dz_ws_bad(wave_mom, row / rows_per_x)) a = __builtin_nanf("");      // (block-uniform)
    if (i < dim) out[(long long)row * dim + i] = a;
}

// ---------------------------------------------------------------------------
// powerset -> multilabel: one_hot(argmax) @ mapping, subsets ordered by size then
// lexicographically, at most 2 speakers per frame.
// ---------------------------------------------------------------------------
__global__ void powerset_kernel(const float* __restrict__ logit, int rows, int classes,
                                int speakers, float* __restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* p = logit + (long long)r * classes;
    int best = 0;
    float bv = p[0];
    for (int c = 1; c < classes; ++c)
        if (p[c] > bv) {
            bv = p[c];
            best = c;
        }
    int a = -1, b2 = -1;
    if (best >= 1 && best <= speakers) {
        a = best - 1;
    } else if (best > speakers) {
        int idx = best - speakers - 1;
        for (int i = 0; i < speakers && a < 0; ++i) {
            const int cnt = speakers - 1 - i;
            if (idx < cnt) {
                a = i;
                b2 = i + 1 + idx;
            } else {
                idx -= cnt;
            }
        }
    }
    float* o = out + (long long)r * speakers;
    for (int s = 0; s < speakers; ++s) o[s] = (s == a || s == b2) ? 1.f : 0.f;
}

// ---------------------------------------------------------------------------
// cdist (cosine, fp64) for N streams: one workgroup per (stream, local speaker);
// wave w takes centroids w, w+4, ...; scipy: 1 - u.v / (|u||v|), clipped to [-1,1].
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void cdist_kernel(const float* __restrict__ emb,
                                                    const double* __restrict__ centers, int k,
                                                    int g, int dim, double* __restrict__ out) {
    const int n = blockIdx.x / k, kk = blockIdx.x - n * k;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const float* e = emb + ((long long)n * k + kk) * dim;
    double ee = 0.0;
    for (int i = l; i < dim; i += 64) ee += (double)e[i] * (double)e[i];
    ee = sqrt(wave_sum(ee));
    for (int gg = w; gg < g; gg += 4) {
        const double* c = centers + ((long long)n * g + gg) * dim;
        double dot = 0.0, cc = 0.0;
        for (int i = l; i < dim; i += 64) {
            const double cv = c[i];
            dot += (double)e[i] * cv;
            cc += cv * cv;
        }
        dot = wave_sum(dot);
        cc = sqrt(wave_sum(cc));
        if (l == 0) {
            double cosv = dot / (ee * cc);
            if (fabs(cosv) > 1.0) cosv = copysign(1.0, cosv);
            out[((long long)n * k + kk) * g + gg] = 1.0 - cosv;
        }
    }
}

}  // namespace

int dz_launch_stats_pool(const float* X, long long xstride, int T, int C, int ldx,
                         const float* weights, int Fw, int rows, int rows_per_x, float* out, int ldo,
                         hipStream_t st) {
    DZ_REQUIRE(rows % rows_per_x == 0, "stats_pool: rows %% rows_per_x != 0");
    const int nx = rows / rows_per_x;
    int kofs = 0;
    while (kofs < rows_per_x) {
        const int kk = rows_per_x - kofs >= 4 ? 4 : rows_per_x - kofs;
        int rc;
        switch (kk) {
            case 4: rc = launch_pool<4>(X, xstride, T, C, ldx, weights, Fw, nx, rows_per_x, kofs, out, ldo, st); break;
            case 3: rc = launch_pool<3>(X, xstride, T, C, ldx, weights, Fw, nx, rows_per_x, kofs, out, ldo, st); break;
            case 2: rc = launch_pool<2>(X, xstride, T, C, ldx, weights, Fw, nx, rows_per_x, kofs, out, ldo, st); break;
            default: rc = launch_pool<1>(X, xstride, T, C, ldx, weights, Fw, nx, rows_per_x, kofs, out, ldo, st); break;
        }
        if (rc) return rc;
        kofs += kk;
    }
    return 0;
}

// ---------------------------------------------------------------------------
// pool_combine: the tile pieces of a chunk (means, centred second moments and weight sums left by
// the pooled epilogue of tdnn5, k_gemm_pre.hip) -> mean | std of paper Eq. 1 (pyannote StatsPool with
// weights: var = sum w (x - mean)^2 / (v1 - v2 / v1)).  Chan et al.'s pairwise update, in f64, pieces in
// fixed order.
// ---------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void pool_combine_kernel(const float* __restrict__ part, const float* __restrict__ s0,
                                                           int K, int np, int P, int T, int C, int Npad,
                                                           float* __restrict__ out, int ldo) {
    const int c = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;       // row = chunk * K + speaker
    if (c >= C) return;
    const int b = row / K, k = row - b * K;
    const int first = (b * P) / 128, last = (b * P + T - 1) / 128;
    double v1 = 0.0, v2 = 0.0, mean = 0.0, M2 = 0.0;
    for (int piece = 0; piece <= last - first; ++piece) {
        const float* so = s0 + (((long long)b * np + piece) * K + k) * 2;
        const double w1 = (double)so[0];
        if (!(w1 > 0.0)) continue;
        const float* pp = part + ((((long long)b * np + piece) * K + k) * Npad + c) * 2;
        const double mp = (double)pp[0], m2p = (double)pp[1];
        const double tot = v1 + w1, delta = mp - mean;
        M2 += m2p + delta * delta * (v1 * w1 / tot);
        mean += delta * (w1 / tot);
        v1 = tot;
        v2 += (double)so[1];
    }
    float* o = out + (long long)row * ldo;
    o[c] = (float)mean;
    o[C + c] = (float)sqrt(M2 / (v1 - v2 / v1));
}
}  // namespace

int dz_launch_pool_combine(const float* part, const float* s0, int nx, int K, int np, int P, int T, int C,
                           int Npad, float* out, int ldo, hipStream_t st) {
    DZ_REQUIRE(part && s0 && out && nx >= 1 && K >= 1 && K <= 4 && np == dz_pool_pieces(P), "pool_combine: bad arguments");
    DZ_LAUNCH(pool_combine_kernel, dim3((C + 255) / 256, nx * K), dim3(256), 0, st, part, s0, K, np, P, T, C, Npad, out,
              ldo);
    DZ_HIP(hipGetLastError());
    return 0;
}

int dz_launch_osp(const float* seg, int B, int F, int K, float gamma, float beta, int normalize,
                  int speaker_major, float* out, hipStream_t st) {
    DZ_REQUIRE(K >= 1 && K <= 8, "osp: 1 <= speakers <= 8 (got %d)", K);
    const size_t lds = sizeof(float) * ((size_t)F * K + 2 * K);
    DZ_REQUIRE(lds <= 64 * 1024, "osp: %d frames x %d speakers do not fit in LDS", F, K);
    DZ_LAUNCH(osp_kernel, dim3(B), dim3(256), lds, st, seg, F, K, gamma, beta, normalize,
                       speaker_major, out);
    DZ_HIP(hipGetLastError());
    return 0;
}

// m1 [B*F][128] (MLP output), cw [>= classes][128], cb [classes] -> seg [B][F][K]; wout (optional)
// [B][K][F] OSP weights
int dz_launch_seg_head(const float* m1, const float* cw, const float* cb, int B, int F, int classes, int K,
                       int powerset, float* seg, float gamma, float beta, int normalize, float* wout,
                       hipStream_t st, const float* wave_mom) {
    DZ_REQUIRE(classes >= 1 && classes <= 8 && K >= 1 && K <= 8 && (powerset || K == classes),
               "seg_head: classes %d / speakers %d", classes, K);
    const size_t lds = sizeof(float) * ((size_t)SH_ROWS * SH_PITCH + (size_t)F * K + 2 * K);
    DZ_REQUIRE(lds <= 64 * 1024, "seg_head: %d frames x %d speakers do not fit in LDS", F, K);
    DZ_LAUNCH(seg_head_kernel, dim3(B), dim3(256), lds, st, m1, cw, cb, F, classes, K, powerset, seg, gamma,
              beta, normalize, wout, wave_mom);
    DZ_HIP(hipGetLastError());
    return 0;
}

int dz_launch_l2norm(float* x, int rows, int dim, float norm, hipStream_t st) {
    DZ_LAUNCH(l2norm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, rows, dim, norm);
    DZ_HIP(hipGetLastError());
    return 0;
}

int dz_launch_splitk_finish(const float* parts, int nsplit, long long stride, int rows, int dim,
                            int normalize, float* out, hipStream_t st, const float* wave_mom, int rows_per_x) {
    DZ_REQUIRE(dim <= 512, "splitk_finish: dim %d > 512", dim);
    DZ_REQUIRE(rows_per_x >= 1, "splitk_finish: rows_per_x %d", rows_per_x);
    DZ_LAUNCH(splitk_finish_kernel, dim3(rows), dim3(512), 0, st, parts, nsplit,
                       stride, rows, dim, normalize, out, wave_mom, rows_per_x);
    DZ_HIP(hipGetLastError());
    return 0;
}

int dz_launch_powerset(const float* logp, int rows, int classes, int speakers, float* out,
                       hipStream_t st) {
    DZ_REQUIRE(classes == 1 + speakers + speakers * (speakers - 1) / 2,
               "powerset: %d classes is not 'at most 2 of %d speakers'", classes, speakers);
    DZ_LAUNCH(powerset_kernel, dim3((rows + 255) / 256), dim3(256), 0, st, logp, rows,
                       classes, speakers, out);
    DZ_HIP(hipGetLastError());
    return 0;
}

int dz_launch_cdist(const float* emb, const double* centers, int n, int k, int g, int dim,
                    double* out, hipStream_t st) {
    DZ_LAUNCH(cdist_kernel, dim3(n * k), dim3(256), 0, st, emb, centers, k, g, dim, out);
    DZ_HIP(hipGetLastError());
    return 0;
}
