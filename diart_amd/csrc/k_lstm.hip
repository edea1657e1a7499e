// Persistent bidirectional-LSTM recurrence (hidden 128) for the segmentation network
// (pyannote PyanNet's nn.LSTM(60,128,4,bidirectional), called from
// /root/reference/src/diart/models.py:133; SURVEY.md Appendix A.1 step 2, kernel K5).
//
// One workgroup = 16 chunks ("sequences") x one direction, resident for all T steps.
//   gates^T[512][16] = W_hh[512][128] . h^T[128][16]  (+ x-projection computed beforehand)
// W_hh never leaves the register file: 8 waves, wave w owns hidden units [16w,16w+16) for
// all four gates -> 4 gates x 32 k-steps = 128 A-fragments of v_mfma_f32_16x16x4_f32 per
// lane.  h_t is exchanged through a double-buffered LDS image laid out [k/32][seq][36] so
// that a lane's 32 consecutive k come from 8 conflict-free ds_read_b128.  k is permuted
// (lane quarter q covers k in [32q, 32q+32)) identically for both operands.
// The cell state and the four gate pre-activations of (unit, seq) live in the MFMA
// accumulator layout, so the gate non-linearities need no data movement.
#include "dz_common.h"

namespace {

constexpr int HS_LD = 36;  // 32 used + 4 pad: row pitch of 9 sixteen-byte slots (odd)

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(512) void lstm_rec_kernel(const float* __restrict__ gx,
                                                       const float* __restrict__ whh,
                                                       float* __restrict__ hout, int B, int T) {
    __shared__ __attribute__((aligned(16))) float hs[2][4][16][HS_LD];
    const int dir = blockIdx.y;
    const int b0 = blockIdx.x * 16;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, li = l & 15, q = l >> 4;

    // A fragments: wreg[g][ks] = W[g*128 + 16w + li][32q + ks]
    float wreg[4][32];
    {
        const float* Wd = whh + (long long)dir * 512 * 128;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float* row = Wd + (long long)(g * 128 + 16 * w + li) * 128 + 32 * q;
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
                const float4 v = reinterpret_cast<const float4*>(row)[k4];
                wreg[g][4 * k4 + 0] = v.x;
                wreg[g][4 * k4 + 1] = v.y;
                wreg[g][4 * k4 + 2] = v.z;
                wreg[g][4 * k4 + 3] = v.w;
            }
        }
    }
    for (int i = tid; i < 2 * 4 * 16 * HS_LD; i += 512) (&hs[0][0][0][0])[i] = 0.f;

    // this lane's sequence (column li) and hidden units u0..u0+3 (rows 4q..4q+3 of wave w)
    int bseq = b0 + li;
    const bool seq_valid = bseq < B;
    if (!seq_valid) bseq = B - 1;
    const int u0 = 16 * w + 4 * q;
    const float* gbase = gx + (long long)bseq * T * 1024 + dir * 512 + u0;
    float* hbase = hout + (long long)bseq * T * 256 + dir * 128 + u0;

    float cst[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 gnext[4];
    {
        const int tt = dir ? T - 1 : 0;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            gnext[g] = *reinterpret_cast<const f32x4*>(gbase + (long long)tt * 1024 + g * 128);
    }
    __syncthreads();

    for (int s = 0; s < T; ++s) {
        const int cur = s & 1;
        const int tt = dir ? T - 1 - s : s;
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = gnext[g];
        if (s + 1 < T) {  // prefetch next step's x-projection while the MFMAs run
            const int tn = dir ? tt - 1 : tt + 1;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                gnext[g] = *reinterpret_cast<const f32x4*>(gbase + (long long)tn * 1024 + g * 128);
        }
        // B fragments: hreg[ks] = h[seq li][32q + ks]
        float hreg[32];
        {
            const float* hp = &hs[cur][q][li][0];
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
                const f32x4 v = reinterpret_cast<const f32x4*>(hp)[k4];
                hreg[4 * k4 + 0] = v[0];
                hreg[4 * k4 + 1] = v[1];
                hreg[4 * k4 + 2] = v[2];
                hreg[4 * k4 + 3] = v[3];
            }
        }
#pragma unroll
        for (int ks = 0; ks < 32; ++ks)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = DZ_MFMA(wreg[g][ks], hreg[ks], acc[g]);

        // PyTorch gate order i, f, g, o
        f32x4 hv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ig = sigmoidf(acc[0][r]);
            const float fg = sigmoidf(acc[1][r]);
            const float gg = tanhf(acc[2][r]);
            const float og = sigmoidf(acc[3][r]);
            cst[r] = fg * cst[r] + ig * gg;
            hv[r] = og * tanhf(cst[r]);
        }
        // unit u0 -> k = u0: plane u0/32 = w>>1, offset u0%32 = (w&1)*16 + 4q
        *reinterpret_cast<f32x4*>(&hs[cur ^ 1][w >> 1][li][(w & 1) * 16 + 4 * q]) = hv;
        if (seq_valid) *reinterpret_cast<f32x4*>(hbase + (long long)tt * 256) = hv;
        __syncthreads();
    }
}

}  // namespace

int dz_launch_lstm(const float* gx, const float* whh, float* hout, int B, int T, hipStream_t st) {
    dim3 grid((B + 15) / 16, 2);
    hipLaunchKernelGGL(lstm_rec_kernel, grid, dim3(512), 0, st, gx, whh, hout, B, T);
    DZ_HIP(hipGetLastError());
    return 0;
}
