// Persistent bidirectional-LSTM recurrence (hidden 128) for the segmentation network
// (pyannote PyanNet's nn.LSTM(60,128,4,bidirectional), called from
// /root/reference/src/diart/models.py:133; SURVEY.md Appendix A.1 step 2, kernel K5).
//
// The recurrence is a chain of T dependent 512x128 mat-vec products per (chunk, direction):
// latency bound, so the layout is chosen to (1) spread the chains over as many CUs as there
// are chains and (2) keep W_hh out of memory altogether.
//
//   one workgroup (512 threads) = ONE chunk x ONE direction, resident for all T steps
//   W_hh (512 x 128 f32 = 256 KiB) lives in the CU's vector registers: 128 per lane
//   lane (u, p): hidden unit u = 16*wave + lane/4, k-quarter p = lane%4
//                holds W[gate][u][16*jj + 4*p + e] for the 4 gates, jj < 8, e < 4
//   step:  h_{t-1} is read from LDS as 8 x ds_read_b128 (the 4 lanes of a quad read one
//          64-byte line: conflict-free), 64 v_pk_fma_f32 per lane, 3 DPP adds fold the four
//          k-quarters so that lane p ends with the complete pre-activation of gate p, one
//          exp+rcp per lane, a quad broadcast (4 DPP moves) gives every lane i,f,g,o.
//
// f32 MFMA runs at the f32 VALU rate on gfx950 (MI355X_MICROARCH.md: 64 FLOP/clk/SIMD both),
// so the matrix pipe buys nothing here, while its 16-column minimum would force 16 chunks
// through one CU (the first version of this kernel: 8 workgroups for 64 chunks, 5.9 us/step).
// With one chain per CU, 64 chunks x 2 directions occupy 128 of the 256 CUs and the x-vector
// TDNN stack of the same step runs beside it on the rest.
#include "dz_common.h"
#include <stdlib.h>

namespace {

// quad_perm DPP controls
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr int DPP_XOR1 = 0xB1;  // [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;  // [2,3,0,1]
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// sigmoid through the hardware exp2 / rcp (1 ulp each)
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}

// LDS-only workgroup barrier: __syncthreads() also drains vmcnt (its fence covers global
// memory), which would put the ~1 us completion latency of the h_t global store on the
// critical path of every step.  Only the LDS image of h_t has to be visible to the other waves.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

constexpr int RING = 16;   // h_t history kept in LDS (slot (t+1) & 15); flushed every 8 steps
constexpr int BLK = 8;     // steps per flush
constexpr int PF = 4;      // x-projection prefetch distance in steps

// UM: gx columns are unit-major (dir*512 + unit*4 + gate: what dz_seg_forward's projection GEMM
// writes, one fully contiguous 2 KiB row read per step) instead of PyTorch's gate-major
// (dir*512 + gate*128 + unit)
//
// NC = chains (chunks) per workgroup.  NC = 1: one chain per CU, the shortest step (config 5 latency,
// small batches).  NC = 2 (round 3): two chunks share the W_hh registers of one
// workgroup — the 64 packed FMAs of a lane run once per chain (16 independent accumulator chains
// instead of 8, the second chain's LDS / DPP / transcendental latencies under the first one's FMAs) and
// one barrier serves both; a 64-chunk batch then occupies 64 CUs instead of 128 for ~1.25x the layer
// time, i.e. 0.63x the CU-time, and leaves the other CUs ENTIRELY to the GEMM / convolution workgroups
// (which at 240 - 256 registers cannot sit beside a recurrence workgroup at all).
// PK: the contraction as 64 v_pk_fma_f32 per step (true) or 128 plain v_fma_f32 (false).  Alone the packed form
// is faster; BESIDE A WAVE THAT ISSUES MFMAs — where this kernel spends most of its life in the 64-stream pipeline —
// packed f32 VALU is the slow one (MI355X_MICROARCH.md, "price of one filler beside MFMAs"; measured here, round 4,
// tools/rec_contention.py: 187 us alone -> 320 - 355 us beside MFMA-issuing waves, while LDS-DMA, fragment reads and
// barriers of a neighbour cost it nothing).
template <bool UM, int NC, bool PK = true>
__global__ __launch_bounds__(512) void lstm_rec_kernel(const float* __restrict__ gx,
                                                       const float* __restrict__ whh,
                                                       float* __restrict__ hout,
                                                       unsigned short* __restrict__ hsp,
                                                       long long hplane, int B, int T) {
    __shared__ __attribute__((aligned(16))) float hs[NC][RING][128];
    const int dir = blockIdx.y;
    const int tid = threadIdx.x, p = tid & 3, u = tid >> 2;
    int bc[NC];                               // chunk of chain c (an odd batch's last workgroup repeats its chunk)
    bool live[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int b = blockIdx.x * NC + c;
        live[c] = b < B;
        bc[c] = live[c] ? b : B - 1;
    }

    // slot j of lane p holds gate (j ^ p): the three DPP adds below then need no selects.
    // Weights are kept as k-PAIRS (f32x2 in an even-aligned register pair) so the contraction is
    // 64 v_pk_fma_f32 per step instead of 128 v_fma_f32 (measured, tools/ubench/fma_rate.hip: the packed
    // form retires 1.3 - 1.4x the FLOPs of plain v_fma_f32 per cycle at two waves per SIMD).
    f32x2 wreg[4][16];
    {
        const float* Wd = whh + (long long)dir * 512 * 128;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* row = Wd + (long long)((j ^ p) * 128 + u) * 128 + 4 * p;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const float4 v = *reinterpret_cast<const float4*>(row + 16 * jj);
                wreg[j][2 * jj + 0] = (f32x2){v.x, v.y};
                wreg[j][2 * jj + 1] = (f32x2){v.z, v.w};
            }
        }
    }
    if (tid < 128) {
#pragma unroll
        for (int c = 0; c < NC; ++c) hs[c][0][tid] = 0.f;
    }

    // lane p owns gate p (PyTorch order i, f, g, o); g = tanh(x) = 2*sigmoid(2x) - 1
    const float act_scale = (p == 2) ? 2.f : 1.f;
    const float act_shift = (p == 2) ? -1.f : 0.f;
    // step s works on frame tt(s) = s (forward) or T-1-s (backward)
    const long long tstep = dir ? -1024 : 1024;
    const float* gptr[NC];
    long long hbase[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        gptr[c] = gx + ((long long)bc[c] * T + (dir ? T - 1 : 0)) * 1024 + dir * 512 + (UM ? u * 4 + p : p * 128 + u);
        // h_t goes out as f32 (hout) and / or as the two f16 planes a k_gemm_pre.hip consumer reads
        // (hsp: hi = f16(h), lo = f16((h - hi) * 2^11) hplane elements further)
        hbase[c] = (long long)bc[c] * T * 256 + dir * 128;
    }
    // flush role of this thread: step i = tid / 64 of the block, units 2*(tid % 64), +1
    const int fl_i = tid >> 6, fl_u = (tid & 63) * 2;

    float cst[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) cst[c] = 0.f;
    auto gload = [&](int c, int s) { return gptr[c][(long long)(s < T ? s : T - 1) * tstep]; };
    auto step = [&](int s, const float (&gcur)[NC]) {
        // acc[c][j] = (sum over even k, sum over odd k) of gate slot j of chain c
        f32x2 acc[NC][4];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[c][j] = (f32x2){0.f, 0.f};
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            f32x4 hv[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) hv[c] = *reinterpret_cast<const f32x4*>(&hs[c][s & (RING - 1)][4 * p + 16 * jj]);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const f32x2 h01 = {hv[c][0], hv[c][1]}, h23 = {hv[c][2], hv[c][3]};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (PK) {
                        acc[c][j] = __builtin_elementwise_fma(wreg[j][2 * jj + 0], h01, acc[c][j]);
                        acc[c][j] = __builtin_elementwise_fma(wreg[j][2 * jj + 1], h23, acc[c][j]);
                    } else {
                        // inline asm: hipcc's SLP vectoriser packs adjacent scalar FMAs right back into v_pk_fma_f32
                        float a0 = acc[c][j][0], a1 = acc[c][j][1];
                        asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(wreg[j][2 * jj + 0][0]), "v"(hv[c][0]));
                        asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(wreg[j][2 * jj + 0][1]), "v"(hv[c][1]));
                        asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(wreg[j][2 * jj + 1][0]), "v"(hv[c][2]));
                        asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(wreg[j][2 * jj + 1][1]), "v"(hv[c][3]));
                        acc[c][j][0] = a0;
                        acc[c][j][1] = a1;
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float s0_ = acc[c][0][0] + acc[c][0][1], s1_ = acc[c][1][0] + acc[c][1][1],
                        s2_ = acc[c][2][0] + acc[c][2][1], s3_ = acc[c][3][0] + acc[c][3][1];
            // fold the k-quarters: lane p ends with gate p (slot j of lane q is gate j ^ q)
            const float a0 = s0_ + dpp<DPP_XOR1>(s1_);
            const float a1 = s2_ + dpp<DPP_XOR1>(s3_);
            const float pre = a0 + dpp<DPP_XOR2>(a1) + gcur[c];
            const float act = act_scale * fast_sigmoid(act_scale * pre) + act_shift;
            const float ig = dpp<0x00>(act), fg = dpp<0x55>(act), gg = dpp<0xAA>(act),
                        og = dpp<0xFF>(act);
            cst[c] = fg * cst[c] + ig * gg;
            if (p == 0) hs[c][(s + 1) & (RING - 1)][u] = og * (2.f * fast_sigmoid(2.f * cst[c]) - 1.f);
        }
        lds_barrier();
    };
    // h of steps s0 .. s0+n-1 -> global (coalesced 256 B per step-row).  Nothing waits on these
    // stores: a ring slot is only rewritten 8 barriers after it was flushed.
    auto flush = [&](int s0, int n) {
        const int s = s0 + fl_i;
        if (fl_i < n) {
            const int tt = dir ? T - 1 - s : s;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (!live[c]) continue;
                const float2 v = *reinterpret_cast<const float2*>(&hs[c][(s + 1) & (RING - 1)][fl_u]);
                const long long o = hbase[c] + (long long)tt * 256 + fl_u;
                if (hout) *reinterpret_cast<float2*>(hout + o) = v;
                if (hsp) {
                    const f32x2 x = {v.x, v.y};
                    const f16x2 hi = __builtin_convertvector(x, f16x2);
                    const f16x2 lo = __builtin_convertvector((x - __builtin_convertvector(hi, f32x2)) * 2048.f, f16x2);
                    // planes in the kb-major order of dz_kb(): [256 / 32][B * T rows][32]
                    const long long ok = dz_kb((long long)bc[c] * T + tt, dir * 128 + fl_u, hplane >> 8);
                    *reinterpret_cast<f16x2*>(hsp + ok) = hi;
                    *reinterpret_cast<f16x2*>(hsp + hplane + ok) = lo;
                }
            }
        }
    };

    float gq[NC][PF];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < PF; ++i) gq[c][i] = gload(c, i);
    __syncthreads();

    int s0 = 0;
    for (; s0 + BLK <= T; s0 += BLK) {
#pragma unroll
        for (int i = 0; i < BLK; ++i) {
            float gcur[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                gcur[c] = gq[c][i % PF];
                gq[c][i % PF] = gload(c, s0 + i + PF);
            }
            step(s0 + i, gcur);
        }
        flush(s0, BLK);
    }
    for (int s = s0; s < T; ++s) {
        float gcur[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            gcur[c] = gq[c][0];
#pragma unroll
            for (int i = 0; i + 1 < PF; ++i) gq[c][i] = gq[c][i + 1];
            gq[c][PF - 1] = gload(c, s + PF);
        }
        step(s, gcur);
    }
    flush(s0, T - s0);
}

}  // namespace

int dz_launch_lstm(const float* gx, const float* whh, float* hout, void* hsplit, long long hplane,
                   int B, int T, int unit_major, hipStream_t st) {
    unsigned short* hsp = reinterpret_cast<unsigned short*>(hsplit);
    DZ_REQUIRE(hout || hsp, "lstm: no output");
    DZ_REQUIRE(hplane % 256 == 0 && (!hsplit || hplane >= (long long)B * T * 256),
               "lstm: the kb-major planes need hplane = rows * 256 with rows >= B * T");
    // chains per workgroup: DZ_LSTM_NC=2 selects the two-chunk form (EXPERIMENT).  Measured on MI355X, 64
    // chunks (gpurun_out/visit_r3k.log): alone 273.8 us per layer on 64 CUs against 169.7 us on 128 (0.81x the
    // CU-time, 1.61x the latency), and in the 64-stream pipeline 1.265 / 1.279 ms per step against 1.239 /
    // 1.229 in the same visit: the longer dependent chain of a lane costs more than the freed CUs return
    // (at 196 registers the two-chunk workgroup also keeps every GEMM workgroup off its CU).  Default: 1.
#ifdef DZ_EXPERIMENTS
    const char* e_nc = dz_exp_env("DZ_LSTM_NC");        // (read per launch: the tests switch it in-process)
    // DZ_LSTM_PK=0: plain v_fma_f32 in the contraction (see the PK template parameter)
    const char* e_pk = dz_exp_env("DZ_LSTM_PK");
    if (e_nc && e_nc[0] == '2') {
        dim3 grid((B + 1) / 2, 2);
        if (unit_major)
            DZ_LAUNCH((lstm_rec_kernel<true, 2>), grid, dim3(512), 0, st, gx, whh, hout, hsp, hplane, B, T);
        else
            DZ_LAUNCH((lstm_rec_kernel<false, 2>), grid, dim3(512), 0, st, gx, whh, hout, hsp, hplane, B, T);
        DZ_HIP(hipGetLastError());
        return 0;
    }
    if (e_pk && e_pk[0] == '0') {
        dim3 grid(B, 2);
        if (unit_major)
            DZ_LAUNCH((lstm_rec_kernel<true, 1, false>), grid, dim3(512), 0, st, gx, whh, hout, hsp, hplane, B, T);
        else
            DZ_LAUNCH((lstm_rec_kernel<false, 1, false>), grid, dim3(512), 0, st, gx, whh, hout, hsp, hplane, B, T);
        DZ_HIP(hipGetLastError());
        return 0;
    }
#endif
    dim3 grid(B, 2);
    if (unit_major)
        DZ_LAUNCH((lstm_rec_kernel<true, 1>), grid, dim3(512), 0, st, gx, whh, hout, hsp, hplane, B, T);
    else
        DZ_LAUNCH((lstm_rec_kernel<false, 1>), grid, dim3(512), 0, st, gx, whh, hout, hsp, hplane, B, T);
    DZ_HIP(hipGetLastError());
    return 0;
}
