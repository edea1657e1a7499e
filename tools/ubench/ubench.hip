// Instruction-rate micro-benchmarks for gfx950 (one workgroup on one CU): cycles per wave-instruction
// for plain / packed / transcendental VALU ops and f16 MFMAs, alone and interleaved, at 1 and 2
// waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/ubench.hip -o tools/ubench/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
#define REP16(X) REP8(X) REP8(X)

template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
    float a0 = threadIdx.x * 1e-3f + 0.5f, a1 = a0 + 0.1f, a2 = a0 + 0.2f, a3 = a0 + 0.3f;
    float a4 = a0 + 0.4f, a5 = a0 + 0.5f, a6 = a0 + 0.6f, a7 = a0 + 0.7f;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    f16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(a0 + i); fb[i] = (_Float16)(a1 - i); }
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {   // 16 independent v_fma_f32 (8 chains x 2)
            REP8(asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %2, %2, %2, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
                 asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %2, %2, %2, %3" : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 1) {   // v_exp_f32 x 32
            REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (MODE == 2) {   // v_rcp_f32 x 32
            REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (MODE == 3) {   // 32 MFMA 16x16x32 f16 on 8 accumulators
            REP8(c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c0, 0, 0, 0);
                 c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c1, 0, 0, 0);
                 c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c2, 0, 0, 0);
                 c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c3, 0, 0, 0);)
        } else if (MODE == 4) {   // 32 MFMA on 2 accumulators (dependent pairs 16 apart)
            REP16(c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c0, 0, 0, 0);
                  c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c1, 0, 0, 0);)
        } else if (MODE == 5) {   // 32 x (MFMA + 2 plain VALU)
            REP8(c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c0, 0, 0, 0);
                 asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %0" : "+v"(a0), "+v"(a1));
                 c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c1, 0, 0, 0);
                 asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %0" : "+v"(a2), "+v"(a3));
                 c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c2, 0, 0, 0);
                 asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %0" : "+v"(a4), "+v"(a5));
                 c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c3, 0, 0, 0);
                 asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %0" : "+v"(a6), "+v"(a7));)
        } else if (MODE == 6) {   // 32 x (MFMA + 4 plain VALU)
            REP8(c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c0, 0, 0, 0);
                 asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %0\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %2" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
                 c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c1, 0, 0, 0);
                 asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %0\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %2" : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
                 c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c2, 0, 0, 0);
                 asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %0\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %2" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
                 c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c3, 0, 0, 0);
                 asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %0\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %2" : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 7) {   // 32 x (MFMA + 1 v_exp + 1 v_rcp)
            REP8(c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c0, 0, 0, 0);
                 asm volatile("v_exp_f32 %0, %0\n v_rcp_f32 %1, %1" : "+v"(a0), "+v"(a1));
                 c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c1, 0, 0, 0);
                 asm volatile("v_exp_f32 %0, %0\n v_rcp_f32 %1, %1" : "+v"(a2), "+v"(a3));
                 c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c2, 0, 0, 0);
                 asm volatile("v_exp_f32 %0, %0\n v_rcp_f32 %1, %1" : "+v"(a4), "+v"(a5));
                 c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c3, 0, 0, 0);
                 asm volatile("v_exp_f32 %0, %0\n v_rcp_f32 %1, %1" : "+v"(a6), "+v"(a7));)
        } else if (MODE == 8) {   // 16 v_pk_fma_f32
            f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %0, %1\n v_pk_fma_f32 %2, %2, %2, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
            a0 = p0[0] + p1[1]; a2 = p2[0] + p3[1];
        } else if (MODE == 9) {   // 32 MFMA 32x32x16 f16? (uses 16-reg acc) skip: plain 16 exp + 16 fma mix
            REP8(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %1, %0\n v_exp_f32 %2, %2\n v_fma_f32 %3, %3, %3, %2" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        }
    }
    long long t1 = clock64();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    for (int i = 0; i < 4; ++i) s += c0[i] + c1[i] + c2[i] + c3[i] + c4[i] + c5[i] + c6[i] + c7[i];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int MODE>
void run(const char* name, int ops_per_iter) {
    float* out; long long* cyc;
    hipMalloc(&out, 1024 * 4); hipMalloc(&cyc, 16 * 8);
    for (int threads : {64, 256, 512, 1024}) {
        const int iters = 2000;
        k<MODE><<<1, threads>>>(out, cyc, 10);
        k<MODE><<<1, threads>>>(out, cyc, iters);
        hipDeviceSynchronize();
        long long h[16];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        long long mx = 0;
        for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? h[w] : mx;
        // clock64 = s_memtime: 100 MHz constant clock on gfx9?  print raw and per-op
        printf("%-34s waves/SIMD %.2f : %8.3f ticks per wave-op (x%d ops), per-SIMD op interval %8.3f ticks\n", name,
               threads / 256.0, (double)mx / iters / ops_per_iter, ops_per_iter,
               (double)mx / iters / ops_per_iter / (threads >= 256 ? threads / 256.0 : 1.0));
    }
    hipFree(out); hipFree(cyc);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int wc = 0; hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0);
    printf("device %s, clockRate %d kHz, wall clock rate %d kHz (clock64 ticks)\n", p.gcnArchName, p.clockRate, wc);
    run<0>("v_fma_f32 (independent)", 16);
    run<8>("v_pk_fma_f32", 16);
    run<1>("v_exp_f32", 32);
    run<2>("v_rcp_f32", 32);
    run<9>("v_exp + v_fma alternating", 32);
    run<3>("mfma 16x16x32 f16, 4 acc", 32);
    run<4>("mfma 16x16x32 f16, 2 acc", 32);
    run<5>("mfma + 2 fma", 32);
    run<6>("mfma + 4 fma", 32);
    run<7>("mfma + exp + rcp", 32);
    return 0;
}
