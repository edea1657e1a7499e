// v_mfma_f32_32x32x16_f16: issue interval vs the distance between two MFMAs on the SAME accumulator
// (NACC accumulators used round-robin), at 1 and 2 waves per SIMD; wall time over the whole chip.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_dep.hip -o tools/ubench/mfma_dep
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 1e-3f + i); b[i] = (_Float16)(1.f - i * 0.1f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 12 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
}

template <int NACC>
void run() {
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {256, 512}) {
        const int iters = 20000, grid = 256;
        k<NACC><<<grid, threads>>>(out, 100);
        hipEventRecord(e0);
        k<NACC><<<grid, threads>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)iters * 12;                    // MFMAs per wave
        const double wps = threads / 256.0;
        const double flop = n * (threads / 64) * grid * 2.0 * 32 * 32 * 16;
        printf("accumulators %d (same-accumulator distance %d) waves/SIMD %.0f : %7.1f TFLOP/s, %6.1f clk@2.4GHz per MFMA per SIMD\n",
               NACC, NACC, wps, flop / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (n * wps));
    }
}

int main() {
    run<1>(); run<2>(); run<3>(); run<4>(); run<6>();
    return 0;
}
