// f32 VALU issue rate on gfx950, measured by WALL time over the whole chip (hipEvents), not s_memtime:
// plain v_fma_f32 vs v_pk_fma_f32, N independent accumulators per lane, 1 / 2 / 4 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/fma_rate.hip -o tools/ubench/fma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE, int NACC>
__global__ void k(float* out, int iters) {
    float a[16];
    f32x2 p[8];
    const float x = threadIdx.x * 1e-9f + 1.0f, y = 1e-9f;
    for (int i = 0; i < 16; ++i) a[i] = i * 1e-3f;
    for (int i = 0; i < 8; ++i) p[i] = (f32x2){i * 1e-3f, i * 2e-3f};
    const f32x2 xx = {x, x}, yy = {y, y};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < NACC; ++i) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(x), "v"(y));
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < NACC; ++i) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(p[i]) : "v"(xx), "v"(yy));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a[i];
    for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
    if (s == 123.456f) out[0] = s;
}

template <int MODE, int NACC>
void run(const char* name) {
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {256, 512, 1024}) {
        const int iters = 20000, grid = 256;
        k<MODE, NACC><<<grid, threads>>>(out, 100);
        hipEventRecord(e0);
        k<MODE, NACC><<<grid, threads>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double ops = (double)iters * 4 * NACC;                       // wave-instructions per wave
        const double flop = ops * threads * grid * (MODE ? 4.0 : 2.0);
        // per-SIMD issue interval assuming 2.4 GHz: time * 2.4e9 / (ops * waves_per_simd)
        const double wps = threads / 256.0;
        printf("%-14s acc %2d waves/SIMD %.0f : %7.2f TFLOP/s, %5.2f clk@2.4GHz per SIMD instruction\n", name, NACC, wps,
               flop / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (ops * wps));
    }
}

int main() {
    run<0, 4>("v_fma_f32");
    run<0, 8>("v_fma_f32");
    run<0, 16>("v_fma_f32");
    run<1, 2>("v_pk_fma_f32");
    run<1, 4>("v_pk_fma_f32");
    run<1, 8>("v_pk_fma_f32");
    return 0;
}
