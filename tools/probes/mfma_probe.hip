// Probe: issue rate of v_mfma_f32_16x16x4_f32 with NACC independent accumulators, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(256) probe(float *out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 64 / NACC; ++m)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu) {
    float *out;
    hipMalloc(&out, 256 * 256 * 8 * 4);
    int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NACC><<<256 * blocks_per_cu, 256>>>(out, 10, 1.f, 1.f);
    hipEventRecord(e0);
    probe<NACC><<<256 * blocks_per_cu, 256>>>(out, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double mfma_per_simd = (double)iters * 64 * blocks_per_cu;  // each block: 4 waves, one per SIMD
    double tf = 2048.0 * iters * 64 * 4 * 256 * blocks_per_cu / (ms * 1e-3) / 1e12;
    printf("NACC=%d waves/SIMD=%d: %.3f ms, %.1f ns per MFMA per SIMD (32 cyc @2.4GHz = 13.3 ns), %.1f TFLOP/s\n", NACC, blocks_per_cu, ms,
           ms * 1e6 / mfma_per_simd, tf);
    hipFree(out);
}
int main() {
    for (int w = 1; w <= 4; w *= 2) { run<1>(w); run<2>(w); run<4>(w); run<8>(w); }
    return 0;
}
