// Probe: issue rate of v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16 with NACC independent accumulators (a chain of
// dependent MFMAs per accumulator, CH consecutive MFMAs on one accumulator before moving on), W waves per SIMD, random-ish data.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC, int CH, bool BIG>
__global__ void __launch_bounds__(256) probe(float *out, int iters, float a0, float b0) {
    using Acc = typename std::conditional<BIG, f32x16, f32x4>::type;
    Acc acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < (BIG ? 16 : 4); ++j) acc[i][j] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(a0 + 0.37f * ((threadIdx.x * 7 + j * 3) % 11) - 1.5f); b[j] = (__bf16)(b0 + 0.21f * ((threadIdx.x * 5 + j) % 13) - 1.2f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 48 / (NACC * CH); ++m)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    if constexpr (BIG) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                    else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
                }
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < (BIG ? 16 : 4); ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int CH, bool BIG>
void run(int blocks_per_cu) {
    float *out;
    hipMalloc(&out, 256 * 256 * 8 * 4);
    int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NACC, CH, BIG><<<256 * blocks_per_cu, 256>>>(out, 10, 1.f, 1.f);
    hipEventRecord(e0);
    probe<NACC, CH, BIG><<<256 * blocks_per_cu, 256>>>(out, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double mfma_per_simd = (double)iters * 48 * blocks_per_cu;  // each block: 4 waves, one per SIMD
    double flop = (BIG ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32);
    double tf = flop * iters * 48 * 4 * 256 * blocks_per_cu / (ms * 1e-3) / 1e12;
    printf("%s NACC=%d chain=%d waves/SIMD=%d: %.3f ms, %.1f ns per MFMA per SIMD, %.0f TFLOP/s\n", BIG ? "32x32x16" : "16x16x32", NACC, CH, blocks_per_cu, ms,
           ms * 1e6 / mfma_per_simd, tf);
    hipFree(out);
}
int main() {
    for (int w = 1; w <= 2; ++w) {
        run<1, 1, true>(w); run<2, 1, true>(w); run<4, 1, true>(w); run<4, 6, true>(w); run<4, 3, true>(w); run<8, 1, true>(w);
        run<1, 1, false>(w); run<2, 1, false>(w); run<4, 1, false>(w); run<4, 6, false>(w); run<8, 1, false>(w);
    }
    return 0;
}
