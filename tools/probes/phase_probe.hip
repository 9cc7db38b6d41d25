// Probe: W waves per SIMD alternate a non-MFMA phase (dependent LDS round trips + NV VALU ops) and an MFMA phase
// (NM x v_mfma_f32_16x16x4_f32 on two alternating accumulators).  How full does the MFMA pipe get?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NM, int NV, int NL, bool PHASED>
__global__ void __launch_bounds__(256) probe(float *out, int iters, float a0, float b0) {
    __shared__ float s[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) s[i] = (float)(i & 63);
    __syncthreads();
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    float a = a0 + threadIdx.x, b = b0;
    int idx = threadIdx.x;
    float v = 1.0f;
    for (int it = 0; it < iters; ++it) {
        // non-MFMA phase: NL dependent LDS round trips, NV dependent VALU ops
#pragma unroll
        for (int l = 0; l < NL; ++l) idx = ((int)s[idx & 4095] + idx + 1) & 4095;
#pragma unroll
        for (int k = 0; k < NV; ++k) v = v * 1.0001f + (float)idx;
        if (PHASED) __builtin_amdgcn_sched_barrier(0);
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
#pragma unroll
        for (int m = 0; m < NM / 2; ++m) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0);
        }
        if (PHASED) __builtin_amdgcn_sched_barrier(0);
        acc0 += c0; acc1 += c1;
        a += v * 1e-20f;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[1] + v + idx;
}
template <int NM, int NV, int NL, bool PHASED>
void run(int w) {
    float *out;
    hipMalloc(&out, 256 * 256 * 8 * 4);
    int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NM, NV, NL, PHASED><<<256 * w, 256>>>(out, 10, 1.f, 1.f);
    hipEventRecord(e0);
    probe<NM, NV, NL, PHASED><<<256 * w, 256>>>(out, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double ns_item = ms * 1e6 / iters;                 // per item per wave (all waves concurrent)
    double mfma_ns = 32.0 / 2.4 * NM * w;              // pipe time needed per round at 2.4 GHz
    printf("NM=%2d NV=%3d NL=%d phased=%d waves/SIMD=%d: %.0f ns per item-round, MFMA pipe needs %.0f ns -> util %.2f\n", NM, NV, NL,
           (int)PHASED, w, ns_item, mfma_ns, mfma_ns / ns_item);
    hipFree(out);
}
int main() {
    for (int w = 1; w <= 4; ++w) run<16, 40, 3, true>(w);
    for (int w = 1; w <= 4; ++w) run<16, 40, 3, false>(w);
    for (int w = 1; w <= 3; ++w) run<64, 40, 3, true>(w);
    for (int w = 1; w <= 4; ++w) run<16, 0, 0, true>(w);
    for (int w = 1; w <= 4; ++w) run<16, 40, 0, true>(w);
    for (int w = 1; w <= 4; ++w) run<16, 0, 3, true>(w);
    return 0;
}
