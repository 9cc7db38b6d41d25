// Stand-alone reproducer of the round-6 determinism finding (profiles/round6_determinism_soak.txt, section 8); no part of the product.
// VICTIM: every thread runs a chain of 512 dependent packed-fp32 instructions twice from the same registers and compares the two results bit for
// bit -- once with an op_sel swizzle of src1 (v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[1,0]), once plain.
// DISTURBER (three other streams): waves that issue one kind of MFMA back to back from registers -- no memory traffic, no LDS.
// On MI355X the swizzled chain disagrees with itself next to v_mfma_f32_16x16x32_bf16 (less often next to 32x32x16_bf16), never alone, never next to
// the fp32 MFMAs; the plain chain never does.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/pk_opsel_repro tools/probes/pk_opsel_repro.hip && tools/probes/pk_opsel_repro
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool SWIZZLE>
__device__ __noinline__ float chain(float a, float b, float c, float d) {
    f32x2 x = {a, b}, y = {1.0f + c * 0.0001f, 1.0f - d * 0.0001f};
    for (int i = 0; i < 512; ++i) {
        if constexpr (SWIZZLE) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(x) : "v"(x), "v"(y));
        else asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y));
    }
    return x.x + x.y;
}

template <bool SWIZZLE>
__global__ void __launch_bounds__(128) victim(int iters, unsigned long long *mismatch, unsigned long long *lanes48) {
    const int t = blockIdx.x * 128 + threadIdx.x;
    float a = 0.37f + (float)(t % 977) * 0.0113f, b = 1.21f - (float)(t % 131) * 0.0071f, c = 0.05f + (float)(t % 17) * 0.031f, d = 0.9f + (float)(t % 29) * 0.013f;
    unsigned long long bad = 0;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        const float r1 = chain<SWIZZLE>(a, b, c, d);
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        const float r2 = chain<SWIZZLE>(a, b, c, d);
        bad += __float_as_uint(r1) != __float_as_uint(r2);
        a += 0.001f;
    }
    if (bad) {
        atomicAdd(mismatch, bad);
        if ((threadIdx.x & 63) >= 48) atomicAdd(lanes48, bad);
    }
}

template <int KIND>
__global__ void __launch_bounds__(256) disturber(int iters, float *sink) {
    const float t = (float)threadIdx.x * 1e-3f;
    bf16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(t + k); b[k] = (__bf16)(1.f - t * k); }
    f32x16 c16 = {};
    f32x4 c4 = {};
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0) { c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4, 0, 0, 0); c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c4, 0, 0, 0); }
        if constexpr (KIND == 1) { c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c16, 0, 0, 0); c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c16, 0, 0, 0); }
        if constexpr (KIND == 2) { c16 = __builtin_amdgcn_mfma_f32_32x32x2f32(t, 1.f - t, c16, 0, 0, 0); c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f + t, t, c4, 0, 0, 0); }
    }
    if (c16[0] + c4[3] == 123.f) sink[0] = c16[1];
}

// the same inside ONE kernel: waves 0-1 of a workgroup run the chain, waves 2-3 issue v_mfma_f32_16x16x32_bf16 back to back
template <bool SWIZZLE>
__global__ void __launch_bounds__(256) mixed(int iters, unsigned long long *mismatch) {
    const int wave = threadIdx.x >> 6;
    if (wave >= 2) {
        const float t = (float)threadIdx.x * 1e-3f;
        bf16x8 a, b;
        for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(t + k); b[k] = (__bf16)(1.f - t * k); }
        f32x4 c4 = {};
        for (int i = 0; i < iters * 700; ++i) { c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4, 0, 0, 0); c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c4, 0, 0, 0); }
        if (c4[3] == 123.f) mismatch[1] = 1;
        return;
    }
    const int t = blockIdx.x * 128 + threadIdx.x;
    float a = 0.37f + (float)(t % 977) * 0.0113f, b = 1.21f - (float)(t % 131) * 0.0071f, c = 0.05f + (float)(t % 17) * 0.031f, d = 0.9f + (float)(t % 29) * 0.013f;
    unsigned long long bad = 0;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        const float r1 = chain<SWIZZLE>(a, b, c, d);
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        const float r2 = chain<SWIZZLE>(a, b, c, d);
        bad += __float_as_uint(r1) != __float_as_uint(r2);
        a += 0.001f;
    }
    if (bad) atomicAdd(mismatch, bad);
}

int main() {
    hipStream_t sv, sd[3];
    hipStreamCreate(&sv);
    for (auto &s : sd) hipStreamCreate(&s);
    unsigned long long *cnt;
    float *sink;
    hipMalloc(&cnt, 16);
    hipMalloc(&sink, 16);
    const char *dn[] = {"nothing", "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_32x32x16_bf16", "fp32 MFMAs (32x32x2 + 16x16x4)"};
    printf("%-34s %-28s %-28s\n", "disturber (3 streams)", "swizzled chain: mismatches", "plain chain: mismatches");
    for (int kind = -1; kind < 3; ++kind) {
        unsigned long long res[2][2] = {{0, 0}, {0, 0}};
        for (int sw = 0; sw < 2; ++sw) {
            hipMemset(cnt, 0, 16);
            hipDeviceSynchronize();
            for (int round = 0; round < 100; ++round) {
                for (auto &s : sd) {
                    if (kind == 0) disturber<0><<<512, 256, 0, s>>>(20000, sink);
                    if (kind == 1) disturber<1><<<512, 256, 0, s>>>(20000, sink);
                    if (kind == 2) disturber<2><<<512, 256, 0, s>>>(20000, sink);
                }
                for (int l = 0; l < 6; ++l) {
                    if (sw == 0) victim<true><<<128, 128, 0, sv>>>(24, cnt, cnt + 1);
                    else victim<false><<<128, 128, 0, sv>>>(24, cnt, cnt + 1);
                }
                hipDeviceSynchronize();
            }
            hipMemcpy(res[sw], cnt, 16, hipMemcpyDeviceToHost);
        }
        printf("%-34s %10llu (%llu in lanes 48-63)   %10llu\n", dn[kind + 1], res[0][0], res[0][1], res[1][0]);
    }
    printf("(per cell: 100 rounds x 6 launches x 16384 threads x 24 double evaluations = 236 M)\n");
    for (int sw = 0; sw < 2; ++sw) {  // one kernel, one stream: chain waves and MFMA waves in the same workgroups
        hipMemset(cnt, 0, 16);
        for (int round = 0; round < 100; ++round) {
            if (sw == 0) mixed<true><<<1024, 256, 0, sv>>>(24, cnt);
            else mixed<false><<<1024, 256, 0, sv>>>(24, cnt);
            hipDeviceSynchronize();
        }
        unsigned long long r[2];
        hipMemcpy(r, cnt, 16, hipMemcpyDeviceToHost);
        printf("one kernel, waves 0-1 chain / waves 2-3 v_mfma_f32_16x16x32_bf16, %s chain: %llu mismatches of %llu double evaluations\n", sw == 0 ? "swizzled" : "plain", r[0],
               100ull * 1024 * 128 * 24);
    }
    return 0;
}
