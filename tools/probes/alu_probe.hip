// Which instruction class goes wrong next to the bf16 3x3 dense convolution?  (tools/soak_alu.py; profiles/round6_determinism_soak.txt)
// Every thread evaluates a function of its thread id twice, per iteration, from the same register inputs and counts the disagreements.
//   mode 0  fused multiply-add chains (v_fma_f32)
//   mode 1  packed fp32 arithmetic (v_pk_mul_f32 / v_pk_add_f32)
//   mode 2  divisions (v_div_scale / v_rcp / v_div_fmas / v_div_fixup)
//   mode 3  comparisons feeding a counter through divergent branches
//   mode 4  a thread-private LDS array written and read back with run-time indices
//   mode 5  atan2f
//   mode 6  v_pk_mul_f32 chains (inline asm: exactly this instruction, with and without neg / op_sel modifiers)
//   mode 7  v_pk_add_f32 chains        mode 8  v_pk_fma_f32 chains        mode 9  v_mul_f32 / v_add_f32 chains (the scalar forms of mode 6 / 7)
//   mode 10-13  mode 7's chain with s_nop 0 / s_nop 3 / one / two independent VALU instructions between dependent v_pk_add_f32
//   mode 16-21  ONE form each: v_pk_mul_f32 plain / neg / op_sel_hi:[1,0] / op_sel:[0,1] op_sel_hi:[1,0] / op_sel:[1,0] op_sel_hi:[0,1]; v_pk_add_f32 op_sel:[0,1] op_sel_hi:[0,1]
//   mode 14  unmodified v_pk_add_f32 only        mode 15  two alternating accumulators (a result is read two instructions later)
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/probes/libalu_probe.so tools/probes/alu_probe.hip
#include <hip/hip_runtime.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__device__ __noinline__ float eval(float a, float b, float c, float d, float *lds) {
    if constexpr (MODE == 0) {
        float x = a;
        for (int i = 0; i < 64; ++i) x = __builtin_fmaf(x, b, c) * 0.5f + d;
        return x;
    } else if constexpr (MODE == 1) {
        f32x2 x = {a, b}, y = {c, d};
        for (int i = 0; i < 64; ++i) { x = x * y + y; y = y * f32x2{0.5f, 0.25f} + x * f32x2{0.125f, 0.0625f}; x = x - y * f32x2{0.75f, 0.5f}; }
        return x.x + x.y + y.x + y.y;
    } else if constexpr (MODE == 2) {
        float x = a;
        for (int i = 0; i < 32; ++i) x = (x * b - c) / (d + x * 0.001f + 3.0f) + a / (b + 2.0f + x * x);
        return x;
    } else if constexpr (MODE == 3) {
        int n = 0;
        float x = a, y = b;
        for (int i = 0; i < 64; ++i) {
            x = x * 1.0009765625f + c * 0.01f; y = y * 0.9990234375f - d * 0.01f;
            if (fminf(x, y) <= fmaxf(c, d) && x * (-y) > 0.f) { ++n; x = -x * 0.5f; }
            else if (fabsf(x - y) < 0.37f) { n += 3; y = y + 0.11f; }
        }
        return (float)n + x;
    } else if constexpr (MODE == 4) {
        int n = 0;
        for (int i = 0; i < 24; ++i) { const float v = a * (float)(i + 1) + b; if (v - floorf(v) < 0.7f) { lds[n * 128] = v; ++n; } }
        for (int k = 1; k < n; ++k) {  // insertion sort, as in the overlap polygon
            const float v = lds[k * 128];
            int m = k;
            while (m > 0 && lds[(m - 1) * 128] > v) { lds[m * 128] = lds[(m - 1) * 128]; --m; }
            lds[m * 128] = v;
        }
        float s = 0.f;
        for (int k = 0; k < n; ++k) s = s * 1.000001f + lds[k * 128] * (float)(k + 1);
        return s;
    } else if constexpr (MODE == 5) {
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += atan2f(a + (float)i * c, b - (float)i * d);
        return s;
    } else if constexpr (MODE == 6) {
        f32x2 x = {a, b}, y = {1.0f + c * 0.001f, 1.0f - d * 0.001f}, z = {0.999f, 1.001f};
        for (int i = 0; i < 128; ++i) {
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y));
            asm volatile("v_pk_mul_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(x) : "v"(x), "v"(z));
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(x) : "v"(x), "v"(y));
            asm volatile("v_pk_mul_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(x) : "v"(x), "v"(z));
        }
        return x.x + x.y;
    } else if constexpr (MODE == 7) {
        f32x2 x = {a, b}, y = {c * 0.01f, -d * 0.01f};
        for (int i = 0; i < 128; ++i) {
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y));
            asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(x) : "v"(x), "v"(y));
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1]" : "=v"(x) : "v"(x), "v"(y));
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(x) : "v"(x), "v"(y));
        }
        return x.x + x.y;
    } else if constexpr (MODE == 8) {
        f32x2 x = {a, b}, y = {1.0f + c * 0.001f, 1.0f - d * 0.001f}, z = {0.001f, -0.001f};
        for (int i = 0; i < 256; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(x) : "v"(x), "v"(y), "v"(z));
        return x.x + x.y;
    } else if constexpr (MODE >= 10 && MODE <= 13) {  // the v_pk_add_f32 chain of mode 7 with something between two dependent instructions
        f32x2 x = {a, b}, y = {c * 0.01f, -d * 0.01f};
        float w = c;
        for (int i = 0; i < 256; ++i) {
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y));
            if constexpr (MODE == 10) asm volatile("s_nop 0");
            if constexpr (MODE == 11) asm volatile("s_nop 3");
            if constexpr (MODE == 12) asm volatile("v_add_f32 %0, %1, %1" : "=v"(w) : "v"(w));   // an independent VALU instruction
            if constexpr (MODE == 13) asm volatile("v_add_f32 %0, %1, %1\n\tv_add_f32 %0, %0, %1" : "=v"(w) : "v"(w));
            asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(x) : "v"(x), "v"(y));
            if constexpr (MODE == 10) asm volatile("s_nop 0");
            if constexpr (MODE == 11) asm volatile("s_nop 3");
            if constexpr (MODE == 12) asm volatile("v_add_f32 %0, %1, %1" : "=v"(w) : "v"(w));
            if constexpr (MODE == 13) asm volatile("v_add_f32 %0, %1, %1\n\tv_add_f32 %0, %0, %1" : "=v"(w) : "v"(w));
        }
        return x.x + x.y + (w == 12345.f ? 1.f : 0.f);
    } else if constexpr (MODE >= 16 && MODE <= 21) {  // one form per mode, 512 dependent instructions
        f32x2 x = {a, b}, y = {1.0f + c * 0.0001f, 1.0f - d * 0.0001f};
        for (int i = 0; i < 512; ++i) {
            if constexpr (MODE == 16) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y));
            if constexpr (MODE == 17) asm volatile("v_pk_mul_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(x) : "v"(x), "v"(y));
            if constexpr (MODE == 18) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(x) : "v"(x), "v"(y));
            if constexpr (MODE == 19) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(x) : "v"(x), "v"(y));
            if constexpr (MODE == 20) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(x) : "v"(x), "v"(y));
            if constexpr (MODE == 21) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1]" : "=v"(x) : "v"(x), "v"(y));
        }
        return x.x + x.y;
    } else if constexpr (MODE == 14) {  // plain (unmodified) v_pk_add_f32 only
        f32x2 x = {a, b}, y = {c * 0.01f, -d * 0.01f};
        for (int i = 0; i < 512; ++i) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y));
        return x.x + x.y;
    } else if constexpr (MODE == 15) {  // v_pk_add_f32 whose result is NOT the next one's input (two alternating accumulators: distance 2)
        f32x2 x = {a, b}, x2 = {b, a}, y = {c * 0.01f, -d * 0.01f};
        for (int i = 0; i < 256; ++i) {
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(x2) : "v"(x2), "v"(y));
        }
        return x.x + x.y + x2.x + x2.y;
    } else {
        float x = a, y = 1.0f + c * 0.001f, z = d * 0.01f;
        for (int i = 0; i < 256; ++i) {
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(z));
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(z));
        }
        return x;
    }
}

template <int MODE>
__global__ void __launch_bounds__(128) probe(int iters, unsigned long long *mismatch, float *sink) {
    __shared__ float s_arr[24 * 128];
    const int t = blockIdx.x * 128 + threadIdx.x;
    float a = 0.37f + (float)(t % 977) * 0.0113f, b = 1.21f - (float)(t % 131) * 0.0071f, c = 0.05f + (float)(t % 17) * 0.031f, d = 0.9f + (float)(t % 29) * 0.013f;
    unsigned long long bad = 0;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        const float r1 = eval<MODE>(a, b, c, d, s_arr + threadIdx.x);
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        const float r2 = eval<MODE>(a, b, c, d, s_arr + threadIdx.x);
        bad += __float_as_uint(r1) != __float_as_uint(r2);
        acc += r1;
        a += 0.001f;
    }
    if (bad) atomicAdd(mismatch, bad);
    if (acc == 123.456f) sink[0] = acc;
}

// ---- synthetic disturbers: waves that issue ONE kind of instruction back to back (no memory traffic), to find what the victim needs next to it
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ void __launch_bounds__(256) disturb(int iters, float *sink) {
    const float t = (float)threadIdx.x * 1e-3f;
    if constexpr (KIND == 0) {  // v_mfma_f32_32x32x16_bf16 (the bf16 dense convolution's)
        bf16x8 a, b;
        for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(t + k); b[k] = (__bf16)(1.f - t * k); }
        f32x16 c0 = {}, c1 = {};
        for (int i = 0; i < iters; ++i) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0); }
        if (c0[0] + c1[3] == 123.f) sink[0] = c0[1];
    } else if constexpr (KIND == 1) {  // v_mfma_f32_16x16x32_bf16 (the bf16 sparse convolutions')
        bf16x8 a, b;
        for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(t + k); b[k] = (__bf16)(1.f - t * k); }
        f32x4v c0 = {}, c1 = {};
        for (int i = 0; i < iters; ++i) { c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c1, 0, 0, 0); }
        if (c0[0] + c1[3] == 123.f) sink[0] = c0[1];
    } else if constexpr (KIND == 2) {  // v_mfma_f32_32x32x2_f32 / 16x16x4_f32 (the fp32 kernels')
        f32x16 c0 = {};
        f32x4v c1 = {};
        for (int i = 0; i < iters; ++i) { c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(t, 1.f - t, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f + t, t, c1, 0, 0, 0); }
        if (c0[0] + c1[3] == 123.f) sink[0] = c0[1];
    } else {  // plain VALU work (v_fma_f32 chains)
        float x = t, y = 1.f - t;
        for (int i = 0; i < iters * 8; ++i) { x = __builtin_fmaf(x, 0.999f, y); y = __builtin_fmaf(y, 1.001f, -x); }
        if (x + y == 123.f) sink[0] = x;
    }
}
extern "C" int alu_disturb_run(int kind, int blocks, int iters, float *sink, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    switch (kind) {
        case 0: disturb<0><<<blocks, 256, 0, s>>>(iters, sink); break;
        case 1: disturb<1><<<blocks, 256, 0, s>>>(iters, sink); break;
        case 2: disturb<2><<<blocks, 256, 0, s>>>(iters, sink); break;
        default: disturb<3><<<blocks, 256, 0, s>>>(iters, sink); break;
    }
    return (int)hipGetLastError();
}

extern "C" int alu_probe_run(int mode, int blocks, int iters, unsigned long long *mismatch, float *sink, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    switch (mode) {
        case 0: probe<0><<<blocks, 128, 0, s>>>(iters, mismatch, sink); break;
        case 1: probe<1><<<blocks, 128, 0, s>>>(iters, mismatch, sink); break;
        case 2: probe<2><<<blocks, 128, 0, s>>>(iters, mismatch, sink); break;
        case 3: probe<3><<<blocks, 128, 0, s>>>(iters, mismatch, sink); break;
        case 4: probe<4><<<blocks, 128, 0, s>>>(iters, mismatch, sink); break;
        case 5: probe<5><<<blocks, 128, 0, s>>>(iters, mismatch, sink); break;
        case 6: probe<6><<<blocks, 128, 0, s>>>(iters, mismatch, sink); break;
        case 7: probe<7><<<blocks, 128, 0, s>>>(iters, mismatch, sink); break;
        case 8: probe<8><<<blocks, 128, 0, s>>>(iters, mismatch, sink); break;
        case 9: probe<9><<<blocks, 128, 0, s>>>(iters, mismatch, sink); break;
        case 10: probe<10><<<blocks, 128, 0, s>>>(iters, mismatch, sink); break;
        case 11: probe<11><<<blocks, 128, 0, s>>>(iters, mismatch, sink); break;
        case 12: probe<12><<<blocks, 128, 0, s>>>(iters, mismatch, sink); break;
        case 13: probe<13><<<blocks, 128, 0, s>>>(iters, mismatch, sink); break;
        case 14: probe<14><<<blocks, 128, 0, s>>>(iters, mismatch, sink); break;
        case 15: probe<15><<<blocks, 128, 0, s>>>(iters, mismatch, sink); break;
        case 16: probe<16><<<blocks, 128, 0, s>>>(iters, mismatch, sink); break;
        case 17: probe<17><<<blocks, 128, 0, s>>>(iters, mismatch, sink); break;
        case 18: probe<18><<<blocks, 128, 0, s>>>(iters, mismatch, sink); break;
        case 19: probe<19><<<blocks, 128, 0, s>>>(iters, mismatch, sink); break;
        case 20: probe<20><<<blocks, 128, 0, s>>>(iters, mismatch, sink); break;
        default: probe<21><<<blocks, 128, 0, s>>>(iters, mismatch, sink); break;
    }
    return (int)hipGetLastError();
}
