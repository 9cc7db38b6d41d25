// Gather cost model of the gfx950 vector-memory path (tuning probe, not part of the product).
// Every wave issues REPS x UNROLL buffer_load_dwordx4 with a controlled lane -> address pattern over a window of a given size
// and reports bytes per clock per CU.  Patterns (row = 64 B unless stated):
//   0 linear     lane l -> base + 16 l                                   (1 KB contiguous per instruction)
//   1 quadrow    lane 4p+q -> row(p) * 64 + 16 q, 16 random rows         (each quad = one full row)
//   2 mfma       lane p+16q -> row(p) * 64 + 16 q, 16 random rows        (MFMA fragment order: adjacent lanes = different rows)
//   3 mfma_seq   as 2, rows consecutive (row(p) = r0 + p, r0 random)
//   4 mfma_half  as 2, odd rows out of range (bounds-checked zeros)
//   5 quad_half  as 1, odd rows out of range
//   6 mfma256    lane p+16q -> row(p) * 256 + 16 q (256-byte rows, first 64 B piece), random rows
//   8 all_oob    every lane out of range;  9 one_lane: lane 0 loads, the other 63 are out of range
//   10 mfma_half_masked  as 4, but the odd rows' lanes are switched off (exec mask) instead of being sent out of range
//   11 mfma256_half_oob  as 6, odd rows out of range;  12 mfma256_half_masked  as 6, odd rows' lanes switched off
//   13 mfma_quarter_masked  as 2, three of four rows' lanes switched off
//   7 blocked    lane p+16q -> (r0 + p) * 16 + q * 256 within a 1 KB block: chunk-major blocked layout, consecutive rows
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/gather_probe tools/probes/gather_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool MASKED = false>
__global__ void __launch_bounds__(1024) probe(const unsigned char *buf, unsigned bytes, const unsigned *offs, int reps, unsigned *sink, long long *cycles) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(buf), 0, (int)bytes, 0x00020000);
    const int lane = threadIdx.x & 63;
    const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned *o = offs + ((size_t)gw * reps) * 64 * UNROLL + lane;
    u32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        unsigned v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = o[(r * UNROLL + u) * 64];
        u32x4 d[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if constexpr (MASKED) {
                d[u] = (u32x4){0u, 0u, 0u, 0u};
                if (v[u] != 0xffffffffu) d[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, v[u], 0, 0);  // exec-masked: the other lanes issue nothing
            } else {
                d[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, v[u], 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= d[u];
    }
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cycles[gw] = t1 - t0;
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

int main(int argc, char **argv) {
    const int waves_per_cu = argc > 1 ? atoi(argv[1]) : 16;
    const int reps = 64, UNROLL = 8;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    const int wg_waves = waves_per_cu > 16 ? 16 : waves_per_cu;
    const int n_wg = n_cu * (waves_per_cu / wg_waves);
    const int n_waves = n_wg * wg_waves;
    const size_t max_bytes = 512u << 20;
    unsigned char *buf;
    hipMalloc(&buf, max_bytes);
    hipMemset(buf, 1, max_bytes);
    unsigned *offs, *sink;
    long long *cyc;
    const size_t n_off = (size_t)n_waves * reps * UNROLL * 64;
    hipMalloc(&offs, n_off * 4);
    hipMalloc(&sink, 4);
    hipMalloc(&cyc, n_waves * 8);
    std::vector<unsigned> h(n_off);
    std::vector<long long> hc(n_waves);
    const char *names[] = {"linear", "quadrow", "mfma", "mfma_seq", "mfma_half", "quad_half", "mfma256", "blocked", "all_oob", "one_lane",
                           "mfma_half_masked", "mfma256_half_oob", "mfma256_half_masked", "mfma_quarter_masked"};
    const size_t windows[] = {16u << 10, 2u << 20, 24u << 20};
    printf("%d CUs, %d waves per CU, %d loads of 1 KB per wave\n", n_cu, waves_per_cu, reps * UNROLL);
    for (size_t win : windows) {
        for (int pat = 0; pat < 14; ++pat) {
            // each CU's waves work in their own window slice when the window is small (L1 case), else share the whole window
            srand(1234 + pat);
            for (int w = 0; w < n_waves; ++w) {
                const size_t base = win <= (64u << 10) ? ((size_t)(w / wg_waves) * win) % (max_bytes - win) : 0;
                for (int i = 0; i < reps * UNROLL; ++i) {
                    unsigned rows[16];
                    const unsigned rb = (pat == 6 || pat == 11 || pat == 12) ? 256 : 64;
                    const unsigned nrows = (unsigned)(win / rb);
                    const unsigned r0 = (unsigned)(rand() % (nrows - 16));
                    for (int p = 0; p < 16; ++p) rows[p] = (pat == 3 || pat == 7) ? r0 + p : (unsigned)(rand() % nrows);
                    for (int l = 0; l < 64; ++l) {
                        unsigned off;
                        switch (pat) {
                            case 0: off = (unsigned)base + (r0 & ~15u) * 64 + 16 * l; break;
                            case 1: off = (unsigned)base + rows[l >> 2] * 64 + 16 * (l & 3); break;
                            case 2: case 3: off = (unsigned)base + rows[l & 15] * 64 + 16 * (l >> 4); break;
                            case 4: off = (l & 1) ? 0xffffff00u : (unsigned)base + rows[l & 15] * 64 + 16 * (l >> 4); break;
                            case 5: off = ((l >> 2) & 1) ? 0xffffff00u : (unsigned)base + rows[l >> 2] * 64 + 16 * (l & 3); break;
                            case 6: off = (unsigned)base + rows[l & 15] * 256 + 16 * (l >> 4); break;
                            case 8: off = 0xffffff00u; break;                                                         // every lane out of range
                            case 10: off = (l & 1) ? 0xffffffffu : (unsigned)base + rows[l & 15] * 64 + 16 * (l >> 4); break;
                            case 11: off = (l & 1) ? 0xffffff00u : (unsigned)base + rows[l & 15] * 256 + 16 * (l >> 4); break;
                            case 12: off = (l & 1) ? 0xffffffffu : (unsigned)base + rows[l & 15] * 256 + 16 * (l >> 4); break;
                            case 13: off = (l & 3) ? 0xffffffffu : (unsigned)base + rows[l & 15] * 64 + 16 * (l >> 4); break;
                            case 9: off = l == 0 ? (unsigned)base + rows[0] * 64 : 0xffffff00u; break;               // one lane loads, 63 out of range
                            default: off = (unsigned)base + (r0 >> 4) * 1024 + ((r0 & 15) + (l & 15)) * 16 + (l >> 4) * 256; break;  // may run into the next block: still contiguous per chunk
                        }
                        h[((size_t)w * reps * UNROLL + i) * 64 + l] = off;
                    }
                }
            }
            hipMemcpy(offs, h.data(), n_off * 4, hipMemcpyHostToDevice);
            const unsigned bytes = (unsigned)(max_bytes - 1);
            const bool masked = pat == 10 || pat == 12 || pat == 13;
            auto kern = masked ? probe<UNROLL, true> : probe<UNROLL, false>;
            for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kern, dim3(n_wg), dim3(wg_waves * 64), 0, 0, buf, bytes, offs, reps, sink, cyc);
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kern, dim3(n_wg), dim3(wg_waves * 64), 0, 0, buf, bytes, offs, reps, sink, cyc);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(hc.data(), cyc, n_waves * 8, hipMemcpyDeviceToHost);
            double mean = 0;
            for (int w = 0; w < n_waves; ++w) mean += (double)hc[w];
            mean /= n_waves;
            const double kb = (double)reps * UNROLL;  // KB per wave
            const double frac = (pat == 4 || pat == 5 || pat == 10 || pat == 11 || pat == 12) ? 0.5 : (pat == 8 ? 0.0 : (pat == 9 ? 1.0 / 64 : (pat == 13 ? 0.25 : 1.0)));
            printf("window %7.1f MB  %-10s  %7.1f cycles per 1-KB load per wave  -> %6.1f B/clk/CU (requested %s)  kernel %.1f us  %.2f TB/s\n", win / 1048576.0, names[pat],
                   mean / kb, 1024.0 * frac * waves_per_cu / (mean / kb), frac < 1 ? "half" : "all", ms * 1e3,
                   (double)n_waves * kb * 1024 * frac / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
