// Sparse convolution apply for gfx950: output-stationary gather + MFMA GEMM, no scatter, no atomics.
//
// Replaces spconv 1.0's indice_conv / indice_subm_conv (gather -> per-tap cuBLAS GEMM -> scatter-add) used by
// the 21 convolutions of det3d/models/backbones/scn.py:99-141, fused with the folded BatchNorm1d, the residual
// add and the ReLU that follow them (scn.py:67-78).
//
//   out[o,:] = act( sum_k in[nbr[k][o],:] @ W[k] + bias (+ residual[o,:]) )
//
// Work decomposition: a wave owns 16*RG consecutive output rows (rows are spatially sorted by the index, so
// the rows a wave gathers are close in memory) and all COUT columns; accumulators stay in registers for all K
// taps.  The rulebook tile of the workgroup ([K][64*RG] int32) is staged once through LDS.  Per tap and per
// 16-row group a wave-wide ballot skips the MFMAs when no row of the group has that neighbour.  A operands are
// gathered straight into MFMA fragment layout with one 16-byte load per lane (row = lane&15, 16-byte chunk =
// lane>>4); B operands come from weights pre-packed in fragment order, one coalesced 1 KiB load per wave
// instruction (L2-resident: at most 1.8 MB per layer).  fp32 uses v_mfma_f32_16x16x4_f32 (exact fp32 fma
// chain), bf16 uses v_mfma_f32_16x16x32_bf16 / 16x16x16 with fp32 accumulation.
#include "fd_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned short f2bf(float v) {
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ inline float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

constexpr int kMaxTaps = 27;

// ------------------------------------------------------------------------------------------------- fp32
template <int CIN, int COUT, int RG>
__global__ void __launch_bounds__(256) spconv_f32(const float *__restrict__ in, const float4 *__restrict__ wp,
                                                  const float *__restrict__ bias, const float *__restrict__ residual, int relu,
                                                  const int *__restrict__ nbr, int64_t nbr_stride, int K, int n_out,
                                                  const int *__restrict__ n_out_dev, float *__restrict__ out) {
    constexpr int ROWS_B = 64 * RG;  // rows per workgroup
    constexpr int NB = COUT / 16, NC = CIN / 16;
    __shared__ int s_nbr[kMaxTaps * ROWS_B];
    const int tile = blockIdx.x;  // index order: contiguous XCD chunks concentrate the dense regions on a few XCDs (measured -10%)
    const int row0 = tile * ROWS_B;
    n_out = fd::device_count(n_out, n_out_dev);  // capacity launch (fd_common.h): workgroups past the device's count leave
    if (row0 >= n_out) return;
    for (int t = threadIdx.x; t < K * ROWS_B; t += 256) {
        int k = t / ROWS_B, r = t - k * ROWS_B;
        int64_t o = (int64_t)row0 + r;
        s_nbr[t] = (o < n_out) ? nbr[(int64_t)k * nbr_stride + o] : -1;  // rows >= n_out of the table are never read
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lrow = lane & 15, lq = lane >> 4;
    const int wrow = wave * 16 * RG;
    if (row0 + wrow >= n_out) return;

    f32x4 acc[RG][NB];
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[g][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int k = 0; k < K; ++k) {
        int idx[RG];
        bool any[RG];
        bool any_all = false;
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            idx[g] = s_nbr[k * ROWS_B + wrow + g * 16 + lrow];
            any[g] = __ballot(idx[g] >= 0) != 0ull;
            any_all = any_all || any[g];
        }
        if (!any_all) continue;
        const float4 *wk = wp + (int64_t)k * NC * NB * 64 + lane;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float4 a[RG];
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                a[g] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx[g] >= 0) a[g] = *reinterpret_cast<const float4 *>(in + (int64_t)idx[g] * CIN + c * 16 + lq * 4);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float4 b = wk[(c * NB + nb) * 64];
#pragma unroll
                for (int g = 0; g < RG; ++g) {
                    if (any[g]) {
                        acc[g][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].x, b.x, acc[g][nb], 0, 0, 0);
                        acc[g][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].y, b.y, acc[g][nb], 0, 0, 0);
                        acc[g][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].z, b.z, acc[g][nb], 0, 0, 0);
                        acc[g][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].w, b.w, acc[g][nb], 0, 0, 0);
                    }
                }
            }
        }
    }
    // epilogue: C/D layout col = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
    for (int g = 0; g < RG; ++g) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int col = nb * 16 + lrow;
            const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + wrow + g * 16 + lq * 4 + r;
                if (row < n_out) {
                    float v = acc[g][nb][r] + bv;
                    if (residual) v += residual[(int64_t)row * COUT + col];
                    if (relu) v = fmaxf(v, 0.0f);
                    out[(int64_t)row * COUT + col] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------- bf16
// CIN >= 32: chunks of 32 channels through v_mfma_f32_16x16x32_bf16 (8 bf16 = 16 B per lane).
// CIN == 16: one chunk of 16 channels through v_mfma_f32_16x16x16_bf16 (4 bf16 = 8 B per lane).
template <int CIN, int COUT, int RG>
__global__ void __launch_bounds__(256) spconv_bf16(const unsigned short *__restrict__ in, const void *__restrict__ wp_,
                                                   const float *__restrict__ bias, const unsigned short *__restrict__ residual,
                                                   int relu, const int *__restrict__ nbr, int64_t nbr_stride, int K, int n_out,
                                                   const int *__restrict__ n_out_dev, unsigned short *__restrict__ out) {
    constexpr int ROWS_B = 64 * RG;
    constexpr int NB = COUT / 16;
    constexpr bool WIDE = CIN >= 32;
    constexpr int NC = WIDE ? CIN / 32 : 1;
    __shared__ int s_nbr[kMaxTaps * ROWS_B];
    const int tile = blockIdx.x;  // index order: contiguous XCD chunks concentrate the dense regions on a few XCDs (measured -10%)
    const int row0 = tile * ROWS_B;
    n_out = fd::device_count(n_out, n_out_dev);  // capacity launch (fd_common.h): workgroups past the device's count leave
    if (row0 >= n_out) return;
    for (int t = threadIdx.x; t < K * ROWS_B; t += 256) {
        int k = t / ROWS_B, r = t - k * ROWS_B;
        int64_t o = (int64_t)row0 + r;
        s_nbr[t] = (o < n_out) ? nbr[(int64_t)k * nbr_stride + o] : -1;  // rows >= n_out of the table are never read
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lrow = lane & 15, lq = lane >> 4;
    const int wrow = wave * 16 * RG;
    if (row0 + wrow >= n_out) return;

    f32x4 acc[RG][NB];
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[g][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int k = 0; k < K; ++k) {
        int idx[RG];
        bool any[RG];
        bool any_all = false;
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            idx[g] = s_nbr[k * ROWS_B + wrow + g * 16 + lrow];
            any[g] = __ballot(idx[g] >= 0) != 0ull;
            any_all = any_all || any[g];
        }
        if (!any_all) continue;
        if constexpr (WIDE) {
            const bf16x8 *wk = reinterpret_cast<const bf16x8 *>(wp_) + (int64_t)k * NC * NB * 64 + lane;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                bf16x8 a[RG];
#pragma unroll
                for (int g = 0; g < RG; ++g) {
                    uint4 raw = make_uint4(0u, 0u, 0u, 0u);
                    if (idx[g] >= 0) raw = *reinterpret_cast<const uint4 *>(in + (int64_t)idx[g] * CIN + c * 32 + lq * 8);
                    a[g] = __builtin_bit_cast(bf16x8, raw);
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const bf16x8 b = wk[(c * NB + nb) * 64];
#pragma unroll
                    for (int g = 0; g < RG; ++g)
                        if (any[g]) acc[g][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[g], b, acc[g][nb], 0, 0, 0);
                }
            }
        } else {
            const s16x4 *wk = reinterpret_cast<const s16x4 *>(wp_) + (int64_t)k * NB * 64 + lane;
            s16x4 a[RG];
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                uint2 raw = make_uint2(0u, 0u);
                if (idx[g] >= 0) raw = *reinterpret_cast<const uint2 *>(in + (int64_t)idx[g] * CIN + lq * 4);
                a[g] = __builtin_bit_cast(s16x4, raw);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const s16x4 b = wk[nb * 64];
#pragma unroll
                for (int g = 0; g < RG; ++g)
                    if (any[g]) acc[g][nb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[g], b, acc[g][nb], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < RG; ++g) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int col = nb * 16 + lrow;
            const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + wrow + g * 16 + lq * 4 + r;
                if (row < n_out) {
                    float v = acc[g][nb][r] + bv;
                    if (residual) v += bf2f(residual[(int64_t)row * COUT + col]);
                    if (relu) v = fmaxf(v, 0.0f);
                    out[(int64_t)row * COUT + col] = f2bf(v);
                }
            }
        }
    }
}

struct LaunchArgs {
    const void *in, *wp;
    const float *bias;
    const void *residual;
    int relu;
    const int *nbr;
    int64_t nbr_stride;
    int K, n_out;
    const int *n_out_dev;
    void *out;
    hipStream_t stream;
};

template <int CIN, int COUT, int RG>
void launch(const LaunchArgs &a, int dtype) {
    const int rows_b = 64 * RG;
    dim3 grid((unsigned)((a.n_out + rows_b - 1) / rows_b));
    if (dtype == 0)
        hipLaunchKernelGGL((spconv_f32<CIN, COUT, RG>), grid, dim3(256), 0, a.stream, (const float *)a.in, (const float4 *)a.wp, a.bias,
                           (const float *)a.residual, a.relu, a.nbr, a.nbr_stride, a.K, a.n_out, a.n_out_dev, (float *)a.out);
    else
        hipLaunchKernelGGL((spconv_bf16<CIN, COUT, RG>), grid, dim3(256), 0, a.stream, (const unsigned short *)a.in, a.wp, a.bias,
                           (const unsigned short *)a.residual, a.relu, a.nbr, a.nbr_stride, a.K, a.n_out, a.n_out_dev, (unsigned short *)a.out);
}

template <int CIN, int COUT>
void launch_rg(const LaunchArgs &a, int dtype, int rg) {
    if (rg >= 4) launch<CIN, COUT, 4>(a, dtype);
    else if (rg == 2) launch<CIN, COUT, 2>(a, dtype);
    else launch<CIN, COUT, 1>(a, dtype);
}

int pick_rg(int64_t n_out, int cin, int cout) {
    const int forced = fd::tuning(fd::kTuneSpconvRG);  // tuning / test override (fd_tuning_set)
    if (forced > 0) return forced;
    // Measured on MI355X (tools/spconv_bench.py, 300k-point cloud): the kernel is bound by the L2->L1 weight stream
    // (every wave reads all of W[k] per tap), so more rows per wave (RG) help until the wave count drops below
    // ~2 per SIMD, where gather latency is no longer hidden.  128-wide layers have few rows: RG 1; 64-wide: RG 2.
    const int64_t waves2 = (n_out + 31) / 32;
    int rg = (cin * cout >= 128 * 128) ? 1 : 2;
    if (rg == 2 && waves2 < 256 * 4 * 2) rg = 1;
    return rg;
}

}  // namespace

// fp32 32 -> 32: the 32x32x2 fragment layout of fd_spconv_c32.hip follows the 16x16x4 one in the same buffer
inline bool has_c32_layout(int cin, int cout, int dtype) { return dtype == 0 && cout == 32 && cin == 32; }
// bf16 with 16 input channels: the tap-pair layout of fd_spconv_bf16.hip (two taps stacked along K = 32) follows the 16x16x16 one
inline int64_t pair_layout_elems(int K, int cin, int cout, int dtype) { return (dtype == 1 && cin == 16) ? (int64_t)((K + 1) / 2) * 32 * cout : 0; }

extern "C" size_t fd_spconv_packed_weight_bytes(int K, int cin, int cout, int dtype) {
    if (K <= 0 || cin <= 0 || cout <= 0) return 0;
    return (size_t)K * cin * cout * (dtype == 0 ? 4 : 2) * (has_c32_layout(cin, cout, dtype) ? 2 : 1) + (size_t)pair_layout_elems(K, cin, cout, dtype) * 2 +
           (dtype == 1 ? fd::spconv_bf16_win_weight_bytes(K, cin, cout) : 0);
}

extern "C" int fd_spconv_pack_weight(const float *w, int K, int cin, int cout, int dtype, void *dst) {
    FD_REQUIRE(w && dst, "fd_spconv_pack_weight: null argument");
    FD_REQUIRE(K >= 1 && K <= kMaxTaps, "fd_spconv_pack_weight: K must be in [1,27]");
    FD_REQUIRE(cin % 16 == 0 && cout % 16 == 0 && cin >= 16 && cin <= 128 && cout >= 16 && cout <= 128,
               "fd_spconv_pack_weight: channels must be multiples of 16 in [16,128] (got %d -> %d)", cin, cout);
    FD_REQUIRE(dtype == 0 || dtype == 1, "fd_spconv_pack_weight: dtype must be 0 (f32) or 1 (bf16)");
    const int NB = cout / 16;
    auto W = [&](int k, int ci, int co) { return w[((int64_t)k * cin + ci) * cout + co]; };
    auto tobf = [](float v) -> uint16_t {
        union { float f; uint32_t u; } cvt;
        cvt.f = v;
        uint32_t u = cvt.u;
        u += 0x7fffu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    };
    if (dtype == 0) {
        const int NC = cin / 16;
        float *d = (float *)dst;
        for (int k = 0; k < K; ++k)
            for (int c = 0; c < NC; ++c)
                for (int nb = 0; nb < NB; ++nb)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 4; ++j)
                            d[((((int64_t)k * NC + c) * NB + nb) * 64 + lane) * 4 + j] =
                                W(k, 16 * c + 4 * (lane >> 4) + j, 16 * nb + (lane & 15));
        if (has_c32_layout(cin, cout, dtype)) {  // [K][cin / 8][lane][4]: channel 8 c + 4 (lane / 32) + j, output channel lane % 32
            float *e = d + (int64_t)K * cin * cout;
            for (int k = 0; k < K; ++k)
                for (int c = 0; c < cin / 8; ++c)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 4; ++j)
                            e[(((int64_t)k * (cin / 8) + c) * 64 + lane) * 4 + j] = W(k, 8 * c + 4 * (lane >> 5) + j, lane & 31);
        }
    } else if (cin >= 32) {
        FD_REQUIRE(cin % 32 == 0, "fd_spconv_pack_weight: bf16 needs cin 16 or a multiple of 32");
        const int NC = cin / 32;
        uint16_t *d = (uint16_t *)dst;
        for (int k = 0; k < K; ++k)
            for (int c = 0; c < NC; ++c)
                for (int nb = 0; nb < NB; ++nb)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j)
                            d[((((int64_t)k * NC + c) * NB + nb) * 64 + lane) * 8 + j] =
                                tobf(W(k, 32 * c + 8 * (lane >> 4) + j, 16 * nb + (lane & 15)));
        // the LDS-window kernel's 32x32x16 fragment order follows (fd_spconv_bf16win.hip)
        if (fd::spconv_bf16_win_weight_bytes(K, cin, cout)) fd::spconv_bf16_win_pack(w, K, cin, cout, +tobf, d + (int64_t)K * cin * cout);
    } else {
        uint16_t *d = (uint16_t *)dst;
        for (int k = 0; k < K; ++k)
            for (int nb = 0; nb < NB; ++nb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j)
                        d[(((int64_t)k * NB + nb) * 64 + lane) * 4 + j] = tobf(W(k, 4 * (lane >> 4) + j, 16 * nb + (lane & 15)));
        // tap pairs [ceil(K/2)][NB][lane][8]: quads 0,1 = channels 0..7 / 8..15 of tap 2u, quads 2,3 = those of tap 2u + 1 (zeros past K)
        uint16_t *e = d + (int64_t)K * cin * cout;
        for (int u = 0; u < (K + 1) / 2; ++u)
            for (int nb = 0; nb < NB; ++nb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int q = lane >> 4, k = 2 * u + (q >> 1);
                        e[(((int64_t)u * NB + nb) * 64 + lane) * 8 + j] = k < K ? tobf(W(k, 8 * (q & 1) + j, 16 * nb + (lane & 15))) : (uint16_t)0;
                    }
    }
    return FD_OK;
}

extern "C" int fd_spconv_apply(const void *in_feats, int64_t n_in, const void *wpacked, const float *bias, const void *residual, int relu,
                               const int32_t *nbr, int64_t nbr_stride, const int32_t *ranges, int n_ranges, int K, int64_t n_out,
                               const int32_t *n_out_dev, int64_t n_expected, int cin, int cout, int dtype, void *out_feats, fd_stream_t stream) {
    FD_REQUIRE(K >= 1 && K <= kMaxTaps, "fd_spconv_apply: K must be in [1,27]");
    FD_REQUIRE(dtype == 0 || dtype == 1, "fd_spconv_apply: dtype must be 0 (f32) or 1 (bf16)");
    FD_REQUIRE(n_out >= 0 && n_out <= nbr_stride && n_out < (1ll << 31), "fd_spconv_apply: n_out out of range");
    FD_REQUIRE(n_ranges >= 0 && (ranges == nullptr || n_ranges >= 1), "fd_spconv_apply: ranges needs n_ranges >= 1");
    if (n_expected <= 0 || n_expected > n_out) n_expected = n_out;  // only steers launch heuristics
    if (n_out == 0) return FD_OK;  // an empty active set (empty cloud): nothing to compute, buffers may be null
    FD_REQUIRE(in_feats && wpacked && nbr && out_feats, "fd_spconv_apply: null argument");
    if (dtype == 0 && !fd::tuning(fd::kTuneSpconvV1) && cin == 16 && fd::tuning(fd::kTuneF32ResRG) >= 0) {
        // the 16-channel level: resident weights + register accumulators + empty-item skipping (fd_spconv_f32r.hip); "f32_res_rg" = -1
        // keeps the pair-compacting kernel for A/B runs
        if (fd::spconv_f32_res16_dispatch((const float *)in_feats, wpacked, bias, (const float *)residual, relu, nbr, nbr_stride, K, n_in, (int)n_out,
                                          n_out_dev, n_expected, cin, cout, (float *)out_feats, fd::as_stream(stream)))
            return fd::check_launch("fd_spconv_apply(f32 resident)");
    }
    if (dtype == 0 && !fd::tuning(fd::kTuneSpconvV1) && has_c32_layout(cin, cout, dtype) && fd::tuning(fd::kTuneSpconvC32) >= 0) {
        // 32-column layers: 32-pair items on the 32x32x2 MFMA (fd_spconv_c32.hip); its weight layout follows the 16x16x4 one
        const void *w32 = (const char *)wpacked + (size_t)K * cin * cout * 4;
        if (fd::spconv_f32_c32_dispatch((const float *)in_feats, w32, bias, (const float *)residual, relu, nbr, nbr_stride, K, n_in, (int)n_out,
                                        n_out_dev, cin, cout, (float *)out_feats, ranges, n_ranges, fd::as_stream(stream)))
            return fd::check_launch("fd_spconv_apply(c32)");
    }
    if (dtype == 0 && !fd::tuning(fd::kTuneSpconvV1)) {
        // fp32 is MFMA-bound: the pair-compacting kernel (fd_spconv_v2.hip) feeds the matrix core no zero rows
        if (fd::spconv_f32_compact_dispatch((const float *)in_feats, wpacked, bias, (const float *)residual, relu, nbr, nbr_stride, K, n_in,
                                            (int)n_out, n_out_dev, cin, cout, (float *)out_feats, ranges, n_ranges, fd::as_stream(stream)))
            return fd::check_launch("fd_spconv_apply(compact)");
    }
    if (dtype == 1 && fd::tuning(fd::kTuneBf16GP) >= 0) {
        // bf16: register accumulators + LDS-shared weights (fd_spconv_bf16.hip); 16 input channels read the tap-pair weight layout
        const void *w = cin == 16 ? (const void *)((const char *)wpacked + (size_t)K * cin * cout * 2) : wpacked;
        // 64 -> 64, 128 -> 128 (the SubM layers of levels 2 and 3): LDS window of input rows + 32-row register tiles
        // (fd_spconv_bf16win.hip); "bf16_win" = -1 keeps the RING / RESIDENT kernels (A/B runs, variant tests)
        // The window [row - HALO, row + TM + HALO] only holds the neighbours when input rows lie around the output rows, i.e. for a SubM
        // convolution (27 taps over the SAME row set).  The strided 128 -> 128 extra_conv (K = 3, stride (2,1,1)) has the shape but not the
        // property: every lane would take the exec-masked global gather next to a window staged for nothing -> RING kernel.
        // ("bf16_win" = 2 / FD_BF16_WIN=2 sends such a layer to the window kernel all the same: the A/B of this rule)
        const bool subm_like = (K == 27 && n_in == n_out) || fd::tuning(fd::kTuneBf16Win) == 2;
        if (fd::tuning(fd::kTuneBf16Win) >= 0 && subm_like && fd::spconv_bf16_win_weight_bytes(K, cin, cout) &&
            fd::spconv_bf16_win_dispatch(in_feats, (const char *)wpacked + (size_t)K * cin * cout * 2, bias, residual, relu, nbr, nbr_stride, K, n_in, (int)n_out,
                                         n_out_dev, n_expected, cin, cout, out_feats, fd::as_stream(stream)))
            return fd::check_launch("fd_spconv_apply(bf16 window)");
        if (fd::spconv_bf16_ws_dispatch(in_feats, w, bias, residual, relu, nbr, nbr_stride, K, n_in, (int)n_out, n_out_dev, n_expected, cin, cout, out_feats,
                                        fd::as_stream(stream)))
            return fd::check_launch("fd_spconv_apply(bf16 ws)");
        // "strict" (fd_tuning_set / FD_STRICT=1; tests and tuning runs): a shape the default kernel family covers must not fall
        // through to the older kernels silently (round 3: 32 -> 64 did, because its LDS request did not fit)
        if (fd::tuning(fd::kTuneStrict) && n_in * cin * 2 < (1ll << 31)) {
            fd::set_error("fd_spconv_apply: strict mode: the bf16 kernel (fd_spconv_bf16.hip) did not take %d -> %d, K = %d", cin, cout, K);
            return FD_EINVAL;
        }
    }
    // what is left below: the register-resident output-stationary kernels of round 1 -- the path for feature matrices of 2 GB and
    // more (the kernels above address rows through 32-bit buffer offsets) and the A/B partner of the variant tests
    LaunchArgs a{in_feats, wpacked, bias, residual, relu, nbr, nbr_stride, K, (int)n_out, n_out_dev, out_feats, fd::as_stream(stream)};
    const int rg = pick_rg(n_expected, cin, cout);
    const int key = cin * 1000 + cout;
    switch (key) {
        case 16016: launch_rg<16, 16>(a, dtype, rg); break;
        case 16032: launch_rg<16, 32>(a, dtype, rg); break;
        case 32032: launch_rg<32, 32>(a, dtype, rg); break;
        case 32064: launch_rg<32, 64>(a, dtype, rg); break;
        case 64064: launch_rg<64, 64>(a, dtype, rg); break;
        case 64128: launch_rg<64, 128>(a, dtype, rg); break;
        case 128128: launch_rg<128, 128>(a, dtype, rg); break;
        default:
            fd::set_error("fd_spconv_apply: unsupported channels %d -> %d", cin, cout);
            return FD_EINVAL;
    }
    return fd::check_launch("fd_spconv_apply");
}
