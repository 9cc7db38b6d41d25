// fp32 sparse convolution for the 16-channel level (16 -> 16 SubM x 5, 16 -> 32 strided): weights resident in LDS, register
// accumulators, free-running waves, empty (16 rows, tap) items skipped.
//
// Replaces spconv 1.0's indice_conv / indice_subm_conv for conv_input / conv1 / the first strided convolution of
// det3d/models/backbones/scn.py:99-112 in the fp32 configuration, fused with the folded BatchNorm1d, residual add and ReLU.
//
// The pair-compacting kernel (fd_spconv_v2.hip) is built for MFMA-bound layers; on this level it is not the matrix pipe that
// costs: 30 us per 16 -> 16 launch for 2.5 us of MFMA (11 % busy, profiles/round3_pmc_summary.txt) -- list staging, in-place
// compaction, LDS accumulators and the chunk epilogue per 128 rows.  The level is also the sparsest: 4.4 pairs per row, 65 % of the
// (16-row group, tap) items have no pair at all.  This is the formulation the bf16 path uses there (fd_spconv_bf16.hip,
// RESIDENT): all 27 taps' weights (27 - 54 KB) are staged in LDS once per persistent 16-wave workgroup, a wave owns 16 * RG rows
// and all output columns with the accumulators in registers, walks only the taps that have a pair among its rows (one wave-wide
// OR over the rulebook slice), gathers 64-byte rows with bounds-checked buffer loads DEPTH items ahead, and there is no barrier
// after the staging.  v_mfma_f32_16x16x4_f32, exact fp32 FMA chains; summation order: taps ascending, channels ascending --
// fixed, independent of RG / grid (a different order than the compacting kernel's: the two agree to fp32 rounding, tested at 1e-4).
#include "fd_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxTaps = 27;

template <int COUT, int RG, int DEPTH, int NW>
__global__ void __launch_bounds__(NW * 64) spconv_f32_res16(const float *__restrict__ in, const f32x4 *__restrict__ wp, const float *__restrict__ bias,
                                                         const float *__restrict__ residual, int relu, const int *__restrict__ nbr, int64_t nbr_stride,
                                                         int K, int n_out, const int *__restrict__ n_out_dev, float *__restrict__ out, unsigned in_bytes, int interleave) {
    constexpr int NB = COUT / 16, ROWS = 16 * RG;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4 *s_w = reinterpret_cast<f32x4 *>(smem);                              // [K][NB][64]: fd_spconv_pack_weight's fp32 fragment order
    constexpr int kSliceInts = (kMaxTaps + 1) * ROWS;
    constexpr int kWaveInts = 2 * kSliceInts + ROWS;                           // two slice buffers + the 'no neighbour' row
    int *s_nbr = reinterpret_cast<int *>(s_w + kMaxTaps * NB * 64);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: what derives from it stays in SGPRs
    const int lrow = lane & 15, lq = lane >> 4;
    n_out = fd::device_count(n_out, n_out_dev);
    int *s = s_nbr + wave * kWaveInts;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (int)in_bytes, 0x00020000);
    const unsigned lane_off = (unsigned)(lq * 16);  // channels 4 lq .. + 3 of a 64-byte row

    // a contiguous chunk of tiles per workgroup (XCD-contiguous eighths, see fd_spconv_bf16.hip), walked NW tiles at a time
    const unsigned lb = fd::xcd_swizzle(blockIdx.x, gridDim.x);
    const int n_tiles = (n_out + ROWS - 1) / ROWS;
    const int tpb = (n_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t_lo = (int)lb * tpb, t_hi = t_lo + tpb < n_tiles ? t_lo + tpb : n_tiles;
    int tile_first = t_lo + wave;
    int n_iter = tile_first < t_hi ? (t_hi - tile_first + NW - 1) / NW : 0;
    int tile_step = NW;
    if (interleave) {  // all workgroups sweep the row range together: tile = it * (grid * NW) + block * NW + wave (44.0 -> 40.9 us on two clouds
                       // against XCD-contiguous chunks per workgroup, round 6; "f32_res_nw" = 1 brings the chunks back for A/B runs)
        tile_first = (int)blockIdx.x * NW + wave;
        tile_step = (int)gridDim.x * NW;
        n_iter = tile_first < n_tiles ? (n_tiles - tile_first + tile_step - 1) / tile_step : 0;
    }

    constexpr int NPRE = (kMaxTaps * ROWS + 63) / 64;
    static_assert(NPRE * 64 <= kSliceInts, "a slice buffer takes whole DMA instructions");
    auto request_slice = [&](int it) {
        const int row0 = (tile_first + it * tile_step) * ROWS;
        int *dst = s + (it & 1) * kSliceInts;
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            if (i * 64 < K * ROWS) {  // (uniform)
                const int t = lane + i * 64;
                int k = t / ROWS;
                const int r = t - k * ROWS;
                k = k < K ? k : K - 1;
                int o = row0 + r;
                o = o < n_out ? o : n_out - 1;  // masked on use
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) int *)(nbr + (int64_t)k * nbr_stride + o),
                                                 (__attribute__((address_space(3))) int *)(dst + i * 64), 4, 0, 0);
            }
        }
    };
    for (int r = lane; r < ROWS; r += 64) s[2 * kSliceInts + r] = -1;  // the 'no neighbour' row
    // The residual rows of a tile are requested ONE TILE AHEAD, together with its rulebook slice, and the accumulators start from
    // bias + residual: the epilogue has no global load left.  (Round 3-5 loaded them in the epilogue: one exposed memory round trip per
    // 16-row tile of a wave whose whole tile takes ~5 us.)
    f32x4 r_next[RG][NB];
    auto request_residual = [&](int it) {
        const int row0 = (tile_first + it * tile_step) * ROWS;
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            const int row = row0 + 16 * g + lrow;
            const int64_t rb = (int64_t)(row < n_out ? row : 0) * COUT;  // (clamped address; rows past the end are never stored)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) r_next[g][nb] = *reinterpret_cast<const f32x4 *>(residual + rb + 16 * nb + 4 * lq);
        }
    };
    if (n_iter > 0) {
        request_slice(0);
        if (residual) request_residual(0);
    }
    // the weights travel together with the first slice (one round trip at the start of a workgroup's life instead of two)
    for (int i = tid; i < K * NB * 64; i += NW * 64) s_w[i] = wp[i];
    __syncthreads();

    for (int it = 0; it < n_iter; ++it) {
        const int row0 = (tile_first + it * tile_step) * ROWS;
        // Wait for this tile's slice (LDS-DMA), not for everything: vector-memory operations of a wave complete in issue order on gfx9, and
        // the only ones issued after the slice (+ residual) request that can still be pending are the previous tile's RG * NB output stores
        // (its gathers were consumed).  vmcnt(0) here made every tile wait for the previous tile's store round trip.
        if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RG * NB) : "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        f32x4 r_cur[RG][NB];
#pragma unroll
        for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) r_cur[g][nb] = residual ? r_next[g][nb] : (f32x4){0.f, 0.f, 0.f, 0.f};
        if (it + 1 < n_iter) {
            request_slice(it + 1);
            if (residual) request_residual(it + 1);
        }
        const int *sl = s + (it & 1) * kSliceInts;
        bool valid[RG];
#pragma unroll
        for (int g = 0; g < RG; ++g) valid[g] = row0 + 16 * g + lrow < n_out;

        f32x4 acc[RG][NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (bias) bv = *reinterpret_cast<const f32x4 *>(bias + 16 * nb + 4 * lq);
#pragma unroll
            for (int g = 0; g < RG; ++g) acc[g][nb] = bv + r_cur[g][nb];
        }
        // taps that have a pair among this wave's rows (OR over the lanes' entries of the slice)
        unsigned mine = 0u;
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int t = lane + i * 64;
            if (i * 64 < K * ROWS && t < K * ROWS) {
                const int tap = t / ROWS, r = t - tap * ROWS;
                if (sl[t] >= 0 && row0 + r < n_out) mine |= 1u << tap;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mine |= (unsigned)__shfl_xor((int)mine, o);
        unsigned rem = (unsigned)__builtin_amdgcn_readfirstlane((int)mine);
        const int n_steps = __builtin_popcount(rem);
        auto next_step = [&]() -> int {
            const int t = rem ? __builtin_ctz(rem) : K;  // K = 'no step': its entries are the 'no neighbour' row
            rem &= rem - 1u;
            return t;
        };
        auto fetch_idx = [&](int t, int(&e)[RG]) {
            const int *p = (t < K ? sl + t * ROWS : s + 2 * kSliceInts) + lrow;
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                const int v = p[16 * g];
                e[g] = valid[g] ? v : -1;
            }
        };
        auto issue = [&](u32x4(&dst)[RG], const int(&e)[RG]) {
#pragma unroll
            for (int g = 0; g < RG; ++g) dst[g] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ((unsigned)e[g] << 6) + lane_off, 0, 0);  // -1: out of range, zeros
        };
        u32x4 a_r[DEPTH][RG];
        int t_r[DEPTH], e_next[RG];
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d) {
            t_r[d] = next_step();
            fetch_idx(t_r[d], e_next);
            issue(a_r[d], e_next);
        }
        int t_n = next_step();
        fetch_idx(t_n, e_next);
        for (int i0 = 0; i0 < n_steps; i0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int t = t_r[d];
                issue(a_r[(d + DEPTH - 1) % DEPTH], e_next);
                t_r[(d + DEPTH - 1) % DEPTH] = t_n;
                t_n = next_step();
                fetch_idx(t_n, e_next);
                const int tw = t < K ? t : K - 1;
                const f32x4 *wsrc = s_w + (tw * NB) * 64 + lane;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const f32x4 wf = wsrc[nb * 64];
#pragma unroll
                    for (int g = 0; g < RG; ++g) {
                        const f32x4 b = __builtin_bit_cast(f32x4, a_r[d][g]);
                        acc[g][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[0], b[0], acc[g][nb], 0, 0, 0);
                        acc[g][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[1], b[1], acc[g][nb], 0, 0, 0);
                        acc[g][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[2], b[2], acc[g][nb], 0, 0, 0);
                        acc[g][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[3], b[3], acc[g][nb], 0, 0, 0);
                    }
                }
            }
        }
        // ---- epilogue: lane (row lrow of the group, quad lq) holds channels 16 nb + 4 lq .. + 3 of its row
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            const int row = row0 + 16 * g + lrow;
            const int64_t rb = (int64_t)(row < n_out ? row : 0) * COUT;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                f32x4 v = acc[g][nb];
                if (relu) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
                }
                if (row < n_out) *reinterpret_cast<f32x4 *>(out + rb + 16 * nb + 4 * lq) = v;
            }
        }
    }
}

template <int COUT, int RG, int DEPTH, int NW>
int launch_res16(const float *in, const void *wp, const float *bias, const float *residual, int relu, const int *nbr, int64_t nbr_stride, int K, int n_out,
                 const int *n_out_dev, int64_t n_expected, float *out, unsigned in_bytes, hipStream_t stream) {
    constexpr size_t lds = (size_t)kMaxTaps * (COUT / 16) * 1024 + (size_t)NW * (2 * (kMaxTaps + 1) + 1) * 16 * RG * 4;
    static_assert(lds <= 160 * 1024, "LDS request");
    constexpr int per_cu = (int)((160 * 1024) / lds) < 32 / NW ? (int)((160 * 1024) / lds) : 32 / NW;  // resident workgroups per CU (LDS, 32 waves)
    auto kern = spconv_f32_res16<COUT, RG, DEPTH, NW>;
    static std::atomic<uint64_t> lds_set{0};
    if (lds > 65536 && !fd::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, lds_set)) return 0;
    constexpr int64_t wg_rows = NW * 16 * RG;
    int64_t grid = (n_expected + wg_rows - 1) / wg_rows;
    const int64_t cap = (int64_t)fd::device_cu_count() * (per_cu > 0 ? per_cu : 1);  // persistent workgroups, every resident slot of every CU
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, stream, in, (const f32x4 *)wp, bias, residual, relu, nbr, nbr_stride, K, n_out, n_out_dev,
                       out, in_bytes, fd::tuning(fd::kTuneF32ResNW) == 1 ? 0 : 1);
    return 1;
}

}  // namespace

namespace fd {
// 16 input channels, 16 or 32 output channels.  1 = launched, 0 = not this kernel's shape.
int spconv_f32_res16_dispatch(const float *in, const void *wp, const float *bias, const float *residual, int relu, const int *nbr, int64_t nbr_stride, int K,
                              int64_t n_in_bound, int n_out, const int *n_out_dev, int64_t n_expected, int cin, int cout, float *out, hipStream_t stream) {
    // 16 -> 32 (the strided convolution into level 1: 272k output rows, 2.4 pairs per row) is the pair-compacting kernel's: measured
    // 63 us here against 44 us there (profiles/round4_serial_step_summary.txt of the first run); "f32_res_rg" >= 32 forces it for A/B runs
    if (cin != 16 || (cout != 16 && cout != 32) || n_in_bound * 64 >= (1ll << 31)) return 0;
    if (cout == 32 && fd::tuning(fd::kTuneF32ResRG) < 32) return 0;
    const unsigned in_bytes = (unsigned)(n_in_bound * 64);
    const int rg = fd::tuning(fd::kTuneF32ResRG);
    // Measured and dropped in round 6 (profiles/round6_tiles_prototype.txt): gather ring depth 8 / 12 / 16 (24.6 -> 26.9 / 28.2 / 32.1 us), two workgroups of 14
    // waves per CU (29.4 vs 27.3 us), two row groups per wave with interleaved tiles (43.6 vs 41.1 us on two clouds).
#define FD_RES(CO, RGV) launch_res16<CO, RGV, 4, 16>(in, wp, bias, residual, relu, nbr, nbr_stride, K, n_out, n_out_dev, n_expected, out, in_bytes, stream)
    if (cout == 16) return rg >= 2 ? FD_RES(16, 2) : FD_RES(16, 1);
    return FD_RES(32, 1);
#undef FD_RES
}
}  // namespace fd
