// Densify for gfx950: SparseConvTensor.dense() + view(N, C*D, H, W) (det3d/models/backbones/scn.py:165-168).
// Output-stationary: one wave per (b,y,x) BEV cell reads the cell's occupancy word and writes all C*D output
// channels exactly once (zeros where inactive), so no memset pass and no scatter; element strides select NCHW
// or channels-last memory.
#include "fd_common.h"

namespace {

__device__ inline unsigned short f2bf(float v) {
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

template <bool IN_BF16, bool OUT_BF16>
__global__ void __launch_bounds__(256) densify_kernel(const void *__restrict__ feats, int C, const unsigned long long *__restrict__ words,
                                                      const int *__restrict__ prefix, fd::IndexGeom g, void *__restrict__ out,
                                                      int64_t sb, int64_t sc, int64_t sy, int64_t sx, int64_t n_rows) {
    // thread -> (cell, channel): channel index fastest so channels-last stores coalesce; for NCHW the x index
    // of neighbouring cells is 8 apart in the tiled column order, writes go through L2 either way.
    const int CD = C * g.D;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t cell = t / CD;
    int ch = (int)(t - cell * CD);  // ch = c*D + d
    if (cell >= g.num_cols()) return;
    int b, y, x;
    fd::col_to_byx(g, cell, b, y, x);
    if (y >= g.H || x >= g.W) return;
    const int c = ch / g.D, d = ch - c * g.D;
    const unsigned long long w = words[cell];
    float v = 0.0f;
    if ((w >> d) & 1ull) {
        const int row = prefix[cell] + __popcll(w & ((1ull << d) - 1ull));
        if (row < n_rows) {  // (a capacity-sized level whose count overflowed: rows past the feature matrix read as zero)
            if (IN_BF16) v = __uint_as_float(((unsigned)reinterpret_cast<const unsigned short *>(feats)[(int64_t)row * C + c]) << 16);
            else v = reinterpret_cast<const float *>(feats)[(int64_t)row * C + c];
        }
    }
    const int64_t o = b * sb + ch * sc + y * sy + x * sx;
    if (OUT_BF16) reinterpret_cast<unsigned short *>(out)[o] = f2bf(v);
    else reinterpret_cast<float *>(out)[o] = v;
}

// Channels-last output (what the plan's first convolution reads; stride_c == 1): a thread writes VEC consecutive channels of one cell as ONE
// 16-byte store (4 floats / 8 bf16) instead of one element -- the generic kernel above moves 4 (fp32) or 2 (bf16) bytes per lane and store
// instruction and ran at 1.0 / 0.5 TB/s of writes (64 us per two-map pass in both dtypes, round 5).  Same values, same zeros.
template <bool IN_BF16, bool OUT_BF16>
__global__ void __launch_bounds__(256) densify_nhwc_vec(const void *__restrict__ feats, int C, const unsigned long long *__restrict__ words,
                                                        const int *__restrict__ prefix, fd::IndexGeom g, void *__restrict__ out, int64_t sb, int64_t sy,
                                                        int64_t sx, int64_t n_rows) {
    constexpr int VEC = OUT_BF16 ? 8 : 4;
    const int CD = C * g.D, per_cell = CD / VEC;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t cell = t / per_cell;
    const int ch0 = (int)(t - cell * per_cell) * VEC;
    if (cell >= g.num_cols()) return;
    int b, y, x;
    fd::col_to_byx(g, cell, b, y, x);
    if (y >= g.H || x >= g.W) return;
    const unsigned long long w = words[cell];
    const int base = prefix[cell];
    float v[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int ch = ch0 + j, c = ch / g.D, d = ch - c * g.D;
        v[j] = 0.0f;
        if ((w >> d) & 1ull) {
            const int row = base + __popcll(w & ((1ull << d) - 1ull));
            if (row < n_rows) {
                if (IN_BF16) v[j] = __uint_as_float(((unsigned)reinterpret_cast<const unsigned short *>(feats)[(int64_t)row * C + c]) << 16);
                else v[j] = reinterpret_cast<const float *>(feats)[(int64_t)row * C + c];
            }
        }
    }
    const int64_t o = b * sb + y * sy + x * sx + ch0;
    if (OUT_BF16) {
        uint4 pk;
        pk.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
        pk.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
        pk.z = (unsigned)f2bf(v[4 % VEC]) | ((unsigned)f2bf(v[5 % VEC]) << 16);
        pk.w = (unsigned)f2bf(v[6 % VEC]) | ((unsigned)f2bf(v[7 % VEC]) << 16);
        *reinterpret_cast<uint4 *>(reinterpret_cast<unsigned short *>(out) + o) = pk;
    } else {
        *reinterpret_cast<float4 *>(reinterpret_cast<float *>(out) + o) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// NCHW float32 output (the reference layout): one workgroup per 8x8 spatial tile of the index (64 consecutive
// columns).  Phase 1 stages the tile's features in LDS with row-contiguous reads; phase 2 writes channel planes with
// lane = cell, so each group of 8 lanes stores 32 contiguous bytes of an image row (the generic kernel above stores 4).
template <bool IN_BF16>
__global__ void __launch_bounds__(256) densify_nchw_tile(const void *__restrict__ feats, int C, const unsigned long long *__restrict__ words,
                                                         const int *__restrict__ prefix, fd::IndexGeom g, float *__restrict__ out,
                                                         int64_t sb, int64_t sc, int64_t sy, int64_t sx, int64_t n_rows) {
    extern __shared__ float s_tile[];  // [64 cells][CH + 1], CH = channels of this workgroup (blockIdx.y selects the chunk)
    const int CD = C * g.D, CH = CD / gridDim.y, ch0 = blockIdx.y * CH, ld = CH + 1;
    const int64_t cell0 = (int64_t)blockIdx.x * 64;
    __shared__ unsigned long long s_w[64];
    __shared__ int s_p[64];
    if (threadIdx.x < 64) {
        s_w[threadIdx.x] = words[cell0 + threadIdx.x];
        s_p[threadIdx.x] = prefix[cell0 + threadIdx.x];
    }
    __syncthreads();
    // chunk = channels [ch0, ch0 + CH) with channel = c*D + d  ->  c in [ch0/D, (ch0+CH)/D), all d
    const int Cc = CH / g.D, c0 = ch0 / g.D;
    for (int i = threadIdx.x; i < 64 * CH; i += 256) {
        const int cell = i / CH, r = i - cell * CH;
        const int d = r / Cc, c = c0 + (r - d * Cc);  // feature rows are contiguous in c
        const unsigned long long w = s_w[cell];
        float v = 0.0f;
        if ((w >> d) & 1ull) {
            const int row = s_p[cell] + __popcll(w & ((1ull << d) - 1ull));
            if (row < n_rows) {
                if (IN_BF16) v = __uint_as_float(((unsigned)reinterpret_cast<const unsigned short *>(feats)[(int64_t)row * C + c]) << 16);
                else v = reinterpret_cast<const float *>(feats)[(int64_t)row * C + c];
            }
        }
        s_tile[cell * ld + (c - c0) * g.D + d] = v;
    }
    __syncthreads();
    int b, y0, x0;
    fd::col_to_byx(g, cell0, b, y0, x0);
    const int cell = threadIdx.x & 63, y = y0 + (cell >> 3), x = x0 + (cell & 7);
    if (y >= g.H || x >= g.W) return;
    const int64_t o0 = b * sb + y * sy + x * sx;
    for (int ch = threadIdx.x >> 6; ch < CH; ch += 4) out[o0 + (ch0 + ch) * sc] = s_tile[cell * ld + ch];
}

}  // namespace

extern "C" int fd_densify(const void *feats, int c, int dtype, const uint64_t *words, const int32_t *prefix, int B, int D, int H, int W,
                          void *out, int out_dtype, int64_t stride_b, int64_t stride_c, int64_t stride_y, int64_t stride_x, int64_t n_rows,
                          fd_stream_t stream) {
    FD_REQUIRE(words && prefix && out, "fd_densify: null argument");  // feats may be null when no cell is active
    FD_REQUIRE(c > 0 && D > 0 && D <= 64, "fd_densify: bad shape");
    fd::IndexGeom g = fd::make_geom(B, D, H, W);
    int64_t total = g.num_cols() * c * D;
    dim3 grid((unsigned)((total + 255) / 256));
    const unsigned long long *wd = (const unsigned long long *)words;
    hipStream_t s = fd::as_stream(stream);
    int chunks = 1;
    while ((c % (2 * chunks)) == 0 && (size_t)64 * (c * D / chunks + 1) * sizeof(float) > 40 * 1024) chunks *= 2;
    const size_t tile_lds = (size_t)64 * (c * D / chunks + 1) * sizeof(float);
    if (out_dtype == 0 && stride_x == 1 && tile_lds <= 60 * 1024) {  // NCHW float32: tile kernel with wider stores
        const dim3 tgrid((unsigned)(g.num_cols() / 64), (unsigned)chunks);
        if (dtype == 0)
            hipLaunchKernelGGL((densify_nchw_tile<false>), tgrid, dim3(256), tile_lds, s, feats, c, wd, prefix, g, (float *)out, stride_b, stride_c, stride_y, stride_x, n_rows);
        else
            hipLaunchKernelGGL((densify_nchw_tile<true>), tgrid, dim3(256), tile_lds, s, feats, c, wd, prefix, g, (float *)out, stride_b, stride_c, stride_y, stride_x, n_rows);
        return fd::check_launch("fd_densify(tile)");
    }
    {
        const int vec = out_dtype == 1 ? 8 : 4;
        if (stride_c == 1 && (dtype == 0 || dtype == 1) && (out_dtype == 0 || out_dtype == 1) && (c * D) % vec == 0 && stride_b % vec == 0 && stride_y % vec == 0 &&
            stride_x % vec == 0 && ((uintptr_t)out & 15) == 0) {
            const dim3 vgrid((unsigned)((total / vec + 255) / 256));
#define FD_DV(I, O) hipLaunchKernelGGL((densify_nhwc_vec<I, O>), vgrid, dim3(256), 0, s, feats, c, wd, prefix, g, out, stride_b, stride_y, stride_x, n_rows)
            if (dtype == 0 && out_dtype == 0) FD_DV(false, false);
            else if (dtype == 0) FD_DV(false, true);
            else if (out_dtype == 0) FD_DV(true, false);
            else FD_DV(true, true);
#undef FD_DV
            return fd::check_launch("fd_densify(nhwc)");
        }
    }
    if (dtype == 0 && out_dtype == 0)
        hipLaunchKernelGGL((densify_kernel<false, false>), grid, dim3(256), 0, s, feats, c, wd, prefix, g, out, stride_b, stride_c, stride_y, stride_x, n_rows);
    else if (dtype == 0 && out_dtype == 1)
        hipLaunchKernelGGL((densify_kernel<false, true>), grid, dim3(256), 0, s, feats, c, wd, prefix, g, out, stride_b, stride_c, stride_y, stride_x, n_rows);
    else if (dtype == 1 && out_dtype == 0)
        hipLaunchKernelGGL((densify_kernel<true, false>), grid, dim3(256), 0, s, feats, c, wd, prefix, g, out, stride_b, stride_c, stride_y, stride_x, n_rows);
    else if (dtype == 1 && out_dtype == 1)
        hipLaunchKernelGGL((densify_kernel<true, true>), grid, dim3(256), 0, s, feats, c, wd, prefix, g, out, stride_b, stride_c, stride_y, stride_x, n_rows);
    else {
        fd::set_error("fd_densify: bad dtype");
        return FD_EINVAL;
    }
    return fd::check_launch("fd_densify");
}
