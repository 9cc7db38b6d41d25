// PointPillars reader: PillarFeatureNet.forward (det3d/models/readers/pillar_encoder.py:113-164) with its
// PFNLayers (:38-55) as ONE kernel: decoration (cluster offset, pillar-centre offset, optional range), padding mask,
// Linear + eval BatchNorm1d + ReLU + max over the pillar's points, the [x, max] concatenation and the second PFNLayer.
// The reference materialises [M, P, 10] -> [M, P, 32] -> [M, P, 64] -> [M, P, 64] in HBM; here a pillar lives in one
// wave's LDS slice and only the [M, 64] result is written (HBM: P*ndim*4 B read + 64*4 B written per pillar).
//
// One wave per pillar.  Layer 1: lane = (point group g, unit u), its weight row in registers, the decorated point
// broadcast from LDS.  Layer 2 (64 units): lane = unit; the half of the input that is the repeated per-pillar max
// (pillar_encoder.py:52-54) contributes one per-pillar constant, so only U1 products per point remain.
#include "fd_common.h"

namespace {

constexpr int kMaxP = 32;    // points per pillar (shipped pp configs: 20)
constexpr int kFin = 16;     // decorated feature slots (ndim + 5 [+1]) zero padded
constexpr int kWaves = 4;

struct PillarArgs {
    const float *voxels;
    const int *num_points;
    const int *coors;  // [M,4] (b,z,y,x)
    const int *n_dev;
    long long m_max;
    int P, ndim, fin, with_distance;
    float vx, vy, x_off, y_off;
    const float *w1, *scale1, *shift1;
    const float *w2, *scale2, *shift2;
    void *out;
    int out_stride, out_bf16;
};

__device__ inline unsigned short f2bf_rn(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <int U1, bool TWO>
__global__ void __launch_bounds__(kWaves * 64) pillar_encode(PillarArgs a) {
    constexpr int G = 64 / U1;           // point groups in layer 1
    constexpr int U2 = 64;
    __shared__ float s_raw[kWaves][kMaxP * 8];
    __shared__ __attribute__((aligned(16))) float s_f[kWaves][kMaxP][kFin];
    __shared__ __attribute__((aligned(16))) float s_x1[kWaves][kMaxP + 1][U1];  // row P = max over points
    __shared__ float s_mean[kWaves][4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long n = a.n_dev ? min((long long)*a.n_dev, a.m_max) : a.m_max;
    const int u = lane % U1, g = lane / U1;

    float w1[kFin];
#pragma unroll
    for (int c = 0; c < kFin; ++c) w1[c] = c < a.fin ? a.w1[u * a.fin + c] : 0.0f;
    const float sc1 = a.scale1[u], sh1 = a.shift1[u];
    float w2[TWO ? 2 * U1 : 1];
    float sc2 = 0.f, sh2 = 0.f;
    if (TWO) {
#pragma unroll
        for (int c = 0; c < 2 * U1; ++c) w2[c] = a.w2[lane * 2 * U1 + c];
        sc2 = a.scale2[lane];
        sh2 = a.shift2[lane];
    }
    const int P = a.P, nd = a.ndim;
    const long long n_iter = (n + (long long)gridDim.x * kWaves - 1) / ((long long)gridDim.x * kWaves);
    for (long long it = 0; it < n_iter; ++it) {
        const long long m = (it * gridDim.x + blockIdx.x) * kWaves + wave;
        const bool live = m < n;
        // ---- the pillar's raw rows, its mean and the decorated + masked features
        if (live)
            for (int i = lane; i < P * nd; i += 64) s_raw[wave][i] = a.voxels[m * P * nd + i];
        __syncthreads();
        const int cnt = live ? a.num_points[m] : 0;
        if (live && lane < 3) {
            float s = 0.f;
            for (int p = 0; p < P; ++p) s += s_raw[wave][p * nd + lane];  // sum over all P rows (padding rows are zero)
            s_mean[wave][lane] = s / (float)cnt;                           // pillar_encoder.py:120-122
        }
        __syncthreads();
        if (live) {
            const float cx = (float)a.coors[m * 4 + 3] * a.vx + a.x_off;   // :128-133
            const float cy = (float)a.coors[m * 4 + 2] * a.vy + a.y_off;
            for (int i = lane; i < P * kFin; i += 64) {
                const int p = i / kFin, c = i % kFin;
                const float *r = &s_raw[wave][p * nd];
                float v = 0.f;
                if (c < nd) v = r[c];
                else if (c < nd + 3) v = r[c - nd] - s_mean[wave][c - nd];
                else if (c == nd + 3) v = r[0] - cx;
                else if (c == nd + 4) v = r[1] - cy;
                else if (c == nd + 5 && a.with_distance) v = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
                s_f[wave][p][c] = p < cnt ? v : 0.f;                       // :146-149 padding mask
            }
        }
        __syncthreads();
        // ---- PFNLayer 1: Linear(no bias) -> BN -> ReLU, max over ALL P rows (masked rows contribute relu(shift))
        float mx = -INFINITY;
        if (live) {
            for (int p = g; p < P; p += G) {
                const float4 *f4 = reinterpret_cast<const float4 *>(&s_f[wave][p][0]);
                float acc = 0.f;
#pragma unroll
                for (int q = 0; q < kFin / 4; ++q) {
                    const float4 f = f4[q];
                    acc = fmaf(f.x, w1[4 * q], acc);
                    acc = fmaf(f.y, w1[4 * q + 1], acc);
                    acc = fmaf(f.z, w1[4 * q + 2], acc);
                    acc = fmaf(f.w, w1[4 * q + 3], acc);
                }
                const float y = fmaxf(fmaf(acc, sc1, sh1), 0.f);
                mx = fmaxf(mx, y);
                if (TWO) s_x1[wave][p][u] = y;
            }
        }
        if (G >= 2) mx = fmaxf(mx, __shfl_xor(mx, U1));
        if (G >= 4) mx = fmaxf(mx, __shfl_xor(mx, 2 * U1));
        if (!TWO) {
            if (live && g == 0) {
                if (a.out_bf16) static_cast<unsigned short *>(a.out)[m * a.out_stride + u] = f2bf_rn(mx);
                else static_cast<float *>(a.out)[m * a.out_stride + u] = mx;
            }
            __syncthreads();
            continue;
        }
        if (live && g == 0) s_x1[wave][P][u] = mx;
        __syncthreads();
        // ---- PFNLayer 2 (last): input row = [x1[p], max1]; lane = output unit
        if (live) {
            float base = 0.f;
            {
                const float4 *m4 = reinterpret_cast<const float4 *>(&s_x1[wave][P][0]);
#pragma unroll
                for (int q = 0; q < U1 / 4; ++q) {
                    const float4 f = m4[q];
                    base = fmaf(f.x, w2[U1 + 4 * q], base);
                    base = fmaf(f.y, w2[U1 + 4 * q + 1], base);
                    base = fmaf(f.z, w2[U1 + 4 * q + 2], base);
                    base = fmaf(f.w, w2[U1 + 4 * q + 3], base);
                }
            }
            float m2 = -INFINITY;
            for (int p = 0; p < P; ++p) {
                const float4 *x4 = reinterpret_cast<const float4 *>(&s_x1[wave][p][0]);
                float acc = base;
#pragma unroll
                for (int q = 0; q < U1 / 4; ++q) {
                    const float4 f = x4[q];
                    acc = fmaf(f.x, w2[4 * q], acc);
                    acc = fmaf(f.y, w2[4 * q + 1], acc);
                    acc = fmaf(f.z, w2[4 * q + 2], acc);
                    acc = fmaf(f.w, w2[4 * q + 3], acc);
                }
                m2 = fmaxf(m2, fmaxf(fmaf(acc, sc2, sh2), 0.f));
            }
            if (a.out_bf16) static_cast<unsigned short *>(a.out)[m * a.out_stride + lane] = f2bf_rn(m2);
            else static_cast<float *>(a.out)[m * a.out_stride + lane] = m2;
        }
        __syncthreads();
    }
    (void)U2;
}

}  // namespace

extern "C" int fd_pillar_encode(const float *voxels, const int32_t *num_points, const int32_t *coors4, const int32_t *n_dev,
                                int64_t m_max, int max_points, int ndim, int with_distance, float vx, float vy, float x_offset,
                                float y_offset, const float *w1, const float *scale1, const float *shift1, int units1,
                                const float *w2, const float *scale2, const float *shift2, int units2, int out_dtype, void *out,
                                int out_stride, fd_stream_t stream_) {
    FD_REQUIRE(m_max >= 0 && m_max < (1ll << 30), "fd_pillar_encode: m_max out of range");
    if (m_max == 0) return FD_OK;
    FD_REQUIRE(voxels && num_points && coors4 && w1 && scale1 && shift1 && out, "fd_pillar_encode: null argument");
    FD_REQUIRE(max_points >= 1 && max_points <= kMaxP, "fd_pillar_encode: max_points must be in [1,%d]", kMaxP);
    FD_REQUIRE(ndim >= 3 && ndim <= 8, "fd_pillar_encode: ndim must be in [3,8]");
    const int fin = ndim + 5 + (with_distance ? 1 : 0);
    FD_REQUIRE(fin <= kFin, "fd_pillar_encode: ndim + 5 (+1) must be <= %d", kFin);
    FD_REQUIRE(out_dtype == 0 || out_dtype == 1, "fd_pillar_encode: out_dtype must be 0 (f32) or 1 (bf16)");
    const bool two = w2 != nullptr;
    FD_REQUIRE(!two || (scale2 && shift2 && units2 == 64 && units1 == 32),
               "fd_pillar_encode: two PFN layers are supported as 32 (+32 max) -> 64 units (num_filters=[64,64])");
    FD_REQUIRE(two || units1 == 64 || units1 == 32 || units1 == 16, "fd_pillar_encode: a single PFN layer needs 16/32/64 units");
    FD_REQUIRE(out_stride >= (two ? units2 : units1), "fd_pillar_encode: out_stride too small");
    PillarArgs a{voxels, num_points, coors4, n_dev, (long long)m_max, max_points, ndim, fin, with_distance ? 1 : 0, vx, vy, x_offset,
                 y_offset, w1, scale1, shift1, w2, scale2, shift2, out, out_stride, out_dtype};
    const unsigned grid = (unsigned)std::min<int64_t>((m_max + kWaves - 1) / kWaves, 256 * 8);
    hipStream_t st = fd::as_stream(stream_);
    if (two) hipLaunchKernelGGL((pillar_encode<32, true>), dim3(grid), dim3(kWaves * 64), 0, st, a);
    else if (units1 == 64) hipLaunchKernelGGL((pillar_encode<64, false>), dim3(grid), dim3(kWaves * 64), 0, st, a);
    else if (units1 == 32) hipLaunchKernelGGL((pillar_encode<32, false>), dim3(grid), dim3(kWaves * 64), 0, st, a);
    else hipLaunchKernelGGL((pillar_encode<16, false>), dim3(grid), dim3(kWaves * 64), 0, st, a);
    return fd::check_launch("fd_pillar_encode");
}

// -------------------------------------------------------------------------------------------------------------------
// PointPillarsScatter.forward (pillar_encoder.py:186-221): canvas[b, :, y*nx + x] = pillar row, everything else zero.
// One thread per (row, channel); consecutive threads walk the channels of one pillar, so the NHWC canvas (bf16 path)
// gets contiguous 128 B stores and the NCHW canvas one 4 B store per channel plane.
namespace {

template <bool IN_BF16, bool OUT_BF16>
__global__ void __launch_bounds__(256) pillar_scatter(const void *__restrict__ feats, int C, int feat_stride, const int *__restrict__ coors,
                                                      const int *__restrict__ n_dev, long long m_max, int B, int H, int W,
                                                      void *__restrict__ out, long long sb, long long sc, long long sy, long long sx) {
    const long long n = n_dev ? min((long long)*n_dev, m_max) : m_max;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long m = t / C;
    const int c = (int)(t - m * C);
    if (m >= n) return;
    const int b = coors[m * 4], y = coors[m * 4 + 2], x = coors[m * 4 + 3];
    if ((unsigned)b >= (unsigned)B || (unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W) return;
    float v;
    if (IN_BF16) v = __uint_as_float((unsigned)static_cast<const unsigned short *>(feats)[m * feat_stride + c] << 16);
    else v = static_cast<const float *>(feats)[m * feat_stride + c];
    const long long o = b * sb + c * sc + y * sy + x * sx;
    if (OUT_BF16) static_cast<unsigned short *>(out)[o] = f2bf_rn(v);
    else static_cast<float *>(out)[o] = v;
}

}  // namespace

extern "C" int fd_pillar_scatter(const void *feats, int c, int feat_stride, int dtype, const int32_t *coors4, const int32_t *n_dev,
                                 int64_t m_max, int B, int H, int W, void *out, int out_dtype, int64_t stride_b, int64_t stride_c,
                                 int64_t stride_y, int64_t stride_x, int zero_first, fd_stream_t stream_) {
    FD_REQUIRE(out && B > 0 && H > 0 && W > 0 && c > 0, "fd_pillar_scatter: bad shape / null output");
    FD_REQUIRE((dtype == 0 || dtype == 1) && (out_dtype == 0 || out_dtype == 1), "fd_pillar_scatter: dtype must be 0 (f32) or 1 (bf16)");
    FD_REQUIRE(m_max >= 0 && m_max * c < (1ll << 40), "fd_pillar_scatter: m_max out of range");
    hipStream_t st = fd::as_stream(stream_);
    if (zero_first) {
        // the canvas is a dense (possibly permuted) tensor: B*C*H*W elements from `out`
        const size_t bytes = (size_t)B * c * H * W * (out_dtype ? 2 : 4);
        if (bytes % 4 == 0) fd::fill_words(out, 0u, bytes / 4, st);
        else if (hipMemsetAsync(out, 0, bytes, st) != hipSuccess) return fd::check_launch("fd_pillar_scatter(memset)");
    }
    if (m_max == 0) return FD_OK;
    FD_REQUIRE(feats && coors4 && feat_stride >= c, "fd_pillar_scatter: null argument");
    const dim3 grid((unsigned)((m_max * c + 255) / 256));
#define FD_PS(I, O)                                                                                                              \
    hipLaunchKernelGGL((pillar_scatter<I, O>), grid, dim3(256), 0, st, feats, c, feat_stride, coors4, n_dev, (long long)m_max, B, H, W, out, \
                       (long long)stride_b, (long long)stride_c, (long long)stride_y, (long long)stride_x)
    if (dtype == 0 && out_dtype == 0) FD_PS(false, false);
    else if (dtype == 0) FD_PS(false, true);
    else if (out_dtype == 0) FD_PS(true, false);
    else FD_PS(true, true);
#undef FD_PS
    return fd::check_launch("fd_pillar_scatter");
}
