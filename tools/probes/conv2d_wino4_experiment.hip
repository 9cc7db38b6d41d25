// EXPERIMENT, not built into the library: Winograd F(4x4,3x3) in fp32 on MFMA (round 3).
// Result: correct (error vs float64 2e-5 .. 9e-5 of the largest output per layer, 20-30x that of F(2x2,3x3)), but with it on the
// RPN / head layers the end-to-end neck output of config 5 moves from 7.0e-4 to 1.16e-3 of the fp32 oracle -- outside the 1e-3
// budget of the path -- and config 2 from 5.2e-4 to 7.9e-4.  Speed of this first version: equal to the best F(2x2,3x3) tile (the
// same weight-stream bound, see fd_conv2d_wino_pc.hip).  Dropped; kept as a record of the kernel and its numbers.
// 3x3 stride-1 pad-1 convolution in fp32 by Winograd F(4x4, 3x3) on the matrix cores (NHWC, fused bias + ReLU).
//
// The fp32 RPN / CenterHead layers (det3d/models/necks/rpn.py:124-159, det3d/models/bbox_heads/center_head.py:129-143,344-349)
// are bound by fp32 MFMA issue (= the vector rate).  F(2x2,3x3) (fd_conv2d_wino.hip) needs 16 multiplies per 4 outputs = 4 per
// output; F(4x4,3x3) needs 36 per 16 = 2.25: 1.78x less matrix work for the same layer, with transforms of about the same cost
// per output.  The price is rounding: the transform matrices have entries up to 8 (inputs) / 1/24 (filters), the result carries an
// error of a few 1e-6 of the largest intermediate instead of a few 1e-7 -- inside the 1e-3 budget of the path, measured against
// float64 in the tests (tests/test_gpu_parity.py::test_conv2d_wino4_f32_vs_torch).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A     d = 6x6 input patch, g = 3x3 filter, Y = 4x4 outputs, 36 products xi = (r, c)
//   M[xi][tile][co] = sum_ci U[xi][ci][co] * V[xi][tile][ci]
//
//   * a workgroup owns 4 x 4 Winograd tiles (16 x 16 output pixels) times 64 output channels; wave w owns 16 channels for all
//     16 tiles and all 36 xi: 36 accumulator quads (144 registers) live in the register file for the whole channel loop;
//   * per 16-channel slice the input transform runs in two passes: pass 1 reads the 18 x 18 patch straight from global memory
//     (one item = tile row ty, patch column x, channel quad: six rows -> the six values of T = B^T d for that column; requested a
//     slice ahead, under the previous slice's MFMAs) and leaves T in LDS [ty][r][x]; pass 2 turns rows of T into V = T B
//     in the MFMA operand layout [xi][tile][16 channels].  The raw patch never goes through LDS (64 KB per workgroup, two per CU);
//   * 36 xi-steps of 4 MFMAs each (v_mfma_f32_16x16x4_f32, transposed: A operand = U fragment, B operand = V fragment), U streamed
//     from L2 through a register ring;
//   * epilogue: lane (tile j, quad q) holds all 36 xi of four consecutive output channels of tile j: the output transform
//     A^T M A is lane-local; bias, ReLU, 16-byte NHWC stores with channel offset (concat).
#include "fd_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Wino4Params {
    int B, H, W, Cin, Cout_pad, Cout_real, cout_total, co_off, relu;
    int tiles_x, tiles_y;
    unsigned x_bytes;
    int dbg;
};

// one column (or row) of B^T d:  B^T = [[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]]
__device__ __forceinline__ void bt6(const f32x4 (&d)[6], f32x4 (&t)[6]) {
    t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    const f32x4 a = d[4] - 4.f * d[2], b = d[3] - 4.f * d[1];
    t[1] = a + b;
    t[2] = a - b;
    const f32x4 c = d[4] - d[2], e = 2.f * (d[3] - d[1]);
    t[3] = c + e;
    t[4] = c - e;
    t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}

constexpr int NTILE = 16;                       // Winograd tiles per workgroup: 16 CONSECUTIVE tiles of the row-major tile order
constexpr int XCOLS = 4 * NTILE + 4;            // patch columns: at most two runs of tiles (a strip may wrap to the next tile row), 2 halo columns each
constexpr int N1 = XCOLS * 4;                   // pass-1 items (column, channel quad) = 272
constexpr int NX = N1 - 256;                    // items beyond one per thread (16): by LDS-DMA
constexpr int T_ROW = (XCOLS + XCOLS / 4) * 64; // one row r of T: [x][16 channels], 64 B of padding per 4 columns (pass 2 reads 4 tiles = stride 4 columns at once)
constexpr int T_BYTES = 6 * T_ROW;
constexpr int V_BYTES = 36 * NTILE * 64;        // V [xi][tile][16 channels]
constexpr int X_BYTES = 6 * NX * 16;
__device__ __forceinline__ int t_off(int x) { return x * 64 + (x >> 2) * 64; }

__global__ void __launch_bounds__(256, 2) conv2d_wino4_f32(const float *__restrict__ x, const float4 *__restrict__ wp, const float *__restrict__ bias,
                                                           float *__restrict__ y, Wino4Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // T | V | DMA scratch
    unsigned char *s_t = smem, *s_v = smem + T_BYTES, *s_x = smem + T_BYTES + V_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lm = lane & 15, lq = lane >> 4;
    const int n0 = blockIdx.y * 64 + wave * 16;
    const int nslices = p.Cin / 16;
    // the strip: tiles [t0, t0 + 16) = run 1 (n1 tiles from (b1, ty1, tx1) to the end of that tile row at most) + run 2 (the rest,
    // from column 0 of the next tile row -- of the next image after the last row); tiles_x >= 16, so there is no third run
    const int tpi = p.tiles_x * p.tiles_y, total = p.B * tpi;
    const int t0 = blockIdx.x * NTILE;
    const int b1 = t0 / tpi, ty1 = (t0 % tpi) / p.tiles_x, tx1 = t0 % p.tiles_x;
    const int n1 = min(NTILE, p.tiles_x - tx1), c1 = 4 * n1 + 2;
    const int t2 = t0 + n1;
    const int n2 = t2 < total ? NTILE - n1 : 0;
    const int b2 = t2 / tpi, ty2 = (t2 % tpi) / p.tiles_x;

    // ---- pass 1 operands of the next slice: six patch rows of this thread's (column, channel quad) item
    // (bounds-checked buffer loads: a row or column outside the image gets an offset past the range -> zeros, no branches around
    // the loads).  256 of the 272 items live in registers (one per thread); the last 16 (threads 0..15 of wave 0) travel by
    // LDS-DMA into a scratch [r][item][16 B] instead of costing every thread another 24 registers.
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, (int)p.x_bytes, 0x00020000);
    f32x4 stage[6];
    const unsigned row_bytes = (unsigned)p.W * (unsigned)p.Cin * 4u;
    auto item_setup = [&](int id, unsigned &off, unsigned &mask) {
        const int q = id & 3, xg = id >> 2;
        const bool second = xg >= c1;
        const int xr = second ? xg - c1 : xg;
        const bool live = second ? (n2 > 0 && xr < 4 * n2 + 2) : true;
        const int ix = (second ? 0 : 4 * tx1) - 1 + xr, iy0 = 4 * (second ? ty2 : ty1) - 1;
        off = (unsigned)((((int64_t)(second ? b2 : b1) * p.H + iy0) * p.W + ix) * p.Cin + q * 4) * 4u;  // may wrap: only used under mask
        mask = 0;
        if (live && ix >= 0 && ix < p.W)
#pragma unroll
            for (int r = 0; r < 6; ++r) mask |= (iy0 + r >= 0 && iy0 + r < p.H) ? 1u << r : 0u;
    };
    unsigned off0, rmask0, off1 = 0, rmask1 = 0;
    item_setup(tid, off0, rmask0);
    if (tid < NX) item_setup(tid + 256, off1, rmask1);
    auto load_slice = [&](int s) {
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const unsigned off = (rmask0 >> r & 1) ? off0 + r * row_bytes + s * 64 : 0xffffff00u;
            stage[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0));
        }
        if (tid < NX) {
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const unsigned off = (rmask1 >> r & 1) ? off1 + r * row_bytes + s * 64 : 0u;  // masked rows read a valid address, zeroed on use
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) unsigned *)(reinterpret_cast<const unsigned char *>(x) + off),
                                                 (__attribute__((address_space(3))) unsigned *)(s_x + r * NX * 16), 16, 0, 0);
            }
        }
    };
    auto pass1_item = [&](int id, const f32x4 (&d)[6]) {
        const int q = id & 3, xg = id >> 2;
        f32x4 tt[6];
        bt6(d, tt);
#pragma unroll
        for (int r = 0; r < 6; ++r) *reinterpret_cast<f32x4 *>(s_t + r * T_ROW + t_off(xg) + q * 16) = tt[r];
    };
    auto pass1 = [&]() {
        pass1_item(tid, stage);
        if (tid < NX) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA of this wave's own lanes
            f32x4 d[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                d[r] = *reinterpret_cast<const f32x4 *>(s_x + (r * NX + tid) * 16);
                if (!(rmask1 >> r & 1)) d[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            pass1_item(tid + 256, d);
        }
    };
    // pass 2: item = (row r of T, tile, channel quad), quad and tile fastest: V[r][0..5] = T[r][xs .. xs + 5] B, xs = the tile's first
    // patch column (run 2 starts after run 1's halo).  16 lanes write 256 contiguous bytes of V and read 4 tiles x 64 B of T at a
    // stride of 5 x 64 B (the padding): both conflict-free.
    auto pass2 = [&]() {
        for (int w = tid; w < NTILE * 4 * 6; w += 256) {
            const int q = w & 3, tile = (w >> 2) & 15, r = w >> 6;
            const int xs = tile < n1 ? 4 * tile : 4 * tile + 2;
            const unsigned char *src = s_t + r * T_ROW + q * 16;
            f32x4 d[6], v[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) d[c] = *reinterpret_cast<const f32x4 *>(src + t_off(xs + c));
            bt6(d, v);
            unsigned char *dst = s_v + (size_t)tile * 64 + q * 16;
#pragma unroll
            for (int c = 0; c < 6; ++c) *reinterpret_cast<f32x4 *>(dst + (size_t)(r * 6 + c) * NTILE * 64) = v[c];
        }
    };

    f32x4 acc[36];
#pragma unroll
    for (int xi = 0; xi < 36; ++xi) acc[xi] = f32x4{0.f, 0.f, 0.f, 0.f};

    // packed weights: [Cout_pad/16][slice][xi][lane] x 16 bytes
    const int nb = n0 >> 4;
    const float4 *wb = wp + (int64_t)(nb < (p.Cout_pad >> 4) ? nb : (p.Cout_pad >> 4) - 1) * ((int64_t)nslices * 36 * 64) + lane;
    const int total_steps = nslices * 36;
    const unsigned vbase = (unsigned)(lm * 64 + lq * 16);  // this lane's tile and channel quad
    constexpr int RW = 6;  // weight ring (divides 36): a xi-step has 4 MFMAs (128 cycles), an L2 round trip is 700+ cycles
    float4 bw[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) bw[r] = wb[(int64_t)(r < total_steps ? r : 0) * 64];

    load_slice(0);
    for (int s = 0; s < nslices; ++s) {
        if (!(p.dbg & 1)) pass1();
        __syncthreads();
        if (!(p.dbg & 2)) pass2();
        __syncthreads();
        if (s + 1 < nslices) load_slice(s + 1);  // travels under this slice's MFMAs (issued after pass 2: its 48 registers and pass 2's are not live together)
        if (p.dbg & 4) { __syncthreads(); continue; }
        float4 a[2];
        a[0] = *reinterpret_cast<const float4 *>(s_v + vbase);
#pragma unroll
        for (int xi = 0; xi < 36; ++xi) {
            const int step = s * 36 + xi;
            if (xi + 1 < 36) a[(xi + 1) & 1] = *reinterpret_cast<const float4 *>(s_v + vbase + (xi + 1) * NTILE * 64);
            __builtin_amdgcn_sched_barrier(0);
            acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[xi % RW].x, a[xi & 1].x, acc[xi], 0, 0, 0);
            acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[xi % RW].y, a[xi & 1].y, acc[xi], 0, 0, 0);
            acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[xi % RW].z, a[xi & 1].z, acc[xi], 0, 0, 0);
            acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[xi % RW].w, a[xi & 1].w, acc[xi], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const int ns = step + RW < total_steps ? step + RW : 0;
            bw[xi % RW] = wb[(int64_t)ns * 64];
        }
        __syncthreads();  // T and V are rewritten by the next slice
    }

    // ---- output transform + epilogue: Y = A^T M A, A^T = [[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]]
    const int co = n0 + lq * 4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias && co < p.Cout_real) {
        bv.x = bias[co];
        if (co + 1 < p.Cout_real) bv.y = bias[co + 1];
        if (co + 2 < p.Cout_real) bv.z = bias[co + 2];
        if (co + 3 < p.Cout_real) bv.w = bias[co + 3];
    }
    const int gt = t0 + lm;  // this lane's tile
    const bool tile_ok = gt < total;
    const int b = gt / tpi, oy = 4 * ((gt % tpi) / p.tiles_x), ox = 4 * (gt % p.tiles_x);
    const bool wide = ((p.cout_total | p.co_off) & 3) == 0;
    // R = A^T M (4 x 6), one output row i at a time, then Y[i][j] = sum_c R[i][c] A^T[j][c]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x4 R[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const f32x4 m0 = acc[0 * 6 + c], m1 = acc[1 * 6 + c], m2 = acc[2 * 6 + c], m3 = acc[3 * 6 + c], m4 = acc[4 * 6 + c], m5 = acc[5 * 6 + c];
            if (i == 0) R[c] = m0 + m1 + m2 + m3 + m4;
            else if (i == 1) R[c] = (m1 - m2) + 2.f * (m3 - m4);
            else if (i == 2) R[c] = (m1 + m2) + 4.f * (m3 + m4);
            else R[c] = (m1 - m2) + 8.f * (m3 - m4) + m5;
        }
        f32x4 Y[4];
        Y[0] = R[0] + R[1] + R[2] + R[3] + R[4];
        Y[1] = (R[1] - R[2]) + 2.f * (R[3] - R[4]);
        Y[2] = (R[1] + R[2]) + 4.f * (R[3] + R[4]);
        Y[3] = (R[1] - R[2]) + 8.f * (R[3] - R[4]) + R[5];
        if (!tile_ok || co >= p.Cout_real || oy + i >= p.H) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (ox + j >= p.W) continue;
            float4 v = make_float4(Y[j][0] + bv.x, Y[j][1] + bv.y, Y[j][2] + bv.z, Y[j][3] + bv.w);
            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            float *dst = y + (((int64_t)b * p.H + oy + i) * p.W + ox + j) * p.cout_total + p.co_off + co;
            if (wide && co + 3 < p.Cout_real) {
                *reinterpret_cast<float4 *>(dst) = v;
            } else {
                dst[0] = v.x;
                if (co + 1 < p.Cout_real) dst[1] = v.y;
                if (co + 2 < p.Cout_real) dst[2] = v.z;
                if (co + 3 < p.Cout_real) dst[3] = v.w;
            }
        }
    }
}

}  // namespace

extern "C" size_t fd_conv2d_wino4_f32_packed_weight_bytes(int cout, int cin) {
    if (cout <= 0 || cin <= 0 || cin % 16) return 0;
    return ((size_t)cout + 63) / 64 * 64 * cin * 36 * 4;
}

// w: [cout][cin][3][3] float32 -> U = G g G^T (6 x 6) in fragment order [cout_pad/16][cin/16][xi][lane][4]:
// lane = (co & 15) + 16 q holds U[xi][16 s + 4 q + 0..3][co];  G = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]
extern "C" int fd_conv2d_wino4_f32_pack_weight(const float *w, int cout, int cin, void *dst) {
    FD_REQUIRE(w && dst, "fd_conv2d_wino4_f32_pack_weight: null argument");
    FD_REQUIRE(cin % 16 == 0 && cout > 0, "fd_conv2d_wino4_f32_pack_weight: need cin %% 16 == 0");
    const int cout_pad = (cout + 63) / 64 * 64, nsl = cin / 16;
    static const double G[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    float *d = (float *)dst;
    for (int nb = 0; nb < cout_pad / 16; ++nb)
        for (int s = 0; s < nsl; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    const int co = nb * 16 + (lane & 15), ci = s * 16 + 4 * (lane >> 4) + j;
                    double U[6][6] = {};
                    if (co < cout) {
                        const float *g = w + ((int64_t)co * cin + ci) * 9;
                        double Gg[6][3];
                        for (int r = 0; r < 6; ++r)
                            for (int c = 0; c < 3; ++c) Gg[r][c] = G[r][0] * g[0 * 3 + c] + G[r][1] * g[1 * 3 + c] + G[r][2] * g[2 * 3 + c];
                        for (int r = 0; r < 6; ++r)
                            for (int c = 0; c < 6; ++c) U[r][c] = Gg[r][0] * G[c][0] + Gg[r][1] * G[c][1] + Gg[r][2] * G[c][2];
                    }
                    for (int xi = 0; xi < 36; ++xi)
                        d[((((int64_t)nb * nsl + s) * 36 + xi) * 64 + lane) * 4 + j] = (float)U[xi / 6][xi % 6];
                }
    return FD_OK;
}

extern "C" int fd_conv2d_wino4_nhwc_f32(const float *x, int B, int H, int W, int cin, const void *wpacked, const float *bias, int cout, int relu,
                                        float *y, int cout_total, int co_off, fd_stream_t stream) {
    FD_REQUIRE(x && wpacked && y, "fd_conv2d_wino4_nhwc_f32: null argument");
    FD_REQUIRE(cin % 16 == 0 && cin >= 16, "fd_conv2d_wino4_nhwc_f32: cin must be a multiple of 16 (got %d)", cin);
    FD_REQUIRE(B > 0 && H > 0 && W > 0 && cout > 0, "fd_conv2d_wino4_nhwc_f32: bad shape");
    FD_REQUIRE(co_off >= 0 && co_off + cout <= cout_total, "fd_conv2d_wino4_nhwc_f32: channel window [%d, %d) outside the %d output channels", co_off,
               co_off + cout, cout_total);
    Wino4Params p;
    p.B = B; p.H = H; p.W = W; p.Cin = cin;
    p.Cout_real = cout;
    p.Cout_pad = (cout + 63) / 64 * 64;
    p.cout_total = cout_total; p.co_off = co_off; p.relu = relu;
    const int64_t xb = (int64_t)B * H * W * cin * 4;
    FD_REQUIRE(xb < 0xffffff00ll, "fd_conv2d_wino4_nhwc_f32: the input must be smaller than 4 GB (32-bit buffer offsets), got %lld bytes", (long long)xb);
    p.x_bytes = (unsigned)xb;
    { const char *e = getenv("FD_WINO4_DBG"); p.dbg = e ? atoi(e) : 0; }
    p.tiles_x = (W + 3) / 4;
    p.tiles_y = (H + 3) / 4;
    FD_REQUIRE(p.tiles_x >= NTILE, "fd_conv2d_wino4_nhwc_f32: needs W >= %d (a strip of %d tiles spans at most two tile rows), got %d", 4 * NTILE - 3, NTILE, W);
    const size_t lds = (size_t)T_BYTES + V_BYTES + X_BYTES;
    static std::atomic<uint64_t> lds_set{0};
    if (lds > 65536 && !fd::ensure_dynamic_lds(reinterpret_cast<const void *>(conv2d_wino4_f32), lds, lds_set)) {
        fd::set_error("fd_conv2d_wino4_nhwc_f32: the runtime refused the dynamic LDS request");
        return FD_EINVAL;
    }
    dim3 grid((unsigned)(((int64_t)p.tiles_x * p.tiles_y * B + NTILE - 1) / NTILE), (unsigned)((cout + 63) / 64));
    hipLaunchKernelGGL(conv2d_wino4_f32, grid, dim3(256), lds, fd::as_stream(stream), x, (const float4 *)wpacked, bias, y, p);
    return fd::check_launch("fd_conv2d_wino4_nhwc_f32");
}
