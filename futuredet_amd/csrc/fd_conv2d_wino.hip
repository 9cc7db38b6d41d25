// 3x3 stride-1 pad-1 convolution in fp32 by Winograd F(2x2, 3x3) on the matrix cores (NHWC, fused bias + ReLU).
//
// The fp32 RPN / CenterHead layers (det3d/models/necks/rpn.py:124-159, det3d/models/bbox_heads/center_head.py:129-143,344-349)
// are bound by the fp32 MFMA rate, which is the vector-ALU rate: the direct implicit GEMM (fd_conv2d_f32.hip) tops out at the
// same ~105 TFLOP/s "direct-equivalent" as MIOpen's vector-ALU Winograd.  Winograd needs 16 multiplies per 2x2 outputs
// instead of 36, and the 16 element-wise products are 16 independent GEMMs over the input channels -- MFMA work:
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      d = 4x4 input patch, g = 3x3 filter, Y = 2x2 outputs
//   M[xi][tile][co] = sum_ci U[xi][ci][co] * V[xi][tile][ci]      xi = 0..15
//
//   * a workgroup owns TY x TX Winograd tiles (2 TY x 2 TX output pixels) times 64 * NBW output channels; wave w owns NBW
//     blocks of 16 channels for all tiles and all 16 xi: 16 * NTB * NBW accumulator quads (256 registers at NTB = NBW = 2)
//     live in the register file for the whole channel loop;
//   * per 16-channel slice: the raw (2 TY + 2) x (2 TX + 2) input patch goes to LDS, every thread turns a share of it into
//     V = B^T d B (adds / subtracts only) in a second LDS image laid out [xi][tile][16 channels], and the 16 xi-steps run
//     NTB * NBW * 4 MFMAs each (v_mfma_f32_16x16x4_f32, transposed: A operand = U fragment, B operand = V fragment);
//   * U = G g G^T is precomputed on the host into fragment order (fd_conv2d_wino_f32_pack_weight) and streamed from L2 one
//     xi-step ahead;
//   * epilogue: lane (tile j, quad q) holds all 16 xi of four consecutive output channels of tile j, so the output
//     transform A^T M A is lane-local arithmetic; bias, ReLU and four 16-byte NHWC stores follow (same placement arguments
//     as the direct kernel: channel offset into a wider tensor).
// fp32 Winograd F(2x2,3x3) rounds differently from a direct sum (error of a few ulp of the largest partial product); the
// parity tests bound it element-wise against float64.
#include "fd_common.h"

#ifndef FD_WINO_OCC16
#define FD_WINO_OCC16 3
#endif

namespace {

// phase timeline for tuning builds (tools/probes/build_trace.sh, -DFD_V2_TRACE): thread 0 of every workgroup accumulates the
// s_memtime cycles of each phase over its slices
#ifdef FD_V2_TRACE
__device__ unsigned long long *g_wtrace;
#define FD_WT(var) const unsigned long long var = __builtin_readcyclecounter()
#define FD_WDECL unsigned long long wacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define FD_WADD(i, v) wacc[i] += (v)
#define FD_WFLUSH                                                                                              \
    do {                                                                                                       \
        if (threadIdx.x == 0 && g_wtrace)                                                                      \
            for (int i_ = 0; i_ < 8; ++i_) g_wtrace[(size_t)(blockIdx.x + blockIdx.y * gridDim.x) * 8 + i_] = wacc[i_]; \
    } while (0)
#else
#define FD_WT(var)
#define FD_WDECL
#define FD_WADD(i, v)
#define FD_WFLUSH
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WinoParams {
    int B, H, W, Cin, Cout_pad, Cout_real, cout_total, co_off, relu;
    int tiles_x, tiles_y;
};

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// Variants with <= 128 accumulator registers are compiled for two (<= 64: three) workgroups per CU: one workgroup's input
// transform and slice hand-over then run under the other one's MFMAs (128->128 at 180 x 180: 91 us at one workgroup per
// CU, 75 at two, 61 at three; four need <= 128 registers, which spills: 101 us)
template <int TY, int TX, int NBW>
__global__ void __launch_bounds__(256, (TY * TX * NBW <= 16) ? FD_WINO_OCC16 : (TY * TX * NBW <= 32) ? 2 : 1) conv2d_wino_f32(const float *__restrict__ x, const float4 *__restrict__ wp, const float *__restrict__ bias,
                                                       float *__restrict__ y, WinoParams p) {
    constexpr int NTILE = TY * TX, NTB = NTILE / 16;
    static_assert(NTILE % 16 == 0, "tiles per workgroup must fill whole MFMA blocks");
    constexpr int PH = 2 * TY + 2, PW = 2 * TX + 2, PP = PH * PW;
    constexpr int NCHUNK = PP * 4, NLOAD = (NCHUNK + 255) / 256;
    constexpr int NT = 64 * NBW;
    constexpr int RAW_BYTES = PP * 64, V_BYTES = 16 * NTILE * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // raw patch | V
    unsigned char *s_raw = smem, *s_v = smem + RAW_BYTES;

    FD_WDECL;
    FD_WT(tstart);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: what derives from it stays in SGPRs
    const int lm = lane & 15, lq = lane >> 4;
    int t = blockIdx.x;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y; const int b = t / p.tiles_y;
    const int n0 = blockIdx.y * NT + wave * NBW * 16;
    const int oy0 = ty * 2 * TY, ox0 = tx * 2 * TX;
    const int nslices = p.Cin / 16;

    float4 stage[NLOAD];
    auto load_slice = [&](int s) {
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int id = tid + i * 256;
            stage[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (id < NCHUNK) {
                const int pix = id >> 2, q = id & 3;
                const int iy = oy0 - 1 + pix / PW, ix = ox0 - 1 + pix % PW;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                    stage[i] = *reinterpret_cast<const float4 *>(x + (((int64_t)b * p.H + iy) * p.W + ix) * p.Cin + s * 16 + q * 4);
            }
        }
    };
    auto store_raw = [&]() {
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int id = tid + i * 256;
            if (id < NCHUNK) *reinterpret_cast<float4 *>(s_raw + id * 16) = stage[i];
        }
    };
    // input transform: work item = (tile, channel quad, row pair of V): reads 3 rows x 4 columns of the tile's 4x4 patch,
    // writes 2 rows x 4 columns of V = B^T d B, B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]].  (Splitting the items
    // further by column pairs saves registers but costs more LDS instructions: 61 -> 66 us on 128->128 at 180 x 180.)
    auto transform = [&]() {
        for (int w = tid; w < NTILE * 4 * 2; w += 256) {
            const int half = w & 1, q = (w >> 1) & 3, tile = w >> 3;
            const int py = (tile / TX) * 2, px = (tile % TX) * 2;
            const unsigned char *src = s_raw + ((py + half) * PW + px) * 64 + q * 16;  // rows half .. half+2 of the patch
            float4 d[3][4];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) d[r][c] = *reinterpret_cast<const float4 *>(src + (r * PW + c) * 64);
            // rows of T = B^T d:  half 0 -> T0 = d0 - d2, T1 = d1 + d2;  half 1 (patch rows 1..3) -> T2 = d2 - d1, T3 = d1 - d3
            float4 ta[4], tb[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (half == 0) { ta[c] = f4sub(d[0][c], d[2][c]); tb[c] = f4add(d[1][c], d[2][c]); }
                else { ta[c] = f4sub(d[1][c], d[0][c]); tb[c] = f4sub(d[0][c], d[2][c]); }
            }
            unsigned char *dst = s_v + (size_t)tile * 64 + q * 16;
            auto put = [&](int row, const float4(&tr)[4]) {
                const float4 v0 = f4sub(tr[0], tr[2]), v1 = f4add(tr[1], tr[2]), v2 = f4sub(tr[2], tr[1]), v3 = f4sub(tr[1], tr[3]);
                *reinterpret_cast<float4 *>(dst + (size_t)(row * 4 + 0) * NTILE * 64) = v0;
                *reinterpret_cast<float4 *>(dst + (size_t)(row * 4 + 1) * NTILE * 64) = v1;
                *reinterpret_cast<float4 *>(dst + (size_t)(row * 4 + 2) * NTILE * 64) = v2;
                *reinterpret_cast<float4 *>(dst + (size_t)(row * 4 + 3) * NTILE * 64) = v3;
            };
            put(half * 2, ta);
            put(half * 2 + 1, tb);
        }
    };

    f32x4 acc[16][NTB][NBW];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int i = 0; i < NTB; ++i)
#pragma unroll
            for (int j = 0; j < NBW; ++j) acc[xi][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // packed weights: [Cout_pad/16][slice][xi][lane] x 16 bytes
    const int64_t w_nb_stride = (int64_t)nslices * 16 * 64;
    const float4 *wb[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const int nb = (n0 >> 4) + j;
        wb[j] = wp + (int64_t)(nb < (p.Cout_pad >> 4) ? nb : (p.Cout_pad >> 4) - 1) * w_nb_stride + lane;
    }
    const int total_steps = nslices * 16;
    const unsigned vbase = (unsigned)(lm * 64 + lq * 16);  // this lane's tile (inside a block of 16) and channel quad

    load_slice(0);
    // Weight ring: a xi-step has only NTB * NBW * 4 MFMAs (128 cycles at NTB = NBW = 1) while an L2 round trip is ~1000
    // cycles, so fragments are requested RW steps ahead (one step ahead the kernel waited ~800 cycles per step: 61 us for
    // 128->128 at 180 x 180; the ring makes it MFMA / transform bound).  16 % RW == 0 keeps the slot index static.
    constexpr int RW = (NTB * NBW >= 4) ? 2 : 4;
    float4 bw[RW][NBW];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int j = 0; j < NBW; ++j) bw[r][j] = wb[j][(int64_t)(r < total_steps ? r : 0) * 64];
    FD_WT(tk0);
    for (int s = 0; s < nslices; ++s) {
        FD_WT(t0);
        store_raw();
        __syncthreads();
        FD_WT(t1);
        if (s + 1 < nslices) load_slice(s + 1);  // travels under this slice's transform + MFMAs
        transform();
        FD_WT(t2);
        __syncthreads();
        FD_WT(t3);
        float4 a[2][NTB];
#pragma unroll
        for (int i = 0; i < NTB; ++i) a[0][i] = *reinterpret_cast<const float4 *>(s_v + vbase + i * 16 * 64);
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            const int step = s * 16 + xi;
            if (xi + 1 < 16) {
#pragma unroll
                for (int i = 0; i < NTB; ++i)
                    a[(xi + 1) & 1][i] = *reinterpret_cast<const float4 *>(s_v + vbase + ((xi + 1) * NTILE + i * 16) * 64);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the fragment prefetch in front of the MFMA block (see fd_conv2d_f32.hip)
#define FD_KSTEP(C)                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < NTB; ++i) _Pragma("unroll") for (int j = 0; j < NBW; ++j)                   \
        acc[xi][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[xi % RW][j].C, a[xi & 1][i].C, acc[xi][i][j], 0, 0, 0);
            FD_KSTEP(x) FD_KSTEP(y) FD_KSTEP(z) FD_KSTEP(w)
#undef FD_KSTEP
            __builtin_amdgcn_sched_barrier(0);
            {   // refill the slot just consumed with the fragment of RW steps ahead
                const int ns = step + RW < total_steps ? step + RW : 0;
#pragma unroll
                for (int j = 0; j < NBW; ++j) bw[xi % RW][j] = wb[j][(int64_t)ns * 64];
            }
        }
        FD_WT(t4);
        __syncthreads();  // V and the raw patch are rewritten by the next slice
        FD_WT(t5);
        FD_WADD(0, t1 - t0); FD_WADD(1, t2 - t1); FD_WADD(2, t3 - t2); FD_WADD(3, t4 - t3); FD_WADD(4, t5 - t4);
    }
    FD_WT(tk1);
    FD_WADD(5, ((unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) << 32) | (unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)));
    FD_WADD(6, tstart);
    // ---- output transform + epilogue: Y = A^T M A, A^T = [[1,1,1,0],[0,1,-1,-1]]
    const bool wide = ((p.cout_total | p.co_off) & 3) == 0;
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const int co = n0 + j * 16 + lq * 4;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias && co < p.Cout_real) {
            bv.x = bias[co];
            if (co + 1 < p.Cout_real) bv.y = bias[co + 1];
            if (co + 2 < p.Cout_real) bv.z = bias[co + 2];
            if (co + 3 < p.Cout_real) bv.w = bias[co + 3];
        }
#pragma unroll
        for (int i = 0; i < NTB; ++i) {
            const int tile = i * 16 + lm;
            const int oy = oy0 + (tile / TX) * 2, ox = ox0 + (tile % TX) * 2;
            f32x4 r0[4], r1[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                r0[c] = acc[c][i][j] + acc[4 + c][i][j] + acc[8 + c][i][j];
                r1[c] = acc[4 + c][i][j] - acc[8 + c][i][j] - acc[12 + c][i][j];
            }
            f32x4 yv[2][2];
            yv[0][0] = r0[0] + r0[1] + r0[2];
            yv[0][1] = r0[1] - r0[2] - r0[3];
            yv[1][0] = r1[0] + r1[1] + r1[2];
            yv[1][1] = r1[1] - r1[2] - r1[3];
            if (co >= p.Cout_real) continue;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    if (oy + dy >= p.H || ox + dx >= p.W) continue;
                    float4 v = make_float4(yv[dy][dx][0] + bv.x, yv[dy][dx][1] + bv.y, yv[dy][dx][2] + bv.z, yv[dy][dx][3] + bv.w);
                    if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    float *dst = y + (((int64_t)b * p.H + oy + dy) * p.W + ox + dx) * p.cout_total + p.co_off + co;
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
    FD_WT(tend);
    FD_WADD(7, tend);
    FD_WFLUSH;
}

template <int TY, int TX, int NBW>
bool launch_wino(const float *x, const void *wp, const float *bias, float *y, WinoParams p, hipStream_t stream) {
    constexpr int PH = 2 * TY + 2, PW = 2 * TX + 2;
    const size_t lds = (size_t)PH * PW * 64 + (size_t)16 * TY * TX * 64;
    p.tiles_x = (p.W + 2 * TX - 1) / (2 * TX);
    p.tiles_y = (p.H + 2 * TY - 1) / (2 * TY);
    auto kern = conv2d_wino_f32<TY, TX, NBW>;
    static std::atomic<uint64_t> lds_set{0};
    if (lds > 65536 && !fd::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, lds_set)) return false;
    dim3 grid((unsigned)(p.tiles_x * p.tiles_y * p.B), (unsigned)((p.Cout_real + 64 * NBW - 1) / (64 * NBW)));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, x, (const float4 *)wp, bias, y, p);
    return true;
}

constexpr int kNumWinoTiles = 7;

}  // namespace

#ifdef FD_V2_TRACE
extern "C" int fd_debug_set_wino_trace(void *p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_wtrace), &p, sizeof(p)) == hipSuccess ? 0 : -1; }
#endif

extern "C" int fd_conv2d_wino_f32_num_tiles(void) { return kNumWinoTiles; }

extern "C" size_t fd_conv2d_wino_f32_packed_weight_bytes(int cout, int cin) {
    if (cout <= 0 || cin <= 0 || cin % 16) return 0;
    return ((size_t)cout + 63) / 64 * 64 * cin * 16 * 4;
}

// w: [cout][cin][3][3] float32 -> U = G g G^T in fragment order [cout_pad/16][cin/16][xi][lane][4]:
// lane = (co & 15) + 16 * q holds U[xi][16 s + 4 q + 0..3][co];  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
extern "C" int fd_conv2d_wino_f32_pack_weight(const float *w, int cout, int cin, void *dst) {
    FD_REQUIRE(w && dst, "fd_conv2d_wino_f32_pack_weight: null argument");
    FD_REQUIRE(cin % 16 == 0 && cout > 0, "fd_conv2d_wino_f32_pack_weight: need cin %% 16 == 0");
    const int cout_pad = (cout + 63) / 64 * 64, nsl = cin / 16;
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    float *d = (float *)dst;
    for (int nb = 0; nb < cout_pad / 16; ++nb)
        for (int s = 0; s < nsl; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    const int co = nb * 16 + (lane & 15), ci = s * 16 + 4 * (lane >> 4) + j;
                    double U[4][4] = {};
                    if (co < cout) {
                        const float *g = w + ((int64_t)co * cin + ci) * 9;
                        double Gg[4][3];
                        for (int r = 0; r < 4; ++r)
                            for (int c = 0; c < 3; ++c) Gg[r][c] = G[r][0] * g[0 * 3 + c] + G[r][1] * g[1 * 3 + c] + G[r][2] * g[2 * 3 + c];
                        for (int r = 0; r < 4; ++r)
                            for (int c = 0; c < 4; ++c) U[r][c] = Gg[r][0] * G[c][0] + Gg[r][1] * G[c][1] + Gg[r][2] * G[c][2];
                    }
                    for (int xi = 0; xi < 16; ++xi)
                        d[((((int64_t)nb * nsl + s) * 16 + xi) * 64 + lane) * 4 + j] = (float)U[xi >> 2][xi & 3];
                }
    return FD_OK;
}

extern "C" int fd_conv2d_wino_nhwc_f32(const float *x, int B, int H, int W, int cin, const void *wpacked, const float *bias, int cout, int relu,
                                       float *y, int cout_total, int co_off, int tile, fd_stream_t stream) {
    FD_REQUIRE(x && wpacked && y, "fd_conv2d_wino_nhwc_f32: null argument");
    FD_REQUIRE(cin % 16 == 0 && cin >= 16, "fd_conv2d_wino_nhwc_f32: cin must be a multiple of 16 (got %d)", cin);
    FD_REQUIRE(B > 0 && H > 0 && W > 0 && cout > 0, "fd_conv2d_wino_nhwc_f32: bad shape");
    FD_REQUIRE(tile >= 0 && tile <= kNumWinoTiles, "fd_conv2d_wino_nhwc_f32: tile must be 0..%d", kNumWinoTiles);
    FD_REQUIRE(co_off >= 0 && co_off + cout <= cout_total, "fd_conv2d_wino_nhwc_f32: channel window [%d, %d) outside the %d output channels", co_off,
               co_off + cout, cout_total);
    WinoParams p;
    p.B = B; p.H = H; p.W = W; p.Cin = cin;
    p.Cout_real = cout;
    p.Cout_pad = (cout + 63) / 64 * 64;
    p.cout_total = cout_total; p.co_off = co_off; p.relu = relu;
    p.tiles_x = p.tiles_y = 0;
    hipStream_t s = fd::as_stream(stream);
    if (tile == 0) tile = 6;  // 8 x 8 pixels x 64 channels, three workgroups per CU: the fastest shape on every RPN / head layer measured
    bool ok;
    switch (tile) {
        case 1: ok = launch_wino<4, 8, 2>(x, wpacked, bias, y, p, s); break;   // 8 x 16 pixels x 128 channels, 256 accumulator registers
        case 2: ok = launch_wino<6, 8, 2>(x, wpacked, bias, y, p, s); break;   // 12 x 16 pixels x 128 channels, 384
        case 3: ok = launch_wino<4, 8, 1>(x, wpacked, bias, y, p, s); break;   // 8 x 16 pixels x 64 channels, 128
        case 4: ok = launch_wino<8, 8, 1>(x, wpacked, bias, y, p, s); break;   // 16 x 16 pixels x 64 channels, 256
        case 5: ok = launch_wino<4, 4, 2>(x, wpacked, bias, y, p, s); break;   // 8 x 8 pixels x 128 channels, 128 (two workgroups per CU)
        case 7: {  // strips of 32 tiles x 64 channels, producer + consumer waves (fd_conv2d_wino_pc.hip); needs W >= 63
            const int rc = fd::wino_pc_launch(x, wpacked, bias, y, B, H, W, cin, cout, relu, cout_total, co_off, s);
            if (rc == 1) {
                fd::set_error("fd_conv2d_wino_nhwc_f32: tile 7 needs W >= 63 and an input below 2 GB (got W = %d)", W);
                return FD_EINVAL;
            }
            ok = rc == 0;
            break;
        }
        default: ok = launch_wino<4, 4, 1>(x, wpacked, bias, y, p, s); break;  // 8 x 8 pixels x 64 channels, 64 (three per CU)
    }
    if (!ok) {
        fd::set_error("fd_conv2d_wino_nhwc_f32: the runtime refused tile %d's dynamic LDS request", tile);
        return FD_EINVAL;
    }
    return fd::check_launch("fd_conv2d_wino_nhwc_f32");
}
