// Dense 2-D convolution for gfx950 in fp32 (fp32 in, fp32 MFMA, fp32 out), NHWC, fused bias (+folded BN) + ReLU.
//
// Serves the RPN neck and CenterHead convolutions (det3d/models/necks/rpn.py:81-159,
// det3d/models/bbox_heads/center_head.py:104-143,344-349) of the fp32 configurations (BASELINE configs[1], [4]).  MIOpen's
// fp32 path for these shapes is an assembly Winograd kernel on the vector ALU (SQ_VALU_MFMA_BUSY_CYCLES = 0, see
// profiles/) plus separate bias / ReLU / concat passes; this is a direct implicit GEMM on v_mfma_f32_16x16x4_f32:
//
//   * a workgroup computes a TH x TW patch of output pixels (flattened to MB = ceil(TH*TW/16) blocks of 16) times
//     NT = 16 * NBW * WAVES_N output channels; a wave owns NBW blocks of 16 channels for MB / WAVES_M pixel blocks
//     (wide layers: 4 waves side by side over the channels, each with all pixel blocks; the narrow final head
//     convolutions: 4 waves over the pixel blocks), so the accumulators never leave the register file;
//   * for every 16-channel slice of the input the (TH-1)*S+KS by (TW-1)*S+KS halo patch is staged in LDS once (64 bytes
//     per pixel) and all KS*KS taps read their fragments from it with ds_read_b128 at a per-lane base plus an immediate
//     tap offset -- the 9x im2col blow-up never touches HBM or L2.  (No swizzle: a b128 lane group covers 16 consecutive
//     pixels x 2 chunks = a 2-way bank conflict, and with one 1 KiB fragment read per 4 * NBW MFMAs of 32 cycles the
//     LDS is nowhere near busy; a swizzle would cost ~9 vector-ALU instructions per read next to the MFMAs.)
//   * the product is issued transposed (A operand = weight fragment, B operand = pixel fragment): lane (pixel j, quad q)
//     ends up with four consecutive output channels of pixel j, so the epilogue adds bias, applies ReLU and stores
//     16 bytes per lane straight into the NHWC tensor -- no transpose through LDS;
//   * weights are pre-packed in fragment order and streamed from L2 with one coalesced 1 KiB load per wave instruction,
//     one (slice, tap) step ahead of their MFMAs; the next slice's patch is fetched into registers while the current
//     slice's MFMAs run (double-buffered LDS);
//   * output placement (y*osy+ooy, x*osx+oox, co_off + co) in a tensor with cout_total channels expresses the RPN's
//     concat and the 2x2 stride-2 transposed convolution (four 1x1 convolutions writing interleaved pixels) in place.
// The tile shape is chosen per layer: `tile` = 0 lets a makespan estimate pick among the instantiated shapes (180 x 180
// maps with 128 channels: 12 x 12 tiles = 225 workgroups for 256 compute units, no padded pixel rows); the host plan
// (futuredet_amd/dense_bf16.py) instead times every shape once per layer and passes the winner.
#include "fd_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvParamsF {
    int B, H, W, Cin, Ho, Wo, Cout_pad, Cout_real, cout_total, co_off, pad, relu;
    int osy, osx, ooy, oox;  // output pixel mapping
    int tiles_x, tiles_y;
    int cin_stride;  // channels of an input pixel in memory (= Cin unless grouped)
    int cout_sub;    // > 0: pixel shuffle -- output channel c is sub-convolution c / cout_sub (its pixel offset (dy, dx) =
                     // (sub / osx, sub % osx)) and channel c % cout_sub: ConvTranspose2d(k, stride k) as ONE 1x1 convolution
    int groups;      // > 1: grouped convolution, blockIdx.y = group: input channels [g*Cin, (g+1)*Cin), 16 (padded) outputs
    int gcnt[8], goff[8];  // real output channels of each group and where they go in the output tensor
};

// CS = 16-channel sub-slices staged per barrier (3x3: 2 -> 32 channels, 1x1: 4 -> 64 channels: a 1x1 step has no taps to
// amortise the patch hand-over over)
template <int KS, int S, int TH, int TW, int NBW, int WAVES_N, int CS>
__global__ void __launch_bounds__(256) conv2d_nhwc_f32(const float *__restrict__ x, const float4 *__restrict__ wp, const float *__restrict__ bias,
                                                       float *__restrict__ y, ConvParamsF p) {
    constexpr int MPIX = TH * TW, MB_ALL = (MPIX + 15) / 16;
    constexpr int WAVES_M = 4 / WAVES_N, MB = MB_ALL / WAVES_M;  // pixel blocks per wave
    static_assert(WAVES_N * WAVES_M == 4 && MB * WAVES_M == MB_ALL, "pixel blocks must split evenly over the waves");
    constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS, PP = PH * PW;
    constexpr int PIXB = 64 * CS;                       // bytes per pixel of a staged slice
    constexpr int NCHUNK = PP * 4 * CS;                 // 16-byte chunks per staged slice
    constexpr int NLOAD = (NCHUNK + 255) / 256;
    constexpr int NT = 16 * NBW * WAVES_N;              // output channels per workgroup
    constexpr int TAPS = KS * KS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 x PP x PIXB bytes

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: what derives from it stays in SGPRs
    const int lm = lane & 15, lq = lane >> 4;
    int t = blockIdx.x;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y; const int b = t / p.tiles_y;
    const int wn = wave % WAVES_N, wm = wave / WAVES_N;
    const int n0 = blockIdx.y * NT + wn * NBW * 16;     // first output channel of this wave
    const int cin0 = p.groups > 1 ? (int)blockIdx.y * p.Cin : 0;  // grouped: this group's slice of the input channels
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - p.pad, ix0 = ox0 * S - p.pad;
    const int nslices = (p.Cin / 16 + CS - 1) / CS;  // staged slices; the last one may be partly past Cin (zero-filled)

    float4 stage[NLOAD];
    auto load_slice = [&](int s) {
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int id = tid + i * 256;
            stage[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (id < NCHUNK) {
                const int pix = id / (4 * CS), q = id % (4 * CS);
                const int iy = iy0 + pix / PW, ix = ix0 + pix % PW;
                const int ch = s * 16 * CS + q * 4;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && ch < p.Cin)
                    stage[i] = *reinterpret_cast<const float4 *>(x + (((int64_t)b * p.H + iy) * p.W + ix) * p.cin_stride + cin0 + ch);
            }
        }
    };
    auto store_slice = [&](int buf) {
        unsigned char *dst = smem + buf * (PP * PIXB);
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int id = tid + i * 256;
            if (id < NCHUNK) *reinterpret_cast<float4 *>(dst + id * 16) = stage[i];  // pixel-major, chunk-minor = id order
        }
    };

    f32x4 acc[MB][NBW];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NBW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // LDS byte offset of this lane's pixel (tap 0) and chunk in every pixel block of the wave (row-major inside the tile;
    // lanes past the tile re-read its last pixel)
    int abase[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        int m = (wm * MB + i) * 16 + lm;
        m = m < MPIX ? m : MPIX - 1;
        abase[i] = ((m / TW) * S * PW + (m % TW) * S) * PIXB + lq * 16;
    }
    // packed weights: [Cout_pad/16][slice][tap][lane] x 16 bytes
    const int64_t w_nb_stride = (int64_t)(p.Cin / 16) * TAPS * 64;
    const float4 *wb[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const int nb = (n0 >> 4) + j;  // blocks past the packed width (a tile wider than the layer) re-read the last one
        wb[j] = wp + (int64_t)(nb < (p.Cout_pad >> 4) ? nb : (p.Cout_pad >> 4) - 1) * w_nb_stride + lane;
    }
    const int total_steps = (p.Cin / 16) * TAPS;

    load_slice(0);
    store_slice(0);
    __syncthreads();
    float4 bw[NBW], bw_next[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j) bw[j] = wb[j][0];
    if (nslices > 1) load_slice(1);  // registers hold slice s+1 while slice s is computed
    for (int s = 0; s < nslices; ++s) {
        const unsigned char *src = smem + (s & 1) * (PP * PIXB);
        // pixel fragments are read one step ahead of their MFMAs (the LDS round trip of MB reads would otherwise sit in
        // front of every step's MFMA block: one wave per SIMD has nobody else to fill it)
        float4 a[2][MB];
#pragma unroll
        for (int i = 0; i < MB; ++i) a[0][i] = *reinterpret_cast<const float4 *>(src + abase[i]);
        constexpr int STEPS = CS * TAPS;  // steps of a staged slice: sub-slice major, tap minor (= the packed weight order)
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const int step = s * STEPS + st;
            if (step >= total_steps) break;  // (only a trailing partial slice ends early)
            {   // weights of the next step travel under this step's MFMAs
                const int ns = step + 1 < total_steps ? step + 1 : 0;
#pragma unroll
                for (int j = 0; j < NBW; ++j) bw_next[j] = wb[j][(int64_t)ns * 64];
            }
            if (st + 1 < STEPS) {
                constexpr int kZero = 0;
                const int sub = (st + 1) / TAPS + kZero, tap = (st + 1) % TAPS;
                const int toff = ((tap / KS) * PW + (tap % KS)) * PIXB + sub * 64;  // compile-time: the ds_read offset field
#pragma unroll
                for (int i = 0; i < MB; ++i) a[(st + 1) & 1][i] = *reinterpret_cast<const float4 *>(src + abase[i] + toff);
            }
            // pin both prefetches HERE: left alone, hipcc sinks the weight loads below the last MFMA that reads the old
            // registers (to coalesce the copy at the end of the step) and the fragment reads next to them, which puts an
            // L2 round trip and an LDS round trip in front of every step
            __builtin_amdgcn_sched_barrier(0);
            // k-step outermost: MB * NBW independent accumulators lie between two MFMAs on the same one (the dependent-
            // accumulator latency of v_mfma_f32_16x16x4_f32 is 40 cycles against 32 of issue)
#define FD_KSTEP(C)                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < MB; ++i) _Pragma("unroll") for (int j = 0; j < NBW; ++j)                    \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[j].C, a[st & 1][i].C, acc[i][j], 0, 0, 0);
            FD_KSTEP(x) FD_KSTEP(y) FD_KSTEP(z) FD_KSTEP(w)
#undef FD_KSTEP
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NBW; ++j) bw[j] = bw_next[j];
        }
        // patch hand-over at the END of a slice (vector-memory loads return in order: issued here the next patch has a full
        // slice of MFMAs to land and only the weight loads of the next slice's first steps queue behind it)
        if (s + 1 < nslices) store_slice((s + 1) & 1);
        if (s + 2 < nslices) load_slice(s + 2);
        __syncthreads();
    }
    // epilogue: lane (pixel lm of block i, quad lq) holds channels n0 + 16 j + 4 lq .. + 3 of its pixel
    const int64_t Hy = (int64_t)p.Ho * p.osy, Wy = (int64_t)p.Wo * p.osx;
    const bool wide = ((p.cout_total | p.co_off) & 3) == 0 && p.groups <= 1;  // 16-byte aligned channel quads
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const int co = n0 + j * 16 + lq * 4;          // channel in the (virtual) output of this launch
        int co_bias = co, co_out = co, limit = p.Cout_real, ooy = p.ooy, oox = p.oox;
        if (p.cout_sub > 0) {                          // pixel shuffle: (sub-convolution, channel)
            const int sub = co / p.cout_sub;
            co_out = co_bias = co - sub * p.cout_sub;
            limit = p.cout_sub;
            ooy = sub / p.osx;
            oox = sub - ooy * p.osx;
            if (co >= p.Cout_real) continue;
        } else if (p.groups > 1) {                     // grouped: local channel -> the group's slot in the output
            const int g = blockIdx.y, cl = co - g * NT;
            co_out = p.goff[g] + cl;
            limit = p.goff[g] + p.gcnt[g];
        }
        if (co_out >= limit) continue;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) {
            bv.x = bias[co_bias];
            if (co_out + 1 < limit) bv.y = bias[co_bias + 1];
            if (co_out + 2 < limit) bv.z = bias[co_bias + 2];
            if (co_out + 3 < limit) bv.w = bias[co_bias + 3];
        }
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int m = (wm * MB + i) * 16 + lm;
            const int oy = oy0 + m / TW, ox = ox0 + m % TW;
            if (m >= MPIX || oy >= p.Ho || ox >= p.Wo) continue;
            float4 v = make_float4(acc[i][j][0] + bv.x, acc[i][j][1] + bv.y, acc[i][j][2] + bv.z, acc[i][j][3] + bv.w);
            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            const int64_t yy = (int64_t)oy * p.osy + ooy, xx = (int64_t)ox * p.osx + oox;
            float *dst = y + (((int64_t)b * Hy + yy) * Wy + xx) * p.cout_total + p.co_off + co_out;
            if (wide && co_out + 3 < limit) {
                *reinterpret_cast<float4 *>(dst) = v;
            } else {
                dst[0] = v.x;
                if (co_out + 1 < limit) dst[1] = v.y;
                if (co_out + 2 < limit) dst[2] = v.z;
                if (co_out + 3 < limit) dst[3] = v.w;
            }
        }
    }
}

template <int KS, int S, int TH, int TW, int NBW, int WAVES_N>
void launch_f32(const float *x, const void *wp, const float *bias, float *y, ConvParamsF p, hipStream_t stream) {
    constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS;
    constexpr int NT = 16 * NBW * WAVES_N;
    // channels staged per barrier: 64 for 1x1, 32 for 3x3 stride 1, 16 for the large stride-2 patches
    constexpr int CS = KS == 1 ? 4 : (S == 1 ? 2 : 1);
    const size_t lds = 2 * (size_t)PH * PW * 64 * CS;
    p.tiles_x = (p.Wo + TW - 1) / TW;
    p.tiles_y = (p.Ho + TH - 1) / TH;
    auto kern = conv2d_nhwc_f32<KS, S, TH, TW, NBW, WAVES_N, CS>;
    static std::atomic<uint64_t> lds_set{0};
    if (lds > 65536) (void)fd::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, lds_set);
    dim3 grid((unsigned)(p.tiles_x * p.tiles_y * p.B), (unsigned)(p.groups > 1 ? p.groups : (p.Cout_real + NT - 1) / NT));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, x, (const float4 *)wp, bias, y, p);
}

struct TileChoice { int th, tw, nbw, wn; };
constexpr TileChoice kTiles[] = {{12, 12, 2, 4}, {8, 16, 2, 4}, {5, 15, 2, 4}, {8, 8, 2, 4},   // 128 channels per workgroup
                                 {12, 12, 1, 4}, {8, 16, 1, 4}, {5, 15, 1, 4}, {8, 8, 1, 4},   // 64
                                 {8, 16, 2, 1}, {8, 16, 1, 1}, {8, 8, 2, 1}, {8, 8, 1, 1},     // 32 / 16: the final head convolutions
                                 {8, 16, 2, 2}, {8, 16, 4, 2}, {8, 8, 2, 2}, {8, 8, 4, 2}};    // 2 x 2 wave grids: 64 / 128 channels
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

// makespan estimate in MFMA-block units (16 pixels x 16 channels x one 16-channel slice-tap) per SIMD: workgroups are dealt
// to the compute units in rounds; a workgroup costs (pixel blocks per wave) x NBW, plus a prologue / epilogue term
inline double tile_cost(int Ho, int Wo, int B, int cout, const TileChoice &c, int n_cu) {
    const int nt = 16 * c.nbw * c.wn;
    const int64_t wgs = (int64_t)((Ho + c.th - 1) / c.th) * ((Wo + c.tw - 1) / c.tw) * B * ((cout + nt - 1) / nt);
    const int mb = ((c.th * c.tw + 15) / 16) / (4 / c.wn);
    const int64_t rounds = (wgs + n_cu - 1) / n_cu;
    return (double)rounds * (mb * c.nbw + 0.35);
}

// ---------------------------------------------------------------------------------------------- pointwise (1x1) convolution
// A 1x1 convolution is the GEMM [pixels x Cin] x [Cin x Cout] on NHWC rows that are already contiguous: no halo, so no LDS
// staging at all.  A workgroup takes 16 * NPB consecutive pixels; its four waves take NCB 16-channel blocks each.  Per
// 16-channel slice a lane loads one float4 of its pixel's row per pixel block (the B operand of the transposed product) and
// one float4 of packed weights per channel block (A operand, the fd_conv2d_f32_pack_weight layout with one tap), two
// slices in flight.  Same epilogue as the direct kernel (bias, ReLU, channel-offset / pixel-stride / pixel-shuffle placement).
// Measured (us): 128->256 at 180 x 180 47 (direct kernel with a 1x1 tap) -> 36; 256->256 at 90 x 90 35 -> 17 (MIOpen: 26 / 16).
// Row-contiguous epilogue of the pointwise kernels (round 6).  The accumulators leave a wave as lane (pixel lm, quad lq) = four channels of one
// pixel: a store instruction then touches 16 pixels with 64 bytes each.  Through a 16 x (16 NCB) LDS tile per wave the same values leave as
// lane -> 16 consecutive bytes of a pixel's channel run: four pixels x 64 NCB bytes contiguous per instruction.  Taken when the wave's
// 16 NCB channels are all real, 16-byte placed and inside one sub-convolution of a pixel shuffle; otherwise the element-wise epilogue runs.
constexpr int kEpPitch = 68;  // floats per staged pixel row (64 + 4: 16-byte aligned, rows on different bank groups)
template <int NPB, int NCB>
__device__ inline bool pointwise_epilogue_rows(const f32x4 (&acc)[NPB][NCB], const ConvParamsF &p, const float *__restrict__ bias, float *__restrict__ y,
                                               int64_t px0, int64_t n_px, int nb0, int lane, float *__restrict__ s_ep /* this wave's [16][kEpPitch] */) {
    static_assert(16 * NCB <= 64, "tile width");
    const int lm = lane & 15, lq = lane >> 4;
    const int co_base = nb0 * 16;
    if (((p.cout_total | p.co_off) & 3) != 0 || co_base + 16 * NCB > p.Cout_real) return false;
    int co_out_base = co_base, ooy = p.ooy, oox = p.oox;
    if (p.cout_sub > 0) {
        const int sub = co_base / p.cout_sub;
        if ((co_base + 16 * NCB - 1) / p.cout_sub != sub) return false;
        co_out_base = co_base - sub * p.cout_sub;
        ooy = sub / p.osx;
        oox = sub - ooy * p.osx;
    }
    float4 bv[NCB];
#pragma unroll
    for (int j = 0; j < NCB; ++j) bv[j] = bias ? *reinterpret_cast<const float4 *>(bias + co_out_base + 16 * j + 4 * lq) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t Hy = (int64_t)p.Ho * p.osy, Wy = (int64_t)p.Wo * p.osx;
    constexpr int LPP = 4 * NCB;        // lanes per pixel when reading back (16-byte pieces of its 16 NCB channels)
    constexpr int PPI = 64 / LPP;       // pixels per store instruction
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
#pragma unroll
        for (int j = 0; j < NCB; ++j) {
            float4 v = make_float4(acc[i][j][0] + bv[j].x, acc[i][j][1] + bv[j].y, acc[i][j][2] + bv[j].z, acc[i][j][3] + bv[j].w);
            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4 *>(s_ep + lm * kEpPitch + 16 * j + 4 * lq) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int r = 0; r < 16 / PPI; ++r) {
            const int pl = r * PPI + lane / LPP, c4 = lane % LPP;
            const float4 v = *reinterpret_cast<const float4 *>(s_ep + pl * kEpPitch + 4 * c4);
            const int64_t px = px0 + i * 16 + pl;
            if (px < n_px) {
                const int ox = (int)(px % p.Wo);
                const int64_t t = px / p.Wo;
                const int oy = (int)(t % p.Ho);
                const int64_t b = t / p.Ho;
                const int64_t yy = (int64_t)oy * p.osy + ooy, xx = (int64_t)ox * p.osx + oox;
                *reinterpret_cast<float4 *>(y + ((b * Hy + yy) * Wy + xx) * p.cout_total + p.co_off + co_out_base + 4 * c4) = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    return true;
}

template <int NPB, int NCB>
__global__ void __launch_bounds__(256) conv1x1_f32(const float *__restrict__ x, const float4 *__restrict__ wp, const float *__restrict__ bias,
                                                   float *__restrict__ y, ConvParamsF p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: what derives from it stays in SGPRs
    const int lm = lane & 15, lq = lane >> 4;
    const int64_t n_px = (int64_t)p.B * p.Ho * p.Wo;
    const int64_t px0 = (int64_t)blockIdx.x * (16 * NPB);
    const int nb0 = ((int)blockIdx.y * 4 + wave) * NCB;  // first 16-channel block of this wave
    if (nb0 * 16 >= p.Cout_pad) return;                   // (no barrier in this kernel)
    const int nsl = p.Cin / 16;
    const float *xp[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        int64_t px = px0 + i * 16 + lm;
        px = px < n_px ? px : n_px - 1;  // padding pixels read the last row; their results are not stored
        xp[i] = x + px * p.cin_stride + lq * 4;
    }
    const float4 *wq[NCB];
#pragma unroll
    for (int j = 0; j < NCB; ++j) {
        const int nb = nb0 + j < (p.Cout_pad >> 4) ? nb0 + j : (p.Cout_pad >> 4) - 1;
        wq[j] = wp + (int64_t)nb * nsl * 64 + lane;
    }
    f32x4 acc[NPB][NCB];
#pragma unroll
    for (int i = 0; i < NPB; ++i)
#pragma unroll
        for (int j = 0; j < NCB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 b0[NPB], b1[NPB], a0[NCB], a1[NCB];  // two slices in flight (four measured slower: registers, not latency, bind)
    auto load = [&](int s, float4(&bb)[NPB], float4(&aa)[NCB]) {
        const int sc = s < nsl ? s : nsl - 1;
#pragma unroll
        for (int i = 0; i < NPB; ++i) bb[i] = *reinterpret_cast<const float4 *>(xp[i] + sc * 16);
#pragma unroll
        for (int j = 0; j < NCB; ++j) aa[j] = wq[j][(int64_t)sc * 64];
    };
// (one accumulator's four k-steps back to back: measured faster than walking all accumulators per k-step, 36 vs 45 us on
// 128->256 at 180 x 180)
#define FD_PW_STEP(BB, AA)                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    _Pragma("unroll") for (int i = 0; i < NPB; ++i) _Pragma("unroll") for (int j = 0; j < NCB; ++j) {                  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(AA[j].x, BB[i].x, acc[i][j], 0, 0, 0);                        \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(AA[j].y, BB[i].y, acc[i][j], 0, 0, 0);                        \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(AA[j].z, BB[i].z, acc[i][j], 0, 0, 0);                        \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(AA[j].w, BB[i].w, acc[i][j], 0, 0, 0);                        \
    }                                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);
    load(0, b0, a0);
    for (int s = 0; s < nsl; s += 2) {
        load(s + 1, b1, a1);
        FD_PW_STEP(b0, a0)
        if (s + 1 >= nsl) break;
        load(s + 2, b0, a0);
        FD_PW_STEP(b1, a1)
    }
#undef FD_PW_STEP
    // epilogue: lane (pixel lm of block i, quad lq) holds channels 16 (nb0 + j) + 4 lq .. + 3 of its pixel
    __shared__ __attribute__((aligned(16))) float s_ep[4][16 * kEpPitch];
    if (pointwise_epilogue_rows<NPB, NCB>(acc, p, bias, y, px0, n_px, nb0, lane, s_ep[wave])) return;
    const int64_t Hy = (int64_t)p.Ho * p.osy, Wy = (int64_t)p.Wo * p.osx;
    const bool wide = ((p.cout_total | p.co_off) & 3) == 0;
#pragma unroll
    for (int j = 0; j < NCB; ++j) {
        const int co = (nb0 + j) * 16 + lq * 4;
        int co_out = co, limit = p.Cout_real, ooy = p.ooy, oox = p.oox;
        if (co >= p.Cout_real) continue;
        if (p.cout_sub > 0) {  // pixel shuffle: (sub-convolution, channel)
            const int sub = co / p.cout_sub;
            co_out = co - sub * p.cout_sub;
            limit = p.cout_sub;
            ooy = sub / p.osx;
            oox = sub - ooy * p.osx;
        }
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) {
            bv.x = bias[co_out];
            if (co_out + 1 < limit) bv.y = bias[co_out + 1];
            if (co_out + 2 < limit) bv.z = bias[co_out + 2];
            if (co_out + 3 < limit) bv.w = bias[co_out + 3];
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            const int64_t px = px0 + i * 16 + lm;
            if (px >= n_px) continue;
            const int ox = (int)(px % p.Wo);
            const int64_t t = px / p.Wo;
            const int oy = (int)(t % p.Ho);
            const int64_t b = t / p.Ho;
            float4 v = make_float4(acc[i][j][0] + bv.x, acc[i][j][1] + bv.y, acc[i][j][2] + bv.z, acc[i][j][3] + bv.w);
            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            const int64_t yy = (int64_t)oy * p.osy + ooy, xx = (int64_t)ox * p.osx + oox;
            float *dst = y + ((b * Hy + yy) * Wy + xx) * p.cout_total + p.co_off + co_out;
            if (wide && co_out + 3 < limit) {
                *reinterpret_cast<float4 *>(dst) = v;
            } else {
                dst[0] = v.x;
                if (co_out + 1 < limit) dst[1] = v.y;
                if (co_out + 2 < limit) dst[2] = v.z;
                if (co_out + 3 < limit) dst[3] = v.w;
            }
        }
    }
}

// The same GEMM with the pixel operand shared through LDS (round 6).  In conv1x1_f32 every wave loads the workgroup's pixels itself: 4 pixel +
// 4 weight fragments of 1 KB per wave and 16-channel slice for 64 MFMAs -- 32 KB per 2048 MFMA cycles and CU, which is what the vector-memory
// path delivers (16 B/clk): 45 % MFMA-busy.  Here the four waves stage the pixels once (1 KB each) and read them back from LDS: 5 KB per wave
// and slice from memory instead of 8.  Same MFMA sequence per output element: bit-identical to conv1x1_f32 (the plan times both per layer).
template <int NPB, int NCB>
__global__ void __launch_bounds__(256) conv1x1_lds_f32(const float *__restrict__ x, const float4 *__restrict__ wp, const float *__restrict__ bias,
                                                   float *__restrict__ y, ConvParamsF p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: what derives from it stays in SGPRs
    const int lm = lane & 15, lq = lane >> 4;
    const int64_t n_px = (int64_t)p.B * p.Ho * p.Wo;
    const int64_t px0 = (int64_t)blockIdx.x * (16 * NPB);
    const int nb0 = ((int)blockIdx.y * 4 + wave) * NCB;  // first 16-channel block of this wave
    const bool active = nb0 * 16 < p.Cout_pad;            // (an idle wave still stages pixels and meets the barriers)
    const int nsl = p.Cin / 16;
    // the workgroup's 16 * NPB pixels x 16 channels of a slice, staged ONCE per slice by all four waves (thread -> pixel tid / 4, piece tid % 4)
    // instead of loaded by each of them: pixel pitch 80 bytes (the 16 pixels of a block land on distinct bank groups), two buffers
    constexpr int kPitch = 20;  // floats
    __shared__ __attribute__((aligned(16))) float s_x[2][16 * NPB * kPitch];
    const int spx = tid >> 2, spc = tid & 3;
    const float *xsrc = nullptr;
    if (spx < 16 * NPB) {
        int64_t px = px0 + spx;
        px = px < n_px ? px : n_px - 1;
        xsrc = x + px * p.cin_stride + spc * 4;
    }
    const float4 *wq[NCB];
#pragma unroll
    for (int j = 0; j < NCB; ++j) {
        const int nb = nb0 + j < (p.Cout_pad >> 4) ? nb0 + j : (p.Cout_pad >> 4) - 1;
        wq[j] = wp + (int64_t)nb * nsl * 64 + lane;
    }
    f32x4 acc[NPB][NCB];
#pragma unroll
    for (int i = 0; i < NPB; ++i)
#pragma unroll
        for (int j = 0; j < NCB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 a0[NCB], a1[NCB], xr;
    auto load_w = [&](int s, float4(&aa)[NCB]) {
        const int sc = s < nsl ? s : nsl - 1;
#pragma unroll
        for (int j = 0; j < NCB; ++j) aa[j] = wq[j][(int64_t)sc * 64];
    };
    auto load_x = [&](int s) {
        const int sc = s < nsl ? s : nsl - 1;
        if (xsrc) xr = *reinterpret_cast<const float4 *>(xsrc + sc * 16);
    };
    auto stage_x = [&](int buf) {
        if (xsrc) *reinterpret_cast<float4 *>(&s_x[buf][spx * kPitch + spc * 4]) = xr;
    };
#define FD_PW_STEP(BUF, AA)                                                                                            \
    {                                                                                                                  \
        float4 bb[NPB];                                                                                                \
        _Pragma("unroll") for (int i = 0; i < NPB; ++i) bb[i] = *reinterpret_cast<const float4 *>(&s_x[BUF][(i * 16 + lm) * kPitch + lq * 4]); \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        _Pragma("unroll") for (int i = 0; i < NPB; ++i) _Pragma("unroll") for (int j = 0; j < NCB; ++j) {              \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(AA[j].x, bb[i].x, acc[i][j], 0, 0, 0);                    \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(AA[j].y, bb[i].y, acc[i][j], 0, 0, 0);                    \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(AA[j].z, bb[i].z, acc[i][j], 0, 0, 0);                    \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(AA[j].w, bb[i].w, acc[i][j], 0, 0, 0);                    \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    }
    load_x(0);
    load_w(0, a0);
    stage_x(0);
    __syncthreads();
    for (int s = 0; s < nsl; s += 2) {
        load_x(s + 1);       // slice s + 1 travels under the MFMAs of slice s ...
        load_w(s + 1, a1);
        FD_PW_STEP(0, a0)
        stage_x(1);          // ... and is handed over behind them (buffer 1 was last read in the previous iteration, before its barrier)
        __syncthreads();
        if (s + 1 >= nsl) break;
        load_x(s + 2);
        load_w(s + 2, a0);
        FD_PW_STEP(1, a1)
        stage_x(0);
        __syncthreads();
    }
#undef FD_PW_STEP
    if (!active) return;
    // epilogue: lane (pixel lm of block i, quad lq) holds channels 16 (nb0 + j) + 4 lq .. + 3 of its pixel
    __shared__ __attribute__((aligned(16))) float s_ep[4][16 * kEpPitch];
    if (pointwise_epilogue_rows<NPB, NCB>(acc, p, bias, y, px0, n_px, nb0, lane, s_ep[wave])) return;
    const int64_t Hy = (int64_t)p.Ho * p.osy, Wy = (int64_t)p.Wo * p.osx;
    const bool wide = ((p.cout_total | p.co_off) & 3) == 0;
#pragma unroll
    for (int j = 0; j < NCB; ++j) {
        const int co = (nb0 + j) * 16 + lq * 4;
        int co_out = co, limit = p.Cout_real, ooy = p.ooy, oox = p.oox;
        if (co >= p.Cout_real) continue;
        if (p.cout_sub > 0) {  // pixel shuffle: (sub-convolution, channel)
            const int sub = co / p.cout_sub;
            co_out = co - sub * p.cout_sub;
            limit = p.cout_sub;
            ooy = sub / p.osx;
            oox = sub - ooy * p.osx;
        }
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) {
            bv.x = bias[co_out];
            if (co_out + 1 < limit) bv.y = bias[co_out + 1];
            if (co_out + 2 < limit) bv.z = bias[co_out + 2];
            if (co_out + 3 < limit) bv.w = bias[co_out + 3];
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            const int64_t px = px0 + i * 16 + lm;
            if (px >= n_px) continue;
            const int ox = (int)(px % p.Wo);
            const int64_t t = px / p.Wo;
            const int oy = (int)(t % p.Ho);
            const int64_t b = t / p.Ho;
            float4 v = make_float4(acc[i][j][0] + bv.x, acc[i][j][1] + bv.y, acc[i][j][2] + bv.z, acc[i][j][3] + bv.w);
            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            const int64_t yy = (int64_t)oy * p.osy + ooy, xx = (int64_t)ox * p.osx + oox;
            float *dst = y + ((b * Hy + yy) * Wy + xx) * p.cout_total + p.co_off + co_out;
            if (wide && co_out + 3 < limit) {
                *reinterpret_cast<float4 *>(dst) = v;
            } else {
                dst[0] = v.x;
                if (co_out + 1 < limit) dst[1] = v.y;
                if (co_out + 2 < limit) dst[2] = v.z;
                if (co_out + 3 < limit) dst[3] = v.w;
            }
        }
    }
}

// ... and with the WHOLE pixel rows staged once (round 6): the per-slice forms above fetch 64 bytes of each 4 Cin-byte pixel row per slice -- half
// of every 128-byte line, eight times over -- and wait for it in every slice.  Here a workgroup copies its 16 NPB rows (all channels) to LDS with
// fully coalesced 16-byte loads, one barrier, and the slice loop reads pixels from LDS and only the weights from memory.  LDS = 16 NPB x (Cin + 4)
// floats (+ the epilogue tile): launched only when that fits.  Same MFMA sequence: bit-identical.
template <int NPB, int NCB>
__global__ void __launch_bounds__(256) conv1x1_rows_f32(const float *__restrict__ x, const float4 *__restrict__ wp, const float *__restrict__ bias,
                                                   float *__restrict__ y, ConvParamsF p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: what derives from it stays in SGPRs
    const int lm = lane & 15, lq = lane >> 4;
    const int64_t n_px = (int64_t)p.B * p.Ho * p.Wo;
    const int64_t px0 = (int64_t)blockIdx.x * (16 * NPB);
    const int nb0 = ((int)blockIdx.y * 4 + wave) * NCB;  // first 16-channel block of this wave
    const bool active = nb0 * 16 < p.Cout_pad;            // (an idle wave still stages pixels and meets the barriers)
    const int nsl = p.Cin / 16;
    // the workgroup's 16 * NPB pixel rows, ALL input channels, staged once: lane -> 16 consecutive bytes of a row, so every load instruction
    // moves whole 128-byte lines (the per-slice forms fetch 64 bytes of every pixel row per slice); row pitch Cin + 4 floats
    extern __shared__ __attribute__((aligned(16))) float s_rows[];
    const int pitch = p.Cin + 4;
    {
        const int c4n = p.Cin / 4;
        for (int t = tid; t < 16 * NPB * c4n; t += 256) {
            const int pr = t / c4n, c4 = t - pr * c4n;
            int64_t px = px0 + pr;
            px = px < n_px ? px : n_px - 1;
            *reinterpret_cast<float4 *>(&s_rows[pr * pitch + 4 * c4]) = *reinterpret_cast<const float4 *>(x + px * p.cin_stride + 4 * c4);
        }
    }
    const float4 *wq[NCB];
#pragma unroll
    for (int j = 0; j < NCB; ++j) {
        const int nb = nb0 + j < (p.Cout_pad >> 4) ? nb0 + j : (p.Cout_pad >> 4) - 1;
        wq[j] = wp + (int64_t)nb * nsl * 64 + lane;
    }
    f32x4 acc[NPB][NCB];
#pragma unroll
    for (int i = 0; i < NPB; ++i)
#pragma unroll
        for (int j = 0; j < NCB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 a0[NCB], a1[NCB];
    auto load_w = [&](int s, float4(&aa)[NCB]) {
        const int sc = s < nsl ? s : nsl - 1;
#pragma unroll
        for (int j = 0; j < NCB; ++j) aa[j] = wq[j][(int64_t)sc * 64];
    };
#define FD_PW_STEP(S, AA)                                                                                              \
    {                                                                                                                  \
        float4 bb[NPB];                                                                                                \
        _Pragma("unroll") for (int i = 0; i < NPB; ++i) bb[i] = *reinterpret_cast<const float4 *>(&s_rows[(i * 16 + lm) * pitch + (S) * 16 + lq * 4]); \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        _Pragma("unroll") for (int i = 0; i < NPB; ++i) _Pragma("unroll") for (int j = 0; j < NCB; ++j) {              \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(AA[j].x, bb[i].x, acc[i][j], 0, 0, 0);                    \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(AA[j].y, bb[i].y, acc[i][j], 0, 0, 0);                    \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(AA[j].z, bb[i].z, acc[i][j], 0, 0, 0);                    \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(AA[j].w, bb[i].w, acc[i][j], 0, 0, 0);                    \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    }
    load_w(0, a0);
    __syncthreads();
    if (!active) return;  // (no barrier below)
    for (int s = 0; s < nsl; s += 2) {
        load_w(s + 1, a1);
        FD_PW_STEP(s, a0)
        if (s + 1 >= nsl) break;
        load_w(s + 2, a0);
        FD_PW_STEP(s + 1, a1)
    }
#undef FD_PW_STEP
    // epilogue: lane (pixel lm of block i, quad lq) holds channels 16 (nb0 + j) + 4 lq .. + 3 of its pixel
    __shared__ __attribute__((aligned(16))) float s_ep[4][16 * kEpPitch];
    if (pointwise_epilogue_rows<NPB, NCB>(acc, p, bias, y, px0, n_px, nb0, lane, s_ep[wave])) return;
    const int64_t Hy = (int64_t)p.Ho * p.osy, Wy = (int64_t)p.Wo * p.osx;
    const bool wide = ((p.cout_total | p.co_off) & 3) == 0;
#pragma unroll
    for (int j = 0; j < NCB; ++j) {
        const int co = (nb0 + j) * 16 + lq * 4;
        int co_out = co, limit = p.Cout_real, ooy = p.ooy, oox = p.oox;
        if (co >= p.Cout_real) continue;
        if (p.cout_sub > 0) {  // pixel shuffle: (sub-convolution, channel)
            const int sub = co / p.cout_sub;
            co_out = co - sub * p.cout_sub;
            limit = p.cout_sub;
            ooy = sub / p.osx;
            oox = sub - ooy * p.osx;
        }
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) {
            bv.x = bias[co_out];
            if (co_out + 1 < limit) bv.y = bias[co_out + 1];
            if (co_out + 2 < limit) bv.z = bias[co_out + 2];
            if (co_out + 3 < limit) bv.w = bias[co_out + 3];
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            const int64_t px = px0 + i * 16 + lm;
            if (px >= n_px) continue;
            const int ox = (int)(px % p.Wo);
            const int64_t t = px / p.Wo;
            const int oy = (int)(t % p.Ho);
            const int64_t b = t / p.Ho;
            float4 v = make_float4(acc[i][j][0] + bv.x, acc[i][j][1] + bv.y, acc[i][j][2] + bv.z, acc[i][j][3] + bv.w);
            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            const int64_t yy = (int64_t)oy * p.osy + ooy, xx = (int64_t)ox * p.osx + oox;
            float *dst = y + ((b * Hy + yy) * Wy + xx) * p.cout_total + p.co_off + co_out;
            if (wide && co_out + 3 < limit) {
                *reinterpret_cast<float4 *>(dst) = v;
            } else {
                dst[0] = v.x;
                if (co_out + 1 < limit) dst[1] = v.y;
                if (co_out + 2 < limit) dst[2] = v.z;
                if (co_out + 3 < limit) dst[3] = v.w;
            }
        }
    }
}

// pointwise variants: (pixel blocks, channel blocks per wave); tile ids kNumTiles + 1 ...
constexpr int kNumPointwise = 8;
inline int launch_pointwise(const float *x, const void *wp, const float *bias, float *y, const ConvParamsF &p, int variant, hipStream_t stream) {
    const int64_t n_px = (int64_t)p.B * p.Ho * p.Wo;
    auto go = [&](auto kern, int npb, int ncb) {
        dim3 grid((unsigned)((n_px + 16 * npb - 1) / (16 * npb)), (unsigned)((p.Cout_pad / 16 + 4 * ncb - 1) / (4 * ncb)));
        hipLaunchKernelGGL(kern, grid, dim3(256), 0, stream, x, (const float4 *)wp, bias, y, p);
    };
    auto go_rows = [&](auto kern, int npb, int ncb) -> int {
        const size_t lds = (size_t)16 * npb * (p.Cin + 4) * sizeof(float);
        if (lds > 46 * 1024) return 0;  // (next to the 17-KB epilogue tile: 64 KB of static + dynamic LDS at most, no opt-in needed)
        dim3 grid((unsigned)((n_px + 16 * npb - 1) / (16 * npb)), (unsigned)((p.Cout_pad / 16 + 4 * ncb - 1) / (4 * ncb)));
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, x, (const float4 *)wp, bias, y, p);
        return 1;
    };
    switch (variant) {
        case 0: go(conv1x1_f32<4, 4>, 4, 4); break;  // 64 pixels x 256 channels per workgroup
        case 1: go(conv1x1_f32<2, 4>, 2, 4); break;  // 32 x 256
        case 2: go(conv1x1_f32<4, 2>, 4, 2); break;  // 64 x 128
        case 3: go(conv1x1_f32<2, 2>, 2, 2); break;  // 32 x 128
        case 4: go(conv1x1_lds_f32<4, 4>, 4, 4); break;  // 64 x 256, pixels shared through LDS
        case 5: go(conv1x1_lds_f32<2, 4>, 2, 4); break;  // 32 x 256, pixels shared through LDS
        case 6: return go_rows(conv1x1_rows_f32<4, 4>, 4, 4);  // 64 x 256, whole pixel rows staged once
        case 7: return go_rows(conv1x1_rows_f32<2, 4>, 2, 4);  // 32 x 256, whole pixel rows staged once
        default: return 0;
    }
    return 1;
}

template <int KS, int S>
int dispatch_tile(const float *x, const void *wp, const float *bias, float *y, const ConvParamsF &p, int tile, hipStream_t stream) {
    const int n_cu = fd::device_cu_count();
    int best = -1;
    double bc = 1e30;
    if constexpr (KS == 1) {
        // plain NHWC input (no channel groups): the pointwise GEMM kernel; tile 0 picks the variant that fills the chip
        if (p.groups <= 1 && (tile == 0 || tile > kNumTiles)) {
            int v = tile > kNumTiles ? tile - kNumTiles - 1 : -1;
            if (v < 0) {
                const int64_t n_px = (int64_t)p.B * p.Ho * p.Wo;
                const int wide_wgs = (int)((n_px + 63) / 64) * ((p.Cout_pad / 16 + 15) / 16);
                v = wide_wgs >= n_cu ? 0 : 3;  // (measured: 128->256 at 180 x 180 -> variant 0, 256->256 at 90 x 90 -> variant 3)
            }
            return launch_pointwise(x, wp, bias, y, p, v, stream);
        }
    }
    if (tile >= 1 && tile <= kNumTiles) {
        best = tile - 1;  // the caller's choice (e.g. measured once per layer shape by the host plan)
    } else {
        for (int i = 0; i < kNumTiles; ++i) {
            const double c = tile_cost(p.Ho, p.Wo, p.B, p.Cout_real, kTiles[i], n_cu);
            if (c < bc) { bc = c; best = i; }
        }
    }
    switch (best) {
        case 0: launch_f32<KS, S, 12, 12, 2, 4>(x, wp, bias, y, p, stream); break;
        case 1: launch_f32<KS, S, 8, 16, 2, 4>(x, wp, bias, y, p, stream); break;
        case 2: launch_f32<KS, S, 5, 15, 2, 4>(x, wp, bias, y, p, stream); break;
        case 3: launch_f32<KS, S, 8, 8, 2, 4>(x, wp, bias, y, p, stream); break;
        case 4: launch_f32<KS, S, 12, 12, 1, 4>(x, wp, bias, y, p, stream); break;
        case 5: launch_f32<KS, S, 8, 16, 1, 4>(x, wp, bias, y, p, stream); break;
        case 6: launch_f32<KS, S, 5, 15, 1, 4>(x, wp, bias, y, p, stream); break;
        case 7: launch_f32<KS, S, 8, 8, 1, 4>(x, wp, bias, y, p, stream); break;
        case 8: launch_f32<KS, S, 8, 16, 2, 1>(x, wp, bias, y, p, stream); break;
        case 9: launch_f32<KS, S, 8, 16, 1, 1>(x, wp, bias, y, p, stream); break;
        case 10: launch_f32<KS, S, 8, 8, 2, 1>(x, wp, bias, y, p, stream); break;
        case 11: launch_f32<KS, S, 8, 8, 1, 1>(x, wp, bias, y, p, stream); break;
        case 12: launch_f32<KS, S, 8, 16, 2, 2>(x, wp, bias, y, p, stream); break;
        case 13: launch_f32<KS, S, 8, 16, 4, 2>(x, wp, bias, y, p, stream); break;
        case 14: launch_f32<KS, S, 8, 8, 2, 2>(x, wp, bias, y, p, stream); break;
        case 15: launch_f32<KS, S, 8, 8, 4, 2>(x, wp, bias, y, p, stream); break;
        default: return 0;
    }
    return 1;
}

}  // namespace

extern "C" size_t fd_conv2d_f32_packed_weight_bytes(int cout, int cin, int ks) {
    if (cout <= 0 || cin <= 0 || cin % 16 || (ks != 1 && ks != 3)) return 0;
    const size_t cout_pad = ((size_t)cout + 63) / 64 * 64;
    return cout_pad * cin * ks * ks * 4;
}

// w: [cout][cin][ks][ks] float32 (torch Conv2d layout) -> [cout_pad/16][cin/16][tap][lane][4] float32:
// lane = (co & 15) + 16 * q holds w[co][16 s + 4 q + 0..3][tap]
extern "C" int fd_conv2d_f32_pack_weight(const float *w, int cout, int cin, int ks, void *dst) {
    FD_REQUIRE(w && dst, "fd_conv2d_f32_pack_weight: null argument");
    FD_REQUIRE(cin % 16 == 0 && (ks == 1 || ks == 3) && cout > 0, "fd_conv2d_f32_pack_weight: need cin %% 16 == 0 and ks in {1,3}");
    const int cout_pad = (cout + 63) / 64 * 64, nsl = cin / 16, taps = ks * ks;
    float *d = (float *)dst;
    for (int nb = 0; nb < cout_pad / 16; ++nb)
        for (int s = 0; s < nsl; ++s)
            for (int tap = 0; tap < taps; ++tap)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        const int co = nb * 16 + (lane & 15);
                        const int ci = s * 16 + 4 * (lane >> 4) + j;
                        d[((((int64_t)nb * nsl + s) * taps + tap) * 64 + lane) * 4 + j] =
                            co < cout ? w[(((int64_t)co * cin + ci) * ks + tap / ks) * ks + tap % ks] : 0.0f;
                    }
    return FD_OK;
}

extern "C" int fd_conv2d_f32_num_tiles(void) { return kNumTiles + kNumPointwise; }  // (the last kNumPointwise ids: 1x1 convolutions only)

extern "C" int fd_conv2d_nhwc_f32(const float *x, int B, int H, int W, int cin, const void *wpacked, const float *bias, int cout, int ks,
                                  int stride, int pad, int relu, float *y, int cout_total, int co_off, int osy, int osx, int ooy, int oox,
                                  int tile, fd_stream_t stream) {
    FD_REQUIRE(x && wpacked && y, "fd_conv2d_nhwc_f32: null argument");
    FD_REQUIRE(cin % 16 == 0 && cin >= 16, "fd_conv2d_nhwc_f32: cin must be a multiple of 16 (got %d)", cin);
    FD_REQUIRE((ks == 3 && (stride == 1 || stride == 2) && pad == 1) || (ks == 1 && stride == 1 && pad == 0),
               "fd_conv2d_nhwc_f32: supported: 3x3 stride 1|2 pad 1, 1x1 stride 1 pad 0");
    FD_REQUIRE(B > 0 && H > 0 && W > 0 && cout > 0 && osy >= 1 && osx >= 1, "fd_conv2d_nhwc_f32: bad shape");
    FD_REQUIRE(co_off >= 0 && co_off + cout <= cout_total, "fd_conv2d_nhwc_f32: channel window [%d, %d) outside the %d output channels", co_off,
               co_off + cout, cout_total);
    FD_REQUIRE(tile >= 0 && tile <= kNumTiles + (ks == 1 ? kNumPointwise : 0), "fd_conv2d_nhwc_f32: tile must be 0 (library heuristic) or 1..%d for this kernel size",
               kNumTiles + (ks == 1 ? kNumPointwise : 0));
    ConvParamsF p;
    p.B = B; p.H = H; p.W = W; p.Cin = cin;
    p.Ho = (H + 2 * pad - ks) / stride + 1;
    p.Wo = (W + 2 * pad - ks) / stride + 1;
    p.Cout_real = cout;
    p.Cout_pad = (cout + 63) / 64 * 64;
    p.cout_total = cout_total; p.co_off = co_off; p.pad = pad; p.relu = relu;
    p.osy = osy; p.osx = osx; p.ooy = ooy; p.oox = oox;
    p.tiles_x = p.tiles_y = 0;
    p.cin_stride = cin; p.cout_sub = 0; p.groups = 1;
    hipStream_t s = fd::as_stream(stream);
    int ok;
    if (ks == 3 && stride == 1) ok = dispatch_tile<3, 1>(x, wpacked, bias, y, p, tile, s);
    else if (ks == 3) ok = dispatch_tile<3, 2>(x, wpacked, bias, y, p, tile, s);
    else ok = dispatch_tile<1, 1>(x, wpacked, bias, y, p, tile, s);
    FD_REQUIRE(ok, "fd_conv2d_nhwc_f32: no tile shape for cout %d", cout);
    return fd::check_launch("fd_conv2d_nhwc_f32");
}

// ConvTranspose2d(k, stride k) (the RPN's upsampling deblock, det3d/models/necks/rpn.py:98-110) as ONE 1x1 convolution to
// k*k*cout_sub virtual channels with a pixel-shuffle epilogue: the input is staged once for all k*k sub-convolutions.
// Weights: fd_conv2d_f32_pack_weight of [(dy, dx, co), cin, 1, 1]; bias [cout_sub].
extern "C" int fd_conv2d_shuffle_nhwc_f32(const float *x, int B, int H, int W, int cin, const void *wpacked, const float *bias, int cout_sub, int k,
                                          int relu, float *y, int cout_total, int co_off, int tile, fd_stream_t stream) {
    FD_REQUIRE(x && wpacked && y, "fd_conv2d_shuffle_nhwc_f32: null argument");
    FD_REQUIRE(cin % 16 == 0 && cin >= 16 && cout_sub > 0 && cout_sub % 4 == 0 && k >= 2 && k <= 4 && B > 0 && H > 0 && W > 0,
               "fd_conv2d_shuffle_nhwc_f32: need cin %% 16 == 0, cout_sub %% 4 == 0, 2 <= k <= 4");
    FD_REQUIRE(tile >= 0 && tile <= kNumTiles + kNumPointwise, "fd_conv2d_shuffle_nhwc_f32: tile must be 0..%d", kNumTiles + kNumPointwise);
    FD_REQUIRE(co_off >= 0 && co_off + cout_sub <= cout_total, "fd_conv2d_shuffle_nhwc_f32: channel window [%d, %d) outside the %d output channels",
               co_off, co_off + cout_sub, cout_total);
    ConvParamsF p;
    p.B = B; p.H = H; p.W = W; p.Cin = cin; p.Ho = H; p.Wo = W;
    p.Cout_real = cout_sub * k * k;
    p.Cout_pad = (p.Cout_real + 63) / 64 * 64;
    p.cout_total = cout_total; p.co_off = co_off; p.pad = 0; p.relu = relu;
    p.osy = k; p.osx = k; p.ooy = 0; p.oox = 0;
    p.tiles_x = p.tiles_y = 0;
    p.cin_stride = cin; p.cout_sub = cout_sub; p.groups = 1;
    FD_REQUIRE((dispatch_tile<1, 1>(x, wpacked, bias, y, p, tile, fd::as_stream(stream))), "fd_conv2d_shuffle_nhwc_f32: no tile shape");
    return fd::check_launch("fd_conv2d_shuffle_nhwc_f32");
}

// Grouped 3x3 stride-1 convolution with at most 16 outputs per group: the final convolutions of the CenterHead branches
// (det3d/models/bbox_heads/center_head.py:129-143: reg 2, height 1, dim 3, rot 2, vel 2T, hm 1, each on its own 64 channels)
// in one launch, instead of one block-diagonal dense convolution that multiplies by zeros 5/6 of the time.
// x [B,H,W,groups*cin_g]; weights: fd_conv2d_f32_pack_weight of [groups*16, cin_g, 3, 3] (each group padded to 16 outputs);
// bias [groups*16]; counts_host[g] real outputs of group g, written back to back from co_off.
extern "C" int fd_conv2d_grouped_nhwc_f32(const float *x, int B, int H, int W, int groups, int cin_g, const void *wpacked, const float *bias,
                                          const int *counts_host, int relu, float *y, int cout_total, int co_off, int tile, fd_stream_t stream) {
    FD_REQUIRE(x && wpacked && y && counts_host, "fd_conv2d_grouped_nhwc_f32: null argument");
    FD_REQUIRE(groups >= 1 && groups <= 8 && cin_g % 16 == 0 && cin_g >= 16 && B > 0 && H > 0 && W > 0,
               "fd_conv2d_grouped_nhwc_f32: need 1 <= groups <= 8 and cin_g %% 16 == 0");
    ConvParamsF p;
    p.B = B; p.H = H; p.W = W; p.Cin = cin_g; p.Ho = H; p.Wo = W;
    p.Cout_real = groups * 16;
    p.Cout_pad = (p.Cout_real + 63) / 64 * 64;
    p.cout_total = cout_total; p.co_off = co_off; p.pad = 1; p.relu = relu;
    p.osy = p.osx = 1; p.ooy = p.oox = 0;
    p.tiles_x = p.tiles_y = 0;
    p.cin_stride = groups * cin_g; p.cout_sub = 0; p.groups = groups > 1 ? groups : 2;  // (groups == 1 still takes the table path)
    int off = 0;
    for (int g = 0; g < 8; ++g) {
        const int c = g < groups ? counts_host[g] : 0;
        FD_REQUIRE(c >= 0 && c <= 16, "fd_conv2d_grouped_nhwc_f32: a group has at most 16 outputs (got %d)", c);
        p.gcnt[g] = c;
        p.goff[g] = off;
        off += c;
    }
    FD_REQUIRE(co_off >= 0 && co_off + off <= cout_total, "fd_conv2d_grouped_nhwc_f32: channel window [%d, %d) outside the %d output channels", co_off,
               co_off + off, cout_total);
    if (groups == 1) p.groups = 1, p.Cout_real = counts_host[0];
    // one 16-channel block per workgroup: the pixel-split layouts (tile 10 = 8x16 pixels, 12 = 8x8)
    if (tile != 12) tile = 10;
    hipStream_t s = fd::as_stream(stream);
    if (p.groups > 1) {
        if (tile == 10) launch_f32<3, 1, 8, 16, 1, 1>(x, wpacked, bias, y, p, s);
        else launch_f32<3, 1, 8, 8, 1, 1>(x, wpacked, bias, y, p, s);
    } else {
        FD_REQUIRE((dispatch_tile<3, 1>(x, wpacked, bias, y, p, 0, s)), "fd_conv2d_grouped_nhwc_f32: no tile shape");
    }
    return fd::check_launch("fd_conv2d_grouped_nhwc_f32");
}
