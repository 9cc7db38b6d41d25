// Dense 2-D convolution for gfx950, bf16 in / fp32 accumulate / bf16 out, NHWC, fused bias (+folded BN) + ReLU.
//
// Serves the RPN neck and CenterHead convolutions (det3d/models/necks/rpn.py:81-140,
// det3d/models/bbox_heads/center_head.py:104-143,344-349) in the bf16 configuration: MIOpen in this ROCm build only
// offers im2col + GEMM (or a naive kernel) for bf16 on gfx950, which spends more time in Im2d2Col / bias / ReLU
// passes than in the GEMM.  This is a direct implicit GEMM on v_mfma_f32_32x32x16_bf16:
//
//   * a workgroup computes an 8 x 16 patch of output pixels (M = 128) x NT output channels;
//   * for every 32-channel slice of the input it stages the (8-1)*S+KS by (16-1)*S+KS input halo patch in LDS once
//     (16-byte chunks XOR-swizzled by pixel so the ds_read_b128 fragment reads are bank-conflict free) and all
//     KS*KS taps read their A fragments from it -- the 9x im2col blow-up never touches HBM or L2;
//   * the next slice's patch is fetched into registers while the current slice's MFMAs run (double-buffered LDS);
//   * weights are pre-packed in MFMA B-fragment order and streamed from L2 with one coalesced 1 KiB load per
//     wave instruction (a layer's weights are <= 1.2 MB);
//   * epilogue: + bias, ReLU, round to bf16, store at (y*osy+ooy, x*osx+oox, co_off + co) of a tensor with
//     cout_total channels -- which is how the RPN's concat and the 2x2 stride-2 transposed conv (four 1x1 convs
//     writing interleaved pixels) are expressed without extra passes.
#include "fd_common.h"

#include <algorithm>
#include <map>
#include <mutex>
#include <utility>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TH = 8, TW = 16;  // output pixels per workgroup (M = 128)

// tuning builds (tools/probes/build_trace.sh): thread 0 of every workgroup accumulates cycles per phase:
// [0] prologue (first patch in LDS), [1] step loops, [2] slice hand-over (store + next loads), [3] barrier, [4] epilogue, [5] whole
// ablation builds (tools/probes/build_exp.sh fd_conv2d <tag> -DFD_CONV_EXP=<mask>; results are wrong on purpose): 1 = no weight
// fragment loads inside the step loop, 2 = no A-fragment LDS reads inside the step loop, 4 = no MFMAs
#ifndef FD_CONV_EXP
#define FD_CONV_EXP 0
#endif
#ifndef FD_CONV_RING  // weight fragments in flight per column block (tuning builds)
#define FD_CONV_RING 6
#endif
#ifndef FD_CONV_WAVES  // waves per SIMD the stride-1 tile kernel is compiled for (tuning builds)
#define FD_CONV_WAVES 2
#endif
#ifdef FD_V2_TRACE
__device__ unsigned long long *g_ctrace;
#define FD_CT(var) const unsigned long long var = __builtin_readcyclecounter()
#define FD_CADD(i, v) cacc[i] += (v)
#else
#define FD_CT(var)
#define FD_CADD(i, v)
#endif

__device__ inline unsigned short f2bf(float v) {
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

struct ConvParams {
    int B, H, W, Cin, Ho, Wo, Cout_pad, Cout_real, cout_total, co_off, pad, relu;
    int osy, osx, ooy, oox;  // output pixel mapping
    int tiles_x, tiles_y;
    unsigned w_bytes;
    // mixed tiling of a stride-1 layer (GEN instances): fh x fw whole 8 x 16 tiles, then the columns right of them as tiles of th_r x rw
    // pixels, then the rows below as tiles of bh x tw_b pixels (each <= 128 pixels, sides <= 32); see mixed_tiles()
    int fh, fw, n_right, th_r, rw, n_bottom, tw_b, bh, tiles_img;
};

#ifdef FD_V2_TRACE
#define FD_EPI_TRACE_DECL unsigned long long c_mid = 0;
#define FD_EPI_TRACE_ARG , &c_mid
#else
#define FD_EPI_TRACE_DECL
#define FD_EPI_TRACE_ARG
#endif

// Epilogue shared by the tile and the strip kernel: + bias, ReLU, round to bf16, store at (y*osy+ooy, x*osx+oox, co_off + co).
// `out_pixel(m, oy, ox)` maps pixel m (0 .. 127) of the workgroup to its output coordinates and says whether it exists.
// C/D layout of 32x32: col (channel) = lane & 31, row (pixel) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
template <int WMT, int WNT, int WAVES_M, int WAVES_N, typename PixelFn>
__device__ __forceinline__ void conv_bf16_epilogue(const f32x16 (&acc)[WMT][WNT], unsigned char *smem, const ConvParams &p, int b,
                                                   const float *__restrict__ bias, unsigned short *__restrict__ y, const PixelFn &out_pixel,
                                                   unsigned long long *t_mid = nullptr) {
    constexpr int NT = WNT * WAVES_N * 32, M = WMT * WAVES_M * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int lm = lane & 31, lk = lane >> 5;
    const int n0 = blockIdx.y * NT + wn * WNT * 32;
    const int64_t Hy = (int64_t)p.Ho * p.osy, Wy = (int64_t)p.Wo * p.osx;
    const bool wide = ((p.cout_total | p.co_off) & 7) == 0 && (p.Cout_real & 7) == 0;  // 16-byte aligned channel runs
    if (wide) {
        // A lane owns one channel of 16 pixels, i.e. 2-byte stores if written directly (64 store instructions per
        // wave, issue-bound).  Transpose the tile through LDS instead and store 16 bytes (8 channels) per lane.
        unsigned short *s_out = reinterpret_cast<unsigned short *>(smem);  // [128 pixels][NT channels] bf16
        // Lanes l and l ^ 1 hold adjacent channels of the same 16 pixels.  Per pixel pair (r, r + 1): bias, ReLU, one packed
        // convert (round to nearest even, v_cvt_pk_bf16_f32), the neighbour's pair by a DPP quad permute, one byte permute ->
        // the even lane owns the (channel pair, pixel r) dword, the odd lane (channel pair, pixel r + 1): 4-byte LDS writes at
        // compile-time offsets from a per-lane base.  (The older form rounded by hand and exchanged through ds_bpermute: 5.1 k
        // cycles for this phase, tools/conv_bf16_trace.py.)
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        const bool odd = lm & 1;
        const unsigned sel = odd ? 0x03020706u : 0x05040100u;  // v_perm_b32(nb, mine), low half first: odd -> {nb.hi16, mine.hi16}, even -> {mine.lo16, nb.lo16}
#pragma unroll
        for (int j = 0; j < WNT; ++j) {
            const int cl = (wn * WNT + j) * 32 + lm;  // channel inside the block
            const int co = blockIdx.y * NT + cl;
            const float bv = (bias && co < p.Cout_real) ? bias[co] : 0.0f;
#pragma unroll
            for (int i = 0; i < WMT; ++i) {
                unsigned char *dst = reinterpret_cast<unsigned char *>(s_out) + ((((wm * WMT + i) * 32 + (odd ? 1 : 0) + 4 * lk) * NT + (cl & ~1)) << 1);
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f32x2_t v = {acc[i][j][r] + bv, acc[i][j][r + 1] + bv};
                    if (p.relu) { v[0] = fmaxf(v[0], 0.0f); v[1] = fmaxf(v[1], 0.0f); }
                    const unsigned mine = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
                    const unsigned nb = (unsigned)__builtin_amdgcn_mov_dpp((int)mine, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
                    const unsigned packed = __builtin_amdgcn_perm(nb, mine, sel);
                    // pixel of this dword: (r & 3) + 8 (r >> 2) + 4 lk (+ 1 on the odd lane: inside `dst`)
                    *reinterpret_cast<unsigned *>(dst + ((((r & 3) + 8 * (r >> 2)) * NT) << 1)) = packed;
                }
            }
        }
        __syncthreads();
#ifdef FD_V2_TRACE
        if (t_mid) *t_mid = __builtin_readcyclecounter();
#endif
        constexpr int C8 = NT / 8;
        for (int id = tid; id < M * C8; id += 256) {
            const int m = id / C8, c8 = id - m * C8;
            const int co = blockIdx.y * NT + c8 * 8;
            int oy, ox;
            if (out_pixel(m, oy, ox) && co < p.Cout_real) {
                const int64_t yy = (int64_t)oy * p.osy + p.ooy, xx = (int64_t)ox * p.osx + p.oox;
                *reinterpret_cast<uint4 *>(y + (((int64_t)b * Hy + yy) * Wy + xx) * p.cout_total + p.co_off + co) =
                    *reinterpret_cast<const uint4 *>(s_out + m * NT + c8 * 8);
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
        const int co = n0 + j * 32 + lm;
        const float bv = (bias && co < p.Cout_real) ? bias[co] : 0.0f;
#pragma unroll
        for (int i = 0; i < WMT; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (wm * WMT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                int oy, ox;
                if (out_pixel(m, oy, ox) && co < p.Cout_real) {
                    float v = acc[i][j][r] + bv;
                    if (p.relu) v = fmaxf(v, 0.0f);
                    const int64_t yy = (int64_t)oy * p.osy + p.ooy, xx = (int64_t)ox * p.osx + p.oox;
                    y[(((int64_t)b * Hy + yy) * Wy + xx) * p.cout_total + p.co_off + co] = f2bf(v);
                }
            }
        }
    }
}

// WMT x WNT MFMA tiles (32 x 32) per wave, waves arranged WAVES_M x WAVES_N (product 4); M tile = 128 pixels
// GEN (stride 1 only): the workgroup's tile is one of the three shapes of the layer's MIXED tiling (ConvParams::fh ...): 180 x 180 is
// 23 x 12 = 276 ragged 8 x 16 tiles but 242 whole ones + 6 of 32 x 4 + 6 of 4 x 32 = 254 -- one workgroup per compute unit for one
// map, 508 for two on the 512 slots two resident workgroups per unit give (the 40 workgroups of 552 that had to wait for a slot made
// a two-map layer take 28 us instead of 18: start times by s_memrealtime, tools/conv_bf16_trace.py).  The tile's height / width and
// the patch pitch become run-time values: pixel -> (row, column) by a 16-bit reciprocal, one A-fragment base per kernel ROW (the
// column and k-half offsets stay immediates), nothing new inside the step loop; the arithmetic order per output pixel is unchanged
// (bit-identical to the fixed tiling, tested).
template <int KS>
constexpr int gen_max_patch() { return KS == 1 ? 128 : 204; }  // (4 x 32 tile: 6 x 34 patch pixels)

template <int KS, int S, int WMT, int WNT, int WAVES_M, int WAVES_N, bool GEN = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(S == 1 ? FD_CONV_WAVES : 2))) conv2d_nhwc_bf16(const unsigned short *__restrict__ x, const bf16x8 *__restrict__ wp,
                                                        const float *__restrict__ bias, unsigned short *__restrict__ y, ConvParams p) {
    static_assert(WAVES_M * WAVES_N == 4 && WMT * WAVES_M * 32 == TH * TW, "tile shape");
    static_assert(!GEN || S == 1, "mixed tiles: stride 1 only");
    constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS, PP = GEN ? gen_max_patch<KS>() : PH * PW;
    constexpr int NCHUNK16 = PP * 4;                    // 16-byte chunks per 32-channel patch (GEN: of the largest one)
    constexpr int NLOAD = (NCHUNK16 + 255) / 256;
    constexpr int NT = WNT * WAVES_N * 32;              // output channels per workgroup
    constexpr int PIXB = 80;                            // LDS bytes per patch pixel (see store_slice)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 x PP x PIXB bytes

    // the wave index as a SCALAR: everything derived from it (this wave's channel block, hence the weight-fragment offsets) then
    // lives in SGPRs.  With `tid >> 6` the compiler keeps it in a VGPR and every fragment load of the step loop costs a 64-bit
    // VALU add + v_readfirstlane + s_nop 4 -- and a VALU instruction waits for the MFMAs in flight on its SIMD
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef FD_V2_TRACE
    unsigned long long cacc[6] = {0, 0, 0, 0, 0, 0};
#endif
    FD_CT(c_start);
#ifdef FD_V2_TRACE
    const unsigned long long rt_start = __builtin_amdgcn_s_memrealtime();  // 100 MHz, one counter for the whole device
#endif
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    int b, oy0, ox0, th = TH, tw = TW;  // (uniform)
    if constexpr (GEN) {
        int t = blockIdx.x % p.tiles_img;
        b = blockIdx.x / p.tiles_img;
        const int n_core = p.fh * p.fw;
        if (t < n_core) {
            oy0 = (t / p.fw) * TH; ox0 = (t % p.fw) * TW;
        } else if (t < n_core + p.n_right) {
            oy0 = (t - n_core) * p.th_r; ox0 = p.fw * TW;
            th = min(p.th_r, p.fh * TH - oy0); tw = p.rw;
        } else {
            oy0 = p.fh * TH; ox0 = (t - n_core - p.n_right) * p.tw_b;
            th = p.bh; tw = min(p.tw_b, p.Wo - ox0);
        }
    } else {
        int t = blockIdx.x;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y; b = t / p.tiles_y;
        oy0 = ty * TH; ox0 = tx * TW;
    }
    const int n0 = blockIdx.y * NT + wn * WNT * 32;     // first output channel of this wave
    const int iy0 = oy0 * S - p.pad, ix0 = ox0 * S - p.pad;
    const int nslices = p.Cin / 32;
    // GEN: run-time tile and patch shape; x / d for x < 256, d <= 34 as (x * ceil(65536 / d)) >> 16 (exact: x (ceil - 65536 / d) < 65536 / d)
    const int pw = GEN ? tw + KS - 1 : PW, n_pix = th * tw, n_patch = (th + KS - 1) * pw;
    const unsigned rcp_tw = (65536u + (unsigned)tw - 1u) / (unsigned)tw, rcp_pw = (65536u + (unsigned)pw - 1u) / (unsigned)pw;

    // per-thread patch chunk assignment: chunk id -> (pixel, q)
    uint4 stage[NLOAD];
    auto load_slice = [&](int s) {
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int id = tid + i * 256;
            stage[i] = make_uint4(0u, 0u, 0u, 0u);
            if (id < NCHUNK16) {
                const int pix = id >> 2, q = id & 3;
                int py, px;
                if constexpr (GEN) {
                    py = (int)(((unsigned)pix * rcp_pw) >> 16);
                    px = pix - py * pw;
                } else {
                    py = pix / PW; px = pix % PW;
                }
                const int iy = iy0 + py, ix = ix0 + px;
                if ((!GEN || pix < n_patch) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                    stage[i] = *reinterpret_cast<const uint4 *>(x + (((int64_t)b * p.H + iy) * p.W + ix) * p.Cin + s * 32 + q * 8);
            }
        }
    };
    // LDS patch layout: pixel pitch PIXB = 80 bytes (64 of data + 16 of padding).  20 dwords per pixel: 16 pixels with distinct
    // (index mod 16) hit 16 distinct 4-dword bank groups (20 p mod 64 = 4 (5 p mod 16)), so the ds_read_b128 fragment reads are
    // conflict free -- and, unlike the XOR swizzle this replaces, the address is LINEAR in the pixel index: the A fragment of
    // every (tap, k-half) step is one per-lane base register + an immediate offset, no VALU instruction in the step loop.
    auto store_slice = [&](int buf) {
        unsigned char *dst = smem + buf * (PP * PIXB);
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int id = tid + i * 256;
            if (id < NCHUNK16) {
                const int pix = id >> 2, q = id & 3;
                *reinterpret_cast<uint4 *>(dst + pix * PIXB + (q << 4)) = stage[i];
            }
        }
    };

    f32x16 acc[WMT][WNT];
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // A-fragment geometry: lane -> output pixel of each of its WMT M-tiles, k half = lane >> 5
    const int lm = lane & 31, lk = lane >> 5;
    int prow[WMT], pcol[WMT];
#pragma unroll
    for (int i = 0; i < WMT; ++i) {
        int m = (wm * WMT + i) * 32 + lm;  // pixel index inside the tile, row-major
        if constexpr (GEN) {
            m = m < n_pix ? m : 0;         // (slots past the tile read pixel 0; their results are not stored)
            prow[i] = (int)(((unsigned)m * rcp_tw) >> 16);
            pcol[i] = m - prow[i] * tw;
        } else {
            prow[i] = (m / TW) * S;
            pcol[i] = (m % TW) * S;
        }
    }
    // packed weights: [Cout_pad/32][slice][tap][ksub][lane] x 16 bytes
    const int64_t w_nt_stride = (int64_t)nslices * KS * KS * 2 * 64;

    load_slice(0);
    store_slice(0);
    __syncthreads();
    // The weight stream of one 32-channel column block is a linear array over (slice, tap, k-half): a register ring
    // keeps RING fragments in flight ahead of their MFMAs (an L2 hit costs ~1000 cycles, a (tap, k-half) step has only
    // WMT*WNT MFMAs), and the ring keeps running across the slice barrier.
    constexpr int ITERS = KS * KS * 2;
    constexpr int RING = (ITERS % 6 == 0) ? FD_CONV_RING : 2;
    const int total_iters = nslices * ITERS;
    // fragment loads: buffer loads with the wave-uniform part of the address in the scalar offset (no vector instruction per load)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8 *>(wp), 0, (int)p.w_bytes, 0x00020000);
    const unsigned lane16 = lane * 16;
    auto wfrag = [&](int j, int it) {
        const int soff = __builtin_amdgcn_readfirstlane((int)((((n0 >> 5) + j) * w_nt_stride + (int64_t)it * 64) * 16));
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, soff, 0));
    };
    bf16x8 bw[RING][WNT];
#pragma unroll
    for (int r = 0; r < RING; ++r)
#pragma unroll
        for (int j = 0; j < WNT; ++j) bw[r][j] = wfrag(j, r < total_iters ? r : 0);
    // per-lane LDS base of the A fragments (tap (0, 0), k-half 0) relative to the slice buffer; a step adds a compile-time offset
    unsigned abase[WMT];
#pragma unroll
    for (int i = 0; i < WMT; ++i) abase[i] = (unsigned)((prow[i] * pw + pcol[i]) * PIXB + lk * 16);
    const unsigned row_pitch = (unsigned)(pw * PIXB);  // GEN: the step loop adds it per kernel row (one base register per row)
    if (nslices > 1) load_slice(1);  // registers hold slice s+1 while slice s is computed
    FD_CT(c_pro);
    FD_CADD(0, c_pro - c_start);
    for (int s = 0; s < nslices; ++s) {
        FD_CT(c0);
        // A fragments are read ahead of their MFMAs (LDS latency ~130 cycles, a step has 128 cycles of MFMA).  Their LDS
        // offsets (tap, k-half and swizzle applied) are per-lane constants computed once (aoff): no address arithmetic in the loop --
        // every VALU instruction here is a slot the MFMAs do not get.  The reads and the weight requests sit BETWEEN the MFMA groups
        // of a step (fences): an MFMA occupies the pipe for 32 cycles but issues in 4, what is issued in its shadow is free; lumped
        // after the step's MFMAs the same instructions cost ~100 cycles per 128-cycle step (tools/conv_bf16_trace.py: 4870 cycles
        // per slice for 2304 of MFMA before).
        unsigned ab[GEN ? KS : 1][WMT];
#pragma unroll
        for (int i = 0; i < WMT; ++i) {
            ab[0][i] = abase[i] + (unsigned)((s & 1) * (PP * PIXB));
            if constexpr (GEN) {
#pragma unroll
                for (int ky = 1; ky < KS; ++ky) ab[ky][i] = ab[0][i] + (unsigned)ky * row_pitch;
            }
        }
        auto read_a = [&](int it, bf16x8(&a)[WMT]) {
            const int tap = it >> 1, ks = it & 1;
            if constexpr (GEN) {
                const unsigned imm = (unsigned)((tap % KS) * PIXB + ks * 32);  // compile-time: the ds_read offset field
#pragma unroll
                for (int i = 0; i < WMT; ++i) a[i] = *reinterpret_cast<const bf16x8 *>(smem + ab[tap / KS][i] + imm);
            } else {
                const unsigned imm = (unsigned)(((tap / KS) * PW + tap % KS) * PIXB + ks * 32);  // compile-time: the ds_read offset field
#pragma unroll
                for (int i = 0; i < WMT; ++i) a[i] = *reinterpret_cast<const bf16x8 *>(smem + ab[0][i] + imm);
            }
        };
        bf16x8 a[3][WMT];  // ring of three: the fragments of step it + 2 are requested in the middle of step it (1.5 steps = 190 cycles of lead)
        read_a(0, a[0]);
        if (ITERS > 1) read_a(1, a[1]);
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int slot = it % RING;  // ITERS % RING == 0, so the slot sequence is the same in every slice
            const int nxt = s * ITERS + it + RING;
            const int nsel = nxt < total_iters ? nxt : 0;
#pragma unroll
            for (int j = 0; j < WNT; ++j) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < WMT; ++i) {
                    if constexpr (FD_CONV_EXP & 4) asm volatile("" : "+v"(acc[i][j]) : "v"(a[it % 3][i]), "v"(bw[slot][j]));
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[it % 3][i], bw[slot][j], acc[i][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(FD_CONV_EXP & 2)) {
                    if (j == 0 && it + 2 < ITERS) read_a(it + 2, a[(it + 2) % 3]);
                }
                if constexpr (!(FD_CONV_EXP & 1)) bw[slot][j] = wfrag(j, nsel);  // the slot just consumed takes the fragment of RING steps ahead
            }
        }
        // Patch loads go at the END of a slice: vector-memory loads return in order, so a patch load issued at the top
        // would sit in front of this slice's weight-ring loads and stall them for its whole latency; issued here it has a
        // full slice of MFMAs to land and only the ring loads of the next slice's first steps queue behind it.
        FD_CT(c1);
        if (s + 1 < nslices) store_slice((s + 1) & 1);
        if (s + 2 < nslices) load_slice(s + 2);
        FD_CT(c2);
        __syncthreads();
        FD_CT(c3);
        FD_CADD(1, c1 - c0); FD_CADD(2, c2 - c1); FD_CADD(3, c3 - c2);
    }
    FD_CT(c_epi);
    const auto out_pixel = [&](int m, int &oy, int &ox) {  // pixel m of the tile, row-major
        if constexpr (GEN) {
            const int r = (int)(((unsigned)m * rcp_tw) >> 16);
            oy = oy0 + r;
            ox = ox0 + (m - r * tw);
            return m < n_pix;
        } else {
            oy = oy0 + m / TW;
            ox = ox0 + m % TW;
            return oy < p.Ho && ox < p.Wo;
        }
    };
    FD_EPI_TRACE_DECL
    conv_bf16_epilogue<WMT, WNT, WAVES_M, WAVES_N>(acc, smem, p, b, bias, y, out_pixel FD_EPI_TRACE_ARG);
#ifdef FD_V2_TRACE
    if (tid == 0 && g_ctrace) {
        const unsigned long long c_end = __builtin_readcyclecounter();
        unsigned long long *o = g_ctrace + (size_t)(blockIdx.x + blockIdx.y * gridDim.x) * 8;
        o[0] = cacc[0]; o[1] = cacc[1]; o[2] = cacc[2]; o[3] = cacc[3]; o[4] = c_end - c_epi; o[5] = c_end - c_start; o[6] = c_mid - c_epi; o[7] = rt_start;
    }
#endif
}

template <int KS, int S, int WMT, int WNT, int WAVES_M, int WAVES_N, bool GEN = false>
void launch_conv(const void *x, const void *wp, const float *bias, void *y, const ConvParams &p, hipStream_t stream) {
    constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS;
    constexpr int NT = WNT * WAVES_N * 32;
    size_t lds = 2 * (size_t)(GEN ? gen_max_patch<KS>() : PH * PW) * 80;
    if (lds < (size_t)TH * TW * NT * 2) lds = (size_t)TH * TW * NT * 2;  // the epilogue transposes the tile through LDS
    auto kern = conv2d_nhwc_bf16<KS, S, WMT, WNT, WAVES_M, WAVES_N, GEN>;
    static std::atomic<uint64_t> lds_set{0};
    if (lds > 65536) (void)fd::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, lds_set);
    dim3 grid((unsigned)((GEN ? p.tiles_img : p.tiles_x * p.tiles_y) * p.B), (unsigned)(p.Cout_pad / NT));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, (const unsigned short *)x, (const bf16x8 *)wp, bias, (unsigned short *)y, p);
}


// ------------------------------------------------------------------------------------------------------------ strip kernel
// Stride-1 convolutions on STRIPS: a workgroup takes M = 128 CONSECUTIVE pixels of the row-major pixel order of one image
// instead of an 8 x 16 tile.  180 x 180 is 23 x 12 = 276 tiles for 256 compute units -- 20 units carry two workgroups and the
// layer waits for them (tools: 242 workgroups 15.5 us, 276 workgroups 20.0 us) -- but 254 strips with 0.3 % padding: one
// workgroup per unit, one round.  A strip touches at most three image rows; the input patch is, per patch row, the hull of the
// columns its pixels need, rows packed one after the other in LDS (80-byte pixel pitch as above).  Because every A-fragment
// address is a per-lane base (one per kernel row ky: the patch rows have different origins) + a compile-time offset, an
// arbitrary pixel -> patch mapping costs nothing in the step loop; the loop itself, the weight ring and the arithmetic order
// are those of the tile kernel (bit-identical results).
struct StripParams {
    int spi;      // strips per image
    int max_pix;  // patch pixels of the largest strip: the slice buffers are max_pix * 80 bytes apart
};

template <int KS, int WMT, int WNT, int WAVES_M, int WAVES_N, int NLOAD>
__global__ void __launch_bounds__(256) conv2d_strip_bf16(const unsigned short *__restrict__ x, const bf16x8 *__restrict__ wp,
                                                         const float *__restrict__ bias, unsigned short *__restrict__ y, ConvParams p, StripParams sp) {
    constexpr int M = WMT * WAVES_M * 32;
    static_assert(WAVES_M * WAVES_N == 4 && M == 128, "strip shape");
    constexpr int NT = WNT * WAVES_N * 32;
    constexpr int PIXB = 80;
    constexpr int RS_MAX = 3, PH_MAX = RS_MAX + KS - 1;  // image rows a strip may touch (host: Wo >= 64), patch rows
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 x max_pix x PIXB bytes

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef FD_V2_TRACE
    unsigned long long cacc[6] = {0, 0, 0, 0, 0, 0};
#endif
    FD_CT(c_start);
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int b = blockIdx.x / sp.spi, sidx = blockIdx.x - b * sp.spi;
    const int n0 = blockIdx.y * NT + wn * WNT * 32;
    const int nslices = p.Cin / 32;
    const int Wo = p.Wo, HW = p.Ho * p.Wo;
    // strip geometry (all wave-uniform)
    const int p0 = sidx * M;
    const int nvalid = HW - p0 < M ? HW - p0 : M;
    const int y0 = p0 / Wo, x0 = p0 - y0 * Wo;
    const int y1 = (p0 + nvalid - 1) / Wo, x1 = p0 + nvalid - 1 - y1 * Wo;
    int lo[PH_MAX], rowpix[PH_MAX + 1];  // patch row r = input row y0 - pad + r holds patch columns lo[r] .. (column c = input column c - pad)
    rowpix[0] = 0;
#pragma unroll
    for (int r = 0; r < PH_MAX; ++r) {
        int l = 1 << 30, h = -1;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            const int yy = y0 + r - ky;  // the output row that reads patch row r with kernel row ky
            if (yy >= y0 && yy <= y1) {
                const int xa = yy == y0 ? x0 : 0, xb = yy == y1 ? x1 : Wo - 1;
                l = xa < l ? xa : l;
                h = xb + KS - 1 > h ? xb + KS - 1 : h;
            }
        }
        lo[r] = h >= 0 ? l : 0;
        rowpix[r + 1] = rowpix[r] + (h >= 0 ? h - l + 1 : 0);
    }
    const int total_chunks = rowpix[PH_MAX] * 4;
    auto row_origin = [&](int r, int &lo_r, int &rp_r) {  // (per-lane r)
        lo_r = lo[0]; rp_r = 0;
#pragma unroll
        for (int rr = 1; rr < PH_MAX; ++rr)
            if (r >= rr) { lo_r = lo[rr]; rp_r = rowpix[rr]; }
    };
    // patch chunks of this thread: chunk id = 4 * (LDS pixel index) + 16-byte quarter; global byte offset inside the image (slice 0)
    // or a value past the buffer (-> zeros: padding pixels), LDS byte offset or ~0 (no chunk)
    unsigned goff[NLOAD], loff[NLOAD];
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
        const int id = tid + i * 256;
        int r = 0;
#pragma unroll
        for (int rr = 1; rr < PH_MAX; ++rr) r += id >= rowpix[rr] * 4 ? 1 : 0;
        int lo_r, rp_r;
        row_origin(r, lo_r, rp_r);
        const int o = id - rp_r * 4, q = o & 3;
        const int iy = y0 - p.pad + r, ix = lo_r + (o >> 2) - p.pad;
        const bool have = id < total_chunks;
        const bool inb = have && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        goff[i] = inb ? (unsigned)(((iy * p.W + ix) * p.Cin + q * 8) * 2) : 0x80000000u;
        loff[i] = have ? (unsigned)((id >> 2) * PIXB + (q << 4)) : ~0u;
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(x) + (int64_t)b * p.H * p.W * p.Cin, 0,
                                                                          p.H * p.W * p.Cin * 2, 0x00020000);
    const unsigned buf_bytes = (unsigned)sp.max_pix * PIXB;
    uint4 stage[NLOAD];
    auto load_slice = [&](int s) {
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) stage[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xrs, goff[i], s * 64, 0));
    };
    auto store_slice = [&](int buf) {
        unsigned char *dst = smem + buf * buf_bytes;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i)
            if (loff[i] != ~0u) *reinterpret_cast<uint4 *>(dst + loff[i]) = stage[i];
    };

    f32x16 acc[WMT][WNT];
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // A-fragment bases: lane -> pixel lm of each of its WMT M-tiles (pixels past the strip's end read the last pixel's window),
    // one base per kernel row
    const int lm = lane & 31, lk = lane >> 5;
    unsigned abase[WMT][KS];
#pragma unroll
    for (int i = 0; i < WMT; ++i) {
        int m = (wm * WMT + i) * 32 + lm;
        m = m < nvalid ? m : nvalid - 1;
        const int xx = x0 + m;
        const int dy = (xx >= Wo ? 1 : 0) + (xx >= 2 * Wo ? 1 : 0);
        const int xc = xx - dy * Wo;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            int lo_r, rp_r;
            row_origin(dy + ky, lo_r, rp_r);
            abase[i][ky] = (unsigned)((rp_r + xc - lo_r) * PIXB + lk * 16);
        }
    }
    const int64_t w_nt_stride = (int64_t)nslices * KS * KS * 2 * 64;

    load_slice(0);
    constexpr int ITERS = KS * KS * 2;
    constexpr int RING = (ITERS % 6 == 0) ? FD_CONV_RING : 2;
    const int total_iters = nslices * ITERS;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8 *>(wp), 0, (int)p.w_bytes, 0x00020000);
    const unsigned lane16 = lane * 16;
    auto wfrag = [&](int j, int it) {
        const int soff = __builtin_amdgcn_readfirstlane((int)((((n0 >> 5) + j) * w_nt_stride + (int64_t)it * 64) * 16));
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, soff, 0));
    };
    // the weight ring is requested BEHIND the first patch slice (loads return in order) and before its LDS stores wait for it:
    // the two round trips overlap
    bf16x8 bw[RING][WNT];
#pragma unroll
    for (int r = 0; r < RING; ++r)
#pragma unroll
        for (int j = 0; j < WNT; ++j) bw[r][j] = wfrag(j, r < total_iters ? r : 0);
    store_slice(0);
    __syncthreads();
    if (nslices > 1) load_slice(1);
    FD_CT(c_pro);
    FD_CADD(0, c_pro - c_start);
    for (int s = 0; s < nslices; ++s) {
        FD_CT(c0);
        unsigned ab[WMT][KS];
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) ab[i][ky] = abase[i][ky] + (unsigned)(s & 1) * buf_bytes;
        auto read_a = [&](int it, bf16x8(&a)[WMT]) {
            const int tap = it >> 1, ks = it & 1;
            const unsigned imm = (unsigned)((tap % KS) * PIXB + ks * 32);  // compile-time: the ds_read offset field
#pragma unroll
            for (int i = 0; i < WMT; ++i) a[i] = *reinterpret_cast<const bf16x8 *>(smem + ab[i][tap / KS] + imm);
        };
        bf16x8 a[3][WMT];
        read_a(0, a[0]);
        if (ITERS > 1) read_a(1, a[1]);
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int slot = it % RING;
            const int nxt = s * ITERS + it + RING;
            const int nsel = nxt < total_iters ? nxt : 0;
#pragma unroll
            for (int j = 0; j < WNT; ++j) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < WMT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[it % 3][i], bw[slot][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (j == 0 && it + 2 < ITERS) read_a(it + 2, a[(it + 2) % 3]);
                bw[slot][j] = wfrag(j, nsel);
            }
        }
        FD_CT(c1);
        if (s + 1 < nslices) store_slice((s + 1) & 1);
        if (s + 2 < nslices) load_slice(s + 2);
        FD_CT(c2);
        __syncthreads();
        FD_CT(c3);
        FD_CADD(1, c1 - c0); FD_CADD(2, c2 - c1); FD_CADD(3, c3 - c2);
    }
    FD_CT(c_epi);
    const auto out_pixel = [&](int m, int &oy, int &ox) {  // pixel m of the strip = pixel p0 + m of the image
        const int xx = x0 + m;
        const int dy = (xx >= Wo ? 1 : 0) + (xx >= 2 * Wo ? 1 : 0);
        oy = y0 + dy;
        ox = xx - dy * Wo;
        return m < nvalid;
    };
    FD_EPI_TRACE_DECL
    conv_bf16_epilogue<WMT, WNT, WAVES_M, WAVES_N>(acc, smem, p, b, bias, y, out_pixel FD_EPI_TRACE_ARG);
#ifdef FD_V2_TRACE
    if (tid == 0 && g_ctrace) {
        const unsigned long long c_end = __builtin_readcyclecounter();
        unsigned long long *o = g_ctrace + (size_t)(blockIdx.x + blockIdx.y * gridDim.x) * 8;
        o[0] = cacc[0]; o[1] = cacc[1]; o[2] = cacc[2]; o[3] = cacc[3]; o[4] = c_end - c_epi; o[5] = c_end - c_start; o[6] = c_mid - c_epi; o[7] = c_end;
    }
#endif
}

// patch pixels of the largest strip of an H x W image (the host side of the kernel's hull computation)
template <int KS>
int strip_max_pix(int Ho, int Wo) {
    constexpr int M = 128;
    const int HW = Ho * Wo;
    int best = 0;
    for (int p0 = 0; p0 < HW; p0 += M) {
        const int nvalid = HW - p0 < M ? HW - p0 : M;
        const int y0 = p0 / Wo, x0 = p0 - y0 * Wo, y1 = (p0 + nvalid - 1) / Wo, x1 = p0 + nvalid - 1 - y1 * Wo;
        int pix = 0;
        for (int r = 0; r < y1 - y0 + KS; ++r) {
            int l = 1 << 30, h = -1;
            for (int ky = 0; ky < KS; ++ky) {
                const int yy = y0 + r - ky;
                if (yy >= y0 && yy <= y1) {
                    const int xa = yy == y0 ? x0 : 0, xb = yy == y1 ? x1 : Wo - 1;
                    l = xa < l ? xa : l;
                    h = xb + KS - 1 > h ? xb + KS - 1 : h;
                }
            }
            if (h >= 0) pix += h - l + 1;
        }
        best = pix > best ? pix : best;
    }
    return best;
}

template <int KS, int WMT, int WNT, int WAVES_M, int WAVES_N, int NLOAD>
void launch_strip(const void *x, const void *wp, const float *bias, void *y, const ConvParams &p, const StripParams &sp, hipStream_t stream) {
    constexpr int NT = WNT * WAVES_N * 32;
    size_t lds = 2 * (size_t)sp.max_pix * 80;
    if (lds < (size_t)128 * NT * 2) lds = (size_t)128 * NT * 2;
    auto kern = conv2d_strip_bf16<KS, WMT, WNT, WAVES_M, WAVES_N, NLOAD>;
    static std::atomic<uint64_t> lds_set{0};
    if (lds > 65536) (void)fd::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, lds_set);
    dim3 grid((unsigned)(sp.spi * p.B), (unsigned)(p.Cout_pad / NT));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, (const unsigned short *)x, (const bf16x8 *)wp, bias, (unsigned short *)y, p, sp);
}

// stride-1 layers: the strip kernel when the image is wide enough for its three-row patch, the patch fits and the chunk table
// has a variant; channel block = the widest that still gives every compute unit a workgroup.  Returns 0 -> the tile kernel.
//
// OPT-IN (fd_tuning_set("conv_strip", 1)): measured on MI355X (profiles/round4_bf16_dense_strips.txt) a layer alone is 15-20 %
// faster on strips when they make it one round of workgroups (128 -> 128 at 180 x 180 20.3 -> 17.1 us, 256 -> 256 at 90 x 90
// 22.6 -> 18.1) and a sweep's latency drops 1.5 % (config 3: 1.80 -> 1.77 ms) -- but with four sweeps in flight, which is what
// the throughput figure measures, the strips' 80-116 KB of LDS per workgroup (a tile: 29 KB) keep the other sweeps' kernels off
// the compute unit: config 3 800 -> 771 sweeps/s, config 5 (270 x 270: several rounds either way, bigger patch) 463 -> 443.
template <int KS>
int dispatch_strip(const void *x, const void *wp, const float *bias, void *y, const ConvParams &p, hipStream_t stream) {
    if (fd::tuning(fd::kTuneConvStrip) <= 0 || p.Wo < 64 || p.Ho != p.H || p.Wo != p.W) return 0;
    if ((int64_t)p.H * p.W * p.Cin * 2 >= (1ll << 31)) return 0;  // (the patch loads address one image through a 2 GB buffer window)
    StripParams sp;
    sp.spi = (p.Ho * p.Wo + 127) / 128;
    // (the hull walk is O(strips) on the host; cache per shape)
    static std::mutex mu;
    static std::map<std::pair<int, int>, int> cache;
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = cache.find({p.Ho, p.Wo});
        if (it == cache.end()) it = cache.emplace(std::make_pair(p.Ho, p.Wo), strip_max_pix<KS>(p.Ho, p.Wo)).first;
        sp.max_pix = it->second;
    }
    const int need = (sp.max_pix * 4 + 255) / 256;
    if (need > 12 || 2 * (size_t)sp.max_pix * 80 > 160 * 1024) return 0;
    const int force = fd::tuning(fd::kTuneConvNT);
    const int n_cu = fd::device_cu_count();
    const int64_t strips = (int64_t)sp.spi * p.B;
    int nt = 32;
    if (p.Cout_pad % 64 == 0) nt = 64;
    if (p.Cout_pad % 128 == 0 && (strips * (p.Cout_pad / 128) * 4 >= 3 * n_cu || p.Cout_pad % 64 != 0)) nt = 128;
    if (nt == 64 && strips * (p.Cout_pad / 64) * 4 < 3 * n_cu) nt = 32;
    // a layer of one or two slices that is several rounds of workgroups anyway is prologue-bound: the strip's patch (hull of up
    // to three full-width rows) costs more to fetch than a tile's (64 -> 384 at 180 x 180: 26.4 us on tiles, 28.5 on strips)
    if (p.Cin < 128 && strips * (p.Cout_pad / nt) > 2 * n_cu && force == 0) return 0;
    if (force == 128 && p.Cout_pad % 128 == 0) nt = 128;
    if (force == 64 && p.Cout_pad % 64 == 0) nt = 64;
    if (force == 32) nt = 32;
#define FD_STRIP(NL)                                                                                    \
    if (nt == 128) launch_strip<KS, 2, 2, 2, 2, NL>(x, wp, bias, y, p, sp, stream);                     \
    else if (nt == 64) launch_strip<KS, 1, 2, 4, 1, NL>(x, wp, bias, y, p, sp, stream);                 \
    else launch_strip<KS, 1, 1, 4, 1, NL>(x, wp, bias, y, p, sp, stream);
    if (need <= 4) { FD_STRIP(4) }
    else if (need <= 8) { FD_STRIP(8) }
    else { FD_STRIP(12) }
#undef FD_STRIP
    return 1;
}

template <int KS, int S>
int dispatch_nt(const void *x, const void *wp, const float *bias, void *y, const ConvParams &p, hipStream_t stream) {
    const int force = fd::tuning(fd::kTuneConvNT);  // tuning override
    if constexpr (S == 1) {
        // the mixed tiling where it has fewer tiles than the ragged grid ("conv_strip" = -1: never -- A/B runs and the identity test)
        if (p.tiles_img < p.tiles_x * p.tiles_y && fd::tuning(fd::kTuneConvStrip) >= 0) {
            if (p.Cout_pad % 128 == 0 && force != 64 && force != 32) launch_conv<KS, S, 2, 2, 2, 2, true>(x, wp, bias, y, p, stream);
            else if (p.Cout_pad % 64 == 0 && force != 32) launch_conv<KS, S, 1, 2, 4, 1, true>(x, wp, bias, y, p, stream);
            else launch_conv<KS, S, 1, 1, 4, 1, true>(x, wp, bias, y, p, stream);
            return 1;
        }
    }
    if (p.Cout_pad % 128 == 0 && force != 64 && force != 32) launch_conv<KS, S, 2, 2, 2, 2>(x, wp, bias, y, p, stream);
    else if (p.Cout_pad % 64 == 0 && force != 32) launch_conv<KS, S, 1, 2, 4, 1>(x, wp, bias, y, p, stream);
    else launch_conv<KS, S, 1, 1, 4, 1>(x, wp, bias, y, p, stream);
    return 1;
}

// the mixed tiling of an Ho x Wo map (ConvParams::fh ...): whole 8 x 16 tiles, the columns right of them, the rows below
void mixed_tiles(ConvParams &p) {
    p.fh = p.Ho / TH; p.fw = p.Wo / TW;
    p.rw = p.Wo - p.fw * TW;
    p.bh = p.Ho - p.fh * TH;
    p.th_r = p.rw > 0 ? std::min(32, (TH * TW) / p.rw) : 1;
    p.n_right = (p.rw > 0 && p.fh > 0) ? (p.fh * TH + p.th_r - 1) / p.th_r : 0;
    p.tw_b = p.bh > 0 ? std::min(32, (TH * TW) / p.bh) : 1;
    p.n_bottom = p.bh > 0 ? (p.Wo + p.tw_b - 1) / p.tw_b : 0;
    p.tiles_img = p.fh * p.fw + p.n_right + p.n_bottom;
}

}  // namespace

#ifdef FD_V2_TRACE
extern "C" int fd_debug_set_conv_trace(void *p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_ctrace), &p, sizeof(p)) == hipSuccess ? 0 : -1; }
#endif

extern "C" size_t fd_conv2d_packed_weight_bytes(int cout, int cin, int ks) {
    if (cout <= 0 || cin <= 0 || cin % 32 || (ks != 1 && ks != 3)) return 0;
    const size_t cout_pad = ((size_t)cout + 31) / 32 * 32;
    return cout_pad * cin * ks * ks * 2;
}

// w: [cout][cin][ks][ks] float32 (torch Conv2d layout) -> [cout_pad/32][cin/32][tap][ksub][lane][8] bf16
extern "C" int fd_conv2d_pack_weight(const float *w, int cout, int cin, int ks, void *dst) {
    FD_REQUIRE(w && dst, "fd_conv2d_pack_weight: null argument");
    FD_REQUIRE(cin % 32 == 0 && (ks == 1 || ks == 3) && cout > 0, "fd_conv2d_pack_weight: need cin %% 32 == 0 and ks in {1,3}");
    const int cout_pad = (cout + 31) / 32 * 32, nsl = cin / 32, taps = ks * ks;
    uint16_t *d = (uint16_t *)dst;
    auto tobf = [](float v) {
        union { float f; uint32_t u; } c;
        c.f = v;
        uint32_t u = c.u;
        u += 0x7fffu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    };
    for (int nt = 0; nt < cout_pad / 32; ++nt)
        for (int s = 0; s < nsl; ++s)
            for (int tap = 0; tap < taps; ++tap)
                for (int ksub = 0; ksub < 2; ++ksub)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int co = nt * 32 + (lane & 31);
                            const int ci = s * 32 + ksub * 16 + 8 * (lane >> 5) + j;
                            const float v = co < cout ? w[(((int64_t)co * cin + ci) * ks + tap / ks) * ks + tap % ks] : 0.0f;
                            d[((((((int64_t)nt * nsl + s) * taps + tap) * 2 + ksub) * 64 + lane) * 8) + j] = tobf(v);
                        }
    return FD_OK;
}

extern "C" int fd_conv2d_nhwc_bf16(const void *x, int B, int H, int W, int cin, const void *wpacked, const float *bias, int cout, int ks,
                                   int stride, int pad, int relu, void *y, int cout_total, int co_off, int osy, int osx, int ooy, int oox,
                                   fd_stream_t stream) {
    FD_REQUIRE(x && wpacked && y, "fd_conv2d_nhwc_bf16: null argument");
    FD_REQUIRE(cin % 32 == 0 && cin >= 32, "fd_conv2d_nhwc_bf16: cin must be a multiple of 32 (got %d)", cin);
    FD_REQUIRE((ks == 3 && (stride == 1 || stride == 2) && pad == 1) || (ks == 1 && stride == 1 && pad == 0),
               "fd_conv2d_nhwc_bf16: supported: 3x3 stride 1|2 pad 1, 1x1 stride 1 pad 0");
    FD_REQUIRE(B > 0 && H > 0 && W > 0 && cout > 0 && osy >= 1 && osx >= 1, "fd_conv2d_nhwc_bf16: bad shape");
    ConvParams p;
    p.B = B; p.H = H; p.W = W; p.Cin = cin;
    p.Ho = (H + 2 * pad - ks) / stride + 1;
    p.Wo = (W + 2 * pad - ks) / stride + 1;
    p.Cout_real = cout;
    p.Cout_pad = (cout + 31) / 32 * 32;
    p.cout_total = cout_total; p.co_off = co_off; p.pad = pad; p.relu = relu;
    p.osy = osy; p.osx = osx; p.ooy = ooy; p.oox = oox;
    {
        const size_t wb_ = fd_conv2d_packed_weight_bytes(cout, cin, ks);
        FD_REQUIRE(wb_ < (1ull << 31), "fd_conv2d_nhwc_bf16: packed weights of 2 GB and more are not supported");
        p.w_bytes = (unsigned)wb_;
    }
    p.tiles_x = (p.Wo + TW - 1) / TW;
    p.tiles_y = (p.Ho + TH - 1) / TH;
    mixed_tiles(p);
    hipStream_t s = fd::as_stream(stream);
    if (ks == 3 && stride == 1) {
        if (!dispatch_strip<3>(x, wpacked, bias, y, p, s)) dispatch_nt<3, 1>(x, wpacked, bias, y, p, s);
    } else if (ks == 3) {
        dispatch_nt<3, 2>(x, wpacked, bias, y, p, s);
    } else if (!dispatch_strip<1>(x, wpacked, bias, y, p, s)) {
        dispatch_nt<1, 1>(x, wpacked, bias, y, p, s);
    }
    return fd::check_launch("fd_conv2d_nhwc_bf16");
}
