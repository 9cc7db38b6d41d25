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

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TH = 8, TW = 16;  // output pixels per workgroup (M = 128)

// tuning builds (tools/probes/build_trace.sh): thread 0 of every workgroup accumulates cycles per phase:
// [0] prologue (first patch in LDS), [1] step loops, [2] slice hand-over (store + next loads), [3] barrier, [4] epilogue, [5] whole
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
};

// WMT x WNT MFMA tiles (32 x 32) per wave, waves arranged WAVES_M x WAVES_N (product 4); M tile = 128 pixels
template <int KS, int S, int WMT, int WNT, int WAVES_M, int WAVES_N>
__global__ void __launch_bounds__(256) conv2d_nhwc_bf16(const unsigned short *__restrict__ x, const bf16x8 *__restrict__ wp,
                                                        const float *__restrict__ bias, unsigned short *__restrict__ y, ConvParams p) {
    static_assert(WAVES_M * WAVES_N == 4 && WMT * WAVES_M * 32 == TH * TW, "tile shape");
    constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS, PP = PH * PW;
    constexpr int NCHUNK16 = PP * 4;                    // 16-byte chunks per 32-channel patch
    constexpr int NLOAD = (NCHUNK16 + 255) / 256;
    constexpr int NT = WNT * WAVES_N * 32;              // output channels per workgroup
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 x PP x 64 bytes

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef FD_V2_TRACE
    unsigned long long cacc[6] = {0, 0, 0, 0, 0, 0};
#endif
    FD_CT(c_start);
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    int t = blockIdx.x;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y; const int b = t / p.tiles_y;
    const int n0 = blockIdx.y * NT + wn * WNT * 32;     // first output channel of this wave
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - p.pad, ix0 = ox0 * S - p.pad;
    const int nslices = p.Cin / 32;

    // per-thread patch chunk assignment: chunk id -> (pixel, q)
    uint4 stage[NLOAD];
    auto load_slice = [&](int s) {
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int id = tid + i * 256;
            stage[i] = make_uint4(0u, 0u, 0u, 0u);
            if (id < NCHUNK16) {
                const int pix = id >> 2, q = id & 3;
                const int iy = iy0 + pix / PW, ix = ix0 + pix % PW;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                    stage[i] = *reinterpret_cast<const uint4 *>(x + (((int64_t)b * p.H + iy) * p.W + ix) * p.Cin + s * 32 + q * 8);
            }
        }
    };
    auto store_slice = [&](int buf) {
        unsigned char *dst = smem + buf * (PP * 64);
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int id = tid + i * 256;
            if (id < NCHUNK16) {
                const int pix = id >> 2, q = id & 3;
                *reinterpret_cast<uint4 *>(dst + pix * 64 + ((q ^ ((pix >> 2) & 3)) << 4)) = stage[i];
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
        const int m = (wm * WMT + i) * 32 + lm;  // pixel index inside the 8 x 16 tile, row-major
        prow[i] = (m / TW) * S;
        pcol[i] = (m % TW) * S;
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
    constexpr int RING = (ITERS % 6 == 0) ? 6 : 2;
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
    // per-lane LDS offsets of the A fragments of every step (tap, k-half), relative to the slice buffer
    unsigned aoff[ITERS][WMT];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int tap = it >> 1, ks = it & 1;
        const int ky = tap / KS, kx = tap % KS;
#pragma unroll
        for (int i = 0; i < WMT; ++i) {
            const int pix = (prow[i] + ky) * PW + pcol[i] + kx;
            const int q = ks * 2 + lk;
            aoff[it][i] = (unsigned)(pix * 64 + ((q ^ ((pix >> 2) & 3)) << 4));
        }
    }
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
        const unsigned sbase = (unsigned)((s & 1) * (PP * 64));
        auto read_a = [&](int it, bf16x8(&a)[WMT]) {
#pragma unroll
            for (int i = 0; i < WMT; ++i) a[i] = *reinterpret_cast<const bf16x8 *>(smem + sbase + aoff[it][i]);
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
                for (int i = 0; i < WMT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[it % 3][i], bw[slot][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (j == 0 && it + 2 < ITERS) read_a(it + 2, a[(it + 2) % 3]);
                bw[slot][j] = wfrag(j, nsel);  // the slot just consumed takes the fragment of RING steps ahead
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
    // epilogue.  C/D layout of 32x32: col (channel) = lane & 31, row (pixel) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
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
        FD_CT(c_mid);
        constexpr int C8 = NT / 8;
        for (int id = tid; id < TH * TW * C8; id += 256) {
            const int m = id / C8, c8 = id - m * C8;
            const int oy = oy0 + m / TW, ox = ox0 + m % TW;
            const int co = blockIdx.y * NT + c8 * 8;
            if (oy < p.Ho && ox < p.Wo && co < p.Cout_real) {
                const int64_t yy = (int64_t)oy * p.osy + p.ooy, xx = (int64_t)ox * p.osx + p.oox;
                *reinterpret_cast<uint4 *>(y + (((int64_t)b * Hy + yy) * Wy + xx) * p.cout_total + p.co_off + co) =
                    *reinterpret_cast<const uint4 *>(s_out + m * NT + c8 * 8);
            }
        }
#ifdef FD_V2_TRACE
        if (tid == 0 && g_ctrace) {
            const unsigned long long c_end = __builtin_readcyclecounter();
            unsigned long long *o = g_ctrace + (size_t)(blockIdx.x + blockIdx.y * gridDim.x) * 8;
            o[0] = cacc[0]; o[1] = cacc[1]; o[2] = cacc[2]; o[3] = cacc[3]; o[4] = c_end - c_epi; o[5] = c_end - c_start; o[6] = c_mid - c_epi; o[7] = c_end;
        }
#endif
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
                const int oy = oy0 + m / TW, ox = ox0 + m % TW;
                if (oy < p.Ho && ox < p.Wo && co < p.Cout_real) {
                    float v = acc[i][j][r] + bv;
                    if (p.relu) v = fmaxf(v, 0.0f);
                    const int64_t yy = (int64_t)oy * p.osy + p.ooy, xx = (int64_t)ox * p.osx + p.oox;
                    y[(((int64_t)b * Hy + yy) * Wy + xx) * p.cout_total + p.co_off + co] = f2bf(v);
                }
            }
        }
    }
}

template <int KS, int S, int WMT, int WNT, int WAVES_M, int WAVES_N>
void launch_conv(const void *x, const void *wp, const float *bias, void *y, const ConvParams &p, hipStream_t stream) {
    constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS;
    constexpr int NT = WNT * WAVES_N * 32;
    size_t lds = 2 * (size_t)PH * PW * 64;
    if (lds < (size_t)TH * TW * NT * 2) lds = (size_t)TH * TW * NT * 2;  // the epilogue transposes the tile through LDS
    auto kern = conv2d_nhwc_bf16<KS, S, WMT, WNT, WAVES_M, WAVES_N>;
    static std::atomic<uint64_t> lds_set{0};
    if (lds > 65536) (void)fd::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, lds_set);
    dim3 grid((unsigned)(p.tiles_x * p.tiles_y * p.B), (unsigned)(p.Cout_pad / NT));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, (const unsigned short *)x, (const bf16x8 *)wp, bias, (unsigned short *)y, p);
}

template <int KS, int S>
int dispatch_nt(const void *x, const void *wp, const float *bias, void *y, const ConvParams &p, hipStream_t stream) {
    const int force = fd::tuning(fd::kTuneConvNT);  // tuning override
    if (p.Cout_pad % 128 == 0 && force != 64 && force != 32) launch_conv<KS, S, 2, 2, 2, 2>(x, wp, bias, y, p, stream);
    else if (p.Cout_pad % 64 == 0 && force != 32) launch_conv<KS, S, 1, 2, 4, 1>(x, wp, bias, y, p, stream);
    else launch_conv<KS, S, 1, 1, 4, 1>(x, wp, bias, y, p, stream);
    return 1;
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
    hipStream_t s = fd::as_stream(stream);
    if (ks == 3 && stride == 1) dispatch_nt<3, 1>(x, wpacked, bias, y, p, s);
    else if (ks == 3) dispatch_nt<3, 2>(x, wpacked, bias, y, p, s);
    else dispatch_nt<1, 1>(x, wpacked, bias, y, p, s);
    return fd::check_launch("fd_conv2d_nhwc_bf16");
}
