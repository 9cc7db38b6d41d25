// fp32 sparse convolution on the bf16 matrix pipe: split operands ("3 x bf16"), fp32 accumulate.
//
// Replaces spconv 1.0's indice_conv / indice_subm_conv for the wide convolutions of det3d/models/backbones/scn.py:109-141
// in the fp32 configuration, fused with the folded BatchNorm1d, residual add and ReLU (scn.py:67-78).
//
// Why: on gfx950 an fp32-input MFMA runs at the fp32 VECTOR rate, 1/16 of the bf16 MFMA rate, and the native fp32 kernel
// (fd_spconv_v2.hip) sits at that bound on the 64- and 128-channel layers (70 % matrix-pipe busy, flat for three rounds).
// Here every fp32 value is the exact sum of three bf16 pieces,
//     x = xh + xm + xl,   xh = rn_bf16(x), xm = rn_bf16(x - xh), xl = rn_bf16(x - xh - xm)
// (both subtractions are exact; 24 significand bits = 3 x 8 and round-to-nearest leaves no remainder for any finite x whose
// pieces stay in the normal range), and a product keeps the six cross terms of weight >= 2^-16:
//     x w  ~=  xh wh + (xh wm + xm wh) + (xh wl + xm wm + xl wh)        dropped: xm wl + xl wm + xl wl  <= 3 * 2^-26 |x w|,
// each term one v_mfma_f32_32x32x16_bf16 (bf16 products are exact in fp32; fp32 accumulate): 6/16 of the fp32 MFMA time at
// fp32-class accuracy (error table: profiles/round4_split_error_table.txt; tests/test_gpu_split.py gates it against the
// native kernel's error vs float64).
//
// STORAGE.  The three pieces are the storage format of the feature matrices between split layers ("planes", dtype code 2
// of the C ABI): row r = [xh[C] | xm[C] | xl[C]] bf16, 6 C bytes -- a lossless re-encoding of the fp32 row (fd_planes_to_rows
// returns it bit for bit).  A layer splits its OUTPUT once in the epilogue (a handful of VALU instructions per element); the
// consumers gather bf16 fragments that go into the MFMA as they arrive.  The first version of this kernel
// (tools/probes/spconv_split_inreg_experiment.hip) gathered fp32 rows and split them in registers: every element is then
// split once per (output row, tap) pair -- 27 x more often -- and the ~80 VALU instructions per 48 MFMAs made the non-MFMA
// part of the kernel longer than the MFMA part (measured by leaving parts out: no-MFMA 165 us vs ~125 us of MFMA on 128 -> 128,
// 250 us together; profiles/round4_split_v1_*.txt).
// Not representable: |x| within half a bf16 ulp of FLT_MAX (xh rounds to infinity) and non-finite x: both become NaN where the
// fp32 kernel carries inf -- a non-finite feature either way.
//
// Formulation = the bf16 kernel's (fd_spconv_bf16.hip): a WAVE owns 32 * NT consecutive output rows and all output columns,
// accumulators stay in registers over all taps (no atomics, no LDS accumulators); W (three fragment planes, split at pack
// time) is shared through an LDS ring of three stages, a stage = one (tap, 32-channel chunk): 2 MFMA steps x COUT/32 blocks x
// 3 planes of 1 KB (24 KB at 128 columns), one barrier per stage; a stage's weights become visible a whole stage before they
// are multiplied, so fragment reads run one (step, block) item ahead across stage boundaries.  Gathered fragments are single
// buffered per MFMA step: as soon as a step's MFMAs are issued its registers take the same step of the NEXT stage (the other
// step's MFMAs cover the round trip).  Rulebook slices arrive by LDS-DMA a tile ahead.  A missing neighbour is a byte offset
// past the buffer: zeros.  Summation order: taps ascending, chunks ascending, steps ascending, cross terms (wh xl, wh xm,
// wh xh, wm xm, wm xh, wl xh) -- fixed: deterministic, independent of NT.
#include "fd_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxTaps = 27;

using fd::join3x4;
using fd::split3x4;

// Tuning builds only (tools/probes/build_split_exp.sh compiles this file with -DFD_SPLIT_EXP=<mask> into separate libraries; the
// product library contains none of it): parts of the main loop left out at compile time, results wrong, to see what a part
// costs.  1 no gathers, 2 no MFMAs, 8 no stage barrier, 16 no W ring traffic, 32 no W fragment reads.
#ifndef FD_SPLIT_EXP
#define FD_SPLIT_EXP 0
#endif

// NT = 32-row tiles per wave, NW = waves per workgroup.  OUT_PLANES: output (and residual) rows as planes, else float32.
template <int CIN, int COUT, int NT, int NW, bool OUT_PLANES>
__global__ void __launch_bounds__(NW * 64) spconv_p3_ws(const unsigned short *__restrict__ in, const u32x4 *__restrict__ wp,
                                                        const float *__restrict__ bias, const void *__restrict__ residual_, int relu,
                                                        const int *__restrict__ nbr, int64_t nbr_stride, int K, int n_out,
                                                        const int *__restrict__ n_out_dev, void *__restrict__ out_, unsigned in_bytes) {
    constexpr int NCU = CIN / 32;        // 32-channel chunks (stages) per tap; a stage = two K = 16 MFMA steps
    constexpr int NBL = COUT / 32;       // 32-column blocks
    constexpr int NI = 2 * NBL;          // (step, block) items of a stage, step-major
    constexpr int FRS = NI * 3;          // fragments (1 KB each) of a stage: [step u][block][plane h, m, l]
    constexpr int ROWS = 32 * NT;
    constexpr unsigned kRowBytes = 6 * CIN;  // input row: three planes of CIN bf16
    constexpr int NWR = (FRS + NW - 1) / NW;  // fragments of a stage that one wave moves
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4 *s_w = reinterpret_cast<u32x4 *>(smem);                         // [3][FRS][64]: ring of three stages
    constexpr int kSliceInts = (kMaxTaps + 1) * ROWS;
    constexpr int kSliceBufs = NT >= 3 ? 1 : 2;                           // (three and four tiles per wave: LDS holds one slice buffer;
                                                                          //  such launches make one or two passes, each waits for its slice)
    constexpr int kWaveInts = kSliceBufs * kSliceInts + ROWS;             // slice buffer(s) + the 'no neighbour' row
    int *s_nbr = reinterpret_cast<int *>(s_w + 3 * FRS * 64);             // [NW][kWaveInts], wave-private
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: what derives from it stays in SGPRs
    const int lrow = lane & 31, lh = lane >> 5;
    n_out = fd::device_count(n_out, n_out_dev);
    int *s = s_nbr + wave * kWaveInts;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(in), 0, (int)in_bytes, 0x00020000);
    const unsigned lane_off = (unsigned)(lh * 16);
    const int S = K * NCU;  // stages

    // rows: contiguous per workgroup, XCD-contiguous eighths (see fd_spconv_bf16.hip)
    const unsigned lb = fd::xcd_swizzle(blockIdx.x, gridDim.x);
    const int rows_per_wg = (((n_out + (int)gridDim.x - 1) / (int)gridDim.x) + 31) & ~31;
    const int64_t wg_r0 = (int64_t)lb * rows_per_wg;
    if (wg_r0 >= n_out) return;  // (uniform for the workgroup)
    const int wg_r1 = (int)(wg_r0 + rows_per_wg < n_out ? wg_r0 + rows_per_wg : n_out);
    const int n_iter = (rows_per_wg + NW * ROWS - 1) / (NW * ROWS);

    // this wave's share of W[stage st] (stages past the end re-read the last one: their gathered rows are all zero)
    auto wload = [&](int st, u32x4(&dst)[NWR]) {
        st = st < S ? st : S - 1;
#pragma unroll
        for (int i = 0; i < NWR; ++i) {
            const int f = wave + i * NW;
            if (FRS % NW == 0 || f < FRS) dst[i] = wp[((int64_t)st * FRS + f) * 64 + lane];
        }
    };
    auto wstore = [&](int slot, const u32x4(&src)[NWR]) {
#pragma unroll
        for (int i = 0; i < NWR; ++i) {
            const int f = wave + i * NW;
            if (FRS % NW == 0 || f < FRS) s_w[(slot * FRS + f) * 64 + lane] = src[i];
        }
    };
    auto tile_rows = [&](int it, int &row0) {
        const int64_t r = wg_r0 + ((int64_t)it * NW + wave) * ROWS;
        row0 = (int)(r < wg_r1 ? r : wg_r1);
    };
    constexpr int NPRE = (kMaxTaps * ROWS + 63) / 64;
    static_assert(NPRE * 64 <= kSliceInts, "a slice buffer takes whole DMA instructions");
    auto request_slice = [&](int it) {
        int row0;
        tile_rows(it, row0);
        int *dst = s + (it & (kSliceBufs - 1)) * kSliceInts;
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
    for (int r = lane; r < ROWS; r += 64) s[kSliceBufs * kSliceInts + r] = -1;  // the 'no neighbour' row
    if (kSliceBufs == 2) request_slice(0);

    for (int it = 0; it < n_iter; ++it) {
        int row0;
        tile_rows(it, row0);
        if (kSliceBufs == 1) request_slice(it);  // (all reads of the previous pass are complete: LDS operations of a wave execute in order)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (kSliceBufs == 2 && it + 1 < n_iter) request_slice(it + 1);
        const int *sl = s + (it & (kSliceBufs - 1)) * kSliceInts;
        bool valid[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) valid[t] = row0 + 32 * t + lrow < wg_r1;

        // lane (row lrow, half lh) register v of block b = output channel 32 b + 8 (v / 4) + 4 lh + v % 4 of its row
        f32x16 acc[NT][NBL];
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (bias) bv = *reinterpret_cast<const f32x4 *>(bias + 32 * b + 8 * q + 4 * lh);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[t][b][4 * q + i] = bv[i];
                }
            }
        }

        // byte offset of this lane's 16 bytes of plane 0, step 0 of stage st for every tile (stages past the end: the 'no
        // neighbour' row).  -1 * kRowBytes = 2^32 - kRowBytes; adding an in-row offset stays below 2^32: out of range -> zeros.
        auto fetch_off = [&](int st, unsigned(&off)[NT]) {
            const int tap = st / NCU, c = st - tap * NCU;
            const int *p = (tap < K ? sl + tap * ROWS : s + kSliceBufs * kSliceInts) + lrow;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int v = p[32 * t];
                const int e = valid[t] ? v : -1;
                off[t] = (unsigned)e * kRowBytes + lane_off + (unsigned)(c * 64);
            }
        };
        // step u of a stage: one 16-byte fragment per plane = channels 32 c + 16 u + 8 lh .. + 7 of the lane's row
        auto issue = [&](u32x4(&dst)[NT][3], const unsigned(&off)[NT], int u) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) dst[t][pl] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[t] + (unsigned)(pl * 2 * CIN + 32 * u), 0, 0);
            }
        };

        u32x4 x[2][NT][3];  // B fragments of the current stage: [step][tile][plane]
        unsigned off_next[NT];
        u32x4 wr[NWR];
        fetch_off(0, off_next);
        issue(x[0], off_next, 0);
        issue(x[1], off_next, 1);
        fetch_off(1, off_next);
        // W ring of three stages: W[st + 2] is stored at the top of stage st (its slot was last read in stage st - 1) and becomes
        // visible with the barrier that ends stage st -- a whole stage before it is multiplied.
        wload(0, wr);
        wstore(0, wr);
        wload(1, wr);
        wstore(1, wr);
        wload(2, wr);
        __syncthreads();
        bf16x8 wf[2][3];  // fragment triples (h, m, l) of the item being multiplied and the next one
        int slot = 0;     // ring slot of the current stage
        auto wfetch = [&](int sl_, int item, bf16x8(&dst)[3]) {
            const u32x4 *src = s_w + (sl_ * FRS + item * 3) * 64 + lane;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) dst[pl] = __builtin_bit_cast(bf16x8, src[pl * 64]);
        };
        wfetch(0, 0, wf[0]);
        if (FD_SPLIT_EXP & 32) wfetch(0, 1, wf[1]);
        for (int st = 0; st < S; ++st) {
            const int slot_next = slot == 2 ? 0 : slot + 1, slot_store = slot_next == 2 ? 0 : slot_next + 1;
            if (!(FD_SPLIT_EXP & 16)) {
                wstore(slot_store, wr);  // W[st + 2], requested one stage ago
                wload(st + 3, wr);
            }
#pragma unroll
            for (int item = 0; item < NI; ++item) {
                const int u = item / NBL, b = item - u * NBL;
                if (!(FD_SPLIT_EXP & 32)) {
                    if (item + 1 < NI) wfetch(slot, item + 1, wf[(item + 1) & 1]);
                    else wfetch(slot_next, 0, wf[0]);
                }
                const bf16x8(&w)[3] = wf[item & 1];
#define FD_TERM(P, Q)                                                                                          \
    if (!(FD_SPLIT_EXP & 2)) _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t][b] =                       \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[P], __builtin_bit_cast(bf16x8, x[u][t][Q]), acc[t][b], 0, 0, 0);
                FD_TERM(0, 2)
                FD_TERM(0, 1)
                FD_TERM(0, 0)
                FD_TERM(1, 1)
                FD_TERM(1, 0)
                FD_TERM(2, 0)
#undef FD_TERM
                if (b == NBL - 1) {
                    // this step's fragments are consumed: its registers take the same step of the next stage
                    if (!(FD_SPLIT_EXP & 1)) issue(x[u], off_next, u);
                    if (u == 1) fetch_off(st + 2, off_next);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            slot = slot_next;
            if (!(FD_SPLIT_EXP & 8)) __syncthreads();
        }

        // ---- epilogue
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int row = row0 + 32 * t + lrow;
            const int64_t rbase = (int64_t)(row < wg_r1 ? row : 0) * COUT;
#pragma unroll
            for (int b = 0; b < NBL; ++b) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {acc[t][b][4 * q], acc[t][b][4 * q + 1], acc[t][b][4 * q + 2], acc[t][b][4 * q + 3]};
                    const int col = 32 * b + 8 * q + 4 * lh;
                    if constexpr (OUT_PLANES) {
                        const unsigned short *res = reinterpret_cast<const unsigned short *>(residual_);
                        unsigned short *out = reinterpret_cast<unsigned short *>(out_);
                        if (res) {
                            const unsigned short *rp = res + rbase * 3 + col;
                            v += join3x4(*reinterpret_cast<const bf16x4 *>(rp), *reinterpret_cast<const bf16x4 *>(rp + COUT),
                                         *reinterpret_cast<const bf16x4 *>(rp + 2 * COUT));
                        }
                        if (relu) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
                        }
                        bf16x4 h, m, l;
                        split3x4(v, h, m, l);
                        if (row < wg_r1) {
                            unsigned short *op = out + rbase * 3 + col;
                            *reinterpret_cast<bf16x4 *>(op) = h;
                            *reinterpret_cast<bf16x4 *>(op + COUT) = m;
                            *reinterpret_cast<bf16x4 *>(op + 2 * COUT) = l;
                        }
                    } else {
                        const float *res = reinterpret_cast<const float *>(residual_);
                        float *out = reinterpret_cast<float *>(out_);
                        if (res) v += *reinterpret_cast<const f32x4 *>(res + rbase + col);
                        if (relu) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
                        }
                        if (row < wg_r1) *reinterpret_cast<f32x4 *>(out + rbase + col) = v;
                    }
                }
            }
        }
    }
}

struct SplitArgs {
    const void *in;
    const void *wp;
    const float *bias;
    const void *residual;
    int relu;
    const int *nbr;
    int64_t nbr_stride;
    int K, n_out;
    const int *n_out_dev;
    void *out;
    unsigned in_bytes;
    int64_t n_expected;
    hipStream_t stream;
};

template <int COUT, int NT, int NW>
constexpr size_t split_lds_bytes() {
    return (size_t)3 * (COUT / 16) * 3 * 1024 + (size_t)NW * ((NT >= 3 ? 1 : 2) * (kMaxTaps + 1) + 1) * 32 * NT * 4;
}

template <int CIN, int COUT, int NT, int NW, bool OUT_PLANES>
struct SplitKernel {
    static int wgs_per_cu() {
        static std::atomic<int> cached{0};
        int v = cached.load(std::memory_order_relaxed);
        if (v) return v > 0 ? v : 0;
        constexpr size_t lds = split_lds_bytes<COUT, NT, NW>();
        auto kern = spconv_p3_ws<CIN, COUT, NT, NW, OUT_PLANES>;
        int nb = 0;
        static std::atomic<uint64_t> lds_set{0};
        if (lds > 160 * 1024 || (lds > 65536 && !fd::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, lds_set)) ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, NW * 64, lds) != hipSuccess) {
            (void)hipGetLastError();
            nb = 0;
        }
        cached.store(nb > 0 ? nb : -1, std::memory_order_relaxed);
        return nb;
    }
    static int64_t grid_for(int64_t n) {
        // persistent workgroups, one per compute unit (the whole W set streams through every workgroup once per pass: more,
        // smaller workgroups would multiply that traffic), fewer when the level is small
        int64_t grid = fd::device_cu_count();
        const int64_t most = (n + NW * 32 - 1) / (NW * 32);
        if (grid > most) grid = most;
        return grid < 1 ? 1 : grid;
    }
    // relative cost of a launch for n rows: passes x (tiles per wave + the per-pass skeleton), -1 when the variant cannot run
    static int64_t cost(int64_t n) {
        if (wgs_per_cu() <= 0) return -1;
        const int64_t grid = grid_for(n);
        const int64_t rows_per_wg = (((n + grid - 1) / grid) + 31) & ~31ll;
        const int64_t passes = (rows_per_wg + NW * 32 * NT - 1) / (NW * 32 * NT);
        return passes * (NT * 8 + 3);
    }
    static bool launch(const SplitArgs &a) {
        if (wgs_per_cu() <= 0) return false;
        const int64_t grid = grid_for(a.n_expected);
        constexpr size_t lds = split_lds_bytes<COUT, NT, NW>();
        hipLaunchKernelGGL((spconv_p3_ws<CIN, COUT, NT, NW, OUT_PLANES>), dim3((unsigned)grid), dim3(NW * 64), lds, a.stream, (const unsigned short *)a.in,
                           (const u32x4 *)a.wp, a.bias, a.residual, a.relu, a.nbr, a.nbr_stride, a.K, a.n_out, a.n_out_dev, a.out, a.in_bytes);
        return true;
    }
};

// tiles per wave: the variant with the least passes x (tiles + skeleton) for the expected row count (nt > 0: forced)
template <int CIN, int COUT, bool OUT_PLANES>
bool launch_split(const SplitArgs &a, int nt) {
    constexpr int kMaxNT = COUT >= 128 ? 1 : COUT == 64 ? 2 : 4;  // (register budget of two waves per SIMD)
    if (nt <= 0) {
        int64_t best = -1;
        auto consider = [&](int v, int64_t c) {
            if (c >= 0 && (best < 0 || c < best)) { best = c; nt = v; }
        };
        consider(1, SplitKernel<CIN, COUT, 1, 8, OUT_PLANES>::cost(a.n_expected));
        if constexpr (kMaxNT >= 2) consider(2, SplitKernel<CIN, COUT, 2, 8, OUT_PLANES>::cost(a.n_expected));
        if constexpr (kMaxNT >= 3) consider(3, SplitKernel<CIN, COUT, 3, 8, OUT_PLANES>::cost(a.n_expected));
        if constexpr (kMaxNT >= 4) consider(4, SplitKernel<CIN, COUT, 4, 8, OUT_PLANES>::cost(a.n_expected));
        if (nt <= 0) return false;
    }
    if constexpr (kMaxNT >= 4) {
        if (nt >= 4) return SplitKernel<CIN, COUT, 4, 8, OUT_PLANES>::launch(a);
    }
    if constexpr (kMaxNT >= 3) {
        if (nt >= 3) return SplitKernel<CIN, COUT, 3, 8, OUT_PLANES>::launch(a);
    }
    if constexpr (kMaxNT >= 2) {
        if (nt >= 2) return SplitKernel<CIN, COUT, 2, 8, OUT_PLANES>::launch(a);
    }
    return SplitKernel<CIN, COUT, 1, 8, OUT_PLANES>::launch(a);
}

// ---- float32 rows <-> planes (format conversion at the edges of a split region, tests)
__global__ void __launch_bounds__(256) rows_to_planes_kernel(const float *__restrict__ src, unsigned short *__restrict__ dst, int64_t n, int c4,
                                                             const int *__restrict__ n_dev) {
    if (n_dev) { const int64_t m = *n_dev; n = m < n ? (m > 0 ? m : 0) : n; }
    const int64_t total = n * c4;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t r = t / c4;
        const int q = (int)(t - r * c4);
        const f32x4 v = *reinterpret_cast<const f32x4 *>(src + (r * c4 + q) * 4);
        bf16x4 h, m, l;
        split3x4(v, h, m, l);
        unsigned short *op = dst + r * 12 * c4 + 4 * q;
        *reinterpret_cast<bf16x4 *>(op) = h;
        *reinterpret_cast<bf16x4 *>(op + 4 * c4) = m;
        *reinterpret_cast<bf16x4 *>(op + 8 * c4) = l;
    }
}
__global__ void __launch_bounds__(256) planes_to_rows_kernel(const unsigned short *__restrict__ src, float *__restrict__ dst, int64_t n, int c4,
                                                             const int *__restrict__ n_dev) {
    if (n_dev) { const int64_t m = *n_dev; n = m < n ? (m > 0 ? m : 0) : n; }
    const int64_t total = n * c4;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t r = t / c4;
        const int q = (int)(t - r * c4);
        const unsigned short *ip = src + r * 12 * c4 + 4 * q;
        *reinterpret_cast<f32x4 *>(dst + (r * c4 + q) * 4) =
            join3x4(*reinterpret_cast<const bf16x4 *>(ip), *reinterpret_cast<const bf16x4 *>(ip + 4 * c4), *reinterpret_cast<const bf16x4 *>(ip + 8 * c4));
    }
}

}  // namespace

extern "C" int fd_rows_to_planes(const float *src, int64_t n, int c, const int32_t *n_dev, void *dst, fd_stream_t stream) {
    FD_REQUIRE(c > 0 && c % 4 == 0, "fd_rows_to_planes: channels must be a multiple of 4");
    if (n <= 0) return FD_OK;
    FD_REQUIRE(src && dst, "fd_rows_to_planes: null argument");
    int64_t blocks = (n * (c / 4) + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(rows_to_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, fd::as_stream(stream), src, (unsigned short *)dst, n, c / 4, n_dev);
    return fd::check_launch("fd_rows_to_planes");
}

extern "C" int fd_planes_to_rows(const void *src, int64_t n, int c, const int32_t *n_dev, float *dst, fd_stream_t stream) {
    FD_REQUIRE(c > 0 && c % 4 == 0, "fd_planes_to_rows: channels must be a multiple of 4");
    if (n <= 0) return FD_OK;
    FD_REQUIRE(src && dst, "fd_planes_to_rows: null argument");
    int64_t blocks = (n * (c / 4) + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(planes_to_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, fd::as_stream(stream), (const unsigned short *)src, dst, n, c / 4, n_dev);
    return fd::check_launch("fd_planes_to_rows");
}

namespace fd {
// planes in; out_planes 1: planes out (residual planes), 0: float32 out (residual float32).  1 = launched, 0 = shape not covered.
int spconv_p3_dispatch(const void *in, const void *wp, const float *bias, const void *residual, int relu, const int *nbr, int64_t nbr_stride, int K,
                       int64_t n_in_bound, int n_out, const int *n_out_dev, int64_t n_expected, int cin, int cout, int out_planes, void *out,
                       hipStream_t stream) {
    if (n_in_bound * cin * 6 >= (1ll << 31)) return 0;  // the 'missing neighbour' offset must lie beyond the buffer
    SplitArgs a{in, wp, bias, residual, relu, nbr, nbr_stride, K, n_out, n_out_dev, out, (unsigned)(n_in_bound * cin * 6), n_expected, stream};
    const int nt = fd::tuning(fd::kTuneSplitRG);
#define FD_SHAPE(CI, CO) \
    case CI * 1000 + CO: return (out_planes ? launch_split<CI, CO, true>(a, nt) : launch_split<CI, CO, false>(a, nt)) ? 1 : 0;
    switch (cin * 1000 + cout) {
        FD_SHAPE(32, 32)
        FD_SHAPE(32, 64)
        FD_SHAPE(64, 64)
        FD_SHAPE(64, 128)
        FD_SHAPE(128, 128)
        default: return 0;
    }
#undef FD_SHAPE
}

// host side of the operand split (pack time): w -> (h, m, l) bf16 bit patterns, round-to-nearest-even at every level
void split3_host(float w, uint16_t &h, uint16_t &m, uint16_t &l) {
    auto tobf = [](float v) {
        union { float f; uint32_t u; } c;
        c.f = v;
        uint32_t u = c.u;
        if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);  // inf / nan: truncate
        u += 0x7fffu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    };
    auto tof = [](uint16_t b) {
        union { float f; uint32_t u; } c;
        c.u = (uint32_t)b << 16;
        return c.f;
    };
    h = tobf(w);
    const float r1 = w - tof(h);
    m = tobf(r1);
    const float r2 = r1 - tof(m);
    l = tobf(r2);
}
}  // namespace fd
