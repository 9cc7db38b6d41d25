// fp32 sparse convolution on the bf16 matrix pipe: split operands ("3 x bf16"), fp32 accumulate.  dtype code 2 of
// fd_spconv_pack_weight / fd_spconv_apply: features, bias, residual and output are float32 exactly as for dtype 0.
//
// Replaces spconv 1.0's indice_conv / indice_subm_conv for the wide convolutions of det3d/models/backbones/scn.py:119-141
// (32 -> 64 ... 128 -> 128) in the fp32 configuration, fused with the folded BatchNorm1d, residual add and ReLU (scn.py:67-78).
//
// Why: on gfx950 an fp32-input MFMA runs at the fp32 VECTOR rate, 1/16 of the bf16 MFMA rate, and the native fp32 kernel
// (fd_spconv_v2.hip) sits at that bound on the 64- and 128-channel layers (70 % matrix-pipe busy, flat for three rounds).
// Here every fp32 operand is written as the exact sum of three bf16 pieces,
//     x = xh + xm + xl,   xh = rn_bf16(x), xm = rn_bf16(x - xh), xl = rn_bf16(x - xh - xm)   (both subtractions are exact;
//     24 significand bits = 3 x 8, round-to-nearest leaves |x - xh - xm - xl| = 0 for every finite x whose pieces do not
//     overflow -- see below),
// and the product keeps the six cross terms of weight >= 2^-16:
//     x w  ~=  xh wh + (xh wm + xm wh) + (xh wl + xm wm + xl wh)        dropped: xm wl + xl wm + xl wl  <= 3 * 2^-26 |x w|,
// each term one v_mfma_f32_16x16x32_bf16 (bf16 products are exact in fp32; fp32 accumulate).  Six bf16 MFMAs replace
// sixteen-bf16-MFMAs-worth of fp32 MFMA time: 6/16 of the matrix-pipe time at fp32-class accuracy (the error table is
// profiles/round4_split_error_table.txt; tests gate it against the native kernel's error vs float64).
// Weights are split once at pack time (three fragment planes); the gathered rows are split in registers, 11 plain VALU
// instructions per pair of values, which -- unlike next to an fp32 MFMA -- co-execute with the bf16 MFMAs of the same wave.
// Not representable: |x| within half a bf16 ulp of FLT_MAX (xh rounds to infinity) and non-finite x (inf - inf): both give NaN
// where the fp32 kernel gives inf / NaN -- same class of result (a non-finite feature), different payload.
//
// Formulation = the bf16 kernel's (fd_spconv_bf16.hip) on v_mfma_f32_32x32x16_bf16 (the 16x16x32 form of the first version was
// ISSUE-bound: 2.6 other instructions per 16-cycle MFMA, 55 % matrix-pipe busy, profiles/round4_split_v1_sq_counters.txt;
// the 32-cycle form does the same multiply-adds with half the MFMA instructions): a WAVE owns 32 * NT consecutive output rows and all output columns,
// accumulators stay in registers over all taps (no atomics, no LDS accumulators); W is shared through a double-buffered LDS
// ring whose stage is one (tap, 32-channel chunk): 3 planes x COUT/16 fragments (24 KB at 128 columns), one barrier per
// stage; rulebook slices arrive by LDS-DMA a tile ahead; the gather of stage s + 2 is issued during stage s, the rows of
// stage s + 1 are split during the MFMAs of stage s.  A missing neighbour is a byte offset past the buffer: zeros.
// Summation order: taps ascending, chunks ascending, cross terms (wh xl, wh xm, wh xh, wm xm, wm xh, wl xh) -- fixed:
// deterministic, independent of RG / NW.
#include "fd_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxTaps = 27;

struct Planes {
    bf16x8 h, m, l;
};

// eight floats (two 16-byte pieces of a row) -> three bf16x8 planes, round-to-nearest-even at every level
__device__ __forceinline__ Planes split3(const u32x4 &p0, const u32x4 &p1) {
    const f32x4 a = __builtin_bit_cast(f32x4, p0), b = __builtin_bit_cast(f32x4, p1);
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    Planes r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 v = {x[2 * i], x[2 * i + 1]};
        const bf16x2 h = __builtin_convertvector(v, bf16x2);
        const f32x2 r1 = v - __builtin_convertvector(h, f32x2);
        const bf16x2 m = __builtin_convertvector(r1, bf16x2);
        const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
        const bf16x2 l = __builtin_convertvector(r2, bf16x2);
        r.h[2 * i] = h[0]; r.h[2 * i + 1] = h[1];
        r.m[2 * i] = m[0]; r.m[2 * i + 1] = m[1];
        r.l[2 * i] = l[0]; r.l[2 * i + 1] = l[1];
    }
    return r;
}

// NT = 32-row tiles per wave, NW = waves per workgroup, PD = (step, block) items the weight-fragment reads run ahead.
template <int CIN, int COUT, int NT, int NW, int PD>
__global__ void __launch_bounds__(NW * 64) spconv_f32s_ws(const float *__restrict__ in, const u32x4 *__restrict__ wp,
                                                          const float *__restrict__ bias, const float *__restrict__ residual, int relu,
                                                          const int *__restrict__ nbr, int64_t nbr_stride, int K, int n_out,
                                                          const int *__restrict__ n_out_dev, float *__restrict__ out, unsigned in_bytes) {
    constexpr int NCU = CIN / 32;        // 32-channel chunks (stages) per tap; a stage = two K = 16 MFMA steps
    constexpr int NBL = COUT / 32;       // 32-column blocks
    constexpr int FRS = 2 * NBL * 3;     // fragments (1 KB each) of a stage: [step u][block][plane h, m, l]
    constexpr int ROWS = 32 * NT;
    constexpr int kRowShift = CIN == 32 ? 7 : CIN == 64 ? 8 : 9;  // log2(bytes of an input row)
    static_assert(CIN * 4 == (1 << kRowShift), "CIN must be 32, 64 or 128");
    constexpr int NWR = (FRS + NW - 1) / NW;  // fragments of a stage that one wave moves
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4 *s_w = reinterpret_cast<u32x4 *>(smem);                         // [3][FRS][64]: ring of three stages
    constexpr int kSliceInts = (kMaxTaps + 1) * ROWS;
    constexpr int kWaveInts = 2 * kSliceInts + ROWS;                      // two slice buffers + the 'no neighbour' row
    int *s_nbr = reinterpret_cast<int *>(s_w + 3 * FRS * 64);             // [NW][kWaveInts], wave-private
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lh = lane >> 5;
    n_out = fd::device_count(n_out, n_out_dev);
    int *s = s_nbr + wave * kWaveInts;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (int)in_bytes, 0x00020000);
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
    if (lane < ROWS) s[2 * kSliceInts + lane] = -1;  // the 'no neighbour' row
    if (ROWS > 64 && lane + 64 < ROWS) s[2 * kSliceInts + lane + 64] = -1;
    request_slice(0);

    for (int it = 0; it < n_iter; ++it) {
        int row0;
        tile_rows(it, row0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (it + 1 < n_iter) request_slice(it + 1);
        const int *sl = s + (it & 1) * kSliceInts;
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

        // byte offsets of this lane's piece of stage st for every tile (stages past the end: the 'no neighbour' row)
        auto fetch_off = [&](int st, unsigned(&off)[NT]) {
            const int tap = st / NCU, c = st - tap * NCU;
            const int *p = (tap < K ? sl + tap * ROWS : s + 2 * kSliceInts) + lrow;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int v = p[32 * t];
                const int e = valid[t] ? v : -1;
                off[t] = ((unsigned)e << kRowShift) + lane_off + (unsigned)(c * 128);  // -1: just below 2^32, out of range -> zeros
            }
        };
        // four 16-byte pieces per lane and stage: piece j = bytes [32 j + 16 lh, + 16) of the row's 128-byte chunk; pieces
        // 2 u, 2 u + 1 are the eight K values of MFMA step u (fd_spconv_pack_weight orders the weights the same way)
        auto issue = [&](u32x4(&dst)[NT][4], const unsigned(&off)[NT]) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[t][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[t] + 32 * j, 0, 0);
            }
        };

        u32x4 raw[2][NT][4];  // gathered fp32 pieces of stage s + 1 / s + 2
        Planes x[2][NT][2];   // split rows of stage s / s + 1, per MFMA step
        unsigned off_next[NT];
        u32x4 wr[NWR];
        fetch_off(0, off_next);
        issue(raw[0], off_next);
        fetch_off(1, off_next);
        issue(raw[1], off_next);
        fetch_off(2, off_next);
        // W ring of three stages: W[st + 2] is stored at the top of stage st (its slot was last read in stage st - 1) and becomes
        // visible with the barrier that ends stage st -- a whole stage before it is multiplied, so the fragment reads of a stage's
        // first items can be issued during the previous stage and no stage starts with an exposed LDS round trip.
        wload(0, wr);
        wstore(0, wr);
        wload(1, wr);
        wstore(1, wr);
        wload(2, wr);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int u = 0; u < 2; ++u) x[0][t][u] = split3(raw[0][t][2 * u], raw[0][t][2 * u + 1]);
        }
        __syncthreads();
        constexpr int NI = 2 * NBL;                          // items of a stage: (step u, block b), u-major
        static_assert(PD >= 1 && PD <= NI && (2 * NI) % (PD + 1) == 0, "fragment ring: static register slots per unrolled stage pair");
        constexpr int kPieces = 2 * NT;                      // (tile, step) pieces to split per stage
        constexpr int kShare = (kPieces + NI - 1) / NI;
        bf16x8 wf[PD + 1][3];                                // fragment triples (h, m, l) of the items in flight
        int slot = 0;                                        // ring slot of the current stage
        auto wfetch = [&](int sl, int item, bf16x8(&dst)[3]) {
            const u32x4 *src = s_w + (sl * FRS + item * 3) * 64 + lane;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) dst[pl] = __builtin_bit_cast(bf16x8, src[pl * 64]);
        };
#pragma unroll
        for (int i = 0; i < PD; ++i) wfetch(0, i, wf[i]);
        for (int s0 = 0; s0 < S; s0 += 2) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int st = s0 + d;
                const int slot_next = slot == 2 ? 0 : slot + 1, slot_store = slot_next == 2 ? 0 : slot_next + 1;
                wstore(slot_store, wr);  // W[st + 2], requested one stage ago
                wload(st + 3, wr);
                // raw[d] held stage st (split during the previous stage): it takes the gather of stage st + 2
                issue(raw[d], off_next);
                fetch_off(st + 3, off_next);
                // The rows of stage st + 1 are split under this stage's MFMAs, one share per (step, block) item; item J's weight
                // fragments were requested PD items earlier.  Nothing else crosses an item boundary (sched_barrier: left to
                // itself hipcc hoists all of a stage's fragment reads, 96 registers at 128 columns, and spills).
#pragma unroll
                for (int item = 0; item < NI; ++item) {
                    const int J = d * NI + item;  // (static: register slots of the fragment ring)
                    const int u = item / NBL, b = item - u * NBL;
                    if (item + PD < NI) wfetch(slot, item + PD, wf[(J + PD) % (PD + 1)]);
                    else wfetch(slot_next, item + PD - NI, wf[(J + PD) % (PD + 1)]);
#pragma unroll
                    for (int pc = item * kShare; pc < (item + 1) * kShare && pc < kPieces; ++pc) {
                        const int t = pc >> 1, uu = pc & 1;
                        x[(d + 1) & 1][t][uu] = split3(raw[(d + 1) & 1][t][2 * uu], raw[(d + 1) & 1][t][2 * uu + 1]);
                    }
                    const bf16x8(&w)[3] = wf[J % (PD + 1)];
#define FD_TERM(P, X) \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[P], x[d][t][u].X, acc[t][b], 0, 0, 0);
                    FD_TERM(0, l)
                    FD_TERM(0, m)
                    FD_TERM(0, h)
                    FD_TERM(1, m)
                    FD_TERM(1, h)
                    FD_TERM(2, h)
#undef FD_TERM
                    __builtin_amdgcn_sched_barrier(0);
                }
                slot = slot_next;
                __syncthreads();
            }
        }

        // ---- epilogue (residual pieces are loaded where they are added: a prefetched block next to the accumulators costs
        //      30-60 registers at 128 columns)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int row = row0 + 32 * t + lrow;
#pragma unroll
            for (int b = 0; b < NBL; ++b) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {acc[t][b][4 * q], acc[t][b][4 * q + 1], acc[t][b][4 * q + 2], acc[t][b][4 * q + 3]};
                    const int64_t o = (int64_t)(row < wg_r1 ? row : 0) * COUT + 32 * b + 8 * q + 4 * lh;
                    if (residual) v += *reinterpret_cast<const f32x4 *>(residual + o);
                    if (relu) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
                    }
                    if (row < wg_r1) *reinterpret_cast<f32x4 *>(out + o) = v;
                }
            }
        }
    }
}

struct SplitArgs {
    const float *in;
    const void *wp;
    const float *bias, *residual;
    int relu;
    const int *nbr;
    int64_t nbr_stride;
    int K, n_out;
    const int *n_out_dev;
    float *out;
    unsigned in_bytes;
    int64_t n_expected;
    hipStream_t stream;
};

template <int COUT, int NT, int NW>
constexpr size_t split_lds_bytes() {
    return (size_t)3 * (COUT / 16) * 3 * 1024 + (size_t)NW * (2 * (kMaxTaps + 1) + 1) * 32 * NT * 4;
}

template <int CIN, int COUT, int NT, int NW, int PD>
struct SplitKernel {
    static int wgs_per_cu() {
        static std::atomic<int> cached{0};
        int v = cached.load(std::memory_order_relaxed);
        if (v) return v > 0 ? v : 0;
        constexpr size_t lds = split_lds_bytes<COUT, NT, NW>();
        auto kern = spconv_f32s_ws<CIN, COUT, NT, NW, PD>;
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
    // relative cost of a launch: passes x row groups per pass (every pass walks all stages whatever its fill)
    static int64_t cost(int64_t n) {
        const int wpc = wgs_per_cu();
        if (wpc <= 0) return -1;
        const int64_t grid = grid_for(n, wpc);
        const int64_t rows_per_wg = (((n + grid - 1) / grid) + 31) & ~31ll;
        const int64_t passes = (rows_per_wg + NW * 32 * NT - 1) / (NW * 32 * NT);
        return passes * NT * 16 + passes;  // (+ the per-pass skeleton: prologue, epilogue, barriers)
    }
    static int64_t grid_for(int64_t n, int wpc) {
        // persistent workgroups: all the device holds at once, as long as each gets at least one 32-row tile per wave
        int64_t grid = (int64_t)wpc * fd::device_cu_count();
        const int64_t most = (n + NW * 32 - 1) / (NW * 32);
        if (grid > most) grid = most;
        return grid < 1 ? 1 : grid;
    }
    static bool launch(const SplitArgs &a) {
        const int wpc = wgs_per_cu();
        if (wpc <= 0) return false;
        const int64_t grid = grid_for(a.n_expected, wpc);
        constexpr size_t lds = split_lds_bytes<COUT, NT, NW>();
        hipLaunchKernelGGL((spconv_f32s_ws<CIN, COUT, NT, NW, PD>), dim3((unsigned)grid), dim3(NW * 64), lds, a.stream, a.in, (const u32x4 *)a.wp, a.bias, a.residual,
                           a.relu, a.nbr, a.nbr_stride, a.K, a.n_out, a.n_out_dev, a.out, a.in_bytes);
        return true;
    }
};

// tiles per wave / waves per workgroup: nt = 1 -> 8 waves x 32 rows (two waves per SIMD); nt = 2 -> 4 waves x 64 rows (one wave per SIMD)
template <int CIN, int COUT>
bool launch_split(const SplitArgs &a, int nt) {
    constexpr int kDeep = COUT >= 64 ? 3 : 1;
    if (nt >= 3) return SplitKernel<CIN, COUT, 2, 4, 1>::launch(a);
    if (nt >= 2) return SplitKernel<CIN, COUT, 2, 4, kDeep>::launch(a);
    return SplitKernel<CIN, COUT, 1, 8, 1>::launch(a);
}

}  // namespace

namespace fd {
// returns 1 when launched, 0 when the shape is not covered
int spconv_f32s_dispatch(const float *in, const void *wp, const float *bias, const float *residual, int relu, const int *nbr, int64_t nbr_stride, int K,
                         int64_t n_in_bound, int n_out, const int *n_out_dev, int64_t n_expected, int cin, int cout, float *out, hipStream_t stream) {
    if (n_in_bound * cin * 4 >= (1ll << 31)) return 0;  // the 'missing neighbour' offset must lie beyond the buffer
    SplitArgs a{in, wp, bias, residual, relu, nbr, nbr_stride, K, n_out, n_out_dev, out, (unsigned)(n_in_bound * cin * 4), n_expected, stream};
    const int rg = fd::tuning(fd::kTuneSplitRG);
    switch (cin * 1000 + cout) {
        case 32032: return launch_split<32, 32>(a, rg) ? 1 : 0;
        case 32064: return launch_split<32, 64>(a, rg) ? 1 : 0;
        case 64064: return launch_split<64, 64>(a, rg) ? 1 : 0;
        case 64128: return launch_split<64, 128>(a, rg) ? 1 : 0;
        case 128128: return launch_split<128, 128>(a, rg) ? 1 : 0;
        default: return 0;
    }
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
