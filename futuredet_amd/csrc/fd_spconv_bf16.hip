// bf16 sparse convolution, round-3 formulation: register accumulators, pipelined bounds-checked gathers, weights shared
// through LDS ("ws").
//
// Replaces spconv 1.0's indice_conv / indice_subm_conv for the 21 convolutions of det3d/models/backbones/scn.py:99-141
// in the bf16 configurations (BASELINE configs[2..4]), fused with the folded BatchNorm1d, residual add and ReLU (scn.py:67-78).
//
// What the counters said about the two older bf16 kernels of fd_spconv.hip (profiles/round3_bf16_before_*): bf16 MFMA is
// 16x the fp32 rate, so no layer is matrix-bound (7 % of the bf16 peak) and none is HBM-bound (the spatially sorted rows keep
// the gather in L1 / L2); waves spent 86 % of their cycles waiting on the vector-memory path.  A first rewrite that only
// pipelined the gathers (every wave streaming W[k] from L1) did not help: the per-CU L1 is the contended resource, and at
// least half of what went through it was weights -- every wave re-read the whole weight set (54 KB ... 864 KB, more than the
// 32 KB L1) for its 16-64 rows.  So:
//
//   * weights never go through L1 more than once per workgroup.  RESIDENT kernels (Cin * Cout <= 32 * 64: the whole [K] set
//     is <= 108 KB) stage all taps in LDS once; workgroups are persistent (one per CU, 16 waves) and their waves walk over
//     tiles on their own -- no barrier after the staging.  RING kernels (64 -> 64, 64 -> 128, 128 -> 128: 8 ... 32 KB per tap)
//     keep a double-buffered W[tap] in LDS: during step s every wave stores its share of W[s + 1] (requested one step earlier)
//     and requests W[s + 2]; one barrier per tap, never waiting on memory.  MFMA weight operands are 16-byte LDS reads in
//     fragment order (conflict-free), shared by the RG row groups of a wave.
//   * a WAVE owns 16 * RG consecutive output rows and ALL output columns; accumulators stay in registers over all taps: no
//     atomics, no LDS traffic for features or accumulators.  Its rulebook slice ([K][16 RG] int32) goes through a wave-private
//     LDS region (coalesced loads -> fragment-order reads).
//   * the gather of tap t + DEPTH - 1 is issued before the MFMAs of tap t (ring of DEPTH register slots).  A missing neighbour
//     (-1) becomes a byte offset just below 2^32: the buffer bounds check returns zeros -- no exec mask, no branch.
//   * the product is issued TRANSPOSED (A operand = weight fragment, B operand = gathered rows): a lane ends up with four
//     consecutive output channels of one row, so bias / residual / output move in 8-byte pieces.
//   * CIN = 16 (first stage and its down-sampling conv): two taps share one K = 32 MFMA -- lane quads 0,1 gather the 16
//     channels of tap 2u, quads 2,3 those of tap 2u + 1; fd_spconv_pack_weight stacks the weights of a tap pair along K.
//
// v_mfma_f32_16x16x32_bf16, fp32 accumulate; summation order: taps ascending, 32-channel chunks ascending -- fixed, so the
// result is deterministic and does not depend on RG / DEPTH / waves per workgroup / RESIDENT vs RING (tested bit for bit).
#include "fd_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxTaps = 27;

template <int CIN, int COUT, int RG, int DEPTH, int NW, bool RING>
__global__ void __launch_bounds__(NW * 64) spconv_bf16_ws(const unsigned short *__restrict__ in, const u32x4 *__restrict__ wp,
                                                          const float *__restrict__ bias, const unsigned short *__restrict__ residual, int relu,
                                                          const int *__restrict__ nbr, int64_t nbr_stride, int K, int n_out,
                                                          const int *__restrict__ n_out_dev, unsigned short *__restrict__ out, unsigned in_bytes, int exp_mode) {
    constexpr bool PAIR = CIN == 16;          // two taps per K = 32 MFMA
    constexpr int NCU = PAIR ? 1 : CIN / 32;  // 32-channel chunks (MFMAs along K) per step
    constexpr int NB = COUT / 16;
    constexpr int FR = NCU * NB;              // weight fragments (1 KB each) of a step
    constexpr int ROWS = 16 * RG;             // rows of a wave's tile
    constexpr int kRowShift = CIN == 16 ? 5 : CIN == 32 ? 6 : CIN == 64 ? 7 : 8;  // log2(bytes of an input row)
    static_assert(CIN * 2 == (1 << kRowShift), "CIN must be 16, 32, 64 or 128");
    static_assert(DEPTH % 2 == 0 && DEPTH >= 2, "ring slots alternate with a static parity");
    constexpr int NWR = (FR + NW - 1) / NW;   // RING: fragments of a step that one wave moves
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int T = PAIR ? (K + 1) >> 1 : K;    // steps
    u32x4 *s_w = reinterpret_cast<u32x4 *>(smem);                                // RESIDENT: [T][FR][64]; RING: [2][FR][64]
    constexpr int kSliceInts = (kMaxTaps + 1) * ROWS;                             // one slice buffer (whole 64-entry DMA instructions)
    constexpr int kWaveInts = 2 * kSliceInts + ROWS;                              // two slice buffers + the 'no neighbour' row
    int *s_nbr = reinterpret_cast<int *>(s_w + (RING ? 2 : T) * FR * 64);        // [NW][kWaveInts], wave-private
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: what derives from it stays in SGPRs
    const int lrow = lane & 15, lq = lane >> 4;
    n_out = fd::device_count(n_out, n_out_dev);  // capacity launch (fd_common.h)
    int *s = s_nbr + wave * kWaveInts;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(in), 0, (int)in_bytes, 0x00020000);
    const unsigned lane_off = PAIR ? (unsigned)((lq & 1) * 16) : (unsigned)(lq * 16);

    // ---- tile loop bounds
    int tile_first, tile_step, n_iter;  // RESIDENT: tiles of this wave; RING: passes of this workgroup
    int wg_r1 = n_out;                   // RING: end of this workgroup's row range
    int64_t wg_r0 = 0;
    // Workgroup b runs on XCD b % 8 (observed dispatch order; only speed depends on it).  Logical block ids are dealt so that
    // each XCD owns one contiguous eighth of the rows: every row costs the same here (no zero skipping), so contiguous
    // shares are balanced, and an XCD's 4 MB L2 then holds the feature rows its CUs gather instead of the whole level.
    const unsigned lb = exp_mode ? blockIdx.x : fd::xcd_swizzle(blockIdx.x, gridDim.x);
    if constexpr (RING) {
        const int rows_per_wg = (((n_out + (int)gridDim.x - 1) / (int)gridDim.x) + 15) & ~15;
        wg_r0 = (int64_t)lb * rows_per_wg;
        if (wg_r0 >= n_out) return;  // (uniform for the workgroup)
        wg_r1 = (int)(wg_r0 + rows_per_wg < n_out ? wg_r0 + rows_per_wg : n_out);
        n_iter = (rows_per_wg + NW * ROWS - 1) / (NW * ROWS);
        tile_first = tile_step = 0;
    } else {
        // all taps' weights -> LDS, once per (persistent) workgroup
        for (int i = tid; i < T * FR * 64; i += NW * 64) s_w[i] = wp[i];
        __syncthreads();
        const int n_tiles = (n_out + ROWS - 1) / ROWS;
        if (exp_mode) {
            tile_first = blockIdx.x * NW + wave;
            tile_step = gridDim.x * NW;
            n_iter = tile_first < n_tiles ? (n_tiles - tile_first + tile_step - 1) / tile_step : 0;
        } else {  // a contiguous chunk of tiles per workgroup, walked NW tiles at a time
            const int tpb = (n_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
            const int t_lo = (int)lb * tpb, t_hi = t_lo + tpb < n_tiles ? t_lo + tpb : n_tiles;
            tile_first = t_lo + wave;
            tile_step = NW;
            n_iter = tile_first < t_hi ? (t_hi - tile_first + NW - 1) / NW : 0;
        }
    }

    auto issue = [&](u32x4(&dst)[RG][NCU], const int(&e)[RG]) {
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            const unsigned voff = ((unsigned)e[g] << kRowShift) + lane_off;  // -1 -> just below 2^32: out of range, reads zeros
#pragma unroll
            for (int c = 0; c < NCU; ++c) dst[g][c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + c * 64, 0, 0);
        }
    };
    // RING: this wave's share of W[step t] (steps past the end re-read the last one: their gathered rows are all zero)
    auto wload = [&](int t, u32x4(&dst)[NWR]) {
        t = t < T ? t : T - 1;
#pragma unroll
        for (int i = 0; i < NWR; ++i) {
            const int f = wave + i * NW;
            if (FR % NW == 0 || f < FR) dst[i] = wp[((int64_t)t * FR + f) * 64 + lane];
        }
    };
    auto wstore = [&](int slot, const u32x4(&src)[NWR]) {
#pragma unroll
        for (int i = 0; i < NWR; ++i) {
            const int f = wave + i * NW;
            if (FR % NW == 0 || f < FR) s_w[(slot * FR + f) * 64 + lane] = src[i];
        }
    };

    // Rulebook slice of a tile ([K][ROWS] int32) -> the wave's LDS region by LDS-DMA (global_load_lds_dword: lane l of
    // instruction i lands at slot 64 i + l, which is exactly the coalesced [tap][row] order), requested a whole tile ahead into
    // the other of two buffers: no registers, no ds_write, and no tile starts with a chain of dependent memory round trips.
    auto tile_rows = [&](int it, int &row0, int &row_end) {
        if constexpr (RING) {
            const int64_t r = wg_r0 + ((int64_t)it * NW + wave) * ROWS;
            row0 = (int)(r < wg_r1 ? r : wg_r1);
            row_end = wg_r1;
        } else {
            row0 = (tile_first + it * tile_step) * ROWS;
            row_end = n_out;
        }
    };
    constexpr int NPRE = (kMaxTaps * ROWS + 63) / 64;
    static_assert(NPRE * 64 <= kSliceInts, "a slice buffer takes whole DMA instructions");
    auto request_slice = [&](int it) {
        int row0, row_end;
        tile_rows(it, row0, row_end);
        int *dst = s + (it & 1) * kSliceInts;
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            if (i * 64 < K * ROWS) {  // (uniform)
                const int t = lane + i * 64;
                int k = t / ROWS;
                const int r = t - k * ROWS;
                k = k < K ? k : K - 1;
                int o = row0 + r;
                o = o < n_out ? o : n_out - 1;  // rows >= n_out of the table are never read (n_out >= 1 here); masked on use
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) int *)(nbr + (int64_t)k * nbr_stride + o),
                                                 (__attribute__((address_space(3))) int *)(dst + i * 64), 4, 0, 0);
            }
        }
    };
    if (lane < ROWS) s[2 * kSliceInts + lane] = -1;  // the 'no neighbour' row
    if (n_iter > 0) request_slice(0);

    for (int it = 0; it < n_iter; ++it) {
        int row0, row_end;
        tile_rows(it, row0, row_end);
        // this tile's slice has landed (nothing else of this wave is in flight here); the next tile's starts travelling
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (it + 1 < n_iter) request_slice(it + 1);
        const int *sl = s + (it & 1) * kSliceInts;
        bool valid[RG];
#pragma unroll
        for (int g = 0; g < RG; ++g) valid[g] = row0 + 16 * g + lrow < row_end;

        // bias enters through the accumulators' initial value
        f32x4 acc[RG][NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (bias) bv = *reinterpret_cast<const f32x4 *>(bias + 16 * nb + 4 * lq);
#pragma unroll
            for (int g = 0; g < RG; ++g) acc[g][nb] = bv;
        }

        // rulebook entries of step t for this lane's row of every group (steps past the end read the 'no neighbour' row)
        auto fetch_idx = [&](int t, int(&e)[RG]) {
            const int tap = PAIR ? 2 * t + (lq >> 1) : t;
            const int *p = (tap < K ? sl + tap * ROWS : s + 2 * kSliceInts) + lrow;
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                const int v = p[16 * g];
                e[g] = valid[g] ? v : -1;
            }
        };
        u32x4 a_r[DEPTH][RG][NCU];
        int e_next[RG];
        if constexpr (!RING) {
            // ---- RESIDENT: steps without a single pair in this wave's rows are skipped (no gather instruction, no MFMA): an
            // out-of-range lane costs the gather path as much as a loaded one (tools/probes/gather_probe.hip), and 65 % of the
            // (16 rows, tap) items of the first stage are empty, most of them in the down-sampling convolutions.  The step
            // mask is the OR over the lanes' entries of the slice; the walk over its set bits is scalar work.
            unsigned mine = 0u;
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                const int t = lane + i * 64;
                if (i * 64 < K * ROWS && t < K * ROWS) {
                    const int tap = t / ROWS, r = t - tap * ROWS;
                    if (sl[t] >= 0 && row0 + r < row_end) mine |= 1u << (PAIR ? tap >> 1 : tap);
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mine |= (unsigned)__shfl_xor((int)mine, o);
            unsigned rem = (unsigned)__builtin_amdgcn_readfirstlane((int)mine);
            const int n_steps = __builtin_popcount(rem);
            auto next_step = [&]() -> int {
                const int t = rem ? __builtin_ctz(rem) : T;  // T = 'no step': its entries are the 'no neighbour' row
                rem &= rem - 1u;
                return t;
            };
            int t_r[DEPTH];
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
                    const int tw = t < T ? t : T - 1;
                    const u32x4 *wsrc = s_w + (tw * FR) * 64 + lane;
#pragma unroll
                    for (int c = 0; c < NCU; ++c) {
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            const bf16x8 wf = __builtin_bit_cast(bf16x8, wsrc[(c * NB + nb) * 64]);
#pragma unroll
                            for (int g = 0; g < RG; ++g)
                                acc[g][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8, a_r[d][g][c]), acc[g][nb], 0, 0, 0);
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int d = 0; d < DEPTH - 1; ++d) {
                fetch_idx(d, e_next);
                issue(a_r[d], e_next);
            }
            fetch_idx(DEPTH - 1, e_next);
            u32x4 wr[NWR];
            wload(0, wr);
            wstore(0, wr);
            wload(1, wr);
            __syncthreads();
            for (int t0 = 0; t0 < T; t0 += DEPTH) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    const int t = t0 + d;
                    wstore((d + 1) & 1, wr);  // W[t + 1], requested one step ago, for the step after the coming barrier
                    wload(t + 2, wr);
                    // the slot freed by step t - 1 takes the gather of step t + DEPTH - 1; the entries of step t + DEPTH are read
                    // now and used one iteration later (no LDS round trip in front of a gather)
                    issue(a_r[(d + DEPTH - 1) % DEPTH], e_next);
                    fetch_idx(t + DEPTH, e_next);
                    const u32x4 *wsrc = s_w + ((d & 1) * FR) * 64 + lane;
                    // Weight fragments in batches of up to 8, each batch requested in one go BEFORE its MFMAs and the next batch before
                    // the current one is multiplied: left to itself hipcc reads two fragments, waits, multiplies, reads the next two --
                    // an exposed LDS round trip per four MFMAs, with both waves of a SIMD in lock step behind the tap barrier
                    // (phase trace, tools/win_trace.py: a 64 -> 64 tap took ~2000 cycles for 512 of MFMA).
                    constexpr int FB = (CIN * COUT >= 128 * 128 && RG >= 2) ? 4 : (FR < 8 ? FR : 8), NBATCH = FR / FB;  // (128 -> 128 with two row groups: 8-fragment double buffers spill)
                    static_assert(FR % FB == 0, "whole batches");
                    bf16x8 wfb[2][FB];
#pragma unroll
                    for (int i = 0; i < FB; ++i) wfb[0][i] = __builtin_bit_cast(bf16x8, wsrc[i * 64]);
#pragma unroll
                    for (int bt = 0; bt < NBATCH; ++bt) {
                        if (bt + 1 < NBATCH) {
#pragma unroll
                            for (int i = 0; i < FB; ++i) wfb[(bt + 1) & 1][i] = __builtin_bit_cast(bf16x8, wsrc[((bt + 1) * FB + i) * 64]);
                        }
#pragma unroll
                        for (int i = 0; i < FB; ++i) {
                            const int f = bt * FB + i, c = f / NB, nb = f - c * NB;
#pragma unroll
                            for (int g = 0; g < RG; ++g)
                                acc[g][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfb[bt & 1][i], __builtin_bit_cast(bf16x8, a_r[d][g][c]), acc[g][nb], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    __syncthreads();
                }
            }
        }

        // ---- epilogue: lane (row lrow of the group, quad lq) holds channels 16 nb + 4 lq .. + 3 of its row.  All residual
        // pieces are requested before the first is used (one exposed round trip per tile, not one per piece).
        bf16x4 res_r[RG][NB];
        if (residual) {
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                int row = row0 + 16 * g + lrow;
                row = row < row_end ? row : (row_end > 0 ? row_end - 1 : 0);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    res_r[g][nb] = *reinterpret_cast<const bf16x4 *>(residual + (int64_t)row * COUT + 16 * nb + 4 * lq);
            }
        }
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            const int row = row0 + 16 * g + lrow;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                f32x4 v = acc[g][nb];
                if (residual) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += (float)res_r[g][nb][i];
                }
                if (relu) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
                }
                bf16x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
                if (row < row_end) *reinterpret_cast<bf16x4 *>(out + (int64_t)row * COUT + 16 * nb + 4 * lq) = o;
            }
        }
        // (the next tile's rulebook slice overwrites this wave's LDS region: all of its reads above are complete -- LDS
        //  operations of one wave execute in order)
    }
}

struct WsArgs {
    const void *in, *wp;
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

template <int CIN, int COUT, int RG, int NW, bool RING>
constexpr size_t ws_lds_bytes(int K) {
    const int T = CIN == 16 ? (K + 1) / 2 : K;
    const int FR = (CIN == 16 ? 1 : CIN / 32) * (COUT / 16);
    return (size_t)(RING ? 2 : T) * FR * 1024 + (size_t)NW * (2 * (kMaxTaps + 1) + 1) * 16 * RG * 4;
}

// One instantiation: its LDS request and how many of its workgroups a CU holds (asked from the runtime once).
template <int CIN, int COUT, int RG, int DEPTH, int NW, bool RING>
struct WsKernel {
    static int wgs_per_cu(int K) {
        // The RESIDENT kernels' LDS request grows with K (all taps' weights are staged), so the answer is cached per K; the
        // dynamic-LDS limit of the instantiation is raised ONCE per device to the most any K can ask for (capped at the CU's
        // 160 KB) -- a limit set for a small K and then reused for a larger one made that launch fail (advisor, round 3).
        static std::atomic<int> cached[kMaxTaps + 1] = {};
        K = K < 1 ? 1 : (K > kMaxTaps ? kMaxTaps : K);
        std::atomic<int> &c = cached[K];
        int v = c.load(std::memory_order_relaxed);
        if (v) return v > 0 ? v : 0;
        const size_t lds = ws_lds_bytes<CIN, COUT, RG, NW, RING>(K);
        const size_t lds_limit = ws_lds_bytes<CIN, COUT, RG, NW, RING>(kMaxTaps) < (size_t)160 * 1024 ? ws_lds_bytes<CIN, COUT, RG, NW, RING>(kMaxTaps) : (size_t)160 * 1024;
        auto kern = spconv_bf16_ws<CIN, COUT, RG, DEPTH, NW, RING>;
        int nb = 0;
        static std::atomic<uint64_t> lds_set{0};
        if (lds > 160 * 1024 || (lds_limit > 65536 && !fd::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds_limit, lds_set)) ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, NW * 64, lds) != hipSuccess) {
            (void)hipGetLastError();
            nb = 0;
        }
        c.store(nb > 0 ? nb : -1, std::memory_order_relaxed);
        return nb;
    }
    static int64_t rows_per_round(int K) { return (int64_t)wgs_per_cu(K) * fd::device_cu_count() * NW * 16 * RG; }  // rows all resident workgroups cover in one pass
    static bool launch(const WsArgs &a) {
        const int wpc = wgs_per_cu(a.K);
        if (wpc <= 0) return false;
        // persistent workgroups, as many as the device holds at once; fewer when the level is small (a capacity launch is
        // sized by the capacity; the kernel splits the device's actual row count over whatever grid it gets)
        const int64_t wg_rows = (int64_t)NW * 16 * RG;
        int64_t grid = (a.n_out + wg_rows - 1) / wg_rows;
        const int64_t cap = (int64_t)wpc * fd::device_cu_count();
        if (grid > cap) grid = cap;
        if (grid < 1) grid = 1;
        const size_t lds = ws_lds_bytes<CIN, COUT, RG, NW, RING>(a.K);
        hipLaunchKernelGGL((spconv_bf16_ws<CIN, COUT, RG, DEPTH, NW, RING>), dim3((unsigned)grid), dim3(NW * 64), lds, a.stream, (const unsigned short *)a.in, (const u32x4 *)a.wp, a.bias, (const unsigned short *)a.residual, a.relu, a.nbr, a.nbr_stride,
                           a.K, a.n_out, a.n_out_dev, (unsigned short *)a.out, a.in_bytes, (!RING && fd::tuning(fd::kTuneBf16NW) != 1) ? 1 : 0);
        // RESIDENT kernels: tiles interleaved over the grid -- all workgroups sweep the row range together (round 6, in the config-3 sweep: 32 -> 32
        // 280.5 -> 274.2 us, 16 -> 16 88.5 -> 83.3, 32 -> 64 71.1 -> 67.7, 16 -> 32 46.8 -> 42.4 per two-cloud pass; "bf16_nw" = 1 brings the
        // XCD-contiguous chunks back for A/B runs).  RING kernels keep their contiguous row ranges per workgroup.
        return true;
    }
};

// RESIDENT shapes (the whole weight set in LDS, 16 free-running waves per workgroup): rg in {1, 2, 4}; the register file of a
// 1024-thread workgroup (128 per lane) excludes 4 row groups for 64 output columns and the deep ring next to 4 groups.
template <int CIN, int COUT>
bool launch_resident(const WsArgs &a, int rg, int depth) {
    if constexpr (CIN == 32 && COUT == 64) {
        // 27 taps x 4 KB of weights = 108 KB: next to them only 8 waves' rulebook slices fit the CU's 160 KB (16 waves: 166 KB --
        // that variant refused every K = 27 launch of round 3 and the layer silently ran the round-2 kernel).  512-thread
        // workgroups have 256 registers per lane: two row groups x 64 columns fit.
        if (a.K > 9) return rg >= 2 ? WsKernel<CIN, COUT, 2, 2, 8, false>::launch(a) : WsKernel<CIN, COUT, 1, 4, 8, false>::launch(a);
    }
    if constexpr (COUT <= 32) {
        if (rg >= 4) return WsKernel<CIN, COUT, 4, 2, 16, false>::launch(a);
    }
    if constexpr (COUT <= 32) {  // (two row groups x 64 columns do not fit the 128 registers of a 1024-thread workgroup)
        if (rg >= 2) {
            if constexpr (CIN == 16) {
                if (depth >= 4) return WsKernel<CIN, COUT, 2, 4, 16, false>::launch(a);
            }
            return WsKernel<CIN, COUT, 2, 2, 16, false>::launch(a);
        }
    }
    return depth >= 4 ? WsKernel<CIN, COUT, 1, 4, 16, false>::launch(a) : WsKernel<CIN, COUT, 1, 2, 16, false>::launch(a);
}

// RING shapes (W[tap] double-buffered in LDS, 8 waves per workgroup, one barrier per tap).  rg = 0: the smallest number of
// row groups per wave with which the resident workgroups cover the level in ONE pass (a second, partly empty pass would cost
// a whole walk over the taps), else the largest.
template <int CIN, int COUT>
bool launch_ring(const WsArgs &a, int rg, int depth) {
    constexpr int kMaxRG = COUT > 64 ? 2 : 4;  // (128 output columns: 3 row groups spill)
    if (rg <= 0) {
        const int64_t n = a.n_expected;
        if (WsKernel<CIN, COUT, 1, 2, 8, true>::rows_per_round(a.K) >= n) rg = 1;
        else if (WsKernel<CIN, COUT, 2, 2, 8, true>::rows_per_round(a.K) >= n) rg = 2;
        else if (kMaxRG >= 3 && WsKernel<CIN, COUT, kMaxRG >= 3 ? 3 : 2, 2, 8, true>::rows_per_round(a.K) >= n) rg = 3;
        else rg = kMaxRG >= 4 ? 3 : kMaxRG;  // two passes either way: three row groups per wave (384-row passes) measured best on
                                              // 64 -> 64 at 155k rows (68 us; rg 2: 72, rg 4: 79 -- tools/spconv_bench.py, round 4)
    }
    if (fd::tuning(fd::kTuneBf16NW) == 4) {  // experiment: 4-wave workgroups (two per CU, barriers decoupled), rg row groups per wave
        if (rg >= 4) return WsKernel<CIN, COUT, 4, 2, 4, true>::launch(a);
        if (rg >= 3) return WsKernel<CIN, COUT, 3, 2, 4, true>::launch(a);
        return WsKernel<CIN, COUT, 2, 2, 4, true>::launch(a);
    }
    if constexpr (kMaxRG >= 4) {
        if (rg >= 4) return WsKernel<CIN, COUT, 4, 2, 8, true>::launch(a);
        if (rg >= 3) return WsKernel<CIN, COUT, 3, 2, 8, true>::launch(a);
    }
    if (rg >= 2) return WsKernel<CIN, COUT, 2, 2, 8, true>::launch(a);
    if constexpr (CIN * COUT <= 64 * 64) {
        if (depth >= 4) return WsKernel<CIN, COUT, 1, 4, 8, true>::launch(a);
    }
    return WsKernel<CIN, COUT, 1, 2, 8, true>::launch(a);
}

}  // namespace

namespace fd {
// returns 1 when launched, 0 when the shape is not covered (caller falls back to the older kernels)
int spconv_bf16_ws_dispatch(const void *in, const void *wp, const float *bias, const void *residual, int relu, const int *nbr, int64_t nbr_stride, int K,
                            int64_t n_in_bound, int n_out, const int *n_out_dev, int64_t n_expected, int cin, int cout, void *out, hipStream_t stream) {
    if (n_in_bound * cin * 2 >= (1ll << 31)) return 0;  // the 'missing neighbour' offset must lie beyond the buffer
    WsArgs a{in, wp, bias, residual, relu, nbr, nbr_stride, K, n_out, n_out_dev, out, (unsigned)(n_in_bound * cin * 2), n_expected, stream};
    int rg = fd::tuning(fd::kTuneBf16RG), depth = fd::tuning(fd::kTuneBf16Depth);  // 0 = the heuristics below
    const bool ring = cin * cout > 32 * 64;
    if (!ring) {
        if (rg <= 0) rg = 1;  // 16 waves x 16 rows: the most waves in flight per CU (measured best on every RESIDENT shape)
        if (depth <= 0) depth = cin == 16 ? 2 : 4;
    }
    const int key = cin * 1000 + cout;
    bool ok = false;
    switch (key) {
        // (a RESIDENT variant whose LDS request does not fit falls back to fewer rows per wave)
        case 16016: for (; rg >= 1 && !(ok = launch_resident<16, 16>(a, rg, depth)); rg >>= 1) {} break;
        case 16032: for (; rg >= 1 && !(ok = launch_resident<16, 32>(a, rg, depth)); rg >>= 1) {} break;
        case 32032: for (; rg >= 1 && !(ok = launch_resident<32, 32>(a, rg, depth)); rg >>= 1) {} break;
        case 32064:
            if (fd::tuning(fd::kTuneBf16RG) <= 0) rg = 2;  // (8-wave workgroups: 32 rows per wave keep 256 rows per workgroup pass)
            for (; rg >= 1 && !(ok = launch_resident<32, 64>(a, rg, depth)); rg >>= 1) {}
            break;
        case 64064: ok = launch_ring<64, 64>(a, rg, depth); break;
        case 64128: ok = launch_ring<64, 128>(a, rg, depth); break;
        case 128128: ok = launch_ring<128, 128>(a, rg, depth); break;
        default: return 0;
    }
    return ok ? 1 : 0;
}
}  // namespace fd
