// bf16 submanifold convolution with an LDS window of input rows (the 64- and 128-channel levels of configs 3 / 4 / 5).
//
// Replaces spconv 1.0's indice_subm_conv for the SubM convolutions of det3d/models/backbones/scn.py:119-141 in the bf16
// configurations; same arithmetic, summation order and results (bit for bit) as the RING kernels of fd_spconv_bf16.hip, which
// stay the path for strided convolutions and for feature matrices of 2 GB and more.
//
// What bounded fd_spconv_bf16.hip (profiles/round3_gather_probe.txt, round3_bf16_ws_sq_counters.txt): the per-CU vector-memory
// path.  It issues one gather instruction per (16 rows, tap, 32-channel chunk) whether the neighbours exist or not, and a
// dwordx4 wave instruction costs the texture path >= 32 cycles per CU whatever its lanes do (45-63 cycles for 16 scattered
// 64-byte pieces).  But the rows are spatially sorted: 81-92 % of the pairs of a 256-row tile lie inside [tile - 64, tile + 256
// + 64) in index order, and only 15-25 % of the (16-row group, tap) items have a pair outside (tools/window_stats.py,
// profiles/round4_window_stats.txt).  So:
//   * per pass a workgroup loads the WINDOW [row0 - H, row0 + NW * 16 * RG + H) once, linearly (16 bytes per lane, consecutive
//     lanes consecutive pieces), into LDS, 16-byte pieces XOR-swizzled by the row so that the MFMA-fragment reads of 16 nearly
//     consecutive rows spread over the banks; one extra all-zero row stands for a missing neighbour;
//   * per (group, tap) the B operand is read from the window (ds_read_b128: ~10 x cheaper for the CU than the gather) unless one
//     of the 16 rows' neighbours lies outside it -- then (wave-uniform branch) the item takes the bounds-checked global gather of
//     the old kernel for all its lanes.  Correctness never depends on locality: a rulebook without any is just slower;
//   * everything else as in fd_spconv_bf16.hip's RING form: a wave owns 16 * RG rows and all output columns, register accumulators
//     over all taps, W[tap] double-buffered in LDS with one barrier per tap, rulebook slices by LDS-DMA, operands fetched DEPTH
//     taps ahead, transposed MFMA (v_mfma_f32_16x16x32_bf16, A = weights).
#include "fd_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxTaps = 27;

// Tuning builds only (tools/probes/build_win_trace.sh, -DFD_WIN_TRACE): lane 0 of wave 0 of every workgroup accumulates shader cycles
// per phase: [0] window + slice staging, [1] operand issue + index fetch, [2] MFMA block, [3] barrier wait, [4] epilogue, [5] whole
// kernel, [6] weight ring moves, [7] passes.
#ifdef FD_WIN_TRACE
__device__ unsigned long long *g_wintrace;
#define FD_WT(var) const unsigned long long var = __builtin_readcyclecounter()
#define FD_WADD(i, v) do { if (tid == 0) wacc[i] += (v); } while (0)
#else
#define FD_WT(var)
#define FD_WADD(i, v)
#endif

template <int CIN, int COUT, int RG, int NW, int HALO>
__global__ void __launch_bounds__(NW * 64) spconv_bf16_win(const unsigned short *__restrict__ in, const u32x4 *__restrict__ wp,
                                                           const float *__restrict__ bias, const unsigned short *__restrict__ residual, int relu,
                                                           const int *__restrict__ nbr, int64_t nbr_stride, int K, int n_out,
                                                           const int *__restrict__ n_out_dev, unsigned short *__restrict__ out, unsigned in_bytes, int n_in) {
    constexpr int DEPTH = 2;
    constexpr int NCU = CIN / 32;             // 32-channel chunks (MFMAs along K) per tap
    constexpr int NB = COUT / 16;
    constexpr int FR = NCU * NB;              // weight fragments (1 KB each) of a tap
    constexpr int ROWS = 16 * RG;             // rows of a wave's tile
    constexpr int TM = NW * ROWS;             // rows of a workgroup pass
    constexpr int WIN = TM + 2 * HALO;        // window rows; slot WIN = the all-zero row
    constexpr int PIECES = CIN / 8;           // 16-byte pieces of a row (4 per 32-channel chunk)
    constexpr int kRowShift = CIN == 32 ? 6 : CIN == 64 ? 7 : 8;  // log2(bytes of an input row)
    static_assert(CIN * 2 == (1 << kRowShift), "CIN must be 32, 64 or 128");
    constexpr int NWR = (FR + NW - 1) / NW;   // fragments of a tap that one wave moves
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4 *s_w = reinterpret_cast<u32x4 *>(smem);                                   // [2][FR][64]
    u32x4 *s_win = s_w + 2 * FR * 64;                                               // [(WIN + 1) * PIECES] swizzled pieces
    constexpr int kSliceInts = (kMaxTaps + 1) * ROWS;
    constexpr int kWaveInts = kSliceInts + ROWS;                                    // one slice buffer + the 'no neighbour' row
    int *s_nbr = reinterpret_cast<int *>(s_win + (WIN + 1) * PIECES);               // [NW][kWaveInts], wave-private
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: what derives from it stays in SGPRs
    const int lrow = lane & 15, lq = lane >> 4;
    n_out = fd::device_count(n_out, n_out_dev);
    int *s = s_nbr + wave * kWaveInts;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(in), 0, (int)in_bytes, 0x00020000);
    const unsigned lane_off = (unsigned)(lq * 16);
    const int T = K;

#ifdef FD_WIN_TRACE
    unsigned long long wacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    FD_WT(t_start);
    const unsigned lb = fd::xcd_swizzle(blockIdx.x, gridDim.x);
    const int rows_per_wg = (((n_out + (int)gridDim.x - 1) / (int)gridDim.x) + 15) & ~15;
    const int64_t wg_r0 = (int64_t)lb * rows_per_wg;
    if (wg_r0 >= n_out) return;  // (uniform for the workgroup)
    const int wg_r1 = (int)(wg_r0 + rows_per_wg < n_out ? wg_r0 + rows_per_wg : n_out);
    const int n_iter = (rows_per_wg + TM - 1) / TM;

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
    constexpr int NPRE = (kMaxTaps * ROWS + 63) / 64;
    static_assert(NPRE * 64 <= kSliceInts, "a slice buffer takes whole DMA instructions");
    for (int r = lane; r < ROWS; r += 64) s[kSliceInts + r] = -1;  // the 'no neighbour' row
    // the all-zero window row
    for (int i = tid; i < PIECES; i += NW * 64) s_win[WIN * PIECES + i] = (u32x4){0u, 0u, 0u, 0u};

    for (int it = 0; it < n_iter; ++it) {
        const int pass_r0 = (int)(wg_r0 + (int64_t)it * TM);        // first row of this pass (uniform)
        const int64_t r = (int64_t)pass_r0 + wave * ROWS;
        const int row0 = (int)(r < wg_r1 ? r : wg_r1);
        const int win0 = pass_r0 - HALO;                            // first row of the window (may be negative: zeros)
        FD_WT(t_p0);
        __syncthreads();  // every wave is done with the previous pass's window, slices and weight ring
        // ---- rulebook slice of this wave's tile by LDS-DMA, window rows by plain loads (consecutive lanes = consecutive 16-byte
        //      pieces of consecutive rows: linear 1-KB loads), stored at their swizzled position
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            if (i * 64 < K * ROWS) {  // (uniform)
                const int t = lane + i * 64;
                int k = t / ROWS;
                const int rr = t - k * ROWS;
                k = k < K ? k : K - 1;
                int o = row0 + rr;
                o = o < n_out ? o : n_out - 1;  // masked on use
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) int *)(nbr + (int64_t)k * nbr_stride + o),
                                                 (__attribute__((address_space(3))) int *)(s + i * 64), 4, 0, 0);
            }
        }
        {
            constexpr int NPC = (WIN * PIECES + NW * 64 - 1) / (NW * 64);
            u32x4 wv[NPC];
#pragma unroll
            for (int i = 0; i < NPC; ++i) {
                const int pc = tid + i * NW * 64;                  // piece index in the window, row-major
                const int64_t grow = (int64_t)win0 + pc / PIECES;
                const bool ok = pc < WIN * PIECES && grow >= 0 && grow < n_in;
                const unsigned off = ok ? (unsigned)((grow << kRowShift) + (pc % PIECES) * 16) : 0xfffffff0u;  // out of range: zeros
                wv[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NPC; ++i) {
                const int pc = tid + i * NW * 64;
                const int sl_ = pc / PIECES, p = pc % PIECES;
                if (pc < WIN * PIECES) s_win[sl_ * PIECES + (p ^ (sl_ & (PIECES - 1)))] = wv[i];
            }
        }
        // W[t + 1] is stored at the top of tap t from a register set that was loaded TWO taps earlier (two sets, by tap parity):
        // a tap of 16-32 MFMAs per wave is shorter than an L2 round trip, and with one tap of lead (fd_spconv_bf16.hip) every
        // tap opened by waiting for its weights
        u32x4 wr[2][NWR];
        wload(0, wr[0]);
        wstore(0, wr[0]);
        wload(1, wr[0]);
        wload(2, wr[1]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the slice DMA)
        __syncthreads();
        FD_WT(t_p1);
        FD_WADD(0, t_p1 - t_p0);
        FD_WADD(7, 1);

        bool valid[RG];
#pragma unroll
        for (int g = 0; g < RG; ++g) valid[g] = row0 + 16 * g + lrow < wg_r1;
        const bool wave_live = row0 < wg_r1;  // (wave-uniform) a wave past the end of the range only moves weights and meets the barriers
        f32x4 acc[RG][NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (bias) bv = *reinterpret_cast<const f32x4 *>(bias + 16 * nb + 4 * lq);
#pragma unroll
            for (int g = 0; g < RG; ++g) acc[g][nb] = bv;
        }

        // rulebook entries of tap t for this lane's row of every group (taps past the end: the 'no neighbour' row)
        auto fetch_idx = [&](int t, int(&e)[RG]) {
            const int *p = (t < K ? s + t * ROWS : s + kSliceInts) + lrow;
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                const int v = p[16 * g];
                e[g] = valid[g] ? v : -1;
            }
        };
        // B operands of a tap: from the window when all 16 neighbours of the group are inside it (or missing), else the global gather.
        // The two sources land in SEPARATE register sets (l: LDS, m: memory) that are selected at the MFMA: a register that is
        // written by a buffer load on one path and by an LDS read on the other makes hipcc wait for every outstanding vector-memory
        // load (write-after-write on an unknown path) before each LDS read -- i.e. for the weight load issued a few instructions
        // earlier, a full L2 round trip per tap (phase trace of the first version: 560 cycles per tap in this function).
        auto issue = [&](u32x4(&dl)[RG][NCU], u32x4(&dm)[RG][NCU], unsigned long long &farmask, const int(&e)[RG]) {
            farmask = 0ull;
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                const unsigned slot = (unsigned)(e[g] - win0);
                const bool far = e[g] >= 0 && slot >= (unsigned)WIN;
                const bool any_far = __builtin_amdgcn_ballot_w64(far) != 0ull;  // (wave-uniform)
                if (any_far) {
                    farmask |= 1ull << g;
                    const unsigned voff = ((unsigned)e[g] << kRowShift) + lane_off;  // -1 -> just below 2^32: out of range, reads zeros
#pragma unroll
                    for (int c = 0; c < NCU; ++c) dm[g][c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + c * 64, 0, 0);
                }
                // (always: an item that takes the gather reads the zero row here -- no branch around LDS reads)
                const unsigned sl_ = (e[g] >= 0 && !any_far) ? slot : (unsigned)WIN;
                const u32x4 *rowp = s_win + sl_ * PIECES;
                const unsigned sw = sl_ & (PIECES - 1);
#pragma unroll
                for (int c = 0; c < NCU; ++c) dl[g][c] = rowp[(unsigned)(c * 4 + lq) ^ sw];
            }
        };

        u32x4 a_r[DEPTH][RG][NCU], m_r[DEPTH][RG][NCU];
        unsigned long long far_r[DEPTH];
        int e_next[RG];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            far_r[d] = 0ull;
#pragma unroll
            for (int g = 0; g < RG; ++g)
#pragma unroll
                for (int c = 0; c < NCU; ++c) m_r[d][g][c] = (u32x4){0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d) {
            fetch_idx(d, e_next);
            issue(a_r[d], m_r[d], far_r[d], e_next);
        }
        fetch_idx(DEPTH - 1, e_next);
        for (int t0 = 0; t0 < T; t0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int t = t0 + d;
                FD_WT(t_a);
                wstore((d + 1) & 1, wr[d]);  // W[t + 1], requested two taps ago, for the tap after the coming barrier
                wload(t + 3, wr[d]);
                FD_WT(t_b);
                FD_WADD(6, t_b - t_a);
                if (wave_live) {
                    issue(a_r[(d + DEPTH - 1) % DEPTH], m_r[(d + DEPTH - 1) % DEPTH], far_r[(d + DEPTH - 1) % DEPTH], e_next);
                    fetch_idx(t + DEPTH, e_next);
                    FD_WT(t_c);
                    FD_WADD(1, t_c - t_b);
                    const u32x4 *wsrc = s_w + ((d & 1) * FR) * 64 + lane;
                    // operands of this tap: the window's, or the gathered ones of an item that had a far neighbour (wave-uniform select)
                    u32x4 bop[RG][NCU];
#pragma unroll
                    for (int g = 0; g < RG; ++g) {
                        const bool gf = (far_r[d] >> g) & 1ull;
#pragma unroll
                        for (int c = 0; c < NCU; ++c) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) bop[g][c][i] = gf ? m_r[d][g][c][i] : a_r[d][g][c][i];
                        }
                    }
#pragma unroll
                    for (int c = 0; c < NCU; ++c) {
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            const bf16x8 wf = __builtin_bit_cast(bf16x8, wsrc[(c * NB + nb) * 64]);
#pragma unroll
                            for (int g = 0; g < RG; ++g)
                                acc[g][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8, bop[g][c]), acc[g][nb], 0, 0, 0);
                        }
                    }
                    FD_WT(t_d);
                    FD_WADD(2, t_d - t_c);
                }
                FD_WT(t_e);
                __syncthreads();
                FD_WT(t_f);
                FD_WADD(3, t_f - t_e);
            }
        }
        FD_WT(t_ep);

        // ---- epilogue: lane (row lrow of the group, quad lq) holds channels 16 nb + 4 lq .. + 3 of its row
        bf16x4 res_r[RG][NB];
        if (residual) {
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                int row = row0 + 16 * g + lrow;
                row = row < wg_r1 ? row : (wg_r1 > 0 ? wg_r1 - 1 : 0);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) res_r[g][nb] = *reinterpret_cast<const bf16x4 *>(residual + (int64_t)row * COUT + 16 * nb + 4 * lq);
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
                if (row < wg_r1) *reinterpret_cast<bf16x4 *>(out + (int64_t)row * COUT + 16 * nb + 4 * lq) = o;
            }
        }
        FD_WT(t_ee);
        FD_WADD(4, t_ee - t_ep);
    }
#ifdef FD_WIN_TRACE
    if (tid == 0 && g_wintrace) {
        wacc[5] = __builtin_readcyclecounter() - t_start;
        for (int i = 0; i < 8; ++i) g_wintrace[(size_t)blockIdx.x * 8 + i] = wacc[i];
    }
#endif
}

template <int CIN, int COUT, int RG, int NW, int HALO>
constexpr size_t win_lds_bytes() {
    return (size_t)2 * (CIN / 32) * (COUT / 16) * 1024 + (size_t)(NW * 16 * RG + 2 * HALO + 1) * CIN * 2 + (size_t)NW * ((kMaxTaps + 1) + 1) * 16 * RG * 4;
}

template <int CIN, int COUT, int RG, int NW, int HALO>
int launch_win(const void *in, const void *wp, const float *bias, const void *residual, int relu, const int *nbr, int64_t nbr_stride, int K, int n_out,
               const int *n_out_dev, int64_t n_expected, void *out, unsigned in_bytes, int n_in, hipStream_t stream) {
    constexpr size_t lds = win_lds_bytes<CIN, COUT, RG, NW, HALO>();
    static_assert(lds <= 160 * 1024, "LDS request of the windowed kernel");
    auto kern = spconv_bf16_win<CIN, COUT, RG, NW, HALO>;
    static std::atomic<uint64_t> lds_set{0};
    if (lds > 65536 && !fd::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, lds_set)) return 0;
    // persistent workgroups, one per CU (the window + weight ring take most of a CU's LDS); fewer when the level is small
    constexpr int64_t wg_rows = (int64_t)NW * 16 * RG;
    int64_t grid = (n_expected + wg_rows - 1) / wg_rows;
    const int64_t cap = fd::device_cu_count();
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, stream, (const unsigned short *)in, (const u32x4 *)wp, bias, (const unsigned short *)residual,
                       relu, nbr, nbr_stride, K, n_out, n_out_dev, (unsigned short *)out, in_bytes, n_in);
    return 1;
}

}  // namespace

#ifdef FD_WIN_TRACE
extern "C" int fd_debug_set_win_trace(void *p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_wintrace), &p, sizeof(p)) == hipSuccess ? 0 : -1; }
#endif

namespace fd {
// SubM convolutions (input rows = output rows) of the 64- and 128-channel levels.  1 = launched, 0 = not this kernel's case.
int spconv_bf16_win_dispatch(const void *in, const void *wp, const float *bias, const void *residual, int relu, const int *nbr, int64_t nbr_stride, int K,
                             int64_t n_in_bound, int n_out, const int *n_out_dev, int64_t n_expected, int cin, int cout, void *out, hipStream_t stream) {
    if (n_in_bound * cin * 2 >= (1ll << 31) || n_in_bound != n_out || K != kMaxTaps) return 0;
    const unsigned in_bytes = (unsigned)(n_in_bound * cin * 2);
    if (cin == 64 && cout == 64)
        return launch_win<64, 64, 2, 8, 64>(in, wp, bias, residual, relu, nbr, nbr_stride, K, n_out, n_out_dev, n_expected, out, in_bytes, (int)n_in_bound, stream);
    if (cin == 128 && cout == 128)
        return launch_win<128, 128, 1, 8, 64>(in, wp, bias, residual, relu, nbr, nbr_stride, K, n_out, n_out_dev, n_expected, out, in_bytes, (int)n_in_bound, stream);
    return 0;
}
}  // namespace fd
