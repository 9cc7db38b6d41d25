// bf16 sparse convolution, round-5 formulation for the wide levels: an LDS WINDOW of input rows + big register tiles.
//
// Replaces spconv 1.0's indice_subm_conv for the SubM convolutions of det3d/models/backbones/scn.py:119-141 (64 -> 64, 128 -> 128:
// 9 of the 21 convolutions, 0.52 of the 0.84 ms the bf16 sparse convolutions took in round 4) in the bf16
// configurations (BASELINE configs[2..4]), fused with the folded BatchNorm1d, residual add and ReLU (scn.py:67-78).
//
// What bounded the RING kernels of fd_spconv_bf16.hip, both at once (rounds 3 and 4):
//   (a) the per-CU texture path: one 16-byte gather instruction per (16 rows, tap, 32 channels) costs 45-64 cycles whatever its
//       lanes do, and the 27 taps re-read nearly the same rows: 64 KB per tap and CU at 16 B/clk = 4096 cycles for 2048 of MFMA
//       on 128 -> 128, 1600 for 512 on 64 -> 64;
//   (b) LDS bytes per MFMA: a wave owned 16-48 rows and read the whole W[tap] from LDS for them -- eight waves, all at the same
//       moment behind the tap barrier.
// Round 4's window kernel removed (a) and made (b) worse (16 rows per wave); this one does both:
//   * a workgroup = NW waves (8: two per SIMD, half the register file each; 4: one per SIMD, "bf16_nw") owns TM = NW * 32 * RGS
//     consecutive output rows per pass and stages the input rows [first - HALO, last + HALO] ONCE, by LDS-DMA (global_load_lds_dwordx4:
//     no registers, no ds_write), 16-byte pieces XOR-swizzled by the row.  Rows are spatially sorted (fd_index.hip), so 90 % of all
//     pairs lie inside; a (32-row group, tap) item with a neighbour outside takes the bounds-checked global gather for its lanes
//     instead (exec-masked) -- correctness never depends on locality;
//   * a wave owns 32 * RGS rows and ALL output columns: accumulators in registers over all taps (v_mfma_f32_32x32x16_bf16,
//     transposed: A = weights), one 1-KB weight fragment read from LDS feeds RGS MFMAs of 32 cycles, one 1-KB row fragment COUT / 32.
//     That is (1 / RGS + 32 / COUT) KB of LDS reads per MFMA, four SIMDs wide: 160 B/clk on 128 -> 128 at RGS = 1, 128 B/clk on
//     64 -> 64 at RGS = 2, against the port's 128 -- the 128-channel loop is LDS-bound just above its MFMA time (a (tap, 64-channel)
//     step of the two waves of a SIMD: 1024 cycles of MFMA, 1250 of LDS, measured 1330-1480); RGS = 2 there needs 128 accumulator + 64
//     operand registers per wave and spills at eight waves, and runs at the in-order issue of a lone wave at four (62 us vs 57-59);
//   * weights travel global -> registers -> LDS through a ring of three (tap, 64-channel) stages, two steps ahead (plain loads +
//     ds_write: hipcc answers every wait with vmcnt(0) while an LDS-DMA is pending); rulebook entries come straight from global memory
//     into registers a tap ahead (coalesced 128-byte reads, no LDS slice);
//   * the B operand of step u + 1 (LDS reads or gathers) is requested in the gaps behind the MFMAs of step u (register double buffer).
// Per output element the summation order is fixed (taps ascending, 16-channel MFMA steps ascending), so results do not depend on
// RGS, on the grid, or on which items took the gather (tested bit for bit); they differ from the RING kernels' by fp32 summation
// order only (both are checked against the bf16 oracle layer by layer, within one bf16 ulp).
#include "fd_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxTaps = 27;
constexpr int kStages = 3;

#ifdef FD_WIN_TRACE  // tuning builds only: lane 0 of wave 0 accumulates shader cycles per phase (tools/bf16win_trace.py)
__device__ unsigned long long *g_wintrace;
#define FD_WT(var) const unsigned long long var = __builtin_readcyclecounter()
#define FD_WADD(i, v) do { if (tid == 0) wacc[i] += (v); } while (0)
#else
#define FD_WT(var)
#define FD_WADD(i, v)
#endif

template <int CIN, int COUT>
struct WinShape {
    static constexpr int SC = CIN < 64 ? CIN : 64;  // input channels of a step (one W stage)
    static constexpr int H = CIN / SC;              // steps per tap
    static constexpr int KS = SC / 16;              // MFMA k-steps per step
    static constexpr int CB = COUT / 32;            // 32-column blocks
    static constexpr int ROWB = CIN * 2;            // bytes of an input row
    static constexpr int PIECES = CIN / 8;          // 16-byte pieces of a row
    static constexpr int RP = 16 / PIECES > 0 ? 16 / PIECES : 1;  // rows per 256 bytes (the 16 bank groups of a ds_read_b128)
    static constexpr int SB = SC * COUT * 2;        // bytes of a W stage
};

template <int CIN, int COUT, int RGS, int HALO, int NW>
constexpr size_t win_lds_bytes() {
    return (size_t)kStages * WinShape<CIN, COUT>::SB + (size_t)(NW * 32 * RGS + 2 * HALO + 1) * WinShape<CIN, COUT>::ROWB;
}

// NW waves per workgroup: 4 = one per SIMD with the whole register file (512), 8 = two per SIMD with half of it each
template <int CIN, int COUT, int RGS, int HALO, int NW>
__global__ void __launch_bounds__(NW * 64) spconv_bf16_win(const unsigned short *__restrict__ in, const unsigned char *__restrict__ wp,
                                                               const float *__restrict__ bias, const unsigned short *__restrict__ residual, int relu,
                                                               const int *__restrict__ nbr, int64_t nbr_stride, int K, int n_out,
                                                               const int *__restrict__ n_out_dev, unsigned short *__restrict__ out, unsigned in_bytes,
                                                               int n_in) {
    using S = WinShape<CIN, COUT>;
    constexpr int kWaves = NW;
    constexpr int H = S::H, KS = S::KS, CB = S::CB, ROWB = S::ROWB, PIECES = S::PIECES, RP = S::RP, SB = S::SB;
    constexpr int kRowShift = CIN == 32 ? 6 : CIN == 64 ? 7 : 8;
    static_assert(ROWB == (1 << kRowShift), "CIN must be 32, 64 or 128");
    static_assert(COUT % 32 == 0 && COUT <= 128, "COUT must be 32, 64 or 128");
    constexpr int TM = kWaves * 32 * RGS;      // output rows of a workgroup pass
    constexpr int WIN = TM + 2 * HALO;         // window rows; row WIN = the all-zero row
    constexpr int NCH = WIN * ROWB / 1024;     // 1-KB DMA chunks of the window
    static_assert((WIN * ROWB) % 1024 == 0, "the window is a whole number of DMA instructions");
    constexpr int WCH = SB / 1024;             // 1-KB DMA chunks of a W stage
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char *s_w = smem;                       // [kStages][SB]: fragment (k-step s, column block cb) at (s * CB + cb) * 1024
    unsigned char *s_win = smem + kStages * SB;      // [(WIN + 1) * ROWB]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 31, kg = lane >> 5;
    n_out = fd::device_count(n_out, n_out_dev);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(in), 0, (int)in_bytes, 0x00020000);
    const int U = K * H;                             // steps
#ifdef FD_WIN_TRACE
    unsigned long long wacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    FD_WT(t_start);

    // rows of this workgroup: XCD b % 8 owns one contiguous eighth of the level (fd::xcd_swizzle; speed only)
    const unsigned lb = fd::xcd_swizzle(blockIdx.x, gridDim.x);
    const int rows_per_wg = (((n_out + (int)gridDim.x - 1) / (int)gridDim.x) + 31) & ~31;
    const int64_t wg_r0 = (int64_t)lb * rows_per_wg;
    if (wg_r0 >= n_out) return;  // (uniform for the workgroup)
    const int wg_r1 = (int)(wg_r0 + rows_per_wg < n_out ? wg_r0 + rows_per_wg : n_out);
    const int n_iter = (wg_r1 - (int)wg_r0 + TM - 1) / TM;

    // the all-zero window row (read by lanes whose neighbour does not exist)
    for (int i = tid; i < ROWB / 16; i += kWaves * 64) reinterpret_cast<u32x4 *>(s_win + WIN * ROWB)[i] = (u32x4){0u, 0u, 0u, 0u};

    // W of step u -> ring stage u % kStages (steps past the end re-read the last one: their rows are all zero).  ``part`` / ``parts``: the
    // wave's DMA instructions are dealt over the k-steps of the running step (one per ~8 MFMAs): issued back to back they fill the
    // texture queue and the wave sits in the issue stage while its matrix pipe drains (phase trace: 900 of 2150 cycles per step)
    auto w_issue = [&](int u, int part, int parts) {
        const int uu = u < U ? u : U - 1;
        const unsigned char *src = wp + (int64_t)uu * SB + lane * 16;
        unsigned char *dst = s_w + (u % kStages) * SB;
        constexpr int PER = (WCH + kWaves - 1) / kWaves;  // DMA instructions per wave and stage
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if (i % parts != part) continue;
            const int c = wave + i * kWaves;
            if (WCH % kWaves == 0 || c < WCH)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + c * 1024),
                                                 (__attribute__((address_space(3))) void *)(dst + c * 1024), 16, 0, 0);
        }
    };

    constexpr int kWPer = (WCH + kWaves - 1) / kWaves;  // 1-KB pieces of a W stage per wave
    auto w_load = [&](int u, int i, u32x4(&wr)[kWPer]) {
        const int uu = u < U ? u : U - 1;
        const int c = wave + i * kWaves;
        if (WCH % kWaves == 0 || c < WCH) wr[i] = *reinterpret_cast<const u32x4 *>(wp + (int64_t)uu * SB + c * 1024 + lane * 16);
    };
    auto w_store = [&](int u, int i, const u32x4(&wr)[kWPer]) {
        const int c = wave + i * kWaves;
        if (WCH % kWaves == 0 || c < WCH) *reinterpret_cast<u32x4 *>(s_w + (u % kStages) * SB + c * 1024 + lane * 16) = wr[i];
    };

    for (int it = 0; it < n_iter; ++it) {
        const int pass_r0 = (int)wg_r0 + it * TM;  // (uniform)
        const int win0 = pass_r0 - HALO;           // global row of window slot 0 (may be negative)
        const int row0 = pass_r0 + wave * 32 * RGS;  // first row of this wave
        FD_WT(t_p0);
        __syncthreads();  // every wave is done with the previous pass's window and weight ring (and the zero row is written)
        // ---- window rows -> LDS by DMA: lane l of chunk c fills the 16-byte slot q = 64 c + l = (row r, position pp), with the
        //      piece pp ^ swz(r) of global row win0 + r.  Rows outside [0, n_in) are clamped: no rulebook entry ever points at them.
        for (int c = wave; c < NCH; c += kWaves) {
            const int q = c * 64 + lane;
            const int r = q / PIECES, pp = q % PIECES;
            int64_t grow = (int64_t)win0 + r;
            grow = grow < 0 ? 0 : (grow < n_in ? grow : n_in - 1);
            const int piece = pp ^ ((r / RP) % PIECES);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(reinterpret_cast<const unsigned char *>(in) + (grow << kRowShift) + piece * 16),
                (__attribute__((address_space(3))) void *)(s_win + c * 1024), 16, 0, 0);
        }
        w_issue(0, 0, 1);  // W(0) by DMA, complete before the first barrier; W(1) into the ring registers (stored during step 0)
        u32x4 wr[kWPer];
#pragma unroll
        for (int i = 0; i < kWPer; ++i) w_load(1, i, wr);

        // rulebook entries of tap t for this lane's row of every group (taps past the end and rows past the range: no neighbour)
        bool valid[RGS];
#pragma unroll
        for (int g = 0; g < RGS; ++g) valid[g] = row0 + 32 * g + ln < wg_r1;
        auto fetch_idx = [&](int t, int(&e)[RGS]) {
            const int tt = t < K ? t : K - 1;
#pragma unroll
            for (int g = 0; g < RGS; ++g) {
                int o = row0 + 32 * g + ln;
                o = o < n_out ? o : n_out - 1;
                e[g] = nbr[(int64_t)tt * nbr_stride + o];
            }
        };
        auto copy_idx = [&](int(&d)[RGS], const int(&e)[RGS]) {
#pragma unroll
            for (int g = 0; g < RGS; ++g) d[g] = e[g];
        };
        auto mask_idx = [&](int t, int(&e)[RGS]) {
#pragma unroll
            for (int g = 0; g < RGS; ++g) e[g] = (valid[g] && t < K) ? e[g] : -1;
        };
        // B operand of step (tap entries e, half h), row group g: LDS window reads for every lane -- a lane whose neighbour does not exist
        // or lies outside the window reads the zero row -- then, for the lanes with a neighbour OUTSIDE the window only (exec-masked: the
        // texture path charges for the active lanes' cache lines, not per instruction as it does for out-of-range lanes), the global gather
        // into the same registers.  26 % of the (32-row group, tap) items of the wide levels have such a lane (the y-neighbours across an
        // 8 x 8 index tile), but only 8 % of the pairs.
        auto load_b_group = [&](const int(&e)[RGS], int h, u32x4(&b)[RGS][KS], int g) {
            const int ev = e[g];
            const int slot = ev - win0;
            const bool inwin = ev >= 0 && (unsigned)slot < (unsigned)WIN;
            const bool far = ev >= 0 && !inwin;
            const unsigned base = inwin ? (unsigned)slot * ROWB : (unsigned)(WIN * ROWB);
            const unsigned sw = inwin ? (unsigned)((slot / RP) % PIECES) : 0u;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const unsigned p = (unsigned)(h * (S::SC / 8) + 2 * s + kg);
                b[g][s] = *reinterpret_cast<const u32x4 *>(s_win + base + ((p ^ sw) << 4));
            }
            if (far) {
                const unsigned voff = (unsigned)ev << kRowShift;
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    b[g][s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + (unsigned)(h * (S::SC / 8) + 2 * s + kg) * 16u, 0, 0);
            }
        };
        auto load_b = [&](const int(&e)[RGS], int h, u32x4(&b)[RGS][KS]) {
#pragma unroll
            for (int g = 0; g < RGS; ++g) load_b_group(e, h, b, g);
        };

        // bias enters through the accumulators' initial value: lane (row ln of a group, half kg) holds channels
        // 32 cb + 8 j + 4 kg + (0..3) in registers 4 j .. 4 j + 3
        f32x16 acc[RGS][CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            f32x16 bv;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 q = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (bias) q = *reinterpret_cast<const f32x4 *>(bias + 32 * cb + 8 * j + 4 * kg);
                bv[4 * j] = q[0]; bv[4 * j + 1] = q[1]; bv[4 * j + 2] = q[2]; bv[4 * j + 3] = q[3];
            }
#pragma unroll
            for (int g = 0; g < RGS; ++g) acc[g][cb] = bv;
        }

        int e_cur[RGS], e_nxt[RGS];
        fetch_idx(0, e_cur);
        fetch_idx(1, e_nxt);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the window and of W stage 0 has landed
        __syncthreads();
        mask_idx(0, e_cur);
        mask_idx(1, e_nxt);
        u32x4 bq[2][RGS][KS];
        load_b(e_cur, 0, bq[0]);
        FD_WT(t_p1);
        FD_WADD(0, t_p1 - t_p0);

        // one step: requests for step u + 1 / u + 2 first, then the MFMAs of step u, then the hand-over barrier
        // One step: the MFMAs of step u (B operand ``b``) with the requests for what comes next placed INTO THE GAPS between them.  A wave
        // is in-order: while it waits to issue the next MFMA (32 cycles behind the previous one) nothing else of it issues, and while it
        // issues a block of requests its matrix pipe drains -- a request block in front of an MFMA block made a 1024-cycle step take 1600
        // (phase trace, round 5).  So every MFMA is followed by one small piece of request work and a scheduling fence:
        //   * the CB weight fragments of k-step s + 1 (one LDS read per gap; a fragment feeds only RGS MFMAs, an LDS read takes ~130 cycles),
        //   * this wave's share of the weight ring: the 16-byte pieces of W(u + 1), loaded a step ago, are stored to their LDS stage in the
        //     first k-step, those of W(u + 2) requested in the last one (plain loads + ds_write, not LDS-DMA: hipcc's wait-count pass treats
        //     DMA and register loads as unordered event types and answers every wait for a gathered row or a rulebook entry with vmcnt(0)
        //     while a DMA is pending -- a full L2 round trip per step),
        //   * the B operand of step u + 1 (entries ``e_b``, half ``h_b``, into ``bn``): row group g in k-step g mod KS, split into address
        //     arithmetic / LDS reads / the exec-masked gather of the lanes whose neighbour lies outside the window,
        //   * in the last k-step, the rulebook entries of tap ``t_f`` into ``e_f`` (t_f < 0: none).
        auto step = [&](int u, const u32x4(&b)[RGS][KS], const int(&e_b)[RGS], int h_b, u32x4(&bn)[RGS][KS], int t_f, int(&e_f)[RGS]) {
            constexpr int PER = kWPer;
            const unsigned char *wst = s_w + (u % kStages) * SB + lane * 16;
            bf16x8 a[2][CB];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) a[0][cb] = *reinterpret_cast<const bf16x8 *>(wst + cb * 1024);
            constexpr int NM = CB * RGS;  // MFMAs (= gaps) of a k-step
            unsigned g_base[RGS], g_sw[RGS], g_voff[RGS];
            bool g_far[RGS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const int cb = m / RGS, g = m % RGS;
                    acc[g][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s & 1][cb], __builtin_bit_cast(bf16x8, b[g][s]), acc[g][cb], 0, 0, 0);
                    // ---- the piece of request work of gap m
                    if (m < CB && s + 1 < KS) a[(s + 1) & 1][m] = *reinterpret_cast<const bf16x8 *>(wst + ((s + 1) * CB + m) * 1024);
                    // row groups gg with gg mod KS == s (the j-th of them): three consecutive gaps behind the DMA's (arithmetic, LDS reads,
                    // gather) when the k-step has that many, else all three pieces in one gap
                    constexpr int JMAX = (RGS + KS - 1) / KS;
                    constexpr bool SPREAD = NM - CB >= 3 * JMAX;
#pragma unroll
                    for (int gg = s; gg < RGS; gg += KS) {
                        const int j = gg / KS;
                        const int m0 = SPREAD ? CB + 3 * j : (CB + j < NM ? CB + j : NM - 1);
                        const int m1 = SPREAD ? m0 + 1 : m0, m2 = SPREAD ? m0 + 2 : m0;
                        if (m == m0) {
                            const int ev = e_b[gg];
                            const int slot = ev - win0;
                            const bool inwin = ev >= 0 && (unsigned)slot < (unsigned)WIN;
                            g_far[gg] = ev >= 0 && !inwin;
                            g_base[gg] = inwin ? (unsigned)slot * ROWB : (unsigned)(WIN * ROWB);
                            g_sw[gg] = inwin ? (unsigned)((slot / RP) % PIECES) : 0u;
                            g_voff[gg] = (unsigned)ev << kRowShift;
                        }
                        if (m == m1) {
#pragma unroll
                            for (int ss = 0; ss < KS; ++ss) {
                                const unsigned p = (unsigned)(h_b * (S::SC / 8) + 2 * ss + kg);
                                bn[gg][ss] = *reinterpret_cast<const u32x4 *>(s_win + g_base[gg] + ((p ^ g_sw[gg]) << 4));
                            }
                        }
                        if (m == m2) {
                            if (g_far[gg]) {
#pragma unroll
                                for (int ss = 0; ss < KS; ++ss)
                                    bn[gg][ss] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, g_voff[gg] + (unsigned)(h_b * (S::SC / 8) + 2 * ss + kg) * 16u, 0, 0);
                            }
                        }
                    }
                    if (s == KS - 1 && m == 0 && t_f >= 0) fetch_idx(t_f, e_f);
                    // this wave's share of the weight ring: W(u + 1), requested a step ago, goes to its LDS stage in the first k-step (the stage
                    // was last read in step u - 2); W(u + 2) is requested in the last k-step into the same registers
                    if (s == 0) {
#pragma unroll
                        for (int i = 0; i < PER; ++i)
                            if (m == (i < NM ? i : NM - 1)) w_store(u + 1, i, wr);
                    }
                    if (s == KS - 1) {
#pragma unroll
                        for (int i = 0; i < PER; ++i)
                            if (m == (NM - PER + i > 0 ? NM - PER + i : 0) || (NM < PER && m == NM - 1 && i >= NM)) w_load(u + 2, i, wr);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        // Hand-over between steps: this wave's pieces of W(u + 1) are in LDS (ds_write complete), then the workgroup barrier.  No wait on
        // vector memory here: W(u + 2) stays in flight across it (__syncthreads() would add s_waitcnt vmcnt(0)).
        auto hand_over = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

        // (tuning builds: [6] = cycles in the steps, [7] = in the hand-over wait + barrier)
#ifdef FD_WIN_TRACE
#define FD_STEP(ST)            \
    {                          \
        FD_WT(q1);             \
        ST;                    \
        FD_WT(q2);             \
        hand_over();           \
        FD_WT(q3);             \
        FD_WADD(6, q2 - q1);   \
        FD_WADD(7, q3 - q2);   \
    }
#else
#define FD_STEP(ST) \
    {               \
        ST;         \
        hand_over(); \
    }
#endif
        if constexpr (H == 1) {
            // two taps per loop body (static register double buffer); an odd K runs one dummy tap: its entries are all 'missing'
            for (int t = 0; t < K; t += 2) {
                // step t: B(t) in bq[0]; e_nxt = entries of tap t + 1 -> B(t + 1) into bq[1]; entries of tap t + 2 -> e_cur
                FD_STEP(step(t, bq[0], e_nxt, 0, bq[1], t + 2, e_cur));
                mask_idx(t + 2, e_cur);
                // step t + 1: B(t + 1) in bq[1]; e_cur = entries of tap t + 2 -> B(t + 2) into bq[0]; entries of tap t + 3 -> e_nxt
                FD_STEP(step(t + 1, bq[1], e_cur, 0, bq[0], t + 3, e_nxt));
                mask_idx(t + 3, e_nxt);
            }
        } else {
            // one tap per loop body: steps (t, 0) and (t, 1)
            int e_f[RGS];
            for (int t = 0; t < K; ++t) {
                FD_STEP(step(2 * t, bq[0], e_cur, 1, bq[1], -1, e_f));          // B(t, 1) from the current tap's entries
                FD_STEP(step(2 * t + 1, bq[1], e_nxt, 0, bq[0], t + 2, e_f));   // B(t + 1, 0) from the next tap's; entries of tap t + 2
                copy_idx(e_cur, e_nxt);
                copy_idx(e_nxt, e_f);
                mask_idx(t + 2, e_nxt);
            }
        }
#undef FD_STEP
        FD_WT(t_p2);
        FD_WADD(1, t_p2 - t_p1);

        __syncthreads();  // (every wave is through its last step: the ring and the window are free)
        // ---- epilogue through LDS (the window and the ring are free: every wave is behind the last hand-over barrier).  A wave's tile is
        //      32 RGS rows x COUT bf16 in a wave-private region, row pitch COUT * 2 + 16 bytes (conflict-free for both access shapes):
        //      the residual rows arrive by 16-byte row-contiguous loads and are read back as the 8-byte pieces of the accumulator layout
        //      (lane (row ln, half kg), registers 4 j .. 4 j + 3 = channels 32 cb + 8 j + 4 kg .. + 3); the results go the other way and
        //      leave as 16-byte row-contiguous stores (8-byte pieces at a 2 COUT-byte stride straight from the registers cost 18 k of
        //      the 160 k cycles of a workgroup: phase trace, round 5).
        constexpr int PITCH = COUT * 2 + 16, P16 = COUT / 8, WROWS = 32 * RGS;
        constexpr int NIT = WROWS * P16 / 64;  // 16-byte pieces of the wave's tile per lane
        static_assert((size_t)kWaves * WROWS * PITCH <= win_lds_bytes<CIN, COUT, RGS, HALO, NW>(), "the epilogue tiles fit the window + ring");
        unsigned char *s_ep = smem + wave * (WROWS * PITCH);
        if (residual) {
#pragma unroll
            for (int i0 = 0; i0 < NIT; i0 += 8) {
                u32x4 rv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (i0 + i < NIT) {
                        const int q = (i0 + i) * 64 + lane, r = q / P16, pc = q % P16;
                        int row = row0 + r;
                        row = row < wg_r1 ? row : (wg_r1 > 0 ? wg_r1 - 1 : 0);
                        rv[i] = *reinterpret_cast<const u32x4 *>(residual + (int64_t)row * COUT + pc * 8);
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (i0 + i < NIT) {
                        const int q = (i0 + i) * 64 + lane, r = q / P16, pc = q % P16;
                        *reinterpret_cast<u32x4 *>(s_ep + r * PITCH + pc * 16) = rv[i];
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
#pragma unroll
        for (int g = 0; g < RGS; ++g) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned char *sp = s_ep + (32 * g + ln) * PITCH + (32 * cb + 8 * j + 4 * kg) * 2;
                    f32x4 v = (f32x4){acc[g][cb][4 * j], acc[g][cb][4 * j + 1], acc[g][cb][4 * j + 2], acc[g][cb][4 * j + 3]};
                    if (residual) {
                        const bf16x4 rr = *reinterpret_cast<const bf16x4 *>(sp);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] += (float)rr[i];
                    }
                    if (relu) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
                    }
                    bf16x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
                    *reinterpret_cast<bf16x4 *>(sp) = o;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int q = i * 64 + lane, r = q / P16, pc = q % P16;
            const u32x4 v = *reinterpret_cast<const u32x4 *>(s_ep + r * PITCH + pc * 16);
            if (row0 + r < wg_r1) *reinterpret_cast<u32x4 *>(out + (int64_t)(row0 + r) * COUT + pc * 8) = v;
        }
        FD_WT(t_p3);
        FD_WADD(2, t_p3 - t_p2);
    }
#ifdef FD_WIN_TRACE
    FD_WT(t_end);
    if (tid == 0 && g_wintrace) {
        unsigned long long *dst = g_wintrace + (size_t)blockIdx.x * 8;
        wacc[3] = t_end - t_start;
        wacc[4] = (unsigned long long)n_iter;
        for (int i = 0; i < 8; ++i) dst[i] = wacc[i];
    }
#endif
}

struct WinArgs {
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
    int n_in;
    hipStream_t stream;
};

template <int CIN, int COUT, int RGS, int HALO, int NW = 4>
bool win_launch(const WinArgs &a) {
    constexpr int kWaves = NW;
    constexpr size_t lds = win_lds_bytes<CIN, COUT, RGS, HALO, NW>();
    static_assert(lds <= 160 * 1024, "window + weight ring must fit the CU's LDS");
    auto kern = spconv_bf16_win<CIN, COUT, RGS, HALO, NW>;
    static std::atomic<uint64_t> lds_set{0};
    if (lds > 65536 && !fd::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, lds_set)) return false;
    constexpr int TM = kWaves * 32 * RGS;
    int64_t grid = ((int64_t)a.n_out + TM - 1) / TM;
    const int64_t cus = fd::device_cu_count();  // one workgroup per CU (LDS), persistent over its passes
    if (grid > cus) grid = cus;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kWaves * 64), lds, a.stream, (const unsigned short *)a.in, (const unsigned char *)a.wp, a.bias,
                       (const unsigned short *)a.residual, a.relu, a.nbr, a.nbr_stride, a.K, a.n_out, a.n_out_dev, (unsigned short *)a.out, a.in_bytes,
                       a.n_in);
    return true;
}

// row groups per wave: the fewest with which one pass of the resident workgroups covers the level; beyond the largest tile the
// rows are spread evenly over the passes (a last pass with a few rows costs a whole walk over the taps)
inline int pick_rgs(int64_t n_expected, int rgs_max, int kWaves = 4) {
    const int64_t cus = fd::device_cu_count();
    const int64_t per_cu = (n_expected + cus - 1) / cus;
    const int64_t tm_max = (int64_t)kWaves * 32 * rgs_max;
    const int64_t passes = (per_cu + tm_max - 1) / tm_max;
    const int64_t per_pass = (per_cu + (passes > 0 ? passes : 1) - 1) / (passes > 0 ? passes : 1);
    int rgs = (int)((per_pass + kWaves * 32 - 1) / (kWaves * 32));
    return rgs < 1 ? 1 : (rgs > rgs_max ? rgs_max : rgs);
}

}  // namespace

#ifdef FD_WIN_TRACE
extern "C" int fd_debug_set_wintrace(void *p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_wintrace), &p, sizeof(p)) == hipSuccess ? 0 : -1; }
#endif

namespace fd {
// bytes of the window kernels' weight layout for (K, cin, cout); 0 = shape not covered
size_t spconv_bf16_win_weight_bytes(int K, int cin, int cout) {
    if (!((cin == 64 && cout == 64) || (cin == 128 && cout == 128))) return 0;
    return (size_t)K * cin * cout * 2;
}

// [K][H halves][KS k-steps][CB column blocks][lane][8] bf16: lane (m = lane % 32, kg = lane / 32) element j of fragment
// (tap, h, s, cb) = W[tap][input channel SC h + 16 s + 8 kg + j][output channel 32 cb + m] -- the A operand of v_mfma_f32_32x32x16_bf16
void spconv_bf16_win_pack(const float *w, int K, int cin, int cout, uint16_t (*tobf)(float), uint16_t *dst) {
    const int SC = cin < 64 ? cin : 64, H = cin / SC, KS = SC / 16, CB = cout / 32;
    for (int k = 0; k < K; ++k)
        for (int h = 0; h < H; ++h)
            for (int s = 0; s < KS; ++s)
                for (int cb = 0; cb < CB; ++cb)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int ci = SC * h + 16 * s + 8 * (lane >> 5) + j, co = 32 * cb + (lane & 31);
                            dst[((((((int64_t)k * H + h) * KS + s) * CB + cb) * 64) + lane) * 8 + j] = tobf(w[((int64_t)k * cin + ci) * cout + co]);
                        }
}

// returns 1 when launched, 0 when the shape is not covered (the caller takes the RING / RESIDENT kernels)
int spconv_bf16_win_dispatch(const void *in, const void *wp_win, const float *bias, const void *residual, int relu, const int *nbr, int64_t nbr_stride, int K,
                             int64_t n_in_bound, int n_out, const int *n_out_dev, int64_t n_expected, int cin, int cout, void *out, hipStream_t stream) {
    if (n_in_bound * cin * 2 >= (1ll << 31) || n_in_bound < 1 || K < 1 || K > kMaxTaps) return 0;
    WinArgs a{in, wp_win, bias, residual, relu, nbr, nbr_stride, K, n_out, n_out_dev, out, (unsigned)(n_in_bound * cin * 2), (int)n_in_bound, stream};
    const int forced = fd::tuning(fd::kTuneBf16RG);  // 0 = heuristic
    bool ok = false;
    // Eight waves (two per SIMD, 256 registers each, one or two 32-row groups) against four (one per SIMD, up to four groups): with two
    // waves a SIMD's matrix pipe is fed while the other wave sits in an LDS / vector-memory wait -- 59 vs 64 us on 128 -> 128, 64 vs 67-75 on
    // 64 -> 64 (tools/spconv_bench.py, round 5) although every weight fragment read from LDS then feeds half as many MFMAs.
    // "bf16_nw" = 4 selects the four-wave shapes (A/B runs).
    const bool nw4 = fd::tuning(fd::kTuneBf16NW) == 4;
    if (cin == 128 && cout == 128 && !nw4) {
        ok = win_launch<128, 128, 1, 64, 8>(a);
    } else if (cin == 64 && cout == 64 && !nw4) {
        const int rgs = forced > 0 ? (forced > 2 ? 2 : forced) : pick_rgs(n_expected, 2, 8);
        ok = rgs >= 2 ? win_launch<64, 64, 2, 128, 8>(a) : win_launch<64, 64, 1, 128, 8>(a);
    } else if (cin == 128 && cout == 128) {
        const int rgs = forced > 0 ? (forced > 2 ? 2 : forced) : pick_rgs(n_expected, 2);
        ok = rgs >= 2 ? win_launch<128, 128, 2, 64>(a) : win_launch<128, 128, 1, 64>(a);
    } else if (cin == 64 && cout == 64) {
        const int rgs = forced > 0 ? (forced > 4 ? 4 : forced) : pick_rgs(n_expected, 4);  // (5 row groups spill: 160 + 160 accumulator / operand registers)
        switch (rgs) {
            case 1: ok = win_launch<64, 64, 1, 128>(a); break;
            case 2: ok = win_launch<64, 64, 2, 128>(a); break;
            case 3: ok = win_launch<64, 64, 3, 128>(a); break;
            default: ok = win_launch<64, 64, 4, 128>(a); break;
        }
    }
    // (32 -> 32 on this formulation -- a weight ring with a barrier per tap -- measured 66 us against the 42 us of the RESIDENT kernel of
    //  fd_spconv_bf16.hip; a second form with the weight set resident in LDS, a sliding window and no barrier inside a pass was built,
    //  parity-green, and removed: LDS-bound (1.5 KB of operands per 32-cycle MFMA) at the RESIDENT kernel's time -- 45.8 vs 42.3 us, two
    //  clouds 70 vs 70; profiles/round5_bf16win32_negative.txt)
    return ok ? 1 : 0;
}
}  // namespace fd
