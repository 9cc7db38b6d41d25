// fp32 sparse convolution, second formulation: tile-level pair compaction.
//
// fp32 MFMA (v_mfma_f32_16x16x4_f32) runs at the fp32 vector rate, so the fp32 layers are MFMA-bound and
// every zero row fed to the matrix core is lost time.  With 27 taps only 18-63 % of the (output row, tap)
// pairs exist, and a 16-row output group almost never lacks a tap entirely, so the register-resident
// output-stationary kernel (fd_spconv.hip) spends 40-80 % of its MFMAs on zeros.  This kernel removes them:
//
//   * a workgroup owns one RANGE of consecutive output rows (spatially sorted, so their inputs are close) and walks it
//     in chunks of at most TM = 128 rows.  Ranges are either equal row counts or, for rulebooks shared by several
//     convolutions, equal WORK (fd_spconv_ranges: MFMA groups per 32-row block, prefix sum, quantiles), and there is a
//     whole number of them per compute unit: no tail round, no heavy / light tile lottery, and -- unlike a work-sorted
//     tile permutation -- neighbouring workgroups still gather neighbouring input rows.  While a chunk's MFMAs run,
//     the next chunk's rulebook slice is already on its way into registers;
//   * the rulebook tile [K][TM] is staged in LDS and compacted IN PLACE per tap with wave ballots /
//     prefix popcounts into lists of (input row << 8 | local output row) entries, tails filled with -1 --
//     the "LDS-staged rulebook tile";
//   * accumulators for the whole chunk live in LDS ([TM + 1][COUT] fp32, 64 KB at COUT = 128; row TM is a
//     scratch row that absorbs the padding lanes so the accumulator traffic needs no exec masking).  The MFMA is
//     issued TRANSPOSED (A operand = weight fragment, B operand = gathered rows): lane (pair j, quad q) then holds four
//     consecutive output channels of pair j's row, so one 16-byte LDS read and one 16-byte write per 16-column block
//     move the accumulators (the untransposed layout needs four 4-byte accesses each way and four row addresses);
//     16-byte slots are XOR-swizzled by the row so the 16 rows of a lane group land on distinct banks;
//   * each wave owns a column slice of the tile (and, for narrow COUT, a row subset) and walks a flattened
//     work list of (tap, 16-pair group) items: gather the 16 input rows straight into MFMA A-fragment layout
//     (one 16-byte bounds-checked buffer load per lane and 16-channel chunk, prefetched DEPTH items ahead with
//     exact vmcnt accounting), read the 16 accumulator rows from LDS as the MFMA C operand, run the
//     CIN/4 x NBW MFMAs, write D back.  The wave's slice of W[tap] stays in registers for the whole tap
//     (double buffered one tap ahead when it is small).  A wave never shares an accumulator element with another
//     wave: no atomics, no barriers in the main loop, fixed summation order (taps ascending) -> deterministic.
//   * epilogue: bias (+ residual) (+ ReLU) on the LDS tile, written out with 16-byte row-contiguous stores.
#include "fd_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxTaps = 27;

// Phase timeline for tuning builds only (tools/probes/build_trace.sh compiles this file with -DFD_V2_TRACE into a
// separate library; the product library contains none of it): thread 0 of every workgroup stamps s_memtime at the
// phase boundaries of its first chunk.
#ifdef FD_V2_TRACE
__device__ unsigned long long *g_trace;
#define FD_T(i)                                                                                  \
    do {                                                                                         \
        if (threadIdx.x == 0 && g_trace) g_trace[(size_t)blockIdx.x * 16 + (i)] = __builtin_readcyclecounter(); \
    } while (0)
#define FD_TV(i, v)                                                                    \
    do {                                                                               \
        if (threadIdx.x == 0 && g_trace) g_trace[(size_t)blockIdx.x * 16 + (i)] = (v); \
    } while (0)
#else
#define FD_T(i)
#define FD_TV(i, v)
#endif


template <int CIN, int COUT, int TM, int DEPTH>
__global__ void __launch_bounds__(256) spconv_f32_compact(const float *__restrict__ in, const float4 *__restrict__ wp,
                                                          const float *__restrict__ bias, const float *__restrict__ residual, int relu,
                                                          const int *__restrict__ nbr, int64_t nbr_stride, int K, int n_out,
                                                          const int *__restrict__ n_out_dev, float *__restrict__ out, unsigned in_bytes,
                                                          const int *__restrict__ ranges, int rows_per_range) {
    constexpr int NB = COUT / 16, NC = CIN / 16;
    constexpr int WC = NB == 8 ? 4 : (NB >= 2 ? 2 : 1);  // column splits across the 4 waves (64 columns: 2 x 32)
    constexpr int NBW = NB / WC;          // 16-column blocks per wave
    // 32 / 64 columns: instead of two row halves (whose separately compacted lists pad 17 % more MFMA rows) the two waves of a
    // column slice split the TAPS (even / odd) over the full tile and accumulate into private copies of the tile that
    // the epilogue adds up; the LDS for the second copy is there anyway (these layers run two workgroups per CU).
    constexpr int TS = (NB == 2 || NB == 4) ? 2 : 1;  // tap splits (16 columns keep four row quarters: measured faster)
    constexpr int WR = 4 / (WC * TS);             // row splits
    constexpr int RW = TM / WR;                   // rows in a wave's row set
    constexpr int NACC = NBW;  // one accumulator chain per column block (dependent 16x16x4 MFMAs issue back to back at full rate)
    static_assert((TM == 128 || TM == 64) && RW >= 16, "local row uses 8 bits (0..TM, TM = scratch row)");
    constexpr int kMaxItems = kMaxTaps * (TM / 16);           // per wave: taps x 16-row groups
    constexpr int kPad = (int)(0xffffff00u | (unsigned)TM);  // list padding: input offset out of range, local row = TM (scratch row)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *s_list = reinterpret_cast<int *>(smem);                                            // [K][TM] raw nbr, then compacted entries
    unsigned short *s_items = reinterpret_cast<unsigned short *>(s_list + kMaxTaps * TM);   // [4 waves][kMaxItems]
    unsigned char *s_cnt = reinterpret_cast<unsigned char *>(s_items + 4 * kMaxItems);      // [K][4] (<= 128 each)
    int *s_pad = reinterpret_cast<int *>(s_cnt + 112);                                      // 16 padding entries (tail of the work list)
    float *s_acc = reinterpret_cast<float *>(s_pad + 16);                                   // [TM + 1][COUT], 16-byte aligned for TM = 64 and 128

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: what derives from it stays in SGPRs
    // ---- this workgroup's row range, cut into equal chunks of at most TM rows (multiples of 16)
    int r_begin, r_end;
    if (n_out_dev) n_out = fd::device_count(n_out, n_out_dev);  // capacity launch (see fd_common.h)
    if (ranges) {
        r_begin = ranges[blockIdx.x];
        r_end = ranges[blockIdx.x + 1];
    } else {
        if (n_out_dev)  // the equal-rows split is made here, from the device's count
            rows_per_range = (((n_out + (int)gridDim.x - 1) / (int)gridDim.x) + 15) & ~15;
        const int64_t b = (int64_t)blockIdx.x * rows_per_range;
        r_begin = (int)(b < n_out ? b : n_out);
        r_end = (int)(b + rows_per_range < n_out ? b + rows_per_range : n_out);
    }
    if (r_end > n_out) r_end = n_out;
    if (r_begin >= r_end) return;
    FD_T(0);
    const int n_chunks = (r_end - r_begin + TM - 1) / TM;
    const int chunk_rows = (((r_end - r_begin + n_chunks - 1) / n_chunks) + 15) & ~15;
    // rulebook slice of a chunk, one register per 256 entries; loads are branch-free (clamped address, select on use)
    constexpr int NPRE = (kMaxTaps * TM + 255) / 256;
    int pre[NPRE];
    auto fetch_slice = [&](int row0) {
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int t = tid + i * 256;
            int k = t / TM;
            const int r = t - k * TM;
            k = k < K ? k : K - 1;
            int64_t o = (int64_t)row0 + r;
            o = o < nbr_stride ? o : nbr_stride - 1;
            pre[i] = nbr[(int64_t)k * nbr_stride + o];
        }
    };
    fetch_slice(r_begin);

    const int lrow = lane & 15, lq = lane >> 4;
    const int wc = wave % WC, wr = TS == 1 ? wave / WC : 0, ts = TS == 1 ? 0 : wave / WC;
    const int cb = wc * NBW * 16;
    constexpr int kRowShift = COUT == 16 ? 6 : COUT == 32 ? 7 : COUT == 64 ? 8 : 9;  // log2(COUT * 4)
    static_assert((COUT * 4) == (1 << kRowShift), "COUT must be 16, 32, 64 or 128");
    // 16-byte slot swizzle of the accumulator tile: slot ^= f(row) so that 16 different rows at one column slot spread
    // over the 16 slots of a 256-byte bank row (COUT 64/128: row & 15; 32: two rows per bank row; 16: four)
    constexpr int kSwzShift = COUT >= 64 ? 0 : COUT == 32 ? 1 : 2;
    constexpr unsigned kSwzMask = COUT >= 64 ? 15u : COUT == 32 ? 7u : 3u;
    unsigned char *acc_bytes = reinterpret_cast<unsigned char *>(s_acc + ts * (TM + 1) * COUT);  // this wave's tile copy
    const unsigned slot0 = (unsigned)(cb >> 2) + (unsigned)lq;  // 16-byte slot of this lane's 4 channels in block nw = 0
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (int)in_bytes, 0x00020000);
    unsigned short *items = s_items + wave * kMaxItems;

    for (int chunk = 0; chunk < n_chunks; ++chunk) {
    const int row0 = r_begin + chunk * chunk_rows;
    const int n_rows = (r_end - row0) < chunk_rows ? (r_end - row0) : chunk_rows;
    if (n_rows <= 0) break;
    // ---- stage the prefetched slice, clear the accumulators
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
        const int t = tid + i * 256;
        const int r = t % TM;
        if (t < K * TM) s_list[t] = r < n_rows ? pre[i] : -1;
    }
    // Accumulator tile.  For COUT <= 64 the tile starts from bias + residual instead of zero: the epilogue then has no global
    // load left (it used to issue one dependent residual load per 256 rows x 4 channels -- 4 to 8 exposed memory round trips per
    // chunk, as long as the chunk's whole MFMA loop on the 32-channel layers: the "unexplained" 47 % dependency stall of round 2).
    // All loads of a chunk go out back to back here and land during the list staging.  (128 columns: 64 registers per thread
    // would be needed; that layer is MFMA-bound and keeps the epilogue form.)
    constexpr bool kInitAcc = COUT <= 64;
    if constexpr (kInitAcc) {
        constexpr int C4i = COUT / 4, NINIT = TM * C4i / 256;
        static_assert(TM * C4i % 256 == 0, "whole passes");
        float4 iv[NINIT];
#pragma unroll
        for (int i = 0; i < NINIT; ++i) {
            const int t = tid + i * 256, c4 = t % C4i;
            iv[i] = bias ? reinterpret_cast<const float4 *>(bias)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (residual) {
            float4 rv[NINIT];
#pragma unroll
            for (int i = 0; i < NINIT; ++i) {
                const int t = tid + i * 256, r = t / C4i, c4 = t - r * C4i;
                const int rr = r < n_rows ? r : n_rows - 1;  // (clamped address, selected on use)
                rv[i] = reinterpret_cast<const float4 *>(residual + (int64_t)(row0 + rr) * COUT)[c4];
            }
#pragma unroll
            for (int i = 0; i < NINIT; ++i) { iv[i].x += rv[i].x; iv[i].y += rv[i].y; iv[i].z += rv[i].z; iv[i].w += rv[i].w; }
        }
#pragma unroll
        for (int i = 0; i < NINIT; ++i) {
            const int t = tid + i * 256, r = t / C4i, c4 = t - r * C4i;
            const int ts4 = r * C4i + (c4 ^ (int)(((unsigned)r >> kSwzShift) & kSwzMask));
            reinterpret_cast<float4 *>(s_acc)[ts4] = r < n_rows ? iv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // the scratch row of copy 0 and the other tile copies start from zero
        for (int t = TM * C4i + tid; t < TS * (TM + 1) * COUT / 4; t += 256) reinterpret_cast<float4 *>(s_acc)[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        for (int t = tid; t < TS * (TM + 1) * COUT / 4; t += 256) reinterpret_cast<float4 *>(s_acc)[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid < 16) s_pad[tid] = kPad;
    __syncthreads();
    if (chunk == 0) FD_T(1);
    // ---- in-place compaction: wave w takes taps w, w+4, ...; tails are filled with kPad
    for (int k = wave; k < K; k += 4)
    {
#pragma unroll
        for (int wr = 0; wr < WR; ++wr) {
            const int base = k * TM + wr * RW;
            int count = 0;
            int v[(RW + 63) / 64];
#pragma unroll
            for (int h = 0; h < (RW + 63) / 64; ++h) {
                const int r = h * 64 + lane;
                v[h] = (r < RW) ? s_list[base + r] : -1;
            }
#pragma unroll
            for (int h = 0; h < (RW + 63) / 64; ++h) {
                const int r = h * 64 + lane;
                if (r < RW) s_list[base + r] = kPad;
            }
#pragma unroll
            for (int h = 0; h < (RW + 63) / 64; ++h) {
                const int r = h * 64 + lane;
                const unsigned long long m = __ballot(v[h] >= 0);
                const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
                if (v[h] >= 0) s_list[base + pos] = (v[h] << 8) | (wr * RW + r);
                count += __popcll(m);
            }
            if (lane == 0) s_cnt[k * 4 + wr] = (unsigned char)count;
        }
    }
    __syncthreads();
    if (chunk == 0) FD_T(2);
    // the next chunk's slice travels while this chunk computes
    if (chunk + 1 < n_chunks) fetch_slice(row0 + chunk_rows);

    // ---- flattened work list of this wave: one item = 16 compacted pairs of one tap, code = (tap << 3) | group
    int n_items;
    unsigned long long tapmask;
    {
        const int ng = (lane < K && (lane % TS) == ts) ? ((int)s_cnt[lane * 4 + wr] + 15) >> 4 : 0;
        tapmask = __ballot(ng > 0);
        int inc = ng;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int u = __shfl_up(inc, off);
            if (lane >= off) inc += u;
        }
        n_items = __builtin_amdgcn_readfirstlane(__shfl(inc, 63));
        for (int g = 0; g < ng; ++g) items[inc - ng + g] = (unsigned short)((lane << 3) | g);
    }
    // wave-local LDS hand-off (items written above are read below by other lanes of the same wave)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    // ring of DEPTH prefetched items.  Gathers are buffer loads with hardware bounds checking: a padding lane / a
    // slot past the end of the work list gets an out-of-range offset and reads zeros, so prefetch AND compute are
    // branch-free (exec-masked loads forced vmcnt(0), i.e. no overlap at all; branches around the MFMAs kept the
    // compiler from interleaving the bookkeeping VALU/LDS work with them).
    int k_r[DEPTH];
    int row_r[DEPTH];  // list entry of pair `lrow` of the item: input row << 8 | local output row
    u32x4 a_r[DEPTH][NC];
    // Bookkeeping is kept off the vector ALU (it competes with the MFMAs for issue slots): the item code is
    // wave-uniform (scalar registers), and a list entry needs no compare/select -- the padding entry kPad has all
    // high bits set, so its byte offset lands beyond the buffer (hardware returns zeros) and its row field is the
    // scratch row TM.
    // Bookkeeping is software pipelined so that no LDS round trip sits between two MFMA blocks of a wave (in-order
    // issue: a dependent ds_read chain costs ~1.3k cycles per item under load, as much as the gather itself):
    //   stage A0 (item i+DEPTH+1): read the item code          -> consumed one iteration later
    //   stage A1 (item i+DEPTH)  : read its 16 list entries     -> consumed one iteration later
    //   stage B  (item i+DEPTH-1): issue its gather (buffer loads; a padding entry / a slot past the end of the work list
    //                              has an out-of-range offset: the hardware returns zeros, no branch, no exec mask)
    //   item i                   : request the old accumulator values, run the MFMAs from a zero accumulator, then add
    //                              and write back -- nothing waits in front of the MFMAs.  (LDS float atomics would
    //                              drop the read altogether but ds_add_f32 measured 2-4x slower for the whole kernel.)
    // A wave owns its accumulator elements and LDS operations of a wave execute in order, so the sums are formed in a
    // fixed order (taps ascending): deterministic.
    auto stage_a0 = [&](int it) -> int { return (int)items[it < n_items ? it : 0]; };
    auto stage_a1 = [&](int it, int code_v, int &kk, int &e) {
        const bool v = it < n_items;  // uniform (n_items is in a scalar register)
        const int code = __builtin_amdgcn_readfirstlane(code_v);
        const int ks = v ? (code >> 3) : 0;
        kk = v ? ks : -1;
        // past the end of the work list the (scalar) list pointer selects a block of 16 padding entries: no per-lane select
        const int *lst = v ? s_list + ks * TM + wr * RW + ((code & 7) << 4) : s_pad;
        e = lst[lrow];
    };
    auto gather_offset = [&](int e) -> unsigned {
        // byte offset of the input row = (e >> 8) * CIN * 4, computed on the masked entry without a multiply
        const unsigned hi = (unsigned)e & 0xffffff00u;
        return (CIN >= 64 ? hi << (CIN == 128 ? 1 : 0) : hi >> (CIN == 32 ? 1 : 2)) + (unsigned)(lq * 16);
    };
    auto gather_chunk = [&](unsigned voff, int c) { return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + c * 64, 0, 0); };
    auto load_b = [&](int k, float4(&dst)[NC][NBW]) {
        const float4 *wk = wp + ((int64_t)k * NC * NB + wc * NBW) * 64 + lane;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int nw = 0; nw < NBW; ++nw) dst[c][nw] = wk[(c * NB + nw) * 64];
    };
    // One weight slice in registers.  It is replaced IN PLACE during the last item of a tap: as soon as the MFMAs of a
    // 16-channel chunk have consumed b[c], the same registers are reloaded with the next tap's chunk, so the weights
    // of the next tap arrive under the matrix work of the current one without a second buffer.  (A double buffer
    // swapped at the tap switch costs hipcc a full register copy plus an s_waitcnt vmcnt(0) per switch -- the loop-
    // carried swap cannot be allocated copy-free -- which also drained the gather prefetch ring one item in five.)
    float4 b[NC][NBW];
    static_assert(DEPTH >= 2, "the slot freed by the previous item is refilled during the current item's MFMAs");
    int k_s, e_s, code_s;
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) {
        stage_a1(d, stage_a0(d), k_s, e_s);
        const unsigned vo = gather_offset(e_s);
#pragma unroll
        for (int c = 0; c < NC; ++c) a_r[d][c] = gather_chunk(vo, c);
        k_r[d] = k_s;
        row_r[d] = e_s;
    }
    k_r[DEPTH - 1] = -1;
    row_r[DEPTH - 1] = kPad;
    stage_a1(DEPTH - 1, stage_a0(DEPTH - 1), k_s, e_s);  // staged for the first loop iteration
    code_s = stage_a0(DEPTH);
    if (n_items > 0) load_b(k_r[0], b);  // weights of the first tap (the only exposed weight latency of the chunk)
    if (chunk == 0) { FD_T(3); FD_TV(8, (unsigned long long)n_items); FD_TV(9, (unsigned long long)n_chunks); FD_TV(10, (unsigned long long)n_rows);
                      FD_TV(11, (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11))); FD_TV(12, (unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11))); }
    FD_TV(13, (unsigned long long)(g_trace ? g_trace[(size_t)blockIdx.x * 16 + 13] : 0) + (unsigned long long)n_items);

    // transposed product: A operand = weight fragment (row index = output channel), B operand = gathered rows (column
    // index = pair), so acc[nw] of lane (pair lrow, quad lq) = out[pair][cb + 16 nw + 4 lq .. + 3]
    auto mfma_chunk = [&](const u32x4 &a4, const float4(&bc)[NBW], f32x4(&acc)[NACC]) {
        const float4 av = __builtin_bit_cast(float4, a4);
#pragma unroll
        for (int nw = 0; nw < NBW; ++nw) acc[nw] = __builtin_amdgcn_mfma_f32_16x16x4f32(bc[nw].x, av.x, acc[nw], 0, 0, 0);
#pragma unroll
        for (int nw = 0; nw < NBW; ++nw) acc[nw] = __builtin_amdgcn_mfma_f32_16x16x4f32(bc[nw].y, av.y, acc[nw], 0, 0, 0);
#pragma unroll
        for (int nw = 0; nw < NBW; ++nw) acc[nw] = __builtin_amdgcn_mfma_f32_16x16x4f32(bc[nw].z, av.z, acc[nw], 0, 0, 0);
#pragma unroll
        for (int nw = 0; nw < NBW; ++nw) acc[nw] = __builtin_amdgcn_mfma_f32_16x16x4f32(bc[nw].w, av.w, acc[nw], 0, 0, 0);
    };

    for (int i0 = 0; i0 < n_items; i0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            // accumulator row of this lane's pair (the row field of a padding entry is the scratch row TM): one swizzled
            // 16-byte slot per 16-column block
            const unsigned arow = (unsigned)row_r[d] & 255u;
            const unsigned abase = arow << kRowShift, aswz = (arow >> kSwzShift) & kSwzMask;
            unsigned aoff[NBW];
#pragma unroll
            for (int nw = 0; nw < NBW; ++nw) aoff[nw] = abase + (((slot0 + 4u * nw) ^ aswz) << 4);
            // The old accumulator values are requested first and are the C operand of the MFMA chain: the matrix pipe
            // does the accumulation and D goes back to LDS untouched -- no vector-ALU work on the accumulators at all
            // (VALU instructions of one wave barely overlap the MFMAs of the other waves of its SIMD, so every VALU
            // instruction saved per item is matrix-pipe time gained).  The bookkeeping below covers the LDS latency.
            f32x4 acc[NACC];
#pragma unroll
            for (int nw = 0; nw < NBW; ++nw) acc[nw] = *reinterpret_cast<const f32x4 *>(acc_bytes + aoff[nw]);
            // refill the slot freed by the previous item BEFORE this item's MFMAs (left to itself hipcc sinks the loads
            // below the MFMA block and waits vmcnt(0) for them at the top of the next item), then advance the two
            // bookkeeping stages; their results are first touched in the next iteration.
            const int dn = (d + DEPTH - 1) % DEPTH;  // compile-time after unrolling
            unsigned vo;
            {
                const int it = i0 + d + DEPTH - 1;
                vo = gather_offset(e_s);
                k_r[dn] = k_s;
                row_r[dn] = e_s;
                stage_a1(it + 1, code_s, k_s, e_s);
                code_s = stage_a0(it + 2);
            }
            const int knext = k_r[(d + 1) % DEPTH];                 // tap of the next item (-1 past the end), wave-uniform
            const bool reload = knext >= 0 && knext != k_r[d];      // this is the last item of its tap
            // narrow layers (fewer than 16 MFMAs per item) have too little matrix work between the loads for the
            // interleaving to pay: their gather goes out in one piece in front of the MFMAs
            constexpr bool kSpread = NC * 4 * NBW >= 16;
            if constexpr (!kSpread) {
#pragma unroll
                for (int c = 0; c < NC; ++c) a_r[dn][c] = gather_chunk(vo, c);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (reload) {
                const float4 *wk = wp + ((int64_t)knext * NC * NB + wc * NBW) * 64 + lane;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    mfma_chunk(a_r[d][c], b[c], acc);
                    if constexpr (kSpread) a_r[dn][c] = gather_chunk(vo, c);  // the gather of item i+DEPTH-1 is spread between the MFMAs
#pragma unroll
                    for (int nw = 0; nw < NBW; ++nw) b[c][nw] = wk[(c * NB + nw) * 64];
                    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NBW, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, (kSpread ? 1 : 0) + NBW, 0);
                }
            } else {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    mfma_chunk(a_r[d][c], b[c], acc);
                    if constexpr (kSpread) {
                        a_r[dn][c] = gather_chunk(vo, c);
                        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NBW, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
            }
#pragma unroll
            for (int nw = 0; nw < NBW; ++nw) *reinterpret_cast<f32x4 *>(acc_bytes + aoff[nw]) = acc[nw];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (chunk == 0) FD_T(4);
    __syncthreads();
    if (chunk == 0) FD_T(5);
    // ---- epilogue: whole chunk, float4 per thread, rows contiguous in global memory (swizzled slots in LDS)
    constexpr int C4 = COUT / 4;
    for (int t = tid; t < n_rows * C4; t += 256) {
        const int r = t / C4, c4 = t - r * C4;
        const int row = row0 + r;
        const int ts4 = r * C4 + (c4 ^ (int)(((unsigned)r >> kSwzShift) & kSwzMask));
        float4 v = reinterpret_cast<const float4 *>(s_acc)[ts4];
#pragma unroll
        for (int q = 1; q < TS; ++q) {
            const float4 v2 = reinterpret_cast<const float4 *>(s_acc + q * (TM + 1) * COUT)[ts4];
            v.x += v2.x; v.y += v2.y; v.z += v2.z; v.w += v2.w;
        }
        if constexpr (!kInitAcc) {
            if (bias) {
                const float4 bv = reinterpret_cast<const float4 *>(bias)[c4];
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            }
            if (residual) {
                const float4 rv = reinterpret_cast<const float4 *>(residual + (int64_t)row * COUT)[c4];
                v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
            }
        }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        reinterpret_cast<float4 *>(out + (int64_t)row * COUT)[c4] = v;
    }
    __syncthreads();  // the next chunk re-uses the list and the accumulator tile
    if (chunk == 0) FD_T(6);
    }  // chunk loop
    FD_T(7);
}

constexpr size_t lds_bytes(int tm, int cout) {
    return sizeof(int) * kMaxTaps * tm + sizeof(unsigned short) * 4 * kMaxTaps * (tm / 16) + 112 + 64 +
           sizeof(float) * (tm + 1) * cout * (cout == 32 || cout == 64 ? 2 : 1);
}

// LDS request of one workgroup.  Occupancy is not a lever here: MFMA and non-MFMA instructions of the waves sharing a
// SIMD execute almost serially (128 channels: one workgroup per CU is only 9 % slower than two), and for the 64->64
// layers two workgroups per CU beat the three that would fit (300 -> 278 us: less contention in the gather path), so
// their request is rounded up to just over a third of the CU's 160 KB.
inline size_t lds_request(int cin, int cout, int tm) {
    const size_t lds = lds_bytes(tm, cout);
    const size_t pad = (size_t)fd::tuning(fd::kTuneV2LdsPad);  // occupancy experiments
    if (pad) return lds + pad;
    return (cin == 64 && cout == 64 && tm == 128 && lds < 56 * 1024) ? (size_t)56 * 1024 : lds;
}

template <int CIN, int COUT, int TM, int DEPTH>
int launch_compact(const float *in, const void *wp, const float *bias, const float *residual, int relu, const int *nbr,
                   int64_t nbr_stride, int K, int n_out, const int *n_out_dev, float *out, unsigned in_bytes, const int *ranges, int n_ranges,
                   hipStream_t stream) {
    const size_t lds_req = lds_request(CIN, COUT, TM);
    static std::atomic<uint64_t> lds_set{0};  // devices on which this instantiation has its LDS limit raised
    auto kern = spconv_f32_compact<CIN, COUT, TM, DEPTH>;
    if (fd::tuning(fd::kTuneV2LdsPad)) {  // tuning runs change the request between calls: set it every time
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_req) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
    } else if (!fd::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds_req, lds_set)) {
        return 0;
    }
    int rows_per = 0;
    if (!ranges) {  // equal row counts (multiples of 16): n_ranges == 0 -> one TM-row tile per workgroup
        if (n_ranges <= 0) n_ranges = (n_out + TM - 1) / TM;
        rows_per = (((n_out + n_ranges - 1) / n_ranges) + 15) & ~15;
        if (!n_out_dev) n_ranges = (n_out + rows_per - 1) / rows_per;  // (with a device count the kernel makes the split over n_ranges)
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)n_ranges), dim3(256), lds_req, stream, in, (const float4 *)wp, bias, residual, relu, nbr, nbr_stride, K,
                       n_out, n_out_dev, out, in_bytes, ranges, rows_per);
    return 1;
}

// ---------------------------------------------------------------------------------------------- work-balanced ranges
// work of an 8-row block in 1/32 of a 16-pair MFMA group: 2 per pair, + 1 per (block, tap) that has pairs (its share of
// the list padding: half a group per tap and 128-row chunk), + kRowCost for the per-row cost of prologue / compaction /
// epilogue: the phase trace puts those at ~10 groups per 128-row chunk = 20 units per block (swept 4 ... 48: 20-32 is the
// flat optimum, 128->128 -2.5 %, 64->64 -1 % against the 4 of the first version, which let sparse regions grow ranges past
// 128 rows, i.e. into a second chunk with its own padding and prologue).  An iterative refinement with the kernel's exact
// group count per range was tried and is worse (98 ... 156 groups per wave instead of 120 ... 151): the cost of a range
// jumps by ~23 groups when it crosses 128 rows, which a boundary interpolation cannot follow; with the ranges capped at
// 128 rows (one chunk each) the refinement converges but the layer time does not move (348 vs 352 us): the group count per
// workgroup is not what sets the kernel's duration.
// One wave covers 64 rows = 8 blocks with one ballot per tap; lane i < 8 accumulates block i.
constexpr int kWorkRows = 8;
constexpr int kRowCost = 24;
__global__ void __launch_bounds__(256) block_work_kernel(const int *__restrict__ nbr, int64_t nbr_stride, int K, int n_out,
                                                         const int *__restrict__ n_out_dev, unsigned row_cost, unsigned *__restrict__ work) {
    n_out = fd::device_count(n_out, n_out_dev);
    const int n_blocks = (n_out + kWorkRows - 1) / kWorkRows;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t o = wave * 64 + lane;
    if (wave * 8 >= n_blocks) return;
    unsigned w = row_cost;
    for (int k = 0; k < K; ++k) {
        const bool v = o < n_out && nbr[(int64_t)k * nbr_stride + o] >= 0;
        const unsigned long long m = __ballot(v);
        const unsigned byte = (unsigned)(m >> (8 * (lane & 7))) & 0xffu;
        w += (unsigned)__popc(byte) * 2u + (byte ? 1u : 0u);
    }
    if (lane < 8 && wave * 8 + lane < n_blocks) work[wave * 8 + lane] = w;
}

// single workgroup: prefix over the block works, range j = blocks whose work midpoint falls into the j-th of n_ranges
// equal slices of the total.  Boundaries are multiples of 8 rows; a range may be empty (one block heavier than a slice).
// Round 6: every pass over the works is coalesced -- the 16 waves own contiguous segments and walk them 64 blocks at a time with a
// shuffle scan (the round 2-5 form gave each of the 1024 threads ~40 CONSECUTIVE blocks: 40 strided load instructions per thread, twice,
// between two 1024-wide Hillis-Steele scans; 28 us per launch for a 150-KB table).  Same table bit for bit.
__global__ void __launch_bounds__(1024) range_split_kernel(const unsigned *__restrict__ work, int n_out, const int *__restrict__ n_out_dev,
                                                           int n_ranges, int *__restrict__ ranges) {
    n_out = fd::device_count(n_out, n_out_dev);
    const int n_blocks = (n_out + kWorkRows - 1) / kWorkRows;
    __shared__ unsigned long long s_seg[16];
    __shared__ unsigned long long s_total;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (n_blocks == 0) {
        for (int j = tid; j <= n_ranges; j += 1024) ranges[j] = 0;
        return;
    }
    const int seg = (((n_blocks + 15) / 16) + 63) & ~63;  // blocks per wave (whole 64-block steps)
    const int b_lo = wave * seg, b_hi = min(n_blocks, b_lo + seg);
    // ---- segment sums (coalesced), their exclusive scan, the total
    constexpr int kU = 16;  // loads in flight per lane (a dependent load per 64 blocks costs a memory round trip each: 38 of them per pass before)
    unsigned long long sum = 0;
    for (int c0 = b_lo; c0 < b_hi; c0 += 64 * kU) {
        unsigned wv[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int bb = c0 + 64 * u + lane;
            wv[u] = work[bb < b_hi ? bb : b_hi - 1];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) sum += (c0 + 64 * u + lane < b_hi) ? wv[u] : 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) s_seg[wave] = sum;
    __syncthreads();
    if (tid == 0) {
        unsigned long long run = 0;
        for (int w = 0; w < 16; ++w) { const unsigned long long v = s_seg[w]; s_seg[w] = run; run += v; }
        s_total = run;
    }
    __syncthreads();
    const unsigned long long total = s_total > 0 ? s_total : 1ull;
    // range of a block = floor(midpoint * n_ranges / total), as one double multiply (monotone in the midpoint, the same function in
    // every thread: all the table needs -- results do not depend on the ranges, tests/test_gpu_parity.py)
    const double scale = (double)n_ranges / (2.0 * (double)total);
    auto range_of = [&](unsigned long long ex, unsigned w) -> int {
        const int r = (int)((double)(2ull * ex + w) * scale);
        return r < n_ranges ? r : n_ranges - 1;
    };
    unsigned long long run = s_seg[wave];  // exclusive prefix of this wave's first block
    int prev = -1;                         // range of the block in front of the current one (-1 in front of block 0)
    if (b_lo > 0 && b_lo < n_blocks) {
        const unsigned wp = work[b_lo - 1];
        prev = range_of(run - wp, wp);
    }
    // 512 blocks per step: loaded coalesced into this wave's LDS slice, then lane l walks blocks 8 l .. 8 l + 7 on its own -- one shuffle scan
    // (of the lanes' 8-block sums) per 512 blocks instead of one per 64
    constexpr int kB = 8;
    __shared__ unsigned s_buf[16][64 * (kB + 1)];
    unsigned *buf = s_buf[wave];
    for (int c0 = b_lo; c0 < b_hi; c0 += 64 * kB) {
#pragma unroll
        for (int u = 0; u < kB; ++u) {
            const int e = 64 * u + lane, bb = c0 + e;
            buf[(e / kB) * (kB + 1) + e % kB] = bb < b_hi ? work[bb] : 0u;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        unsigned w[kB];
        unsigned tot = 0;
#pragma unroll
        for (int i = 0; i < kB; ++i) {
            w[i] = buf[lane * (kB + 1) + i];
            tot += w[i];
        }
        unsigned inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        unsigned long long ex = run + (inc - tot);
        // the range of the block in front of this lane's first one: the previous lane's last block (lane 0: the carry)
        const int first = c0 + lane * kB;
        int last_rg = prev;
        {
            unsigned long long e2 = ex;
#pragma unroll
            for (int i = 0; i < kB; ++i) {
                if (first + i < b_hi) last_rg = range_of(e2, w[i]);
                e2 += w[i];
            }
        }
        int before = __shfl_up(last_rg, 1);
        if (lane == 0) before = prev;
        // (a lane whose blocks all lie past the segment end keeps the carry: its last_rg is its predecessor's, handed on below)
        if (first >= b_hi) last_rg = before;
#pragma unroll
        for (int i = 0; i < kB; ++i) {
            const int bb = first + i;
            if (bb < b_hi) {
                const int rg = range_of(ex, w[i]);
                for (int j = before + 1; j <= rg; ++j) ranges[j] = bb * kWorkRows;
                before = rg;
                if (bb == n_blocks - 1)  // the thread owning the last block closes the table
                    for (int j = rg + 1; j <= n_ranges; ++j) ranges[j] = n_out;
            }
            ex += w[i];
        }
        // carry into the next step: the range of the step's last valid block, the step's total
        const int n_valid = min(64 * kB, b_hi - c0);
        prev = __shfl(last_rg, (n_valid - 1) / kB);
        run += __shfl(inc, 63);
        __builtin_amdgcn_wave_barrier();  // (the slice is rewritten by the next step)
    }
}

}  // namespace

#ifdef FD_V2_TRACE
extern "C" int fd_debug_set_trace(void *p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &p, sizeof(p)) == hipSuccess ? 0 : -1; }
#endif

// How many ranges fd_spconv_apply wants for a layer shape (measured on MI355X, 300k-point cloud, tools/spconv_bench.py):
//   * Cout >= 64 (MFMA-bound, two workgroups resident per CU): exactly one range per resident workgroup slot, equal
//     WORK -- every CU gets the same number of MFMA groups and finishes together (64->64: 229 -> 200 us, 128->128:
//     320 -> 308 us); more, shorter ranges lose that (greedy dispatch leaves up to one range of imbalance per CU);
//   * narrower layers are bound by the per-row skeleton (prologue / compaction / epilogue), three to six workgroups
//     are resident per CU and overlap each other's phases: about one 128-row chunk per range, rounded up to a whole
//     number of ranges per CU, equal ROWS (equal-work ranges put several chunks of a sparse region behind each other
//     in one workgroup: 16->16 28 -> 56 us).
extern "C" int fd_spconv_num_ranges(int64_t n_out, int cin, int cout, int dtype) {
    if (n_out <= 0 || dtype != 0) return 0;
    const int n_cu = fd::device_cu_count();
    const int64_t tiles = (n_out + 127) / 128;
    const int mult = fd::tuning(fd::kTuneV2RangesPerCU);  // tuning override
    int64_t per_cu;
    if (mult) per_cu = mult;
    else if (cout >= 64) per_cu = (int64_t)((160 * 1024) / lds_request(cin, cout, 128));
    else per_cu = (tiles + n_cu - 1) / n_cu;
    if (per_cu < 1) per_cu = 1;
    return (int)(per_cu * n_cu);
}

// 1 when the layer shape profits from equal-work ranges (fd_spconv_ranges), 0 when equal rows are the better split
extern "C" int fd_spconv_wants_balanced_ranges(int cin, int cout, int dtype) { return dtype == 0 && cout >= 64 && !fd::tuning(fd::kTuneV2Uniform); }

extern "C" size_t fd_spconv_ranges_workspace_bytes(int64_t n_out) {
    return fd::align_up(sizeof(unsigned) * (size_t)((n_out + kWorkRows - 1) / kWorkRows + 8), 256);
}

extern "C" int fd_spconv_ranges(const int32_t *nbr, int64_t nbr_stride, int K, int64_t n_out, const int32_t *n_out_dev, int n_ranges,
                                int32_t *ranges, void *workspace, size_t workspace_bytes, fd_stream_t stream_) {
    FD_REQUIRE(nbr && ranges && workspace, "fd_spconv_ranges: null argument");
    FD_REQUIRE(n_ranges >= 1 && n_out >= 0 && n_out <= nbr_stride && n_out < (1ll << 31), "fd_spconv_ranges: bad sizes");
    FD_REQUIRE(workspace_bytes >= fd_spconv_ranges_workspace_bytes(n_out), "fd_spconv_ranges: workspace too small");
    hipStream_t stream = fd::as_stream(stream_);
    const int n_blocks = (int)((n_out + kWorkRows - 1) / kWorkRows);
    unsigned *work = (unsigned *)workspace;
    if (n_blocks > 0)
        hipLaunchKernelGGL(block_work_kernel, dim3((unsigned)((n_blocks + 31) / 32)), dim3(256), 0, stream, nbr, nbr_stride, K, (int)n_out, n_out_dev,
                           (unsigned)(fd::tuning(fd::kTuneV2RowCost) ? fd::tuning(fd::kTuneV2RowCost) : kRowCost), work);
    hipLaunchKernelGGL(range_split_kernel, dim3(1), dim3(1024), 0, stream, work, (int)n_out, n_out_dev, n_ranges, ranges);
    return fd::check_launch("fd_spconv_ranges");
}

namespace fd {
// returns 1 when launched, 0 when this shape is not covered (caller falls back to the register kernel)
int spconv_f32_compact_dispatch(const float *in, const void *wp, const float *bias, const float *residual, int relu, const int *nbr,
                                int64_t nbr_stride, int K, int64_t n_in_bound, int n_out, const int *n_out_dev, int cin, int cout, float *out,
                                const int *ranges, int n_ranges, hipStream_t stream) {
    // (input row << 8 | local row) must fit an int32 and the feature matrix a 31-bit buffer range
    if (n_in_bound >= (1ll << 23) || n_in_bound * cin * 4 >= (1ll << 31)) return 0;
    const unsigned in_bytes = (unsigned)(n_in_bound * cin * 4);
    const int dsel = fd::tuning(fd::kTuneV2Depth), tsel = fd::tuning(fd::kTuneV2TM);  // tuning overrides
#define FD_LAUNCH(CI, CO, T, D) \
    launch_compact<CI, CO, T, D>(in, wp, bias, residual, relu, nbr, nbr_stride, K, n_out, n_out_dev, out, in_bytes, ranges, n_ranges, stream)
#define FD_CASE(CI, CO, DDEF, TDEF)                                  \
    if (cin == CI && cout == CO) {                                   \
        const int dd = dsel ? dsel : DDEF, tt = tsel ? tsel : TDEF;  \
        if (tt == 64) {                                              \
            if (dd <= 2) return FD_LAUNCH(CI, CO, 64, 2);            \
            if (dd == 3) return FD_LAUNCH(CI, CO, 64, 3);            \
            return FD_LAUNCH(CI, CO, 64, 4);                         \
        }                                                            \
        if (dd <= 2) return FD_LAUNCH(CI, CO, 128, 2);               \
        if (dd == 3) return FD_LAUNCH(CI, CO, 128, 3);               \
        return FD_LAUNCH(CI, CO, 128, 4);                            \
    }
    FD_CASE(16, 16, 4, 128)
    FD_CASE(16, 32, 4, 128)
    FD_CASE(32, 32, 4, 128)
    FD_CASE(32, 64, 4, 128)
    FD_CASE(64, 64, 3, 128)
    FD_CASE(64, 128, 3, 128)
    FD_CASE(128, 128, 2, 128)
#undef FD_CASE
#undef FD_LAUNCH
    return 0;
}
}  // namespace fd
