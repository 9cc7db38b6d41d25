// fp32 sparse convolution, second formulation: tile-level pair compaction.
//
// fp32 MFMA (v_mfma_f32_16x16x4_f32) runs at the fp32 vector rate, so the fp32 layers are MFMA-bound and
// every zero row fed to the matrix core is lost time.  With 27 taps only 18-63 % of the (output row, tap)
// pairs exist, and a 16-row output group almost never lacks a tap entirely, so the register-resident
// output-stationary kernel (fd_spconv.hip) spends 40-80 % of its MFMAs on zeros.  This kernel removes them:
//
//   * a workgroup owns TM = 128 consecutive output rows (spatially sorted, so their inputs are close);
//   * the rulebook tile [K][TM] is staged in LDS and compacted IN PLACE per tap with wave ballots /
//     prefix popcounts into lists of (input row << 8 | local output row) entries, tails filled with -1 --
//     the "LDS-staged rulebook tile";
//   * accumulators for the whole tile live in LDS ([TM + 1][COUT] fp32, 64 KB at COUT = 128; row TM is a
//     scratch row that absorbs the padding lanes so the accumulator traffic needs no exec masking);
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


template <int CIN, int COUT, int TM, int DEPTH>
__global__ void __launch_bounds__(256) spconv_f32_compact(const float *__restrict__ in, const float4 *__restrict__ wp,
                                                          const float *__restrict__ bias, const float *__restrict__ residual, int relu,
                                                          const int *__restrict__ nbr, int64_t nbr_stride, int K, int n_out,
                                                          float *__restrict__ out, unsigned in_bytes, const int *__restrict__ tile_order) {
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

    // tiles differ in work by up to 2x (dense regions near the sensor): the optional order puts heavy tiles first and
    // pairs them with light ones on a CU (fd_spconv_tile_order); without it tiles run in index order.
    const int tile = tile_order ? tile_order[blockIdx.x] : (int)blockIdx.x;
    const int row0 = tile * TM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int t = tid; t < K * TM; t += 256) {
        int k = t / TM, r = t - k * TM;
        int64_t o = (int64_t)row0 + r;
        s_list[t] = (o < nbr_stride) ? nbr[(int64_t)k * nbr_stride + o] : -1;
    }
    for (int t = tid; t < TS * (TM + 1) * COUT / 4; t += 256) reinterpret_cast<float4 *>(s_acc)[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 16) s_pad[tid] = kPad;
    __syncthreads();
    // ---- in-place compaction: wave w takes taps w, w+4, ...; tails are filled with kPad
    for (int k = wave; k < K; k += 4) {
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

    const int lrow = lane & 15, lq = lane >> 4;
    const int wc = wave % WC, wr = TS == 1 ? wave / WC : 0, ts = TS == 1 ? 0 : wave / WC;
    const int cb = wc * NBW * 16;
    constexpr int kRowShift = COUT == 16 ? 6 : COUT == 32 ? 7 : COUT == 64 ? 8 : 9;  // log2(COUT * 4)
    static_assert((COUT * 4) == (1 << kRowShift), "COUT must be 16, 32, 64 or 128");
    unsigned char *acc_bytes = reinterpret_cast<unsigned char *>(s_acc + ts * (TM + 1) * COUT);  // this wave's tile copy
    const unsigned lane_off = (unsigned)(cb + lrow) * 4u;
    // ---- flattened work list of this wave: one item = 16 compacted pairs of one tap, code = (tap << 3) | group
    unsigned short *items = s_items + wave * kMaxItems;
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
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (int)in_bytes, 0x00020000);
    int k_r[DEPTH];
    i32x4 rows_r[DEPTH];  // entries of the 4 accumulator rows this lane touches (group rows 4*lq .. 4*lq+3)
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
    auto stage_a1 = [&](int it, int code_v, int &kk, int &e, i32x4 &rows) {
        const bool v = it < n_items;  // uniform (n_items is in a scalar register)
        const int code = __builtin_amdgcn_readfirstlane(code_v);
        const int ks = v ? (code >> 3) : 0;
        kk = v ? ks : -1;
        // past the end of the work list the (scalar) list pointer selects a block of 16 padding entries: no per-lane select
        const int *lst = v ? s_list + ks * TM + wr * RW + ((code & 7) << 4) : s_pad;
        e = lst[lrow];
        rows = *reinterpret_cast<const i32x4 *>(lst + lq * 4);
    };
    auto stage_b = [&](int kk, int e, const i32x4 &rows_in, i32x4 &rows, u32x4(&a)[NC]) -> unsigned {
        rows = rows_in;
        // byte offset of the input row = (e >> 8) * CIN * 4, computed on the masked entry without a multiply
        const unsigned hi = (unsigned)e & 0xffffff00u;
        const unsigned voff = (CIN >= 64 ? hi << (CIN == 128 ? 1 : 0) : hi >> (CIN == 32 ? 1 : 2)) + (unsigned)(lq * 16);
        (void)a;
        return voff;
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
    i32x4 rows_s;
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) {
        stage_a1(d, stage_a0(d), k_s, e_s, rows_s);
        const unsigned vo = stage_b(k_s, e_s, rows_s, rows_r[d], a_r[d]);
#pragma unroll
        for (int c = 0; c < NC; ++c) a_r[d][c] = gather_chunk(vo, c);
        k_r[d] = k_s;
    }
    k_r[DEPTH - 1] = -1;
    stage_a1(DEPTH - 1, stage_a0(DEPTH - 1), k_s, e_s, rows_s);  // staged for the first loop iteration
    code_s = stage_a0(DEPTH);
    if (n_items > 0) load_b(k_r[0], b);  // weights of the first tap (the only exposed weight latency of the tile)

    auto mfma_chunk = [&](const u32x4 &a4, const float4(&bc)[NBW], f32x4(&acc)[NACC]) {
        const float4 av = __builtin_bit_cast(float4, a4);
#pragma unroll
        for (int nw = 0; nw < NBW; ++nw) acc[nw] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bc[nw].x, acc[nw], 0, 0, 0);
#pragma unroll
        for (int nw = 0; nw < NBW; ++nw) acc[nw] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bc[nw].y, acc[nw], 0, 0, 0);
#pragma unroll
        for (int nw = 0; nw < NBW; ++nw) acc[nw] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bc[nw].z, acc[nw], 0, 0, 0);
#pragma unroll
        for (int nw = 0; nw < NBW; ++nw) acc[nw] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bc[nw].w, acc[nw], 0, 0, 0);
    };

    for (int i0 = 0; i0 < n_items; i0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            // accumulator rows: the row field of a padding entry is the scratch row TM
            // (byte offsets: one AND and one shift-add per row; the lane's column offset is loop invariant)
            unsigned aoff[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) aoff[r] = (((unsigned)rows_r[d][r] & 255u) << kRowShift) + lane_off;
            // The old accumulator values are requested first and are the C operand of the MFMA chain: the matrix pipe
            // does the accumulation and D goes back to LDS untouched -- no vector-ALU work on the accumulators at all
            // (VALU instructions of one wave barely overlap the MFMAs of the other waves of its SIMD, so every VALU
            // instruction saved per item is matrix-pipe time gained).  The bookkeeping below covers the LDS latency.
            f32x4 acc[NACC];
#pragma unroll
            for (int nw = 0; nw < NBW; ++nw)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[nw][r] = *reinterpret_cast<const float *>(acc_bytes + aoff[r] + nw * 64);
            // refill the slot freed by the previous item BEFORE this item's MFMAs (left to itself hipcc sinks the loads
            // below the MFMA block and waits vmcnt(0) for them at the top of the next item), then advance the two
            // bookkeeping stages; their results are first touched in the next iteration.
            const int dn = (d + DEPTH - 1) % DEPTH;  // compile-time after unrolling
            unsigned vo;
            {
                const int it = i0 + d + DEPTH - 1;
                vo = stage_b(k_s, e_s, rows_s, rows_r[dn], a_r[dn]);
                k_r[dn] = k_s;
                stage_a1(it + 1, code_s, k_s, e_s, rows_s);
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
            for (int nw = 0; nw < NBW; ++nw)
#pragma unroll
                for (int r = 0; r < 4; ++r) *reinterpret_cast<float *>(acc_bytes + aoff[r] + nw * 64) = acc[nw][r];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    // ---- epilogue: whole tile, float4 per thread, rows contiguous
    constexpr int C4 = COUT / 4;
    for (int t = tid; t < TM * C4; t += 256) {
        const int r = t / C4, c4 = t - r * C4;
        const int row = row0 + r;
        if (row >= n_out) break;
        float4 v = reinterpret_cast<const float4 *>(s_acc)[t];
#pragma unroll
        for (int q = 1; q < TS; ++q) {
            const float4 v2 = reinterpret_cast<const float4 *>(s_acc + q * (TM + 1) * COUT)[t];
            v.x += v2.x; v.y += v2.y; v.z += v2.z; v.w += v2.w;
        }
        if (bias) {
            const float4 bv = reinterpret_cast<const float4 *>(bias)[c4];
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        }
        if (residual) {
            const float4 rv = reinterpret_cast<const float4 *>(residual + (int64_t)row * COUT)[c4];
            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
        }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        reinterpret_cast<float4 *>(out + (int64_t)row * COUT)[c4] = v;
    }
}

template <int CIN, int COUT, int TM, int DEPTH>
int launch_compact(const float *in, const void *wp, const float *bias, const float *residual, int relu, const int *nbr,
                   int64_t nbr_stride, int K, int n_out, float *out, unsigned in_bytes, const int *tile_order, hipStream_t stream) {
    const size_t lds = sizeof(int) * kMaxTaps * TM + sizeof(unsigned short) * 4 * kMaxTaps * (TM / 16) + 112 + 64 + sizeof(float) * (TM + 1) * COUT * (COUT == 32 || COUT == 64 ? 2 : 1);
    const size_t lds_pad = (size_t)fd::tuning(fd::kTuneV2LdsPad);  // occupancy experiments
    // Occupancy is not a lever here: MFMA and non-MFMA instructions of the waves sharing a SIMD execute almost serially
    // (128 channels: one workgroup per CU is only 9 % slower than two), and for the 64->64 layers two workgroups per
    // CU beat the three that would fit (300 -> 278 us: less contention in the gather path), so their LDS request is
    // rounded up to just over a third of the CU's 160 KB.
    const size_t lds_req = (CIN == 64 && COUT == 64 && TM == 128 && !lds_pad ? (lds > 56 * 1024 ? lds : (size_t)56 * 1024) : lds) + lds_pad;
    static std::atomic<uint64_t> lds_set{0};  // devices on which this instantiation has its LDS limit raised
    auto kern = spconv_f32_compact<CIN, COUT, TM, DEPTH>;
    if (lds_pad) {  // tuning runs change the request between calls: set it every time
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_req) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
    } else if (!fd::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds_req, lds_set)) {
        return 0;
    }
    dim3 grid((unsigned)((n_out + TM - 1) / TM));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds_req, stream, in, (const float4 *)wp, bias, residual, relu, nbr, nbr_stride, K, n_out, out, in_bytes, tile_order);
    return 1;
}

// ---------------------------------------------------------------------------------------------- tile order
// work of a 128-row tile = number of 16-pair MFMA groups = sum over taps of ceil(valid rows / 16)
__global__ void __launch_bounds__(256) tile_work_kernel(const int *__restrict__ nbr, int64_t nbr_stride, int K, int n_tiles, unsigned *__restrict__ keys) {
    // one workgroup per tile; wave w counts taps w, w+4, ... with two 64-row ballots each (no barrier per tap)
    const int tile = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ int s_work[4];
    int work = 0;
    const int64_t o0 = (int64_t)tile * 128 + lane, o1 = o0 + 64;
    for (int k = wave; k < K; k += 4) {
        const int *row = nbr + (int64_t)k * nbr_stride;
        const bool v0 = o0 < nbr_stride && row[o0] >= 0;
        const bool v1 = o1 < nbr_stride && row[o1] >= 0;
        work += (__popcll(__ballot(v0)) + __popcll(__ballot(v1)) + 15) >> 4;
    }
    if (lane == 0) s_work[wave] = work;
    __syncthreads();
    if (threadIdx.x == 0) {
        work = s_work[0] + s_work[1] + s_work[2] + s_work[3];
        keys[tile] = ((unsigned)work << 20) | (0xfffffu - (unsigned)tile);  // sort key: work desc, tile asc
    }
}

// single workgroup: counting sort by work (at most 27 * 8 = 216 distinct values), heaviest first, then the CU
// pairing rule.  Ties are placed in arrival order of the atomics: the order only steers scheduling, results do not
// depend on it.
__global__ void __launch_bounds__(1024) tile_sort_kernel(const unsigned *__restrict__ keys, int n_tiles, int n_cu, int *__restrict__ sorted,
                                                         int *__restrict__ order) {
    __shared__ int s_hist[256], s_start[256];
    if (threadIdx.x < 256) s_hist[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n_tiles; i += 1024) atomicAdd(&s_hist[min((int)(keys[i] >> 20), 255)], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int w = 255; w >= 0; --w) { s_start[w] = acc; acc += s_hist[w]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_tiles; i += 1024) {
        const int w = min((int)(keys[i] >> 20), 255);
        sorted[atomicAdd(&s_start[w], 1)] = i;
    }
    __syncthreads();
    __threadfence_block();
    // all tiles resident at once (<= 2 per CU): workgroups b and b + n_cu tend to share a CU, so the heaviest n_cu tiles
    // go first and are followed by the rest lightest-first; otherwise plain heaviest-first (greedy LPT by the dispatcher)
    const bool pair = n_tiles <= 2 * n_cu;
    const int nh = n_tiles < n_cu ? n_tiles : n_cu;
    for (int i = threadIdx.x; i < n_tiles; i += 1024) {
        const int src = (!pair || i < nh) ? i : n_tiles - 1 - (i - nh);
        order[i] = sorted[src];
    }
}

}  // namespace

extern "C" int fd_spconv_tile_order(const int32_t *nbr, int64_t nbr_stride, int K, int64_t n_out, int32_t *order, void *workspace,
                                    size_t workspace_bytes, fd_stream_t stream_) {
    FD_REQUIRE(nbr && order && workspace, "fd_spconv_tile_order: null argument");
    const int64_t n_tiles = (n_out + 127) / 128;
    FD_REQUIRE(n_tiles >= 1 && n_tiles < (1 << 20), "fd_spconv_tile_order: supports up to 2^20 tiles of 128 rows (got %lld)", (long long)n_tiles);
    FD_REQUIRE(workspace_bytes >= 2 * sizeof(unsigned) * (size_t)n_tiles, "fd_spconv_tile_order: workspace too small");
    hipStream_t stream = fd::as_stream(stream_);
    const int n_cu = fd::device_cu_count();
    unsigned *keys = (unsigned *)workspace;
    int *sorted = (int *)(keys + n_tiles);
    hipLaunchKernelGGL(tile_work_kernel, dim3((unsigned)n_tiles), dim3(256), 0, stream, nbr, nbr_stride, K, (int)n_tiles, keys);
    hipLaunchKernelGGL(tile_sort_kernel, dim3(1), dim3(1024), 0, stream, keys, (int)n_tiles, n_cu, sorted, order);
    return fd::check_launch("fd_spconv_tile_order");
}

namespace fd {
// returns 1 when launched, 0 when this shape is not covered (caller falls back to the register kernel)
int spconv_f32_compact_dispatch(const float *in, const void *wp, const float *bias, const float *residual, int relu, const int *nbr,
                                int64_t nbr_stride, int K, int64_t n_in_bound, int n_out, int cin, int cout, float *out, const int *tile_order, hipStream_t stream) {
    // (input row << 8 | local row) must fit an int32 and the feature matrix a 31-bit buffer range
    if (n_in_bound >= (1ll << 23) || n_in_bound * cin * 4 >= (1ll << 31)) return 0;
    const unsigned in_bytes = (unsigned)(n_in_bound * cin * 4);
    const int dsel = fd::tuning(fd::kTuneV2Depth), tsel = fd::tuning(fd::kTuneV2TM);  // tuning overrides
#define FD_LAUNCH(CI, CO, T, D) \
    launch_compact<CI, CO, T, D>(in, wp, bias, residual, relu, nbr, nbr_stride, K, n_out, out, in_bytes, (T == 128 ? tile_order : nullptr), stream)
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
