// EXPERIMENT, not built into the library (round 3): the producer / consumer formulation that worked for the dense Winograd
// convs, applied to the 64-channel fp32 sparse conv.  Correct (all sparse-conv parity tests passed with it in the dispatch), but
// 1.8x SLOWER than fd_spconv_v2.hip on the 64 -> 64 level (397 vs 214-240 us).  Phase trace: a consumer wave needs 1120 cycles
// per 16-pair item for 512 cycles of MFMA while the producers idle 80 % of the time.  Why: with one 16-column slice per consumer
// wave an item is ONE chain of 16 dependent v_mfma_f32_16x16x4_f32 (40 cycles each instead of 32: 640), and the item decode /
// accumulator / operand requests of the next item sit in front of the chain (~300 issue cycles) with nobody else on the SIMD to
// cover them -- the column-slice kernel runs two 32-column waves per SIMD (two independent chains each) that cover each other.
// Beating it needs two items of one tap multiplied as interleaved chains per wave and the bookkeeping issued inside the chain;
// even then the bound is ~530-640 cycles per item against the ~620 the shipped kernel already reaches.  Kept as a record.
// fp32 sparse convolution for the 64-channel levels with producer and consumer waves.
//
// Same formulation as fd_spconv_v2.hip -- row ranges walked in chunks of <= 128 output rows, the chunk's rulebook slice compacted
// per tap into lists of (input row, local row) pairs, 16-pair items, accumulator tile in LDS, transposed product
// (v_mfma_f32_16x16x4_f32, A = weight fragment, B = gathered rows), the same packed weights, fixed summation order (taps
// ascending), no atomics -- with the work of a workgroup split by ROLE instead of by column slice and tap parity:
//
//   * producers (waves 4-7): gather every item's 16 input rows ONCE (bounds-checked buffer loads, a padding entry reads zeros)
//     and lay them down in LDS in MFMA operand order; the column-slice kernel gathers each row once per column wave through
//     the CU's vector-memory path (tools/probes/gather_probe.hip: the resource its 32/64-channel layers sit on) and spends
//     ~2.2 non-MFMA instructions per MFMA on item decode and gather addressing in the waves that also multiply;
//   * consumers (waves 0-3, one 16 * NBW-column slice each, one copy of the tile): per item one list entry, the accumulator
//     slot, four operand reads and 16 * NBW MFMAs.  The accumulator round trip (LDS read -> C operand) is taken off the chain:
//     consecutive items of one tap update disjoint rows, so the next item's accumulators are requested while the current item
//     multiplies; only the first item of a tap reads after the previous write;
//   * items travel in steps of G: during step s the consumers multiply the G tiles of buffer s & 1 while every producer wave
//     writes its G / 4 tiles of step s + 1 (gathered during step s - 1) into the other buffer and requests those of step s + 2.
//     One barrier per step.  The prologue (slice staging, compaction, accumulator tile = bias + residual) and the epilogue
//     (ReLU, row-contiguous 16-byte stores) are done by all eight waves.
//
// Results do not depend on the work distribution (ranges, chunking): an output element is the sum of its pairs in tap order,
// each pair's 64 products in channel order, on one accumulator.  They differ in the last bits from fd_spconv_v2.hip for 64
// columns, which adds the even-tap and the odd-tap partial sums of two tile copies.
#include "fd_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxTaps = 27, TM = 128;

// tuning builds (tools/probes/build_trace.sh): lane 0 of the first consumer / producer wave accumulates cycles over all chunks:
// [0] prologue, [1] consumer items, [2] consumer barrier wait, [3] producer work, [4] producer barrier wait, [5] epilogue, [6] items, [7] life
#ifdef FD_V2_TRACE
__device__ unsigned long long *g_sptrace;
#define FD_ST(var) const unsigned long long var = __builtin_readcyclecounter()
#define FD_SADD(i, v) sacc[i] += (v)
#else
#define FD_ST(var)
#define FD_SADD(i, v)
#endif

template <int CIN, int COUT, int G>
__global__ void __launch_bounds__(512) spconv_f32_pc(const float *__restrict__ in, const float4 *__restrict__ wp, const float *__restrict__ bias,
                                                     const float *__restrict__ residual, int relu, const int *__restrict__ nbr, int64_t nbr_stride,
                                                     int K, int n_out, const int *__restrict__ n_out_dev, float *__restrict__ out, unsigned in_bytes,
                                                     const int *__restrict__ ranges, int n_ranges, int ranges_per_wg, int rows_per_wg) {
    static_assert(CIN == 64 && (COUT == 64 || COUT == 128) && G % 4 == 0, "shapes");
    constexpr int NC = CIN / 16, NB = COUT / 16, NBW = NB / 4;  // 16-channel chunks; 16-column blocks; blocks per consumer wave
    constexpr int GP = G / 4;                                   // items per producer wave and step
    constexpr int kMaxItems = kMaxTaps * (TM / 16);
    constexpr int kPad = (int)(0xffffff00u | (unsigned)TM);     // list padding: input offset out of range, local row = TM (scratch row)
    constexpr int kRowShift = COUT == 64 ? 8 : 9;               // log2(COUT * 4)
    constexpr int TILE = NC * 1024;                             // one item's gathered rows in operand order [chunk][lane][16 B]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *s_list = reinterpret_cast<int *>(smem);                                            // [K][TM] raw nbr, then compacted entries
    unsigned short *s_items = reinterpret_cast<unsigned short *>(s_list + kMaxTaps * TM);   // [kMaxItems + 2 G] codes (tap << 3 | group), zeros past the end
    unsigned char *s_cnt = reinterpret_cast<unsigned char *>(s_items + kMaxItems + 2 * G);  // [K] pairs per tap (<= 128)
    int *s_pad = reinterpret_cast<int *>(s_cnt + 32);                                       // 16 padding entries
    int *s_misc = s_pad + 16;                                                               // [4]
    float *s_acc = reinterpret_cast<float *>(s_misc + 4);                                   // [TM + 1][COUT]
    unsigned char *s_tile = reinterpret_cast<unsigned char *>(s_acc + (TM + 1) * COUT);     // [2][G][TILE]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 15, lq = lane >> 4;
    if (n_out_dev) n_out = fd::device_count(n_out, n_out_dev);
    int r_begin, r_end;
    if (ranges) {
        const int b0 = blockIdx.x * ranges_per_wg, b1 = min(b0 + ranges_per_wg, n_ranges);
        r_begin = ranges[b0];
        r_end = ranges[b1];
    } else {
        if (n_out_dev) rows_per_wg = (((n_out + (int)gridDim.x - 1) / (int)gridDim.x) + 15) & ~15;
        const int64_t b = (int64_t)blockIdx.x * rows_per_wg;
        r_begin = (int)(b < n_out ? b : n_out);
        r_end = (int)(b + rows_per_wg < n_out ? b + rows_per_wg : n_out);
    }
    if (r_end > n_out) r_end = n_out;
    if (r_begin >= r_end) return;
    const int n_chunks = (r_end - r_begin + TM - 1) / TM;
    const int chunk_rows = (((r_end - r_begin + n_chunks - 1) / n_chunks) + 15) & ~15;
#ifdef FD_V2_TRACE
    unsigned long long sacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    FD_ST(t_life0);

    // rulebook slice of a chunk: one register per 256 entries of the PRODUCER threads; loads are branch-free (clamped address)
    constexpr int NPRE = (kMaxTaps * TM + 255) / 256;
    const int ptid = tid - 256;
    int pre[NPRE];
    auto fetch_slice = [&](int row0) {
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int t = ptid + i * 256;
            int k = t / TM;
            const int r = t - k * TM;
            k = k < K ? k : K - 1;
            int64_t o = (int64_t)row0 + r;
            o = o < nbr_stride ? o : nbr_stride - 1;
            pre[i] = nbr[(int64_t)k * nbr_stride + o];
        }
    };
    if (wave >= 4) fetch_slice(r_begin);

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (int)in_bytes, 0x00020000);

    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const int row0 = r_begin + chunk * chunk_rows;
        const int n_rows = (r_end - row0) < chunk_rows ? (r_end - row0) : chunk_rows;
        if (n_rows <= 0) break;
        FD_ST(t_p0);
        // ------------------------------------------------------------------------------------------------ prologue
        if (wave >= 4) {
            // producers: the prefetched slice
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                const int t = ptid + i * 256;
                const int r = t % TM;
                if (t < K * TM) s_list[t] = r < n_rows ? pre[i] : -1;
            }
            if (ptid < 16) s_pad[ptid] = kPad;
        } else {
            // consumers: the accumulator tile starts from bias + residual (16-byte slots XOR-swizzled by the row, as in
            // fd_spconv_v2.hip); the scratch row (padding pairs) from zero
            constexpr int C4 = COUT / 4, HALF = TM / 2 * C4 / 256;  // float4 per thread and half tile
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float4 iv[HALF];
#pragma unroll
                for (int i = 0; i < HALF; ++i) {
                    const int t = tid + i * 256 + h * (TM / 2) * C4, c4 = t % C4;
                    iv[i] = bias ? reinterpret_cast<const float4 *>(bias)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                if (residual) {
                    float4 rv[HALF];
#pragma unroll
                    for (int i = 0; i < HALF; ++i) {
                        const int t = tid + i * 256 + h * (TM / 2) * C4, r = t / C4, c4 = t - r * C4;
                        const int rr = r < n_rows ? r : n_rows - 1;
                        rv[i] = reinterpret_cast<const float4 *>(residual + (int64_t)(row0 + rr) * COUT)[c4];
                    }
#pragma unroll
                    for (int i = 0; i < HALF; ++i) { iv[i].x += rv[i].x; iv[i].y += rv[i].y; iv[i].z += rv[i].z; iv[i].w += rv[i].w; }
                }
#pragma unroll
                for (int i = 0; i < HALF; ++i) {
                    const int t = tid + i * 256 + h * (TM / 2) * C4, r = t / C4, c4 = t - r * C4;
                    const int ts4 = r * C4 + (c4 ^ (r & 15));
                    reinterpret_cast<float4 *>(s_acc)[ts4] = r < n_rows ? iv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            if (tid < C4) reinterpret_cast<float4 *>(s_acc)[TM * C4 + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        // in-place compaction per tap: wave w takes taps w, w + 8, ...; tails are filled with kPad
        for (int k = wave; k < K; k += 8) {
            const int base = k * TM;
            const int v0 = s_list[base + lane], v1 = s_list[base + 64 + lane];
            s_list[base + lane] = kPad;
            s_list[base + 64 + lane] = kPad;
            const unsigned long long m0 = __ballot(v0 >= 0), m1 = __ballot(v1 >= 0);
            const unsigned long long lt = (1ull << lane) - 1ull;
            const int c0 = __popcll(m0);
            if (v0 >= 0) s_list[base + __popcll(m0 & lt)] = (v0 << 8) | lane;
            if (v1 >= 0) s_list[base + c0 + __popcll(m1 & lt)] = (v1 << 8) | (64 + lane);
            if (lane == 0) s_cnt[k] = (unsigned char)(c0 + __popcll(m1));
        }
        __syncthreads();
        if (wave >= 4 && chunk + 1 < n_chunks) fetch_slice(row0 + chunk_rows);  // the next chunk's slice travels while this chunk computes
        // flattened work list: one item = 16 compacted pairs of one tap, code = (tap << 3) | group; every wave scans the counts
        // (it needs n_items), wave 0 writes the list
        int n_items;
        {
            const int ng = lane < K ? ((int)s_cnt[lane] + 15) >> 4 : 0;
            int inc = ng;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int u = __shfl_up(inc, off);
                if (lane >= off) inc += u;
            }
            n_items = __builtin_amdgcn_readfirstlane(__shfl(inc, 63));
            if (wave == 0) {
                for (int g = 0; g < ng; ++g) s_items[inc - ng + g] = (unsigned short)((lane << 3) | g);
                if (lane < 2 * G) s_items[n_items + lane] = 0;  // (slots past the end are read, never used)
            }
        }
        __syncthreads();
        const int n_steps = (n_items + G - 1) / G;
        FD_ST(t_p1);
        FD_SADD(0, t_p1 - t_p0); FD_SADD(6, (unsigned long long)n_items);

        // ------------------------------------------------------------------------------------------------ main loop
        if (wave >= 4) {
            // ---- producers: wave pw owns the items G s + GP pw .. + GP - 1 of every step s
            const int pw = wave - 4;
            u32x4 ga[GP][NC];
            auto gather = [&](int st) {
#pragma unroll
                for (int u = 0; u < GP; ++u) {
                    const int it = st * G + pw * GP + u;
                    const int code = it < n_items ? (int)s_items[it] : 0;
                    const int *lst = it < n_items ? s_list + (code >> 3) * TM + ((code & 7) << 4) : s_pad;  // (uniform select)
                    const unsigned e = (unsigned)lst[lrow];
                    const unsigned vo = (e & 0xffffff00u) + (unsigned)(lq * 16);  // (input row) * CIN * 4 = row << 8 for 64 channels
#pragma unroll
                    for (int c = 0; c < NC; ++c) ga[u][c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo + c * 64, 0, 0);
                }
            };
            auto put = [&](int st) {
#pragma unroll
                for (int u = 0; u < GP; ++u) {
                    unsigned char *dst = s_tile + ((st & 1) * G + pw * GP + u) * TILE + lane * 16;
#pragma unroll
                    for (int c = 0; c < NC; ++c) *reinterpret_cast<u32x4 *>(dst + c * 1024) = ga[u][c];
                }
            };
            gather(0);
            put(0);
            gather(1);
            __syncthreads();
            for (int st = 0; st < n_steps; ++st) {
                FD_ST(q0);
                put(st + 1);      // (gathered during step st - 1; steps past the end hold zeros nobody reads)
                gather(st + 2);
                FD_ST(q1);
                __syncthreads();
                FD_ST(q2);
                FD_SADD(3, q1 - q0); FD_SADD(4, q2 - q1);
            }
        } else {
            // ---- consumers: wave cw owns the columns 16 NBW cw .. + 16 NBW - 1
            const int cw = wave;
            unsigned char *acc_bytes = reinterpret_cast<unsigned char *>(s_acc);
            const unsigned slot0 = (unsigned)(cw * NBW * 4) + (unsigned)lq;  // 16-byte slot of this lane's 4 channels in block nw = 0
            auto acc_off = [&](unsigned e, unsigned(&off)[NBW]) {
                const unsigned arow = e & 255u;
#pragma unroll
                for (int nw = 0; nw < NBW; ++nw) off[nw] = (arow << kRowShift) + (((slot0 + 4u * nw) ^ (arow & 15u)) << 4);
            };
            auto entry_of = [&](int it) -> unsigned {  // list entry of this lane's pair of item `it` (padding past the end)
                const int code = (int)s_items[it];
                const int *lst = it < n_items ? s_list + (code >> 3) * TM + ((code & 7) << 4) : s_pad;
                return (unsigned)lst[lrow];
            };
            auto tap_of = [&](int it) -> int { return it < n_items ? (int)s_items[it] >> 3 : -1; };
            float4 b[NC][NBW];  // this wave's weight slice of the current tap
            auto load_b = [&](int k) {
                const float4 *wk = wp + ((int64_t)k * NC * NB + cw * NBW) * 64 + lane;
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int nw = 0; nw < NBW; ++nw) b[c][nw] = wk[(c * NB + nw) * 64];
            };
            // Software pipeline, with the same LDS operations on every path (a conditional request makes hipcc wait for ALL
            // outstanding LDS traffic before the next use): while item i multiplies, the code of item i + 3, the list entry of i + 2
            // (from the code read one item earlier), the accumulators of i + 1 and the operand fragments of i + 1 are requested.
            // The accumulator request is only USED inside a tap (disjoint rows); at a tap change the value may be stale and the
            // item re-reads its accumulators first thing -- the one conditional read, placed in front of the unconditional ones.
            // (the code is wave-uniform, but left to itself hipcc moves it to a scalar register -- s_waitcnt lgkmcnt(0) +
            //  v_readfirstlane -- right behind the read: a whole LDS round trip per item; the empty asm keeps it in a vector
            //  register until the next item uses it)
            auto code_of = [&](int it) -> int {
                int v = (int)s_items[it];
                asm volatile("" : "+v"(v));
                return v;
            };
            auto entry_from = [&](int it, int code_v) -> unsigned {
                const int code = __builtin_amdgcn_readfirstlane(code_v);
                const int *lst = it < n_items ? s_list + (code >> 3) * TM + ((code & 7) << 4) : s_pad;
                return (unsigned)lst[lrow];
            };
            int k_cur = tap_of(0), k_next = tap_of(1);
            if (n_items > 0) load_b(k_cur);
            unsigned e_cur = entry_of(0), e_next = entry_of(1);
            int code_n2 = code_of(2);
            f32x4 accp[2][NBW];  // accumulators of the current / the next item (parity of the item index: G is even)
            u32x4 a[2][NC];      // operand fragments of the current / the next item
            bool have = false;   // the current item's accumulators were requested during the previous item (wave-uniform)
            __syncthreads();
            for (int st = 0; st < n_steps; ++st) {
                FD_ST(c0);
                const unsigned char *tiles = s_tile + (st & 1) * G * TILE + lane * 16;
#pragma unroll
                for (int c = 0; c < NC; ++c) a[0][c] = *reinterpret_cast<const u32x4 *>(tiles + c * 1024);  // first item of the step: exposed
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const int it = st * G + j;
                    if (it >= n_items) break;  // (uniform; only the last step is partial)
                    unsigned aoff[NBW], noff[NBW];
                    acc_off(e_cur, aoff);
                    acc_off(e_next, noff);
                    if (!have) {
#pragma unroll
                        for (int nw = 0; nw < NBW; ++nw) accp[j & 1][nw] = *reinterpret_cast<const f32x4 *>(acc_bytes + aoff[nw]);
                    }
                    const int code_n3 = code_of(it + 3);
                    const unsigned e_n2 = entry_from(it + 2, code_n2);
#pragma unroll
                    for (int nw = 0; nw < NBW; ++nw) accp[(j + 1) & 1][nw] = *reinterpret_cast<const f32x4 *>(acc_bytes + noff[nw]);
#pragma unroll
                    for (int c = 0; c < NC; ++c) a[(j + 1) & 1][c] = *reinterpret_cast<const u32x4 *>(tiles + ((j + 1) % G) * TILE + c * 1024);
                    const bool same = k_next == k_cur;
                    const bool reload = !same && k_next >= 0;  // last item of its tap: the next tap's weights replace b[c] behind its MFMAs
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const float4 av = __builtin_bit_cast(float4, a[j & 1][c]);
#define FD_K(C)                                        \
    _Pragma("unroll") for (int nw = 0; nw < NBW; ++nw) \
        accp[j & 1][nw] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[c][nw].C, av.C, accp[j & 1][nw], 0, 0, 0);
                        FD_K(x) FD_K(y) FD_K(z) FD_K(w)
#undef FD_K
                        if (reload) {
                            const float4 *wk = wp + ((int64_t)k_next * NC * NB + cw * NBW) * 64 + lane;
#pragma unroll
                            for (int nw = 0; nw < NBW; ++nw) b[c][nw] = wk[(c * NB + nw) * 64];
                        }
                    }
#pragma unroll
                    for (int nw = 0; nw < NBW; ++nw) *reinterpret_cast<f32x4 *>(acc_bytes + aoff[nw]) = accp[j & 1][nw];
                    __builtin_amdgcn_sched_barrier(0);
                    have = same;
                    k_cur = k_next;
                    k_next = it + 2 < n_items ? __builtin_amdgcn_readfirstlane(code_n2) >> 3 : -1;
                    e_cur = e_next;
                    e_next = e_n2;
                    code_n2 = code_n3;
                }
                FD_ST(c1);
                __syncthreads();
                FD_ST(c2);
                FD_SADD(1, c1 - c0); FD_SADD(2, c2 - c1);
            }
        }
        // ------------------------------------------------------------------------------------------------ epilogue
        FD_ST(t_e0);
        __syncthreads();
        constexpr int C4 = COUT / 4;
        for (int t = tid; t < n_rows * C4; t += 512) {
            const int r = t / C4, c4 = t - r * C4;
            const int ts4 = r * C4 + (c4 ^ (r & 15));
            float4 v = reinterpret_cast<const float4 *>(s_acc)[ts4];
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            reinterpret_cast<float4 *>(out + (int64_t)(row0 + r) * COUT)[c4] = v;
        }
        __syncthreads();  // the next chunk re-uses the lists and the tile
        FD_ST(t_e1);
        FD_SADD(5, t_e1 - t_e0);
    }
#ifdef FD_V2_TRACE
    if (g_sptrace && (tid == 0 || tid == 256)) {
        unsigned long long *o = g_sptrace + (size_t)blockIdx.x * 8;
        if (tid == 0) { o[0] = sacc[0]; o[1] = sacc[1]; o[2] = sacc[2]; o[5] = sacc[5]; o[6] = sacc[6]; o[7] = __builtin_readcyclecounter() - t_life0; }
        else { o[3] = sacc[3]; o[4] = sacc[4]; }
    }
#endif
}

template <int COUT, int G>
constexpr size_t pc_lds_bytes() {
    return sizeof(int) * kMaxTaps * TM + sizeof(unsigned short) * (kMaxTaps * (TM / 16) + 2 * G) + 32 + 64 + 16 + sizeof(float) * (TM + 1) * COUT +
           (size_t)2 * G * 4096;
}

template <int COUT, int G>
int launch_pc(const float *in, const void *wp, const float *bias, const float *residual, int relu, const int *nbr, int64_t nbr_stride, int K, int n_out,
              const int *n_out_dev, float *out, unsigned in_bytes, const int *ranges, int n_ranges, hipStream_t stream) {
    constexpr size_t lds = (pc_lds_bytes<COUT, G>() + 255) / 256 * 256;
    static std::atomic<uint64_t> lds_set{0};
    auto kern = spconv_f32_pc<64, COUT, G>;
    if (!fd::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, lds_set)) return 0;
    // one workgroup per CU (LDS): with a range table the workgroup takes `per` consecutive ranges as one; without, equal rows
    const int n_cu = fd::device_cu_count();
    int grid, per = 1, rows_per = 0;
    if (ranges) {
        per = (n_ranges + n_cu - 1) / n_cu;
        grid = (n_ranges + per - 1) / per;
    } else {
        grid = n_cu;
        rows_per = (((n_out + grid - 1) / grid) + 15) & ~15;
        if (rows_per < 16) rows_per = 16;
        if (!n_out_dev) grid = (n_out + rows_per - 1) / rows_per;
    }
    if (grid <= 0) return 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, stream, in, (const float4 *)wp, bias, residual, relu, nbr, nbr_stride, K, n_out,
                       n_out_dev, out, in_bytes, ranges, n_ranges, per, rows_per);
    return 1;
}

}  // namespace

#ifdef FD_V2_TRACE
extern "C" int fd_debug_set_spconv_pc_trace(void *p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_sptrace), &p, sizeof(p)) == hipSuccess ? 0 : -1; }
#endif

namespace fd {
// Returns 1 when launched (64 -> 64 and 64 -> 128, fp32), 0 when the shape is not handled here.
int spconv_f32_pc_dispatch(const float *in, const void *wp, const float *bias, const float *residual, int relu, const int *nbr, int64_t nbr_stride, int K,
                           int64_t n_in_bound, int n_out, const int *n_out_dev, int cin, int cout, float *out, const int *ranges, int n_ranges,
                           hipStream_t stream) {
    if (cin != 64 || (cout != 64 && cout != 128) || K > kMaxTaps) return 0;
    if (n_in_bound >= (1ll << 23) || n_in_bound * cin * 4 >= (1ll << 31)) return 0;  // (input row << 8 | local row) in an int32; 31-bit buffer range
    const unsigned in_bytes = (unsigned)(n_in_bound * cin * 4);
    if (cout == 64) return launch_pc<64, 8>(in, wp, bias, residual, relu, nbr, nbr_stride, K, n_out, n_out_dev, out, in_bytes, ranges, n_ranges, stream);
    return launch_pc<128, 4>(in, wp, bias, residual, relu, nbr, nbr_stride, K, n_out, n_out_dev, out, in_bytes, ranges, n_ranges, stream);
}
}  // namespace fd
