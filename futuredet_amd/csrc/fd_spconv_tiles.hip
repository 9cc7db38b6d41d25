// Sparse convolution on a COMPRESSED, tile-local rulebook: the narrow levels (64-byte feature rows: 16 fp32 or 32 bf16 channels).
//
// Replaces spconv 1.0's indice_conv / indice_subm_conv for conv_input / conv1 (16 -> 16), the strided 16 -> 32 convolution and -- in the bf16
// configurations -- conv2's 32 -> 32 layers of det3d/models/backbones/scn.py:99-121, fused with the folded BatchNorm1d, residual add and ReLU.
//
// What bound the round 3-5 kernels of these layers (fd_spconv_f32r.hip, fd_spconv_bf16.hip RESIDENT): not HBM, not the matrix pipe, not
// latency (round 6: a deeper gather ring, a second workgroup per CU, the residual / slice requested a tile ahead -- none moved the 16 -> 16
// layer) but the per-CU vector-memory path.  A wave gathered its 16 rows once per TAP that has any pair among them: 9.5 bounds-checked
// gather instructions per 16-row tile for 70 pairs -- 46 % of the row slots of every instruction empty, and an instruction of 16 x 64-byte
// rows costs the CU's texture path 64-90 cycles whatever its lanes do (profiles/round3_gather_probe.txt, round5_gather_probe.txt).  The dense
// rulebook nbr[27][n] feeding them was 56-91 % "-1".
//
// Here the rulebook of such a layer is one 64-byte RECORD per 16-row tile -- 27 row masks (bit r of mask t: row r of the tile has a
// neighbour under tap t), the tile's pair count and the offset of its pairs in a packed list (tap-major inside the tile, rows ascending
// inside a tap) -- 4 + 4 x pairs/row bytes per row instead of 108 (fd_rulebook_tiles builds it from nbr[K][n], once per indice_key).  A wave
//   * reads its tile's record and packed input rows (both requested ahead: the record two tiles, the list one),
//   * gathers ONLY the pairs that exist, 16 pairs per instruction, by LDS-DMA (global_load_lds_dwordx4: lane 4 p + q fetches 16-byte piece q of
//     pair p's row; no registers, no ds_write) into a per-wave LDS buffer -- 4.4 full instructions per tile instead of 9.5 half-empty ones,
//   * walks the taps that have a pair: lane (row r, piece q) reads its operand from buffer slot prefix(tap) + popcount(mask & below r) -- or the
//     zero slot when the row lacks the tap (a select, no branch, no exec mask) -- and the MFMAs accumulate in registers, weights resident in LDS.
// Summation order: taps ascending, channels as the MFMA sums them -- the order of the kernels this one replaces (bit-identical results, tested).
// A tile with more pairs than the buffer holds is processed in tap segments (greedy, whole taps); correctness never depends on the data.
#include "fd_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxTaps = 27;
constexpr int kRecWords = 16;  // 14 words of masks (27 x 16 bits), pair count, packed offset

// ------------------------------------------------------------------------------------------------------------------ builder
// One wave per 16-row tile: lane (q = lane / 16, r = lane % 16) looks at tap 4 it + q of row r, so one ballot yields the row masks of four
// taps; the tile's segment of the packed list is claimed with one atomicAdd on a cursor (the layout depends on the order the tiles arrive
// in, the results of the convolution do not).
__global__ void __launch_bounds__(256) rulebook_tiles_kernel(const int *__restrict__ nbr, int64_t nbr_stride, int K, int n_out, const int *__restrict__ n_out_dev,
                                                             unsigned *__restrict__ records, int *__restrict__ packed, unsigned *__restrict__ cursor) {
    n_out = fd::device_count(n_out, n_out_dev);
    const int lane = threadIdx.x & 63, q = lane >> 4, r = lane & 15;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t row = tile * 16 + r;
    if (tile * 16 >= n_out) return;
    constexpr int NIT = (kMaxTaps + 3) / 4;
    int v[NIT];
    unsigned long long bal[NIT];
    int total = 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int t = 4 * it + q;
        v[it] = (t < K && row < n_out) ? nbr[(int64_t)t * nbr_stride + row] : -1;
        bal[it] = __ballot(v[it] >= 0);
        total += __popcll(bal[it]);
    }
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(cursor, (unsigned)total);
    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    int run = 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const unsigned long long m = bal[it];
        if (v[it] >= 0) packed[base + run + __popcll(m & ((1ull << (16 * q)) - 1ull)) + __popcll((m >> (16 * q)) & ((1ull << r) - 1ull))] = v[it];
        run += __popcll(m);
    }
    // record: words 0..13 = the 27 masks (two per word: taps 2 j, 2 j + 1 are adjacent 16-bit fields of ballot j / 2), 14 = pairs, 15 = offset
    if (lane < kRecWords) {
        unsigned w = 0;
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if ((lane >> 1) == it) w = (unsigned)(bal[it] >> (32 * (lane & 1)));
        if (lane == 14) w = (unsigned)total;
        if (lane == 15) w = base;
        records[tile * kRecWords + lane] = w;
    }
}

// ------------------------------------------------------------------------------------------------------------------ consumer
// DT 0: fp32 features, CIN = 16 (v_mfma_f32_16x16x4_f32 x 4 per tap and 16-column block); DT 1: bf16 features, CIN = 32
// (v_mfma_f32_16x16x32_bf16 x 1).  A feature row is 64 bytes = four 16-byte pieces either way.
template <int DT, int COUT, int NW, int CAP>
__global__ void __launch_bounds__(NW * 64) spconv_tiles(const unsigned char *__restrict__ in, const u32x4 *__restrict__ wp, const float *__restrict__ bias,
                                                        const unsigned char *__restrict__ residual, int relu, const unsigned *__restrict__ records,
                                                        const int *__restrict__ packed, int K, int n_out, const int *__restrict__ n_out_dev,
                                                        unsigned char *__restrict__ out, int exp) {
    constexpr int NB = COUT / 16;
    constexpr int OB = DT == 0 ? 4 : 2;              // bytes per output element
    constexpr int GMAX = CAP / 16;                   // gather instructions of a full buffer
    static_assert(CAP % 16 == 0, "whole gather instructions");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4 *s_w = reinterpret_cast<u32x4 *>(smem);                                     // [K][NB][64] fragments (fd_spconv_pack_weight order)
    unsigned char *s_g = smem + (size_t)kMaxTaps * NB * 1024;                          // [NW][(CAP + 1) * 64]: gathered rows, slot CAP = zeros
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 15, lq = lane >> 4;
    n_out = fd::device_count(n_out, n_out_dev);
    unsigned char *g = s_g + (size_t)wave * (CAP + 1) * 64;
    // contiguous chunk of tiles per workgroup (XCD-contiguous eighths), walked NW tiles at a time
    const unsigned lb = fd::xcd_swizzle(blockIdx.x, gridDim.x);
    const int n_tiles = (n_out + 15) >> 4;
    const int tpb = (n_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t_lo = (int)lb * tpb, t_hi = t_lo + tpb < n_tiles ? t_lo + tpb : n_tiles;
    int tile_first = t_lo + wave;
    int n_iter = tile_first < t_hi ? (t_hi - tile_first + NW - 1) / NW : 0;
    int tile_step = NW;
    if (exp & 128) {  // experiment: tiles interleaved over the whole grid (all waves sweep the row range together)
        tile_first = (int)blockIdx.x * NW + wave;
        tile_step = (int)gridDim.x * NW;
        n_iter = tile_first < n_tiles ? (n_tiles - tile_first + tile_step - 1) / tile_step : 0;
    }

    // (an unconditional load at a clamped address: a select on the loaded value would make hipcc wait for the load where it is issued;
    //  lanes 16..63 mirror lanes 0..15, only those are read; a record past the wave's last tile is never used)
    auto load_rec = [&](int it) -> unsigned {
        int tile = tile_first + it * tile_step;
        tile = tile < n_tiles ? tile : n_tiles - 1;
        return records[(int64_t)tile * kRecWords + (lane & (kRecWords - 1))];
    };
    auto rec_word = [&](unsigned rec, int j) -> unsigned { return (unsigned)__builtin_amdgcn_readlane((int)rec, j); };
    // packed input rows of pairs [first, first + count) of a tile, lane 4 p + q <- pair 16 gg + p (count <= CAP; lanes past the end re-read the last)
    auto load_idx = [&](unsigned base, int first, int count, int(&idx)[GMAX]) {
#pragma unroll
        for (int gg = 0; gg < GMAX; ++gg) {
            if (gg * 16 < count) {  // (uniform)
                int p = gg * 16 + (lane >> 2);
                p = p < count ? p : count - 1;
                idx[gg] = packed[(size_t)base + first + p];
            }
        }
    };
    auto issue_gathers = [&](int count, const int(&idx)[GMAX]) {
#pragma unroll
        for (int gg = 0; gg < GMAX; ++gg) {
            if (gg * 16 < count)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(in + ((size_t)(unsigned)idx[gg] << 6) + ((lane & 3) << 4)),
                                                 (__attribute__((address_space(3))) void *)(g + gg * 1024), 16, 0, 0);
        }
    };
    // first segment of a tile: taps [0, t1) whose pairs fit the buffer (all K taps for nearly every tile)
    auto first_segment = [&](unsigned rec, int &t1) -> int {
        const int P_all = (int)rec_word(rec, 14);
        if (P_all <= CAP) {  // nearly every tile (the scan below costs ~8 us per tile: a scalar loop around a variable-lane readlane)
            t1 = K;
            return P_all;
        }
        int pairs = 0;
        t1 = 0;
        while (t1 < K) {
            const unsigned m = (rec_word(rec, t1 >> 1) >> (16 * (t1 & 1))) & 0xffffu;
            const int c = __builtin_popcount(m);
            if (pairs + c > CAP) break;
            pairs += c;
            ++t1;
        }
        return pairs;
    };

    // ---- prologue: zero slot, first records / lists, weights (all requests of the start of a workgroup's life travel together)
    if (lane < 4) reinterpret_cast<u32x4 *>(g + CAP * 64)[lane] = (u32x4){0u, 0u, 0u, 0u};
    unsigned rec_cur = load_rec(0), rec_nxt = load_rec(1);
    for (int i = tid; i < K * NB * 64; i += NW * 64) s_w[i] = wp[i];
    int idx_cur[GMAX];
    int t1_cur = 0, seg_cur = 0;
    if (n_iter > 0) {
        seg_cur = first_segment(rec_cur, t1_cur);
        load_idx(rec_word(rec_cur, 15), 0, seg_cur, idx_cur);
    }
    // residual rows one tile ahead (the accumulators of the fp32 form start from bias + residual, as in fd_spconv_f32r.hip; the bf16 form adds
    // the residual to the finished sum in fp32 as fd_spconv_bf16.hip does -- both orders are the replaced kernels')
    u32x4 res_nxt[NB];
    auto load_res = [&](int it) {
        const int row = (tile_first + it * tile_step) * 16 + lrow;
        const size_t rb = (size_t)(row < n_out ? row : 0) * COUT * OB;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if constexpr (DT == 0) res_nxt[nb] = *reinterpret_cast<const u32x4 *>(residual + rb + (16 * nb + 4 * lq) * 4);
            else {
                const uint2 v2 = *reinterpret_cast<const uint2 *>(residual + rb + (16 * nb + 4 * lq) * 2);
                res_nxt[nb] = (u32x4){v2.x, v2.y, 0u, 0u};
            }
        }
    };
    if (residual && n_iter > 0) load_res(0);
    __syncthreads();

    const unsigned lmask = (1u << lrow) - 1u;
    for (int it = 0; it < ((exp & 64) ? 0 : n_iter); ++it) {
        const int row0 = (tile_first + it * tile_step) * 16;
        const unsigned rec = rec_cur;
        const unsigned base = rec_word(rec, 15);
        const int P = (int)rec_word(rec, 14);
        // ---- gathers of the first segment (their packed rows arrived during the previous tile)
        if (!(exp & 1)) issue_gathers(seg_cur, idx_cur);
        u32x4 res_cur[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) res_cur[nb] = res_nxt[nb];
        f32x4 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (bias) bv = *reinterpret_cast<const f32x4 *>(bias + 16 * nb + 4 * lq);
            if constexpr (DT == 0) {
                if (residual) bv += __builtin_bit_cast(f32x4, res_cur[nb]);
            }
            acc[nb] = bv;
        }
        if (!(exp & 8)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // ---- requests for the tiles ahead fly under this tile's MFMAs: record of tile it + 2, packed rows and residual of tile it + 1
        rec_cur = rec_nxt;
        rec_nxt = load_rec(it + 2);
        int t1_nxt = 0, seg_nxt = 0;
        int idx_nxt[GMAX];
        if (it + 1 < n_iter) {
            seg_nxt = first_segment(rec_cur, t1_nxt);
            if (!(exp & 4)) load_idx(rec_word(rec_cur, 15), 0, seg_nxt, idx_nxt);
            if (residual) load_res(it + 1);
        }

        int t0 = 0, t1 = t1_cur, consumed = 0, seg = seg_cur;
        while (true) {
            int slot0 = 0;  // buffer slot of the first pair of the current tap
#pragma unroll
            for (int t = 0; t < kMaxTaps; ++t) {
                if (t >= t0 && t < t1) {  // (uniform)
                    const unsigned m = (rec_word(rec, t >> 1) >> (16 * (t & 1))) & 0xffffu;
                    if (m && !(exp & 2)) {
                        const bool has = (m >> lrow) & 1u;
                        const int slot = has ? slot0 + __builtin_popcount(m & lmask) : CAP;
                        const u32x4 b = *reinterpret_cast<const u32x4 *>(g + slot * 64 + lq * 16);
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            const u32x4 wf = s_w[(t * NB + nb) * 64 + lane];
                            if constexpr (DT == 0) {
                                const f32x4 wv = __builtin_bit_cast(f32x4, wf), bvv = __builtin_bit_cast(f32x4, b);
                                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[0], bvv[0], acc[nb], 0, 0, 0);
                                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[1], bvv[1], acc[nb], 0, 0, 0);
                                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[2], bvv[2], acc[nb], 0, 0, 0);
                                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[3], bvv[3], acc[nb], 0, 0, 0);
                            } else {
                                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, b), acc[nb], 0, 0, 0);
                            }
                        }
                        slot0 += __builtin_popcount(m);
                    }
                }
            }
            consumed += seg;
            if (t1 >= K || consumed >= P) break;
            // ---- a tile with more pairs than the buffer holds: the next tap segment (rare; its list is fetched on demand)
            t0 = t1;
            seg = 0;
            while (t1 < K) {
                const unsigned m = (rec_word(rec, t1 >> 1) >> (16 * (t1 & 1))) & 0xffffu;
                const int c = __builtin_popcount(m);
                if (seg + c > CAP) break;
                seg += c;
                ++t1;
            }
            int idx_more[GMAX];
            load_idx(base, consumed, seg, idx_more);
            issue_gathers(seg, idx_more);  // (LDS operations of a wave execute in order: the reads of the previous segment are done)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }

        // ---- epilogue: lane (row lrow, quad lq) holds channels 16 nb + 4 lq .. + 3 of its row
        const int row = row0 + lrow;
        const size_t ob = (size_t)(row < n_out ? row : 0) * COUT * OB;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f32x4 v = acc[nb];
            if constexpr (DT == 1) {
                if (residual) {
                    const bf16x4 rv = __builtin_bit_cast(bf16x4, (uint2){res_cur[nb][0], res_cur[nb][1]});
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += (float)rv[i];
                }
            }
            if (relu) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
            }
            if (row < n_out && !(exp & 32)) {
                if constexpr (DT == 0) *reinterpret_cast<f32x4 *>(out + ob + (16 * nb + 4 * lq) * 4) = v;
                else {
                    bf16x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
                    *reinterpret_cast<bf16x4 *>(out + ob + (16 * nb + 4 * lq) * 2) = o;
                }
            }
        }
        t1_cur = t1_nxt;
        seg_cur = seg_nxt;
#pragma unroll
        for (int gg = 0; gg < GMAX; ++gg) idx_cur[gg] = idx_nxt[gg];
    }
}

template <int DT, int COUT, int NW, int CAP>
int launch_tiles(const void *in, const void *wp, const float *bias, const void *residual, int relu, const unsigned *records, const int *packed, int K,
                 int n_out, const int *n_out_dev, int64_t n_expected, void *out, hipStream_t stream) {
    constexpr size_t lds = (size_t)kMaxTaps * (COUT / 16) * 1024 + (size_t)NW * (CAP + 1) * 64;
    static_assert(lds <= 160 * 1024, "LDS request");
    auto kern = spconv_tiles<DT, COUT, NW, CAP>;
    static std::atomic<uint64_t> lds_set{0};
    if (lds > 65536 && !fd::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, lds_set)) return 0;
    int64_t grid = (n_expected + NW * 16 - 1) / (NW * 16);
    const int64_t cap = fd::device_cu_count();  // persistent workgroups, one per CU (the LDS request is the CU's)
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, stream, (const unsigned char *)in, (const u32x4 *)wp, bias, (const unsigned char *)residual, relu,
                       records, packed, K, n_out, n_out_dev, (unsigned char *)out, fd::tuning(fd::kTuneSpconvTiles) > 0 ? fd::tuning(fd::kTuneSpconvTiles) : 0);
    return 1;
}

}  // namespace

extern "C" size_t fd_rulebook_tiles_record_bytes(int64_t n_rows) { return n_rows < 0 ? 0 : (size_t)((n_rows + 15) / 16) * kRecWords * 4; }

extern "C" int fd_rulebook_tiles(const int32_t *nbr, int64_t nbr_stride, int K, int64_t n_out, const int32_t *n_out_dev, void *records, int32_t *packed,
                                 int64_t packed_capacity, uint32_t *cursor, fd_stream_t stream_) {
    FD_REQUIRE(K >= 1 && K <= kMaxTaps, "fd_rulebook_tiles: K must be in [1,27]");
    FD_REQUIRE(n_out >= 0 && n_out <= nbr_stride && n_out < (1ll << 31), "fd_rulebook_tiles: n_out out of range");
    FD_REQUIRE(packed_capacity >= (int64_t)K * n_out && packed_capacity < (1ll << 32), "fd_rulebook_tiles: the packed list needs room for K * n_out entries (< 2^32)");
    if (n_out == 0) return FD_OK;
    FD_REQUIRE(nbr && records && packed && cursor, "fd_rulebook_tiles: null argument");
    FD_REQUIRE(((uintptr_t)records & 63) == 0, "fd_rulebook_tiles: records must be 64-byte aligned");
    hipStream_t stream = fd::as_stream(stream_);
    int rc = fd::fill_words(cursor, 0u, 1, stream);
    if (rc != FD_OK) return rc;
    const int64_t tiles = (n_out + 15) / 16;
    hipLaunchKernelGGL(rulebook_tiles_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, stream, nbr, nbr_stride, K, (int)n_out, n_out_dev, (unsigned *)records, packed,
                       cursor);
    return fd::check_launch("fd_rulebook_tiles");
}

// 1 when fd_spconv_apply_tiles takes the shape (dtype 0 = f32, 1 = bf16)
extern "C" int fd_spconv_tiles_supported(int cin, int cout, int dtype) {
    if (fd::tuning(fd::kTuneSpconvTiles) < 0) return 0;  // "spconv_tiles" = -1: the dense-table kernels everywhere (A/B runs, variant tests)
    if (dtype == 0) return cin == 16 && (cout == 16 || cout == 32);
    if (dtype == 1) return cin == 32 && cout == 32;
    return 0;
}

extern "C" int fd_spconv_apply_tiles(const void *in_feats, int64_t n_in, const void *wpacked, const float *bias, const void *residual, int relu,
                                     const void *records, const int32_t *packed, int K, int64_t n_out, const int32_t *n_out_dev, int64_t n_expected, int cin,
                                     int cout, int dtype, void *out_feats, fd_stream_t stream_) {
    FD_REQUIRE(K >= 1 && K <= kMaxTaps, "fd_spconv_apply_tiles: K must be in [1,27]");
    FD_REQUIRE(fd_spconv_tiles_supported(cin, cout, dtype), "fd_spconv_apply_tiles: unsupported shape %d -> %d (dtype %d): 64-byte feature rows only", cin, cout, dtype);
    FD_REQUIRE(n_out >= 0 && n_out < (1ll << 31) && n_in >= 0 && n_in < (1ll << 26), "fd_spconv_apply_tiles: row counts out of range");
    if (n_expected <= 0 || n_expected > n_out) n_expected = n_out;
    if (n_out == 0) return FD_OK;
    FD_REQUIRE(in_feats && wpacked && records && packed && out_feats, "fd_spconv_apply_tiles: null argument");
    hipStream_t stream = fd::as_stream(stream_);
    const unsigned *rec = (const unsigned *)records;
    int ok = 0;
    if (dtype == 0 && cout == 16) ok = launch_tiles<0, 16, 16, 128>(in_feats, wpacked, bias, residual, relu, rec, packed, K, (int)n_out, n_out_dev, n_expected, out_feats, stream);
    else if (dtype == 0) ok = launch_tiles<0, 32, 16, 96>(in_feats, wpacked, bias, residual, relu, rec, packed, K, (int)n_out, n_out_dev, n_expected, out_feats, stream);
    else ok = launch_tiles<1, 32, 16, 96>(in_feats, wpacked, bias, residual, relu, rec, packed, K, (int)n_out, n_out_dev, n_expected, out_feats, stream);
    if (!ok) {
        fd::set_error("fd_spconv_apply_tiles: the runtime refused the kernel's LDS request");
        return FD_ELAUNCH;
    }
    return fd::check_launch("fd_spconv_apply_tiles");
}
