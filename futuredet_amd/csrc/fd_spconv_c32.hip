// fp32 sparse convolution for the 32 -> 32 layers: 32-pair items on v_mfma_f32_32x32x2_f32.
//
// Same formulation as fd_spconv_v2.hip (row ranges walked in chunks of <= 128 rows, rulebook tile compacted in LDS per tap,
// accumulator tile in LDS, transposed product, no atomics, fixed summation order) with a different unit of work.  The SQ
// counters of the 16x16x4 kernel on 32 -> 32 (profiles/round2_spconv_sq_counters.txt): 3.0 VALU + 3.9 SALU + 0.7 LDS + 0.4
// VMEM instructions per MFMA, waves stalled on a dependency 47 % of their cycles, and no VALU instruction ever co-executes
// with an MFMA on a SIMD -- a 16-pair x 16-column item carries 8 MFMAs (256 matrix-pipe cycles) for ~50 instructions of item
// decode, gather addressing and accumulator hand-over.  Here an item is 32 pairs x all 32 columns: CIN/2 MFMAs of 64 cycles
// (1024 cycles at CIN = 32) for about the same bookkeeping.  Measured (300k-point cloud, 268k rows, 3.17 M pairs): 146 -> 141 us
// per 32 -> 32 layer, +0.8 % on the whole sweep -- far less than the instruction count suggests, so the item overhead is not
// what binds these layers either; with the rows split between two waves (64-row halves, 28 pairs per tap and half on
// average) half of the 32-pair slots were padding and the kernel was slower (158 us); the 16 -> 32 strided layer (2.4 pairs
// per row) stays on the 16-pair kernel (54 vs 70 us).
//   * waves: wave ts handles the taps k with k % 4 == ts over all rows of the chunk and accumulates into its own copy of the
//     tile; the epilogue adds the four copies (80 KB of LDS per workgroup, two workgroups per CU);
//   * B operand (gathered rows): lane (pair n = lane % 32, k-half h = lane / 32) loads the four consecutive channels
//     8 c + 4 h .. + 3 of its pair's input row per 8-channel step c; A operand: the same channels of W[tap] for output
//     channel lane % 32 (fd_spconv_pack_weight appends this layout for Cout = 32);
//   * D: lane (n, h) register v holds output channel 8 (v / 4) + 4 h + v % 4 of pair n -> four 16-byte accumulator slots
//     per lane, XOR-swizzled exactly as the 32-column tile of the v2 kernel, so the two kernels share the epilogue layout.
#include "fd_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxTaps = 27;
constexpr int TM = 128, COUT = 32, TS = 4, WR = 1, RW = TM / WR;

// Tuning builds only (tools/probes/build_exp.sh fd_spconv_c32 img -DFD_SKELETON_IMAGE, tools/skeleton_image_bench.py): the stop-rule
// experiment of the round-4 review.  Mode 1: every (workgroup, chunk) DUMPS what its skeleton phases produce -- the compacted per-tap lists,
// their counts, the four waves' item lists -- to a global image; mode 2: a launch LOADS that image instead of staging the raw rulebook
// slice, compacting it and building the item lists, i.e. it runs as if the lists came ready-made with the rulebook of the indice_key
// (built once, shared by the 4-5 convolutions of a level).  Results of a mode-2 launch are exact.
#ifdef FD_SKELETON_IMAGE
__device__ int *g_img;
__device__ int g_img_mode;
constexpr int kImgChunks = 4;                                   // chunks per workgroup the image has room for
constexpr int kImgInts = kMaxTaps * TM + 28 + 64 + 4;           // list, counts (27 x 4 bytes), items (4 x 32 u16), item counts
#endif

template <int CIN, int DEPTH>
__global__ void __launch_bounds__(256) spconv_f32_c32(const float *__restrict__ in, const float4 *__restrict__ wp, const float *__restrict__ bias,
                                                      const float *__restrict__ residual, int relu, const int *__restrict__ nbr, int64_t nbr_stride,
                                                      int K, int n_out, const int *__restrict__ n_out_dev, float *__restrict__ out, unsigned in_bytes,
                                                      const int *__restrict__ ranges, int rows_per_range) {
    constexpr int NCH = CIN / 8;                               // 8-channel steps = float4 loads per lane and item
    constexpr int kMaxItems = ((kMaxTaps + TS - 1) / TS) * (RW / 32);  // per wave: its taps x 32-pair groups of the chunk
    constexpr int kPad = (int)(0xffffff00u | (unsigned)TM);  // list padding: input offset out of range, local row = TM (scratch row)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *s_list = reinterpret_cast<int *>(smem);                                           // [K][TM] raw nbr, then compacted entries
    unsigned short *s_items = reinterpret_cast<unsigned short *>(s_list + kMaxTaps * TM);  // [4 waves][kMaxItems]
    unsigned char *s_cnt = reinterpret_cast<unsigned char *>(s_items + 4 * 32);            // [K][4] (<= 64 each)
    int *s_pad = reinterpret_cast<int *>(s_cnt + 112);                                     // 32 padding entries
    float *s_acc = reinterpret_cast<float *>(s_pad + 32);                                  // [TS][TM + 1][COUT]
    static_assert(kMaxItems <= 32, "item list slot");

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: what derives from it stays in SGPRs
    if (n_out_dev) n_out = fd::device_count(n_out, n_out_dev);
    int r_begin, r_end;
    if (ranges) {
        r_begin = ranges[blockIdx.x];
        r_end = ranges[blockIdx.x + 1];
    } else {
        if (n_out_dev) rows_per_range = (((n_out + (int)gridDim.x - 1) / (int)gridDim.x) + 15) & ~15;
        const int64_t b = (int64_t)blockIdx.x * rows_per_range;
        r_begin = (int)(b < n_out ? b : n_out);
        r_end = (int)(b + rows_per_range < n_out ? b + rows_per_range : n_out);
    }
    if (r_end > n_out) r_end = n_out;
    if (r_begin >= r_end) return;
    const int n_chunks = (r_end - r_begin + TM - 1) / TM;
    const int chunk_rows = (((r_end - r_begin + n_chunks - 1) / n_chunks) + 15) & ~15;
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
#ifdef FD_SKELETON_IMAGE
    const bool img_any_load = g_img && g_img_mode == 2;
    if (!img_any_load)
#endif
    fetch_slice(r_begin);

    const int ln = lane & 31, lh = lane >> 5;
    const int ts = wave, wr = 0;
    unsigned char *acc_bytes = reinterpret_cast<unsigned char *>(s_acc + ts * (TM + 1) * COUT);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (int)in_bytes, 0x00020000);
    unsigned short *items = s_items + wave * 32;

    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const int row0 = r_begin + chunk * chunk_rows;
        const int n_rows = (r_end - row0) < chunk_rows ? (r_end - row0) : chunk_rows;
        if (n_rows <= 0) break;
        // ---- stage the prefetched slice, clear the accumulators
#ifdef FD_SKELETON_IMAGE
        int *img = g_img ? g_img + ((int64_t)blockIdx.x * kImgChunks + (chunk < kImgChunks ? chunk : kImgChunks - 1)) * kImgInts : nullptr;
        const bool img_load = img && g_img_mode == 2;
        if (img_load) {
            for (int t = tid; t < K * TM / 4; t += 256) reinterpret_cast<int4 *>(s_list)[t] = reinterpret_cast<const int4 *>(img)[t];
            if (tid < 28) reinterpret_cast<int *>(s_cnt)[tid] = img[kMaxTaps * TM + tid];
            if (tid < 64) reinterpret_cast<int *>(s_items)[tid] = img[kMaxTaps * TM + 28 + tid];
        } else
#endif
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int t = tid + i * 256;
            const int r = t % TM;
            if (t < K * TM) s_list[t] = r < n_rows ? pre[i] : -1;
        }
        // tile copy 0 starts from bias + residual (see fd_spconv_v2.hip: no dependent global loads left in the epilogue)
        {
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
                    const int rr = r < n_rows ? r : n_rows - 1;
                    rv[i] = reinterpret_cast<const float4 *>(residual + (int64_t)(row0 + rr) * COUT)[c4];
                }
#pragma unroll
                for (int i = 0; i < NINIT; ++i) { iv[i].x += rv[i].x; iv[i].y += rv[i].y; iv[i].z += rv[i].z; iv[i].w += rv[i].w; }
            }
#pragma unroll
            for (int i = 0; i < NINIT; ++i) {
                const int t = tid + i * 256, r = t / C4i, c4 = t - r * C4i;
                const int ts4 = r * C4i + (c4 ^ (int)(((unsigned)r >> 1) & 7u));
                reinterpret_cast<float4 *>(s_acc)[ts4] = r < n_rows ? iv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            for (int t = TM * C4i + tid; t < TS * (TM + 1) * COUT / 4; t += 256) reinterpret_cast<float4 *>(s_acc)[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tid < 32) s_pad[tid] = kPad;
        __syncthreads();
        // ---- in-place compaction per (tap, row half): wave w takes taps w, w + 4, ...; tails are filled with kPad
#ifdef FD_SKELETON_IMAGE
        for (int k = img_load ? K : wave; k < K; k += 4) {
#else
        for (int k = wave; k < K; k += 4) {
#endif
#pragma unroll
            for (int h = 0; h < WR; ++h) {
                const int base = k * TM + h * RW;
                int v[RW / 64];
#pragma unroll
                for (int g = 0; g < RW / 64; ++g) v[g] = s_list[base + g * 64 + lane];
#pragma unroll
                for (int g = 0; g < RW / 64; ++g) s_list[base + g * 64 + lane] = kPad;
                int count = 0;
#pragma unroll
                for (int g = 0; g < RW / 64; ++g) {
                    const unsigned long long m = __ballot(v[g] >= 0);
                    const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
                    if (v[g] >= 0) s_list[base + pos] = (v[g] << 8) | (h * RW + g * 64 + lane);
                    count += __popcll(m);
                }
                if (lane == 0) s_cnt[k * 4 + h] = (unsigned char)count;
            }
        }
        __syncthreads();
#ifdef FD_SKELETON_IMAGE
        if (!img_any_load)
#endif
        if (chunk + 1 < n_chunks) fetch_slice(row0 + chunk_rows);  // the next chunk's slice travels while this chunk computes

        // ---- work list of this wave: item = 32 compacted pairs of one tap, code = (tap << 3) | group
        int n_items;
#ifdef FD_SKELETON_IMAGE
        if (img_load) {
            n_items = img[kMaxTaps * TM + 28 + 64 + wave];
        } else
#endif
        {
            const int ng = (lane < K && (lane % TS) == ts) ? ((int)s_cnt[lane * 4 + wr] + 31) >> 5 : 0;
            int inc = ng;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int u = __shfl_up(inc, off);
                if (lane >= off) inc += u;
            }
            n_items = __builtin_amdgcn_readfirstlane(__shfl(inc, 63));
            for (int g = 0; g < ng; ++g) items[inc - ng + g] = (unsigned short)((lane << 3) | g);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#ifdef FD_SKELETON_IMAGE
        if (img && g_img_mode == 1 && chunk < kImgChunks) {  // dump: the compacted lists, counts and this wave's items
            __syncthreads();
            for (int t = tid; t < K * TM; t += 256) img[t] = s_list[t];
            if (tid < 28) img[kMaxTaps * TM + tid] = reinterpret_cast<const int *>(s_cnt)[tid];
            if (tid < 64) img[kMaxTaps * TM + 28 + tid] = reinterpret_cast<const int *>(s_items)[tid];
            if (lane == 0) img[kMaxTaps * TM + 28 + 64 + wave] = n_items;
        }
#endif

        // software pipeline as in the v2 kernel: item code one iteration ahead of the list entry, the list entry one ahead
        // of the gather, the gather DEPTH - 1 items ahead of the MFMAs; everything branch-free (a slot past the end of the
        // list reads padding entries: out-of-range gather offset -> zeros, accumulator row TM = scratch row)
        int k_r[DEPTH], row_r[DEPTH];
        u32x4 a_r[DEPTH][NCH];
        auto stage_a0 = [&](int it) -> int { return (int)items[it < n_items ? it : 0]; };
        auto stage_a1 = [&](int it, int code_v, int &kk, int &e) {
            const bool v = it < n_items;
            const int code = __builtin_amdgcn_readfirstlane(code_v);
            const int ks = v ? (code >> 3) : 0;
            kk = v ? ks : -1;
            const int *lst = v ? s_list + ks * TM + wr * RW + ((code & 7) << 5) : s_pad;
            e = lst[ln];
        };
        auto gather_offset = [&](int e) -> unsigned {
            const unsigned hi = (unsigned)e & 0xffffff00u;  // (input row) << 8; row bytes = CIN * 4
            return (CIN == 32 ? hi >> 1 : hi >> 2) + (unsigned)(lh * 16);
        };
        auto gather_step = [&](unsigned voff, int c) { return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + c * 32, 0, 0); };
        float4 b[NCH];  // this tap's weights: channels 8 c + 4 lh .. + 3, output channel ln
        auto load_b = [&](int k) {
            const float4 *wk = wp + (int64_t)k * NCH * 64 + lane;
#pragma unroll
            for (int c = 0; c < NCH; ++c) b[c] = wk[c * 64];
        };
        int k_s, e_s, code_s;
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d) {
            stage_a1(d, stage_a0(d), k_s, e_s);
            const unsigned vo = gather_offset(e_s);
#pragma unroll
            for (int c = 0; c < NCH; ++c) a_r[d][c] = gather_step(vo, c);
            k_r[d] = k_s;
            row_r[d] = e_s;
        }
        k_r[DEPTH - 1] = -1;
        row_r[DEPTH - 1] = kPad;
        stage_a1(DEPTH - 1, stage_a0(DEPTH - 1), k_s, e_s);
        code_s = stage_a0(DEPTH);
        if (n_items > 0) load_b(k_r[0]);

        for (int i0 = 0; i0 < n_items; i0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                // accumulator row of this lane's pair: four swizzled 16-byte slots (channels 8 q + 4 lh .. + 3, q = 0..3)
                const unsigned arow = (unsigned)row_r[d] & 255u;
                const unsigned abase = arow << 7, aswz = (arow >> 1) & 7u;
                unsigned aoff[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) aoff[q] = abase + ((((unsigned)(2 * q + lh)) ^ aswz) << 4);
                f32x16 acc;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4 *>(acc_bytes + aoff[q]);
                    acc[4 * q + 0] = v.x; acc[4 * q + 1] = v.y; acc[4 * q + 2] = v.z; acc[4 * q + 3] = v.w;
                }
                // refill the ring slot freed by the previous item and advance the bookkeeping stages before the MFMAs
                const int dn = (d + DEPTH - 1) % DEPTH;
                {
                    const int it = i0 + d + DEPTH - 1;
                    const unsigned vo = gather_offset(e_s);
                    k_r[dn] = k_s;
                    row_r[dn] = e_s;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) a_r[dn][c] = gather_step(vo, c);
                    stage_a1(it + 1, code_s, k_s, e_s);
                    code_s = stage_a0(it + 2);
                }
                const int knext = k_r[(d + 1) % DEPTH];
                const bool reload = knext >= 0 && knext != k_r[d];  // last item of its tap (wave-uniform)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const float4 av = __builtin_bit_cast(float4, a_r[d][c]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b[c].x, av.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b[c].y, av.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b[c].z, av.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b[c].w, av.w, acc, 0, 0, 0);
                    if (reload) b[c] = (wp + (int64_t)knext * NCH * 64 + lane)[c * 64];  // next tap's step c, under the remaining MFMAs
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4 *>(acc_bytes + aoff[q]) = make_float4(acc[4 * q + 0], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        // ---- epilogue: whole chunk, float4 per thread, rows contiguous in global memory (swizzled slots in LDS)
        constexpr int C4 = COUT / 4;
        for (int t = tid; t < n_rows * C4; t += 256) {
            const int r = t / C4, c4 = t - r * C4;
            const int row = row0 + r;
            const int ts4 = r * C4 + (c4 ^ (int)(((unsigned)r >> 1) & 7u));
            float4 v = reinterpret_cast<const float4 *>(s_acc)[ts4];
#pragma unroll
            for (int q = 1; q < TS; ++q) {
                const float4 v2 = reinterpret_cast<const float4 *>(s_acc + q * (TM + 1) * COUT)[ts4];
                v.x += v2.x; v.y += v2.y; v.z += v2.z; v.w += v2.w;
            }
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            reinterpret_cast<float4 *>(out + (int64_t)row * COUT)[c4] = v;
        }
        __syncthreads();  // the next chunk re-uses the list and the accumulator tile
    }
}

constexpr size_t kLdsC32 = sizeof(int) * kMaxTaps * TM + sizeof(unsigned short) * 4 * 32 + 112 + 128 + sizeof(float) * TS * (TM + 1) * COUT;

template <int CIN, int DEPTH>
int launch_c32(const float *in, const void *wp, const float *bias, const float *residual, int relu, const int *nbr, int64_t nbr_stride, int K,
               int n_out, const int *n_out_dev, float *out, unsigned in_bytes, const int *ranges, int n_ranges, hipStream_t stream) {
    int rows_per = 0;
    if (!ranges) {
        if (n_ranges <= 0) n_ranges = (n_out + TM - 1) / TM;
        rows_per = (((n_out + n_ranges - 1) / n_ranges) + 15) & ~15;
        if (!n_out_dev) n_ranges = (n_out + rows_per - 1) / rows_per;
    }
    static std::atomic<uint64_t> lds_set{0};  // devices on which this instantiation has its dynamic-LDS limit raised (> 64 KB)
    if (!fd::ensure_dynamic_lds(reinterpret_cast<const void *>(spconv_f32_c32<CIN, DEPTH>), kLdsC32, lds_set)) return 0;
    hipLaunchKernelGGL((spconv_f32_c32<CIN, DEPTH>), dim3((unsigned)n_ranges), dim3(256), kLdsC32, stream, in, (const float4 *)wp, bias, residual, relu,
                       nbr, nbr_stride, K, n_out, n_out_dev, out, in_bytes, ranges, rows_per);
    return 1;
}

}  // namespace

#ifdef FD_SKELETON_IMAGE
extern "C" int fd_debug_set_skeleton_image(void *p, int mode) {
    return (hipMemcpyToSymbol(HIP_SYMBOL(g_img), &p, sizeof(p)) == hipSuccess && hipMemcpyToSymbol(HIP_SYMBOL(g_img_mode), &mode, sizeof(mode)) == hipSuccess) ? 0 : -1;
}
extern "C" int fd_debug_skeleton_image_ints(void) { return kImgChunks * kImgInts; }
#endif

namespace fd {
// wp: the 32x32x2 fragment layout ([K][CIN / 8][64 lanes] float4, see fd_spconv_pack_weight).  Returns 1 when launched.
int spconv_f32_c32_dispatch(const float *in, const void *wp, const float *bias, const float *residual, int relu, const int *nbr, int64_t nbr_stride,
                            int K, int64_t n_in_bound, int n_out, const int *n_out_dev, int cin, int cout, float *out, const int *ranges,
                            int n_ranges, hipStream_t stream) {
    if (cout != 32 || cin != 32) return 0;
    if (n_in_bound >= (1ll << 23) || n_in_bound * cin * 4 >= (1ll << 31)) return 0;
    const unsigned in_bytes = (unsigned)(n_in_bound * cin * 4);
    const int depth = fd::tuning(fd::kTuneV2Depth);
    if (depth == 2) return launch_c32<32, 2>(in, wp, bias, residual, relu, nbr, nbr_stride, K, n_out, n_out_dev, out, in_bytes, ranges, n_ranges, stream);
    if (depth == 4) return launch_c32<32, 4>(in, wp, bias, residual, relu, nbr, nbr_stride, K, n_out, n_out_dev, out, in_bytes, ranges, n_ranges, stream);
    return launch_c32<32, 3>(in, wp, bias, residual, relu, nbr, nbr_stride, K, n_out, n_out_dev, out, in_bytes, ranges, n_ranges, stream);
}
}  // namespace fd
