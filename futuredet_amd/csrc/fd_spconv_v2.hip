// fp32 sparse convolution, second formulation: tile-level pair compaction.
//
// fp32 MFMA (v_mfma_f32_16x16x4_f32) runs at the fp32 vector rate, so the fp32 layers are MFMA-bound and
// every zero row fed to the matrix core is lost time.  With 27 taps only 18-63 % of the (output row, tap)
// pairs exist, and a 16-row output group almost never lacks a tap entirely, so the register-resident
// output-stationary kernel (fd_spconv.hip) spends 40-80 % of its MFMAs on zeros.  This kernel removes them:
//
//   * a workgroup owns TM = 128 consecutive output rows (spatially sorted, so their inputs are close);
//   * the rulebook tile [K][TM] is staged in LDS and compacted IN PLACE per tap with wave ballots /
//     prefix popcounts into lists of (input row, local output row) pairs -- the "LDS-staged rulebook tile";
//   * accumulators for the whole tile live in LDS ([TM][COUT] fp32, 64 KB at COUT = 128);
//   * each wave owns a column slice of the tile (and, for narrow COUT, a row subset); per tap it keeps its
//     slice of W[k] in registers, walks the compacted list 16 pairs at a time, gathers the 16 input rows
//     straight into MFMA A-fragment layout (one 16-byte load per lane and 16-channel chunk), reads the 16
//     accumulator rows from LDS as the MFMA C operand, runs the CIN/4 x NBW MFMAs and writes D back.
//     A wave never shares an accumulator element with another wave, so there are no atomics, no barriers in
//     the tap loop, and the summation order per output element is fixed (taps ascending) -> deterministic.
//   * epilogue: bias (+ residual) (+ ReLU) on the LDS tile, written out with 16-byte row-contiguous stores.
#include "fd_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxTaps = 27;
constexpr int kMaxItems = kMaxTaps * 8;  // per wave: taps x (128 rows / 16)

template <int CIN, int COUT, int TM, int DEPTH>
__global__ void __launch_bounds__(256) spconv_f32_compact(const float *__restrict__ in, const float4 *__restrict__ wp,
                                                          const float *__restrict__ bias, const float *__restrict__ residual, int relu,
                                                          const int *__restrict__ nbr, int64_t nbr_stride, int K, int n_out,
                                                          float *__restrict__ out, unsigned in_bytes) {
    constexpr int NB = COUT / 16, NC = CIN / 16;
    constexpr int WC = NB >= 4 ? 4 : NB;  // column splits across the 4 waves
    constexpr int NBW = NB / WC;          // 16-column blocks per wave
    constexpr int WR = 4 / WC;            // row splits
    constexpr int RW = TM / WR;           // rows in a wave's row set
    static_assert(TM <= 128 && RW >= 16, "local row must fit 7 bits");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *s_list = reinterpret_cast<int *>(smem);             // [K][TM]  raw nbr, then compacted (in<<7 | row)
    int *s_cnt = s_list + kMaxTaps * TM;                     // [K][WR]
    unsigned short *s_items = reinterpret_cast<unsigned short *>(s_cnt + kMaxTaps * 4);  // [4 waves][kMaxItems]
    float *s_acc = reinterpret_cast<float *>(s_items + 4 * kMaxItems);  // [TM][COUT], 16-byte aligned (15984 B in)

    const int tile = fd::xcd_swizzle(blockIdx.x, gridDim.x);
    const int row0 = tile * TM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int t = tid; t < K * TM; t += 256) {
        int k = t / TM, r = t - k * TM;
        int64_t o = (int64_t)row0 + r;
        s_list[t] = (o < nbr_stride) ? nbr[(int64_t)k * nbr_stride + o] : -1;
    }
    for (int t = tid; t < TM * COUT / 4; t += 256) reinterpret_cast<float4 *>(s_acc)[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    // ---- in-place compaction: wave w takes taps w, w+4, ...
    for (int k = wave; k < K; k += 4) {
#pragma unroll
        for (int wr = 0; wr < WR; ++wr) {
            const int base = k * TM + wr * RW;
            int count = 0;
#pragma unroll
            for (int h = 0; h < (RW + 63) / 64; ++h) {
                const int r = h * 64 + lane;
                const int v = (r < RW) ? s_list[base + r] : -1;
                const unsigned long long m = __ballot(v >= 0);
                const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
                if (v >= 0) s_list[base + pos] = (v << 7) | (wr * RW + r);
                count += __popcll(m);
            }
            if (lane == 0) s_cnt[k * 4 + wr] = count;
        }
    }
    __syncthreads();

    const int lrow = lane & 15, lq = lane >> 4;
    const int wc = wave % WC, wr = wave / WC;
    const int cb = wc * NBW * 16;
    // ---- flattened work list of this wave: one item = 16 compacted pairs of one tap, code = (tap << 3) | group
    unsigned short *items = s_items + wave * kMaxItems;
    int n_items;
    {
        const int ng = (lane < K) ? (s_cnt[lane * 4 + wr] + 15) >> 4 : 0;
        int inc = ng;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int u = __shfl_up(inc, off);
            if (lane >= off) inc += u;
        }
        n_items = __shfl(inc, 63);
        for (int g = 0; g < ng; ++g) items[inc - ng + g] = (unsigned short)((lane << 3) | g);
    }
    // wave-local LDS hand-off (items written above are read below by other lanes of the same wave)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    // ring of DEPTH prefetched items: entry (input row << 7 | local row, or -1), tap, gathered A fragments.
    // The gathers are buffer loads with hardware bounds checking: a padding lane / a slot past the end of the
    // work list gets an out-of-range offset and reads zeros, so the prefetch is branch-free and the compiler can
    // keep exact vmcnt(N) counts (exec-masked loads forced vmcnt(0), i.e. no overlap at all).
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (int)in_bytes, 0x00020000);
    int e_r[DEPTH], k_r[DEPTH];
    u32x4 a_r[DEPTH][NC];
    auto fetch = [&](int it, int &e, int &kk, u32x4(&a)[NC]) {
        const bool v = it < n_items;
        const int code = items[v ? it : 0];
        const int ks = v ? (code >> 3) : 0;
        kk = v ? ks : -1;
        const int idx = (((code & 7) << 4) + lrow) & (TM - 1);
        const int ent = s_list[ks * TM + wr * RW + idx];
        e = (v && idx < s_cnt[ks * 4 + wr]) ? ent : -1;
        const unsigned voff = e >= 0 ? (unsigned)(e >> 7) * (unsigned)(CIN * 4) + (unsigned)(lq * 16) : in_bytes;
#pragma unroll
        for (int c = 0; c < NC; ++c) a[c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + c * 64, 0, 0);
    };
    auto load_b = [&](int k, float4(&dst)[NC][NBW]) {
        const float4 *wk = wp + ((int64_t)k * NC * NB + wc * NBW) * 64 + lane;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int nw = 0; nw < NBW; ++nw) dst[c][nw] = wk[(c * NB + nw) * 64];
    };
    const unsigned long long tapmask = __ballot(lane < K && s_cnt[(lane < K ? lane : 0) * 4 + wr] > 0);
    // small weight slices are double buffered a whole tap ahead; the 128x128 slice (64 VGPRs) is loaded at the tap
    // switch instead, which keeps the kernel at two waves per SIMD (the LDS tile allows two workgroups per CU)
    constexpr bool BPF = NC * NBW <= 8;
    float4 b[NC][NBW], bn[BPF ? NC : 1][BPF ? NBW : 1];
    int kcur = -1;
    if constexpr (BPF) {
        if (tapmask) load_b(__builtin_ctzll(tapmask), bn);  // weights of the first non-empty tap
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) fetch(d, e_r[d], k_r[d], a_r[d]);

    for (int i0 = 0; i0 < n_items; i0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (k_r[d] >= 0) {  // wave-uniform; no vector-memory op inside except the weight prefetch at a tap switch
                if (k_r[d] != kcur) {
                    kcur = k_r[d];
                    if constexpr (BPF) {
#pragma unroll
                        for (int c = 0; c < NC; ++c)
#pragma unroll
                            for (int nw = 0; nw < NBW; ++nw) b[c][nw] = bn[c][nw];
                        const unsigned long long rest = (kcur + 1 < 64) ? (tapmask >> (kcur + 1)) : 0ull;
                        if (rest) load_b(kcur + 1 + __builtin_ctzll(rest), bn);  // next tap's weights, a whole tap ahead
                    } else {
                        load_b(kcur, b);
                    }
                }
                const int e = e_r[d];
                int orow[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) orow[r] = __shfl(e, lq * 4 + r);  // entry of group row 4*lq + r (-1 = padding)
                f32x4 acc[NBW];
#pragma unroll
                for (int nw = 0; nw < NBW; ++nw)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[nw][r] = orow[r] >= 0 ? s_acc[(orow[r] & 127) * COUT + cb + nw * 16 + lrow] : 0.0f;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float4 av = __builtin_bit_cast(float4, a_r[d][c]);
#pragma unroll
                    for (int nw = 0; nw < NBW; ++nw) acc[nw] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b[c][nw].x, acc[nw], 0, 0, 0);
#pragma unroll
                    for (int nw = 0; nw < NBW; ++nw) acc[nw] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b[c][nw].y, acc[nw], 0, 0, 0);
#pragma unroll
                    for (int nw = 0; nw < NBW; ++nw) acc[nw] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b[c][nw].z, acc[nw], 0, 0, 0);
#pragma unroll
                    for (int nw = 0; nw < NBW; ++nw) acc[nw] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b[c][nw].w, acc[nw], 0, 0, 0);
                }
#pragma unroll
                for (int nw = 0; nw < NBW; ++nw)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (orow[r] >= 0) s_acc[(orow[r] & 127) * COUT + cb + nw * 16 + lrow] = acc[nw][r];
            }
            fetch(i0 + d + DEPTH, e_r[d], k_r[d], a_r[d]);  // refill the slot just consumed (unconditional, branch-free)
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
                   int64_t nbr_stride, int K, int n_out, float *out, unsigned in_bytes, hipStream_t stream) {
    const size_t lds = sizeof(int) * (kMaxTaps * TM + kMaxTaps * 4) + sizeof(unsigned short) * 4 * kMaxItems + sizeof(float) * TM * COUT;
    static bool attr_set = false;
    auto kern = spconv_f32_compact<CIN, COUT, TM, DEPTH>;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        attr_set = true;
    }
    dim3 grid((unsigned)((n_out + TM - 1) / TM));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, in, (const float4 *)wp, bias, residual, relu, nbr, nbr_stride, K, n_out, out, in_bytes);
    return 1;
}

}  // namespace

namespace fd {
// returns 1 when launched, 0 when this shape is not covered (caller falls back to the register kernel)
int spconv_f32_compact_dispatch(const float *in, const void *wp, const float *bias, const float *residual, int relu, const int *nbr,
                                int64_t nbr_stride, int K, int64_t n_in_bound, int n_out, int cin, int cout, float *out, hipStream_t stream) {
    // (input row << 7 | local row) must fit an int32 and the feature matrix a 31-bit buffer range
    if (n_in_bound >= (1ll << 24) || n_in_bound * cin * 4 >= (1ll << 31)) return 0;
    const unsigned in_bytes = (unsigned)(n_in_bound * cin * 4);
#define FD_CASE(CI, CO) \
    if (cin == CI && cout == CO)  \
        return launch_compact<CI, CO, 128, (CI >= 128 ? 2 : (CI >= 64 ? 3 : 4))>(in, wp, bias, residual, relu, nbr, nbr_stride, K, n_out, out, in_bytes, stream);
    FD_CASE(16, 16)
    FD_CASE(16, 32)
    FD_CASE(32, 32)
    FD_CASE(32, 64)
    FD_CASE(64, 64)
    FD_CASE(64, 128)
    FD_CASE(128, 128)
#undef FD_CASE
    return 0;
}
}  // namespace fd
