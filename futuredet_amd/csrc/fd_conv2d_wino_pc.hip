// Winograd F(2x2, 3x3) in fp32 with producer and consumer waves (tile 7 of fd_conv2d_wino_nhwc_f32; same arithmetic, same
// packed weights and bit-identical results as the kernels of fd_conv2d_wino.hip).
//
// What bounds the one-role kernels (fd_conv2d_wino.hip, 49 % of the fp32 MFMA rate at best): a wave that owns 16 channels x 16
// tiles needs one 1-KB weight fragment from L2 per 4 MFMAs (128 cycles); four SIMDs -> one vector-memory instruction per 32
// cycles per CU, which is all the CU's texture path issues (tools/probes/gather_probe.hip: 32 B/clk/CU for linear 1-KB loads).
// Holding more tiles per wave halves that stream but costs the occupancy that hid the input transform behind another
// workgroup's MFMAs.  Here the two jobs run on different waves of ONE persistent workgroup per CU instead:
//
//   * waves 0-3 (consumers): 16 output channels x 32 tiles x 16 xi = 128 accumulator registers each; one weight fragment
//     feeds 8 MFMAs (buffer loads with the uniform part of the address in the scalar offset, ring of 8, requested 6 xi-steps
//     ahead); they never touch the input.  Two xi-steps are interleaved (four independent accumulator chains), and the LDS reads
//     and weight requests sit between the groups of four MFMAs, where they issue in the shadow of the 32-cycle MFMA before them;
//   * waves 4-7 (producers): per 16-channel slice, pass 1 turns the patch columns straight from global memory into T = B^T d
//     (LDS [r][x]), pass 2 turns rows of T into V = T B in the MFMA operand layout [xi][tile][16 channels]; T and V are double
//     buffered: while the consumers multiply step g the producers run pass 2 of step g + 1, pass 1 of g + 2 and issue the loads
//     of g + 4.  One barrier per step.  Their arithmetic is packed (v_pk_add_f32, subtraction by the negate modifier);
//   * a work item = 32 CONSECUTIVE tiles of the row-major tile order (a strip may wrap to the next tile row / image: two runs,
//     each with its own halo columns) times 64 output channels: 180 x 180 -> 254 strips (0.3 % padding instead of the 13 % of
//     8 x 8-pixel blocks), 90 x 90 -> 64 strips.  The items are dealt to ceil(items / rounds) workgroups in whole rounds; a
//     workgroup runs its items as ONE pipeline of (items x slices) steps: the producers are already transforming the next item
//     while the consumers store the current one (output transform lane-local, bias from an LDS copy: a global load there would
//     share the vmcnt queue with the weight ring and wait for all of it).
//
// Measured (profiles/round3_dense_fp32_layers.txt, us, one-role tile 6 -> this): 256->128 @180 117 -> 94; 128->128 @180 63 -> 52;
// 256->256 @90 72 -> 48; 512->64 @180 130 -> 88; 64->384 @180 88 -> 87.  Per 16-channel step the consumers need 4750 cycles for
// 128 MFMAs of 32 (tools/wino_pc_trace.py); the rest is the pipeline fill (6 k cycles per workgroup), the epilogue (4 k per item)
// and the barrier.  The clock under this load is ~2.0 GHz.
#include "fd_common.h"

#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// tuning builds (tools/probes/build_trace.sh, -DFD_V2_TRACE): lane 0 of the first consumer / producer wave of every workgroup
// accumulates cycles per phase: [0] consumer multiply, [1] consumer barrier wait, [2] epilogues, [3] producer work, [4] producer
// barrier wait, [5] start, [6] end, [7] prologue (start -> second barrier passed, consumer side)
#ifdef FD_V2_TRACE
__device__ unsigned long long *g_pctrace;
#define FD_PT(var) const unsigned long long var = __builtin_readcyclecounter()
#define FD_PADD(i, v) pacc[i] += (v)
#else
#define FD_PT(var)
#define FD_PADD(i, v)
#endif

struct PcParams {
    int B, H, W, Cin, Cout_pad, Cout_real, cout_total, co_off, relu;
    int tiles_x, tiles_y;
    unsigned x_bytes, w_bytes;
    int n_strips, n_items;
};

constexpr int NTB = 2, NTILE = 16 * NTB;        // tiles per workgroup
constexpr int XCOLS = 2 * NTILE + 4;            // patch columns of a strip: two runs of tiles at most, two halo columns each (68)
constexpr int N1 = XCOLS * 4;                   // pass-1 items (column, channel quad) = 272
constexpr int T_ROW = (XCOLS + XCOLS / 4) * 64; // one row r of T [x][16 channels]; 64 B of padding per 4 columns (pass 2: 4 tiles = stride 2 columns)
constexpr int T_BYTES = 4 * T_ROW;
constexpr int V_BYTES = 16 * NTILE * 64;        // V [xi][tile][16 channels]
constexpr int kMaxBias = 8192;                  // output channels whose bias fits the LDS copy (32 KB)
constexpr int LDS_BYTES = 2 * T_BYTES + 2 * V_BYTES;  // + 4 * Cout_pad for the bias
__device__ __forceinline__ int t_off(int x) { return x * 64 + (x >> 2) * 64; }

// a - b on packed pairs: v_pk_add_f32 with the negate modifier on the second operand (the compiler packs additions but splits
// subtractions into four v_sub_f32; the fp32 MFMAs run on the same pipe as the vector ALU, so every VALU instruction of the
// producers is a slot the consumers' MFMAs do not get).  Exactly a + (-b): bit-identical to the scalar subtraction.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 sub4(f32x4 a, f32x4 b) {
    f32x2 lo, hi;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(__builtin_shufflevector(a, a, 0, 1)), "v"(__builtin_shufflevector(b, b, 0, 1)));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(__builtin_shufflevector(a, a, 2, 3)), "v"(__builtin_shufflevector(b, b, 2, 3)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

__global__ void __launch_bounds__(512) conv2d_wino_pc_f32(const float *__restrict__ x, const float4 *__restrict__ wp, const float *__restrict__ bias,
                                                          float *__restrict__ y, PcParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // T[2] | V[2]
    unsigned char *s_t = smem, *s_v = smem + 2 * T_BYTES;
    float *s_bias = reinterpret_cast<float *>(smem + LDS_BYTES);  // [Cout_pad], zeros beyond Cout: a global load in the epilogue would
                                                                  // share the vmcnt queue with the weight ring and wait for all of it
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nslices = p.Cin / 16;
    const int tpi = p.tiles_x * p.tiles_y, total = p.B * tpi;
    // work items (strip, block of 64 output channels), item = strip + n_strips * channel block; this workgroup takes items
    // blockIdx.x, + gridDim.x, ... and runs them as ONE pipeline of G = items x slices steps: the producers are already
    // transforming the next item's first slices while the consumers finish (and store) the current one
    const int n_mine = (p.n_items - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int G = n_mine * nslices;

    if (wave >= 4) {
        // ------------------------------------------------------------------------------------------------ producers
        const int pt = tid - 256;
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, (int)p.x_bytes, 0x00020000);
        const unsigned row_bytes = (unsigned)p.W * (unsigned)p.Cin * 4u;
        // A strip: tiles [t0, t0 + 32) = run 1 (n1 tiles from (b1, ty1, tx1), to the end of that tile row at most) + run 2 (the rest,
        // from column 0 of the next tile row -- of the next image after the last row); tiles_x >= 32, so there is no third run.
        // pass-1 item = (patch column xg, channel quad q): four patch rows -> the four values of T = B^T d of that column.
        // Bounds-checked buffer loads: rows / columns outside the image (and columns of tiles beyond the last) read zeros.
        unsigned vo0[4], vo1[4];  // byte offsets of the four patch rows of this thread's item(s) in the strip being loaded; kOob = masked
        constexpr unsigned kOob = 0x80000000u;  // >= the buffer size (checked by the launcher) with or without the slice offset added
        const int xg0 = pt >> 2, xg1 = (pt + 256) >> 2, q = pt & 3;
        auto strip_setup = [&](int item) {
            const int t0 = (item % p.n_strips) * NTILE;
            const int b1 = t0 / tpi, ty1 = (t0 % tpi) / p.tiles_x, tx1 = t0 % p.tiles_x;
            const int n1 = min(NTILE, p.tiles_x - tx1), c1 = 2 * n1 + 2;
            const int t2 = t0 + n1;
            const int n2 = t2 < total ? NTILE - n1 : 0;
            const int b2 = t2 / tpi, ty2 = (t2 % tpi) / p.tiles_x;
            auto one = [&](int xg, bool on, unsigned (&vo)[4]) {
                const bool second = xg >= c1;
                const int xr = second ? xg - c1 : xg;
                const bool live = on && (second ? (n2 > 0 && xr < 2 * n2 + 2) : true);
                const int ix = (second ? 0 : 2 * tx1) - 1 + xr, iy0 = 2 * (second ? ty2 : ty1) - 1;
                const unsigned off = (unsigned)((((int64_t)(second ? b2 : b1) * p.H + iy0) * p.W + ix) * p.Cin + q * 4) * 4u;  // may wrap: only used when valid
#pragma unroll
                for (int r = 0; r < 4; ++r) vo[r] = (live && ix >= 0 && ix < p.W && iy0 + r >= 0 && iy0 + r < p.H) ? off + r * row_bytes : kOob;
            };
            one(xg0, item < p.n_items, vo0);
            one(xg1, item < p.n_items && xg1 < XCOLS, vo1);
        };
        int ld_k = 0, ld_s = 0;  // the next step to load = (my item number, slice)
        strip_setup((int)blockIdx.x);
        f32x4 st[2][2][4];  // [step parity][item of the thread][patch row]
        auto load_step = [&](auto PAR) {  // issues the loads of step (ld_k, ld_s) into stage set PAR and advances
            constexpr int par = decltype(PAR)::value;
            // Every call issues the same eight loads on every path (steps beyond the last and the items beyond one per thread --
            // only lanes 0..15 of the first producer wave have a second one -- read out of range): the compiler's s_waitcnt
            // in front of pass 1 can then leave the eight loads of the OTHER parity in flight; with a load under a condition it
            // has to assume they may not exist and waits for everything, which halves the prefetch distance.
#pragma unroll
            for (int r = 0; r < 4; ++r) st[par][0][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo0[r], ld_s * 64, 0));
#pragma unroll
            for (int r = 0; r < 4; ++r) st[par][1][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo1[r], ld_s * 64, 0));
            if (++ld_s == nslices) {
                ld_s = 0;
                ++ld_k;
                strip_setup(ld_k < n_mine ? (int)blockIdx.x + ld_k * (int)gridDim.x : p.n_items);  // past the end: all offsets out of range
            }
        };
        // T0 = d0 - d2, T1 = d1 + d2, T2 = d2 - d1, T3 = d1 - d3 (the operation order of fd_conv2d_wino.hip: bit-identical V)
        const int p1o0 = t_off(xg0) + q * 16, p1o1 = t_off(xg1) + q * 16;
        auto pass1_item = [&](const f32x4 (&d)[4], unsigned char *dst) {
            *reinterpret_cast<f32x4 *>(dst + 0 * T_ROW) = sub4(d[0], d[2]);
            *reinterpret_cast<f32x4 *>(dst + 1 * T_ROW) = d[1] + d[2];
            *reinterpret_cast<f32x4 *>(dst + 2 * T_ROW) = sub4(d[2], d[1]);
            *reinterpret_cast<f32x4 *>(dst + 3 * T_ROW) = sub4(d[1], d[3]);
        };
        auto pass1 = [&](auto PAR, unsigned char *t) {
            constexpr int par = decltype(PAR)::value;
            pass1_item(st[par][0], t + p1o0);
            if (wave == 4 && pt < N1 - 256) pass1_item(st[par][1], t + p1o1);
        };
        // pass-2 item = (row r of T, tile, channel quad), quad and tile fastest: 16 lanes write 256 contiguous bytes of V and read
        // 4 tiles x 64 B of T at a stride of 128 / 192 B (the padding): conflict-free both ways.  Two items per thread; their LDS
        // offsets only change with the strip (n1 = tiles of run 1 moves the first patch column of the tiles of run 2)
        static_assert(NTILE == 32, "item decoding assumes 32 tiles");
        int p2_left = nslices, p2_k = 0;  // slices of the item being pass-2'ed that are still to do
        int so[2][2], dof[2];             // [item][columns 0-1 | 2-3] of T, V offset
        auto p2_setup = [&](int item) {
            const int n1 = min(NTILE, p.tiles_x - ((item % p.n_strips) * NTILE) % p.tiles_x);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int w = pt + i * 256;
                const int tile = (w >> 2) & (NTILE - 1), r = w >> 7;
                const int xs = tile < n1 ? 2 * tile : 2 * tile + 2;
                so[i][0] = r * T_ROW + q * 16 + t_off(xs);
                so[i][1] = r * T_ROW + q * 16 + t_off(xs + 2);
                dof[i] = (r * 4) * NTILE * 64 + tile * 64 + q * 16;
            }
        };
        p2_setup((int)blockIdx.x);
        auto pass2 = [&](const unsigned char *t, unsigned char *v) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x4 d[4];
                d[0] = *reinterpret_cast<const f32x4 *>(t + so[i][0]);
                d[1] = *reinterpret_cast<const f32x4 *>(t + so[i][0] + 64);
                d[2] = *reinterpret_cast<const f32x4 *>(t + so[i][1]);
                d[3] = *reinterpret_cast<const f32x4 *>(t + so[i][1] + 64);
                unsigned char *dst = v + dof[i];
                *reinterpret_cast<f32x4 *>(dst + 0 * NTILE * 64) = sub4(d[0], d[2]);
                *reinterpret_cast<f32x4 *>(dst + 1 * NTILE * 64) = d[1] + d[2];
                *reinterpret_cast<f32x4 *>(dst + 2 * NTILE * 64) = sub4(d[2], d[1]);
                *reinterpret_cast<f32x4 *>(dst + 3 * NTILE * 64) = sub4(d[1], d[3]);
            }
            if (--p2_left == 0) {
                p2_left = nslices;
                if (++p2_k < n_mine) p2_setup((int)blockIdx.x + p2_k * (int)gridDim.x);
            }
        };
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        // step g: loads two steps before its pass 1, pass 1 one step before its pass 2, pass 2 one step before the consumers
        load_step(P0{});
        load_step(P1{});
        pass1(P0{}, s_t);
        load_step(P0{});
        __syncthreads();
        pass2(s_t, s_v);
        pass1(P1{}, s_t + T_BYTES);  // (of step 1: garbage-free zeros when G == 1, never read)
        load_step(P1{});
        __syncthreads();
#ifdef FD_V2_TRACE
        unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
        auto body = [&](auto PAR, int g) {  // while the consumers multiply step g (parity PAR)
            constexpr int par = decltype(PAR)::value;
            FD_PT(q0);
            if (g + 1 < G) pass2(s_t + (1 - par) * T_BYTES, s_v + (1 - par) * V_BYTES);
            pass1(PAR, s_t + par * T_BYTES);  // (beyond the last step: zeros into a buffer nobody reads again)
            load_step(PAR);
            FD_PT(q1);
            __syncthreads();
            FD_PT(q2);
            FD_PADD(3, q1 - q0); FD_PADD(4, q2 - q1);
        };
        for (int g = 0; g < G; g += 2) {
            body(P0{}, g);
            body(P1{}, g + 1);  // (also when g + 1 == G: an idle step, matched by the consumers' extra barrier -- keeps both parities on every path)
        }
#ifdef FD_V2_TRACE
        if (tid == 256 && g_pctrace) { g_pctrace[(size_t)blockIdx.x * 8 + 3] = pacc[3]; g_pctrace[(size_t)blockIdx.x * 8 + 4] = pacc[4]; }
#endif
        return;
    }

    // ---------------------------------------------------------------------------------------------------- consumers
#ifdef FD_V2_TRACE
    unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    FD_PT(tstart);
    const int lm = lane & 15, lq = lane >> 4;
    const unsigned vbase = (unsigned)(lm * 64 + lq * 16);  // this lane's tile (inside a block of 16) and channel quad
    const int total_steps = nslices * 16;
    // packed weights (fd_conv2d_wino_f32_pack_weight): [Cout_pad/16][slice][xi][lane] x 16 bytes
    // Fragment loads are buffer loads with the wave-uniform part of the address in the scalar offset: no vector instruction per
    // load (a VALU instruction of this wave is a slot its MFMAs do not get).  Ring of 8 slots, slot = step % 8: during the pair of
    // xi-steps (2p, 2p+1) the fragments of steps 2p+6, 2p+7 are requested into the slots the previous pair just released.
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(wp), 0, (int)p.w_bytes, 0x00020000);
    auto weights_of = [&](int item) {  // byte offset of the item's first fragment for this wave
        const int nb = (item / p.n_strips) * 4 + wave;
        return __builtin_amdgcn_readfirstlane((nb < (p.Cout_pad >> 4) ? nb : (p.Cout_pad >> 4) - 1) * total_steps * 1024);
    };
    const unsigned lane16 = lane * 16;
    auto fragment = [&](int base, int step) {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, base + step * 1024, 0));
    };
    constexpr int RW = 8, AHEAD = 6;
    float4 bw[RW];
    int wb = weights_of((int)blockIdx.x);
#pragma unroll
    for (int r = 0; r < AHEAD; ++r) bw[r] = fragment(wb, r);  // (total_steps >= 16)
    const bool wide = ((p.cout_total | p.co_off) & 3) == 0;
    for (int c = tid; c < p.Cout_pad; c += 256) s_bias[c] = (bias && c < p.Cout_real) ? bias[c] : 0.f;
    __syncthreads();
    __syncthreads();
    FD_PT(tpro);
    FD_PADD(7, tpro - tstart);
    int g = 0;
    for (int k = 0; k < n_mine; ++k) {
        const int item = (int)blockIdx.x + k * (int)gridDim.x;
        const int wnext = k + 1 < n_mine ? weights_of(item + (int)gridDim.x) : wb;  // the ring runs into the next item's fragments
        const int co = (item / p.n_strips) * 64 + wave * 16 + lq * 4;
        f32x4 acc[16][NTB];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi)
#pragma unroll
            for (int i = 0; i < NTB; ++i) acc[xi][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < nslices; ++s, ++g) {
            const unsigned char *v = s_v + (g & 1) * V_BYTES;
            FD_PT(c0);
            // Two xi-steps at a time: four independent accumulator chains (xi, tile block) are interleaved, so an MFMA's
            // accumulator was written four MFMAs (128 cycles) earlier.  An MFMA occupies the pipe for 32 cycles but issues in 4:
            // the V-fragment reads of the next pair and the weight requests are placed BETWEEN the groups of four MFMAs (fences),
            // where they issue in the shadow of the MFMA before them -- lumped after the 16 MFMAs they cost ~140 cycles per pair.
            float4 a[2][2][NTB];
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < NTB; ++i) a[0][e][i] = *reinterpret_cast<const float4 *>(v + vbase + (e * NTILE + i * 16) * 64);
#pragma unroll
            for (int x2 = 0; x2 < 8; ++x2) {
                const int xi = 2 * x2, step = s * 16 + xi;
#define FD_KSTEP(C)                                                                                                         \
    _Pragma("unroll") for (int e = 0; e < 2; ++e) _Pragma("unroll") for (int i = 0; i < NTB; ++i)                           \
        acc[xi + e][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[(xi + e) % RW].C, a[x2 & 1][e][i].C, acc[xi + e][i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                FD_KSTEP(x)
                __builtin_amdgcn_sched_barrier(0);
                if (x2 + 1 < 8) {
#pragma unroll
                    for (int i = 0; i < NTB; ++i) a[(x2 + 1) & 1][0][i] = *reinterpret_cast<const float4 *>(v + vbase + ((xi + 2) * NTILE + i * 16) * 64);
                }
                __builtin_amdgcn_sched_barrier(0);
                FD_KSTEP(y)
                __builtin_amdgcn_sched_barrier(0);
                if (x2 + 1 < 8) {
#pragma unroll
                    for (int i = 0; i < NTB; ++i) a[(x2 + 1) & 1][1][i] = *reinterpret_cast<const float4 *>(v + vbase + ((xi + 3) * NTILE + i * 16) * 64);
                }
                __builtin_amdgcn_sched_barrier(0);
                FD_KSTEP(z)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int ns = step + e + AHEAD;
                    bw[(xi + e + AHEAD) % RW] = ns < total_steps ? fragment(wb, ns) : fragment(wnext, ns - total_steps);
                }
                __builtin_amdgcn_sched_barrier(0);
                FD_KSTEP(w)
#undef FD_KSTEP
            }
            __builtin_amdgcn_sched_barrier(0);
            FD_PT(c1);
            __syncthreads();
            FD_PT(c2);
            FD_PADD(0, c1 - c0); FD_PADD(1, c2 - c1);
        }
        wb = wnext;
        FD_PT(e0);

        // ---- output transform + epilogue: Y = A^T M A, A^T = [[1,1,1,0],[0,1,-1,-1]] (lane-local; the order of fd_conv2d_wino.hip).
        // Coordinates: the strip's two runs are wave-uniform (scalar divisions), a lane only selects its run.
        const int t0 = (item % p.n_strips) * NTILE;
        const int b1 = t0 / tpi, ty1 = (t0 % tpi) / p.tiles_x, tx1 = t0 % p.tiles_x;
        const int n1 = min(NTILE, p.tiles_x - tx1);
        const int t2 = t0 + n1;
        const int n2 = t2 < total ? NTILE - n1 : 0;
        const int b2 = t2 / tpi, ty2 = (t2 % tpi) / p.tiles_x;
        float *const y1 = y + (((int64_t)b1 * p.H + 2 * ty1) * p.W + 2 * tx1) * p.cout_total + p.co_off;  // pixel of run 1's first tile
        float *const y2 = y + (((int64_t)b2 * p.H + 2 * ty2) * p.W) * p.cout_total + p.co_off;
        const float4 bv = *reinterpret_cast<const float4 *>(s_bias + co);
        // straight-line stores when nothing of the item can fall outside (every tile of the strip exists, no partial tile at the
        // right / bottom edge, the 64 channels exist, 16-byte aligned channel runs): true for all but the last strip of the RPN /
        // head layers.  The general form below has one exec-masked branch per store.
        const bool interior = wide && n1 + n2 == NTILE && !((p.H | p.W) & 1) && (item / p.n_strips) * 64 + 64 <= p.Cout_real;
#pragma unroll
        for (int i = 0; i < NTB; ++i) {
            const int j = i * 16 + lm;
            const bool first = j < n1;
            const int jj = first ? j : j - n1;
            const int oy = 2 * (first ? ty1 : ty2), ox = 2 * (first ? tx1 + jj : jj);
            const bool live = (first || jj < n2) && co < p.Cout_real;
            float *const yb = (first ? y1 : y2) + 2 * jj * p.cout_total + co;
            f32x4 r0[4], r1[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                r0[c] = acc[c][i] + acc[4 + c][i] + acc[8 + c][i];
                r1[c] = sub4(sub4(acc[4 + c][i], acc[8 + c][i]), acc[12 + c][i]);
            }
            f32x4 yv[2][2];
            yv[0][0] = r0[0] + r0[1] + r0[2];
            yv[0][1] = sub4(sub4(r0[1], r0[2]), r0[3]);
            yv[1][0] = r1[0] + r1[1] + r1[2];
            yv[1][1] = sub4(sub4(r1[1], r1[2]), r1[3]);
            if (interior) {  // (wave-uniform)
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        f32x4 v = yv[dy][dx] + f32x4{bv.x, bv.y, bv.z, bv.w};
                        if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                        *reinterpret_cast<f32x4 *>(yb + (dy * p.W + dx) * p.cout_total) = v;
                    }
                continue;
            }
            if (!live) continue;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    if (oy + dy >= p.H || ox + dx >= p.W) continue;
                    float4 v = make_float4(yv[dy][dx][0] + bv.x, yv[dy][dx][1] + bv.y, yv[dy][dx][2] + bv.z, yv[dy][dx][3] + bv.w);
                    if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    float *dst = yb + (dy * p.W + dx) * p.cout_total;
                    if (wide && co + 3 < p.Cout_real) {
                        *reinterpret_cast<float4 *>(dst) = v;
                    } else {
                        dst[0] = v.x;
                        if (co + 1 < p.Cout_real) dst[1] = v.y;
                        if (co + 2 < p.Cout_real) dst[2] = v.z;
                        if (co + 3 < p.Cout_real) dst[3] = v.w;
                    }
                }
        }
        FD_PT(e1);
        FD_PADD(2, e1 - e0);
    }
    if (G & 1) __syncthreads();  // the producers run their steps in pairs
#ifdef FD_V2_TRACE
    if (tid == 0 && g_pctrace) {
        unsigned long long *o = g_pctrace + (size_t)blockIdx.x * 8;
        o[0] = pacc[0]; o[1] = pacc[1]; o[2] = pacc[2]; o[5] = tstart; o[6] = __builtin_readcyclecounter(); o[7] = pacc[7];
    }
#endif
}

}  // namespace

#ifdef FD_V2_TRACE
extern "C" int fd_debug_set_wino_pc_trace(void *p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_pctrace), &p, sizeof(p)) == hipSuccess ? 0 : -1; }
#endif

namespace fd {

// 0 = launched, 1 = shape not supported by this variant (narrower than one strip, or an input / weight tensor of 2 GB and more), -1 = LDS refused
int wino_pc_launch(const float *x, const void *wp, const float *bias, float *y, int B, int H, int W, int cin, int cout, int relu, int cout_total,
                   int co_off, hipStream_t stream) {
    PcParams p;
    p.B = B; p.H = H; p.W = W; p.Cin = cin;
    p.Cout_real = cout;
    p.Cout_pad = (cout + 63) / 64 * 64;
    p.cout_total = cout_total; p.co_off = co_off; p.relu = relu;
    p.tiles_x = (W + 1) / 2;
    p.tiles_y = (H + 1) / 2;
    const int64_t xb = (int64_t)B * H * W * cin * 4;
    const int64_t wbytes = (int64_t)p.Cout_pad * cin * 16 * 4;
    if (p.tiles_x < NTILE || xb >= 0x80000000ll || wbytes >= 0x80000000ll || p.Cout_pad > kMaxBias) return 1;
    p.x_bytes = (unsigned)xb;
    p.w_bytes = (unsigned)wbytes;
    static std::atomic<uint64_t> lds_set{0};
    const size_t lds = (size_t)LDS_BYTES + (size_t)p.Cout_pad * 4;
    if (!fd::ensure_dynamic_lds(reinterpret_cast<const void *>(conv2d_wino_pc_f32), (size_t)LDS_BYTES + kMaxBias * 4, lds_set)) return -1;  // (the limit, once per device)
    const int64_t n_strips = ((int64_t)p.tiles_x * p.tiles_y * B + NTILE - 1) / NTILE, n_items = n_strips * ((cout + 63) / 64);
    if (n_items >= (1ll << 31)) return 1;
    p.n_strips = (int)n_strips;
    p.n_items = (int)n_items;
    // one workgroup per CU (107 KB of LDS); the items are dealt in whole rounds: ceil(items / rounds) workgroups, each takes <= rounds
    const int n_cu = fd::device_cu_count();
    const int rounds = (int)((n_items + n_cu - 1) / n_cu);
    const unsigned grid = (unsigned)((n_items + rounds - 1) / rounds);
    hipLaunchKernelGGL(conv2d_wino_pc_f32, dim3(grid), dim3(512), lds, stream, x, (const float4 *)wp, bias, y, p);
    return 0;
}

}  // namespace fd
