// CenterPoint decode + rotated NMS for gfx950, fully on the device (no mask download, no host sweep).
//
// Replaces, per (sample, heat-map) group:
//   decode            det3d/models/bbox_heads/center_head.py:621-664   sigmoid / exp / atan2 / cell grid
//   post_processing   center_head.py:699-717                           score + centre-range mask
//   rotate_nms_pcdet  det3d/core/bbox/box_torch_ops.py:248-277          sort desc, pre_max, layout swap
//   nms_gpu           det3d/ops/iou3d_nms/src/iou3d_nms.cpp:90-135 + iou3d_nms_kernel.cu:104-311
// Stages: (1) score keys for every BEV cell, (2) one workgroup per group radix-selects the pre_max best
// keys (3 LDS-histogram passes), compacts them in cell order and bitonic-sorts them in LDS by
// (score desc, cell asc); only those <= pre_max cells get the exp/atan2 box decode, (3) upper-triangular
// 64x64 IoU bit-mask tiles, column boxes staged in LDS, (4) one wave per group runs the greedy sweep:
// the 64 rows of a diagonal tile are resolved with readlane broadcasts, kept rows are OR-ed into a
// lane-distributed "removed" vector with one coalesced 8*col_blocks-byte load each; stops at post_max.
// Compiled with -ffp-contract=off: geometry is evaluated in the reference's operation order.
#include "fd_common.h"

namespace {

constexpr int kSelThreads = 1024;
constexpr int kMaxPre = 4096;  // nms_pre_max_size bound (64 lanes x 64 bits in the sweep)

// test_cfg.circular_nms (center_head.py:722-725, core/utils/circle_nms_jit.py): the suppression predicate of decode group g is
// "squared centre distance <= r[g / per]" instead of the rotated IoU; per = 0: rotated NMS
struct CircleCfg {
    int per;
    float r[16];
};

// ---------------------------------------------------------------- rotated BEV IoU
// Same quantity as det3d/ops/iou3d_nms/src/iou3d_nms_kernel.cu:104-234 (overlap polygon of two rotated rectangles =
// edge/edge crossing points + corners of one lying inside the other, ordered by angle around their mean, fan area),
// organised differently:
//   * everything that depends on ONE box -- sin/cos of the yaw, the four rotated corners, half extents, area, the
//     half diagonal used for far-pair rejection -- is computed once per box into a 64-byte Footprint (the reference
//     recomputes it for every pair: 10 sincos + 8 corner rotations per pair, here none);
//   * an edge/edge test evaluates four "side" values (signed area of a point against an edge line through its origin
//     and precomputed direction) instead of five cross products: side(q0), side(q1) against A's edge and side(p0),
//     side(p1) against B's edge.  Two of the reference's products are exact negations of these
//     (fl(ab - cd) = -fl(cd - ab)), so the strict-crossing decision and the crossing point are bit-identical;
//   * the angular order is an insertion sort on angles computed once per point (the reference bubble-sorts with two
//     atan2f per comparison); both sorts are stable with the same strict comparison, hence the same order.
// What is deliberately kept, because tests/golden/iou.npz (the compiled reference) and the NMS decisions at the
// threshold depend on rounding: the order in which points enter the list (it is the summation order of the mean),
// the operation order inside each expression, float32 throughout, no fma contraction (-ffp-contract=off).
struct Footprint {
    float vx[4], vy[4];  // rotated corners, counter-clockwise from (-hx,-hy)
    float cx, cy, co, si; // centre, cos / sin of the yaw
    float hx, hy, area, reach;  // half extents, w*l, half diagonal
};

__device__ inline Footprint make_footprint(const float *b /*[x,y,z,dx,dy,dz,yaw]*/) {
    Footprint f;
    f.cx = b[0]; f.cy = b[1];
    f.hx = b[3] / 2; f.hy = b[4] / 2;
    f.co = cosf(b[6]); f.si = sinf(b[6]);
    f.area = b[3] * b[4];
    f.reach = 0.5f * sqrtf(b[3] * b[3] + b[4] * b[4]);
    const float lo_x = f.cx - f.hx, hi_x = f.cx + f.hx, lo_y = f.cy - f.hy, hi_y = f.cy + f.hy;
    const float ux[4] = {lo_x, hi_x, hi_x, lo_x}, uy[4] = {lo_y, lo_y, hi_y, hi_y};
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // rotation about the centre: (dx*c - dy*s, dx*s + dy*c) + centre
        const float dx = ux[k] - f.cx, dy = uy[k] - f.cy;
        f.vx[k] = dx * f.co + dy * (-f.si) + f.cx;
        f.vy[k] = dx * f.si + dy * f.co + f.cy;
    }
    return f;
}

// signed area of (point - origin) against direction d: > 0 on one side of the line, < 0 on the other
__device__ inline float side_of(float px, float py, float ox, float oy, float dx, float dy) { return (px - ox) * dy - dx * (py - oy); }

// is (px,py) inside rectangle f grown by the 1e-2 margin?  (un-rotate about the centre, compare with the half extents)
__device__ inline bool inside_margin(const Footprint &f, float px, float py) {
    const float dx = px - f.cx, dy = py - f.cy;
    const float u = dx * f.co + dy * f.si;   // cos(-yaw) = co, -sin(-yaw) = si
    const float v = dy * f.co - dx * f.si;   // dx*sin(-yaw) + dy*cos(-yaw), same rounding
    return fabsf(u) < f.hx + 1e-2f && fabsf(v) < f.hy + 1e-2f;
}

constexpr int kMaxPoly = 24;  // 16 edge pairs can cross + 8 corners can be inside (never all at once)
// The polygon's points and angles are indexed at run time (insertion sort over n points): they live in LDS -- element k of a thread at
// [k * kPolyThreads + tid], conflict-free -- rather than in thread-private arrays, which would be scratch memory (208 bytes per lane).
//
// THIS FILE IS COMPILED WITH -fno-slp-vectorize (futuredet_amd/build.py).  With the SLP vectoriser on, the geometry below becomes ~500
// packed-fp32 instructions per evaluation, 90 of them with an op_sel swizzle of src1 (v_pk_mul_f32 / v_pk_add_f32 ... op_sel:[0,1]) -- and
// that form returns wrong values in lanes 48-63 of a wave while waves of the bf16 3x3 dense convolution (v_mfma_f32_32x32x16_bf16) of
// another stream share the compute unit: with several bf16 sweeps in flight 0.5-3 % of the sweeps came back with a different detection
// list (IoU 0.34 judged "no overlap", 0.02 judged "overlap") while every tensor nms_mask reads was bit-identical; never with one sweep in
// flight, never in fp32.  Found by tools/soak_determinism.py, narrowed with a self-checking build (-DFD_MASK_DEBUG: two evaluations of one
// thread on footprints pinned in registers disagree, lanes 48-63 only), tools/soak_pairs.py (the partner is the bf16 neck plan) and
// tools/soak_alu.py (inline-asm chains: ONLY the op_sel:[0,1] forms fail; plain, neg, op_sel_hi, v_pk_fma_f32 and scalar arithmetic are
// exact).  Without the vectoriser: same IEEE operations, bit-identical results, same 24 us, 0 of 18 000 sweeps differing.
// profiles/round6_determinism_soak.txt; guards: tests/test_gpu_parity.py::test_bf16_sweeps_in_flight_are_deterministic (-m gpu) and
// tests/test_host_surface.py::test_no_packed_fp32_instruction_selects_the_high_half_of_src1 (disassembles the built library).
constexpr int kPolyThreads = 128;                              // workgroup size of the kernels that call footprint_overlap
constexpr int kPolyLdsFloats = 3 * kMaxPoly * kPolyThreads;    // px | py | ang: 36 KB

#ifdef FD_MASK_DEBUG  // tuning build (tools/probes/build_maskdbg.sh): digests of one evaluation, so two evaluations of the same inputs can be compared
__device__ int g_maskdbg[8];
__device__ int g_masklog_n;
__device__ unsigned g_masklog[64 * 16];
struct EvalDigest {
    int n;                // points
    unsigned pts, angs;   // xor over the points' / the angles' bit patterns
    float area;
    unsigned made;        // bit 4 i + j: edge pair (i, j) crossed; bit 16 + 2 k (+ 1): corner k of B (of A) inside the other
    unsigned sides;       // xor over the four side values of every edge pair that passed the extent test
};
#define FD_DG_ARG , EvalDigest *dg = nullptr
#define FD_DG(stmt) do { if (dg) { stmt; } } while (0)
#define FD_DG_PT(x, y) FD_DG(dg->pts ^= __float_as_uint(x) * 3u ^ __float_as_uint(y))
#else
#define FD_DG_ARG
#define FD_DG(stmt)
#define FD_DG_PT(x, y)
#endif
__device__ float footprint_overlap(const Footprint &A, const Footprint &B, float *s_poly /* + threadIdx.x */ FD_DG_ARG) {
    FD_DG(*dg = EvalDigest{});
    auto px = [&](int k) -> float & { return s_poly[k * kPolyThreads]; };
    auto py = [&](int k) -> float & { return s_poly[(kMaxPoly + k) * kPolyThreads]; };
    auto ang = [&](int k) -> float & { return s_poly[(2 * kMaxPoly + k) * kPolyThreads]; };
    float sum_x = 0.f, sum_y = 0.f;
    int n = 0;
    // ---- crossing points, A's edges outer, B's edges inner
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float ax = A.vx[i], ay = A.vy[i], ax1 = A.vx[(i + 1) & 3], ay1 = A.vy[(i + 1) & 3];
        const float ex = ax1 - ax, ey = ay1 - ay;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float bx = B.vx[j], by = B.vy[j], bx1 = B.vx[(j + 1) & 3], by1 = B.vy[(j + 1) & 3];
            // the axis-aligned extents of the two edges must overlap (this also shields the sign tests below from
            // rounding when the edges are far apart but nearly collinear)
            if (!(fminf(ax, ax1) <= fmaxf(bx, bx1) && fminf(bx, bx1) <= fmaxf(ax, ax1) && fminf(ay, ay1) <= fmaxf(by, by1) &&
                  fminf(by, by1) <= fmaxf(ay, ay1)))
                continue;
            const float fx = bx1 - bx, fy = by1 - by;
            const float a0 = side_of(bx, by, ax, ay, ex, ey), a1 = side_of(bx1, by1, ax, ay, ex, ey);   // B's end points against A's edge
            const float b0 = side_of(ax, ay, bx, by, fx, fy), b1 = side_of(ax1, ay1, bx, by, fx, fy);   // A's end points against B's edge
            FD_DG(dg->sides ^= (__float_as_uint(a0) * 3u) ^ (__float_as_uint(a1) * 5u) ^ (__float_as_uint(b0) * 7u) ^ (__float_as_uint(b1) * 11u) ^ (unsigned)(4 * i + j));
            if (!(a0 * (-a1) > 0.f && b0 * (-b1) > 0.f)) continue;  // both pairs strictly on opposite sides
            float qx, qy;
            const float den = a1 - a0;
            if (fabsf(den) > 1e-8f) {  // point dividing B's edge in the ratio of the two side values
                qx = (a1 * bx - a0 * bx1) / den;
                qy = (a1 * by - a0 * by1) / den;
            } else {  // degenerate ratio: intersect the two implicit lines l*x + m*y + c = 0 by Cramer's rule
                const float la = ay - ay1, ma = ax1 - ax, ca = ax * ay1 - ax1 * ay;
                const float lb = by - by1, mb = bx1 - bx, cb = bx * by1 - bx1 * by;
                const float det = la * mb - lb * ma;
                qx = (ma * cb - mb * ca) / det;
                qy = (lb * ca - la * cb) / det;
            }
            px(n) = qx; py(n) = qy; ++n;
            FD_DG_PT(qx, qy);
            FD_DG(dg->made |= 1u << (4 * i + j));
            sum_x = sum_x + qx; sum_y = sum_y + qy;
        }
    }
    // ---- corners of one rectangle inside the other, alternating B's k-th and A's k-th
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (inside_margin(A, B.vx[k], B.vy[k])) {
            sum_x = sum_x + B.vx[k]; sum_y = sum_y + B.vy[k];
            px(n) = B.vx[k]; py(n) = B.vy[k]; ++n;
            FD_DG_PT(B.vx[k], B.vy[k]);
            FD_DG(dg->made |= 1u << (16 + 2 * k));
        }
        if (inside_margin(B, A.vx[k], A.vy[k])) {
            sum_x = sum_x + A.vx[k]; sum_y = sum_y + A.vy[k];
            px(n) = A.vx[k]; py(n) = A.vy[k]; ++n;
            FD_DG_PT(A.vx[k], A.vy[k]);
            FD_DG(dg->made |= 1u << (17 + 2 * k));
        }
    }
    FD_DG(dg->n = n);
    if (n < 3) return 0.f;  // no polygon (the fan below is empty or degenerate: area exactly 0)
    const float mx = sum_x / n, my = sum_y / n;
    // ---- angular order around the mean point: stable insertion sort, ascending, strict comparison
    for (int k = 0; k < n; ++k) ang(k) = atan2f(py(k) - my, px(k) - mx);
    FD_DG(for (int k = 0; k < n; ++k) dg->angs ^= __float_as_uint(ang(k)) * (unsigned)(2 * k + 1));
    for (int k = 1; k < n; ++k) {
        const float a = ang(k), x = px(k), y = py(k);
        int m = k;
        while (m > 0 && ang(m - 1) > a) {
            ang(m) = ang(m - 1); px(m) = px(m - 1); py(m) = py(m - 1);
            --m;
        }
        ang(m) = a; px(m) = x; py(m) = y;
    }
    // ---- fan around the first vertex
    float twice = 0.f;
    const float x0 = px(0), y0 = py(0);
    for (int k = 0; k < n - 1; ++k) {
        const float ux = px(k) - x0, uy = py(k) - y0, wx = px(k + 1) - x0, wy = py(k + 1) - y0;
        twice += ux * wy - uy * wx;
    }
    FD_DG(dg->area = fabsf(twice) * 0.5f);
    return fabsf(twice) * 0.5f;
}

__device__ inline float footprint_iou(const Footprint &A, const Footprint &B, float *s_poly) {
    const float ov = footprint_overlap(A, B, s_poly);
    return ov / fmaxf(A.area + B.area - ov, 1e-8f);
}

// one Footprint per box, stored as four float4 planes ([plane][box]) so a wave's 64 column boxes load coalesced
__device__ inline void store_footprint(float4 *planes, int64_t n_total, int64_t i, const Footprint &f) {
    planes[i] = make_float4(f.vx[0], f.vy[0], f.vx[1], f.vy[1]);
    planes[n_total + i] = make_float4(f.vx[2], f.vy[2], f.vx[3], f.vy[3]);
    planes[2 * n_total + i] = make_float4(f.cx, f.cy, f.co, f.si);
    planes[3 * n_total + i] = make_float4(f.hx, f.hy, f.area, f.reach);
}
__device__ inline Footprint load_footprint(const float4 *planes, int64_t n_total, int64_t i) {
    const float4 p0 = planes[i], p1 = planes[n_total + i], p2 = planes[2 * n_total + i], p3 = planes[3 * n_total + i];
    Footprint f;
    f.vx[0] = p0.x; f.vy[0] = p0.y; f.vx[1] = p0.z; f.vy[1] = p0.w;
    f.vx[2] = p1.x; f.vy[2] = p1.y; f.vx[3] = p1.z; f.vy[3] = p1.w;
    f.cx = p2.x; f.cy = p2.y; f.co = p2.z; f.si = p2.w;
    f.hx = p3.x; f.hy = p3.y; f.area = p3.z; f.reach = p3.w;
    return f;
}

__global__ void __launch_bounds__(256) footprint_kernel(const float *__restrict__ boxes, const int *__restrict__ counts, int n_max, int G,
                                                        float4 *__restrict__ planes) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)G * n_max) return;
    const int g = (int)(t / n_max), i = (int)(t - (int64_t)g * n_max);
    if (counts && i >= counts[g]) return;
    float b[7];
#pragma unroll
    for (int d = 0; d < 7; ++d) b[d] = boxes[t * 7 + d];
    store_footprint(planes, (int64_t)G * n_max, t, make_footprint(b));
}

// ---------------------------------------------------------------- stage 1: score keys
constexpr int kMaxSteps = 16;
struct DecCfg {
    int H, W, HW;
    float osf, vx, vy, px, py, thr;
    float rng[6];
    float iou_thr;
    int pre_max, post_max;
    int hm_channels;
};

// A head map as the kernels read it: element (group g, channel ch, BEV cell) of a float32 or bf16 tensor at
// base[g * gs + ch * cs + cell * ps].  NCHW planes: cs = H*W, ps = 1; a channel slice of an NHWC head output: cs = 1, ps = C_total
// -- which is how the decode reads the convolution plan's output in place (no .float() / .contiguous() passes in between).
struct MapView {
    const void *p;
    int64_t gs, cs, ps;
    int bf16;
};
__device__ inline float ld(const MapView &m, int g, int ch, int cell) {
    const int64_t o = (int64_t)g * m.gs + (int64_t)ch * m.cs + (int64_t)cell * m.ps;
    return m.bf16 ? __uint_as_float(((unsigned)reinterpret_cast<const unsigned short *>(m.p)[o]) << 16) : reinterpret_cast<const float *>(m.p)[o];
}
inline MapView as_view(const fd_map_view &v) { return MapView{v.data, v.group_stride, v.channel_stride, v.cell_stride, v.dtype}; }

__device__ inline void cell_center(const DecCfg &c, const MapView &reg, int g, int cell, float &x, float &y) {
    // center_head.py:641-649: xs = (j + reg_x) * out_size_factor * voxel_size[0] + pc_range[0], left to right in f32
    const int i = cell / c.W, j = cell - i * c.W;
    float xs = __fadd_rn((float)j, ld(reg, g, 0, cell));
    float ys = __fadd_rn((float)i, ld(reg, g, 1, cell));
    x = __fadd_rn(__fmul_rn(__fmul_rn(xs, c.osf), c.vx), c.px);
    y = __fadd_rn(__fmul_rn(__fmul_rn(ys, c.osf), c.vy), c.py);
}

__global__ void __launch_bounds__(256) dec_keys(MapView hm, MapView reg, MapView height, DecCfg c, unsigned *__restrict__ keys) {
    const int g = blockIdx.y;
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= c.HW) return;
    float logit = ld(hm, g, 0, cell);
    for (int ch = 1; ch < c.hm_channels; ++ch) logit = fmaxf(logit, ld(hm, g, ch, cell));  // center_head.py:592: torch.max over the class channels
    const float score = 1.0f / (1.0f + expf(-logit));
    float x, y;
    cell_center(c, reg, g, cell, x, y);
    const float z = ld(height, g, 0, cell);
    const bool ok = score > c.thr && x >= c.rng[0] && y >= c.rng[1] && z >= c.rng[2] && x <= c.rng[3] && y <= c.rng[4] && z <= c.rng[5];
    keys[(int64_t)g * c.HW + cell] = ok ? __float_as_uint(score) : 0u;
}

// ---------------------------------------------------------------- stage 2: select + sort + box decode
__device__ inline int block_sum_1024(int v, int *sm) {
    // returns exclusive prefix in thread order; total in sm[16]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int iv = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int u = __shfl_up(iv, off);
        if (lane >= off) iv += u;
    }
    if (lane == 63) sm[wave] = iv;
    __syncthreads();
    int wo = 0, tot = 0;
    for (int w = 0; w < kSelThreads / 64; ++w) {
        if (w < wave) wo += sm[w];
        tot += sm[w];
    }
    __syncthreads();
    if (threadIdx.x == 0) sm[16] = tot;
    __syncthreads();
    return wo + iv - v;
}

// finds the bin (scanning from the top) where the cumulative count reaches `need`; hist has nbins (<= 4096)
// entries.  Parallel: thread t owns the 4 bins nbins-1-4t .. nbins-4-4t, a block scan gives the count above.
template <int Q = 4>  // bins per thread: nbins <= 1024 Q
__device__ inline void find_bin(const int *hist, int nbins, int need, int *sm_scan, int *sm_out /*[2]: bin, count above*/) {
    __syncthreads();
    int h[Q];
    int local = 0;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int bin = nbins - 1 - (Q * (int)threadIdx.x + q);
        h[q] = bin >= 0 ? hist[bin] : 0;
        local += h[q];
    }
    int acc = block_sum_1024(local, sm_scan);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int bin = nbins - 1 - (Q * (int)threadIdx.x + q);
        if (bin >= 0 && acc < need && acc + h[q] >= need) {
            sm_out[0] = bin;
            sm_out[1] = acc;
        }
        acc += h[q];
    }
    __syncthreads();
}

__global__ void __launch_bounds__(kSelThreads) dec_select(const unsigned *__restrict__ keys_all, MapView reg, MapView height, MapView dim, MapView rot,
                                                          DecCfg c, int npad,
                                                          float *__restrict__ sel_boxes /*[G,pre,7] output layout*/,
                                                          float *__restrict__ nms_boxes /*[G,pre,7] pcdet layout*/,
                                                          float *__restrict__ sel_scores, int *__restrict__ sel_cell,
                                                          int *__restrict__ sel_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long *s_sort = reinterpret_cast<unsigned long long *>(smem);     // [npad]
    int *s_hist = reinterpret_cast<int *>(smem + (size_t)npad * 8);                // [4096]
    int *s_misc = s_hist + 4096;                                                   // [32]
    const int g = blockIdx.x;
    const unsigned *keys = keys_all + (int64_t)g * c.HW;
    const int tid = threadIdx.x;

    // valid count
    int cnt = 0;
    for (int i = tid; i < c.HW; i += kSelThreads) cnt += keys[i] != 0u;
    block_sum_1024(cnt, s_misc);
    const int M = s_misc[16];
    __syncthreads();
    unsigned T = 1u;  // select keys >= T ... refined below
    int take_eq = 0x7fffffff;
    if (M > c.pre_max) {
        unsigned prefix = 0u, pmask = 0u;
        int need = c.pre_max;
        const int shifts[3] = {20, 8, 0};
        const int bits[3] = {12, 12, 8};
        for (int pass = 0; pass < 3; ++pass) {
            const int nb = 1 << bits[pass];
            for (int i = tid; i < nb; i += kSelThreads) s_hist[i] = 0;
            __syncthreads();
            for (int i = tid; i < c.HW; i += kSelThreads) {
                unsigned k = keys[i];
                if (k != 0u && (k & pmask) == prefix) atomicAdd(&s_hist[(k >> shifts[pass]) & (nb - 1)], 1);
            }
            find_bin(s_hist, nb, need, s_misc, s_misc + 20);
            const int b = s_misc[20], above = s_misc[21];
            prefix |= (unsigned)b << shifts[pass];
            pmask |= (unsigned)(nb - 1) << shifts[pass];
            need -= above;
            __syncthreads();
        }
        T = prefix;
        take_eq = need;  // how many keys == T to take, in cell order
    }
    // ordered compaction: keys > T all, keys == T first take_eq (cell order).  (M <= pre_max: T=1 -> all valid)
    for (int i = tid; i < npad; i += kSelThreads) s_sort[i] = 0ull;
    __syncthreads();
    const bool all_valid = (M <= c.pre_max);
    {
        // thread t owns the contiguous cells [t * per, (t + 1) * per): one block scan over the per-thread counts gives every
        // thread its first output position in cell order (the previous version scanned 1024 cells at a time: HW / 1024
        // block scans with three barriers each were most of this kernel's 85 us)
        const int per = (c.HW + kSelThreads - 1) / kSelThreads;
        const int i0 = tid * per, i1 = min(c.HW, i0 + per);
        int n_gt = 0, n_eq = 0;
        for (int i = i0; i < i1; ++i) {
            const unsigned k = keys[i];
            n_gt += all_valid ? (k != 0u) : (k > T);
            n_eq += all_valid ? 0 : (k == T);
        }
        // (two scans: the count of keys == T can reach HW, so the two counts do not pack into one 32-bit scan)
        int pos_gt = block_sum_1024(n_gt, s_misc);
        __syncthreads();
        int rank_eq = block_sum_1024(n_eq, s_misc);
        __syncthreads();
        for (int i = i0; i < i1; ++i) {
            const unsigned k = keys[i];
            const bool fgt = all_valid ? (k != 0u) : (k > T);
            const bool feq = !all_valid && (k == T);
            // slots [0, n_gt) hold the > T keys in cell order; the == T keys fill the window from the back of pre_max
            if (fgt) s_sort[pos_gt++] = ((unsigned long long)k << 32) | (unsigned)(0xffffffffu - (unsigned)i);
            else if (feq) {
                if (rank_eq < take_eq) s_sort[c.pre_max - 1 - rank_eq] = ((unsigned long long)k << 32) | (unsigned)(0xffffffffu - (unsigned)i);
                ++rank_eq;
            }
        }
    }
    __syncthreads();
    const int S = all_valid ? M : c.pre_max;
    // bitonic sort descending over npad entries (zeros sink to the end)
    for (int k2 = 2; k2 <= npad; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < npad; t += kSelThreads) {
                int ixj = t ^ j;
                if (ixj > t) {
                    unsigned long long a = s_sort[t], b = s_sort[ixj];
                    bool desc = ((t & k2) == 0);
                    if (desc ? (a < b) : (a > b)) { s_sort[t] = b; s_sort[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    if (tid == 0) { sel_count[g] = S; sel_count[gridDim.x + g] = M; }  // [G] taken, [G] valid candidates
    // box decode for the selected cells only
    for (int i = tid; i < S; i += kSelThreads) {
        const unsigned long long e = s_sort[i];
        const int cell = (int)(0xffffffffu - (unsigned)(e & 0xffffffffull));
        const float score = __uint_as_float((unsigned)(e >> 32));
        float x, y;
        cell_center(c, reg, g, cell, x, y);
        const float z = ld(height, g, 0, cell);
        const float d0 = expf(ld(dim, g, 0, cell)), d1 = expf(ld(dim, g, 1, cell)), d2 = expf(ld(dim, g, 2, cell));
        const float yaw = atan2f(ld(rot, g, 0, cell), ld(rot, g, 1, cell));
        const int64_t o = ((int64_t)g * c.pre_max + i);
        float *sb = sel_boxes + o * 7;
        sb[0] = x; sb[1] = y; sb[2] = z; sb[3] = d0; sb[4] = d1; sb[5] = d2; sb[6] = yaw;
        float *nb = nms_boxes + o * 7;  // box_torch_ops.py:256-257: [x,y,z,dim1,dim0,dim2,-yaw-pi/2]
        nb[0] = x; nb[1] = y; nb[2] = z; nb[3] = d1; nb[4] = d0; nb[5] = d2;
        nb[6] = __fsub_rn(-yaw, 1.5707963267948966f);
        sel_scores[o] = score;
        sel_cell[o] = cell;
    }
}

// ---------------------------------------------------------------- stages 1-2, round 5: histogram with the keys, selection without a sort
// The round-4 selection (one workgroup per group: four radix passes over the keys, an LDS bitonic sort of the pre_max survivors, the box
// decode) took 52 us of a sweep's 113-us back end.  Now:
//   dec_keys_hist   every block of 1024 cells writes its keys AND a 1024-bin histogram of them -- bins of equal width over the only range a
//                   key can lie in, (score_threshold, 1.0] as float32 bit patterns -- so the selection starts from the histogram;
//   dec_select_hist one workgroup per group sums the block histograms, finds the bin in which the pre_max-th best key lies, takes every key
//                   above that bin, and resolves the (few dozen) keys INSIDE it by ranking them against each other (all pairs, LDS broadcast);
//                   a degenerate map (thousands of equal scores) takes three radix passes over the remaining 37 bits instead.  The
//                   survivors are written UNSORTED as (score bits << 32 | ~cell) words;
//   dec_rank_decode the order is a rank: element i goes to slot #{j : word_j > word_i} (the words are unique) -- spread over
//                   ceil(pre_max / 256) workgroups per group, four threads per element; the thread that owns the element decodes its box
//                   (exp / atan2) and writes box, pcdet-layout footprint, score and cell at the slot.
// Same result as before by construction: the order is (score descending, cell ascending), ties included.
struct KeyBins {
    unsigned base;   // keys are > base (the threshold's bit pattern; 0 for a threshold <= 0)
    int shift;       // bin of key k = (k - base) >> shift, < 1024
};
inline KeyBins make_key_bins(float thr) {
    KeyBins kb;
    union { float f; unsigned u; } c;
    c.f = thr > 0.f ? thr : 0.f;
    kb.base = c.u;
    const unsigned range = 0x3f800000u > kb.base ? 0x3f800000u - kb.base : 1u;  // scores are sigmoids: <= 1.0f
    int bits = 0;
    while ((range >> bits) != 0u) ++bits;
    kb.shift = bits > 10 ? bits - 10 : 0;
    return kb;
}
__device__ inline int key_bin(const KeyBins &kb, unsigned k) {
    const unsigned d = (k - kb.base) >> kb.shift;
    return (int)(d < 1023u ? d : 1023u);  // (a key above 1.0f cannot occur; clamped rather than trusted)
}

// LDS histogram update that survives concentrated keys.  A plain atomicAdd per lane serialises on the bank when many lanes hit one bin --
// and heat-map scores DO concentrate (random-weight and saturated maps: all keys within a few bins): 64 lanes on one address cost 64 LDS
// cycles per instruction, which is where the 52 us of round 4's four radix passes went.  Up to four rounds of "the first active lane's bin:
// one lane adds the number of lanes that share it"; lanes left after that (spread keys: little contention anyway) add for themselves.
__device__ inline void hist_add(int *hist, int bin, bool active) {
    unsigned long long todo = __ballot(active);
#pragma unroll 1
    for (int round = 0; round < 4 && todo; ++round) {
        const int leader = __builtin_ctzll(todo);
        const int b0 = __builtin_amdgcn_readlane(bin, leader);  // (leader is wave-uniform)
        const unsigned long long same = __ballot(active && bin == b0) & todo;
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[b0], (int)__builtin_popcountll(same));
        todo &= ~same;
    }
    if (active && ((todo >> (threadIdx.x & 63)) & 1ull)) atomicAdd(&hist[bin], 1);
}

constexpr int kKeyBlock = 1024;  // cells per block of dec_keys_hist
__global__ void __launch_bounds__(256) dec_keys_hist(MapView hm, MapView reg, MapView height, DecCfg c, KeyBins kb, unsigned *__restrict__ keys,
                                                     int *__restrict__ hist_part /*[G][blocks][1024]*/) {
    __shared__ int s_hist[1024];
    const int g = blockIdx.y, tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) s_hist[tid + 256 * i] = 0;
    __syncthreads();
    unsigned kk[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cell = blockIdx.x * kKeyBlock + tid + 256 * i;
        if (cell < c.HW) {
            float logit = ld(hm, g, 0, cell);
            for (int ch = 1; ch < c.hm_channels; ++ch) logit = fmaxf(logit, ld(hm, g, ch, cell));  // center_head.py:592: torch.max over the class channels
            const float score = 1.0f / (1.0f + expf(-logit));
            float x, y;
            cell_center(c, reg, g, cell, x, y);
            const float z = ld(height, g, 0, cell);
            const bool ok = score > c.thr && x >= c.rng[0] && y >= c.rng[1] && z >= c.rng[2] && x <= c.rng[3] && y <= c.rng[4] && z <= c.rng[5];
            const unsigned k = ok ? __float_as_uint(score) : 0u;
            keys[(int64_t)g * c.HW + cell] = k;
            kk[i] = k;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) hist_add(s_hist, kk[i] ? key_bin(kb, kk[i]) : 0, kk[i] != 0u);  // (outside the divergent part: the helper uses wave ballots)
    __syncthreads();
    int *dst = hist_part + ((int64_t)g * gridDim.x + blockIdx.x) * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[tid + 256 * i] = s_hist[tid + 256 * i];
}

#ifdef FD_DEC_TRACE  // tuning builds: thread 0 of dec_select_hist stamps its phases (tools/decode_trace.py)
__device__ unsigned long long *g_dectrace;
#define FD_DT(i) do { if (threadIdx.x == 0 && g_dectrace) g_dectrace[blockIdx.x * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define FD_DT(i)
#endif
constexpr int kMaxCand = 1024;  // keys of the threshold bin ranked against each other in LDS; more take the radix passes
// (the keys of the group live in LDS for the whole kernel -- 130 KB for the 180 x 180 map -- and every phase is a short loop over them: the
//  round-4 kernel and the first version of this one held them in 32 registers per thread, unrolled every phase 32 times and spilled 392
//  registers to scratch memory: 75 us, of which 60 in the two "trivial" counting / compaction loops)
__global__ void __launch_bounds__(kSelThreads) dec_select_hist(const unsigned *__restrict__ keys_all, const int *__restrict__ hist_part, int n_blocks, DecCfg c,
                                                               KeyBins kb, unsigned long long *__restrict__ sel_words /*[G][pre_max]*/,
                                                               int *__restrict__ sel_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *s_hist = reinterpret_cast<int *>(smem);                                           // [2048]: [0, 1024) the key bins; all of it in the radix passes
    unsigned long long *s_cand = reinterpret_cast<unsigned long long *>(s_hist + 2048);     // [kMaxCand]
    int *s_misc = reinterpret_cast<int *>(s_cand + kMaxCand);                               // [32]
    unsigned *s_keys = reinterpret_cast<unsigned *>(s_misc + 32);                           // [HW rounded up to 4096]
    const int g = blockIdx.x, tid = threadIdx.x;
    const unsigned *keys = keys_all + (int64_t)g * c.HW;
    unsigned long long *out = sel_words + (int64_t)g * c.pre_max;
    const int HWp = (c.HW + 4095) & ~4095;
    FD_DT(0);
    // ---- the group's histogram: bin tid = sum over the blocks (coalesced: consecutive threads, consecutive bins); keys -> LDS
    {
        const int *hp = hist_part + (int64_t)g * n_blocks * 1024 + tid;
        int v = 0;
#pragma unroll 8
        for (int b = 0; b < n_blocks; ++b) v += hp[(int64_t)b * 1024];
        s_hist[tid] = v;
    }
    {
        // 16-byte loads where the four cells exist, all of a thread's loads in flight before the first LDS store (HWp / 4096 <= 8 rounds)
        uint4 kv[8];
        const int last4 = ((c.HW - 1) & ~3) - (((c.HW & 3) != 0 || (((uintptr_t)keys) & 15) != 0) ? 4 : 0);  // last 4-cell piece that is whole and aligned
        const bool al = (((uintptr_t)keys) & 15) == 0 && last4 >= 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {  // unconditional loads at a clamped index (all in flight together); pieces past the whole ones are patched below
            const int i0 = tid * 4 + r * 4 * kSelThreads;
            const int ic = al ? (i0 <= last4 ? i0 : last4) : 0;
            kv[r] = al ? *reinterpret_cast<const uint4 *>(keys + ic) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i0 = tid * 4 + r * 4 * kSelThreads;
            if (i0 < HWp) {
                uint4 v = kv[r];
                if (!al || i0 > last4) {  // the ragged end (and unaligned maps): cell by cell
                    v.x = i0 < c.HW ? keys[i0] : 0u;
                    v.y = i0 + 1 < c.HW ? keys[i0 + 1] : 0u;
                    v.z = i0 + 2 < c.HW ? keys[i0 + 2] : 0u;
                    v.w = i0 + 3 < c.HW ? keys[i0 + 3] : 0u;
                }
                *reinterpret_cast<uint4 *>(s_keys + i0) = v;
            }
        }
    }
    __syncthreads();
    FD_DT(1);
    // one block scan from the top bin down: thread t owns bin 1023 - t.  The bin in which the cumulative count reaches pre_max is the
    // threshold bin (none when fewer keys than pre_max are valid: every key is taken)
    {
        const int h = s_hist[1023 - tid];
        const int acc = block_sum_1024(h, s_misc);  // keys in the bins above this thread's
        const int M_all = s_misc[16];
        __syncthreads();
        if (tid == 0) { s_misc[20] = -1; s_misc[21] = 0; s_misc[22] = 0; s_misc[23] = 0; }
        __syncthreads();
        if (M_all > c.pre_max && acc < c.pre_max && acc + h >= c.pre_max) { s_misc[20] = 1023 - tid; s_misc[21] = acc; }
        __syncthreads();
    }
    const int M = s_misc[16];
    const int bsel = s_misc[20], above = s_misc[21];  // every key in a bin > bsel is taken; bsel = -1: all of them
    const int n_cand = bsel >= 0 ? s_hist[bsel] : 0, need = c.pre_max - above;  // (n_cand >= need by the choice of the bin)
    FD_DT(2);
#ifdef FD_DEC_TRACE
    if (threadIdx.x == 0 && g_dectrace) { g_dectrace[blockIdx.x * 16 + 8] = (unsigned long long)M; g_dectrace[blockIdx.x * 16 + 9] = (unsigned long long)n_cand; g_dectrace[blockIdx.x * 16 + 10] = (unsigned long long)need; }
#endif
    // ---- one pass over the keys: those above the bin go straight to the output, those inside it to the LDS list (or, when the whole bin is
    //      taken, to the output as well).  Slots come from two LDS counters, one atomicAdd per wave and round (ballot + lane rank); the order
    //      of the survivors is free: dec_rank_decode ranks them.  (Two counting loops and two block scans in front of the writes, as first
    //      written, were 10 of this kernel's 17 us.)
    const bool list = bsel >= 0 && need > 0 && n_cand <= kMaxCand && n_cand > need;  // the bin's keys are ranked in LDS
    const bool all_in = bsel >= 0 && n_cand == need;                               // the whole bin is taken
    const int lane = tid & 63;
#pragma unroll 4
    for (int i = tid; i < HWp; i += kSelThreads) {  // (HWp is a multiple of 4096: whole waves run every round)
        const unsigned k = s_keys[i];
        const int b = k ? key_bin(kb, k) : -2;
        const bool is_def = b > bsel, is_cand = b == bsel && b >= 0;
        const unsigned long long w = ((unsigned long long)k << 32) | (unsigned)(0xffffffffu - (unsigned)i);
        const unsigned long long md = __ballot(is_def), mc = __ballot(is_cand);
        if (md) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_misc[22], (int)__builtin_popcountll(md));
            base = __builtin_amdgcn_readfirstlane(base);
            if (is_def) out[base + (int)__builtin_popcountll(md & ((1ull << lane) - 1ull))] = w;
        }
        if (mc && (list || all_in)) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_misc[23], (int)__builtin_popcountll(mc));
            base = __builtin_amdgcn_readfirstlane(base);
            const int slot = base + (int)__builtin_popcountll(mc & ((1ull << lane) - 1ull));
            if (is_cand) {
                if (all_in) out[above + slot] = w;
                else s_cand[slot] = w;
            }
        }
    }
    if (tid == 0) { sel_count[g] = M < c.pre_max ? M : c.pre_max; sel_count[gridDim.x + g] = M; }  // [G] taken, [G] valid candidates
    FD_DT(3);
    if (bsel < 0 || need <= 0 || all_in) return;  // (uniform)
    if (list) {
        const int n_pad = (n_cand + 7) & ~7;
        __syncthreads();
        if (tid >= n_cand && tid < n_pad) s_cand[tid] = 0ull;  // (a zero word is greater than nothing)
        __syncthreads();
        if (tid < n_cand) {
            const unsigned long long w = s_cand[tid];
            int rank = 0;
            for (int jj = 0; jj < n_pad; jj += 8) {  // eight independent LDS reads per round (the plain loop was one dependent round trip per word)
                unsigned long long o[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) o[u] = s_cand[jj + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) rank += o[u] > w;
            }
            if (rank < need) out[above + rank] = w;
        }
        FD_DT(4);
        return;
    }
    // ---- degenerate: more keys in one bin than the LDS list holds (thousands of equal scores).  Radix selection over what is left of the
    //      49-bit value (score bits << 17 | 0x1ffff - cell): the low `shift` key bits and the cell, 11 + 11 + 11 + 4 bits from the top.
    auto low_of = [&](unsigned kk, unsigned cell) -> unsigned long long {
        return ((unsigned long long)((kk - kb.base) & ((1u << kb.shift) - 1u)) << 17) | (unsigned long long)(0x1ffffu - cell);
    };
    int need_r = need;
    unsigned long long prefix = 0ull, pmask = 0ull;
    const int shifts[4] = {26, 15, 4, 0}, bits[4] = {11, 11, 11, 4};
    for (int pass = 0; pass < 4; ++pass) {
        const int nb = 1 << bits[pass];
        __syncthreads();
        for (int i = tid; i < nb; i += kSelThreads) s_hist[i] = 0;
        __syncthreads();
        for (int i = tid; i < HWp; i += kSelThreads) {  // (whole waves run every round: hist_add uses ballots)
            const unsigned k = s_keys[i];
            const bool cand = k != 0u && key_bin(kb, k) == bsel;
            const unsigned long long v = cand ? low_of(k, (unsigned)i) : 0ull;
            hist_add(s_hist, (int)((v >> shifts[pass]) & (unsigned long long)(nb - 1)), cand && (v & pmask) == prefix);
        }
        find_bin<2>(s_hist, nb, need_r, s_misc, s_misc + 20);
        const unsigned long long b = (unsigned long long)s_misc[20];
        need_r -= s_misc[21];
        prefix |= b << shifts[pass];
        pmask |= (unsigned long long)(nb - 1) << shifts[pass];
    }
    __syncthreads();
    // prefix = the smallest selected value (the values are unique): take every key of the bin >= it
    int n_take = 0;
    for (int i = tid; i < HWp; i += kSelThreads) {
        const unsigned k = s_keys[i];
        n_take += (k != 0u && key_bin(kb, k) == bsel && low_of(k, (unsigned)i) >= prefix);
    }
    int tpos = block_sum_1024(n_take, s_misc);
    __syncthreads();
    for (int i = tid; i < HWp; i += kSelThreads) {
        const unsigned k = s_keys[i];
        if (k != 0u && key_bin(kb, k) == bsel && low_of(k, (unsigned)i) >= prefix) out[above + tpos++] = ((unsigned long long)k << 32) | (unsigned)(0xffffffffu - (unsigned)i);
    }
}

constexpr int kRankElems = 64;  // elements per workgroup of dec_rank_decode (four threads each): 16 workgroups for pre_max 1000
__global__ void __launch_bounds__(256) dec_rank_decode(const unsigned long long *__restrict__ sel_words, const int *__restrict__ sel_count, MapView reg,
                                                        MapView height, MapView dim, MapView rot, DecCfg c, int G, float *__restrict__ sel_boxes,
                                                        float4 *__restrict__ planes, float *__restrict__ sel_scores, int *__restrict__ sel_cell) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long *s_w = reinterpret_cast<unsigned long long *>(smem);  // [S]
    const int g = blockIdx.y, tid = threadIdx.x;
    const int S = sel_count[g];
    if ((int)blockIdx.x * kRankElems >= S) return;  // (uniform)
    const unsigned long long *words = sel_words + (int64_t)g * c.pre_max;
    const int S_pad = (S + 31) & ~31;  // (the LDS request is for pre_max rounded up to 32 words; zero words are greater than nothing)
    for (int i = tid; i < S_pad; i += 4 * kRankElems) s_w[i] = i < S ? words[i] : 0ull;
    __syncthreads();
    const int e = blockIdx.x * kRankElems + (tid >> 2), q = tid & 3;
    const bool live = e < S;
    const unsigned long long w = live ? s_w[e] : ~0ull;
    int rank = 0;
    for (int j0 = 0; j0 < S_pad; j0 += 32) {  // thread q of an element takes words q, q + 4, ...: eight independent LDS reads per round
        unsigned long long o[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) o[u] = s_w[j0 + q + 4 * u];
#pragma unroll
        for (int u = 0; u < 8; ++u) rank += o[u] > w;
    }
    rank += __shfl_xor(rank, 1);
    rank += __shfl_xor(rank, 2);
    if (!live || q != 0) return;
    const int cell = (int)(0xffffffffu - (unsigned)(w & 0xffffffffull));
    const float score = __uint_as_float((unsigned)(w >> 32));
    float x, y;
    cell_center(c, reg, g, cell, x, y);
    const float z = ld(height, g, 0, cell);
    const float d0 = expf(ld(dim, g, 0, cell)), d1 = expf(ld(dim, g, 1, cell)), d2 = expf(ld(dim, g, 2, cell));
    const float yaw = atan2f(ld(rot, g, 0, cell), ld(rot, g, 1, cell));
    const int64_t o = (int64_t)g * c.pre_max + rank;
    float *sb = sel_boxes + o * 7;
    sb[0] = x; sb[1] = y; sb[2] = z; sb[3] = d0; sb[4] = d1; sb[5] = d2; sb[6] = yaw;
    const float nb[7] = {x, y, z, d1, d0, d2, __fsub_rn(-yaw, 1.5707963267948966f)};  // box_torch_ops.py:256-257: [x,y,z,dim1,dim0,dim2,-yaw-pi/2]
    store_footprint(planes, (int64_t)G * c.pre_max, o, make_footprint(nb));
    sel_scores[o] = score;
    sel_cell[o] = cell;
}

// ---------------------------------------------------------------- stage 3: IoU mask words (upper triangle)
// One wave per (row, 64-column block): lane = column, so the 64-bit mask word of iou3d_nms_kernel.cu:296-305 is a
// single wave ballot.  The row's footprint is wave-uniform, the columns' footprints are four coalesced float4 loads.
// Pairs whose centres are farther apart than the two half-diagonals (+0.1 m, which covers the 1e-2 corner-inside
// margin) cannot produce a crossing point or an inside corner, so their overlap is exactly 0; they skip the geometry.
#ifdef FD_MASK_DEBUG
// tuning build: every near pair is evaluated again from footprints PINNED in registers (the compiler cannot answer by loading them again), then
// twice more with digests, then once from footprints loaded again; counters: [0] near pairs, [1] second evaluation differs, [2] third differs,
// [3] re-loaded footprints differ bit-wise, [4..7] first digest field that differs (point count, point set, angles, area); the first 64
// disagreements are logged with lane / block / HW_ID (tools/soak_determinism.py prints them)
__device__ inline void mask_pin(Footprint &f) {
    for (int k = 0; k < 4; ++k) { asm volatile("" : "+v"(f.vx[k])); asm volatile("" : "+v"(f.vy[k])); }
    asm volatile("" : "+v"(f.cx)); asm volatile("" : "+v"(f.cy)); asm volatile("" : "+v"(f.co)); asm volatile("" : "+v"(f.si));
    asm volatile("" : "+v"(f.hx)); asm volatile("" : "+v"(f.hy)); asm volatile("" : "+v"(f.area)); asm volatile("" : "+v"(f.reach));
}
__device__ inline void mask_self_check(const float4 *planes, int64_t n_total, int64_t base, int g, int row, int col, const Footprint &cur, const Footprint &oth,
                                       float *s_poly, float thr, bool hit) {
    const bool hit2 = footprint_iou(cur, oth, s_poly) > thr;
    EvalDigest d1, d2;
    (void)footprint_overlap(cur, oth, s_poly, &d1);
    (void)footprint_overlap(cur, oth, s_poly, &d2);
    if (d1.n != d2.n || d1.pts != d2.pts || d1.made != d2.made || d1.sides != d2.sides) {
        const int slot = atomicAdd(&g_masklog_n, 1);
        if (slot < 64) {
            unsigned *o = g_masklog + slot * 16;
            o[0] = (unsigned)g; o[1] = (unsigned)row; o[2] = (unsigned)col; o[3] = (unsigned)d1.n; o[4] = (unsigned)d2.n; o[5] = d1.made; o[6] = d2.made;
            o[7] = d1.sides; o[8] = d2.sides; o[9] = d1.pts; o[10] = d2.pts; o[11] = __float_as_uint(d1.area); o[12] = __float_as_uint(d2.area);
            o[13] = (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_ID
            o[14] = (unsigned)(threadIdx.x); o[15] = (unsigned)blockIdx.x;
        }
    }
    if (d1.n != d2.n) atomicAdd(&g_maskdbg[4], 1);
    else if (d1.pts != d2.pts) atomicAdd(&g_maskdbg[5], 1);
    else if (d1.angs != d2.angs) atomicAdd(&g_maskdbg[6], 1);
    else if (__float_as_uint(d1.area) != __float_as_uint(d2.area)) atomicAdd(&g_maskdbg[7], 1);
    const Footprint cur3 = load_footprint(planes, n_total, base + row), oth3 = load_footprint(planes, n_total, base + col);
    const bool hit3 = footprint_iou(cur3, oth3, s_poly) > thr;
    bool same_bits = true;
    for (int k = 0; k < 4; ++k) same_bits = same_bits && __float_as_uint(cur3.vx[k]) == __float_as_uint(cur.vx[k]) && __float_as_uint(oth3.vx[k]) == __float_as_uint(oth.vx[k]) &&
                                            __float_as_uint(cur3.vy[k]) == __float_as_uint(cur.vy[k]) && __float_as_uint(oth3.vy[k]) == __float_as_uint(oth.vy[k]);
    same_bits = same_bits && __float_as_uint(cur3.area) == __float_as_uint(cur.area) && __float_as_uint(oth3.area) == __float_as_uint(oth.area) &&
                __float_as_uint(cur3.cx) == __float_as_uint(cur.cx) && __float_as_uint(oth3.cx) == __float_as_uint(oth.cx);
    atomicAdd(&g_maskdbg[0], 1);
    if (hit2 != hit) atomicAdd(&g_maskdbg[1], 1);
    if (hit3 != hit) atomicAdd(&g_maskdbg[2], 1);
    if (!same_bits) atomicAdd(&g_maskdbg[3], 1);
}
#define FD_MASK_PIN(f) mask_pin(f)
#define FD_MASK_SELF_CHECK() mask_self_check(planes, n_total, base, g, row, col, cur, oth, s_poly + threadIdx.x, thr, hit)
#else
#define FD_MASK_PIN(f)
#define FD_MASK_SELF_CHECK()
#endif
constexpr int kMaskRows = kPolyThreads / 64;  // rows (waves) per workgroup of nms_mask
__global__ void __launch_bounds__(kPolyThreads) nms_mask(const float4 *__restrict__ planes, int64_t n_total, const int *__restrict__ counts, int n_max,
                                                         int col_blocks, float thr, CircleCfg cc, unsigned long long *__restrict__ mask_all) {
    __shared__ float s_poly[kPolyLdsFloats];
    const int g = blockIdx.z, cb = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * kMaskRows + (threadIdx.x >> 6);
    const int n = counts ? counts[g] : n_max;
    if (row >= n) return;
    const int rb = row >> 6;
    if (cb < rb || cb * 64 >= n) return;  // the sweep never reads words left of the diagonal (iou3d_nms.cpp:127)
    const int64_t base = (int64_t)g * n_max;
    const int col = cb * 64 + lane;
    bool hit = false;
    if (cc.per > 0) {
        // circle_nms_jit.py:21-25 on float32 scalars: dist = (x_i - x_j)**2 + (y_i - y_j)**2 (each product and the sum rounded, no fma:
        // the file is compiled with -ffp-contract=off), suppressed when dist <= thresh -- the threshold is compared with the SQUARED distance
        if (col < n && col > row) {
            const float4 a = planes[2 * n_total + base + row], b = planes[2 * n_total + base + col];
            const float dx = a.x - b.x, dy = a.y - b.y;
            hit = dx * dx + dy * dy <= cc.r[g / cc.per];
        }
    } else if (col < n && col > row) {
        Footprint cur = load_footprint(planes, n_total, base + row), oth = load_footprint(planes, n_total, base + col);
        FD_MASK_PIN(cur); FD_MASK_PIN(oth);
        const float dx = cur.cx - oth.cx, dy = cur.cy - oth.cy;
        const float reach = cur.reach + oth.reach + 0.1f;
        if (dx * dx + dy * dy <= reach * reach) {
            hit = footprint_iou(cur, oth, s_poly + threadIdx.x) > thr;
            FD_MASK_SELF_CHECK();
        }
    }
    const unsigned long long t = __ballot(hit);
    if (lane == 0) mask_all[((int64_t)g * n_max + row) * col_blocks + cb] = t;
}

// ---------------------------------------------------------------- stage 4: greedy sweep, one wave per group
__global__ void __launch_bounds__(64) nms_sweep(const unsigned long long *__restrict__ mask_all, const int *__restrict__ counts, int n_max,
                                                int col_blocks, int post_max, int *__restrict__ keep_all /*[G,post_cap]*/, int post_cap,
                                                int *__restrict__ out_count, const int *__restrict__ totals) {
    const int g = blockIdx.x, lane = threadIdx.x;
    const int n = counts ? counts[g] : n_max;
    const unsigned long long *mask = mask_all + (int64_t)g * n_max * col_blocks;
    int *keep = keep_all + (int64_t)g * post_cap;
    unsigned long long remv = 0ull;  // lane j holds the removed-bits word of column block j
    int kept = 0;
    const int nblk = (n + 63) / 64;
    for (int blk = 0; blk < nblk && kept < post_max; ++blk) {
        const int row = blk * 64 + lane;
        const unsigned long long diag = (row < n) ? mask[(int64_t)row * col_blocks + blk] : 0ull;
        unsigned long long cur = __shfl(remv, blk);
        unsigned long long keep_bits = 0ull;
        const int rows_here = min(64, n - blk * 64);
        for (int t = 0; t < rows_here && kept < post_max; ++t) {
            const unsigned long long dt = __shfl(diag, t);
            if (!((cur >> t) & 1ull)) {
                keep_bits |= 1ull << t;
                cur |= dt;
                if (lane == 0) keep[kept] = blk * 64 + t;
                ++kept;
            }
        }
        // fold the kept rows' masks into the later column blocks
        unsigned long long kb = keep_bits;
        while (kb) {
            const int t = __builtin_ctzll(kb);
            kb &= kb - 1;
            if (lane < col_blocks && lane > blk) remv |= mask[(int64_t)(blk * 64 + t) * col_blocks + lane];
        }
    }
    // circular NMS has no pre-NMS cut in the reference: a group with more candidates than were taken is decided only if post_max rows were
    // kept among the taken ones (greedy in score order: the first post_max kept rows do not depend on later candidates); else count = -1
    if (lane == 0) out_count[g] = (totals && totals[g] > n_max && kept < post_max) ? -1 : kept;
}

__global__ void __launch_bounds__(128) dec_gather(const int *__restrict__ keep_all, const int *__restrict__ kept_count, int post_max, int pre_max,
                                                  const float *__restrict__ sel_boxes, const float *__restrict__ sel_scores,
                                                  const int *__restrict__ sel_cell, float *__restrict__ out_boxes, float *__restrict__ out_scores,
                                                  int *__restrict__ out_cell) {
    const int g = blockIdx.x, k = threadIdx.x;
    if (k >= post_max) return;
    const int64_t o = (int64_t)g * post_max + k;
    if (k < kept_count[g]) {
        const int64_t s = (int64_t)g * pre_max + keep_all[o];
        for (int d = 0; d < 7; ++d) out_boxes[o * 7 + d] = sel_boxes[s * 7 + d];
        out_scores[o] = sel_scores[s];
        out_cell[o] = sel_cell[s];
    } else {
        for (int d = 0; d < 7; ++d) out_boxes[o * 7 + d] = 0.0f;
        out_scores[o] = 0.0f;
        out_cell[o] = -1;
    }
}

__global__ void __launch_bounds__(256) keep_to_i64(const int *__restrict__ keep, const int *__restrict__ count, int n, long long *__restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i < count[0] ? (long long)keep[i] : 0ll;
}

__global__ void __launch_bounds__(kPolyThreads) iou_pairs(const float *__restrict__ a, int na, const float *__restrict__ b, int nb, float *__restrict__ out) {
    __shared__ float s_poly[kPolyLdsFloats];
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)na * nb) return;
    int i = (int)(t / nb), j = (int)(t - (int64_t)i * nb);
    float ba[7], bb[7];
    for (int d = 0; d < 7; ++d) { ba[d] = a[i * 7 + d]; bb[d] = b[j * 7 + d]; }
    out[t] = footprint_iou(make_footprint(ba), make_footprint(bb), s_poly + threadIdx.x);
}

// Final assembly of predict's output (center_head.py:559-570,606-607,672-697): output step s takes the boxes of group
// step_group[s], its two velocity channels vel_channel[s], vel_channel[s] + 1 of that group's velocity map at the kept cells, and the
// label offset label_of[s].  One packed row per kept box: x y z w l h vx vy yaw score label; rows k >= count are zero.
struct AsmSteps {
    int S;
    int group[kMaxSteps], vel_channel[kMaxSteps], label[kMaxSteps];
};

// ---------------------------------------------------------------- stages 4-6, round 5: sweep from LDS + gather + assembly in one launch
// The round-4 sweep (one wave, mask words from global memory: a dependent L2 round trip per diagonal block and per kept row, 21 us),
// dec_gather and assemble_kernel are one kernel now: the workgroup first stages the upper triangle of the group's mask words in LDS
// (128 KB for pre_max 1000), one wave then runs the greedy sweep -- not over all rows but from kept row to kept row: the next row to
// keep is the lowest clear bit of the diagonal word, at most post_max iterations of ~150 cycles -- and all threads write the outputs:
// the kept boxes / scores / cells of the group and, when asked for, the packed rows of every output step that reads this group
// (x y z w l h vx vy yaw score label, the velocity gathered from the map view at the kept cell).
__global__ void __launch_bounds__(1024) nms_sweep_tail(const unsigned long long *__restrict__ mask_all, const int *__restrict__ counts, int n_max, int col_blocks,
                                                       int post_max, const float *__restrict__ sel_boxes, const float *__restrict__ sel_scores,
                                                       const int *__restrict__ sel_cell, float *__restrict__ out_boxes, float *__restrict__ out_scores,
                                                       int *__restrict__ out_cell, int *__restrict__ out_count, MapView vel, AsmSteps st, int B,
                                                       float *__restrict__ packed, int *__restrict__ counts_out, const int *__restrict__ totals) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long *s_mask = reinterpret_cast<unsigned long long *>(smem);              // [n_max][col_blocks]
    int *s_keep = reinterpret_cast<int *>(s_mask + (size_t)n_max * col_blocks);             // [128] kept rows, [128] = their number
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int n = counts ? counts[g] : n_max;
    const unsigned long long *mask = mask_all + (int64_t)g * n_max * col_blocks;
    FD_DT(11);
    {
        // (words left of the diagonal are never read and nms_mask does not write them: whatever is there is copied along, harmlessly;
        //  batches of eight 16-byte loads in flight per thread -- a load-wait-store loop over 128 KB was most of this kernel's 26 us)
        const int n2 = (n * col_blocks + 1) / 2;  // 16-byte pieces (the group's slice starts 16-byte aligned: n_max * col_blocks * 8 bytes, col_blocks even or the tail is one word)
        const bool wide = ((n_max * col_blocks) & 1) == 0;
        if (wide) {
            const ulonglong2 *m2 = reinterpret_cast<const ulonglong2 *>(mask);
            ulonglong2 *s2 = reinterpret_cast<ulonglong2 *>(s_mask);
            for (int i0 = tid; i0 < n2; i0 += 8 * 1024) {
                ulonglong2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {  // unconditional loads at a clamped index: the eight are in flight together (predicated loads
                    const int idx = i0 + u * 1024 < n2 ? i0 + u * 1024 : n2 - 1;  // made hipcc wait for each and park it in scratch memory)
                    v[u] = m2[idx];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (i0 + u * 1024 < n2) s2[i0 + u * 1024] = v[u];
            }
        } else {
            for (int i = tid; i < n * col_blocks; i += 1024) s_mask[i] = mask[i];
        }
    }
    __syncthreads();
    FD_DT(12);
    if (tid < 64) {
        unsigned long long remv = 0ull;  // lane j: removed-bits word of column block j
        int kept = 0;
        const int nblk = (n + 63) / 64;
        for (int blk = 0; blk < nblk && kept < post_max; ++blk) {
            const int rows_here = min(64, n - blk * 64);
            const unsigned long long rowmask = rows_here == 64 ? ~0ull : ((1ull << rows_here) - 1ull);
            // lane t: the diagonal word of row blk * 64 + t (one LDS read per block); the inner loop then runs on v_readlane alone, and the
            // kept rows' other words are OR-ed into remv by LDS reads nobody waits for before the next block (a read + shuffle per kept
            // row, as first written, cost 280 cycles each)
            const unsigned long long diag = lane < rows_here ? s_mask[(size_t)(blk * 64 + lane) * col_blocks + blk] : 0ull;
            auto lane_word = [&](unsigned long long v, int j) -> unsigned long long {  // v of lane j (j uniform): two v_readlane
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffffull), j);
                const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), j);
                return ((unsigned long long)hi << 32) | lo;
            };
            // scalar work only, as little of it as possible (a lone wave issues an instruction every ~5 cycles: the first version of this loop
            // had 32 of them and an LDS store per kept row, 280 cycles): lowest clear bit, that row's diagonal word, clear + mask.
            // A row's diagonal word only has bits ABOVE the row, and everything below t is decided: no 'rows after t' mask is needed.
            unsigned long long avail = ~lane_word(remv, blk) & rowmask, keep_bits = 0ull;
            int room = post_max - kept;
            while (avail && room > 0) {
                const int t = __builtin_ctzll(avail);
                keep_bits |= 1ull << t;
                avail &= avail - 1ull;
                avail &= ~lane_word(diag, t);
                --room;
            }
            // the kept rows of this block, in order, to the keep list: lane t holds bit t
            if ((keep_bits >> lane) & 1ull) s_keep[kept + (int)__builtin_popcountll(keep_bits & ((1ull << lane) - 1ull))] = blk * 64 + lane;
            kept += (int)__builtin_popcountll(keep_bits);
            // fold the kept rows' words into the later column blocks: four independent LDS reads per round
            while (keep_bits) {
                unsigned long long w4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    w4[u] = 0ull;
                    if (keep_bits) {
                        const int t = __builtin_ctzll(keep_bits);
                        keep_bits &= keep_bits - 1ull;
                        if (lane < col_blocks && lane > blk && lane * 64 < n) w4[u] = s_mask[(size_t)(blk * 64 + t) * col_blocks + lane];
                    }
                }
                remv |= (w4[0] | w4[1]) | (w4[2] | w4[3]);
            }
        }
        if (lane == 0) s_keep[128] = kept;
    }
    __syncthreads();
    FD_DT(13);
    const bool undecided = totals && totals[g] > n_max && s_keep[128] < post_max;  // (see nms_sweep: circular NMS only) -> count -1, no rows
    const int kept = undecided ? 0 : s_keep[128];
    const int kept_out = undecided ? -1 : kept;
    if (tid == 0) out_count[g] = kept_out;
    for (int k = tid; k < post_max; k += 1024) {
        const int64_t o = (int64_t)g * post_max + k;
        if (k < kept) {
            const int64_t sidx = (int64_t)g * n_max + s_keep[k];
            for (int d = 0; d < 7; ++d) out_boxes[o * 7 + d] = sel_boxes[sidx * 7 + d];
            out_scores[o] = sel_scores[sidx];
            out_cell[o] = sel_cell[sidx];
        } else {
            for (int d = 0; d < 7; ++d) out_boxes[o * 7 + d] = 0.0f;
            out_scores[o] = 0.0f;
            out_cell[o] = -1;
        }
    }
    if (!packed) return;
    const int grp = g / B, b = g - grp * B;  // decode groups are group-major: [G][B]
    for (int s_i = 0; s_i < st.S; ++s_i) {
        if (st.group[s_i] != grp) continue;
        if (tid == 0) counts_out[b * st.S + s_i] = kept_out;
        for (int k = tid; k < post_max; k += 1024) {
            float *o = packed + (((int64_t)b * st.S + s_i) * post_max + k) * 11;
            if (k < kept) {
                const int64_t sidx = (int64_t)g * n_max + s_keep[k];
                const float *bx = sel_boxes + sidx * 7;
                const int cc = sel_cell[sidx];
                o[0] = bx[0]; o[1] = bx[1]; o[2] = bx[2]; o[3] = bx[3]; o[4] = bx[4]; o[5] = bx[5];
                o[6] = ld(vel, g, st.vel_channel[s_i], cc);
                o[7] = ld(vel, g, st.vel_channel[s_i] + 1, cc);
                o[8] = bx[6];
                o[9] = sel_scores[sidx];
                o[10] = (float)st.label[s_i];
            } else {
#pragma unroll
                for (int d = 0; d < 11; ++d) o[d] = 0.0f;
            }
        }
    }
}

struct DecWs {
    size_t keys, sel_boxes, nms_boxes, sel_scores, sel_cell, sel_count, mask, keep, foot, hist, words, total;
    int col_blocks, npad, key_blocks;
};
DecWs dec_layout(int G, int HW, int pre_max, int post_max) {
    DecWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += fd::align_up(bytes, 256); return o; };
    w.col_blocks = (pre_max + 63) / 64;
    w.npad = 64;
    while (w.npad < pre_max) w.npad <<= 1;
    w.keys = take(sizeof(unsigned) * (size_t)G * HW);
    w.sel_boxes = take(sizeof(float) * 7 * (size_t)G * pre_max);
    w.nms_boxes = take(sizeof(float) * 7 * (size_t)G * pre_max);
    w.sel_scores = take(sizeof(float) * (size_t)G * pre_max);
    w.sel_cell = take(sizeof(int) * (size_t)G * pre_max);
    w.sel_count = take(sizeof(int) * 2 * (size_t)G);  // taken [G] | valid candidates [G]
    w.mask = take(sizeof(unsigned long long) * (size_t)G * pre_max * w.col_blocks);
    w.keep = take(sizeof(int) * (size_t)G * post_max);
    w.foot = take(sizeof(Footprint) * (size_t)G * pre_max);
    w.key_blocks = (HW + kKeyBlock - 1) / kKeyBlock;
    w.hist = take(sizeof(int) * 1024 * (size_t)G * w.key_blocks);
    w.words = take(sizeof(unsigned long long) * (size_t)G * pre_max);
    w.total = off;
    return w;
}

}  // namespace

__global__ void __launch_bounds__(128) assemble_kernel(const float *__restrict__ boxes7, const float *__restrict__ scores, const int *__restrict__ cell,
                                                       const int *__restrict__ count, MapView vel, AsmSteps st, int B, int post,
                                                       float *__restrict__ packed, int *__restrict__ counts_out) {
    const int b = blockIdx.x / st.S, s = blockIdx.x - b * st.S, k = threadIdx.x;
    const int g = st.group[s];
    const int64_t src = (int64_t)g * B + b;  // decode groups are group-major: [G][B]
    const int n = count[src];
    if (k == 0) counts_out[b * st.S + s] = n;
    if (k >= post) return;
    float *o = packed + (((int64_t)b * st.S + s) * post + k) * 11;
    if (k < n) {
        const float *bx = boxes7 + (src * post + k) * 7;
        const int c = cell[src * post + k];
        // the velocity map of group g, sample b: group index of the view = g * B + b
        const float vx = ld(vel, (int)src, st.vel_channel[s], c), vy = ld(vel, (int)src, st.vel_channel[s] + 1, c);
        o[0] = bx[0]; o[1] = bx[1]; o[2] = bx[2]; o[3] = bx[3]; o[4] = bx[4]; o[5] = bx[5];
        o[6] = vx; o[7] = vy; o[8] = bx[6];
        o[9] = scores[src * post + k];
        o[10] = (float)st.label[s];
    } else {
#pragma unroll
        for (int d = 0; d < 11; ++d) o[d] = 0.0f;
    }
}

extern "C" size_t fd_decode_workspace_bytes(int G, const fd_decode_cfg *cfg) {
    if (!cfg || G <= 0) return 0;
    return dec_layout(G, cfg->H * cfg->W, cfg->nms_pre_max, cfg->nms_post_max).total;
}

#ifdef FD_DEC_TRACE
extern "C" int fd_debug_set_dectrace(void *p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_dectrace), &p, sizeof(p)) == hipSuccess ? 0 : -1; }
#endif

namespace {
// shared body of fd_centerpoint_decode_maps / fd_centerpoint_decode_packed (vel == nullptr: no packed assembly)
int decode_impl(const fd_map_view *hm, const fd_map_view *reg, const fd_map_view *height, const fd_map_view *dim, const fd_map_view *rot, int G,
                const fd_decode_cfg *cfg, float *out_boxes7, float *out_scores, int32_t *out_cell, int32_t *out_count, void *workspace,
                size_t workspace_bytes, const fd_map_view *vel, int B, const AsmSteps *steps, float *packed, int32_t *counts_out, hipStream_t stream) {
    FD_REQUIRE(hm && reg && height && dim && rot && cfg && out_boxes7 && out_scores && out_cell && out_count, "fd_centerpoint_decode: null argument");
    FD_REQUIRE(hm->data && reg->data && height->data && dim->data && rot->data, "fd_centerpoint_decode: null map");
    for (const fd_map_view *v : {hm, reg, height, dim, rot}) FD_REQUIRE(v->dtype == 0 || v->dtype == 1, "fd_centerpoint_decode: map dtype must be 0 (f32) or 1 (bf16)");
    FD_REQUIRE(G > 0 && cfg->H > 0 && cfg->W > 0, "fd_centerpoint_decode: bad shape");
    FD_REQUIRE(cfg->nms_pre_max >= 1 && cfg->nms_pre_max <= kMaxPre, "fd_centerpoint_decode: nms_pre_max must be in [1,%d]", kMaxPre);
    FD_REQUIRE(cfg->nms_post_max >= 1 && cfg->nms_post_max <= 128, "fd_centerpoint_decode: nms_post_max must be in [1,128]");
    // key 0 means "no candidate" and make_key_bins() starts the bins at the threshold's bit pattern: with a negative threshold a score
    // that underflowed to exactly 0.0f would be dropped although `score > threshold` keeps it (center_head.py:666).  Not representable -> refused.
    FD_REQUIRE(cfg->score_threshold >= 0.f, "fd_centerpoint_decode: score_threshold must be >= 0 (got %g)", (double)cfg->score_threshold);
    // the in-register selection (dec_select_hist) rests on three limits; the dispatch condition below (HW <= 32 * kSelThreads) implies all of them
    static_assert(32 * kSelThreads == 8 * 4096, "dec_select_hist walks a group's keys in 8 rounds of 4096 (4 per thread)");
    static_assert(32 * kSelThreads <= (1 << 17), "a cell index must fit the 17-bit field of the 49-bit ranking word");
    DecCfg c;
    c.H = cfg->H; c.W = cfg->W; c.HW = cfg->H * cfg->W;
    c.osf = cfg->out_size_factor; c.vx = cfg->voxel_x; c.vy = cfg->voxel_y; c.px = cfg->pc_x; c.py = cfg->pc_y;
    c.thr = cfg->score_threshold;
    for (int i = 0; i < 6; ++i) c.rng[i] = cfg->center_range[i];
    c.iou_thr = cfg->nms_iou_threshold;
    c.pre_max = cfg->nms_pre_max; c.post_max = cfg->nms_post_max;
    c.hm_channels = cfg->hm_channels > 1 ? cfg->hm_channels : 1;
    FD_REQUIRE(c.hm_channels <= 16, "fd_centerpoint_decode: hm_channels must be <= 16");
    CircleCfg cc;
    cc.per = 0;
    for (float &r : cc.r) r = 0.f;
    FD_REQUIRE(cfg->nms_kind == 0 || cfg->nms_kind == 1, "fd_centerpoint_decode: nms_kind must be 0 (rotated) or 1 (circular), got %d", cfg->nms_kind);
    if (cfg->nms_kind == 1) {
        FD_REQUIRE(cfg->n_radius >= 1 && cfg->n_radius <= 16 && G % cfg->n_radius == 0,
                   "fd_centerpoint_decode: circular NMS needs 1 <= n_radius <= 16 dividing G (got %d, G = %d)", cfg->n_radius, G);
        cc.per = G / cfg->n_radius;
        for (int i = 0; i < cfg->n_radius; ++i) {
            FD_REQUIRE(cfg->circle_radius[i] >= 0.f, "fd_centerpoint_decode: circle_radius[%d] must be >= 0", i);  // (also refuses NaN)
            cc.r[i] = cfg->circle_radius[i];
        }
    }
    DecWs w = dec_layout(G, c.HW, c.pre_max, c.post_max);
    if (!workspace || workspace_bytes < w.total) {
        fd::set_error("fd_centerpoint_decode: workspace %zu < required %zu", workspace_bytes, w.total);
        return FD_EWORKSPACE;
    }
    char *ws = (char *)workspace;
    unsigned *keys = (unsigned *)(ws + w.keys);
    float *sel_boxes = (float *)(ws + w.sel_boxes), *nms_boxes = (float *)(ws + w.nms_boxes), *sel_scores = (float *)(ws + w.sel_scores);
    int *sel_cell = (int *)(ws + w.sel_cell), *sel_count = (int *)(ws + w.sel_count), *keep = (int *)(ws + w.keep);
    const int *totals = cc.per > 0 ? sel_count + G : nullptr;
    unsigned long long *mask = (unsigned long long *)(ws + w.mask);
    float4 *foot = (float4 *)(ws + w.foot);
    const int64_t n_total = (int64_t)G * c.pre_max;
    const MapView vh = as_view(*hm), vr = as_view(*reg), vz = as_view(*height), vd = as_view(*dim), vt = as_view(*rot);
    if (c.HW <= 32 * kSelThreads) {
        // a group's keys fit the registers of one workgroup: histogram with the keys, selection from the histogram, order by ranking
        int *hist = (int *)(ws + w.hist);
        unsigned long long *words = (unsigned long long *)(ws + w.words);
        const KeyBins kb = make_key_bins(c.thr);
        hipLaunchKernelGGL(dec_keys_hist, dim3(w.key_blocks, G), dim3(256), 0, stream, vh, vr, vz, c, kb, keys, hist);
        const size_t sel_lds = 2048 * 4 + (size_t)kMaxCand * 8 + 32 * 4 + (size_t)((c.HW + 4095) & ~4095) * 4;
        static std::atomic<uint64_t> sel_lds_set{0};
        if (sel_lds > 65536 && !fd::ensure_dynamic_lds(reinterpret_cast<const void *>(dec_select_hist), 156 * 1024, sel_lds_set)) {
            fd::set_error("fd_centerpoint_decode: the runtime refused %zu bytes of LDS for the selection", sel_lds);
            return FD_ELAUNCH;
        }
        hipLaunchKernelGGL(dec_select_hist, dim3(G), dim3(kSelThreads), sel_lds, stream, keys, hist, w.key_blocks, c, kb, words, sel_count);
        hipLaunchKernelGGL(dec_rank_decode, dim3((c.pre_max + kRankElems - 1) / kRankElems, G), dim3(4 * kRankElems), (size_t)((c.pre_max + 31) & ~31) * 8, stream, words, sel_count, vr, vz,
                           vd, vt, c, G, sel_boxes, foot, sel_scores, sel_cell);
    } else {  // larger maps (the 270 x 270 map of the 0.05-m grid has 72 keys per thread: spill): keys streamed from memory, LDS bitonic sort
        hipLaunchKernelGGL(dec_keys, dim3((c.HW + 255) / 256, G), dim3(256), 0, stream, vh, vr, vz, c, keys);
        const size_t lds = (size_t)w.npad * 8 + 4096 * 4 + 32 * 4;
        hipLaunchKernelGGL(dec_select, dim3(G), dim3(kSelThreads), lds, stream, keys, vr, vz, vd, vt, c, w.npad, sel_boxes, nms_boxes, sel_scores, sel_cell,
                           sel_count);
        hipLaunchKernelGGL(footprint_kernel, dim3((unsigned)((n_total + 255) / 256)), dim3(256), 0, stream, nms_boxes, sel_count, c.pre_max, G, foot);
    }
    hipLaunchKernelGGL(nms_mask, dim3((c.pre_max + kMaskRows - 1) / kMaskRows, w.col_blocks, G), dim3(kPolyThreads), 0, stream, foot, n_total, sel_count, c.pre_max, w.col_blocks,
                       c.iou_thr, cc, mask);
    // sweep + gather (+ packed assembly) in one launch when the group's mask words fit the LDS; else the round-4 kernels
    const size_t sweep_lds = (size_t)c.pre_max * w.col_blocks * 8 + 129 * 4 + 12;
    static std::atomic<uint64_t> lds_set{0};
    AsmSteps none;
    none.S = 0;
    const bool fused = sweep_lds <= 152 * 1024 && (sweep_lds <= 65536 || fd::ensure_dynamic_lds(reinterpret_cast<const void *>(nms_sweep_tail), 152 * 1024, lds_set));
    if (fused) {
        hipLaunchKernelGGL(nms_sweep_tail, dim3(G), dim3(1024), sweep_lds, stream, mask, sel_count, c.pre_max, w.col_blocks, c.post_max, sel_boxes, sel_scores,
                           sel_cell, out_boxes7, out_scores, out_cell, out_count, vel ? as_view(*vel) : MapView{nullptr, 0, 0, 0, 0}, steps ? *steps : none,
                           B > 0 ? B : 1, vel ? packed : nullptr, counts_out, totals);
    } else {
        hipLaunchKernelGGL(nms_sweep, dim3(G), dim3(64), 0, stream, mask, sel_count, c.pre_max, w.col_blocks, c.post_max, keep, c.post_max, out_count, totals);
        hipLaunchKernelGGL(dec_gather, dim3(G), dim3(128), 0, stream, keep, out_count, c.post_max, c.pre_max, sel_boxes, sel_scores, sel_cell, out_boxes7,
                           out_scores, out_cell);
        if (vel)
            hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)(B * steps->S)), dim3(128), 0, stream, out_boxes7, out_scores, out_cell, out_count, as_view(*vel),
                               *steps, B, c.post_max, packed, counts_out);
    }
    return fd::check_launch("fd_centerpoint_decode");
}

int fill_steps(AsmSteps &st, int S, const int32_t *step_group, const int32_t *step_vel_channel, const int32_t *step_label, const char *who) {
    FD_REQUIRE(step_group && step_vel_channel && step_label, "%s: null step table", who);
    FD_REQUIRE(S >= 1 && S <= kMaxSteps, "%s: need 1 <= S <= %d", who, kMaxSteps);
    st.S = S;
    for (int s = 0; s < S; ++s) {
        FD_REQUIRE(step_group[s] >= 0 && step_vel_channel[s] >= 0, "%s: negative step entry", who);
        st.group[s] = step_group[s]; st.vel_channel[s] = step_vel_channel[s]; st.label[s] = step_label[s];
    }
    return FD_OK;
}
}  // namespace

extern "C" int fd_centerpoint_decode_maps(const fd_map_view *hm, const fd_map_view *reg, const fd_map_view *height, const fd_map_view *dim,
                                          const fd_map_view *rot, int G, const fd_decode_cfg *cfg, float *out_boxes7, float *out_scores, int32_t *out_cell,
                                          int32_t *out_count, void *workspace, size_t workspace_bytes, fd_stream_t stream_) {
    return decode_impl(hm, reg, height, dim, rot, G, cfg, out_boxes7, out_scores, out_cell, out_count, workspace, workspace_bytes, nullptr, 1, nullptr, nullptr,
                       nullptr, fd::as_stream(stream_));
}

// fd_centerpoint_decode_maps + fd_assemble_detections in one call (the sweep kernel's tail writes the packed rows)
extern "C" int fd_centerpoint_decode_packed(const fd_map_view *hm, const fd_map_view *reg, const fd_map_view *height, const fd_map_view *dim,
                                            const fd_map_view *rot, const fd_map_view *vel, int G, int B, const fd_decode_cfg *cfg, int S,
                                            const int32_t *step_group, const int32_t *step_vel_channel, const int32_t *step_label, float *out_boxes7,
                                            float *out_scores, int32_t *out_cell, int32_t *out_count, float *packed, int32_t *counts_out, void *workspace,
                                            size_t workspace_bytes, fd_stream_t stream_) {
    FD_REQUIRE(vel && vel->data && packed && counts_out, "fd_centerpoint_decode_packed: null argument");
    FD_REQUIRE(vel->dtype == 0 || vel->dtype == 1, "fd_centerpoint_decode_packed: map dtype must be 0 (f32) or 1 (bf16)");
    FD_REQUIRE(B >= 1 && G >= B && G % B == 0, "fd_centerpoint_decode_packed: G must be a multiple of B (decode groups are group-major: g = group * B + sample)");
    AsmSteps st;
    const int rc = fill_steps(st, S, step_group, step_vel_channel, step_label, "fd_centerpoint_decode_packed");
    if (rc != FD_OK) return rc;
    for (int s = 0; s < S; ++s) FD_REQUIRE(st.group[s] < G / B, "fd_centerpoint_decode_packed: step group out of range");
    return decode_impl(hm, reg, height, dim, rot, G, cfg, out_boxes7, out_scores, out_cell, out_count, workspace, workspace_bytes, vel, B, &st, packed, counts_out,
                       fd::as_stream(stream_));
}

// the round-1 signature: five [G, C, H, W] float32 tensors with their group strides (NCHW planes)
extern "C" int fd_centerpoint_decode(const float *hm, int64_t hm_gs, const float *reg, int64_t reg_gs, const float *height, int64_t h_gs,
                                     const float *dim, int64_t dim_gs, const float *rot, int64_t rot_gs, int G, const fd_decode_cfg *cfg,
                                     float *out_boxes7, float *out_scores, int32_t *out_cell, int32_t *out_count, void *workspace,
                                     size_t workspace_bytes, fd_stream_t stream_) {
    FD_REQUIRE(cfg, "fd_centerpoint_decode: null argument");
    const int64_t hw = (int64_t)cfg->H * cfg->W;
    const fd_map_view v[5] = {{hm, hm_gs, hw, 1, 0}, {reg, reg_gs, hw, 1, 0}, {height, h_gs, hw, 1, 0}, {dim, dim_gs, hw, 1, 0}, {rot, rot_gs, hw, 1, 0}};
    return fd_centerpoint_decode_maps(&v[0], &v[1], &v[2], &v[3], &v[4], G, cfg, out_boxes7, out_scores, out_cell, out_count, workspace, workspace_bytes,
                                      stream_);
}

extern "C" int fd_assemble_detections(const float *boxes7, const float *scores, const int32_t *cell, const int32_t *count, const fd_map_view *vel, int B,
                                      int post_max, int S, const int32_t *step_group, const int32_t *step_vel_channel, const int32_t *step_label,
                                      float *packed, int32_t *counts_out, fd_stream_t stream) {
    FD_REQUIRE(boxes7 && scores && cell && count && vel && vel->data && step_group && step_vel_channel && step_label && packed && counts_out,
               "fd_assemble_detections: null argument");
    FD_REQUIRE(B >= 1 && S >= 1 && S <= kMaxSteps && post_max >= 1 && post_max <= 128, "fd_assemble_detections: need 1 <= S <= %d, 1 <= post_max <= 128", kMaxSteps);
    FD_REQUIRE(vel->dtype == 0 || vel->dtype == 1, "fd_assemble_detections: map dtype must be 0 (f32) or 1 (bf16)");
    AsmSteps st;
    const int rc = fill_steps(st, S, step_group, step_vel_channel, step_label, "fd_assemble_detections");
    if (rc != FD_OK) return rc;
    hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)(B * S)), dim3(128), 0, fd::as_stream(stream), boxes7, scores, cell, count, as_view(*vel), st, B, post_max,
                       packed, counts_out);
    return fd::check_launch("fd_assemble_detections");
}

extern "C" size_t fd_nms_workspace_bytes(int n) {
    if (n <= 0) return 256;
    size_t cb = (size_t)(n + 63) / 64;
    return fd::align_up(sizeof(unsigned long long) * (size_t)n * cb, 256) + fd::align_up(sizeof(int) * (size_t)n, 256) +
           fd::align_up(sizeof(Footprint) * (size_t)n, 256);
}

extern "C" int fd_rotated_nms(const float *boxes7, int n, float thresh, int64_t *keep, int32_t *out_count, void *workspace,
                              size_t workspace_bytes, fd_stream_t stream_) {
    FD_REQUIRE(out_count && (n == 0 || (boxes7 && keep)), "fd_rotated_nms: null argument");
    FD_REQUIRE(n >= 0 && n <= kMaxPre, "fd_rotated_nms: n must be in [0,%d]", kMaxPre);
    hipStream_t stream = fd::as_stream(stream_);
    if (n == 0) {
        fd::fill_words(out_count, 0u, 1, stream);
        return fd::check_launch("fd_rotated_nms");
    }
    if (!workspace || workspace_bytes < fd_nms_workspace_bytes(n)) {
        fd::set_error("fd_rotated_nms: workspace too small");
        return FD_EWORKSPACE;
    }
    const int cb = (n + 63) / 64;
    unsigned long long *mask = (unsigned long long *)workspace;
    int *keep32 = (int *)((char *)workspace + fd::align_up(sizeof(unsigned long long) * (size_t)n * cb, 256));
    float4 *foot = (float4 *)((char *)keep32 + fd::align_up(sizeof(int) * (size_t)n, 256));
    hipLaunchKernelGGL(footprint_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, boxes7, (const int *)nullptr, n, 1, foot);
    hipLaunchKernelGGL(nms_mask, dim3((n + kMaskRows - 1) / kMaskRows, cb, 1), dim3(kPolyThreads), 0, stream, foot, (int64_t)n, (const int *)nullptr, n, cb, thresh, CircleCfg{0, {}}, mask);
    hipLaunchKernelGGL(nms_sweep, dim3(1), dim3(64), 0, stream, mask, (const int *)nullptr, n, cb, n, keep32, n, out_count, (const int *)nullptr);
    hipLaunchKernelGGL(keep_to_i64, dim3((n + 255) / 256), dim3(256), 0, stream, keep32, out_count, n, (long long *)keep);
    return fd::check_launch("fd_rotated_nms");
}

extern "C" int fd_boxes_iou_bev(const float *a7, int na, const float *b7, int nb, float *out, fd_stream_t stream) {
    FD_REQUIRE(na >= 0 && nb >= 0, "fd_boxes_iou_bev: negative size");
    if (na == 0 || nb == 0) return FD_OK;
    FD_REQUIRE(a7 && b7 && out, "fd_boxes_iou_bev: null argument");
    int64_t total = (int64_t)na * nb;
    hipLaunchKernelGGL(iou_pairs, dim3((unsigned)((total + kPolyThreads - 1) / kPolyThreads)), dim3(kPolyThreads), 0, fd::as_stream(stream), a7, na, b7, nb, out);
    return fd::check_launch("fd_boxes_iou_bev");
}

#ifdef FD_MASK_DEBUG
extern "C" int fd_debug_mask_log(unsigned *out1024) { return hipMemcpyFromSymbol(out1024, HIP_SYMBOL(g_masklog), 4096) == hipSuccess ? 0 : -1; }
extern "C" int fd_debug_mask_counters(int *out8) { return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_maskdbg), 32) == hipSuccess ? 0 : -1; }
#endif
