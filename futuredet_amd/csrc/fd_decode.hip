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

__device__ float footprint_overlap(const Footprint &A, const Footprint &B) {
    float px[kMaxPoly], py[kMaxPoly], ang[kMaxPoly];
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
            px[n] = qx; py[n] = qy; ++n;
            sum_x = sum_x + qx; sum_y = sum_y + qy;
        }
    }
    // ---- corners of one rectangle inside the other, alternating B's k-th and A's k-th
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (inside_margin(A, B.vx[k], B.vy[k])) {
            sum_x = sum_x + B.vx[k]; sum_y = sum_y + B.vy[k];
            px[n] = B.vx[k]; py[n] = B.vy[k]; ++n;
        }
        if (inside_margin(B, A.vx[k], A.vy[k])) {
            sum_x = sum_x + A.vx[k]; sum_y = sum_y + A.vy[k];
            px[n] = A.vx[k]; py[n] = A.vy[k]; ++n;
        }
    }
    if (n < 3) return 0.f;  // no polygon (the fan below is empty or degenerate: area exactly 0)
    const float mx = sum_x / n, my = sum_y / n;
    // ---- angular order around the mean point: stable insertion sort, ascending, strict comparison
    for (int k = 0; k < n; ++k) ang[k] = atan2f(py[k] - my, px[k] - mx);
    for (int k = 1; k < n; ++k) {
        const float a = ang[k], x = px[k], y = py[k];
        int m = k;
        while (m > 0 && ang[m - 1] > a) {
            ang[m] = ang[m - 1]; px[m] = px[m - 1]; py[m] = py[m - 1];
            --m;
        }
        ang[m] = a; px[m] = x; py[m] = y;
    }
    // ---- fan around the first vertex
    float twice = 0.f;
    for (int k = 0; k < n - 1; ++k) {
        const float ux = px[k] - px[0], uy = py[k] - py[0], wx = px[k + 1] - px[0], wy = py[k + 1] - py[0];
        twice += ux * wy - uy * wx;
    }
    return fabsf(twice) * 0.5f;
}

__device__ inline float footprint_iou(const Footprint &A, const Footprint &B) {
    const float ov = footprint_overlap(A, B);
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
    if (tid == 0) sel_count[g] = S;
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

// The same selection with the keys of a group held in REGISTERS (PER per thread, cell = tid + 1024 j: coalesced loads, one pass over
// global memory instead of six) and ties resolved inside the radix selection: the selection runs over the 49-bit value
// (score bits << 17 | 0x1ffff - cell), which is unique per cell and orders equal scores by ascending cell -- exactly "all keys > T,
// then the first take_eq keys == T in cell order" of dec_select above, without its two extra passes and second block scan.
// Four histogram passes (12 + 12 + 12 + 13 bits), one block scan for the output slots (slot order is free: the bitonic sort over the
// unique (score, cell) words fixes the final order), then the same sort and box decode.  74 -> ~30 us on the saturated 180 x 180 map.
template <int PER>
__global__ void __launch_bounds__(kSelThreads) dec_select_reg(const unsigned *__restrict__ keys_all, MapView reg, MapView height, MapView dim, MapView rot,
                                                              DecCfg c, int npad, float *__restrict__ sel_boxes, float *__restrict__ nms_boxes,
                                                              float *__restrict__ sel_scores, int *__restrict__ sel_cell, int *__restrict__ sel_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long *s_sort = reinterpret_cast<unsigned long long *>(smem);     // [npad]
    int *s_hist = reinterpret_cast<int *>(smem + (size_t)npad * 8);                // [8192]
    int *s_misc = s_hist + 8192;                                                   // [32]
    const int g = blockIdx.x;
    const unsigned *keys = keys_all + (int64_t)g * c.HW;
    const int tid = threadIdx.x;
    unsigned k[PER];
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int cell = tid + j * kSelThreads;
        k[j] = cell < c.HW ? keys[cell] : 0u;
        cnt += k[j] != 0u;
    }
    block_sum_1024(cnt, s_misc);
    const int M = s_misc[16];
    __syncthreads();
    // the value a cell competes with: (score bits, 0x1ffff - cell), 49 bits, unique per cell -- kept as its two 32-bit halves
    // (no 64-bit temporaries: 32 .. 72 keys per thread leave little room in the 128 registers of a 1024-thread workgroup)
    unsigned Tk = 0u, Tc = 0u;  // select (k, ci) >= (Tk, Tc); M <= pre_max: every valid key (k >= 1)
    if (M > c.pre_max) {
        int need = c.pre_max;
        unsigned p0 = 0u, p1 = 0u, p2 = 0u;
        auto radix_pass = [&](int nb, auto digit_of, auto matches) -> unsigned {
            for (int i = tid; i < nb; i += kSelThreads) s_hist[i] = 0;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const unsigned ci = 0x1ffffu - (unsigned)(tid + j * kSelThreads);
                if (k[j] != 0u && matches(k[j], ci)) atomicAdd(&s_hist[digit_of(k[j], ci)], 1);
            }
            find_bin<8>(s_hist, nb, need, s_misc, s_misc + 20);
            const unsigned b = (unsigned)s_misc[20];
            need -= s_misc[21];
            __syncthreads();
            return b;
        };
        p0 = radix_pass(4096, [](unsigned kk, unsigned) { return (int)(kk >> 20); }, [](unsigned, unsigned) { return true; });
        p1 = radix_pass(4096, [](unsigned kk, unsigned) { return (int)((kk >> 8) & 0xfffu); }, [&](unsigned kk, unsigned) { return (kk >> 20) == p0; });
        const unsigned k24 = (p0 << 12) | p1;
        p2 = radix_pass(4096, [](unsigned kk, unsigned ci) { return (int)(((kk & 0xffu) << 4) | (ci >> 13)); }, [&](unsigned kk, unsigned) { return (kk >> 8) == k24; });
        Tk = (k24 << 8) | (p2 >> 4);
        const unsigned c4 = p2 & 0xfu;
        const unsigned p3 = radix_pass(8192, [](unsigned, unsigned ci) { return (int)(ci & 0x1fffu); }, [&](unsigned kk, unsigned ci) { return kk == Tk && (ci >> 13) == c4; });
        Tc = (c4 << 13) | p3;  // (need == 1 here: the values are unique)
    } else {
        Tk = 1u;
    }
    auto selected = [&](unsigned kk, unsigned ci) { return kk != 0u && (kk > Tk || (kk == Tk && ci >= Tc)); };
    for (int i = tid; i < npad; i += kSelThreads) s_sort[i] = 0ull;
    int n_sel = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) n_sel += selected(k[j], 0x1ffffu - (unsigned)(tid + j * kSelThreads));
    int pos = block_sum_1024(n_sel, s_misc);
    const int S = s_misc[16];  // = min(M, pre_max)
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const unsigned cell = (unsigned)(tid + j * kSelThreads);
        if (selected(k[j], 0x1ffffu - cell)) s_sort[pos++] = ((unsigned long long)k[j] << 32) | (unsigned)(0xffffffffu - cell);
    }
    __syncthreads();
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
    if (tid == 0) sel_count[g] = S;
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

// ---------------------------------------------------------------- stage 3: IoU mask words (upper triangle)
// One wave per (row, 64-column block): lane = column, so the 64-bit mask word of iou3d_nms_kernel.cu:296-305 is a
// single wave ballot.  The row's footprint is wave-uniform, the columns' footprints are four coalesced float4 loads.
// Pairs whose centres are farther apart than the two half-diagonals (+0.1 m, which covers the 1e-2 corner-inside
// margin) cannot produce a crossing point or an inside corner, so their overlap is exactly 0; they skip the geometry.
__global__ void __launch_bounds__(256) nms_mask(const float4 *__restrict__ planes, int64_t n_total, const int *__restrict__ counts, int n_max,
                                                int col_blocks, float thr, unsigned long long *__restrict__ mask_all) {
    const int g = blockIdx.z, cb = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = counts ? counts[g] : n_max;
    if (row >= n) return;
    const int rb = row >> 6;
    if (cb < rb || cb * 64 >= n) return;  // the sweep never reads words left of the diagonal (iou3d_nms.cpp:127)
    const int64_t base = (int64_t)g * n_max;
    const int col = cb * 64 + lane;
    bool hit = false;
    if (col < n && col > row) {
        const Footprint cur = load_footprint(planes, n_total, base + row), oth = load_footprint(planes, n_total, base + col);
        const float dx = cur.cx - oth.cx, dy = cur.cy - oth.cy;
        const float reach = cur.reach + oth.reach + 0.1f;
        if (dx * dx + dy * dy <= reach * reach) hit = footprint_iou(cur, oth) > thr;
    }
    const unsigned long long t = __ballot(hit);
    if (lane == 0) mask_all[((int64_t)g * n_max + row) * col_blocks + cb] = t;
}

// ---------------------------------------------------------------- stage 4: greedy sweep, one wave per group
__global__ void __launch_bounds__(64) nms_sweep(const unsigned long long *__restrict__ mask_all, const int *__restrict__ counts, int n_max,
                                                int col_blocks, int post_max, int *__restrict__ keep_all /*[G,post_cap]*/, int post_cap,
                                                int *__restrict__ out_count) {
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
    if (lane == 0) out_count[g] = kept;
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

__global__ void __launch_bounds__(256) iou_pairs(const float *__restrict__ a, int na, const float *__restrict__ b, int nb, float *__restrict__ out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)na * nb) return;
    int i = (int)(t / nb), j = (int)(t - (int64_t)i * nb);
    float ba[7], bb[7];
    for (int d = 0; d < 7; ++d) { ba[d] = a[i * 7 + d]; bb[d] = b[j * 7 + d]; }
    out[t] = footprint_iou(make_footprint(ba), make_footprint(bb));
}

struct DecWs {
    size_t keys, sel_boxes, nms_boxes, sel_scores, sel_cell, sel_count, mask, keep, foot, total;
    int col_blocks, npad;
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
    w.sel_count = take(sizeof(int) * (size_t)G);
    w.mask = take(sizeof(unsigned long long) * (size_t)G * pre_max * w.col_blocks);
    w.keep = take(sizeof(int) * (size_t)G * post_max);
    w.foot = take(sizeof(Footprint) * (size_t)G * pre_max);
    w.total = off;
    return w;
}

}  // namespace

// Final assembly of predict's output (center_head.py:559-570,606-607,672-697): output step s takes the boxes of group
// step_group[s], its two velocity channels vel_channel[s], vel_channel[s] + 1 of that group's velocity map at the kept cells, and the
// label offset label_of[s].  One packed row per kept box: x y z w l h vx vy yaw score label; rows k >= count are zero.
struct AsmSteps {
    int S;
    int group[kMaxSteps], vel_channel[kMaxSteps], label[kMaxSteps];
};
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

extern "C" int fd_centerpoint_decode_maps(const fd_map_view *hm, const fd_map_view *reg, const fd_map_view *height, const fd_map_view *dim,
                                          const fd_map_view *rot, int G, const fd_decode_cfg *cfg, float *out_boxes7, float *out_scores, int32_t *out_cell,
                                          int32_t *out_count, void *workspace, size_t workspace_bytes, fd_stream_t stream_) {
    FD_REQUIRE(hm && reg && height && dim && rot && cfg && out_boxes7 && out_scores && out_cell && out_count, "fd_centerpoint_decode: null argument");
    FD_REQUIRE(hm->data && reg->data && height->data && dim->data && rot->data, "fd_centerpoint_decode: null map");
    for (const fd_map_view *v : {hm, reg, height, dim, rot}) FD_REQUIRE(v->dtype == 0 || v->dtype == 1, "fd_centerpoint_decode: map dtype must be 0 (f32) or 1 (bf16)");
    FD_REQUIRE(G > 0 && cfg->H > 0 && cfg->W > 0, "fd_centerpoint_decode: bad shape");
    FD_REQUIRE(cfg->nms_pre_max >= 1 && cfg->nms_pre_max <= kMaxPre, "fd_centerpoint_decode: nms_pre_max must be in [1,%d]", kMaxPre);
    FD_REQUIRE(cfg->nms_post_max >= 1 && cfg->nms_post_max <= 128, "fd_centerpoint_decode: nms_post_max must be in [1,128]");
    DecCfg c;
    c.H = cfg->H; c.W = cfg->W; c.HW = cfg->H * cfg->W;
    c.osf = cfg->out_size_factor; c.vx = cfg->voxel_x; c.vy = cfg->voxel_y; c.px = cfg->pc_x; c.py = cfg->pc_y;
    c.thr = cfg->score_threshold;
    for (int i = 0; i < 6; ++i) c.rng[i] = cfg->center_range[i];
    c.iou_thr = cfg->nms_iou_threshold;
    c.pre_max = cfg->nms_pre_max; c.post_max = cfg->nms_post_max;
    c.hm_channels = cfg->hm_channels > 1 ? cfg->hm_channels : 1;
    FD_REQUIRE(c.hm_channels <= 16, "fd_centerpoint_decode: hm_channels must be <= 16");
    DecWs w = dec_layout(G, c.HW, c.pre_max, c.post_max);
    if (!workspace || workspace_bytes < w.total) {
        fd::set_error("fd_centerpoint_decode: workspace %zu < required %zu", workspace_bytes, w.total);
        return FD_EWORKSPACE;
    }
    hipStream_t stream = fd::as_stream(stream_);
    char *ws = (char *)workspace;
    unsigned *keys = (unsigned *)(ws + w.keys);
    float *sel_boxes = (float *)(ws + w.sel_boxes), *nms_boxes = (float *)(ws + w.nms_boxes), *sel_scores = (float *)(ws + w.sel_scores);
    int *sel_cell = (int *)(ws + w.sel_cell), *sel_count = (int *)(ws + w.sel_count), *keep = (int *)(ws + w.keep);
    unsigned long long *mask = (unsigned long long *)(ws + w.mask);
    const MapView vh = as_view(*hm), vr = as_view(*reg), vz = as_view(*height), vd = as_view(*dim), vt = as_view(*rot);
    hipLaunchKernelGGL(dec_keys, dim3((c.HW + 255) / 256, G), dim3(256), 0, stream, vh, vr, vz, c, keys);
    if (c.HW <= 32 * kSelThreads) {  // a group's keys fit the registers of one workgroup (72 keys per thread, the 270 x 270 map, spill)
        const size_t lds = (size_t)w.npad * 8 + 8192 * 4 + 32 * 4;
#define FD_SEL(PER)                                                                                                                              \
    hipLaunchKernelGGL(dec_select_reg<PER>, dim3(G), dim3(kSelThreads), lds, stream, keys, vr, vz, vd, vt, c, w.npad, sel_boxes, nms_boxes, sel_scores, \
                       sel_cell, sel_count)
        if (c.HW <= 8 * kSelThreads) FD_SEL(8);
        else FD_SEL(32);
#undef FD_SEL
    } else {
        const size_t lds = (size_t)w.npad * 8 + 4096 * 4 + 32 * 4;
        hipLaunchKernelGGL(dec_select, dim3(G), dim3(kSelThreads), lds, stream, keys, vr, vz, vd, vt, c, w.npad, sel_boxes, nms_boxes, sel_scores, sel_cell,
                           sel_count);
    }
    float4 *foot = (float4 *)(ws + w.foot);
    const int64_t n_total = (int64_t)G * c.pre_max;
    hipLaunchKernelGGL(footprint_kernel, dim3((unsigned)((n_total + 255) / 256)), dim3(256), 0, stream, nms_boxes, sel_count, c.pre_max, G, foot);
    hipLaunchKernelGGL(nms_mask, dim3((c.pre_max + 3) / 4, w.col_blocks, G), dim3(256), 0, stream, foot, n_total, sel_count, c.pre_max, w.col_blocks,
                       c.iou_thr, mask);
    hipLaunchKernelGGL(nms_sweep, dim3(G), dim3(64), 0, stream, mask, sel_count, c.pre_max, w.col_blocks, c.post_max, keep, c.post_max, out_count);
    hipLaunchKernelGGL(dec_gather, dim3(G), dim3(128), 0, stream, keep, out_count, c.post_max, c.pre_max, sel_boxes, sel_scores, sel_cell,
                       out_boxes7, out_scores, out_cell);
    return fd::check_launch("fd_centerpoint_decode");
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
    st.S = S;
    for (int s = 0; s < S; ++s) {
        FD_REQUIRE(step_group[s] >= 0 && step_vel_channel[s] >= 0, "fd_assemble_detections: negative step entry");
        st.group[s] = step_group[s]; st.vel_channel[s] = step_vel_channel[s]; st.label[s] = step_label[s];
    }
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
    hipLaunchKernelGGL(nms_mask, dim3((n + 3) / 4, cb, 1), dim3(256), 0, stream, foot, (int64_t)n, (const int *)nullptr, n, cb, thresh, mask);
    hipLaunchKernelGGL(nms_sweep, dim3(1), dim3(64), 0, stream, mask, (const int *)nullptr, n, cb, n, keep32, n, out_count);
    hipLaunchKernelGGL(keep_to_i64, dim3((n + 255) / 256), dim3(256), 0, stream, keep32, out_count, n, (long long *)keep);
    return fd::check_launch("fd_rotated_nms");
}

extern "C" int fd_boxes_iou_bev(const float *a7, int na, const float *b7, int nb, float *out, fd_stream_t stream) {
    FD_REQUIRE(na >= 0 && nb >= 0, "fd_boxes_iou_bev: negative size");
    if (na == 0 || nb == 0) return FD_OK;
    FD_REQUIRE(a7 && b7 && out, "fd_boxes_iou_bev: null argument");
    int64_t total = (int64_t)na * nb;
    hipLaunchKernelGGL(iou_pairs, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, fd::as_stream(stream), a7, na, b7, nb, out);
    return fd::check_launch("fd_boxes_iou_bev");
}
