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

// ---------------------------------------------------------------- rotated IoU (iou3d_nms_kernel.cu:14-234)
struct Pt { float x, y; };
__device__ inline float cross2(Pt a, Pt b) { return a.x * b.y - a.y * b.x; }
__device__ inline float cross3(Pt p1, Pt p2, Pt p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }
__device__ inline int check_rect_cross(Pt p1, Pt p2, Pt q1, Pt q2) {
    return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
           fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}
__device__ inline int check_in_box2d(const float *box, Pt p) {
    const float MARGIN = 1e-2f;
    float cx = box[0], cy = box[1];
    float ac = cosf(-box[6]), as = sinf(-box[6]);
    float rx = (p.x - cx) * ac + (p.y - cy) * (-as);
    float ry = (p.x - cx) * as + (p.y - cy) * ac;
    return (fabsf(rx) < box[3] / 2 + MARGIN && fabsf(ry) < box[4] / 2 + MARGIN);
}
__device__ inline int intersection(Pt p1, Pt p0, Pt q1, Pt q0, Pt &ans) {
    const float EPS = 1e-8f;
    if (check_rect_cross(p0, p1, q0, q1) == 0) return 0;
    float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0), s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > EPS) {
        ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        float D = a0 * b1 - a1 * b0;
        ans.x = (b0 * c1 - b1 * c0) / D;
        ans.y = (a1 * c0 - a0 * c1) / D;
    }
    return 1;
}
__device__ inline void rotate_around_center(Pt c, float ac, float as, Pt &p) {
    float nx = (p.x - c.x) * ac + (p.y - c.y) * (-as) + c.x;
    float ny = (p.x - c.x) * as + (p.y - c.y) * ac + c.y;
    p.x = nx; p.y = ny;
}
__device__ inline int point_cmp(Pt a, Pt b, Pt c) { return atan2f(a.y - c.y, a.x - c.x) > atan2f(b.y - c.y, b.x - c.x); }

__device__ float box_overlap(const float *box_a, const float *box_b) {
    float a_angle = box_a[6], b_angle = box_b[6];
    float a_dx_half = box_a[3] / 2, b_dx_half = box_b[3] / 2, a_dy_half = box_a[4] / 2, b_dy_half = box_b[4] / 2;
    float a_x1 = box_a[0] - a_dx_half, a_y1 = box_a[1] - a_dy_half, a_x2 = box_a[0] + a_dx_half, a_y2 = box_a[1] + a_dy_half;
    float b_x1 = box_b[0] - b_dx_half, b_y1 = box_b[1] - b_dy_half, b_x2 = box_b[0] + b_dx_half, b_y2 = box_b[1] + b_dy_half;
    Pt center_a = {box_a[0], box_a[1]}, center_b = {box_b[0], box_b[1]};
    Pt ca[5] = {{a_x1, a_y1}, {a_x2, a_y1}, {a_x2, a_y2}, {a_x1, a_y2}, {0, 0}};
    Pt cb[5] = {{b_x1, b_y1}, {b_x2, b_y1}, {b_x2, b_y2}, {b_x1, b_y2}, {0, 0}};
    float a_cos = cosf(a_angle), a_sin = sinf(a_angle), b_cos = cosf(b_angle), b_sin = sinf(b_angle);
    for (int k = 0; k < 4; ++k) {
        rotate_around_center(center_a, a_cos, a_sin, ca[k]);
        rotate_around_center(center_b, b_cos, b_sin, cb[k]);
    }
    ca[4] = ca[0]; cb[4] = cb[0];
    Pt cross_points[16];
    Pt poly_center = {0, 0};
    int cnt = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            Pt ans;
            if (intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], ans)) {
                cross_points[cnt] = ans;
                poly_center.x = poly_center.x + ans.x;
                poly_center.y = poly_center.y + ans.y;
                cnt++;
            }
        }
    for (int k = 0; k < 4; ++k) {
        if (check_in_box2d(box_a, cb[k])) {
            poly_center.x = poly_center.x + cb[k].x; poly_center.y = poly_center.y + cb[k].y;
            cross_points[cnt] = cb[k]; cnt++;
        }
        if (check_in_box2d(box_b, ca[k])) {
            poly_center.x = poly_center.x + ca[k].x; poly_center.y = poly_center.y + ca[k].y;
            cross_points[cnt] = ca[k]; cnt++;
        }
    }
    poly_center.x /= cnt;
    poly_center.y /= cnt;
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i)
            if (point_cmp(cross_points[i], cross_points[i + 1], poly_center)) {
                Pt t = cross_points[i]; cross_points[i] = cross_points[i + 1]; cross_points[i + 1] = t;
            }
    float area = 0;
    for (int k = 0; k < cnt - 1; ++k) {
        Pt u = {cross_points[k].x - cross_points[0].x, cross_points[k].y - cross_points[0].y};
        Pt v = {cross_points[k + 1].x - cross_points[0].x, cross_points[k + 1].y - cross_points[0].y};
        area += cross2(u, v);
    }
    return (float)(fabsf(area) / 2.0);
}
__device__ inline float iou_bev(const float *box_a, const float *box_b) {
    float sa = box_a[3] * box_a[4], sb = box_b[3] * box_b[4];
    float s_overlap = box_overlap(box_a, box_b);
    return s_overlap / fmaxf(sa + sb - s_overlap, 1e-8f);
}

// ---------------------------------------------------------------- stage 1: score keys
struct DecCfg {
    int H, W, HW;
    float osf, vx, vy, px, py, thr;
    float rng[6];
    float iou_thr;
    int pre_max, post_max;
};

__device__ inline void cell_center(const DecCfg &c, const float *reg, int cell, float &x, float &y) {
    // center_head.py:641-649: xs = (j + reg_x) * out_size_factor * voxel_size[0] + pc_range[0], left to right in f32
    const int i = cell / c.W, j = cell - i * c.W;
    float xs = __fadd_rn((float)j, reg[cell]);
    float ys = __fadd_rn((float)i, reg[c.HW + cell]);
    x = __fadd_rn(__fmul_rn(__fmul_rn(xs, c.osf), c.vx), c.px);
    y = __fadd_rn(__fmul_rn(__fmul_rn(ys, c.osf), c.vy), c.py);
}

__global__ void __launch_bounds__(256) dec_keys(const float *__restrict__ hm, int64_t hm_gs, const float *__restrict__ reg, int64_t reg_gs,
                                                const float *__restrict__ height, int64_t h_gs, DecCfg c, unsigned *__restrict__ keys) {
    const int g = blockIdx.y;
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= c.HW) return;
    const float logit = hm[g * hm_gs + cell];
    const float score = 1.0f / (1.0f + expf(-logit));
    float x, y;
    cell_center(c, reg + g * reg_gs, cell, x, y);
    const float z = height[g * h_gs + cell];
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
__device__ inline void find_bin(const int *hist, int nbins, int need, int *sm_scan, int *sm_out /*[2]: bin, count above*/) {
    __syncthreads();
    int h[4];
    int local = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int bin = nbins - 1 - (4 * (int)threadIdx.x + q);
        h[q] = bin >= 0 ? hist[bin] : 0;
        local += h[q];
    }
    int acc = block_sum_1024(local, sm_scan);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int bin = nbins - 1 - (4 * (int)threadIdx.x + q);
        if (bin >= 0 && acc < need && acc + h[q] >= need) {
            sm_out[0] = bin;
            sm_out[1] = acc;
        }
        acc += h[q];
    }
    __syncthreads();
}

__global__ void __launch_bounds__(kSelThreads) dec_select(const unsigned *__restrict__ keys_all, const float *__restrict__ reg, int64_t reg_gs,
                                                          const float *__restrict__ height, int64_t h_gs, const float *__restrict__ dim,
                                                          int64_t dim_gs, const float *__restrict__ rot, int64_t rot_gs, DecCfg c, int npad,
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
    int run_gt = 0, run_eq = 0;
    const bool all_valid = (M <= c.pre_max);
    for (int base = 0; base < c.HW; base += kSelThreads) {
        const int i = base + tid;
        unsigned k = i < c.HW ? keys[i] : 0u;
        int fgt = all_valid ? (k != 0u) : (k > T);
        int feq = all_valid ? 0 : (k == T);
        // two scans packed in one: counts <= 1024 each fit in 16 bits
        int packed = fgt | (feq << 16);
        int ex = block_sum_1024(packed, s_misc);
        int tot = s_misc[16];
        __syncthreads();
        int pos_gt = run_gt + (ex & 0xffff), rank_eq = run_eq + (ex >> 16);
        run_gt += tot & 0xffff;
        run_eq += tot >> 16;
        if (fgt) {
            // slots [0, n_gt) hold the > T keys; position known only after all chunks for == keys, so place
            // > keys from the front and == keys from the back of the pre_max window
            s_sort[pos_gt] = ((unsigned long long)k << 32) | (unsigned)(0xffffffffu - (unsigned)i);
        } else if (feq && rank_eq < take_eq) {
            s_sort[c.pre_max - 1 - rank_eq] = ((unsigned long long)k << 32) | (unsigned)(0xffffffffu - (unsigned)i);
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
        cell_center(c, reg + g * reg_gs, cell, x, y);
        const float z = height[g * h_gs + cell];
        const float *dm = dim + g * dim_gs;
        const float d0 = expf(dm[cell]), d1 = expf(dm[c.HW + cell]), d2 = expf(dm[2 * c.HW + cell]);
        const float *rt = rot + g * rot_gs;
        const float yaw = atan2f(rt[cell], rt[c.HW + cell]);
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
// single wave ballot.  Pairs whose centres are farther apart than the two half-diagonals (+0.1 m, which covers
// the reference's 1e-2 corner-inside MARGIN) cannot produce an intersection point or an inside corner, so the
// reference routine returns exactly 0 for them; they skip the geometry.
__global__ void __launch_bounds__(256) nms_mask(const float *__restrict__ boxes_all, const int *__restrict__ counts, int n_max, int col_blocks,
                                                float thr, unsigned long long *__restrict__ mask_all) {
    const int g = blockIdx.z, cb = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = counts ? counts[g] : n_max;
    if (row >= n) return;
    const int rb = row >> 6;
    if (cb < rb || cb * 64 >= n) return;  // the sweep never reads words left of the diagonal (iou3d_nms.cpp:127)
    const float *boxes = boxes_all + (int64_t)g * n_max * 7;
    float cur[7], oth[7];
#pragma unroll
    for (int d = 0; d < 7; ++d) cur[d] = boxes[(int64_t)row * 7 + d];
    const int col = cb * 64 + lane;
    bool hit = false;
    if (col < n && col > row) {
#pragma unroll
        for (int d = 0; d < 7; ++d) oth[d] = boxes[(int64_t)col * 7 + d];
        const float dx = cur[0] - oth[0], dy = cur[1] - oth[1];
        const float ra = 0.5f * sqrtf(cur[3] * cur[3] + cur[4] * cur[4]), rb2 = 0.5f * sqrtf(oth[3] * oth[3] + oth[4] * oth[4]);
        const float reach = ra + rb2 + 0.1f;
        if (dx * dx + dy * dy <= reach * reach) hit = iou_bev(cur, oth) > thr;
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
    out[t] = iou_bev(ba, bb);
}

struct DecWs {
    size_t keys, sel_boxes, nms_boxes, sel_scores, sel_cell, sel_count, mask, keep, total;
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
    w.total = off;
    return w;
}

}  // namespace

extern "C" size_t fd_decode_workspace_bytes(int G, const fd_decode_cfg *cfg) {
    if (!cfg || G <= 0) return 0;
    return dec_layout(G, cfg->H * cfg->W, cfg->nms_pre_max, cfg->nms_post_max).total;
}

extern "C" int fd_centerpoint_decode(const float *hm, int64_t hm_gs, const float *reg, int64_t reg_gs, const float *height, int64_t h_gs,
                                     const float *dim, int64_t dim_gs, const float *rot, int64_t rot_gs, int G, const fd_decode_cfg *cfg,
                                     float *out_boxes7, float *out_scores, int32_t *out_cell, int32_t *out_count, void *workspace,
                                     size_t workspace_bytes, fd_stream_t stream_) {
    FD_REQUIRE(hm && reg && height && dim && rot && cfg && out_boxes7 && out_scores && out_cell && out_count, "fd_centerpoint_decode: null argument");
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
    hipLaunchKernelGGL(dec_keys, dim3((c.HW + 255) / 256, G), dim3(256), 0, stream, hm, hm_gs, reg, reg_gs, height, h_gs, c, keys);
    const size_t lds = (size_t)w.npad * 8 + 4096 * 4 + 32 * 4;
    hipLaunchKernelGGL(dec_select, dim3(G), dim3(kSelThreads), lds, stream, keys, reg, reg_gs, height, h_gs, dim, dim_gs, rot, rot_gs, c, w.npad,
                       sel_boxes, nms_boxes, sel_scores, sel_cell, sel_count);
    hipLaunchKernelGGL(nms_mask, dim3((c.pre_max + 3) / 4, w.col_blocks, G), dim3(256), 0, stream, nms_boxes, sel_count, c.pre_max, w.col_blocks,
                       c.iou_thr, mask);
    hipLaunchKernelGGL(nms_sweep, dim3(G), dim3(64), 0, stream, mask, sel_count, c.pre_max, w.col_blocks, c.post_max, keep, c.post_max, out_count);
    hipLaunchKernelGGL(dec_gather, dim3(G), dim3(128), 0, stream, keep, out_count, c.post_max, c.pre_max, sel_boxes, sel_scores, sel_cell,
                       out_boxes7, out_scores, out_cell);
    return fd::check_launch("fd_centerpoint_decode");
}

extern "C" size_t fd_nms_workspace_bytes(int n) {
    if (n <= 0) return 256;
    size_t cb = (size_t)(n + 63) / 64;
    return fd::align_up(sizeof(unsigned long long) * (size_t)n * cb, 256) + fd::align_up(sizeof(int) * (size_t)n, 256);
}

extern "C" int fd_rotated_nms(const float *boxes7, int n, float thresh, int64_t *keep, int32_t *out_count, void *workspace,
                              size_t workspace_bytes, fd_stream_t stream_) {
    FD_REQUIRE(out_count && (n == 0 || (boxes7 && keep)), "fd_rotated_nms: null argument");
    FD_REQUIRE(n >= 0 && n <= kMaxPre, "fd_rotated_nms: n must be in [0,%d]", kMaxPre);
    hipStream_t stream = fd::as_stream(stream_);
    if (n == 0) {
        (void)hipMemsetAsync(out_count, 0, sizeof(int32_t), stream);
        return fd::check_launch("fd_rotated_nms");
    }
    if (!workspace || workspace_bytes < fd_nms_workspace_bytes(n)) {
        fd::set_error("fd_rotated_nms: workspace too small");
        return FD_EWORKSPACE;
    }
    const int cb = (n + 63) / 64;
    unsigned long long *mask = (unsigned long long *)workspace;
    int *keep32 = (int *)((char *)workspace + fd::align_up(sizeof(unsigned long long) * (size_t)n * cb, 256));
    hipLaunchKernelGGL(nms_mask, dim3((n + 3) / 4, cb, 1), dim3(256), 0, stream, boxes7, (const int *)nullptr, n, cb, thresh, mask);
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
