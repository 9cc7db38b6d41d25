// Voxelizer + fused mean reader for gfx950.
//
// Reproduces the sequential semantics of the reference voxelizer
// (det3d/ops/point_cloud/point_cloud_ops.py:7-55: first-come voxel ids, first max_voxels voxels, first
// max_points points per voxel in input order, float32 floor((p-lo)/vs) with a true division) with a
// deterministic parallel schedule:
//   K1 hash   : point -> cell key -> open-addressing slot (atomicCAS), atomicMin(first point), atomicAdd(count)
//   K2-K3 scan: voxel id = rank of the voxel's first point among all first points (exclusive scan over
//               points); the same scan carries the per-voxel point counts -> CSR bucket offsets
//   K4 fill   : point index -> its voxel's bucket (order inside a bucket is arbitrary)
//   K5 emit   : one lane per voxel selects the max_points smallest point indices of its bucket with an
//               unrolled insertion network, copies / sums the points in ascending order
// Wave-level ballots/prefix counts keep the atomics to one per wave where possible (the compiler folds the
// per-lane atomicAdd(…,1) on a uniform address; the hash itself is lane-divergent by nature).
#include "fd_common.h"

namespace {

constexpr int kMaxP = 64;    // bound on max_points (vox_emit is instantiated for 16 / 32 / 64 slots)
constexpr int kMaxNd = 8;    // compile-time bound on point width
constexpr int kScanThreads = 256;
constexpr int kScanItems = 4;
constexpr int kScanTile = kScanThreads * kScanItems;

struct VoxParams {
    float lo[3], vs[3];
    int grid[3];  // x,y,z
    int ndim, max_points;
    int max_voxels;
    unsigned mask;  // hash table size - 1
};

__device__ inline unsigned hash_key(int key) { return (unsigned)key * 2654435761u; }

__global__ void __launch_bounds__(256) vox_hash(const float *__restrict__ pts, int n, const int *__restrict__ n_dev, VoxParams p,
                                                int *__restrict__ keys, int *__restrict__ first,
                                                int *__restrict__ cnt, int *__restrict__ pslot) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    n = fd::device_count(n, n_dev);
    if (i >= n) return;
    const float *q = pts + (int64_t)i * p.ndim;
    int c[3];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        // float32, true division, floor -- not a reciprocal multiply (cell membership at edges)
        float v = floorf(__fdiv_rn(__fsub_rn(q[j], p.lo[j]), p.vs[j]));
        ok = ok && (v >= 0.0f) && (v < (float)p.grid[j]);
        c[j] = (int)v;
    }
    if (!ok) {
        pslot[i] = -1;
        return;
    }
    int key = (c[2] * p.grid[1] + c[1]) * p.grid[0] + c[0];  // (z,y,x) row-major
    unsigned slot = (hash_key(key) >> 7) & p.mask;
    while (true) {
        int prev = atomicCAS(&keys[slot], -1, key);
        if (prev == -1 || prev == key) break;
        slot = (slot + 1) & p.mask;
    }
    atomicMin(&first[slot], i);
    atomicAdd(&cnt[slot], 1);
    pslot[i] = (int)slot;
}

// block-wide exclusive scan of two ints per thread (kScanThreads threads)
__device__ inline void block_scan2(int &a, int &b, int &ta, int &tb, int *sm /*[2*4+2]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int ia = a, ib = b;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int ua = __shfl_up(ia, off), ub = __shfl_up(ib, off);
        if (lane >= off) { ia += ua; ib += ub; }
    }
    if (lane == 63) { sm[wave] = ia; sm[4 + wave] = ib; }
    __syncthreads();
    int wa = 0, wb = 0, sa = 0, sb = 0;
#pragma unroll
    for (int w = 0; w < kScanThreads / 64; ++w) {
        if (w < wave) { wa += sm[w]; wb += sm[4 + w]; }
        sa += sm[w]; sb += sm[4 + w];
    }
    __syncthreads();
    a = wa + ia - a;  // exclusive
    b = wb + ib - b;
    ta = sa; tb = sb;
}

// block-wide sums of two ints per thread, returned to every thread
__device__ inline void block_sum2(int &a, int &b, int *sm /*[8]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off);
        b += __shfl_xor(b, off);
    }
    if (lane == 0) { sm[wave] = a; sm[4 + wave] = b; }
    __syncthreads();
    a = 0; b = 0;
#pragma unroll
    for (int w = 0; w < kScanThreads / 64; ++w) { a += sm[w]; b += sm[4 + w]; }
    __syncthreads();
}

__device__ inline void load_flags(const int *pslot, const int *first, const int *cnt, int i, int n, int &f, int &c) {
    f = 0; c = 0;
    if (i < n) {
        int s = pslot[i];
        if (s >= 0 && first[s] == i) { f = 1; c = cnt[s]; }
    }
}

__global__ void __launch_bounds__(kScanThreads) vox_scan1(const int *__restrict__ pslot, const int *__restrict__ first,
                                                          const int *__restrict__ cnt, int n, const int *__restrict__ n_dev,
                                                          int *__restrict__ bsum) {
    __shared__ int sm[10];
    n = fd::device_count(n, n_dev);
    int base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    int f = 0, c = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        int ff, cc;
        load_flags(pslot, first, cnt, base + k, n, ff, cc);
        f += ff; c += cc;
    }
    int tf, tc;
    block_scan2(f, c, tf, tc, sm);
    if (threadIdx.x == 0) { bsum[2 * blockIdx.x] = tf; bsum[2 * blockIdx.x + 1] = tc; }
}

// (no second pass over the block sums: a block of the third pass sums the totals of the blocks before it itself -- a few hundred
// values -- which costs less than the launch it replaces)
__global__ void __launch_bounds__(kScanThreads) vox_scan3(const int *__restrict__ pslot, const int *__restrict__ first,
                                                          const int *__restrict__ cnt, int n, const int *__restrict__ n_dev,
                                                          const int *__restrict__ bsum, int max_voxels, int *__restrict__ vid,
                                                          int *__restrict__ boff, int *__restrict__ vslot, int *__restrict__ num_voxels) {
    __shared__ int sm[10];
    n = fd::device_count(n, n_dev);
    int base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    int ff[kScanItems], cc[kScanItems];
    int f = 0, c = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        load_flags(pslot, first, cnt, base + k, n, ff[k], cc[k]);
        f += ff[k]; c += cc[k];
    }
    int tf, tc;
    block_scan2(f, c, tf, tc, sm);
    int pf = 0, pc = 0;  // totals of the blocks in front of this one
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += kScanThreads) { pf += bsum[2 * j]; pc += bsum[2 * j + 1]; }
    block_sum2(pf, pc, sm);
    f += pf;
    c += pc;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (ff[k]) {
            int s = pslot[base + k];
            if (f < max_voxels) {
                vid[s] = f;
                boff[f] = c;
                vslot[f] = s;
            } else {
                vid[s] = -1;
            }
        }
        f += ff[k]; c += cc[k];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kScanThreads - 1) {
        num_voxels[0] = f < max_voxels ? f : max_voxels;
    }
}

__global__ void __launch_bounds__(256) vox_fill(const int *__restrict__ pslot, int n, const int *__restrict__ n_dev,
                                                const int *__restrict__ vid, const int *__restrict__ boff, int *__restrict__ cursor,
                                                int *__restrict__ bucket) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    n = fd::device_count(n, n_dev);
    if (i >= n) return;
    int s = pslot[i];
    if (s < 0) return;
    int v = vid[s];
    if (v < 0) return;
    int pos = boff[v] + atomicAdd(&cursor[v], 1);
    bucket[pos] = i;
}

template <int MAXP>
__global__ void __launch_bounds__(256) vox_emit(const float *__restrict__ pts, VoxParams p, const int *__restrict__ num_voxels,
                                                const int *__restrict__ keys, const int *__restrict__ vslot,
                                                const int *__restrict__ boff, const int *__restrict__ cursor,
                                                const int *__restrict__ bucket, int batch_idx, float *__restrict__ out_voxels,
                                                float *__restrict__ out_mean, int mean_stride, int *__restrict__ out_coors,
                                                int coor_cols, int *__restrict__ out_num) {
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= num_voxels[0]) return;
    const int nv = cursor[v];
    const int *b = bucket + boff[v];
    int sel[MAXP];
#pragma unroll
    for (int k = 0; k < MAXP; ++k) sel[k] = 0x7fffffff;
    for (int t = 0; t < nv; ++t) {
        int x = b[t];
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {  // keep sel[] ascending; x carries the displaced larger value
            int lo = min(sel[k], x), hi = max(sel[k], x);
            sel[k] = lo; x = hi;
        }
    }
    const int np = nv < p.max_points ? nv : p.max_points;
    float sum[kMaxNd];
#pragma unroll
    for (int d = 0; d < kMaxNd; ++d) sum[d] = 0.0f;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        if (k < p.max_points) {
            const bool live = k < np;
            const float *q = pts + (int64_t)(live ? sel[k] : 0) * p.ndim;
#pragma unroll
            for (int d = 0; d < kMaxNd; ++d) {
                if (d < p.ndim) {
                    float val = live ? q[d] : 0.0f;
                    sum[d] = __fadd_rn(sum[d], val);
                    if (out_voxels) out_voxels[((int64_t)v * p.max_points + k) * p.ndim + d] = val;
                }
            }
        }
    }
    if (out_mean) {
        const float cntf = (float)np;
#pragma unroll
        for (int d = 0; d < kMaxNd; ++d)
            if (d < p.ndim) out_mean[(int64_t)v * mean_stride + d] = __fdiv_rn(sum[d], cntf);
        for (int d = p.ndim; d < mean_stride; ++d) out_mean[(int64_t)v * mean_stride + d] = 0.0f;
    }
    int key = keys[vslot[v]];
    int x = key % p.grid[0];
    int t = key / p.grid[0];
    int y = t % p.grid[1];
    int z = t / p.grid[1];
    int *oc = out_coors + (int64_t)v * coor_cols;
    if (coor_cols == 4) { oc[0] = batch_idx; oc[1] = z; oc[2] = y; oc[3] = x; }
    else { oc[0] = z; oc[1] = y; oc[2] = x; }
    out_num[v] = np;
}

struct VoxWs {
    size_t keys, first, cnt, cursor, vid, pslot, bsum, boff, vslot, bucket, total;
    unsigned table;
};

VoxWs vox_layout(int64_t n, int64_t max_voxels) {
    VoxWs w;
    unsigned table = 1024;
    while ((int64_t)table < 2 * n) table <<= 1;
    w.table = table;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += fd::align_up(bytes, 256); return o; };
    w.keys = take(sizeof(int) * table);
    w.first = take(sizeof(int) * table);
    w.cnt = take(sizeof(int) * table);
    w.cursor = take(sizeof(int) * (size_t)(max_voxels + 1));
    w.vid = take(sizeof(int) * table);
    w.pslot = take(sizeof(int) * (size_t)(n + 1));
    w.bsum = take(sizeof(int) * 2 * (size_t)((n + kScanTile - 1) / kScanTile + 1));
    w.boff = take(sizeof(int) * (size_t)(max_voxels + 1));
    w.vslot = take(sizeof(int) * (size_t)(max_voxels + 1));
    w.bucket = take(sizeof(int) * (size_t)(n + 1));
    w.total = off;
    return w;
}

}  // namespace

extern "C" size_t fd_voxelize_workspace_bytes(int64_t n_points, int64_t max_voxels) {
    if (n_points < 0 || max_voxels < 0) return 0;
    return vox_layout(n_points, max_voxels).total;
}

extern "C" int fd_voxelize(const float *points, int64_t n_points, const int32_t *n_points_dev, int ndim, const float *range6, const float *vsize3,
                           int max_points, int64_t max_voxels, int batch_idx, float *out_voxels, float *out_mean,
                           int mean_stride, int32_t *out_coors, int coor_cols, int32_t *out_num_points,
                           int32_t *out_num_voxels, void *workspace, size_t workspace_bytes, fd_stream_t stream_) {
    FD_REQUIRE(range6 && vsize3 && out_coors && out_num_points && out_num_voxels, "fd_voxelize: null argument");
    FD_REQUIRE(n_points >= 0 && n_points < (1ll << 30), "fd_voxelize: n_points out of range");
    FD_REQUIRE(ndim >= 3 && ndim <= kMaxNd, "fd_voxelize: ndim must be in [3,%d]", kMaxNd);
    FD_REQUIRE(max_points >= 1 && max_points <= kMaxP, "fd_voxelize: max_points must be in [1,%d]", kMaxP);
    FD_REQUIRE(max_voxels >= 0 && max_voxels < (1ll << 30), "fd_voxelize: max_voxels out of range");
    FD_REQUIRE(coor_cols == 3 || coor_cols == 4, "fd_voxelize: coor_cols must be 3 or 4");
    FD_REQUIRE(!out_mean || mean_stride >= ndim, "fd_voxelize: mean_stride < ndim");
    hipStream_t stream = fd::as_stream(stream_);
    VoxParams p;
    int64_t cells = 1;
    for (int j = 0; j < 3; ++j) {
        p.lo[j] = range6[j];
        p.vs[j] = vsize3[j];
        // point_cloud_ops.py:24-29: round((hi-lo)/vs) in float32, half-to-even
        volatile float g = (range6[3 + j] - range6[j]) / vsize3[j];
        p.grid[j] = (int)rintf(g);
        FD_REQUIRE(p.grid[j] > 0, "fd_voxelize: empty grid");
        cells *= p.grid[j];
    }
    FD_REQUIRE(cells < (1ll << 31), "fd_voxelize: grid has more than 2^31 cells");
    p.ndim = ndim;
    p.max_points = max_points;
    p.max_voxels = (int)max_voxels;
    if (n_points == 0 || max_voxels == 0) {
        fd::fill_words(out_num_voxels, 0u, 1, stream);
        return fd::check_launch("fd_voxelize(empty)");
    }
    FD_REQUIRE(points, "fd_voxelize: null points");
    VoxWs w = vox_layout(n_points, max_voxels);
    if (workspace_bytes < w.total || !workspace) {
        fd::set_error("fd_voxelize: workspace %zu < required %zu", workspace_bytes, w.total);
        return FD_EWORKSPACE;
    }
    p.mask = w.table - 1;
    char *ws = (char *)workspace;
    int *keys = (int *)(ws + w.keys), *first = (int *)(ws + w.first), *cnt = (int *)(ws + w.cnt);
    int *cursor = (int *)(ws + w.cursor), *vid = (int *)(ws + w.vid), *pslot = (int *)(ws + w.pslot);
    int *bsum = (int *)(ws + w.bsum), *boff = (int *)(ws + w.boff), *vslot = (int *)(ws + w.vslot);
    int *bucket = (int *)(ws + w.bucket);
    const int n = (int)n_points;
    // (a kernel, not hipMemsetAsync: see fd::fill_words; one launch for the three regions -- cnt + cursor are adjacent)
    fd::fill_words3(keys, 0xffffffffu, w.table, first, 0x7f7f7f7fu, w.table, cnt, 0u, (w.vid - w.cnt) / sizeof(int), stream);
    const int nb = (n + 255) / 256;
    const int nsb = (n + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(vox_hash, dim3(nb), dim3(256), 0, stream, points, n, n_points_dev, p, keys, first, cnt, pslot);
    hipLaunchKernelGGL(vox_scan1, dim3(nsb), dim3(kScanThreads), 0, stream, pslot, first, cnt, n, n_points_dev, bsum);
    hipLaunchKernelGGL(vox_scan3, dim3(nsb), dim3(kScanThreads), 0, stream, pslot, first, cnt, n, n_points_dev, bsum, p.max_voxels,
                       vid, boff, vslot, out_num_voxels);
    hipLaunchKernelGGL(vox_fill, dim3(nb), dim3(256), 0, stream, pslot, n, n_points_dev, vid, boff, cursor, bucket);
    const int64_t vmax = n_points < max_voxels ? n_points : max_voxels;
#define FD_EMIT(MP)                                                                                                        \
    hipLaunchKernelGGL(vox_emit<MP>, dim3((unsigned)((vmax + 255) / 256)), dim3(256), 0, stream, points, p, out_num_voxels, keys, \
                       vslot, boff, cursor, bucket, batch_idx, out_voxels, out_mean, mean_stride, out_coors, coor_cols,           \
                       out_num_points)
    if (max_points <= 16) FD_EMIT(16);       // VoxelNet configs: 10
    else if (max_points <= 32) FD_EMIT(32);  // PointPillars configs: 20
    else FD_EMIT(64);                        // points_to_voxel's default of 35
#undef FD_EMIT
    return fd::check_launch("fd_voxelize");
}
