// Voxelizer + fused mean reader for gfx950.
//
// Reproduces the sequential semantics of the reference voxelizer
// (det3d/ops/point_cloud/point_cloud_ops.py:7-55: first-come voxel ids, first max_voxels voxels, first
// max_points points per voxel in input order, float32 floor((p-lo)/vs) with a true division) with a
// deterministic parallel schedule.  Round 5 (the round-1 form moved 210 MB for a 6.3-MB cloud in six launches: three 4-byte tables hit
// by three atomics per point, two passes over the points for the scan, a fill pass with a fourth atomic):
//   vox_init   one 16-byte entry per hash slot {key, aux, (first point << 32 | point count)}: the whole record of a voxel in ONE line;
//              the scan's tile status words
//   vox_hash   point -> cell key -> slot (atomicCAS on the key), then ONE 64-bit CAS that lowers the first point and raises the count;
//              the count before the raise is the point's arrival rank in its voxel (kept: it replaces the fill pass's atomic)
//   vox_scan   single pass with decoupled look-back over the points: voxel id = rank of the voxel's first point among all first
//              points, bucket offset = running sum of the counts of the voxels with more than one point; per kept voxel its key,
//              first point, count and offset go to dense arrays (the emit pass never touches the hash table)
//   vox_place  point -> bucket[offset of its voxel + its arrival rank]: no atomics; skipped outright (device-side flag) when every
//              voxel holds one point
//   vox_emit   one lane per voxel: a single-point voxel IS its first point; otherwise the max_points smallest point indices of the
//              bucket by an unrolled insertion network, summed in ascending order; rows leave through LDS as 16-byte row-contiguous stores
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

struct __attribute__((aligned(16))) VoxEntry {
    int key;                  // cell key, -1 = free
    int aux;                  // after vox_scan: bucket offset of a kept voxel with more than one point, else -1
    unsigned long long fc;    // first point << 32 | points
};
constexpr unsigned long long kFcInit = 0x7fffffffull << 32;

__device__ inline unsigned hash_key(int key) { return (unsigned)key * 2654435761u; }

__global__ void __launch_bounds__(256) vox_init(VoxEntry *__restrict__ table, unsigned n_slots, unsigned long long *__restrict__ status, int n_status) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_slots) {
        uint4 v;
        v.x = 0xffffffffu; v.y = 0u; v.z = 0u; v.w = 0x7fffffffu;  // key -1, aux 0, fc = kFcInit (little endian: low word = count)
        reinterpret_cast<uint4 *>(table)[i] = v;
    }
    if (i < (unsigned)n_status) status[i] = 0ull;
}

__global__ void __launch_bounds__(256) vox_hash(const float *__restrict__ pts, int n, const int *__restrict__ n_dev, VoxParams p,
                                                VoxEntry *__restrict__ table, int *__restrict__ pslot, int *__restrict__ prank) {
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
    const int key = (c[2] * p.grid[1] + c[1]) * p.grid[0] + c[0];  // (z,y,x) row-major
    unsigned slot = (hash_key(key) >> 7) & p.mask;
    bool fresh;
    while (true) {
        const int prev = atomicCAS(&table[slot].key, -1, key);
        fresh = prev == -1;
        if (fresh || prev == key) break;
        slot = (slot + 1) & p.mask;
    }
    // (first, count) in one word: a slot this thread has just claimed still holds the initial value (unless a later point of the voxel
    // got in between: the CAS then returns what is there and the loop retries)
    unsigned long long *fc = &table[slot].fc;
    unsigned long long cur = fresh ? kFcInit : __hip_atomic_load(fc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (true) {
        const unsigned first = (unsigned)(cur >> 32);
        const unsigned long long nw = ((unsigned long long)(first < (unsigned)i ? first : (unsigned)i) << 32) | (unsigned)((unsigned)cur + 1u);
        const unsigned long long prev = atomicCAS(fc, cur, nw);
        if (prev == cur) break;
        cur = prev;
    }
    pslot[i] = (int)slot;
    prank[i] = (int)(unsigned)cur;  // points of this voxel that arrived before this one
}

// ---- single-pass scan over the points (decoupled look-back).  Item of point i: (1, count of its voxel if > 1 else 0) when i is the first
//      point of its voxel, else (0, 0).  A tile = 1024 points; status word of a tile = flag << 62 | voxels << 32 | bucket words, written and
//      read as ONE 8-byte agent-scope access (the value carries its own tag: no fence needed).
constexpr unsigned long long kFlagAgg = 1ull << 62, kFlagPre = 2ull << 62, kFlagMask = 3ull << 62;

__device__ inline void wave_sum2(int &a, int &b) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off);
        b += __shfl_xor(b, off);
    }
}

__global__ void __launch_bounds__(kScanThreads) vox_scan(const int *__restrict__ pslot, VoxEntry *__restrict__ table, int n, const int *__restrict__ n_dev,
                                                         int max_voxels, unsigned long long *__restrict__ status /*[tiles + 1]: [tiles] = tile counter*/,
                                                         int n_tiles, int *__restrict__ vkey, int *__restrict__ vfirst, int *__restrict__ vcnt,
                                                         int *__restrict__ voff, int *__restrict__ num_voxels, int *__restrict__ need_place) {
    __shared__ int sm[2 * (kScanThreads / 64) + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    n = fd::device_count(n, n_dev);
    if (tid == 0) sm[8] = (int)atomicAdd(reinterpret_cast<unsigned long long *>(status + n_tiles), 1ull);  // tiles are numbered in arrival order
    __syncthreads();
    const int tile = sm[8];
    const int base = tile * kScanTile + tid * kScanItems;
    int ff[kScanItems], cc[kScanItems], kk[kScanItems], ss[kScanItems];
    int f = 0, c = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        ff[k] = 0; cc[k] = 0; kk[k] = 0; ss[k] = -1;
        const int i = base + k;
        if (i < n) {
            const int s = pslot[i];
            if (s >= 0) {
                const uint4 e = *reinterpret_cast<const uint4 *>(table + s);  // key, aux, count, first
                if ((int)e.w == i) {
                    ff[k] = 1; cc[k] = (int)e.z; kk[k] = (int)e.x; ss[k] = s;
                }
            }
        }
        f += ff[k];
        c += cc[k] > 1 ? cc[k] : 0;
    }
    // exclusive scan inside the tile
    int iff = f, icc = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int uf = __shfl_up(iff, off), uc = __shfl_up(icc, off);
        if (lane >= off) { iff += uf; icc += uc; }
    }
    if (lane == 63) { sm[wave] = iff; sm[4 + wave] = icc; }
    __syncthreads();
    int wf = 0, wc = 0, tf = 0, tc = 0;
#pragma unroll
    for (int w = 0; w < kScanThreads / 64; ++w) {
        if (w < wave) { wf += sm[w]; wc += sm[4 + w]; }
        tf += sm[w]; tc += sm[4 + w];
    }
    // look-back by the first wave: the totals of all tiles in front of this one
    if (wave == 0) {
        int ef = 0, ec = 0;
        if (tile > 0) {
            if (lane == 0) __hip_atomic_store(status + tile, kFlagAgg | ((unsigned long long)(unsigned)tf << 32) | (unsigned)tc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int back = tile - 1;
            while (true) {
                const int idx = back - lane;
                unsigned long long st = kFlagPre;  // (tiles in front of the first: an empty prefix)
                if (idx >= 0) {
                    do {
                        st = __hip_atomic_load(status + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } while ((st & kFlagMask) == 0ull);
                }
                const unsigned long long pre = __ballot((st & kFlagMask) == kFlagPre);
                const int first_pre = pre ? __builtin_ctzll(pre) : 64;
                int vf = lane <= first_pre ? (int)((st >> 32) & 0x3fffffffull) : 0, vc = lane <= first_pre ? (int)(unsigned)st : 0;
                wave_sum2(vf, vc);
                ef += vf; ec += vc;
                if (pre) break;
                back -= 64;
            }
        }
        if (lane == 0) {
            __hip_atomic_store(status + tile, kFlagPre | ((unsigned long long)(unsigned)(ef + tf) << 32) | (unsigned)(ec + tc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sm[9] = ef; sm[10] = ec;
        }
    }
    __syncthreads();
    f = sm[9] + wf + iff - f;   // exclusive over all points in front of this thread's items
    c = sm[10] + wc + icc - c;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (ff[k]) {
            const bool many = cc[k] > 1;
            if (f < max_voxels) {
                table[ss[k]].aux = many ? c : -1;
                vkey[f] = kk[k];
                vfirst[f] = base + k;
                vcnt[f] = cc[k];
                voff[f] = c;
            } else {
                table[ss[k]].aux = -1;
            }
            f += 1;
            c += many ? cc[k] : 0;
        }
    }
    if (tile == n_tiles - 1 && tid == kScanThreads - 1) {  // (the last tile in arrival order is the last tile of the cloud: ids follow the counter)
        num_voxels[0] = f < max_voxels ? f : max_voxels;
        need_place[0] = c > 0;
    }
}

__global__ void __launch_bounds__(256) vox_place(const int *__restrict__ pslot, const int *__restrict__ prank, int n, const int *__restrict__ n_dev,
                                                 const VoxEntry *__restrict__ table, const int *__restrict__ need_place, int *__restrict__ bucket) {
    if (!need_place[0]) return;  // every voxel holds one point: vox_emit reads the first points
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    n = fd::device_count(n, n_dev);
    if (i >= n) return;
    const int s = pslot[i];
    if (s < 0) return;
    const int off = table[s].aux;
    if (off < 0) return;
    bucket[off + prank[i]] = i;
}

template <int MAXP>
__global__ void __launch_bounds__(256) vox_emit(const float *__restrict__ pts, VoxParams p, const int *__restrict__ num_voxels,
                                                const int *__restrict__ vkey, const int *__restrict__ vfirst, const int *__restrict__ vcnt,
                                                const int *__restrict__ voff, const int *__restrict__ bucket, int batch_idx,
                                                float *__restrict__ out_voxels, float *__restrict__ out_mean, int mean_stride, int *__restrict__ out_coors,
                                                int coor_cols, int *__restrict__ out_num) {
    __shared__ __attribute__((aligned(16))) float s_mean[256 * 16];
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const int nvox = num_voxels[0];
    const bool live = v < nvox;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool via_lds = out_mean && mean_stride <= 16 && (mean_stride & 3) == 0;
    float sum[kMaxNd];
#pragma unroll
    for (int d = 0; d < kMaxNd; ++d) sum[d] = 0.0f;
    int np = 0;
    if (live) {
        const int nv = vcnt[v];
        np = nv < p.max_points ? nv : p.max_points;
        if (nv == 1 && !out_voxels) {
            const float *q = pts + (int64_t)vfirst[v] * p.ndim;
#pragma unroll
            for (int d = 0; d < kMaxNd; ++d)
                if (d < p.ndim) sum[d] = __fadd_rn(0.0f, q[d]);
        } else {
            int sel[MAXP];
#pragma unroll
            for (int k = 0; k < MAXP; ++k) sel[k] = 0x7fffffff;
            if (nv == 1) {
                sel[0] = vfirst[v];
            } else {
                const int *b = bucket + voff[v];
                for (int t = 0; t < nv; ++t) {
                    int x = b[t];
#pragma unroll
                    for (int k = 0; k < MAXP; ++k) {  // keep sel[] ascending; x carries the displaced larger value
                        const int lo = min(sel[k], x), hi = max(sel[k], x);
                        sel[k] = lo; x = hi;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < MAXP; ++k) {
                if (k < p.max_points) {
                    const bool on = k < np;
                    const float *q = pts + (int64_t)(on ? sel[k] : 0) * p.ndim;
#pragma unroll
                    for (int d = 0; d < kMaxNd; ++d) {
                        if (d < p.ndim) {
                            const float val = on ? q[d] : 0.0f;
                            sum[d] = __fadd_rn(sum[d], val);
                            if (out_voxels) out_voxels[((int64_t)v * p.max_points + k) * p.ndim + d] = val;
                        }
                    }
                }
            }
        }
        const int key = vkey[v];
        const int x = key % p.grid[0];
        const int t = key / p.grid[0];
        const int y = t % p.grid[1];
        const int z = t / p.grid[1];
        if (coor_cols == 4) *reinterpret_cast<int4 *>(out_coors + (int64_t)v * 4) = make_int4(batch_idx, z, y, x);
        else { int *oc = out_coors + (int64_t)v * 3; oc[0] = z; oc[1] = y; oc[2] = x; }
        out_num[v] = np;
    }
    if (!out_mean) return;
    const float cntf = (float)np;
    if (via_lds) {
        // a wave's 64 rows of mean_stride floats are contiguous in the output: through LDS, then 16-byte stores in address order
        float *sw = s_mean + wave * 64 * 16;
        if (live) {
#pragma unroll
            for (int d = 0; d < 16; ++d)
                if (d < mean_stride) sw[lane * mean_stride + d] = (d < p.ndim && d < kMaxNd) ? __fdiv_rn(sum[d < kMaxNd ? d : 0], cntf) : 0.0f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int v0 = blockIdx.x * blockDim.x + wave * 64;           // first voxel of this wave
        const int rows = nvox - v0 < 64 ? nvox - v0 : 64;             // live rows of the wave (<= 0: none)
        const int n4 = rows > 0 ? rows * mean_stride / 4 : 0;
        float4 *dst = reinterpret_cast<float4 *>(out_mean + (int64_t)v0 * mean_stride);
        for (int i = lane; i < n4; i += 64) dst[i] = reinterpret_cast<const float4 *>(sw)[i];
    } else if (live) {
#pragma unroll
        for (int d = 0; d < kMaxNd; ++d)
            if (d < p.ndim) out_mean[(int64_t)v * mean_stride + d] = __fdiv_rn(sum[d], cntf);
        for (int d = p.ndim; d < mean_stride; ++d) out_mean[(int64_t)v * mean_stride + d] = 0.0f;
    }
}

struct VoxWs {
    size_t table, pslot, prank, status, vkey, vfirst, vcnt, voff, bucket, flags, total;
    unsigned slots;
    int tiles;
};

VoxWs vox_layout(int64_t n, int64_t max_voxels) {
    VoxWs w;
    unsigned slots = 1024;
    while ((int64_t)slots * 2 < 3 * n) slots <<= 1;  // >= 1.5 n: load factor <= 0.67 when every point is its own voxel
    w.slots = slots;
    w.tiles = (int)((n + kScanTile - 1) / kScanTile);
    if (w.tiles < 1) w.tiles = 1;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += fd::align_up(bytes, 256); return o; };
    w.table = take(sizeof(VoxEntry) * slots);
    w.pslot = take(sizeof(int) * (size_t)(n + 1));
    w.prank = take(sizeof(int) * (size_t)(n + 1));
    w.status = take(sizeof(unsigned long long) * (size_t)(w.tiles + 1));
    w.vkey = take(sizeof(int) * (size_t)(max_voxels + 1));
    w.vfirst = take(sizeof(int) * (size_t)(max_voxels + 1));
    w.vcnt = take(sizeof(int) * (size_t)(max_voxels + 1));
    w.voff = take(sizeof(int) * (size_t)(max_voxels + 1));
    w.bucket = take(sizeof(int) * (size_t)(n + 1));
    w.flags = take(sizeof(int) * 4);
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
    // vox_emit writes 4-column coordinate rows as one int4 and, when mean rows leave through LDS (mean_stride a multiple of 4, <= 16), the
    // mean rows as float4: a caller's sliced / offset buffer must keep the 16-byte alignment those stores assume
    FD_REQUIRE(coor_cols != 4 || ((uintptr_t)out_coors & 15) == 0, "fd_voxelize: out_coors must be 16-byte aligned when coor_cols == 4");
    FD_REQUIRE(!(out_mean && mean_stride <= 16 && (mean_stride & 3) == 0) || ((uintptr_t)out_mean & 15) == 0,
               "fd_voxelize: out_mean must be 16-byte aligned when mean_stride is a multiple of 4 (<= 16)");
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
    FD_REQUIRE(((uintptr_t)workspace & 15) == 0, "fd_voxelize: workspace must be 16-byte aligned");
    p.mask = w.slots - 1;
    char *ws = (char *)workspace;
    VoxEntry *table = (VoxEntry *)(ws + w.table);
    int *pslot = (int *)(ws + w.pslot), *prank = (int *)(ws + w.prank);
    unsigned long long *status = (unsigned long long *)(ws + w.status);
    int *vkey = (int *)(ws + w.vkey), *vfirst = (int *)(ws + w.vfirst), *vcnt = (int *)(ws + w.vcnt), *voff = (int *)(ws + w.voff);
    int *bucket = (int *)(ws + w.bucket), *flags = (int *)(ws + w.flags);
    const int n = (int)n_points;
    const int nb = (n + 255) / 256;
    hipLaunchKernelGGL(vox_init, dim3((w.slots + 255) / 256), dim3(256), 0, stream, table, w.slots, status, w.tiles + 1);
    hipLaunchKernelGGL(vox_hash, dim3(nb), dim3(256), 0, stream, points, n, n_points_dev, p, table, pslot, prank);
    hipLaunchKernelGGL(vox_scan, dim3(w.tiles), dim3(kScanThreads), 0, stream, pslot, table, n, n_points_dev, p.max_voxels, status, w.tiles, vkey, vfirst, vcnt,
                       voff, out_num_voxels, flags);
    hipLaunchKernelGGL(vox_place, dim3(nb), dim3(256), 0, stream, pslot, prank, n, n_points_dev, table, flags, bucket);
    const int64_t vmax = n_points < max_voxels ? n_points : max_voxels;
#define FD_EMIT(MP)                                                                                                                               \
    hipLaunchKernelGGL(vox_emit<MP>, dim3((unsigned)((vmax + 255) / 256)), dim3(256), 0, stream, points, p, out_num_voxels, vkey, vfirst, vcnt, voff, bucket, \
                       batch_idx, out_voxels, out_mean, mean_stride, out_coors, coor_cols, out_num_points)
    if (max_points <= 16) FD_EMIT(16);       // VoxelNet configs: 10
    else if (max_points <= 32) FD_EMIT(32);  // PointPillars configs: 20
    else FD_EMIT(64);                        // points_to_voxel's default of 35
#undef FD_EMIT
    return fd::check_launch("fd_voxelize");
}
