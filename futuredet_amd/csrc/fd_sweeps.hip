// Sweep assembly on the device: the step in front of the voxelizer.
// Replaces the NuScenes branch of LoadPointCloudFromFile.__call__ (det3d/datasets/pipelines/loading.py:107-141)
// with read_file's column cut (:31), remove_close (:36-45) and read_sweep (:47-60).
//
// One thread per raw row.  The kept rows must come out in input order (the voxelizer's first-come rule depends on
// it), so this is a stable stream compaction: pass 1 counts survivors per 256-row block, pass 2 scans the block
// counts in one workgroup, pass 3 recomputes the predicate, ranks survivors inside the block with wave ballots and
// writes them.  Rows [count, n_rows) of the output are filled with +inf so the voxelizer can be launched on the
// n_rows upper bound without a host round trip (an infinite coordinate fails its range test like any outlier).
// HBM-bound: 20 B read twice (the second read hits L2) + 20 B written per row.
#include "fd_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxSweeps = 64;

struct SweepLds {
    long long row_begin[kMaxSweeps + 1];
};

__device__ inline int find_sweep(const SweepLds &s, int n_sweeps, long long row) {
    int lo = 0, hi = n_sweeps - 1;  // last sweep with row_begin <= row
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (s.row_begin[mid] <= row) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ inline void load_sweeps(SweepLds &s, const fd_sweep_desc *__restrict__ d, int n_sweeps) {
    for (int t = threadIdx.x; t < n_sweeps; t += kBlock) s.row_begin[t] = d[t].row_begin;
    if (threadIdx.x == 0) s.row_begin[n_sweeps] = d[n_sweeps - 1].row_end;
    __syncthreads();
}

__device__ inline bool row_kept(const fd_sweep_desc &d, const float *__restrict__ q, float radius) {
    // remove_close (loading.py:41-44): strict comparisons in float32, both axes inside -> dropped
    return !((d.flags & FD_SWEEP_REMOVE_CLOSE) && fabsf(q[0]) < radius && fabsf(q[1]) < radius);
}

__global__ void __launch_bounds__(kBlock) sweep_count(const float *__restrict__ raw, int raw_cols, long long n_rows,
                                                      const fd_sweep_desc *__restrict__ sweeps, int n_sweeps, float radius,
                                                      int *__restrict__ block_cnt) {
    __shared__ SweepLds s;
    __shared__ int s_cnt[kBlock / 64];
    load_sweeps(s, sweeps, n_sweeps);
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    bool keep = false;
    if (i < n_rows && i < s.row_begin[n_sweeps]) keep = row_kept(sweeps[find_sweep(s, n_sweeps, i)], raw + i * raw_cols, radius);
    const unsigned long long m = __ballot(keep);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// exclusive scan of the block counts, one workgroup (n_blocks is a few thousand at most)
__global__ void __launch_bounds__(1024) sweep_scan(int *__restrict__ block_cnt, int n_blocks, int *__restrict__ out_count) {
    __shared__ int s_part[1024];
    const int per = (n_blocks + 1023) / 1024;
    const int b0 = threadIdx.x * per, b1 = min(b0 + per, n_blocks);
    int sum = 0;
    for (int b = b0; b < b1; ++b) sum += block_cnt[b];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = threadIdx.x >= off ? s_part[threadIdx.x - off] : 0;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = s_part[threadIdx.x] - sum;
    for (int b = b0; b < b1; ++b) {
        int c = block_cnt[b];
        block_cnt[b] = run;
        run += c;
    }
    if (threadIdx.x == 1023) *out_count = s_part[1023];
}

__global__ void __launch_bounds__(kBlock) sweep_write(const float *__restrict__ raw, int raw_cols, int keep_cols, long long n_rows,
                                                      const fd_sweep_desc *__restrict__ sweeps, int n_sweeps, float radius,
                                                      const int *__restrict__ block_base, const int *__restrict__ total,
                                                      float *__restrict__ out) {
    __shared__ SweepLds s;
    __shared__ int s_cnt[kBlock / 64];
    load_sweeps(s, sweeps, n_sweeps);
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    const int oc = keep_cols + 1;
    bool keep = false;
    int sw = 0;
    if (i < n_rows && i < s.row_begin[n_sweeps]) {  // n_rows is an upper bound (a fixed-capacity buffer): the descriptors say where the rows end
        sw = find_sweep(s, n_sweeps, i);
        keep = row_kept(sweeps[sw], raw + i * raw_cols, radius);
    }
    const unsigned long long m = __ballot(keep);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_cnt[wave] = __popcll(m);
    __syncthreads();
    int base = block_base[blockIdx.x];
    for (int w = 0; w < wave; ++w) base += s_cnt[w];
    if (keep) {
        const fd_sweep_desc &d = sweeps[sw];
        const float *q = raw + i * raw_cols;
        float *o = out + (long long)(base + __popcll(m & ((1ull << lane) - 1ull))) * oc;
        float v[8];
        for (int c = 0; c < keep_cols; ++c) v[c] = q[c];
        if (d.flags & FD_SWEEP_HAS_TRANSFORM) {
            // read_sweep (loading.py:53-57): float64 4x4 . [x y z 1]^T, rows 0..2 stored back as float32
            const double x = v[0], y = v[1], z = v[2];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                double acc = d.m[4 * r] * x;
                acc = fma(d.m[4 * r + 1], y, acc);
                acc = fma(d.m[4 * r + 2], z, acc);
                acc = fma(d.m[4 * r + 3], 1.0, acc);
                v[r] = (float)acc;
            }
        }
        for (int c = 0; c < keep_cols; ++c) o[c] = v[c];
        o[keep_cols] = d.time;  // float32(time_lag); 0 for the key frame
    }
    if (i < n_rows && i >= (long long)*total) {
        float *o = out + i * oc;
        for (int c = 0; c < oc; ++c) o[c] = __builtin_inff();
    }
}

}  // namespace

extern "C" size_t fd_sweep_assemble_workspace_bytes(int64_t n_rows) {
    if (n_rows < 0) return 0;
    return fd::align_up((size_t)((n_rows + kBlock - 1) / kBlock + 1) * sizeof(int), 256);
}

extern "C" int fd_sweep_assemble(const float *raw, int raw_cols, int keep_cols, int64_t n_rows, const fd_sweep_desc *sweeps_dev,
                                 int n_sweeps, float min_distance, float *out_points, int32_t *out_count, void *workspace,
                                 size_t workspace_bytes, fd_stream_t stream_) {
    FD_REQUIRE(out_count, "fd_sweep_assemble: null out_count");
    FD_REQUIRE(n_rows >= 0 && n_rows < (1ll << 30), "fd_sweep_assemble: n_rows out of range");
    FD_REQUIRE(keep_cols >= 3 && keep_cols <= 7 && raw_cols >= keep_cols, "fd_sweep_assemble: need 3 <= keep_cols <= 7 and raw_cols >= keep_cols");
    FD_REQUIRE(n_sweeps >= 0 && n_sweeps <= kMaxSweeps, "fd_sweep_assemble: n_sweeps must be in [0,%d]", kMaxSweeps);
    hipStream_t stream = fd::as_stream(stream_);
    if (n_rows == 0 || n_sweeps == 0) {
        FD_REQUIRE(n_rows == 0, "fd_sweep_assemble: rows without a sweep descriptor");
        fd::fill_words(out_count, 0u, 1, stream);
        return FD_OK;
    }
    FD_REQUIRE(raw && sweeps_dev && out_points, "fd_sweep_assemble: null argument");
    const size_t need = fd_sweep_assemble_workspace_bytes(n_rows);
    if (!workspace || workspace_bytes < need) {
        fd::set_error("fd_sweep_assemble: workspace %zu < required %zu", workspace_bytes, need);
        return FD_EWORKSPACE;
    }
    int *block_cnt = static_cast<int *>(workspace);
    const int n_blocks = (int)((n_rows + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(sweep_count, dim3(n_blocks), dim3(kBlock), 0, stream, raw, raw_cols, (long long)n_rows, sweeps_dev, n_sweeps,
                       min_distance, block_cnt);
    hipLaunchKernelGGL(sweep_scan, dim3(1), dim3(1024), 0, stream, block_cnt, n_blocks, out_count);
    hipLaunchKernelGGL(sweep_write, dim3(n_blocks), dim3(kBlock), 0, stream, raw, raw_cols, keep_cols, (long long)n_rows, sweeps_dev,
                       n_sweeps, min_distance, block_cnt, out_count, out_points);
    return fd::check_launch("fd_sweep_assemble");
}
