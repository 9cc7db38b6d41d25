// Shared helpers for the gfx950 kernels of libfuturedet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#include "../../include/futuredet_hip.h"

namespace fd {

void set_error(const char *fmt, ...);
int spconv_f32_compact_dispatch(const float *in, const void *wp, const float *bias, const float *residual, int relu, const int *nbr,
                                int64_t nbr_stride, int K, int64_t n_in_bound, int n_out, const int *n_out_dev, int cin, int cout, float *out,
                                const int *ranges, int n_ranges, hipStream_t stream);

// fd_spconv_bf16win.hip: 64 -> 64, 128 -> 128 with an LDS window of input rows and 32-row MFMA tiles; its weight layout
// follows the standard one in the packed buffer (spconv_bf16_win_weight_bytes > 0); 1 = launched, 0 = shape not covered
size_t spconv_bf16_win_weight_bytes(int K, int cin, int cout);
void spconv_bf16_win_pack(const float *w, int K, int cin, int cout, uint16_t (*tobf)(float), uint16_t *dst);
int spconv_bf16_win_dispatch(const void *in, const void *wp_win, const float *bias, const void *residual, int relu, const int *nbr, int64_t nbr_stride, int K,
                             int64_t n_in_bound, int n_out, const int *n_out_dev, int64_t n_expected, int cin, int cout, void *out, hipStream_t stream);
int spconv_bf16_ws_dispatch(const void *in, const void *wp, const float *bias, const void *residual, int relu, const int *nbr, int64_t nbr_stride, int K,
                            int64_t n_in_bound, int n_out, const int *n_out_dev, int64_t n_expected, int cin, int cout, void *out, hipStream_t stream);

// fd_conv2d_wino_pc.hip: 0 = launched, 1 = shape not supported by the variant, -1 = dynamic LDS refused
int wino_pc_launch(const float *x, const void *wp, const float *bias, float *y, int B, int H, int W, int cin, int cout, int relu, int cout_total,
                   int co_off, hipStream_t stream);

// fd_spconv_f32r.hip: fp32, 16 input channels (weights resident in LDS, register accumulators, empty items skipped); 1 = launched
int spconv_f32_res16_dispatch(const float *in, const void *wp, const float *bias, const float *residual, int relu, const int *nbr, int64_t nbr_stride, int K,
                              int64_t n_in_bound, int n_out, const int *n_out_dev, int64_t n_expected, int cin, int cout, float *out, hipStream_t stream);

int spconv_f32_c32_dispatch(const float *in, const void *wp, const float *bias, const float *residual, int relu, const int *nbr,
                            int64_t nbr_stride, int K, int64_t n_in_bound, int n_out, const int *n_out_dev, int cin, int cout, float *out,
                            const int *ranges, int n_ranges, hipStream_t stream);

// Element counts that only the device knows (voxels of a sweep, rows of a sparse level): entry points take the host-side
// CAPACITY plus an optional device pointer to the actual count; kernels are launched for the capacity and work on
// min(capacity, *count).  This is what lets a whole sweep be issued (or captured into one hipGraph) without reading a
// count back to the host.
__device__ __forceinline__ int device_count(int capacity, const int *count_dev) {
    if (!count_dev) return capacity;
    const int n = *count_dev;
    return n < capacity ? (n > 0 ? n : 0) : capacity;
}

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return FD_ELAUNCH;
    }
    return FD_OK;
}

#define FD_REQUIRE(cond, ...)           \
    do {                                \
        if (!(cond)) {                  \
            fd::set_error(__VA_ARGS__); \
            return FD_EINVAL;           \
        }                               \
    } while (0)

inline hipStream_t as_stream(fd_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Fills n_words 32-bit words (p 4-byte aligned) with a kernel (fd_error.hip).  Used instead of hipMemsetAsync wherever the
// call may be captured into a hipGraph: measured on ROCm 7.2 / MI355X, a captured graph whose memset nodes reset the
// voxelizer's hash table hangs on its second replay (the table is not reset, the probe loop never finds a free slot).
int fill_words(void *p, uint32_t value, size_t n_words, hipStream_t stream);
// the same for up to three regions in one launch (a null pointer = no region)
int fill_words3(void *p0, uint32_t v0, size_t n0, void *p1, uint32_t v1, size_t n1, void *p2, uint32_t v2, size_t n2, hipStream_t stream);

// ---- per-process state (fd_error.hip).  The library keeps NO per-call mutable state; what it caches is
//      keyed by device ordinal and published with atomics, so calls from several threads / for several
//      devices of one process are safe.
constexpr int kMaxDevices = 64;
int current_device();   // hipGetDevice (0 when the runtime cannot tell)
int device_cu_count();  // compute units of the current device, cached per device
// Kernels that need more than 64 KB of dynamic LDS must be told so once per (kernel, device).  `done` is the
// kernel instantiation's own bit mask of devices already configured.  Returns false when the runtime refuses.
bool ensure_dynamic_lds(const void *kernel, size_t bytes, std::atomic<uint64_t> &done);
// Tuning / test knobs (fd_tuning_set; initial values are read ONCE from the FD_* environment variables when the
// library is loaded).  0 = the built-in heuristic.
enum TuneKey { kTuneSpconvRG = 0, kTuneSpconvV1, kTuneSpconvBf16V1, kTuneV2Depth, kTuneV2TM, kTuneV2LdsPad, kTuneConvNT, kTuneV2RangesPerCU, kTuneV2Uniform, kTuneV2RowCost, kTuneSpconvC32, kTuneBf16GP, kTuneBf16RG, kTuneBf16Depth, kTuneBf16NW, kTuneStrict, kTuneBf16Win, kTuneF32ResRG, kTuneConvStrip, kTuneF32ResNW, kTuneCount };
int tuning(TuneKey key);

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- sparse index addressing (see include/futuredet_hip.h) ------------------------------------------
struct IndexGeom {
    int B, D, H, W, Ht, Wt;
    __host__ __device__ int64_t num_cols() const { return (int64_t)B * Ht * Wt * 64; }
};
__host__ __device__ inline IndexGeom make_geom(int B, int D, int H, int W) {
    IndexGeom g;
    g.B = B; g.D = D; g.H = H; g.W = W;
    g.Ht = (H + 7) >> 3; g.Wt = (W + 7) >> 3;
    return g;
}
__host__ __device__ inline int64_t col_of(const IndexGeom &g, int b, int y, int x) {
    return ((((int64_t)b * g.Ht + (y >> 3)) * g.Wt + (x >> 3)) << 6) + ((y & 7) << 3) + (x & 7);
}
__device__ inline void col_to_byx(const IndexGeom &g, int64_t col, int &b, int &y, int &x) {
    int in = (int)(col & 63);
    int64_t t = col >> 6;
    int tx = (int)(t % g.Wt);
    t /= g.Wt;
    int ty = (int)(t % g.Ht);
    b = (int)(t / g.Ht);
    y = ty * 8 + (in >> 3);
    x = tx * 8 + (in & 7);
}

// XCD-aware block remap (8 XCDs, round-robin dispatch): consecutive logical tiles land on one XCD so
// spatially adjacent tiles share an L2.  Bijective for any grid size.
__device__ inline unsigned xcd_swizzle(unsigned bid, unsigned nblocks) {
    const unsigned nx = 8;
    unsigned per = nblocks / nx, rem = nblocks % nx;
    unsigned xcd = bid % nx, loc = bid / nx;
    // XCD x owns per(+1 if x<rem) logical tiles
    unsigned start = xcd * per + (xcd < rem ? xcd : rem);
    return start + loc;
}

}  // namespace fd
