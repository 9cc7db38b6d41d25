// Shared helpers for the gfx950 kernels of libfuturedet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#include "../../include/futuredet_hip.h"

namespace fd {

void set_error(const char *fmt, ...);
// out_planes 1: the output rows are written as three bf16 planes (fd_spconv_split.hip's storage format; no residual then)
int spconv_f32_compact_dispatch(const float *in, const void *wp, const float *bias, const float *residual, int relu, const int *nbr,
                                int64_t nbr_stride, int K, int64_t n_in_bound, int n_out, const int *n_out_dev, int cin, int cout, void *out,
                                const int *ranges, int n_ranges, int out_planes, hipStream_t stream);

// fd_spconv_bf16w.hip: SubM convolutions with an LDS window of input rows (64 / 128 channels); 1 = launched, 0 = not its case
int spconv_bf16_win_dispatch(const void *in, const void *wp, const float *bias, const void *residual, int relu, const int *nbr, int64_t nbr_stride, int K,
                             int64_t n_in_bound, int n_out, const int *n_out_dev, int64_t n_expected, int cin, int cout, void *out, hipStream_t stream);
int spconv_bf16_ws_dispatch(const void *in, const void *wp, const float *bias, const void *residual, int relu, const int *nbr, int64_t nbr_stride, int K,
                            int64_t n_in_bound, int n_out, const int *n_out_dev, int64_t n_expected, int cin, int cout, void *out, hipStream_t stream);

// fd_spconv_split.hip: features stored as three bf16 planes per row, split-operand (3 x bf16) arithmetic; out_planes 1: planes out
// (residual planes), 0: float32 out (residual float32); 1 = launched, 0 = shape not covered
int spconv_p3_dispatch(const void *in, const void *wp, const float *bias, const void *residual, int relu, const int *nbr, int64_t nbr_stride, int K,
                       int64_t n_in_bound, int n_out, const int *n_out_dev, int64_t n_expected, int cin, int cout, int out_planes, void *out,
                       hipStream_t stream);
void split3_host(float w, uint16_t &h, uint16_t &m, uint16_t &l);

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
enum TuneKey { kTuneSpconvRG = 0, kTuneSpconvV1, kTuneSpconvBf16V1, kTuneV2Depth, kTuneV2TM, kTuneV2LdsPad, kTuneConvNT, kTuneV2RangesPerCU, kTuneV2Uniform, kTuneV2RowCost, kTuneSpconvC32, kTuneBf16GP, kTuneBf16RG, kTuneBf16Depth, kTuneBf16NW, kTuneSplitRG, kTuneStrict, kTuneBf16Win, kTuneF32ResRG, kTuneConvStrip, kTuneCount };
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

// ---- fp32 value <-> three bf16 pieces (x = h + m + l exactly; fd_spconv_split.hip) ---------------------------------
typedef float fd_f32x4 __attribute__((ext_vector_type(4)));
typedef float fd_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 fd_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 fd_bf16x2 __attribute__((ext_vector_type(2)));
// four floats -> the three bf16x4 pieces, round-to-nearest-even at every level (v_cvt_pk_bf16_f32)
__device__ __forceinline__ void split3x4(const fd_f32x4 &v, fd_bf16x4 &h, fd_bf16x4 &m, fd_bf16x4 &l) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const fd_f32x2 x = {v[2 * i], v[2 * i + 1]};
        const fd_bf16x2 hh = __builtin_convertvector(x, fd_bf16x2);
        const fd_f32x2 r1 = x - __builtin_convertvector(hh, fd_f32x2);
        const fd_bf16x2 mm = __builtin_convertvector(r1, fd_bf16x2);
        const fd_f32x2 r2 = r1 - __builtin_convertvector(mm, fd_f32x2);
        const fd_bf16x2 ll = __builtin_convertvector(r2, fd_bf16x2);
        h[2 * i] = hh[0]; h[2 * i + 1] = hh[1];
        m[2 * i] = mm[0]; m[2 * i + 1] = mm[1];
        l[2 * i] = ll[0]; l[2 * i + 1] = ll[1];
    }
}
__device__ __forceinline__ fd_f32x4 join3x4(const fd_bf16x4 &h, const fd_bf16x4 &m, const fd_bf16x4 &l) {
    return (__builtin_convertvector(h, fd_f32x4) + __builtin_convertvector(m, fd_f32x4)) + __builtin_convertvector(l, fd_f32x4);  // exact
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
