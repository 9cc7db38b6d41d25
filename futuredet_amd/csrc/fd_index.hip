// Sparse index (column occupancy words + prefix counts) and output-stationary rulebook for gfx950.
//
// Replaces spconv 1.0's indice-pair generation (called implicitly from det3d/models/backbones/scn.py:99-141).
// Instead of a hash table + pair lists, an active set is a bitmap: one 64-bit word per (b,y,x) column, bit z
// set when the voxel is active.  An exclusive scan of the popcounts gives every active voxel a row number,
// so rows come out spatially sorted (8x8 column tiles, z fastest) with no sort and no hash probing;
// lookups are two loads + a popcount; the downsampled set of a strided conv is an OR of shifted bits.
#include "fd_common.h"

namespace {

using fd::IndexGeom;

constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;

__global__ void __launch_bounds__(256) idx_mark(const int *__restrict__ coords, const int *__restrict__ n_dev, int64_t n_max,
                                                IndexGeom g, unsigned long long *__restrict__ words) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t n = n_dev ? (int64_t)n_dev[0] : n_max;
    if (n > n_max) n = n_max;
    if (i >= n) return;
    const int4 c = reinterpret_cast<const int4 *>(coords)[i];  // (b,z,y,x)
    if (c.x < 0 || c.x >= g.B || c.y < 0 || c.y >= g.D || c.z < 0 || c.z >= g.H || c.w < 0 || c.w >= g.W) return;
    atomicOr(&words[fd::col_of(g, c.x, c.z, c.w)], 1ull << c.y);
}

struct DownParams {
    int k[3], s[3], p[3];
};

// one thread per OUTPUT column: out word = z-transform(OR of the k x k input columns of its window).  (The scatter form this
// replaces -- one thread per input column, an atomicOr per output column it feeds -- took 23 + 18 + 8 + 5 us for the four
// levels of the backbone: device-scope atomics are served memory-side on this part.  No atomics here, every output column is
// written exactly once, so the output needs no zero fill.)
__global__ void __launch_bounds__(256) idx_down(const unsigned long long *__restrict__ in_words, IndexGeom gi, IndexGeom go,
                                                DownParams dp, unsigned long long *__restrict__ out_words) {
    int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= go.num_cols()) return;
    int b, oy, ox;
    fd::col_to_byx(go, col, b, oy, ox);
    unsigned long long acc = 0;
    if (oy < go.H && ox < go.W) {  // (columns of the 8 x 8 tile padding stay empty)
        for (int ky = 0; ky < dp.k[1]; ++ky) {
            const int iy = oy * dp.s[1] - dp.p[1] + ky;
            if (iy < 0 || iy >= gi.H) continue;
            for (int kx = 0; kx < dp.k[2]; ++kx) {
                const int ix = ox * dp.s[2] - dp.p[2] + kx;
                if (ix < 0 || ix >= gi.W) continue;
                acc |= in_words[fd::col_of(gi, b, iy, ix)];
            }
        }
    }
    // z axis: out bit oz set iff some active z = oz * s - p + kz, kz < k
    unsigned long long zo = 0;
    if (acc) {
        const unsigned long long taps = (1ull << dp.k[0]) - 1ull;
        for (int oz = 0; oz < go.D; ++oz) {
            const int lo = oz * dp.s[0] - dp.p[0];
            const unsigned long long m = lo >= 0 ? (lo < 64 ? taps << lo : 0ull) : taps >> (-lo);
            if (acc & m) zo |= 1ull << oz;
        }
    }
    out_words[col] = zo;
}

__device__ inline int block_excl_scan(int v, int &total, int *sm) {
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
#pragma unroll
    for (int w = 0; w < kScanThreads / 64; ++w) {
        if (w < wave) wo += sm[w];
        tot += sm[w];
    }
    __syncthreads();
    total = tot;
    return wo + iv - v;
}

__global__ void __launch_bounds__(kScanThreads) idx_scan1(const unsigned long long *__restrict__ words, int64_t ncols,
                                                          int *__restrict__ bsum) {
    __shared__ int sm[4];
    int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
        if (base + k < ncols) s += __popcll(words[base + k]);
    int total;
    block_excl_scan(s, total, sm);
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

__global__ void __launch_bounds__(kScanThreads) idx_scan2(int *__restrict__ bsum, int nblocks, int *__restrict__ n_active) {
    __shared__ int sm[4];
    int run = 0;
    for (int base = 0; base < nblocks; base += kScanThreads) {
        int j = base + threadIdx.x;
        int v = j < nblocks ? bsum[j] : 0;
        int total;
        int e = block_excl_scan(v, total, sm);
        if (j < nblocks) bsum[j] = run + e;
        run += total;
    }
    if (threadIdx.x == 0) n_active[0] = run;
}

__global__ void __launch_bounds__(kScanThreads) idx_scan3(const unsigned long long *__restrict__ words, int64_t ncols,
                                                          const int *__restrict__ bsum, int *__restrict__ prefix) {
    __shared__ int sm[4];
    int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int pc[kScanItems];
    int s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        pc[k] = (base + k < ncols) ? __popcll(words[base + k]) : 0;
        s += pc[k];
    }
    int total;
    int e = block_excl_scan(s, total, sm) + bsum[blockIdx.x];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (base + k < ncols) prefix[base + k] = e;
        e += pc[k];
    }
}

__global__ void __launch_bounds__(256) idx_coords(const unsigned long long *__restrict__ words, const int *__restrict__ prefix,
                                                  IndexGeom g, int *__restrict__ coords) {
    int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= g.num_cols()) return;
    unsigned long long w = words[col];
    if (!w) return;
    int b, y, x;
    fd::col_to_byx(g, col, b, y, x);
    int row = prefix[col];
    while (w) {
        int z = __builtin_ctzll(w);
        w &= w - 1;
        reinterpret_cast<int4 *>(coords)[row++] = make_int4(b, z, y, x);
    }
}

// ---- all levels of a pyramid in one set of launches (fd_index_pyramid): the words of every level exist before any prefix is
//      needed, so the three scan phases and the coordinate pass run once over the concatenated blocks of all levels
//      (5 levels: 15 + 5 launches of ~5 us each become 2 + 1)
constexpr int kMaxLevels = 8;
struct PyramidLevels {
    int n;
    int blk0[kMaxLevels + 1];   // first scan block of level l in the fused grid
    int cblk0[kMaxLevels + 1];  // first coordinate block (256 columns each)
    const unsigned long long *words[kMaxLevels];
    int *prefix[kMaxLevels];
    int *coords[kMaxLevels];
    long long coords_rows[kMaxLevels];  // rows of coords[l] (0 = unbounded)
    IndexGeom geom[kMaxLevels];
};

__device__ __forceinline__ int level_of(const int *first, int n, int blk) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < kMaxLevels; ++i)
        if (i < n && blk >= first[i]) l = i;
    return l;
}

__global__ void __launch_bounds__(kScanThreads) idx_scan1_ml(PyramidLevels L, int *__restrict__ bsum) {
    __shared__ int sm[4];
    const int l = level_of(L.blk0, L.n, blockIdx.x);
    const unsigned long long *words = L.words[l];
    const int64_t ncols = L.geom[l].num_cols();
    int64_t base = (int64_t)(blockIdx.x - L.blk0[l]) * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
        if (base + k < ncols) s += __popcll(words[base + k]);
    int total;
    block_excl_scan(s, total, sm);
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

// COORDS: the levels' coordinate tables exist already (capacity-sized levels of the sync-free step): the thread that computes a
// column's prefix writes its rows' coordinates too -- no second pass over words + prefix, one launch less
// (No middle pass over the block sums: a block sums the totals of its level's blocks in front of it itself -- at most a few
// thousand values -- and the last block of a level writes the level's count.)
template <bool COORDS>
__global__ void __launch_bounds__(kScanThreads) idx_scan3_ml(PyramidLevels L, const int *__restrict__ bsum, int *__restrict__ counts) {
    __shared__ int sm[4];
    const int l = level_of(L.blk0, L.n, blockIdx.x);
    const unsigned long long *words = L.words[l];
    int *prefix = L.prefix[l];
    const int64_t ncols = L.geom[l].num_cols();
    int64_t base = (int64_t)(blockIdx.x - L.blk0[l]) * kScanTile + (int64_t)threadIdx.x * kScanItems;
    unsigned long long wv[kScanItems];
    int s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        wv[k] = (base + k < ncols) ? words[base + k] : 0ull;
        s += __popcll(wv[k]);
    }
    int total;
    int e = block_excl_scan(s, total, sm);
    int before = 0;
    for (int j = L.blk0[l] + (int)threadIdx.x; j < (int)blockIdx.x; j += kScanThreads) before += bsum[j];
    int all_before;
    block_excl_scan(before, all_before, sm);
    e += all_before;
    if ((int)blockIdx.x == L.blk0[l + 1] - 1 && threadIdx.x == 0) counts[l] = all_before + total;
    int *coords = COORDS ? L.coords[l] : nullptr;
    const long long cap = (COORDS && L.coords_rows[l] > 0) ? L.coords_rows[l] : 0x7fffffffll;
    int b = 0, y0 = 0, x0 = 0;
    if (COORDS && s) fd::col_to_byx(L.geom[l], base, b, y0, x0);  // (kScanItems = 4 consecutive columns: one row of an 8 x 8 tile)
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (base + k < ncols) prefix[base + k] = e;
        if (COORDS && coords) {
            unsigned long long w = wv[k];
            int row = e;
            while (w) {
                const int z = __builtin_ctzll(w);
                w &= w - 1;
                if (row < cap) reinterpret_cast<int4 *>(coords)[row] = make_int4(b, z, y0, x0 + k);
                ++row;
            }
        }
        e += __popcll(wv[k]);
    }
}

__global__ void __launch_bounds__(256) idx_coords_ml(PyramidLevels L) {
    const int l = level_of(L.cblk0, L.n, blockIdx.x);
    int *coords = L.coords[l];
    if (!coords) return;
    const IndexGeom g = L.geom[l];
    int64_t col = (int64_t)(blockIdx.x - L.cblk0[l]) * blockDim.x + threadIdx.x;
    if (col >= g.num_cols()) return;
    unsigned long long w = L.words[l][col];
    if (!w) return;
    int b, y, x;
    fd::col_to_byx(g, col, b, y, x);
    int row = L.prefix[l][col];
    const long long cap = L.coords_rows[l] > 0 ? L.coords_rows[l] : 0x7fffffffll;
    while (w) {
        int z = __builtin_ctzll(w);
        w &= w - 1;
        if (row < cap) reinterpret_cast<int4 *>(coords)[row] = make_int4(b, z, y, x);  // (an overflowing capacity-sized level stays inside its table)
        ++row;
    }
}

bool fill_levels(PyramidLevels &L, int B, int n_levels, const fd_index_level *levels) {
    L.n = n_levels;
    int64_t blk = 0, cblk = 0;
    for (int l = 0; l < n_levels; ++l) {
        const fd_index_level &lv = levels[l];
        L.geom[l] = fd::make_geom(B, lv.D, lv.H, lv.W);
        L.words[l] = (const unsigned long long *)lv.words;
        L.prefix[l] = lv.prefix;
        L.coords[l] = lv.coords;
        L.coords_rows[l] = lv.coords_rows;
        L.blk0[l] = (int)blk;
        L.cblk0[l] = (int)cblk;
        const int64_t nc = L.geom[l].num_cols();
        blk += (nc + kScanTile - 1) / kScanTile;
        cblk += (nc + 255) / 256;
    }
    for (int l = n_levels; l <= kMaxLevels; ++l) {
        L.blk0[l] = (int)blk;
        L.cblk0[l] = (int)cblk;
    }
    return blk < (1ll << 31) && cblk < (1ll << 31);
}

__device__ inline int lookup_row(const unsigned long long *words, const int *prefix, const IndexGeom &g, int b, int z, int y,
                                 int x) {
    if (z < 0 || z >= g.D || y < 0 || y >= g.H || x < 0 || x >= g.W) return -1;
    int64_t col = fd::col_of(g, b, y, x);
    unsigned long long w = words[col];
    if (!((w >> z) & 1ull)) return -1;
    return prefix[col] + __popcll(w & ((1ull << z) - 1ull));
}

__global__ void __launch_bounds__(256) idx_lookup(const unsigned long long *__restrict__ words, const int *__restrict__ prefix,
                                                  IndexGeom g, const int *__restrict__ coords, const int *__restrict__ n_dev,
                                                  int64_t n_max, int *__restrict__ row_of) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t n = n_dev ? (int64_t)n_dev[0] : n_max;
    if (n > n_max) n = n_max;
    if (i >= n) return;
    const int4 c = reinterpret_cast<const int4 *>(coords)[i];
    int r = -1;
    if (c.x >= 0 && c.x < g.B) r = lookup_row(words, prefix, g, c.x, c.y, c.z, c.w);
    row_of[i] = r;
}

template <bool BF16>
__global__ void __launch_bounds__(256) rows_permute(const float *__restrict__ src, int c_src, const int *__restrict__ row_of,
                                                    const int *__restrict__ n_dev, int64_t n_max, void *__restrict__ dst,
                                                    int c_dst) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t n = n_dev ? (int64_t)n_dev[0] : n_max;
    if (n > n_max) n = n_max;
    int64_t i = t / c_dst;
    int ch = (int)(t % c_dst);
    if (i >= n) return;
    int r = row_of[i];
    if (r < 0) return;
    float v = ch < c_src ? src[i * c_src + ch] : 0.0f;
    if (BF16) {
        unsigned u = __float_as_uint(v);
        u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even
        reinterpret_cast<unsigned short *>(dst)[(int64_t)r * c_dst + ch] = (unsigned short)(u >> 16);
    } else {
        reinterpret_cast<float *>(dst)[(int64_t)r * c_dst + ch] = v;
    }
}

// lookup + permute in one pass: thread (voxel i, channel): row of the voxel in the index, then dst[row][ch] = src[i][ch]
template <bool BF16>
__global__ void __launch_bounds__(256) rows_place(const unsigned long long *__restrict__ words, const int *__restrict__ prefix, IndexGeom g,
                                                  const int *__restrict__ coords, const int *__restrict__ n_dev, int64_t n_max,
                                                  const float *__restrict__ src, int c_src, void *__restrict__ dst, int c_dst) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t n = n_dev ? (int64_t)n_dev[0] : n_max;
    if (n > n_max) n = n_max;
    int64_t i = t / c_dst;
    int ch = (int)(t % c_dst);
    if (i >= n) return;
    const int4 c = reinterpret_cast<const int4 *>(coords)[i];
    if (c.x < 0 || c.x >= g.B) return;
    const int r = lookup_row(words, prefix, g, c.x, c.y, c.z, c.w);
    if (r < 0) return;
    float v = ch < c_src ? src[i * c_src + ch] : 0.0f;
    if (BF16) {
        unsigned u = __float_as_uint(v);
        u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even
        reinterpret_cast<unsigned short *>(dst)[(int64_t)r * c_dst + ch] = (unsigned short)(u >> 16);
    } else {
        reinterpret_cast<float *>(dst)[(int64_t)r * c_dst + ch] = v;
    }
}

// the same with four channels per thread (c_dst % 4 == 0: every caller of the hot path -- 16 padded channels): a quarter of the index
// lookups, 16-byte (float32) / 8-byte (bf16) stores
template <bool BF16>
__global__ void __launch_bounds__(256) rows_place4(const unsigned long long *__restrict__ words, const int *__restrict__ prefix, IndexGeom g,
                                                   const int *__restrict__ coords, const int *__restrict__ n_dev, int64_t n_max,
                                                   const float *__restrict__ src, int c_src, void *__restrict__ dst, int c_dst) {
    const int q4 = c_dst >> 2;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t n = n_dev ? (int64_t)n_dev[0] : n_max;
    if (n > n_max) n = n_max;
    int64_t i = t / q4;
    const int ch = (int)(t - i * q4) * 4;
    if (i >= n) return;
    const int4 c = reinterpret_cast<const int4 *>(coords)[i];
    if (c.x < 0 || c.x >= g.B) return;
    const int r = lookup_row(words, prefix, g, c.x, c.y, c.z, c.w);
    if (r < 0) return;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = ch + k < c_src ? src[i * c_src + ch + k] : 0.0f;
    if (BF16) {
        unsigned short h[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned u = __float_as_uint(v[k]);
            u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even
            h[k] = (unsigned short)(u >> 16);
        }
        *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned short *>(dst) + (int64_t)r * c_dst + ch) =
            make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
    } else {
        *reinterpret_cast<float4 *>(reinterpret_cast<float *>(dst) + (int64_t)r * c_dst + ch) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// One thread per (FOUR consecutive output rows, ky): it requests the words and prefixes of the (up to three) kx columns of its rows
// up front and writes the kz taps of each column as 16-byte stores: 37 vector-memory instructions per four rows and ky where the
// round-1 form (one row and one (ky, kx) per thread, 4-byte stores) issued 72.  Measured neutral (8 launches of a 2-cloud pass: 161 ->
// 150-160 us): the launches write 27 x rows x 4 bytes at 2.0-2.4 TB/s, so what is left is the size of the table, not its construction.
__global__ void __launch_bounds__(256) rulebook_kernel(const unsigned long long *__restrict__ in_words, const int *__restrict__ in_prefix,
                                                       IndexGeom gi, const int *__restrict__ out_coords, const int *__restrict__ n_out_dev,
                                                       int64_t nbr_stride, int64_t out_rows, int fill_tail, int vec, DownParams dp, int *__restrict__ nbr) {
    // rows = the device's count clamped to what the coordinate table HOLDS (out_rows), not to the padded row stride of nbr: an
    // overflowing sweep of a capacity-sized level (count > capacity) must not read coordinates past the level's slice
    const int n_out = fd::device_count((int)(out_rows < 0x7fffffff ? out_rows : 0x7fffffff), n_out_dev);
    // rows written: all of the table's row capacity (tail = -1), or only the device's count when the consumers clamp to it
    // themselves (capacity-sized tables of the sync-free step: the launch must not cost what the capacity suggests)
    const int64_t n_rows = fill_tail ? nbr_stride : (int64_t)n_out;
    const int ky = blockIdx.y;  // rows grid-stride in x, four per thread (vec: every tap's row run starts on 16 bytes)
    for (int64_t o4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; o4 < n_rows; o4 += (int64_t)gridDim.x * blockDim.x * 4) {
        int4 c[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t o = o4 + j < n_out ? o4 + j : (n_out > 0 ? n_out - 1 : 0);  // (clamped address; rows >= n_out are written as -1)
            c[j] = n_out > 0 ? reinterpret_cast<const int4 *>(out_coords)[o] : make_int4(0, 0, 0, 0);  // (b,z,y,x)
        }
        // every word and prefix of the (up to three) kx columns is requested before any is used: two dependent loads per thread, not six
        unsigned long long w[3][4];
        int base[3][4];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iy = c[j].z * dp.s[1] - dp.p[1] + ky;
                const int ix = c[j].w * dp.s[2] - dp.p[2] + kx;
                const bool ok = kx < dp.k[2] && o4 + j < n_out && iy >= 0 && iy < gi.H && ix >= 0 && ix < gi.W;
                const int64_t col = ok ? fd::col_of(gi, c[j].x, iy, ix) : 0;  // (unconditional loads at a clamped address)
                const unsigned long long wv = in_words[col];
                base[kx][j] = in_prefix[col];
                w[kx][j] = ok ? wv : 0ull;
            }
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            if (kx >= dp.k[2]) break;
            for (int kz = 0; kz < dp.k[0]; ++kz) {
                int r[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int iz = c[j].y * dp.s[0] - dp.p[0] + kz;
                    const unsigned long long wv = w[kx][j];
                    r[j] = (iz >= 0 && iz < gi.D && ((wv >> iz) & 1ull)) ? base[kx][j] + __popcll(wv & ((1ull << iz) - 1ull)) : -1;
                }
                int *dst = nbr + ((int64_t)(kz * dp.k[1] + ky) * dp.k[2] + kx) * nbr_stride + o4;
                if (vec && o4 + 3 < n_rows) *reinterpret_cast<int4 *>(dst) = make_int4(r[0], r[1], r[2], r[3]);
                else
                    for (int j = 0; j < 4 && o4 + j < n_rows; ++j) dst[j] = r[j];
            }
        }
    }
}

int fill_dp(DownParams &dp, const int *k, const int *s, const int *p) {
    for (int i = 0; i < 3; ++i) {
        dp.k[i] = k[i]; dp.s[i] = s[i]; dp.p[i] = p[i];
        if (k[i] < 1 || k[i] > 3 || s[i] < 1 || s[i] > 4 || p[i] < 0 || p[i] > 2) return -1;
    }
    return 0;
}

}  // namespace

extern "C" int64_t fd_index_num_cols(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return fd::make_geom(B, 1, H, W).num_cols();
}

extern "C" size_t fd_index_workspace_bytes(int64_t num_cols) {
    // block sums of one scan; fd_index_pyramid scans all its levels with one set of launches and checks the sum itself
    return fd::align_up(sizeof(int) * (size_t)((num_cols + kScanTile - 1) / kScanTile + 1), 256);
}

extern "C" int fd_index_mark(const int32_t *coords, const int32_t *n_dev, int64_t n_max, int B, int D, int H, int W,
                             uint64_t *words, fd_stream_t stream) {
    FD_REQUIRE(coords && words, "fd_index_mark: null argument");
    FD_REQUIRE(B > 0 && D > 0 && D <= 64 && H > 0 && W > 0, "fd_index_mark: bad grid (D must be <= 64)");
    if (n_max <= 0) return FD_OK;
    IndexGeom g = fd::make_geom(B, D, H, W);
    hipLaunchKernelGGL(idx_mark, dim3((unsigned)((n_max + 255) / 256)), dim3(256), 0, fd::as_stream(stream), coords, n_dev, n_max, g,
                       (unsigned long long *)words);
    return fd::check_launch("fd_index_mark");
}

extern "C" int fd_index_downsample(const uint64_t *in_words, int B, int D, int H, int W, const int *ksize3, const int *stride3,
                                   const int *pad3, uint64_t *out_words, fd_stream_t stream) {
    FD_REQUIRE(in_words && out_words && ksize3 && stride3 && pad3, "fd_index_downsample: null argument");
    DownParams dp;
    FD_REQUIRE(fill_dp(dp, ksize3, stride3, pad3) == 0, "fd_index_downsample: unsupported kernel/stride/pad");
    IndexGeom gi = fd::make_geom(B, D, H, W);
    int od[3];
    const int in[3] = {D, H, W};
    for (int i = 0; i < 3; ++i) od[i] = (in[i] + 2 * dp.p[i] - (dp.k[i] - 1) - 1) / dp.s[i] + 1;
    FD_REQUIRE(od[0] > 0 && od[0] <= 64 && od[1] > 0 && od[2] > 0, "fd_index_downsample: bad output grid");
    IndexGeom go = fd::make_geom(B, od[0], od[1], od[2]);
    int64_t ncols = go.num_cols();
    hipLaunchKernelGGL(idx_down, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, fd::as_stream(stream),
                       (const unsigned long long *)in_words, gi, go, dp, (unsigned long long *)out_words);
    return fd::check_launch("fd_index_downsample");
}

extern "C" int fd_index_scan(const uint64_t *words, int64_t num_cols, int32_t *prefix, int32_t *n_active_dev, void *workspace,
                             size_t workspace_bytes, fd_stream_t stream_) {
    FD_REQUIRE(words && prefix && n_active_dev && workspace, "fd_index_scan: null argument");
    FD_REQUIRE(num_cols > 0, "fd_index_scan: empty index");
    if (workspace_bytes < fd_index_workspace_bytes(num_cols)) {
        fd::set_error("fd_index_scan: workspace too small");
        return FD_EWORKSPACE;
    }
    hipStream_t stream = fd::as_stream(stream_);
    int nb = (int)((num_cols + kScanTile - 1) / kScanTile);
    int *bsum = (int *)workspace;
    hipLaunchKernelGGL(idx_scan1, dim3(nb), dim3(kScanThreads), 0, stream, (const unsigned long long *)words, num_cols, bsum);
    hipLaunchKernelGGL(idx_scan2, dim3(1), dim3(kScanThreads), 0, stream, bsum, nb, n_active_dev);
    hipLaunchKernelGGL(idx_scan3, dim3(nb), dim3(kScanThreads), 0, stream, (const unsigned long long *)words, num_cols, bsum, prefix);
    return fd::check_launch("fd_index_scan");
}

extern "C" int fd_index_coords(const uint64_t *words, const int32_t *prefix, int B, int D, int H, int W, int32_t *coords,
                               fd_stream_t stream) {
    FD_REQUIRE(words && prefix && coords, "fd_index_coords: null argument");
    IndexGeom g = fd::make_geom(B, D, H, W);
    int64_t ncols = g.num_cols();
    hipLaunchKernelGGL(idx_coords, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, fd::as_stream(stream),
                       (const unsigned long long *)words, prefix, g, coords);
    return fd::check_launch("fd_index_coords");
}

extern "C" int fd_index_lookup(const uint64_t *words, const int32_t *prefix, int B, int D, int H, int W, const int32_t *coords_in,
                               const int32_t *n_dev, int64_t n_max, int32_t *row_of, fd_stream_t stream) {
    FD_REQUIRE(words && prefix && coords_in && row_of, "fd_index_lookup: null argument");
    if (n_max <= 0) return FD_OK;
    IndexGeom g = fd::make_geom(B, D, H, W);
    hipLaunchKernelGGL(idx_lookup, dim3((unsigned)((n_max + 255) / 256)), dim3(256), 0, fd::as_stream(stream),
                       (const unsigned long long *)words, prefix, g, coords_in, n_dev, n_max, row_of);
    return fd::check_launch("fd_index_lookup");
}

extern "C" int fd_rows_permute(const float *src, int c_src, const int32_t *row_of, const int32_t *n_dev, int64_t n_max, void *dst,
                               int c_dst, int dst_bf16, fd_stream_t stream) {
    FD_REQUIRE(src && row_of && dst, "fd_rows_permute: null argument");
    FD_REQUIRE(c_src > 0 && c_dst >= c_src, "fd_rows_permute: c_dst < c_src");
    if (n_max <= 0) return FD_OK;
    int64_t total = n_max * c_dst;
    dim3 grid((unsigned)((total + 255) / 256));
    if (dst_bf16)
        hipLaunchKernelGGL(rows_permute<true>, grid, dim3(256), 0, fd::as_stream(stream), src, c_src, row_of, n_dev, n_max, dst, c_dst);
    else
        hipLaunchKernelGGL(rows_permute<false>, grid, dim3(256), 0, fd::as_stream(stream), src, c_src, row_of, n_dev, n_max, dst, c_dst);
    return fd::check_launch("fd_rows_permute");
}

extern "C" int fd_rows_place(const uint64_t *words, const int32_t *prefix, int B, int D, int H, int W, const int32_t *coords_in,
                             const int32_t *n_dev, int64_t n_max, const float *src, int c_src, void *dst, int c_dst, int dst_bf16,
                             fd_stream_t stream) {
    FD_REQUIRE(words && prefix && coords_in && src && dst, "fd_rows_place: null argument");
    FD_REQUIRE(c_src > 0 && c_dst >= c_src, "fd_rows_place: c_dst < c_src");
    if (n_max <= 0) return FD_OK;
    IndexGeom g = fd::make_geom(B, D, H, W);
    if (c_dst % 4 == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        dim3 grid4((unsigned)((n_max * (c_dst / 4) + 255) / 256));
        if (dst_bf16)
            hipLaunchKernelGGL(rows_place4<true>, grid4, dim3(256), 0, fd::as_stream(stream), (const unsigned long long *)words, prefix, g, coords_in,
                               n_dev, n_max, src, c_src, dst, c_dst);
        else
            hipLaunchKernelGGL(rows_place4<false>, grid4, dim3(256), 0, fd::as_stream(stream), (const unsigned long long *)words, prefix, g, coords_in,
                               n_dev, n_max, src, c_src, dst, c_dst);
        return fd::check_launch("fd_rows_place");
    }
    dim3 grid((unsigned)((n_max * c_dst + 255) / 256));
    if (dst_bf16)
        hipLaunchKernelGGL(rows_place<true>, grid, dim3(256), 0, fd::as_stream(stream), (const unsigned long long *)words, prefix, g, coords_in,
                           n_dev, n_max, src, c_src, dst, c_dst);
    else
        hipLaunchKernelGGL(rows_place<false>, grid, dim3(256), 0, fd::as_stream(stream), (const unsigned long long *)words, prefix, g, coords_in,
                           n_dev, n_max, src, c_src, dst, c_dst);
    return fd::check_launch("fd_rows_place");
}

extern "C" int fd_rulebook(const uint64_t *in_words, const int32_t *in_prefix, int B, int Din, int Hin, int Win,
                           const int32_t *out_coords, int64_t out_rows, const int32_t *n_out_dev, int64_t nbr_stride, int fill_tail, const int *ksize3,
                           const int *stride3, const int *pad3, int32_t *nbr, fd_stream_t stream) {
    FD_REQUIRE(in_words && in_prefix && out_coords && n_out_dev && nbr && ksize3 && stride3 && pad3, "fd_rulebook: null argument");
    FD_REQUIRE(out_rows >= 0 && out_rows <= nbr_stride, "fd_rulebook: out_rows (rows of out_coords) must lie in [0, nbr_stride]");
    DownParams dp;
    FD_REQUIRE(fill_dp(dp, ksize3, stride3, pad3) == 0, "fd_rulebook: unsupported kernel/stride/pad");
    if (nbr_stride <= 0) return FD_OK;
    IndexGeom gi = fd::make_geom(B, Din, Hin, Win);
    // grid: x = groups of four rows (grid-stride, bounded whatever the row capacity of the table), y = ky
    const int kyn = dp.k[1];
    const int vec = (nbr_stride % 4 == 0) && (reinterpret_cast<uintptr_t>(nbr) % 16 == 0);
    int64_t blocks = ((nbr_stride + 3) / 4 + 255) / 256;
    const int64_t cap = ((int64_t)fd::device_cu_count() * 32 + kyn - 1) / kyn;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(rulebook_kernel, dim3((unsigned)blocks, (unsigned)kyn), dim3(256), 0, fd::as_stream(stream),
                       (const unsigned long long *)in_words, in_prefix, gi, out_coords, n_out_dev, nbr_stride, out_rows, fill_tail, vec, dp, nbr);
    return fd::check_launch("fd_rulebook");
}

// -------------------------------------------------------------------------------------------------------------------
// Whole index pyramid in one call.  The kernels above are a few microseconds each; launched one ctypes call at a time
// the front of a sweep was bound by host launch latency (≈45 calls, GPU idle more than half of the first 0.8 ms).
extern "C" int fd_index_pyramid(const int32_t *coords, const int32_t *n_dev, int64_t n_max_per_sample, int B, int n_levels,
                                const fd_index_level *levels, int32_t *counts_dev, void *workspace, size_t workspace_bytes,
                                fd_stream_t stream) {
    FD_REQUIRE(coords && levels && counts_dev && workspace, "fd_index_pyramid: null argument");
    FD_REQUIRE(n_levels >= 1 && n_levels <= 8 && B >= 1, "fd_index_pyramid: need 1 <= n_levels <= 8 and B >= 1");
    const fd_index_level &l0 = levels[0];
    FD_REQUIRE(l0.words, "fd_index_pyramid: null level buffer");
    // level 0 is marked with atomicOr: its words are cleared here (a kernel of this library -- graph-safe, see fd::fill_words);
    // the words of every other level are overwritten column by column by idx_down
    {
        const int64_t nc0 = fd::make_geom(B, l0.D, l0.H, l0.W).num_cols();
        int rc = fd::fill_words(l0.words, 0u, (size_t)nc0 * 2, fd::as_stream(stream));
        if (rc != FD_OK) return rc;
    }
    for (int b = 0; b < B; ++b) {
        int rc = fd_index_mark(coords + (int64_t)b * n_max_per_sample * 4, n_dev ? n_dev + b : nullptr, n_max_per_sample, B, l0.D, l0.H, l0.W,
                               l0.words, stream);
        if (rc != FD_OK) return rc;
    }
    for (int l = 0; l < n_levels; ++l) {
        const fd_index_level &lv = levels[l];
        FD_REQUIRE(lv.words && lv.prefix, "fd_index_pyramid: null level buffer");
        if (l > 0) {
            const fd_index_level &pv = levels[l - 1];
            int rc = fd_index_downsample(pv.words, B, pv.D, pv.H, pv.W, lv.ksize, lv.stride, lv.pad, lv.words, stream);
            if (rc != FD_OK) return rc;
        }
    }
    PyramidLevels L;
    FD_REQUIRE(fill_levels(L, B, n_levels, levels), "fd_index_pyramid: grid too large");
    const int total_blocks = L.blk0[n_levels];
    if (workspace_bytes < sizeof(int) * (size_t)(total_blocks + 1)) {
        fd::set_error("fd_index_pyramid: workspace %zu < %zu (block sums of all levels)", workspace_bytes, sizeof(int) * (size_t)(total_blocks + 1));
        return FD_EWORKSPACE;
    }
    int *bsum = (int *)workspace;
    hipStream_t st = fd::as_stream(stream);
    hipLaunchKernelGGL(idx_scan1_ml, dim3(total_blocks), dim3(kScanThreads), 0, st, L, bsum);
    // every non-empty level brought its coordinate table: write the coordinates in the same pass (fd_index_pyramid_coords is then
    // not needed); otherwise the caller reads the counts, sizes the tables and calls fd_index_pyramid_coords
    bool fused = true;
    for (int l = 0; l < n_levels; ++l) fused = fused && levels[l].coords != nullptr;
    if (fused) hipLaunchKernelGGL(idx_scan3_ml<true>, dim3(total_blocks), dim3(kScanThreads), 0, st, L, (const int *)bsum, counts_dev);
    else hipLaunchKernelGGL(idx_scan3_ml<false>, dim3(total_blocks), dim3(kScanThreads), 0, st, L, (const int *)bsum, counts_dev);
    return fd::check_launch("fd_index_pyramid");
}

extern "C" int fd_index_pyramid_coords(int B, int n_levels, const fd_index_level *levels, fd_stream_t stream) {
    FD_REQUIRE(levels && n_levels >= 1 && n_levels <= 8, "fd_index_pyramid_coords: bad argument");
    PyramidLevels L;
    FD_REQUIRE(fill_levels(L, B, n_levels, levels), "fd_index_pyramid_coords: grid too large");
    bool any = false;
    for (int l = 0; l < n_levels; ++l) any = any || levels[l].coords != nullptr;  // (NULL: empty level)
    if (!any) return FD_OK;
    hipLaunchKernelGGL(idx_coords_ml, dim3(L.cblk0[n_levels]), dim3(256), 0, fd::as_stream(stream), L);
    return fd::check_launch("fd_index_pyramid_coords");
}
