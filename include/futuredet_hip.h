/*
 * futuredet_hip.h -- C ABI of libfuturedet_hip.so, the MI355X (gfx950) implementation of the
 * FutureDet LiDAR inference hot path.  Plain pointers and sizes only (no torch types); every buffer is
 * caller-owned DEVICE memory unless marked host; every call is asynchronous on `stream` (a hipStream_t
 * passed as void*), performs no allocation and no device synchronisation, and returns 0 on success or a
 * negative FD_E* code (never exit()).  fd_last_error() returns a thread-local description of the last
 * failure.  Calls are re-entrant across threads, streams and devices: a call works on the device that is
 * current on the calling thread (the one `stream` belongs to); the only process-level state is a per-device
 * cache of device attributes / kernel LDS limits (atomics) and the fd_tuning_set knobs.  Sizes that the GPU decides (voxel count, active rows per level, detections) are written to
 * device int32 counters supplied by the caller; host code reads them when it needs them.
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference tree).
 */
#ifndef FUTUREDET_HIP_H
#define FUTUREDET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FD_OK 0
#define FD_EINVAL (-1)    /* bad argument (null pointer, unsupported channel count, ...) */
#define FD_EWORKSPACE (-2) /* workspace too small */
#define FD_ELAUNCH (-3)   /* hipLaunch / runtime error (text in fd_last_error) */

typedef void *fd_stream_t; /* hipStream_t */

int fd_abi_version(void); /* 8 (round 6.  7: + fd_forecast_from_detections / fd_forecast_buffers; fd_sweep_assemble takes n_rows as an upper
                             * bound -- rows past the last descriptor's row_end are dropped.  8: struct fd_decode_cfg grew by nms_kind /
                             * n_radius / circle_radius (circular NMS) -- callers must zero-initialise the struct) */
const char *fd_last_error(void);
/* Tuning / test knobs (no reference counterpart).  0 = built-in heuristic.  Names: "spconv_rg" (rows per wave of the
 * register sparse-conv kernel: 1|2|4), "spconv_v1" (1: fp32 on the register kernel instead of the compacting one),
 * "spconv_bf16_v1", "v2_depth", "v2_tm", "v2_ldspad", "conv_nt" (output channels per workgroup of the bf16 dense conv: 64 | 32),
 * "v2_ranges_per_cu", "v2_uniform", "v2_rowcost", "spconv_c32", "bf16_gp", "bf16_rg", "bf16_depth", "bf16_nw",
 * "strict" (1: a bf16 sparse layer that the gather-pipeline kernels cannot take is an error instead of a fall-back to the register
 * kernels), "bf16_win" (-1: the 64 -> 64 / 128 -> 128 bf16 layers on the RING / RESIDENT kernels instead of the LDS-window
 * kernel of round 5; 2: such a layer on the window kernel even when it is not SubM-like), "bf16_nw" (1: RESIDENT bf16 kernels walk
 * XCD-contiguous tile chunks instead of tiles interleaved over the grid; 4: four-wave window shapes), "f32_res_rg" (-1: 16-channel fp32
 * layers on the pair-compacting kernel instead of the resident-weights one), "f32_res_nw" (1: that kernel on XCD-contiguous tile chunks
 * instead of interleaved tiles), "conv_strip" (1: stride-1 bf16 dense layers on strips of 128
 * consecutive pixels instead of 8 x 16 tiles: faster alone, slower with several sweeps in flight; -1: the ragged grid of 8 x 16
 * tiles instead of the default mixed tiling -- whole 8 x 16 tiles plus edge tiles of other shapes; results identical in all three).  Initial
 * values come from the FD_SPCONV_RG, FD_SPCONV_V1, ... environment variables (FD_ + the upper-case name), read once when the
 * library is loaded; nothing on the launch path calls getenv(). */
int fd_tuning_set(const char *name, int value);

/* ---------------------------------------------------------------------------------------------------
 * Device-side counts.  The number of voxels of a sweep and the number of active rows of every sparse level exist only
 * on the device.  Every entry point on the sweep path therefore takes a host-side CAPACITY (n_points, n_max, n_out,
 * nbr_stride ...) and, optionally, a device pointer to the actual count (n_points_dev, n_dev, n_out_dev): kernels are
 * launched for the capacity (or a bounded grid) and work on min(capacity, *count).  With the counts left on the device a
 * whole sweep -- voxelizer to NMS -- is issued without a host read-back and can be captured into one hipGraph
 * (futuredet_amd.detectors.VoxelNet.forward_points_static).  Passing NULL keeps the plain host-count behaviour.
 * ------------------------------------------------------------------------------------------------- */
/* ---------------------------------------------------------------------------------------------------
 * Voxelizer (+ fused mean reader).
 * Replaces points_to_voxel(points, voxel_size, coors_range, max_points, reverse_index=True, max_voxels)
 *   det3d/ops/point_cloud/point_cloud_ops.py:112-184 (kernel :7-55), called from
 *   det3d/core/input/voxel_generator.py:19-30 <- det3d/datasets/pipelines/preprocess.py:244-271,
 * and, when out_mean != NULL, VoxelFeatureExtractorV3.forward (det3d/models/readers/voxel_encoder.py:17-24)
 * and the batch-index prefix of collate_kitti_multi (det3d/torchie/parallel/collate.py:199-206).
 * Semantics are the reference's sequential ones, computed deterministically in parallel: voxels are numbered
 * in first-occurrence order of the input points, the first max_voxels voxels survive, each keeps its first
 * max_points points in input order; c = floor((p - lo) / vs) in float32 with a true division.
 *   points      [n, ndim] float32 rows (x,y,z,...)          ndim <= 8
 *   n_points_dev  NULL, or device int32[1]: the cloud has min(n_points, *n_points_dev) rows and n_points is the capacity of
 *                 a padded buffer (futuredet_amd.loading.assemble_device) -- see "Device-side counts" below
 *   out_voxels  [max_voxels, max_points, ndim] or NULL      (zero padded, as the reference returns)
 *   out_mean    [max_voxels, mean_stride] or NULL           (sum of kept points / count; cols >= ndim zeroed)
 *   out_coors   [max_voxels, coor_cols]; coor_cols 3 -> (z,y,x); 4 -> (batch_idx,z,y,x)
 *   out_num_points [max_voxels] int32, out_num_voxels device int32[1]
 * ------------------------------------------------------------------------------------------------- */
size_t fd_voxelize_workspace_bytes(int64_t n_points, int64_t max_voxels);
int fd_voxelize(const float *points, int64_t n_points, const int32_t *n_points_dev, int ndim, const float *range6_host,
                const float *voxel_size3_host, int max_points, int64_t max_voxels, int batch_idx,
                float *out_voxels, float *out_mean, int mean_stride, int32_t *out_coors, int coor_cols,
                int32_t *out_num_points, int32_t *out_num_voxels, void *workspace, size_t workspace_bytes,
                fd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Sparse index + rulebook.  Replaces spconv 1.0's get_indice_pairs (invoked implicitly by every
 * SubMConv3d / SparseConv3d with a new indice_key: det3d/models/backbones/scn.py:99,110,115,120,125,130,
 * 135,141) and SparseConvTensor's coordinate bookkeeping (scn.py:154).
 *
 * Native layout (not spconv's pair lists): an active set on a grid (B, D<=64, H, W) is one 64-bit
 * occupancy word per (b,y,x) column (bit z) plus an exclusive prefix count per column; columns are
 * linearised in 8x8 tiles, so row numbers are spatially sorted:
 *     col(b,y,x) = ((b*ceil(H/8) + y/8)*ceil(W/8) + x/8)*64 + (y%8)*8 + x%8
 *     row(b,z,y,x) = prefix[col] + popcount(words[col] & ((1<<z)-1))
 * The rulebook is OUTPUT-stationary: nbr[k][o] = input row feeding output row o through kernel tap k
 * (k row-major over (kz,ky,kx), cross-correlation: input = o*stride - pad + k) or -1.
 * ------------------------------------------------------------------------------------------------- */
int64_t fd_index_num_cols(int B, int H, int W);                 /* length of words[] / prefix[] */
size_t fd_index_workspace_bytes(int64_t num_cols);
/* words must be zeroed by the caller (hipMemsetAsync) before fd_index_mark. */
int fd_index_mark(const int32_t *coords /*[n,4] (b,z,y,x)*/, const int32_t *n_dev /*device count, or NULL*/,
                  int64_t n_max, int B, int D, int H, int W, uint64_t *words, fd_stream_t stream);
/* marks the output set of a strided conv: o = (p + pad - k)/stride where integral and inside out grid; every word of out_words is
 * OVERWRITTEN (gather form, no atomics: nothing is OR-ed into what out_words held before) */
int fd_index_downsample(const uint64_t *in_words, int B, int D, int H, int W, const int *ksize3,
                        const int *stride3, const int *pad3, uint64_t *out_words, fd_stream_t stream);
/* exclusive scan of popcounts -> prefix[], total -> n_active_dev[0] */
int fd_index_scan(const uint64_t *words, int64_t num_cols, int32_t *prefix, int32_t *n_active_dev,
                  void *workspace, size_t workspace_bytes, fd_stream_t stream);
/* coords[row] = (b,z,y,x) for every active row */
int fd_index_coords(const uint64_t *words, const int32_t *prefix, int B, int D, int H, int W,
                    int32_t *coords /*[n_active,4]*/, fd_stream_t stream);
/* row_of[j] = row of coords_in[j] in the index (or -1) */
int fd_index_lookup(const uint64_t *words, const int32_t *prefix, int B, int D, int H, int W,
                    const int32_t *coords_in, const int32_t *n_dev, int64_t n_max, int32_t *row_of,
                    fd_stream_t stream);
/* dst[row_of[j], 0:c_dst] = src[j, 0:c_src] (zero padded to c_dst); rows with row_of<0 skipped */
int fd_rows_permute(const float *src, int c_src, const int32_t *row_of, const int32_t *n_dev, int64_t n_max,
                    void *dst, int c_dst, int dst_bf16, fd_stream_t stream);
/* nbr [K, nbr_stride] int32; out_coords holds out_rows (<= nbr_stride) rows, the kernel works on min(out_rows, *n_out_dev).
 * fill_tail != 0: rows o >= that count are filled with -1 up to nbr_stride;
 * fill_tail == 0: rows >= the count are left untouched (capacity-sized tables whose consumers are given the same device count) */
int fd_rulebook(const uint64_t *in_words, const int32_t *in_prefix, int B, int Din, int Hin, int Win,
                const int32_t *out_coords, int64_t out_rows, const int32_t *n_out_dev, int64_t nbr_stride, int fill_tail, const int *ksize3,
                const int *stride3, const int *pad3, int32_t *nbr, fd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Sparse convolution apply (gather - GEMM - no scatter).  Replaces spconv 1.0 indice_conv /
 * indice_subm_conv as used by the 21 convolutions of SpMiddleResNetFHD (scn.py:99-141) together with the
 * eval-mode BatchNorm1d / ReLU / residual add that follow them (scn.py:67-78,100-101,...): BN is folded
 * into weight/bias by the caller.
 *   out[o,:] = act( sum_k in[nbr[k][o],:] @ W[k] + bias (+ residual[o,:]) )
 *   in_feats  [n_in, cin]  float32 (dtype 0) or bfloat16 (dtype 1);  cin, cout in {16,32,64,128}
 *   wpacked   weights in MFMA fragment order, produced by fd_spconv_pack_weight from [K,cin,cout] float32
 *   bias      [cout] float32 or NULL; residual [n_out,cout] same dtype as out or NULL; relu 0/1
 * ------------------------------------------------------------------------------------------------- */
size_t fd_spconv_packed_weight_bytes(int K, int cin, int cout, int dtype);
int fd_spconv_pack_weight(const float *w_kio_host, int K, int cin, int cout, int dtype, void *wpacked_host);
int fd_spconv_apply(const void *in_feats, int64_t n_in, const void *wpacked, const float *bias, const void *residual,
                    int relu, const int32_t *nbr, int64_t nbr_stride, const int32_t *ranges /* [n_ranges+1] or NULL */,
                    int n_ranges, int K, int64_t n_out, const int32_t *n_out_dev /* NULL, or the device's row count */,
                    int64_t n_expected /* 0, or a typical row count that steers the launch heuristics when n_out is a capacity */,
                    int cin, int cout, int dtype, void *out_feats, fd_stream_t stream);
/* Work distribution of fd_spconv_apply (fp32; no reference counterpart -- spconv launches one GEMM per tap).  Workgroup b
 * computes output rows ranges[b] .. ranges[b+1]-1 (consecutive rows = neighbouring voxels), in chunks of <= 128 rows.
 *   ranges == NULL, n_ranges == 0 : one 128-row tile per workgroup
 *   ranges == NULL, n_ranges  > 0 : n_ranges equal row counts
 *   ranges != NULL                : device table from fd_spconv_ranges: equal WORK (16-pair MFMA groups, counted per
 *                                   8-row block of the rulebook) per range; boundaries are multiples of 8 rows.
 * fd_spconv_num_ranges returns the count the kernel prefers for a layer shape on the current device (a whole number of
 * ranges per compute unit).  The table depends only on the rulebook, so the 4-5 convolutions that share an indice_key
 * share it.  Results never depend on the distribution (fixed summation order per output row). */
int fd_spconv_num_ranges(int64_t n_out, int cin, int cout, int dtype);
int fd_spconv_wants_balanced_ranges(int cin, int cout, int dtype); /* 1: use fd_spconv_ranges; 0: equal rows (ranges = NULL) */
size_t fd_spconv_ranges_workspace_bytes(int64_t n_out);
int fd_spconv_ranges(const int32_t *nbr, int64_t nbr_stride, int K, int64_t n_out, const int32_t *n_out_dev, int n_ranges,
                     int32_t *ranges /*[n_ranges+1]*/, void *workspace, size_t workspace_bytes, fd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Densify.  Replaces SparseConvTensor.dense() + view (scn.py:165-168): out[b, c*D + d, y, x] = feats[row, c].
 * Every output element is written exactly once (zeros where inactive); element strides let the caller
 * choose NCHW or channels-last memory.  out_dtype 0 float32, 1 bfloat16.
 * ------------------------------------------------------------------------------------------------- */
int fd_densify(const void *feats, int c, int dtype, const uint64_t *words, const int32_t *prefix, int B, int D,
               int H, int W, void *out, int out_dtype, int64_t stride_b, int64_t stride_c, int64_t stride_y,
               int64_t stride_x, int64_t n_rows /* rows of feats: an index row >= n_rows (a capacity-sized level that overflowed) reads as zero */,
               fd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Dense 2-D convolution, bf16 NHWC, fp32 accumulate, fused bias + ReLU (hand-written MFMA implicit GEMM).
 * Replaces the cuDNN convolutions of the RPN neck and CenterHead in the bf16 configuration
 * (det3d/models/necks/rpn.py:81-140, det3d/models/bbox_heads/center_head.py:104-143,344-349) with eval-mode
 * BatchNorm folded into weight/bias by the caller.
 *   x [B,H,W,cin] bf16; wpacked from fd_conv2d_pack_weight([cout][cin][ks][ks] float32 host); bias [cout] f32 or NULL
 *   supported: 3x3 stride 1|2 pad 1, 1x1 stride 1 pad 0; cin % 32 == 0
 *   y element (b, oy*osy+ooy, ox*osx+oox, co_off+co) of a [B, Ho*osy, Wo*osx, cout_total] bf16 tensor -- channel
 *   offsets express a concat, pixel strides/offsets express a 2x2 stride-2 transposed conv as four 1x1 convs.
 * ------------------------------------------------------------------------------------------------- */
size_t fd_conv2d_packed_weight_bytes(int cout, int cin, int ks);
int fd_conv2d_pack_weight(const float *w_oihw_host, int cout, int cin, int ks, void *wpacked_host);
int fd_conv2d_nhwc_bf16(const void *x, int B, int H, int W, int cin, const void *wpacked, const float *bias, int cout,
                        int ks, int stride, int pad, int relu, void *y, int cout_total, int co_off, int osy, int osx,
                        int ooy, int oox, fd_stream_t stream);
/* The same convolution in fp32 (fp32 activations / weights / output, v_mfma_f32_16x16x4_f32): the RPN / CenterHead layers of
 * the fp32 configurations, where the reference runs cuDNN (det3d/models/necks/rpn.py:124-159,
 * det3d/models/bbox_heads/center_head.py:129-143,344-349).  cin must be a multiple of 16; same placement arguments. */
size_t fd_conv2d_f32_packed_weight_bytes(int cout, int cin, int ks);
int fd_conv2d_f32_pack_weight(const float *w_oihw_host, int cout, int cin, int ks, void *wpacked_host);
int fd_conv2d_f32_num_tiles(void); /* number of workgroup tile shapes instantiated (valid `tile` values are 1..n) */
int fd_conv2d_nhwc_f32(const float *x, int B, int H, int W, int cin, const void *wpacked, const float *bias, int cout, int ks,
                       int stride, int pad, int relu, float *y, int cout_total, int co_off, int osy, int osx, int ooy, int oox,
                       int tile /* 0 = library heuristic, else 1..fd_conv2d_f32_num_tiles() */, fd_stream_t stream);
/* ConvTranspose2d(k, stride k) -- the RPN's upsampling deblock (det3d/models/necks/rpn.py:98-110) -- as ONE 1x1 convolution to
 * k*k*cout_sub virtual channels with a pixel-shuffle epilogue.  Weights: fd_conv2d_f32_pack_weight of [(dy,dx,co), cin, 1, 1];
 * bias [cout_sub]; y [B, H*k, W*k, cout_total]. */
int fd_conv2d_shuffle_nhwc_f32(const float *x, int B, int H, int W, int cin, const void *wpacked, const float *bias, int cout_sub, int k,
                               int relu, float *y, int cout_total, int co_off, int tile, fd_stream_t stream);
/* Grouped 3x3 stride-1 pad-1 convolution, <= 16 outputs per group: the final convolutions of the six CenterHead branches
 * (det3d/models/bbox_heads/center_head.py:129-143) in one launch.  x [B,H,W,groups*cin_g]; weights: fd_conv2d_f32_pack_weight of
 * [groups*16, cin_g, 3, 3] (each group padded to 16 outputs); bias [groups*16]; counts_host[g] = real outputs of group g,
 * written back to back from channel co_off of y. */
int fd_conv2d_grouped_nhwc_f32(const float *x, int B, int H, int W, int groups, int cin_g, const void *wpacked, const float *bias,
                               const int *counts_host, int relu, float *y, int cout_total, int co_off, int tile, fd_stream_t stream);
/* 3x3 stride-1 pad-1 only: Winograd F(2x2,3x3) with the 16 element-wise products as MFMA GEMMs over the input channels
 * (2.25x fewer multiplies than the direct form).  Weights are transformed (U = G g G^T) and packed by
 * fd_conv2d_wino_f32_pack_weight; cin must be a multiple of 16; output placement: channel offset only.  Every tile gives the
 * same bits; the last one (strips of 32 Winograd tiles, producer + consumer waves) needs W >= 63 and tensors below 2 GB and
 * fails with FD_EINVAL otherwise. */
int fd_conv2d_wino_f32_num_tiles(void);
size_t fd_conv2d_wino_f32_packed_weight_bytes(int cout, int cin);
int fd_conv2d_wino_f32_pack_weight(const float *w_oihw_host, int cout, int cin, void *wpacked_host);
int fd_conv2d_wino_nhwc_f32(const float *x, int B, int H, int W, int cin, const void *wpacked, const float *bias, int cout, int relu,
                            float *y, int cout_total, int co_off, int tile /* 0 = default, else 1..num_tiles */, fd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * CenterPoint decode + rotated NMS.  Replaces CenterHead.predict's per-step decode
 * (det3d/models/bbox_heads/center_head.py:609-673), post_processing (:699-747), rotate_nms_pcdet
 * (det3d/core/bbox/box_torch_ops.py:248-277) and iou3d_nms_cuda.nms_gpu (det3d/ops/iou3d_nms/src/
 * iou3d_nms.cpp:90-135, iou3d_nms_kernel.cu:104-311) -- including the greedy sweep, which runs on the device.
 * One "group" = one (sample, heat-map) pair; G groups are decoded in one call.
 *   hm     [G, HW] float32 logits (single class)           reg [G,2,HW] height [G,1,HW] dim [G,3,HW] rot [G,2,HW]
 *   (channel-planar NCHW slices; *_gstride = element stride between groups)
 *   out_boxes7 [G, post_max, 7]  (x,y,z,w,l,h,yaw as predict emits them, vel excluded)
 *   out_scores [G, post_max], out_cell [G, post_max] int32 (BEV cell index, for gathering vel), out_count [G]
 * ------------------------------------------------------------------------------------------------- */
typedef struct fd_decode_cfg {
    int H, W;
    float out_size_factor, voxel_x, voxel_y, pc_x, pc_y; /* test_cfg.out_size_factor / voxel_size / pc_range */
    float score_threshold;
    float center_range[6];                               /* post_center_limit_range */
    float nms_iou_threshold;
    int nms_pre_max, nms_post_max;
    int hm_channels; /* 0 / 1: one heat-map channel; n > 1: score = max over n channels (center_head.py:589-595, the `classify` head) */
    /* test_cfg.circular_nms (center_head.py:722-725 -> _circle_nms :750-758 -> core/utils/circle_nms_jit.py): nms_kind 1 replaces the
     * rotated-IoU predicate by "squared centre distance <= min_radius" (the reference compares min_radius with the SQUARED distance, in
     * float32); decode group g takes circle_radius[g / (G / n_radius)] (test_cfg.min_radius[task_id]: one entry per group, n_radius must
     * divide G).  The reference applies no pre-NMS cut in this mode; here the nms_pre_max (<= 4096) best candidates are taken, which gives
     * the reference's result whenever a group has no more candidates than that OR nms_post_max boxes are kept among them (the greedy
     * order makes the first kept boxes independent of later candidates).  A group for which neither holds reports count -1 and no rows
     * (out_count and counts_out): never a silently different answer.  Equal scores: the reference's order among them is numpy's
     * unstable argsort reversed, here the lower cell index first. */
    int nms_kind; /* 0: rotated BEV IoU > nms_iou_threshold (rotate_nms_pcdet); 1: circular */
    int n_radius; /* nms_kind 1: entries of circle_radius in use, 1..16 */
    float circle_radius[16];
} fd_decode_cfg;

/* A head map as the decode reads it: element (group g, channel ch, BEV cell) of a float32 (dtype 0) or bf16 (dtype 1) tensor at
 * data[g * group_stride + ch * channel_stride + cell * cell_stride] (strides in elements).  NCHW planes: channel_stride = H*W,
 * cell_stride = 1.  A channel slice of an NHWC convolution output: channel_stride = 1, cell_stride = C_total, data = base + c0 --
 * the decode then reads the head's output in place (center_head.py:559-697 works on the permuted NHWC tensors the same way). */
typedef struct fd_map_view {
    const void *data;
    int64_t group_stride, channel_stride, cell_stride;
    int dtype;
} fd_map_view;

size_t fd_decode_workspace_bytes(int G, const fd_decode_cfg *cfg);
/* fd_centerpoint_decode on map views (any layout / float32 or bf16); the group index of a view is g = group * B + sample */
int fd_centerpoint_decode_maps(const fd_map_view *hm, const fd_map_view *reg, const fd_map_view *height, const fd_map_view *dim,
                               const fd_map_view *rot, int G, const fd_decode_cfg *cfg_host, float *out_boxes7, float *out_scores,
                               int32_t *out_cell, int32_t *out_count, void *workspace, size_t workspace_bytes, fd_stream_t stream);
/* Final assembly of CenterHead.predict's output (center_head.py:559-570 standard head: step s = the shared boxes + velocity
 * channels 2s, 2s+1; :606-607 dense head: step s = task s; :672-697 label offsets and concatenation): for sample b and output step s
 * the kept boxes of decode group step_group[s] * B + b with the velocity read from ``vel`` (a map view whose group index is
 * group * B + sample) at channels step_vel_channel[s], +1 and the label step_label[s], as packed rows
 * [B, S, post_max, 11] = x y z w l h vx vy yaw score label (rows >= count zero) and counts_out [B, S].  step_* are HOST arrays (S <= 16). */
int fd_assemble_detections(const float *boxes7, const float *scores, const int32_t *cell, const int32_t *count, const fd_map_view *vel,
                           int B, int post_max, int S, const int32_t *step_group, const int32_t *step_vel_channel,
                           const int32_t *step_label, float *packed, int32_t *counts_out, fd_stream_t stream);
/* fd_centerpoint_decode_maps + fd_assemble_detections in ONE call: the last kernel of the decode (greedy sweep from LDS, gather of the kept
 * boxes) also writes the packed rows of every output step (arguments as in the two calls above; G = groups * B, decode group
 * g = group * B + sample; step_* are HOST arrays, S <= 16).  out_boxes7 / out_scores / out_cell / out_count receive what
 * fd_centerpoint_decode_maps would write. */
int fd_centerpoint_decode_packed(const fd_map_view *hm, const fd_map_view *reg, const fd_map_view *height, const fd_map_view *dim,
                                 const fd_map_view *rot, const fd_map_view *vel, int G, int B, const fd_decode_cfg *cfg_host, int S,
                                 const int32_t *step_group, const int32_t *step_vel_channel, const int32_t *step_label, float *out_boxes7,
                                 float *out_scores, int32_t *out_cell, int32_t *out_count, float *packed, int32_t *counts_out, void *workspace,
                                 size_t workspace_bytes, fd_stream_t stream);
int fd_centerpoint_decode(const float *hm, int64_t hm_gstride, const float *reg, int64_t reg_gstride,
                          const float *height, int64_t height_gstride, const float *dim, int64_t dim_gstride,
                          const float *rot, int64_t rot_gstride, int G, const fd_decode_cfg *cfg_host,
                          float *out_boxes7, float *out_scores, int32_t *out_cell, int32_t *out_count,
                          void *workspace, size_t workspace_bytes, fd_stream_t stream);

/* Stand-alone rotated NMS with nms_gpu's contract (boxes [n,7] pcdet layout, already score-sorted):
 * keep[0:count] = kept indices (int64), count -> out_count (device int32).  Replaces
 * iou3d_nms_cuda.nms_gpu (iou3d_nms_api.cpp:11-17, iou3d_nms.cpp:90-135). */
size_t fd_nms_workspace_bytes(int n);
int fd_rotated_nms(const float *boxes7, int n, float thresh, int64_t *keep, int32_t *out_count,
                   void *workspace, size_t workspace_bytes, fd_stream_t stream);
/* pairwise rotated BEV IoU, replaces boxes_iou_bev_gpu (iou3d_nms.cpp:49-69) */
int fd_boxes_iou_bev(const float *a7, int na, const float *b7, int nb, float *out, fd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Sweep assembly (the step in front of the voxelizer).  Replaces the NuScenes branch of
 * LoadPointCloudFromFile.__call__ (det3d/datasets/pipelines/loading.py:107-141) together with read_file's
 * column cut (:31), remove_close (:36-45) and read_sweep (:47-60): the raw rows of the key frame and of the
 * sweeps (in the order the reference visits them) are filtered, moved into the key frame's coordinates and
 * given their time column, in input order.
 *   raw          [n_rows, raw_cols] float32, the concatenated contents of the .bin files (raw_cols = 5)
 *   sweeps_dev   [n_sweeps] descriptors IN DEVICE MEMORY, consecutive row ranges starting at row 0.  n_rows is an UPPER BOUND
 *                (ABI 7): rows at or past the last descriptor's row_end are dropped, so a fixed-capacity buffer whose fill is only
 *                known to the descriptors can be assembled by a launch captured once (FullSweepStep, detectors.py)
 *                m = transform_matrix (row-major 4x4, float64 as the reference's np.dot evaluates it; ignored
 *                unless FD_SWEEP_HAS_TRANSFORM), time = float32(time_lag) (0 for the key frame)
 *   out_points   [n_rows, keep_cols + 1] float32; rows [0, *out_count) are the reference's
 *                res["lidar"]["combined"]; rows [*out_count, n_rows) are filled with +inf (outside any range)
 *   out_count    device int32[1]
 * ------------------------------------------------------------------------------------------------- */
#define FD_SWEEP_HAS_TRANSFORM 1 /* sweep["transform_matrix"] is not None (loading.py:53) */
#define FD_SWEEP_REMOVE_CLOSE 2  /* read_sweep's remove_close (loading.py:50); not applied to the key frame (:113) */
typedef struct fd_sweep_desc {
    double m[16];
    int64_t row_begin, row_end;
    float time;
    int32_t flags;
} fd_sweep_desc;
size_t fd_sweep_assemble_workspace_bytes(int64_t n_rows);
int fd_sweep_assemble(const float *raw, int raw_cols, int keep_cols, int64_t n_rows, const fd_sweep_desc *sweeps_dev,
                      int n_sweeps, float min_distance, float *out_points, int32_t *out_count, void *workspace,
                      size_t workspace_bytes, fd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * PointPillars reader.  Replaces PillarFeatureNet.forward (det3d/models/readers/pillar_encoder.py:113-164) and
 * its PFNLayers (:38-55: Linear(bias=False) -> eval BatchNorm1d -> ReLU -> max over the pillar's points
 * [-> concat with the repeated max]) in one launch; nothing of shape [M, P, C] is written to memory.
 *   voxels      [m_max, max_points, ndim] float32 zero padded (fd_voxelize's out_voxels)
 *   num_points  [m_max] int32;  coors4 [m_max, 4] int32 (b,z,y,x);  n_dev: device pillar count or NULL (= m_max)
 *   vx, vy, x_offset, y_offset : PillarFeatureNet.vx/.vy/.x_offset/.y_offset (:107-110)
 *   w1 [units1, ndim+5(+1)] = pfn_layers[0].linear.weight; scale/shift = the eval BatchNorm1d as y = x*scale+shift
 *   w2 [units2, 2*units1]   = pfn_layers[1].linear.weight or NULL for a single layer (num_filters=(64,))
 *   out         [m_max, out_stride] float32 (out_dtype 0) or bf16 (1); rows >= the pillar count are not written
 * ------------------------------------------------------------------------------------------------- */
int fd_pillar_encode(const float *voxels, const int32_t *num_points, const int32_t *coors4, const int32_t *n_dev,
                     int64_t m_max, int max_points, int ndim, int with_distance, float vx, float vy, float x_offset,
                     float y_offset, const float *w1, const float *scale1, const float *shift1, int units1,
                     const float *w2, const float *scale2, const float *shift2, int units2, int out_dtype, void *out,
                     int out_stride, fd_stream_t stream);

/* PointPillarsScatter.forward (pillar_encoder.py:186-221): out[b, c, y, x] = feats[row, c] for coors4[row] = (b,_,y,x),
 * zero elsewhere (zero_first != 0 clears the dense B*c*H*W canvas first).  Strides are in elements, so the same call
 * writes an NCHW float32 canvas (reference layout) or an NHWC bf16 one (hand-written conv path). */
int fd_pillar_scatter(const void *feats, int c, int feat_stride, int dtype, const int32_t *coors4, const int32_t *n_dev,
                      int64_t m_max, int B, int H, int W, void *out, int out_dtype, int64_t stride_b, int64_t stride_c,
                      int64_t stride_y, int64_t stride_x, int zero_first, fd_stream_t stream);


/* ---------------------------------------------------------------------------------------------------
 * Forecast association: the numeric core of `tracker` (det3d/datasets/nuscenes/nuscenes.py:125-257), `match_boxes`
 * (:112-123) and `distance_matrix` (:100-110) for one sweep: T forecast steps with counts[t] <= n_max boxes each.
 *   centers, velocity [T, n_max, 3] float64 (Box.center / Box.velocity);  time_dev [T-1] float64 (seconds between steps)
 *   fwd_idx [n_max, T], fwd_ok [n_max] : forward chain of step-0 box i (index per step) and "not void" flag
 *                                        (every hop <= reject_thresh, :160-173)
 *   bwd_idx [n_max, T], bwd_ok [n_max] : back-cast chain of last-step box i, hop s goes from step T-1-s to T-2-s (:222-237)
 *   match_idx [T, n_max]               : match_boxes: nearest step-t box to step-0 box i
 *   cv_centers [n_max, T, 3]           : constant-velocity forward trajectory of step-0 box i (:183-193)
 *   status int32[1]                    : 1 when some step has no box (the reference then returns no trajectory)
 * ------------------------------------------------------------------------------------------------- */
int fd_forecast_chains(const double *centers, const double *velocity, const int32_t *counts, const double *time_dev, int T,
                       int n_max, double reject_thresh, int32_t *fwd_idx, int32_t *fwd_ok, int32_t *bwd_idx, int32_t *bwd_ok,
                       int32_t *match_idx, double *cv_centers, int32_t *status, fd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Head output -> global-frame boxes: _second_det_to_nusc_box (det3d/datasets/nuscenes/nusc_common.py:167-189: yaw ->
 * -yaw - pi/2 in float32, Quaternion(axis z, that angle), velocity (vx, vy, 0)) followed by _lidar_nusc_box_to_global
 * (:192-216: rotate + translate by the calibrated_sensor record, then by the ego_pose record).  The two records are host
 * arrays (rotation w,x,y,z / translation x,y,z, float64) instead of devkit table look-ups; pass NULL pairs to stay in
 * the lidar frame.  Outputs: center [n,3], quat [n,4] (w,x,y,z), velocity [n,3] float64 (Box.center / .orientation /
 * .velocity), size [n,3] float32 (Box.wlh).
 * ------------------------------------------------------------------------------------------------- */
int fd_det_to_global_boxes(const float *box3d9, int n, const double *cs_rotation4_host, const double *cs_translation3_host,
                           const double *pose_rotation4_host, const double *pose_translation3_host, double *center,
                           double *quat, double *velocity, float *size, fd_stream_t stream);
/* multi_future's grouping (det3d/datasets/nuscenes/nuscenes.py:299-339 with network_split :283-297): boxes whose centres
 * (all three coordinates, distance_matrix :100-110) are closer than match_thresh are linked; ids[i] = index of box i's
 * connected component, components numbered by their smallest member (networkx's enumeration order).  n <= 8192. */
int fd_forecast_groups(const double *centers3, int n, double match_thresh, int32_t *ids, fd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * The three calls above for a whole batch, device-resident and capturable (ABI 7): from the head's packed output
 * (fd_centerpoint_decode_packed: packed [B, T, post, row_floats] float32 rows = box 9 + score + label, counts [B, T]) to what
 * `forecast_boxes` + `tracker` + `multi_future` produce for each sweep (det3d/datasets/nuscenes/nuscenes.py:384-494 with
 * forecast_mode "velocity_dense", :125-257, :299-339; nusc_common.py:167-216), as arrays:
 *   records_dev  [B][14] float64 in DEVICE memory: calibrated_sensor rotation (w,x,y,z), translation (x,y,z), ego_pose rotation,
 *                translation of every sample (the devkit look-ups of nuscenes.py:385-398 are the caller's); NULL = lidar frame.
 *                Device memory, because a captured graph is replayed for other samples: nothing of a sample is a kernel argument
 *   time_dev     [B][T-1] float64: seconds between consecutive forecast steps (get_time, nuscenes.py:399-406)
 *   out->center / quat / velocity [B,T,post,3|4|3] float64, size [B,T,post,3] float32: every slot's global-frame box
 *   out->fwd_idx ... status: fd_forecast_chains' outputs per sample ([B, ...] in front of the shapes documented there, n_max = post)
 *   out->traj_kind / traj_src / traj_first / traj_group [B, 3*post] int32, n_traj [B] (all five or none): the sweep's trajectories in
 *                tracker's order -- kind 0 = forward chain of step-0 box src (fwd_idx[src][:]), 1 = constant-velocity roll-out of
 *                step-0 box src (cv_centers[src][:]), 2 = back-cast chain of last-step box src (bwd_idx[src][::-1]); first = the
 *                step-0 box the trajectory begins with; group = multi_future's forecast_id among the sweep's trajectories (components
 *                of "first boxes closer than match_thresh", numbered by their smallest member).  Entries past n_traj are -1.
 * Three launches (boxes, association, trajectories); bit-identical to the three single-sweep calls.
 * ------------------------------------------------------------------------------------------------- */
typedef struct fd_forecast_buffers {
    double *center, *quat, *velocity;
    float *size;
    int32_t *fwd_idx, *fwd_ok, *bwd_idx, *bwd_ok, *match_idx;
    double *cv_centers;
    int32_t *status;
    int32_t *traj_kind, *traj_src, *traj_first, *traj_group, *n_traj;
} fd_forecast_buffers;
int fd_forecast_from_detections(const float *packed, const int32_t *counts, int B, int T, int post, int row_floats,
                                const double *records_dev, const double *time_dev, double reject_thresh, double match_thresh,
                                const fd_forecast_buffers *out, fd_stream_t stream);

/* The search of process_trajectories (det3d/datasets/nuscenes/nuscenes.py:341-382, forecast_boxes with postprocess=True :465-467):
 * idx[i] = argmin_j || library[j] - queries[i] || over the rows of a trajectory library [n_library, dim] (float64, row =
 * [vx, vy, q0..q3, centre_1 - centre_0, ...]); the first minimum wins like np.argmin.  Exact squared differences instead of the
 * reference's |a|^2 + |b|^2 - 2ab expansion: the two can only disagree between library rows that are equidistant to 1e-15. */
int fd_nearest_rows(const double *library, int n_library, const double *queries, int n_queries, int dim, int32_t *idx, fd_stream_t stream);

/* The whole index pyramid of the backbone in one call (the same launches as fd_index_mark / _downsample / _scan /
 * _coords above, issued back to back): level 0 is marked from the voxelizer's coords of every sample
 * (coords [B * n_max_per_sample, 4], n_dev[b] = voxel count of sample b or NULL), level l > 0 is derived from
 * level l-1 with its ksize/stride/pad (the strided SparseConv3d of scn.py:110,120,130,141); counts_dev[l] receives
 * the active count of level l.  The caller need not initialise any level's words (level 0 is cleared by this call, the other
 * levels are overwritten); all levels are scanned by one
 * set of launches: workspace >= fd_index_workspace_bytes(sum of the levels' fd_index_num_cols + 2048 * n_levels).  fd_index_pyramid_coords materialises coords for the levels whose pointer is
 * set (after the host has read the counts and allocated them).  When EVERY level passed to fd_index_pyramid already has its
 * coords pointer set (capacity-sized tables, coords_rows = capacity), the coordinates are written in the same pass and
 * fd_index_pyramid_coords need not be called. */
typedef struct fd_index_level {
    int32_t D, H, W;
    int32_t ksize[3], stride[3], pad[3]; /* how this level derives from the previous one (unused for level 0) */
    uint64_t *words;
    int32_t *prefix;
    int32_t *coords; /* [n_l, 4] or NULL */
    int64_t coords_rows; /* rows of ``coords``: an active voxel whose row index is >= coords_rows is not written (capacity-sized
                          * levels of the sync-free step; 0 = unbounded, the caller sized coords from the counts) */
} fd_index_level;
int fd_index_pyramid(const int32_t *coords, const int32_t *n_dev, int64_t n_max_per_sample, int B, int n_levels,
                     const fd_index_level *levels_host, int32_t *counts_dev, void *workspace, size_t workspace_bytes,
                     fd_stream_t stream);
int fd_index_pyramid_coords(int B, int n_levels, const fd_index_level *levels_host, fd_stream_t stream);

/* fd_index_lookup + fd_rows_permute in one launch: dst[row(coords_in[i])][:] = src[i][:] (channels >= c_src zeroed) for
 * the first n_dev[0] (<= n_max) voxels; this is how the voxelizer's per-voxel features become rows of the level-0
 * SparseConvTensor (scn.py:154) in index order. */
int fd_rows_place(const uint64_t *words, const int32_t *prefix, int B, int D, int H, int W, const int32_t *coords_in,
                  const int32_t *n_dev, int64_t n_max, const float *src, int c_src, void *dst, int c_dst, int dst_bf16,
                  fd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FUTUREDET_HIP_H */
