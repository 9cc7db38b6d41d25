"""ctypes binding of libfuturedet_hip.so (the C ABI declared in include/futuredet_hip.h).

There is no CPU fallback: loading fails loudly when the library is missing, and every op raises
when its tensors are not on a HIP device.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FD_LIB_PATH") or os.path.join(_HERE, "libfuturedet_hip.so")  # override: tuning builds only
_lib = None

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_i64 = ctypes.c_int64
c_size_t = ctypes.c_size_t
c_float = ctypes.c_float


class MapView(ctypes.Structure):
    """fd_map_view (include/futuredet_hip.h): element (g, ch, cell) at data[g * group_stride + ch * channel_stride + cell * cell_stride]"""
    _fields_ = [("data", ctypes.c_void_p), ("group_stride", ctypes.c_int64), ("channel_stride", ctypes.c_int64), ("cell_stride", ctypes.c_int64),
                ("dtype", ctypes.c_int)]


ABI_VERSION = 8  # include/futuredet_hip.h: fd_abi_version()


class DecodeCfg(ctypes.Structure):  # struct fd_decode_cfg
    _fields_ = [("H", c_int), ("W", c_int), ("out_size_factor", c_float), ("voxel_x", c_float), ("voxel_y", c_float),
                ("pc_x", c_float), ("pc_y", c_float), ("score_threshold", c_float), ("center_range", c_float * 6),
                ("nms_iou_threshold", c_float), ("nms_pre_max", c_int), ("nms_post_max", c_int), ("hm_channels", c_int),
                ("nms_kind", c_int), ("n_radius", c_int), ("circle_radius", c_float * 16)]


class IndexLevel(ctypes.Structure):  # struct fd_index_level
    _fields_ = [("D", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("ksize", ctypes.c_int32 * 3),
                ("stride", ctypes.c_int32 * 3), ("pad", ctypes.c_int32 * 3), ("words", c_void_p), ("prefix", c_void_p),
                ("coords", c_void_p), ("coords_rows", ctypes.c_int64)]


class ForecastBuffers(ctypes.Structure):  # struct fd_forecast_buffers
    _fields_ = [(n, c_void_p) for n in ("center", "quat", "velocity", "size", "fwd_idx", "fwd_ok", "bwd_idx", "bwd_ok", "match_idx", "cv_centers",
                                        "status", "traj_kind", "traj_src", "traj_first", "traj_group", "n_traj")]


# name -> (restype, argtypes); this table is checked against include/futuredet_hip.h by the tests
SIGNATURES = {
    "fd_abi_version": (c_int, []),
    "fd_last_error": (ctypes.c_char_p, []),
    "fd_tuning_set": (c_int, [ctypes.c_char_p, c_int]),
    "fd_voxelize_workspace_bytes": (c_size_t, [c_i64, c_i64]),
    "fd_voxelize": (c_int, [c_void_p, c_i64, c_void_p, c_int, c_void_p, c_void_p, c_int, c_i64, c_int, c_void_p, c_void_p, c_int,
                            c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "fd_index_num_cols": (c_i64, [c_int, c_int, c_int]),
    "fd_index_workspace_bytes": (c_size_t, [c_i64]),
    "fd_index_mark": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "fd_index_downsample": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fd_index_scan": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "fd_index_coords": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "fd_index_lookup": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_i64, c_void_p, c_void_p]),
    "fd_rows_permute": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_i64, c_void_p, c_int, c_int, c_void_p]),
    "fd_rulebook": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_i64, c_void_p, c_i64, c_int, c_void_p, c_void_p,
                            c_void_p, c_void_p, c_void_p]),
    "fd_spconv_packed_weight_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "fd_spconv_pack_weight": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "fd_spconv_apply": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_i64, c_void_p, c_int, c_int, c_i64,
                                c_void_p, c_i64, c_int, c_int, c_int, c_void_p, c_void_p]),
    "fd_spconv_num_ranges": (c_int, [c_i64, c_int, c_int, c_int]),
    "fd_spconv_wants_balanced_ranges": (c_int, [c_int, c_int, c_int]),
    "fd_spconv_ranges_workspace_bytes": (c_size_t, [c_i64]),
    "fd_spconv_ranges": (c_int, [c_void_p, c_i64, c_int, c_i64, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "fd_densify": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_i64,
                           c_i64, c_i64, c_i64, c_i64, c_void_p]),
    "fd_conv2d_packed_weight_bytes": (c_size_t, [c_int, c_int, c_int]),
    "fd_conv2d_pack_weight": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "fd_conv2d_nhwc_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                    c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fd_conv2d_f32_packed_weight_bytes": (c_size_t, [c_int, c_int, c_int]),
    "fd_conv2d_f32_pack_weight": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "fd_conv2d_f32_num_tiles": (c_int, []),
    "fd_conv2d_nhwc_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                   c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fd_conv2d_shuffle_nhwc_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int,
                                           c_void_p]),
    "fd_conv2d_grouped_nhwc_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
                                           c_int, c_void_p]),
    "fd_conv2d_wino_f32_num_tiles": (c_int, []),
    "fd_conv2d_wino_f32_packed_weight_bytes": (c_size_t, [c_int, c_int]),
    "fd_conv2d_wino_f32_pack_weight": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "fd_conv2d_wino_nhwc_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int,
                                        c_void_p]),
    "fd_decode_workspace_bytes": (c_size_t, [c_int, ctypes.POINTER(DecodeCfg)]),
    "fd_centerpoint_decode": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64,
                                      c_int, ctypes.POINTER(DecodeCfg), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_size_t, c_void_p]),
    "fd_centerpoint_decode_maps": (c_int, [ctypes.POINTER(MapView)] * 5 + [c_int, ctypes.POINTER(DecodeCfg), c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_size_t, c_void_p]),
    "fd_centerpoint_decode_packed": (c_int, [ctypes.POINTER(MapView)] * 6 + [c_int, c_int, ctypes.POINTER(DecodeCfg), c_int, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "fd_assemble_detections": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(MapView), c_int, c_int, c_int, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p]),
    "fd_nms_workspace_bytes": (c_size_t, [c_int]),
    "fd_rotated_nms": (c_int, [c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "fd_boxes_iou_bev": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "fd_pillar_encode": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_float, c_float, c_float,
                                 c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                 c_void_p, c_int, c_void_p]),
    "fd_pillar_scatter": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_void_p, c_int,
                                  c_i64, c_i64, c_i64, c_i64, c_int, c_void_p]),
    "fd_forecast_chains": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_double, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fd_det_to_global_boxes": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fd_forecast_groups": (c_int, [c_void_p, c_int, ctypes.c_double, c_void_p, c_void_p]),
    "fd_forecast_from_detections": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, ctypes.c_double, ctypes.c_double,
                                            ctypes.POINTER(ForecastBuffers), c_void_p]),
    "fd_nearest_rows": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "fd_index_pyramid": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, ctypes.POINTER(IndexLevel), c_void_p, c_void_p, c_size_t,
                                 c_void_p]),
    "fd_index_pyramid_coords": (c_int, [c_int, c_int, ctypes.POINTER(IndexLevel), c_void_p]),
    "fd_rows_place": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_i64, c_void_p, c_int, c_void_p,
                              c_int, c_int, c_void_p]),
    "fd_sweep_assemble_workspace_bytes": (c_size_t, [c_i64]),
    "fd_sweep_assemble": (c_int, [c_void_p, c_int, c_int, c_i64, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                  c_size_t, c_void_p]),
}


class FutureDetHipError(RuntimeError):
    pass


def load():
    """Loads the HIP library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise FutureDetHipError(
                "libfuturedet_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `python futuredet_amd/build.py`; there is no CPU fallback for this path." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError = ABI mismatch, let it propagate
            fn.restype = res
            fn.argtypes = args
        if L.fd_abi_version() != ABI_VERSION:  # a stale build would read the structs below with another layout
            raise FutureDetHipError("%s reports ABI %d, these bindings are written for %d: rebuild it (python futuredet_amd/build.py --force)"
                                    % (LIB_PATH, L.fd_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


def check(status, what):
    if status != 0:
        msg = load().fd_last_error()
        raise FutureDetHipError("%s failed (%d): %s" % (what, status, msg.decode() if msg else "?"))
