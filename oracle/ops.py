"""ctypes bindings to oracle/libfd_oracle.so (+ the compiled reference IoU in oracle/_ref).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)


def _ptr(a, t):
    return a.ctypes.data_as(t)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libfd_oracle.so")
        src = os.path.join(_HERE, "fd_oracle.c")
        if not os.path.isfile(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-B", "libfd_oracle.so"])
        L = ctypes.CDLL(so)
        L.fdo_points_to_voxel.restype = ctypes.c_int64
        L.fdo_points_to_voxel.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, _f32p, _f32p, ctypes.c_int,
                                          ctypes.c_int64, _f32p, _i32p, _i32p]
        L.fdo_conv_out_shape.restype = None
        L.fdo_conv_out_shape.argtypes = [_i32p] * 5
        L.fdo_rulebook.restype = ctypes.c_int64
        L.fdo_rulebook.argtypes = [_i32p, ctypes.c_int64, _i32p, _i32p, _i32p, _i32p, ctypes.c_int,
                                   _i32p, _i32p, _i32p]
        L.fdo_indice_conv.restype = None
        L.fdo_indice_conv.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, _f32p, _f32p, _i32p, _i32p,
                                      ctypes.c_int, _f32p, ctypes.c_int64, ctypes.c_int]
        L.fdo_dense.restype = None
        L.fdo_dense.argtypes = [_f32p, _i32p, ctypes.c_int64, ctypes.c_int] + [ctypes.c_int] * 4 + [_f32p]
        L.fdo_iou_bev.restype = ctypes.c_float
        L.fdo_iou_bev.argtypes = [_f32p, _f32p]
        L.fdo_boxes_iou_bev.restype = None
        L.fdo_boxes_iou_bev.argtypes = [_f32p, ctypes.c_int, _f32p, ctypes.c_int, _f32p]
        L.fdo_set_threads.restype = None
        L.fdo_set_threads.argtypes = [ctypes.c_int]
        L.fdo_nms.restype = ctypes.c_int
        L.fdo_nms.argtypes = [_f32p, ctypes.c_int, ctypes.c_float, _i64p]
        L.fdo_assemble_sweeps.restype = ctypes.c_int64
        L.fdo_assemble_sweeps.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, _i64p, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                          _i32p, ctypes.POINTER(ctypes.c_double), ctypes.c_float, _f32p, ctypes.c_int]
        _LIB = L
    return _LIB


def ref_iou_lib():
    """The reference's own iou3d_cpu.cpp compiled into oracle/_ref (None if not built)."""
    global _REF
    if _REF is None:
        so = os.path.join(_HERE, "_ref", "libfd_ref_iou.so")
        if not os.path.isfile(so):
            return None
        import torch  # noqa: F401  (loads libtorch before the dlopen)

        R = ctypes.CDLL(so)
        R.fdref_boxes_iou_bev.restype = ctypes.c_int
        R.fdref_boxes_iou_bev.argtypes = [_f32p, ctypes.c_int, _f32p, ctypes.c_int, _f32p]
        _REF = R
    return _REF


def set_threads(n):
    lib().fdo_set_threads(int(n))


# --------------------------------------------------------------------------------------
def points_to_voxel(points, voxel_size, coors_range, max_points=35, reverse_index=True, max_voxels=20000):
    """Same signature / returns as det3d/ops/point_cloud/point_cloud_ops.py:112-184."""
    assert reverse_index, "only the reverse_index=True branch is on the path (voxel_generator.py:28)"
    points = np.ascontiguousarray(points, dtype=np.float32)
    vs = np.ascontiguousarray(voxel_size, dtype=np.float32)
    rg = np.ascontiguousarray(coors_range, dtype=np.float32)
    n, nd = points.shape
    voxels = np.zeros((max_voxels, max_points, nd), np.float32)
    coors = np.zeros((max_voxels, 3), np.int32)
    num = np.zeros((max_voxels,), np.int32)
    m = lib().fdo_points_to_voxel(_ptr(points, _f32p), n, nd, _ptr(vs, _f32p), _ptr(rg, _f32p), int(max_points),
                                  int(max_voxels), _ptr(voxels, _f32p), _ptr(coors, _i32p), _ptr(num, _i32p))
    assert m >= 0
    return voxels[:m], coors[:m], num[:m]


def conv_out_shape(in_shape, ksize, stride, pad):
    a = [np.ascontiguousarray(x, np.int32) for x in (in_shape, ksize, stride, pad)]
    out = np.zeros(3, np.int32)
    lib().fdo_conv_out_shape(*[_ptr(x, _i32p) for x in a], _ptr(out, _i32p))
    return out


def rulebook(indices, in_shape, ksize, stride, pad, subm):
    """-> (out_indices [n_out,4], pairs [K,2,n], pair_num [K], out_shape[3])"""
    indices = np.ascontiguousarray(indices, np.int32)
    n = indices.shape[0]
    ks = np.ascontiguousarray(ksize, np.int32)
    st = np.ascontiguousarray(stride, np.int32)
    pd = np.ascontiguousarray(pad, np.int32)
    shp = np.ascontiguousarray(in_shape, np.int32)
    K = int(np.prod(ks))
    out_idx = np.zeros((max(n, 1) * (1 if subm else K), 4), np.int32)
    pairs = np.empty((K, 2, max(n, 1)), np.int32)
    pnum = np.zeros(K, np.int32)
    n_out = lib().fdo_rulebook(_ptr(indices, _i32p), n, _ptr(shp, _i32p), _ptr(ks, _i32p), _ptr(st, _i32p),
                               _ptr(pd, _i32p), int(bool(subm)), _ptr(out_idx, _i32p), _ptr(pairs, _i32p),
                               _ptr(pnum, _i32p))
    assert n_out >= 0
    if subm:
        out_shape = shp.copy()
    else:
        out_shape = conv_out_shape(shp, ks, st, pd)
    return out_idx[:n_out].copy(), pairs, pnum, out_shape


def indice_conv(feats, weight, bias, pairs, pair_num, n_out):
    feats = np.ascontiguousarray(feats, np.float32)
    n_in, cin = feats.shape
    w = np.ascontiguousarray(weight, np.float32).reshape(-1, cin, weight.shape[-1])
    K, _, cout = w.shape
    assert pairs.shape[0] == K and pairs.shape[2] == max(n_in, 1)
    out = np.empty((n_out, cout), np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    lib().fdo_indice_conv(_ptr(feats, _f32p), n_in, cin, _ptr(w, _f32p),
                          None if b is None else _ptr(b, _f32p), _ptr(pairs, _i32p), _ptr(pair_num, _i32p), K,
                          _ptr(out, _f32p), n_out, cout)
    return out


def dense(feats, indices, batch_size, spatial_shape):
    feats = np.ascontiguousarray(feats, np.float32)
    indices = np.ascontiguousarray(indices, np.int32)
    n, c = feats.shape
    D, H, W = [int(v) for v in spatial_shape]
    out = np.empty((batch_size, c, D, H, W), np.float32)
    lib().fdo_dense(_ptr(feats, _f32p), _ptr(indices, _i32p), n, c, batch_size, D, H, W, _ptr(out, _f32p))
    return out


def boxes_iou_bev(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    lib().fdo_boxes_iou_bev(_ptr(a, _f32p), a.shape[0], _ptr(b, _f32p), b.shape[0], _ptr(out, _f32p))
    return out


def ref_boxes_iou_bev(a, b):
    R = ref_iou_lib()
    if R is None:
        return None
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    R.fdref_boxes_iou_bev(_ptr(a, _f32p), a.shape[0], _ptr(b, _f32p), b.shape[0], _ptr(out, _f32p))
    return out


def nms(boxes, thresh):
    """boxes [n,7] already score-sorted; returns kept indices (int64), like nms_gpu's keep[:num]."""
    boxes = np.ascontiguousarray(boxes, np.float32)
    n = boxes.shape[0]
    keep = np.zeros(max(n, 1), np.int64)
    k = lib().fdo_nms(_ptr(boxes, _f32p), n, ctypes.c_float(thresh), _ptr(keep, _i64p))
    return keep[:k].copy()


def assemble_sweeps(key_raw, sweep_raws, transforms, time_lags, nsweeps=None, num_point_feature=4, min_distance=1.0):
    """LoadPointCloudFromFile (NuScenes branch, loading.py:107-141) on in-memory file contents: key_raw [n,5] and
    sweep_raws[i] [n_i,5] float32 are what np.fromfile(...).reshape(-1,5) returns for info["lidar_path"] and
    info["sweeps"][i]["lidar_path"]; transforms[i] is a 4x4 or None; time_lags[i] a float.  -> combined [N,5] f32."""
    nsweeps = len(sweep_raws) + 1 if nsweeps is None else nsweeps
    assert nsweeps - 1 == len(sweep_raws)
    order = np.random.default_rng(0).choice(len(sweep_raws), nsweeps - 1, replace=False) if nsweeps > 1 else []  # :128-129
    chunks = [np.asarray(key_raw, np.float32)] + [np.asarray(sweep_raws[i], np.float32) for i in order]
    S = len(chunks)
    mats = np.zeros((S, 16), np.float64)
    flags = np.zeros((S,), np.int32)
    lags = np.zeros((S,), np.float64)
    for j, i in enumerate(order, start=1):
        flags[j] = 2
        if transforms[i] is not None:
            mats[j] = np.asarray(transforms[i], np.float64).reshape(16)
            flags[j] |= 1
        lags[j] = float(time_lags[i])
    raw = np.ascontiguousarray(np.concatenate(chunks), np.float32)
    rows = np.cumsum([0] + [len(c) for c in chunks]).astype(np.int64)
    out = np.empty((max(len(raw), 1), num_point_feature + 1), np.float32)
    n = lib().fdo_assemble_sweeps(_ptr(raw, _f32p), raw.shape[1], num_point_feature, _ptr(rows, _i64p), S,
                                  _ptr(mats, ctypes.POINTER(ctypes.c_double)), _ptr(flags, _i32p),
                                  _ptr(lags, ctypes.POINTER(ctypes.c_double)), ctypes.c_float(min_distance), _ptr(out, _f32p), 1)
    return out[:n].copy()
