"""Torch-tensor front end of the C ABI (include/futuredet_hip.h).  Tensors are device memory plumbing only:
every function passes raw pointers + the current HIP stream to libfuturedet_hip.so.  No CPU fallbacks."""
import ctypes
import threading

import numpy as np
import torch

from . import lib as _lib
from .lib import DecodeCfg, FutureDetHipError, MapView, check


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream on the current device (the raw getter is ~20x cheaper than building a
    torch.cuda.Stream object; this is called once per library launch)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, name, dtype=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise FutureDetHipError("%s must be a tensor on the HIP device (got %s); this path has no CPU implementation"
                                % (name, "cpu tensor" if isinstance(t, torch.Tensor) else type(t)))
    if dtype is not None and t.dtype != dtype:
        raise FutureDetHipError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise FutureDetHipError("%s must be contiguous" % name)
    return t


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class _Workspace(object):
    """Grow-only per-(tag, device, owner) scratch buffers so steady-state steps do no allocation.  The owner is the current
    stream -- sweeps in flight on different streams (bench.py --inflight, serving) must not share scratch memory -- or, inside
    ``with workspace.scope(token)``, the token: a captured whole-sweep graph keeps scratch of its own whatever stream it was
    captured on (torch hands out streams from a pool, so two captures can meet on one stream handle)."""

    def __init__(self):
        self._bufs = {}
        self._scope = threading.local()

    def scope(self, token):
        ws = self

        class _Scope(object):
            def __enter__(self_):
                self_.prev = getattr(ws._scope, "token", None)
                ws._scope.token = token

            def __exit__(self_, *exc):
                ws._scope.token = self_.prev

        return _Scope()

    def release(self, token):
        """Drops the buffers of a scope (the owner of the token is going away)."""
        for key in [k for k in self._bufs if k[2] == ("scope", token)]:
            del self._bufs[key]

    def get(self, tag, nbytes, device):
        token = getattr(self._scope, "token", None)
        key = (tag, device.index, ("scope", token) if token is not None else _stream().value)
        buf = self._bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
            self._bufs[key] = buf
        return buf


workspace = _Workspace()
_DT = {torch.float32: 0, torch.bfloat16: 1}


def set_tuning(name, value):
    """fd_tuning_set: kernel-variant knobs for tuning runs and for the tests that compare variants (0 = heuristic)."""
    check(_lib.load().fd_tuning_set(name.encode(), int(value)), "fd_tuning_set")


# ------------------------------------------------------------------------------------------------ voxelizer
def voxelize(points, voxel_size, coors_range, max_points, max_voxels, batch_idx=0, want_voxels=True, want_mean=False,
             mean_stride=None, coor_cols=3, out=None, n_points_dev=None):
    """Runs fd_voxelize on device points [N, ndim] float32.  Returns a dict of capacity-sized device tensors
    (voxels / mean / coors / num_points) plus ``num_voxels`` (device int32[1]); nothing is synchronised.
    ``out`` may supply pre-allocated (sliced) destination tensors with the same keys.  ``n_points_dev`` (device int32[1]):
    only the first min(N, n_points_dev[0]) rows are points, the rest is padding of a fixed-capacity buffer."""
    L = _lib.load()
    points = _dev(points, "points", torch.float32)
    n, ndim = points.shape
    dev = points.device
    rng = (ctypes.c_float * 6)(*[float(np.float32(v)) for v in coors_range])
    vs = (ctypes.c_float * 3)(*[float(np.float32(v)) for v in voxel_size])
    out = dict(out or {})
    if want_voxels and "voxels" not in out:
        out["voxels"] = torch.empty((max_voxels, max_points, ndim), dtype=torch.float32, device=dev)
    if want_mean and "mean" not in out:
        out["mean"] = torch.empty((max_voxels, mean_stride or ndim), dtype=torch.float32, device=dev)
    if "coors" not in out:
        out["coors"] = torch.empty((max_voxels, coor_cols), dtype=torch.int32, device=dev)
    if "num_points" not in out:
        out["num_points"] = torch.empty((max_voxels,), dtype=torch.int32, device=dev)
    if "num_voxels" not in out:
        out["num_voxels"] = torch.zeros((1,), dtype=torch.int32, device=dev)
    mean = out.get("mean") if want_mean else None
    voxels = out.get("voxels") if want_voxels else None
    ws_bytes = L.fd_voxelize_workspace_bytes(n, max_voxels)
    ws = workspace.get("voxelize", ws_bytes, dev)
    check(L.fd_voxelize(_p(points), n, _p(n_points_dev), ndim, rng, vs, int(max_points), int(max_voxels), int(batch_idx), _p(voxels), _p(mean),
                        int(mean.shape[1]) if mean is not None else 0, _p(out["coors"]), int(out["coors"].shape[1]),
                        _p(out["num_points"]), _p(out["num_voxels"]), _p(ws), ws.numel(), _stream()), "fd_voxelize")
    return out


# ------------------------------------------------------------------------------------------------ sparse index
class SparseIndex(object):
    """Active set on a (B, D, H, W) grid: column occupancy words + prefix counts (+ coords once counted)."""

    def __init__(self, B, D, H, W, device, words=None, prefix=None):
        L = _lib.load()
        self.B, self.D, self.H, self.W = int(B), int(D), int(H), int(W)
        self.ncols = L.fd_index_num_cols(self.B, self.H, self.W)
        self.words = torch.zeros((self.ncols,), dtype=torch.int64, device=device) if words is None else words
        self.prefix = torch.empty((self.ncols,), dtype=torch.int32, device=device) if prefix is None else prefix
        self.n_dev = None     # device int32[1] view
        self.n = None         # host int, set by finalize() (static indexes: the row CAPACITY)
        self.coords = None    # [n,4] int32 (b,z,y,x), rows in index order
        self.device = device
        self.static = False   # True: the count stays on the device (n_dev); n is a capacity, n_expected a typical count
        self.n_expected = 0

    @property
    def spatial_shape(self):
        return [self.D, self.H, self.W]

    def mark(self, coords, n_dev=None, n_max=None):
        L = _lib.load()
        coords = _dev(coords, "coords", torch.int32)
        n_max = coords.shape[0] if n_max is None else n_max
        check(L.fd_index_mark(_p(coords), _p(n_dev), n_max, self.B, self.D, self.H, self.W, _p(self.words), _stream()),
              "fd_index_mark")

    def scan(self, n_dev):
        L = _lib.load()
        ws = workspace.get("index_scan", L.fd_index_workspace_bytes(self.ncols), self.device)
        self.n_dev = n_dev
        check(L.fd_index_scan(_p(self.words), self.ncols, _p(self.prefix), _p(n_dev), _p(ws), ws.numel(), _stream()),
              "fd_index_scan")

    def downsample(self, ksize, stride, pad):
        L = _lib.load()
        od = [(i + 2 * p - (k - 1) - 1) // s + 1 for i, k, s, p in zip(self.spatial_shape, ksize, stride, pad)]
        out = SparseIndex(self.B, od[0], od[1], od[2], self.device)
        check(L.fd_index_downsample(_p(self.words), self.B, self.D, self.H, self.W, (ctypes.c_int * 3)(*ksize),
                                    (ctypes.c_int * 3)(*stride), (ctypes.c_int * 3)(*pad), _p(out.words), _stream()),
              "fd_index_downsample")
        return out

    def finalize(self, n):
        """Called once the host knows the active count: materialises coords."""
        L = _lib.load()
        self.n = int(n)
        self.coords = torch.empty((max(self.n, 1), 4), dtype=torch.int32, device=self.device)[: self.n]
        if self.n:
            check(L.fd_index_coords(_p(self.words), _p(self.prefix), self.B, self.D, self.H, self.W, _p(self.coords),
                                    _stream()), "fd_index_coords")

    def lookup(self, coords, n_dev=None):
        L = _lib.load()
        coords = _dev(coords, "coords", torch.int32)
        row_of = torch.empty((coords.shape[0],), dtype=torch.int32, device=self.device)
        check(L.fd_index_lookup(_p(self.words), _p(self.prefix), self.B, self.D, self.H, self.W, _p(coords), _p(n_dev),
                                coords.shape[0], _p(row_of), _stream()), "fd_index_lookup")
        return row_of

    def rulebook(self, out_index, ksize, stride, pad):
        """nbr [K, stride64] int32: input row (in self) feeding each output row of ``out_index`` per tap."""
        L = _lib.load()
        assert out_index.n is not None
        K = int(ksize[0] * ksize[1] * ksize[2])
        nstride = max(64, (out_index.n + 63) // 64 * 64)
        nbr = torch.empty((K, nstride), dtype=torch.int32, device=self.device)
        static = getattr(out_index, "static", False)
        if out_index.n == 0:
            nbr.fill_(-1)
            return nbr
        # static indexes: the table has the level's row capacity; only the device's count of rows is written (and read)
        check(L.fd_rulebook(_p(self.words), _p(self.prefix), self.B, self.D, self.H, self.W, _p(out_index.coords), out_index.n,
                            _p(out_index.n_dev), nstride, 0 if static else 1, (ctypes.c_int * 3)(*ksize), (ctypes.c_int * 3)(*stride),
                            (ctypes.c_int * 3)(*pad), _p(nbr), _stream()), "fd_rulebook")
        nbr.n_out = out_index.n
        nbr.n_dev = out_index.n_dev if static else None
        nbr.n_expected = out_index.n_expected if static else 0
        return nbr


class _PyramidPlan(object):
    """Static part of build_pyramid for one (B, shape, geometry): level shapes, one words / prefix allocation for all
    levels, the fd_index_level array.  Reused across sweeps so a step pays one memset and two library calls."""

    def __init__(self, B, shape0, geoms, device):
        L = _lib.load()
        self.B, self.geoms, self.device = int(B), geoms, device
        self.shapes = [tuple(int(v) for v in shape0)]
        for ks, st, pd in geoms:
            self.shapes.append(tuple((i + 2 * p - (k - 1) - 1) // s + 1 for i, k, s, p in zip(self.shapes[-1], ks, st, pd)))
        self.ncols = [L.fd_index_num_cols(self.B, sh[1], sh[2]) for sh in self.shapes]
        self.ws_bytes = L.fd_index_workspace_bytes(sum(self.ncols) + 2048 * len(self.ncols))  # block sums of all levels at once

    def row_caps(self, cap0):
        """Upper bounds of the active rows per level given at most ``cap0`` rows on level 0: a strided convolution turns one
        input into at most prod(ceil(k / s)) outputs (the taps k with (p + pad - k) % s == 0), and a level cannot have more
        rows than cells."""
        caps = [min(int(cap0), self.B * int(np.prod(self.shapes[0])))]
        for (ks, st, pd), sh in zip(self.geoms, self.shapes[1:]):
            fan = int(np.prod([-(-k // s) for k, s in zip(ks, st)]))
            caps.append(min(caps[-1] * fan, self.B * int(np.prod(sh))))
        return caps


_pyramid_plans = {}


def build_pyramid(coors, nvox, n_max, B, shape0, geoms, device, static=False, expected=None, row_caps=None):
    """All sparse indexes of the backbone with two library calls and ONE host read: level 0 (D,H,W = shape0) marked
    from the voxelizer output ``coors`` [B*n_max, 4] / ``nvox`` [B]; level l from level l-1 by geoms[l-1] =
    (ksize, stride, pad).  Returns the list of finalised SparseIndex (coords materialised).
    ``static=True``: NO host read -- every level gets its row capacity (_PyramidPlan.row_caps) as ``n``, the counts stay in
    ``n_dev`` and ``expected`` (typical counts, e.g. of an earlier sweep) only steers launch heuristics.  ``row_caps`` (one
    per level) replaces the data-free capacities by smaller ones (a high-water mark with head room): every kernel then works on
    min(capacity, count) rows, and a sweep whose count exceeds a capacity is DETECTABLE -- counts[l] > capacity -- but not
    computed correctly; the caller re-runs it on the eager path (detectors.StaticStep does)."""
    L = _lib.load()
    key = (int(B), tuple(int(v) for v in shape0), tuple((tuple(k), tuple(s), tuple(p)) for k, s, p in geoms), str(device))
    plan = _pyramid_plans.get(key)
    if plan is None:
        plan = _pyramid_plans[key] = _PyramidPlan(B, shape0, geoms, device)
    shapes, ncols = plan.shapes, plan.ncols
    # fresh buffers every sweep (the indexes are handed out and may outlive the call); fd_index_pyramid clears level 0 itself and
    # overwrites the other levels
    words_all = torch.empty((sum(ncols),), dtype=torch.int64, device=device)
    prefix_all = torch.empty((sum(ncols),), dtype=torch.int32, device=device)
    counts = torch.empty((len(shapes),), dtype=torch.int32, device=device)
    idx, off = [], 0
    levels = (_lib.IndexLevel * len(shapes))()
    wp, pp = words_all.data_ptr(), prefix_all.data_ptr()
    for l, (sh, nc) in enumerate(zip(shapes, ncols)):
        ix = SparseIndex.__new__(SparseIndex)
        ix.B, ix.D, ix.H, ix.W, ix.ncols, ix.device = plan.B, sh[0], sh[1], sh[2], nc, device
        ix.words, ix.prefix = words_all[off:off + nc], prefix_all[off:off + nc]
        ix.n_dev, ix.n, ix.coords = counts[l:l + 1], None, None
        ix.level_counts = counts  # all levels' counts in one tensor (n_dev is the slice of this level)
        ix.static, ix.n_expected = False, 0
        lv = levels[l]
        lv.D, lv.H, lv.W = sh
        if l:
            ks, st, pd = geoms[l - 1]
            lv.ksize[:], lv.stride[:], lv.pad[:] = list(ks), list(st), list(pd)
        lv.words, lv.prefix, lv.coords = wp + 8 * off, pp + 4 * off, None
        idx.append(ix)
        off += nc
    ws = workspace.get("index_scan", plan.ws_bytes, device)

    def size_coords(host):
        total = sum(host)
        coords_all = torch.empty((max(total, 1), 4), dtype=torch.int32, device=device)
        o = 0
        for l, ix in enumerate(idx):
            ix.n = int(host[l])
            ix.coords = coords_all[o:o + ix.n]
            levels[l].coords = (coords_all.data_ptr() + 16 * o) if ix.n else None
            levels[l].coords_rows = ix.n
            o += ix.n

    if static:
        host = plan.row_caps(int(B) * int(n_max))
        if row_caps is not None:
            host = [min(int(a), int(b)) for a, b in zip(host, row_caps)]
        for l, ix in enumerate(idx):
            ix.static = True
            ix.n_expected = int(expected[l]) if expected is not None else 0
        # capacities are known before the scan: the coordinate tables go in with the levels and the scan writes them in its last pass
        size_coords(host)
        fused = all(int(h) > 0 for h in host)
        check(L.fd_index_pyramid(_p(coors), _p(nvox), int(n_max), plan.B, len(shapes), levels, _p(counts), _p(ws), ws.numel(), _stream()),
              "fd_index_pyramid")
        if not fused:
            check(L.fd_index_pyramid_coords(plan.B, len(shapes), levels, _stream()), "fd_index_pyramid_coords")
        return idx
    check(L.fd_index_pyramid(_p(coors), _p(nvox), int(n_max), plan.B, len(shapes), levels, _p(counts), _p(ws), ws.numel(), _stream()),
          "fd_index_pyramid")
    size_coords(counts.tolist())  # the only synchronisation of the backbone
    check(L.fd_index_pyramid_coords(plan.B, len(shapes), levels, _stream()), "fd_index_pyramid_coords")
    return idx


def ranges_for(nbr, cin, cout):
    """Work-balanced row ranges of a rulebook for the fp32 kernel (fd_spconv_ranges); computed once per rulebook and
    shared by every convolution that reuses it.  Returns (ranges int32[n+1], n) or (None, n) for an equal-rows split."""
    cached = getattr(nbr, "ranges", None)
    if cached is not None:
        return cached
    L = _lib.load()
    n_out = getattr(nbr, "n_out", None)
    if not n_out:
        return None, 0
    K, nstride = nbr.shape
    n = int(L.fd_spconv_num_ranges(getattr(nbr, "n_expected", 0) or n_out, cin, cout, 0))
    if n <= 0:
        return None, 0
    ranges = torch.empty((n + 1,), dtype=torch.int32, device=nbr.device)
    ws = workspace.get("spconv_ranges", L.fd_spconv_ranges_workspace_bytes(n_out), nbr.device)
    check(L.fd_spconv_ranges(_p(nbr), nstride, K, n_out, _p(getattr(nbr, "n_dev", None)), n, _p(ranges), _p(ws), ws.numel(), _stream()),
          "fd_spconv_ranges")
    nbr.ranges = (ranges, n)
    return nbr.ranges


def rows_permute(src, row_of, c_dst, dtype=torch.float32, n_rows=None, n_dev=None):
    L = _lib.load()
    src = _dev(src, "src", torch.float32)
    row_of = _dev(row_of, "row_of", torch.int32)
    n_rows = src.shape[0] if n_rows is None else n_rows
    dst = torch.zeros((max(n_rows, 1), c_dst), dtype=dtype, device=src.device)[:n_rows]
    if src.shape[0]:
        check(L.fd_rows_permute(_p(src), src.shape[1], _p(row_of), _p(n_dev), src.shape[0], _p(dst), c_dst,
                                _DT[dtype], _stream()), "fd_rows_permute")
    return dst


# ------------------------------------------------------------------------------------------------ sparse conv
def pack_spconv_weight(w_kio, dtype=torch.float32):
    """[K, Cin, Cout] float32 (host or device) -> fragment-ordered packed weights on the same device."""
    L = _lib.load()
    dev = w_kio.device
    w = w_kio.detach().to("cpu", torch.float32).contiguous()
    K, cin, cout = w.shape
    nbytes = L.fd_spconv_packed_weight_bytes(K, cin, cout, _DT[dtype])
    host = torch.empty((nbytes,), dtype=torch.uint8)
    check(L.fd_spconv_pack_weight(ctypes.c_void_p(w.data_ptr()), K, cin, cout, _DT[dtype], ctypes.c_void_p(host.data_ptr())),
          "fd_spconv_pack_weight")
    return host.to(dev)


def spconv_apply(feats, wpacked, bias, nbr, n_out, cout, residual=None, relu=False, out=None, balanced=None):
    L = _lib.load()
    feats = _dev(feats, "feats")
    dt = _DT[feats.dtype]
    cin = feats.shape[1]
    out_dtype, out_cols = feats.dtype, cout
    K, nstride = nbr.shape
    if out is None:
        out = torch.empty((max(n_out, 1), out_cols), dtype=out_dtype, device=feats.device)[:n_out]
    if residual is not None:
        _dev(residual, "residual", out_dtype)
    # fp32: the kernel walks row ranges.  The MFMA-bound SubM layers (Cout >= 64; their rulebook is shared by the 4-5
    # convolutions of a level) get equal-WORK ranges from fd_spconv_ranges; narrow layers and strided convolutions (rulebook
    # used once) take equal row counts -- see fd_spconv_num_ranges for the measurements.
    ranges, n_ranges = None, 0
    n_dev, n_expected = getattr(nbr, "n_dev", None), getattr(nbr, "n_expected", 0)  # static index: n_out is a capacity
    if dt == 0 and n_out > 0:
        if balanced is None:
            balanced = K == 27 and cin == cout and bool(L.fd_spconv_wants_balanced_ranges(cin, cout, 0))
        if balanced == "tiles":      # one 128-row tile per workgroup (tuning comparisons)
            n_ranges = 0
        elif balanced:
            ranges, n_ranges = ranges_for(nbr, cin, cout)
        else:
            n_ranges = int(L.fd_spconv_num_ranges(n_expected or n_out, cin, cout, 0))
    check(L.fd_spconv_apply(_p(feats), feats.shape[0], _p(_dev(wpacked, "wpacked")), _p(bias), _p(residual), int(bool(relu)),
                            _p(_dev(nbr, "nbr", torch.int32)), nstride, _p(ranges), n_ranges, K, n_out, _p(n_dev), int(n_expected or 0),
                            cin, cout, dt, _p(out), _stream()),
          "fd_spconv_apply")
    return out


def densify(feats, index, out_dtype=None, channels_last=False, out=None):
    """SparseConvTensor.dense()+view: [B, C*D, H, W] with channel = c*D + d (``out``: pre-allocated destination)."""
    L = _lib.load()
    feats = _dev(feats, "feats")
    C = feats.shape[1]
    shape = (index.B, C * index.D, index.H, index.W)
    if out is None:
        out_dtype = out_dtype or feats.dtype
        out = torch.empty(shape, dtype=out_dtype, device=feats.device,
                          memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    else:
        assert tuple(out.shape) == shape and out.is_cuda
        out_dtype = out.dtype
    sb, sc, sy, sx = out.stride()
    check(L.fd_densify(_p(feats), C, _DT[feats.dtype], _p(index.words), _p(index.prefix), index.B, index.D, index.H,
                       index.W, _p(out), _DT[out_dtype], sb, sc, sy, sx, int(feats.shape[0]), _stream()), "fd_densify")
    return out


# ------------------------------------------------------------------------------------------------ dense conv (bf16)
def pack_conv2d_weight(w_oihw):
    """[Cout, Cin, k, k] float32 -> MFMA-fragment-ordered bf16 weights on the same device."""
    L = _lib.load()
    dev = w_oihw.device
    w = w_oihw.detach().to("cpu", torch.float32).contiguous()
    cout, cin, ks, _ = w.shape
    nbytes = L.fd_conv2d_packed_weight_bytes(cout, cin, ks)
    if nbytes == 0:
        raise FutureDetHipError("fd_conv2d: unsupported weight shape %s" % (tuple(w.shape),))
    host = torch.empty((nbytes,), dtype=torch.uint8)
    check(L.fd_conv2d_pack_weight(ctypes.c_void_p(w.data_ptr()), cout, cin, ks, ctypes.c_void_p(host.data_ptr())),
          "fd_conv2d_pack_weight")
    return host.to(dev)


def conv2d_nhwc_bf16(x, wpk, bias, cout, ks, stride=1, relu=True, out=None, co_off=0, osy=1, osx=1, ooy=0, oox=0):
    """x [B,H,W,Cin] bf16 contiguous -> y [B,Ho*osy,Wo*osx,Ctot] bf16 (allocated when ``out`` is None)."""
    L = _lib.load()
    x = _dev(x, "x", torch.bfloat16)
    B, H, W, cin = x.shape
    pad = 1 if ks == 3 else 0
    Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    if out is None:
        out = torch.empty((B, Ho * osy, Wo * osx, cout), dtype=torch.bfloat16, device=x.device)
    _dev(out, "out", torch.bfloat16)
    check(L.fd_conv2d_nhwc_bf16(_p(x), B, H, W, cin, _p(wpk), _p(bias), cout, ks, stride, pad, int(bool(relu)), _p(out),
                                out.shape[3], co_off, osy, osx, ooy, oox, _stream()), "fd_conv2d_nhwc_bf16")
    return out


def pack_conv2d_weight_f32(w_oihw):
    """[Cout, Cin, k, k] float32 -> MFMA-fragment-ordered fp32 weights on the same device (fd_conv2d_f32_pack_weight)."""
    L = _lib.load()
    dev = w_oihw.device
    w = w_oihw.detach().to("cpu", torch.float32).contiguous()
    cout, cin, ks, _ = w.shape
    nbytes = L.fd_conv2d_f32_packed_weight_bytes(cout, cin, ks)
    if nbytes == 0:
        raise FutureDetHipError("fd_conv2d_f32: unsupported weight shape %s" % (tuple(w.shape),))
    host = torch.empty((nbytes,), dtype=torch.uint8)
    check(L.fd_conv2d_f32_pack_weight(ctypes.c_void_p(w.data_ptr()), cout, cin, ks, ctypes.c_void_p(host.data_ptr())),
          "fd_conv2d_f32_pack_weight")
    return host.to(dev)


def conv2d_nhwc_f32(x, wpk, bias, cout, ks, stride=1, relu=True, out=None, co_off=0, osy=1, osx=1, ooy=0, oox=0, tile=0):
    """x [B,H,W,Cin] float32 contiguous -> y [B,Ho*osy,Wo*osx,Ctot] float32 (allocated when ``out`` is None).
    ``tile``: 0 = the library's heuristic, 1..conv2d_f32_num_tiles() = that workgroup tile shape."""
    L = _lib.load()
    x = _dev(x, "x", torch.float32)
    B, H, W, cin = x.shape
    pad = 1 if ks == 3 else 0
    Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    if out is None:
        out = torch.empty((B, Ho * osy, Wo * osx, cout), dtype=torch.float32, device=x.device)
    _dev(out, "out", torch.float32)
    check(L.fd_conv2d_nhwc_f32(_p(x), B, H, W, cin, _p(wpk), _p(bias), cout, ks, stride, pad, int(bool(relu)), _p(out),
                               out.shape[3], co_off, osy, osx, ooy, oox, int(tile), _stream()), "fd_conv2d_nhwc_f32")
    return out


def conv2d_f32_num_tiles():
    return int(_lib.load().fd_conv2d_f32_num_tiles())


def conv2d_shuffle_nhwc_f32(x, wpk, bias, cout_sub, k, relu=True, out=None, co_off=0, tile=0):
    """ConvTranspose2d(k, stride k) as one 1x1 convolution + pixel shuffle: x [B,H,W,Cin] -> [B,H*k,W*k,Ctot] float32."""
    L = _lib.load()
    x = _dev(x, "x", torch.float32)
    B, H, W, cin = x.shape
    if out is None:
        out = torch.empty((B, H * k, W * k, cout_sub), dtype=torch.float32, device=x.device)
    _dev(out, "out", torch.float32)
    check(L.fd_conv2d_shuffle_nhwc_f32(_p(x), B, H, W, cin, _p(wpk), _p(bias), int(cout_sub), int(k), int(bool(relu)), _p(out), out.shape[3],
                                       co_off, int(tile), _stream()), "fd_conv2d_shuffle_nhwc_f32")
    return out


def conv2d_grouped_nhwc_f32(x, wpk, bias, counts, cin_g, relu=False, out=None, co_off=0, tile=0):
    """Grouped 3x3 convolution (<= 16 outputs per group): x [B,H,W,groups*cin_g] -> [B,H,W,sum(counts)] float32."""
    L = _lib.load()
    x = _dev(x, "x", torch.float32)
    B, H, W, cin = x.shape
    groups = len(counts)
    assert cin == groups * cin_g
    if out is None:
        out = torch.empty((B, H, W, int(sum(counts))), dtype=torch.float32, device=x.device)
    _dev(out, "out", torch.float32)
    check(L.fd_conv2d_grouped_nhwc_f32(_p(x), B, H, W, groups, int(cin_g), _p(wpk), _p(bias), (ctypes.c_int * groups)(*[int(c) for c in counts]),
                                       int(bool(relu)), _p(out), out.shape[3], co_off, int(tile), _stream()), "fd_conv2d_grouped_nhwc_f32")
    return out


def pack_conv2d_weight_wino(w_oihw):
    """[Cout, Cin, 3, 3] float32 -> Winograd-transformed (G g G^T), fragment-ordered fp32 weights on the same device."""
    L = _lib.load()
    dev = w_oihw.device
    w = w_oihw.detach().to("cpu", torch.float32).contiguous()
    cout, cin, ks, _ = w.shape
    nbytes = L.fd_conv2d_wino_f32_packed_weight_bytes(cout, cin) if ks == 3 else 0
    if nbytes == 0:
        raise FutureDetHipError("fd_conv2d_wino_f32: unsupported weight shape %s" % (tuple(w.shape),))
    host = torch.empty((nbytes,), dtype=torch.uint8)
    check(L.fd_conv2d_wino_f32_pack_weight(ctypes.c_void_p(w.data_ptr()), cout, cin, ctypes.c_void_p(host.data_ptr())),
          "fd_conv2d_wino_f32_pack_weight")
    return host.to(dev)


def conv2d_wino_nhwc_f32(x, wpk, bias, cout, relu=True, out=None, co_off=0, tile=0):
    """3x3 stride-1 pad-1 convolution by Winograd F(2x2,3x3) on MFMA: x [B,H,W,Cin] float32 -> [B,H,W,Ctot] float32."""
    L = _lib.load()
    x = _dev(x, "x", torch.float32)
    B, H, W, cin = x.shape
    if out is None:
        out = torch.empty((B, H, W, cout), dtype=torch.float32, device=x.device)
    _dev(out, "out", torch.float32)
    check(L.fd_conv2d_wino_nhwc_f32(_p(x), B, H, W, cin, _p(wpk), _p(bias), cout, int(bool(relu)), _p(out), out.shape[3], co_off, int(tile),
                                    _stream()), "fd_conv2d_wino_nhwc_f32")
    return out


def conv2d_wino_f32_num_tiles():
    return int(_lib.load().fd_conv2d_wino_f32_num_tiles())


# ------------------------------------------------------------------------------------------------ decode / NMS
CIRCLE_PRE_MAX = 4096  # candidates taken per group under circular NMS (the kernels' bound; the reference applies no cut there)


def make_decode_cfg(H, W, test_cfg, hm_channels=1, group_radius=None):
    """``hm_channels`` > 1: the score of a cell is the maximum over that many heat-map channels (CenterHead's ``classify``
    mode, center_head.py:589-595: torch.max(hm, dim=1) before the sigmoid).  ``group_radius``: test_cfg.circular_nms
    (center_head.py:722-725) -- one ``min_radius`` per decode group; the rotated-IoU predicate is replaced by the centre
    distance and the pre-NMS cut by the kernels' maximum (see fd_decode_cfg in include/futuredet_hip.h)."""
    c = DecodeCfg()
    c.hm_channels = int(hm_channels)
    c.H, c.W = int(H), int(W)
    c.out_size_factor = float(test_cfg["out_size_factor"])
    c.voxel_x, c.voxel_y = float(test_cfg["voxel_size"][0]), float(test_cfg["voxel_size"][1])
    c.pc_x, c.pc_y = float(test_cfg["pc_range"][0]), float(test_cfg["pc_range"][1])
    c.score_threshold = float(test_cfg["score_threshold"])
    for i, v in enumerate(test_cfg["post_center_limit_range"]):
        c.center_range[i] = float(v)
    nms = test_cfg["nms"]
    c.nms_iou_threshold = float(nms["nms_iou_threshold"])
    c.nms_pre_max = int(nms["nms_pre_max_size"])
    c.nms_post_max = int(nms["nms_post_max_size"])
    if group_radius is not None:
        if not 1 <= len(group_radius) <= 16:
            raise ValueError("circular NMS: 1..16 decode groups, got %d radii" % len(group_radius))
        c.nms_kind, c.n_radius = 1, len(group_radius)
        c.nms_pre_max = CIRCLE_PRE_MAX
        for i, r in enumerate(group_radius):
            c.circle_radius[i] = float(r)
    return c


def centerpoint_decode(hm, reg, height, dim, rot, cfg):
    """Inputs are [G, C, H, W] float32 NCHW slices (G groups, C = 1/2/1/3/2).  Returns (boxes7 [G,post,7],
    scores [G,post], cell [G,post] int32, count [G] int32), all on device."""
    L = _lib.load()
    G = hm.shape[0]
    ts = [_dev(t, n, torch.float32) for t, n in ((hm, "hm"), (reg, "reg"), (height, "height"), (dim, "dim"), (rot, "rot"))]
    dev = hm.device
    post = cfg.nms_post_max
    boxes = torch.empty((G, post, 7), dtype=torch.float32, device=dev)
    scores = torch.empty((G, post), dtype=torch.float32, device=dev)
    cell = torch.empty((G, post), dtype=torch.int32, device=dev)
    count = torch.empty((G,), dtype=torch.int32, device=dev)
    ws = workspace.get("decode", L.fd_decode_workspace_bytes(G, ctypes.byref(cfg)), dev)
    args = []
    for t in ts:
        args += [_p(t), t.stride(0)]
    check(L.fd_centerpoint_decode(*args, G, ctypes.byref(cfg), _p(boxes), _p(scores), _p(cell), _p(count), _p(ws),
                                  ws.numel(), _stream()), "fd_centerpoint_decode")
    return boxes, scores, cell, count


def nhwc_channel_view(buf, c0):
    """fd_map_view of channels [c0, ...) of an NHWC buffer [G, H, W, C] (float32 or bf16): the decode reads a head output in place"""
    assert buf.is_cuda and buf.is_contiguous() and buf.dim() == 4 and buf.dtype in (torch.float32, torch.bfloat16)
    G, H, W, C = buf.shape
    return MapView(buf.data_ptr() + int(c0) * buf.element_size(), H * W * C, 1, C, 0 if buf.dtype == torch.float32 else 1)


def centerpoint_decode_views(views, G, cfg, device):
    """fd_centerpoint_decode_maps on five fd_map_view (hm, reg, height, dim, rot); G = groups x samples.  Returns
    (boxes7 [G,post,7], scores [G,post], cell [G,post] int32, count [G] int32)."""
    L = _lib.load()
    post = cfg.nms_post_max
    boxes = torch.empty((G, post, 7), dtype=torch.float32, device=device)
    scores = torch.empty((G, post), dtype=torch.float32, device=device)
    cell = torch.empty((G, post), dtype=torch.int32, device=device)
    count = torch.empty((G,), dtype=torch.int32, device=device)
    ws = workspace.get("decode", L.fd_decode_workspace_bytes(G, ctypes.byref(cfg)), device)
    check(L.fd_centerpoint_decode_maps(*[ctypes.byref(v) for v in views], G, ctypes.byref(cfg), _p(boxes), _p(scores), _p(cell), _p(count), _p(ws),
                                       ws.numel(), _stream()), "fd_centerpoint_decode_maps")
    return boxes, scores, cell, count


def centerpoint_decode_packed(views, vel_view, G, B, cfg, device, step_group, step_vel_channel, step_label):
    """fd_centerpoint_decode_packed: decode + rotated NMS + assembly of the packed result in one call (five launches, the last one
    sweeps, gathers and writes the packed rows).  views = (hm, reg, height, dim, rot) fd_map_view; decode groups are group-major
    (g = group * B + sample).  -> (packed [B, S, post, 11] float32 rows x y z w l h vx vy yaw score label, counts [B, S] int32)"""
    L = _lib.load()
    post, S = cfg.nms_post_max, len(step_group)
    boxes = torch.empty((G, post, 7), dtype=torch.float32, device=device)
    scores = torch.empty((G, post), dtype=torch.float32, device=device)
    cell = torch.empty((G, post), dtype=torch.int32, device=device)
    count = torch.empty((G,), dtype=torch.int32, device=device)
    packed = torch.empty((B, S, post, 11), dtype=torch.float32, device=device)
    counts = torch.empty((B, S), dtype=torch.int32, device=device)
    ws = workspace.get("decode", L.fd_decode_workspace_bytes(G, ctypes.byref(cfg)), device)
    arr = lambda v: (ctypes.c_int32 * S)(*[int(x) for x in v])  # noqa: E731
    check(L.fd_centerpoint_decode_packed(*[ctypes.byref(v) for v in views], ctypes.byref(vel_view), int(G), int(B), ctypes.byref(cfg), S, arr(step_group),
                                         arr(step_vel_channel), arr(step_label), _p(boxes), _p(scores), _p(cell), _p(count), _p(packed), _p(counts), _p(ws),
                                         ws.numel(), _stream()), "fd_centerpoint_decode_packed")
    return packed, counts


def assemble_detections(boxes7, scores, cell, count, vel_view, B, post, step_group, step_vel_channel, step_label):
    """fd_assemble_detections: -> (packed [B, S, post, 11] float32 rows x y z w l h vx vy yaw score label, counts [B, S] int32)"""
    L = _lib.load()
    S = len(step_group)
    dev = boxes7.device
    packed = torch.empty((B, S, post, 11), dtype=torch.float32, device=dev)
    counts = torch.empty((B, S), dtype=torch.int32, device=dev)
    arr = lambda v: (ctypes.c_int32 * S)(*[int(x) for x in v])  # noqa: E731
    check(L.fd_assemble_detections(_p(boxes7), _p(scores), _p(cell), _p(count), ctypes.byref(vel_view), int(B), int(post), S, arr(step_group),
                                   arr(step_vel_channel), arr(step_label), _p(packed), _p(counts), _stream()), "fd_assemble_detections")
    return packed, counts


def rotated_nms(boxes7, thresh):
    """nms_gpu contract on device: boxes [n,7] (pcdet layout, score-sorted) -> (keep int64[n], count int32[1])."""
    L = _lib.load()
    boxes7 = _dev(boxes7, "boxes", torch.float32)
    n = boxes7.shape[0]
    keep = torch.zeros((max(n, 1),), dtype=torch.int64, device=boxes7.device)
    count = torch.zeros((1,), dtype=torch.int32, device=boxes7.device)
    ws = workspace.get("nms", L.fd_nms_workspace_bytes(n), boxes7.device)
    check(L.fd_rotated_nms(_p(boxes7), n, ctypes.c_float(thresh), _p(keep), _p(count), _p(ws), ws.numel(), _stream()),
          "fd_rotated_nms")
    return keep, count


def boxes_iou_bev(a, b):
    L = _lib.load()
    a = _dev(a, "boxes_a", torch.float32)
    b = _dev(b, "boxes_b", torch.float32)
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    check(L.fd_boxes_iou_bev(_p(a), a.shape[0], _p(b), b.shape[0], _p(out), _stream()), "fd_boxes_iou_bev")
    return out


# ------------------------------------------------------------------------------------------------ sweep assembly
SWEEP_DESC = np.dtype([("m", np.float64, (16,)), ("row_begin", np.int64), ("row_end", np.int64), ("time", np.float32),
                       ("flags", np.int32)])  # == struct fd_sweep_desc (include/futuredet_hip.h)
SWEEP_HAS_TRANSFORM, SWEEP_REMOVE_CLOSE = 1, 2


def sweep_descriptors(rows, transforms, time_lags, remove_close):
    """Host side of fd_sweep_assemble: per-sweep records in visit order.  ``rows`` are the cumulative raw-row offsets
    [S+1]; transforms[s] is a 4x4 (any float dtype; promoted to float64 like np.dot with the float64 ones row,
    loading.py:54-56) or None; time_lags[s] a Python/NumPy float; remove_close[s] a bool."""
    S = len(rows) - 1
    d = np.zeros((S,), SWEEP_DESC)
    for s in range(S):
        d["row_begin"][s], d["row_end"][s] = int(rows[s]), int(rows[s + 1])
        flags = 0
        if transforms[s] is not None:
            d["m"][s] = np.asarray(transforms[s], np.float64).reshape(16)
            flags |= SWEEP_HAS_TRANSFORM
        if remove_close[s]:
            flags |= SWEEP_REMOVE_CLOSE
        d["flags"][s] = flags
        d["time"][s] = np.float32(np.float64(time_lags[s]))  # (time_lag * ones).astype(float32), loading.py:58,134
    return d


def assemble_sweeps(raw, desc, keep_cols=4, min_distance=1.0, out=None, count=None, n_sweeps=None):
    """Runs fd_sweep_assemble on device rows ``raw`` [R, raw_cols] float32.  ``desc``: host descriptors (sweep_descriptors; uploaded
    here) or a device uint8 tensor already holding ``n_sweeps`` fd_sweep_desc records (the static form: a captured launch reads
    whatever the caller last copied there; R is then an upper bound, rows past the last descriptor's row_end are dropped).
    Returns (points [R, keep_cols+1] padded with +inf past the count, count int32[1]); ``out`` / ``count`` may be given; no sync."""
    L = _lib.load()
    raw = _dev(raw, "raw", torch.float32)
    R, raw_cols = raw.shape
    dev = raw.device
    if isinstance(desc, torch.Tensor):
        desc_dev = _dev(desc, "desc", torch.uint8)
        n_desc = int(n_sweeps) if n_sweeps is not None else desc_dev.numel() // SWEEP_DESC.itemsize
        assert desc_dev.numel() >= n_desc * SWEEP_DESC.itemsize
    else:
        desc = np.ascontiguousarray(desc)
        assert desc.dtype == SWEEP_DESC
        desc_dev = torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).to(dev, non_blocking=True)
        n_desc = len(desc)
    if out is None:
        out = torch.empty((R, keep_cols + 1), dtype=torch.float32, device=dev)
    else:
        out = _dev(out, "out", torch.float32)
        assert out.shape[0] >= R and out.shape[1] == keep_cols + 1
    if count is None:
        count = torch.empty((1,), dtype=torch.int32, device=dev)
    ws_bytes = L.fd_sweep_assemble_workspace_bytes(R)
    ws = workspace.get("sweeps", ws_bytes, dev)
    check(L.fd_sweep_assemble(_p(raw), raw_cols, int(keep_cols), R, _p(desc_dev), n_desc, float(min_distance), _p(out),
                              _p(count), _p(ws), ws.numel(), _stream()), "fd_sweep_assemble")
    return out, count


# ------------------------------------------------------------------------------------------------ PointPillars reader
def pillar_encode(voxels, num_points, coors4, n_dev, geom, layers, with_distance=False, out_dtype=torch.float32):
    """fd_pillar_encode: voxels [M,P,ndim] f32, num_points [M] i32, coors4 [M,4] i32 (b,z,y,x), n_dev device int32[1] or
    None; geom = (vx, vy, x_offset, y_offset); layers = [(weight [U,Fin] f32, scale [U], shift [U])] (1 or 2 entries).
    Returns [M, U_last] (rows past the count are left untouched: zero-initialised here)."""
    L = _lib.load()
    voxels = _dev(voxels, "voxels", torch.float32)
    M, P, ndim = voxels.shape
    (w1, s1, b1) = layers[0]
    (w2, s2, b2) = layers[1] if len(layers) > 1 else (None, None, None)
    assert len(layers) in (1, 2)
    U = (w2 if w2 is not None else w1).shape[0]
    out = torch.zeros((M, U), dtype=out_dtype, device=voxels.device)
    check(L.fd_pillar_encode(_p(voxels), _p(num_points), _p(coors4), _p(n_dev), M, P, ndim, int(bool(with_distance)),
                             float(geom[0]), float(geom[1]), float(geom[2]), float(geom[3]), _p(w1), _p(s1), _p(b1), w1.shape[0],
                             _p(w2), _p(s2), _p(b2), (w2.shape[0] if w2 is not None else 0), _DT[out_dtype], _p(out), U, _stream()),
          "fd_pillar_encode")
    return out


def pillar_scatter(feats, coors4, n_dev, batch_size, ny, nx, out_dtype=None, channels_last=False, out=None, zero_first=True):
    """fd_pillar_scatter -> [B, C, ny, nx] canvas (NCHW, or channels-last memory when asked)."""
    L = _lib.load()
    feats = _dev(feats, "feats")
    M, C = feats.shape
    if out is None:
        out = torch.empty((batch_size, C, ny, nx), dtype=out_dtype or feats.dtype, device=feats.device,
                          memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    sb, sc, sy, sx = out.stride()
    check(L.fd_pillar_scatter(_p(feats), C, feats.stride(0), _DT[feats.dtype], _p(coors4), _p(n_dev), M, batch_size, ny, nx,
                              _p(out), _DT[out.dtype], sb, sc, sy, sx, int(bool(zero_first)), _stream()), "fd_pillar_scatter")
    return out


# ------------------------------------------------------------------------------------------------ forecast association
def forecast_chains(centers, velocity, counts, time, reject_thresh):
    """fd_forecast_chains on device tensors: centers/velocity [T,n,3] float64, counts [T] int32, time [T-1] float64."""
    L = _lib.load()
    centers = _dev(centers, "centers", torch.float64)
    velocity = _dev(velocity, "velocity", torch.float64)
    T, n, _ = centers.shape
    dev = centers.device
    i32 = dict(dtype=torch.int32, device=dev)
    out = dict(fwd_idx=torch.zeros((n, T), **i32), fwd_ok=torch.zeros((n,), **i32), bwd_idx=torch.zeros((n, T), **i32),
               bwd_ok=torch.zeros((n,), **i32), match_idx=torch.zeros((T, n), **i32),
               cv_centers=torch.zeros((n, T, 3), dtype=torch.float64, device=dev), status=torch.zeros((1,), **i32))
    check(L.fd_forecast_chains(_p(centers), _p(velocity), _p(counts), _p(time), T, n, float(reject_thresh), _p(out["fwd_idx"]),
                               _p(out["fwd_ok"]), _p(out["bwd_idx"]), _p(out["bwd_ok"]), _p(out["match_idx"]), _p(out["cv_centers"]),
                               _p(out["status"]), _stream()), "fd_forecast_chains")
    return out


def det_to_global_boxes(box3d, cs_record=None, pose_record=None):
    """fd_det_to_global_boxes: box3d [n,9] float32 device tensor; cs_record / pose_record = (rotation wxyz, translation xyz)
    or None.  Returns device tensors (center [n,3] f64, quat [n,4] f64, velocity [n,3] f64, size [n,3] f32)."""
    L = _lib.load()
    box3d = _dev(box3d, "box3d", torch.float32)
    n = box3d.shape[0]
    assert box3d.shape[1] == 9, "box3d_lidar rows are (x,y,z,w,l,h,vx,vy,yaw)"
    dev = box3d.device
    center = torch.empty((n, 3), dtype=torch.float64, device=dev)
    quat = torch.empty((n, 4), dtype=torch.float64, device=dev)
    vel = torch.empty((n, 3), dtype=torch.float64, device=dev)
    size = torch.empty((n, 3), dtype=torch.float32, device=dev)

    def rec(r):
        if r is None:
            return None, None
        q, t = (ctypes.c_double * 4)(*[float(v) for v in r[0]]), (ctypes.c_double * 3)(*[float(v) for v in r[1]])
        return q, t

    cq, ct = rec(cs_record)
    pq, pt = rec(pose_record)
    check(L.fd_det_to_global_boxes(_p(box3d), n, cq, ct, pq, pt, _p(center), _p(quat), _p(vel), _p(size), _stream()), "fd_det_to_global_boxes")
    return center, quat, vel, size


def nearest_rows(library, queries):
    """fd_nearest_rows: library [M, D], queries [N, D] float64 device tensors -> int32 [N] index of the nearest library row"""
    L = _lib.load()
    library, queries = _dev(library, "library", torch.float64), _dev(queries, "queries", torch.float64)
    assert library.shape[1] == queries.shape[1]
    idx = torch.zeros((max(queries.shape[0], 1),), dtype=torch.int32, device=queries.device)[: queries.shape[0]]
    check(L.fd_nearest_rows(_p(library), library.shape[0], _p(queries), queries.shape[0], library.shape[1], _p(idx), _stream()), "fd_nearest_rows")
    return idx


def forecast_groups(centers, match_thresh):
    """fd_forecast_groups: centers [n,3] float64 device tensor -> int32 [n] component ids (multi_future's forecast_id)."""
    L = _lib.load()
    centers = _dev(centers, "centers", torch.float64)
    n = centers.shape[0]
    ids = torch.zeros((max(n, 1),), dtype=torch.int32, device=centers.device)[:n]
    check(L.fd_forecast_groups(_p(centers), n, float(match_thresh), _p(ids), _stream()), "fd_forecast_groups")
    return ids


class ForecastOutputs(object):
    """The device buffers of fd_forecast_from_detections for a batch of B sweeps with T steps of up to ``post`` boxes: allocated once,
    written by every call (fixed shapes: they can be the static outputs of a captured graph).  ``host()`` copies them to numpy."""

    FIELDS = (("center", (-1, -2, 3), torch.float64), ("quat", (-1, -2, 4), torch.float64), ("velocity", (-1, -2, 3), torch.float64),
              ("size", (-1, -2, 3), torch.float32), ("fwd_idx", (-3, -1), torch.int32), ("fwd_ok", (-3,), torch.int32),
              ("bwd_idx", (-3, -1), torch.int32), ("bwd_ok", (-3,), torch.int32), ("match_idx", (-1, -3), torch.int32),
              ("cv_centers", (-3, -1, 3), torch.float64), ("status", (), torch.int32), ("traj_kind", (-4,), torch.int32),
              ("traj_src", (-4,), torch.int32), ("traj_first", (-4,), torch.int32), ("traj_group", (-4,), torch.int32), ("n_traj", (), torch.int32))

    def __init__(self, B, T, post, device):
        self.B, self.T, self.post = int(B), int(T), int(post)
        sub = {-1: self.T, -2: self.post, -3: self.post, -4: 3 * self.post}
        # ONE allocation, every array a 16-byte aligned view of it: the whole result goes to the host in one copy (``blob``)
        layout, off = [], 0
        for name, shape, dt in self.FIELDS:
            shape = (self.B,) + tuple(sub.get(d, d) for d in shape)
            nbytes = int(np.prod(shape)) * torch.empty((), dtype=dt).element_size()
            layout.append((name, shape, dt, off, nbytes))
            off = (off + nbytes + 15) & ~15
        self.layout = layout
        self.blob = torch.zeros((off,), dtype=torch.uint8, device=device)
        for name, shape, dt, o, nbytes in layout:
            setattr(self, name, self.blob[o:o + nbytes].view(dt).view(shape))
        self.c_struct = _lib.ForecastBuffers(**{name: getattr(self, name).data_ptr() for name, _, _ in self.FIELDS})

    def views_of(self, blob_host):
        """the named arrays of a HOST copy of ``blob`` (numpy views, no copy)"""
        a = blob_host.numpy() if isinstance(blob_host, torch.Tensor) else np.asarray(blob_host)
        npdt = {torch.float64: np.float64, torch.float32: np.float32, torch.int32: np.int32}
        return {name: a[o:o + nbytes].view(npdt[dt]).reshape(shape) for name, shape, dt, o, nbytes in self.layout}

    def host(self):
        return self.views_of(self.blob.cpu())


def forecast_from_detections(packed, counts, time, records=None, reject_thresh=2.0, match_thresh=0.25, out=None):
    """fd_forecast_from_detections: packed [B,T,post,F>=9] float32, counts [B,T] int32, time [B,T-1] float64, records [B,14] float64 or
    None (all device tensors) -> ForecastOutputs (``out`` is reused when given).  Three launches, no synchronisation."""
    L = _lib.load()
    packed = _dev(packed, "packed", torch.float32)
    counts = _dev(counts, "counts", torch.int32)
    time = _dev(time, "time", torch.float64)
    B, T, post, F = packed.shape
    assert tuple(counts.shape) == (B, T) and tuple(time.shape) == (B, T - 1)
    if records is not None:
        records = _dev(records, "records", torch.float64)
        assert tuple(records.shape) == (B, 14)
    if out is None:
        out = ForecastOutputs(B, T, post, packed.device)
    assert (out.B, out.T, out.post) == (B, T, post)
    check(L.fd_forecast_from_detections(_p(packed), _p(counts), B, T, post, F, _p(records), _p(time), float(reject_thresh), float(match_thresh),
                                        ctypes.byref(out.c_struct), _stream()), "fd_forecast_from_detections")
    return out
