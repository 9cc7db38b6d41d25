"""Detectors (det3d/models/detectors/{base,single_stage,voxelnet}.py).

VoxelNet.forward(example, return_loss) keeps the reference contract (voxelnet.py:33-56).  In addition
``forward_points`` is the device-resident fast path the benchmark measures: raw point clouds (device tensors)
-> fused voxelizer/mean -> sparse backbone -> neck -> head -> decode, with one host read of the five
active-row counts and one of the final detections."""
import contextlib
import os

import numpy as np
import torch
from torch import nn

from . import hip_ops, registry, sparse
from .nn_utils import weights_version
from .registry import DETECTORS

_FORK = os.environ.get("FD_FORK", "0") == "1"  # A/B switch: rulebooks of levels 1-4 on a side stream inside a captured sweep
_NO_GRAPH = bool(os.environ.get("FD_NO_GRAPH"))  # debugging: run neck + head eagerly


class BaseDetector(nn.Module):
    @property
    def with_neck(self):
        return hasattr(self, "neck") and self.neck is not None

    @property
    def with_reader(self):
        return hasattr(self, "reader") and self.reader is not None


@DETECTORS.register_module
class SingleStageDetector(BaseDetector):
    def __init__(self, reader, backbone, neck=None, bbox_head=None, train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__()
        self.reader = registry.build_reader(reader)
        self.backbone = registry.build_backbone(backbone)
        if neck is not None:
            self.neck = registry.build_neck(neck)
        self.bbox_head = registry.build_head(bbox_head)
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg
        self.init_weights(pretrained=pretrained)

    def init_weights(self, pretrained=None):
        if pretrained is None:
            return
        try:
            load_checkpoint(self, pretrained, strict=False)
            print("init weight from {}".format(pretrained))
        except Exception:
            print("no pretrained model at {}".format(pretrained))


_REMOTE_SCHEMES = ("modelzoo://", "torchvision://", "open-mmlab://", "http://", "https://")


def load_checkpoint(model, filename, map_location="cpu", strict=False, logger=None):
    """The contract of det3d/torchie/trainer/checkpoint.py:122-173 for local files: ``filename`` must be an existing file (IOError
    otherwise); the checkpoint is an OrderedDict of tensors or a dict with a "state_dict" entry (RuntimeError otherwise); a
    leading "module." (DataParallel / DDP wrapper) is stripped from every key when the first key carries it; a wrapper model is
    loaded through its ``.module``; returns the checkpoint object.  The reference also downloads modelzoo:// / torchvision:// /
    open-mmlab:// / http(s):// names through torch's model zoo -- this build has no network path and says so instead."""
    import collections
    import os.path as osp

    if isinstance(filename, str) and filename.startswith(_REMOTE_SCHEMES):
        raise NotImplementedError("load_checkpoint: remote checkpoints (%s) are not fetched by this build; download the file and pass its path"
                                  % filename.split("://")[0])
    if not osp.isfile(filename):
        raise IOError("{} is not a checkpoint file".format(filename))
    ckpt = torch.load(filename, map_location=map_location)
    if isinstance(ckpt, collections.OrderedDict):
        sd = ckpt
    elif isinstance(ckpt, dict) and "state_dict" in ckpt:
        sd = ckpt["state_dict"]
    else:
        raise RuntimeError("No state_dict found in checkpoint file {}".format(filename))
    keys = list(sd.keys())
    if keys and keys[0].startswith("module."):
        sd = collections.OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in sd.items())
    target = model.module if hasattr(model, "module") else model
    res = target.load_state_dict(sd, strict=strict)
    if logger is not None and (res.missing_keys or res.unexpected_keys):
        logger.warning("load_checkpoint: missing keys %s, unexpected keys %s", list(res.missing_keys), list(res.unexpected_keys))
    return ckpt


@DETECTORS.register_module
class VoxelNet(SingleStageDetector):
    def __init__(self, reader, backbone, neck, bbox_head, train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__(reader, backbone, neck, bbox_head, train_cfg, test_cfg, pretrained)
        # a captured neck+head graph replays kernels that read the folded / packed weights of the moment of capture
        self.register_load_state_dict_post_hook(lambda m, keys: m.invalidate_caches())
        self.set_precision(torch.float32)

    def invalidate_caches(self):
        """Drops the captured hipGraphs (and, through the sub-modules' own hooks / version keys, every derived weight).
        Called by load_state_dict, set_precision and device moves; call it by hand after writing weights through ``.data``."""
        self.__dict__.pop("_graphs", None)
        for m in (self.neck, self.bbox_head):
            if hasattr(m, "invalidate_caches"):
                m.invalidate_caches()

    def _apply(self, fn, *a, **kw):
        self.__dict__.pop("_graphs", None)
        return super()._apply(fn, *a, **kw)

    def set_precision(self, dtype=torch.float32, channels_last=None):
        """fp32 (default) or bf16 conv features/weights with fp32 accumulation; voxelizer, indexes, decode and
        NMS always stay fp32/int.  ``channels_last`` is accepted for compatibility and ignored: the neck and head run on
        the hand-written NHWC convolutions in both dtypes, so the BEV map is always written channels-last."""
        self.__dict__.pop("_graphs", None)
        self.backbone.compute_dtype = dtype
        self.backbone.dense_channels_last = True
        self.neck.compute_dtype = dtype
        self.bbox_head.compute_dtype = dtype
        return self

    def extract_feat(self, data):
        input_features = self.reader(data["features"], data["num_voxels"])
        x, voxel_feature = self.backbone(input_features, data["coors"], data["batch_size"], data["input_shape"])
        if self.with_neck:
            x = self.neck(x)
        return x, voxel_feature

    def forward(self, example, return_loss=True, **kwargs):
        voxels = example["voxels"]
        coordinates = example["coordinates"]
        num_points_in_voxel = example["num_points"]
        num_voxels = example["num_voxels"]
        batch_size = len(num_voxels)
        data = dict(features=voxels, num_voxels=num_points_in_voxel, coors=coordinates, batch_size=batch_size,
                    input_shape=example["shape"][0])
        bev_map = None
        if self.bbox_head.bev_map:
            bev_map = torch.stack(example["bev_map"], dim=1).float()
        x, _ = self.extract_feat(data)
        preds = self.bbox_head(x, bev_map)
        if return_loss:
            return self.bbox_head.loss(example, preds)
        return self.bbox_head.predict(example, preds, self.test_cfg)

    def _dense_graph(self, B, idx4, dev):
        """hipGraph of neck + head for a given batch size / precision (captured once, after an eager warm-up that lets
        MIOpen pick its kernels).  Returns (graph, static BEV input, static prediction dicts) or None."""
        bb = self.backbone
        dt = bb.dense_dtype or bb.compute_dtype
        # the weights' version is part of the key: after an in-place update (optimizer step, init_weights, BN statistics)
        # the old graph would replay kernels that read freed weight buffers
        ver = weights_version(self.neck, self.bbox_head)
        cache = self.__dict__.setdefault("_graphs", {})
        if cache.get("version") != ver:
            cache.clear()
            cache["version"] = ver
        # one graph (with its own static input / output buffers) per stream: sweeps in flight on different streams replay
        # different graphs
        key = (B, dt, bb.dense_channels_last, idx4.D, idx4.H, idx4.W, dev, torch.cuda.current_stream(dev).cuda_stream)
        if key in cache:
            return cache[key]
        try:
            static_bev = torch.empty((B, 128 * idx4.D, idx4.H, idx4.W), dtype=dt, device=dev,
                                     memory_format=torch.channels_last if bb.dense_channels_last else torch.contiguous_format).zero_()
            cur = torch.cuda.current_stream(dev)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(2):
                    self.bbox_head(self.neck(static_bev), None)
            cur.wait_stream(side)
            g = torch.cuda.CUDAGraph()
            # thread-local capture mode: another thread of the process (RCCL's watchdog in a multi-rank run) may call the
            # HIP runtime while this thread captures
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                preds = self.bbox_head(self.neck(static_bev), None)
            cache[key] = (g, static_bev, preds)
        except Exception as e:  # capture is an optimisation; the eager launches are always available
            print("[futuredet_amd] hipGraph capture of neck+head unavailable (%r); running eagerly" % (e,))
            cache[key] = None
        return cache[key]

    # ------------------------------------------------------------------------------------------------ fast path
    @torch.no_grad()
    def forward_points(self, clouds, voxel_cfg, bev_map=None, padded=True, counts=None, static=False, expected=None, row_caps=None):
        """clouds: list of device float32 [N_i, 5] tensors (one merged multi-sweep cloud per sample);
        voxel_cfg: the config's ``voxel_generator`` dict (range, voxel_size, max_points_in_voxel, max_voxel_num).
        Returns predict_padded()'s tuple (padded=True) or the list of per-sample dicts.
        ``counts``: per cloud a device int32[1] with its number of valid rows (fixed-capacity buffers, loading.assemble_device).
        ``static=True``: no host read-back anywhere -- sparse levels are sized by their row capacities and every kernel takes
        its count from the device, so the call can be captured into one hipGraph (StaticStep); ``expected`` = typical row
        counts per level (launch heuristics only).  Results are identical to the default mode.  ``row_caps``: per-level row
        capacities below the data-free bounds (static mode; see hip_ops.build_pyramid -- the caller checks for overflow)."""
        assert not self.training
        assert not (static and not padded), "the static step returns the padded device tuple"
        mark_stage = getattr(self, "stage_hook", None) or (lambda name: None)
        mark_stage("start")
        dev = clouds[0].device
        B = len(clouds)
        rng, vs = voxel_cfg["range"], voxel_cfg["voxel_size"]
        mv = voxel_cfg["max_voxel_num"]
        max_voxels = int(mv[1] if isinstance(mv, (list, tuple)) else mv)  # eval cap (preprocess.py:256-258)
        max_points = int(voxel_cfg["max_points_in_voxel"])
        cpad = sparse.pad_channels(clouds[0].shape[1])
        mean = torch.empty((B * max_voxels, cpad), dtype=torch.float32, device=dev)
        coors = torch.empty((B * max_voxels, 4), dtype=torch.int32, device=dev)
        npts = torch.empty((B * max_voxels,), dtype=torch.int32, device=dev)
        nvox = torch.empty((B,), dtype=torch.int32, device=dev)  # written by every fd_voxelize call, empty clouds included
        for b, pts in enumerate(clouds):
            sl = slice(b * max_voxels, (b + 1) * max_voxels)
            hip_ops.voxelize(pts, vs, rng, max_points, max_voxels, batch_idx=b, want_voxels=False, want_mean=True,
                             out=dict(mean=mean[sl], coors=coors[sl], num_points=npts[sl], num_voxels=nvox[b:b + 1]),
                             n_points_dev=None if counts is None else counts[b])
        grid = np.round((np.array(rng[3:], np.float32) - np.array(rng[:3], np.float32)) / np.array(vs, np.float32)).astype(np.int64)

        def mark(i0):
            for b in range(B):
                sl = slice(b * max_voxels, (b + 1) * max_voxels)
                i0.mark(coors[sl], n_dev=nvox[b:b + 1], n_max=max_voxels)

        mark_stage("voxelize")
        bb = self.backbone
        idx = bb.build_indexes(mark, B, list(grid), dev, voxels=(coors, nvox, max_voxels), static=static, expected=expected, row_caps=row_caps)
        # (static: idx[l].n is the level's capacity; the counts are read by whoever wants them from level_counts, later)
        self.__dict__["last_level_counts"] = idx[0].level_counts if static else [ix.n for ix in idx]
        mark_stage("index")
        # every row of the level-0 index is exactly one voxel, and fd_rows_place writes all cpad channels of it: no fill
        feats0 = torch.empty((max(idx[0].n, 1), cpad), dtype=bb.compute_dtype, device=dev)[: idx[0].n]
        L = hip_ops._lib.load()
        i0 = idx[0]
        for b in range(B if i0.n > 0 else 0):  # (nothing to place when every cloud of the batch is empty)
            sl = slice(b * max_voxels, (b + 1) * max_voxels)
            hip_ops.check(L.fd_rows_place(hip_ops._p(i0.words), hip_ops._p(i0.prefix), i0.B, i0.D, i0.H, i0.W, hip_ops._p(coors[sl]),
                                          hip_ops._p(nvox[b:b + 1]), max_voxels, hip_ops._p(mean[sl]), cpad, hip_ops._p(feats0), cpad,
                                          hip_ops._DT[bb.compute_dtype], hip_ops._stream()), "fd_rows_place")
        graph = None if (bev_map is not None or _NO_GRAPH or static) else self._dense_graph(B, idx[4], dev)
        if graph is not None:
            # neck + head have static shapes: replay them as one hipGraph (one launch instead of ~25-60)
            g, static_bev, preds = graph
            bb.run_fused(idx, feats0, dense_out=static_bev, fork=static and _FORK)
            mark_stage("sparse_backbone")
            g.replay()
            mark_stage("rpn+head(graph)")
        else:
            x, _ = bb.run_fused(idx, feats0, fork=static and _FORK)
            mark_stage("sparse_backbone")
            taps = self.__dict__.get("debug_taps")  # tools/soak_determinism.py: a dict that keeps the stages' tensors (static memory inside a capture)
            if taps is not None:
                taps.update(mean=mean, coors=coors, num_points=npts, num_voxels=nvox, feats0=feats0, bev=x, idx=idx)
            x = self.neck(x)
            mark_stage("rpn")
            preds = self.bbox_head(x, bev_map)
            mark_stage("head")
            if taps is not None:
                taps.update(neck=x, head=[getattr(pd, "raw", (None,))[0] for pd in preds][0], preds=preds)
        if padded == "packed":  # (packed [B,S,post,11], counts [B,S]): what the multi-GPU gather and the bench move
            out = self.bbox_head.predict_packed(preds, self.test_cfg)
            if out is None:
                from .dist_infer import pack_results
                out = pack_results(*self.bbox_head.predict_padded(preds, self.test_cfg))
        elif padded:
            out = self.bbox_head.predict_padded(preds, self.test_cfg)
        else:
            out = self.bbox_head.predict({"metadata": [None] * B}, preds, self.test_cfg)
        mark_stage("decode")
        return out


class StaticStep(object):
    """One whole sweep -- voxelizer, index pyramid, rulebooks, 21 sparse convolutions, RPN, CenterHead, decode, rotated NMS --
    as ONE hipGraph replay with no host read-back in between.  Shapes are static: the clouds live in a
    fixed-capacity buffer with their row counts on the device, sparse levels are sized by their row capacities
    (hip_ops._PyramidPlan.row_caps) and every kernel takes its count from device memory.

        step = StaticStep(model, voxel_cfg, capacity=400000)       # model: VoxelNet in eval mode on a GPU
        step.warm_up([cloud])                                      # eager sweeps: conv plan choices + typical level counts
        boxes, scores, labels, counts = step([cloud])              # copies the cloud in, replays; outputs are the graph's
                                                                   # static tensors (valid until the next call on this stream)
    One StaticStep belongs to one stream (the one current at capture); sweeps in flight on several streams use one each."""

    def __init__(self, model, voxel_cfg, capacity, batch_size=1, ndim=5, packed=False, row_caps="auto", headroom=1.5):
        """``row_caps``: "auto" (default) sizes the sparse levels by ``headroom`` x the counts of the warm-up clouds (+ 4096 rows): a
        sweep that needs more rows than that is detected from its level counts and re-run on the eager path -- ``step(clouds)`` does
        both (one synchronisation per call); callers that copy ``level_counts`` next to their results pass ``check=False`` and call
        ``overflowed(counts)`` themselves, as bench.py does.  "datafree" sizes the levels by bounds that hold for ANY cloud (no
        overflow possible, no check; ~2 GB of scratch per step at the bench configuration, 1.28 M / 1.43 M rows for levels 1 / 2 where
        a 300k-point cloud has 0.27 M / 0.16 M, and fp32 batches of 8 exceed the fp32 kernel's 2^23-row packing)."""
        self.model, self.voxel_cfg = model, voxel_cfg
        self.padded = "packed" if packed else True  # packed: outputs = (packed [B,S,post,11], counts [B,S]) instead of four tensors
        self.row_caps_mode, self.headroom = row_caps, float(headroom)
        self.caps = None  # per-level row capacities of the captured step (None: data-free bounds)
        self.bev = None   # static [B, 6, H, W] input of a bev_map head (n3dtfm), allocated by the first call that passes one
        self.B, self.capacity, self.ndim = int(batch_size), int(capacity), int(ndim)
        dev = next(model.parameters()).device
        self.points = torch.zeros((self.B, self.capacity, self.ndim), dtype=torch.float32, device=dev)
        self.counts = torch.zeros((self.B,), dtype=torch.int32, device=dev)
        self._counts_host, self._counts_slot = None, 0
        self.expected = None
        self.graph = None
        self.outputs = None
        self.level_counts = None
        self.version = None

    def _version_key(self):
        m = self.model
        return (weights_version(m), getattr(m.backbone, "compute_dtype", None), m.neck.compute_dtype)

    def _load(self, clouds):
        assert len(clouds) == self.B
        if self._counts_host is None:  # ring of pinned count vectors, each with the event of the copy that last read it
            self._counts_host = [[torch.empty((self.B,), dtype=torch.int32).pin_memory(), None] for _ in range(8)]
        slot = self._counts_host[self._counts_slot % len(self._counts_host)]
        self._counts_slot += 1
        host = slot[0]
        if slot[1] is not None:
            slot[1].synchronize()  # (returns at once unless the stream is eight sweeps behind the host)
        for b, c in enumerate(clouds):
            n = int(c.shape[0])
            if n > self.capacity or c.shape[1] != self.ndim:
                raise ValueError("cloud of %d x %d rows does not fit the step's %d x %d buffer" % (n, c.shape[1], self.capacity, self.ndim))
            self.points[b, :n].copy_(c, non_blocking=True)
            host[b] = n
        self.counts.copy_(host, non_blocking=True)  # one small copy for all samples (was: one fill kernel per sample)
        slot[1] = torch.cuda.Event()
        slot[1].record()

    def __del__(self):
        try:
            hip_ops.workspace.release(id(self))
        except Exception:
            pass

    def _set_bev(self, bev_map):
        if bev_map is None:
            return
        if self.bev is None:
            self.bev = torch.zeros_like(bev_map)
            self.graph = None
        self.bev.copy_(bev_map, non_blocking=True)

    def _run(self, static):
        m = self.model
        if static:  # the captured sweep keeps scratch buffers of its own (not those of whatever stream it is captured on)
            with hip_ops.workspace.scope(id(self)):
                return m.forward_points([self.points[b] for b in range(self.B)], self.voxel_cfg, padded=self.padded, bev_map=self.bev,
                                        counts=[self.counts[b:b + 1] for b in range(self.B)], static=True, expected=self.expected, row_caps=self.caps)
        return m.forward_points([self.points[b] for b in range(self.B)], self.voxel_cfg, padded=self.padded, bev_map=self.bev,
                                counts=[self.counts[b:b + 1] for b in range(self.B)], static=static, expected=self.expected)

    def warm_up(self, clouds, n=2, bev_map=None):
        """Eager sweeps on representative clouds: lets the dense-conv plan time its variants and records the level counts that
        steer the sparse-conv launch shapes of the captured step."""
        self._set_bev(bev_map)
        self._load(clouds)
        for _ in range(n):
            self._run(False)
        self.expected = [int(v) for v in self.model.last_level_counts]  # (empty for PointPillars: no sparse levels)
        self.graph = None

    def capture(self):
        m = self.model
        dev = self.points.device
        if self.expected is None:
            raise RuntimeError("StaticStep.warm_up(clouds) must run before the capture")
        mv = self.voxel_cfg["max_voxel_num"]
        cap0 = self.B * int(mv[1] if isinstance(mv, (list, tuple)) else mv)
        self.caps = None
        if self.row_caps_mode != "datafree" and isinstance(m, VoxelNet) and self.expected:
            if self.row_caps_mode == "auto":
                self.caps = [cap0] + [int(self.headroom * e) + 4096 for e in self.expected[1:]]  # (level 0 is capped by the voxelizer itself)
            else:
                self.caps = [int(c) for c in self.row_caps_mode]
        worst = max(self.caps[1:]) if self.caps else cap0 * 8
        if getattr(m.backbone, "compute_dtype", None) == torch.float32 and isinstance(m, VoxelNet) and worst >= (1 << 23):
            # the pair-compacting fp32 kernel packs (input row, local row) into 32 bits: 2^23 input rows at most; a capacity
            # beyond that would send every launch to the slower register kernel
            raise NotImplementedError("StaticStep: %d x max_voxels gives sparse levels a row capacity >= 2^23 (fp32 kernel limit); "
                                      "use forward_points or a smaller batch" % self.B)
        cur = torch.cuda.current_stream(dev)
        # a capture stream of its own: hip_ops' scratch buffers are keyed by stream, so two steps never share scratch memory
        cap = torch.cuda.Stream(device=dev)
        cap.wait_stream(cur)
        with torch.cuda.stream(cap):
            self._run(True)  # allocates capacity-sized scratch outside the graph's pool once
        cur.wait_stream(cap)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
            self.outputs = self._run(True)
            self.level_counts = m.last_level_counts
        self.graph = g
        self.version = self._version_key()

    def __call__(self, clouds, bev_map=None, check=True):
        """Replays the captured sweep and returns the graph's static output tensors (``self.outputs``; valid until the next call).
        With capacities from a high-water mark (row_caps="auto") and ``check`` (the default: always correct) the level counts are read
        back after the replay -- ONE device synchronisation per call -- and an overflowing sweep is re-run on the eager path, its
        results copied INTO the static output tensors, so the returned tensors and ``self.outputs`` always agree.
        ``check=False`` keeps the call asynchronous: the caller copies ``level_counts`` next to its results and asks
        ``overflowed(counts)`` itself (bench.py does); row_caps="datafree" never overflows and never synchronises."""
        self._set_bev(bev_map)
        if self.graph is None or self.version != self._version_key():
            if self.expected is None:
                self.warm_up(clouds)
            self.capture()
        self._load(clouds)
        self.graph.replay()
        if check and self.caps is not None and self.overflowed():
            redo = self.model.forward_points(clouds, self.voxel_cfg, padded=self.padded, bev_map=self.bev)
            for dst, src in zip(self.outputs, redo):  # fixed shapes on both paths (padded / packed results)
                dst.copy_(src)
        return self.outputs

    def overflowed(self, level_counts_host=None):
        """True when the last replay needed more rows on some level than the captured capacities (row_caps="auto"): its outputs
        are then wrong and the sweep has to be re-run eagerly.  Synchronises unless the counts are passed in (host copy of
        ``level_counts`` taken after the replay)."""
        if self.caps is None or self.level_counts is None or not len(self.caps):
            return False
        lc = self.level_counts.cpu().tolist() if level_counts_host is None else [int(v) for v in level_counts_host]
        return any(n > c for n, c in zip(lc, self.caps))

    def run_checked(self, clouds):
        """replay + overflow check (one synchronisation) + eager re-run of an overflowing sweep (what ``step(clouds)`` does by default)"""
        return self(clouds, check=True)


class FullSweepStep(StaticStep):
    """FutureDet end to end as ONE hipGraph replay (SURVEY 8f-2 + A15 + 8f-3):

        raw sweeps [R,5] + per-sweep 4x4 transforms / time lags (HBM)  --fd_sweep_assemble-->  the 10-sweep cloud
        --StaticStep's sweep (voxelizer ... rotated NMS)-->  packed detections
        --fd_forecast_from_detections-->  global-frame boxes, forward / back-cast chains, constant-velocity roll-outs, trajectory list
                                          + forecast ids (hip_ops.ForecastOutputs, device buffers)

    i.e. det3d/datasets/pipelines/loading.py:100-141 in front of the detector and det3d/datasets/nuscenes/nuscenes.py:384-494 (forecast_mode
    "velocity_dense": tracker, :125-257) + multi_future (:299-339) behind it.  File reading and the devkit's token / pose look-ups stay
    with the caller: the raw rows, the descriptors (hip_ops.sweep_descriptors), ``time`` and the two pose records come in as arrays.

        step = FullSweepStep(model, voxel_cfg, capacity=400000, n_sweeps=10)
        step.warm_up(samples); step.capture()
        packed, counts = step(samples)        # samples: one dict per cloud with device tensors raw [R,5] f32, desc uint8 (descriptor
        step.forecast.host()                  # records), optional time [T-1] f64 and records [14] f64
    """

    def __init__(self, model, voxel_cfg, capacity, n_sweeps=10, batch_size=1, keep_cols=4, raw_cols=5, classname="car", min_distance=1.0, **kw):
        super().__init__(model, voxel_cfg, capacity, batch_size=batch_size, ndim=keep_cols + 1, packed=True, **kw)
        dev = self.points.device
        head = model.bbox_head
        self.T = int(head.timesteps if getattr(head, "timesteps", 1) > 1 else getattr(head, "target_timesteps", 7))
        self.n_sweeps, self.keep_cols, self.classname, self.min_distance = int(n_sweeps), int(keep_cols), classname, float(min_distance)
        self.raw = torch.zeros((self.B, self.capacity, raw_cols), dtype=torch.float32, device=dev)
        self.desc = torch.zeros((self.B, self.n_sweeps * hip_ops.SWEEP_DESC.itemsize), dtype=torch.uint8, device=dev)
        self.time = torch.full((self.B, max(self.T - 1, 1)), 0.5, dtype=torch.float64, device=dev)
        ident = torch.tensor([1.0, 0, 0, 0, 0, 0, 0] * 2, dtype=torch.float64, device=dev)
        self.records = ident.repeat(self.B, 1).contiguous()
        self.forecast = None

    def _load(self, samples):
        assert len(samples) == self.B
        for b, smp in enumerate(samples):
            raw, desc = smp["raw"], smp["desc"]
            n = int(raw.shape[0])
            if n > self.capacity or raw.shape[1] != self.raw.shape[2]:
                raise ValueError("%d x %d raw rows do not fit the step's %d x %d buffer" % (n, raw.shape[1], self.capacity, self.raw.shape[2]))
            if desc.numel() != self.desc.shape[1]:
                raise ValueError("the step was built for %d sweeps per cloud" % self.n_sweeps)
            self.raw[b, :n].copy_(raw, non_blocking=True)   # (rows past the last descriptor's row_end are ignored by the assembly)
            self.desc[b].copy_(desc.reshape(-1), non_blocking=True)
            if smp.get("time") is not None:
                self.time[b].copy_(smp["time"], non_blocking=True)
            if smp.get("records") is not None:
                self.records[b].copy_(smp["records"], non_blocking=True)

    def _run(self, static):
        from .forecast import sweep_forecast

        scope = (lambda: hip_ops.workspace.scope(id(self))) if static else contextlib.nullcontext
        with scope():
            for b in range(self.B):  # the cloud and its row count are born in the sweep's static input buffers
                hip_ops.assemble_sweeps(self.raw[b], self.desc[b], keep_cols=self.keep_cols, min_distance=self.min_distance, out=self.points[b],
                                        count=self.counts[b:b + 1], n_sweeps=self.n_sweeps)
        outs = StaticStep._run(self, static)
        if self.T >= 2:
            packed, counts = outs
            self.forecast = sweep_forecast(packed, counts, self.time, self.records, self.classname, out=self.forecast)
        return outs

    def __call__(self, samples, bev_map=None, check=True):
        self._set_bev(bev_map)
        if self.graph is None or self.version != self._version_key():
            if self.expected is None:
                self.warm_up(samples)
            self.capture()
        self._load(samples)
        self.graph.replay()
        if check and self.caps is not None and self.overflowed():  # rare: more rows than the captured capacities -> eager launches
            redo = self._run(False)
            for dst, src in zip(self.outputs, redo):
                dst.copy_(src)
        return self.outputs


@DETECTORS.register_module
class PointPillars(SingleStageDetector):
    """det3d/models/detectors/point_pillars.py:5-50 (the two *_pp_* configs).  Same contract as VoxelNet: forward(example)
    mirrors the reference; forward_points is the device-resident path (voxelizer with point slots -> fused pillar
    reader -> scatter -> RPN -> CenterHead -> decode)."""

    def __init__(self, reader, backbone, neck, bbox_head, train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__(reader, backbone, neck, bbox_head, train_cfg, test_cfg, pretrained)

    def set_precision(self, dtype=torch.float32, channels_last=None):
        self.reader.compute_dtype = dtype
        self.backbone.dense_channels_last = True
        self.neck.compute_dtype = dtype
        self.bbox_head.compute_dtype = dtype
        return self

    def extract_feat(self, data):
        input_features = self.reader(data["features"], data["num_voxels"], data["coors"])
        x = self.backbone(input_features, data["coors"], data["batch_size"], data["input_shape"])
        if self.with_neck:
            x = self.neck(x)
        return x

    def forward(self, example, return_loss=True, **kwargs):
        num_voxels = example["num_voxels"]
        data = dict(features=example["voxels"], num_voxels=example["num_points"], coors=example["coordinates"],
                    batch_size=len(num_voxels), input_shape=example["shape"][0])
        x = self.extract_feat(data)
        preds = self.bbox_head(x)
        if return_loss:
            return self.bbox_head.loss(example, preds)
        return self.bbox_head.predict(example, preds, self.test_cfg)

    @torch.no_grad()
    def forward_points(self, clouds, voxel_cfg, bev_map=None, padded=True, counts=None, static=False, expected=None, row_caps=None):
        """Same contract as VoxelNet.forward_points.  This path never reads anything back (pillar count and point counts stay on
        the device), so ``static`` changes nothing here; ``counts`` = device row counts of fixed-capacity cloud buffers."""
        assert not self.training
        mark_stage = getattr(self, "stage_hook", None) or (lambda name: None)
        mark_stage("start")
        dev = clouds[0].device
        B = len(clouds)
        rng, vs = voxel_cfg["range"], voxel_cfg["voxel_size"]
        mv = voxel_cfg["max_voxel_num"]
        max_voxels = int(mv[1] if isinstance(mv, (list, tuple)) else mv)
        max_points = int(voxel_cfg["max_points_in_voxel"])
        ndim = clouds[0].shape[1]
        self.__dict__["last_level_counts"] = []
        voxels = torch.empty((B * max_voxels, max_points, ndim), dtype=torch.float32, device=dev)
        coors = torch.empty((B * max_voxels, 4), dtype=torch.int32, device=dev)
        npts = torch.empty((B * max_voxels,), dtype=torch.int32, device=dev)
        nvox = torch.empty((B,), dtype=torch.int32, device=dev)  # written by every fd_voxelize call, empty clouds included
        grid = np.round((np.array(rng[3:], np.float32) - np.array(rng[:3], np.float32)) / np.array(vs, np.float32)).astype(np.int64)
        canvas = None
        for b, pts in enumerate(clouds):
            sl = slice(b * max_voxels, (b + 1) * max_voxels)
            hip_ops.voxelize(pts, vs, rng, max_points, max_voxels, batch_idx=b, want_voxels=True, coor_cols=4,
                             out=dict(voxels=voxels[sl], coors=coors[sl], num_points=npts[sl], num_voxels=nvox[b:b + 1]),
                             n_points_dev=None if counts is None else counts[b])
        mark_stage("voxelize")
        if static:
            self.__dict__["last_level_counts"] = nvox
        for b in range(B):
            sl = slice(b * max_voxels, (b + 1) * max_voxels)
            feats = self.reader(voxels[sl], npts[sl], coors[sl], n_dev=nvox[b:b + 1])
            if canvas is None:
                canvas = torch.empty((B, feats.shape[1], int(grid[1]), int(grid[0])), dtype=feats.dtype, device=dev,
                                     memory_format=torch.channels_last if self.backbone.dense_channels_last
                                     else torch.contiguous_format)
            hip_ops.pillar_scatter(feats, coors[sl], nvox[b:b + 1], B, int(grid[1]), int(grid[0]), out=canvas, zero_first=(b == 0))
        mark_stage("pillars")
        x = self.neck(canvas)
        mark_stage("rpn")
        preds = self.bbox_head(x, bev_map)
        mark_stage("head")
        if padded == "packed":  # (packed [B,S,post,11], counts [B,S]): what the multi-GPU gather and the bench move
            out = self.bbox_head.predict_packed(preds, self.test_cfg)
            if out is None:
                from .dist_infer import pack_results
                out = pack_results(*self.bbox_head.predict_padded(preds, self.test_cfg))
        elif padded:
            out = self.bbox_head.predict_padded(preds, self.test_cfg)
        else:
            out = self.bbox_head.predict({"metadata": [None] * B}, preds, self.test_cfg)
        mark_stage("decode")
        return out


@DETECTORS.register_module
class TwoStageDetector(BaseDetector):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("TWO_STAGE is False in every shipped config")
