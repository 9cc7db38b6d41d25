"""Sweep loading in front of the voxelizer, mirroring det3d/datasets/pipelines/loading.py.

read_file                : loading.py:24-33   (nuScenes .bin = float32 rows of 5; the first 4 columns are kept)
sweep_visit_order        : loading.py:128-129 (sweeps are visited in a seeded permutation: default_rng(0).choice)
LoadPointCloudFromFile   : loading.py:100-141 (NuScenesDataset branch): key frame + (nsweeps-1) sweeps ->
                           res["lidar"]["points" | "times" | "combined"]

File reading stays on the host (it is I/O); the raw rows go to the GPU once and remove_close / the 4x4 transform /
the time column / the concatenation run there as one stable compaction (fd_sweep_assemble), so the merged cloud is
born in HBM and feeds the voxelizer with no host round trip (``assemble_device``).  There is no CPU fallback.
"""
import numpy as np
import torch

from . import hip_ops
from .registry import PIPELINES

MIN_DISTANCE = 1.0  # read_sweep, loading.py:48


def read_file(path, tries=2, num_point_feature=4, painted=False):
    if painted:
        raise NotImplementedError("painted point clouds are not produced by any shipped config (painted=False)")
    return np.fromfile(path, dtype=np.float32).reshape(-1, 5)[:, :num_point_feature]


def sweep_visit_order(n_sweeps_available, nsweeps):
    rng = np.random.default_rng(0)
    return [int(i) for i in rng.choice(n_sweeps_available, nsweeps - 1, replace=False)]


def gather_raw(info, nsweeps, reader=None):
    """Host part: reads the key frame and the sweeps in visit order.  Returns (raw [R,5] float32 pinned tensor,
    descriptors) ready for hip_ops.assemble_sweeps."""
    reader = reader or (lambda p: np.fromfile(str(p), dtype=np.float32).reshape(-1, 5))
    assert (nsweeps - 1) == len(info["sweeps"]), "nsweeps {} should equal to list length {}.".format(nsweeps, len(info["sweeps"]))
    chunks = [reader(info["lidar_path"])]
    transforms, lags, close = [None], [0.0], [False]  # key frame: as read, time 0 (loading.py:113-116)
    for i in sweep_visit_order(len(info["sweeps"]), nsweeps):
        sweep = info["sweeps"][i]
        chunks.append(reader(sweep["lidar_path"]))
        transforms.append(sweep["transform_matrix"])
        lags.append(sweep["time_lag"])
        close.append(True)
    rows = np.cumsum([0] + [len(c) for c in chunks]).astype(np.int64)
    raw = torch.empty((int(rows[-1]), 5), dtype=torch.float32, pin_memory=torch.cuda.is_available())
    off = 0
    for c in chunks:
        raw[off:off + len(c)] = torch.from_numpy(np.ascontiguousarray(c, np.float32))
        off += len(c)
    return raw, hip_ops.sweep_descriptors(rows, transforms, lags, close)


def assemble_device(info, nsweeps, device="cuda", reader=None):
    """-> (combined [R, 5] float32 on ``device``, rows past the count are +inf; count int32[1] on device).  No sync:
    pass ``combined`` straight to VoxelNet.forward_points / hip_ops.voxelize (padding rows fall outside any range)."""
    raw, desc = gather_raw(info, nsweeps, reader)
    return hip_ops.assemble_sweeps(raw.to(device, non_blocking=True), desc, keep_cols=4, min_distance=MIN_DISTANCE)


@PIPELINES.register_module
class LoadPointCloudFromFile(object):
    def __init__(self, dataset="KittiDataset", **kwargs):
        self.type = dataset
        self.random_select = kwargs.get("random_select", False)
        self.npoints = kwargs.get("npoints", 16834)
        self.device = kwargs.get("device", "cuda")

    def __call__(self, res, info):
        res["type"] = self.type
        if self.type != "NuScenesDataset":
            raise NotImplementedError("only the NuScenesDataset branch is on the FutureDet path (all shipped configs)")
        if res.get("painted", False):
            raise NotImplementedError("painted point clouds are not produced by any shipped config")
        combined, count = assemble_device(info, res["lidar"]["nsweeps"], self.device)
        n = int(count.item())  # the reference returns exactly-sized arrays; this class keeps that contract (one sync)
        combined = combined[:n]
        res["lidar"]["points"] = combined[:, :4]
        res["lidar"]["times"] = combined[:, 4:5]
        res["lidar"]["combined"] = combined
        return res, info
