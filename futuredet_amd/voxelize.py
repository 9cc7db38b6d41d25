"""Voxelization stage on the HIP voxelizer, behind the reference's three entry points:

points_to_voxel(points, voxel_size, coors_range, max_points=35, reverse_index=True, max_voxels=20000)
        det3d/ops/point_cloud/point_cloud_ops.py:112-184
VoxelGenerator(voxel_size, point_cloud_range, max_num_points, max_voxels).generate(points, max_voxels)
        det3d/core/input/voxel_generator.py:5-46
Voxelization(cfg=voxel_generator_cfg)(res, info)     det3d/datasets/pipelines/preprocess.py:226-271

numpy in -> numpy out (host round trip, like the reference API); device tensor in -> device tensors out.
"""
import numpy as np
import torch

from . import hip_ops
from .registry import PIPELINES


def points_to_voxel(points, voxel_size, coors_range, max_points=35, reverse_index=True, max_voxels=20000, device=None):
    if not reverse_index:
        raise NotImplementedError("only reverse_index=True is on the path (voxel_generator.py:28)")
    as_numpy = isinstance(points, np.ndarray)
    if as_numpy:
        dev = torch.device(device or "cuda")
        pts = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).to(dev)
    else:
        pts = points.float().contiguous()
    out = hip_ops.voxelize(pts, np.asarray(voxel_size, np.float32), np.asarray(coors_range, np.float32), int(max_points),
                           int(max_voxels))
    m = int(out["num_voxels"].cpu()[0])
    voxels, coors, num = out["voxels"][:m], out["coors"][:m], out["num_points"][:m]
    if as_numpy:
        return voxels.cpu().numpy(), coors.cpu().numpy(), num.cpu().numpy()
    return voxels, coors, num


def grid_cells(point_cloud_range, voxel_size):
    """Cells per axis (x, y, z) of a voxel grid: round((hi - lo) / size) evaluated in float32, the rule fd_voxelize applies to the
    same two arrays (include/futuredet_hip.h; point_cloud_ops.py:24-29)."""
    r = np.asarray(point_cloud_range, dtype=np.float32)
    v = np.asarray(voxel_size, dtype=np.float32)
    return np.round((r[3:] - r[:3]) / v).astype(np.int64)


class VoxelGenerator(object):
    """Host-side handle of one voxelizer configuration: generate() is points_to_voxel on the HIP voxelizer with the stored geometry.
    The four read-only attributes are the ones det3d's pipeline stages and configs read (det3d/core/input/voxel_generator.py:5-46)."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000):
        self._spec = dict(voxel_size=np.asarray(voxel_size, dtype=np.float32), point_cloud_range=np.asarray(point_cloud_range, dtype=np.float32),
                          max_num_points_per_voxel=int(max_num_points), grid_size=grid_cells(point_cloud_range, voxel_size))
        self._max_voxels = int(max_voxels)

    def __getattr__(self, name):  # voxel_size, point_cloud_range, max_num_points_per_voxel, grid_size
        spec = self.__dict__.get("_spec", {})
        if name in spec:
            return spec[name]
        raise AttributeError(name)

    def generate(self, points, max_voxels=-1):
        cap = self._max_voxels if max_voxels == -1 else int(max_voxels)
        sp = self._spec
        return points_to_voxel(points, sp["voxel_size"], sp["point_cloud_range"], sp["max_num_points_per_voxel"], True, cap)


@PIPELINES.register_module
class Voxelization(object):
    def __init__(self, **kwargs):
        cfg = kwargs.get("cfg", None)
        self.range = cfg["range"]
        self.voxel_size = cfg["voxel_size"]
        self.max_points_in_voxel = cfg["max_points_in_voxel"]
        mv = cfg["max_voxel_num"]
        self.max_voxel_num = [mv, mv] if isinstance(mv, int) else mv
        self.double_flip = cfg.get("double_flip", False)
        if self.double_flip:
            raise NotImplementedError("DOUBLE_FLIP is False in every shipped config")
        self.voxel_generator = VoxelGenerator(voxel_size=self.voxel_size, point_cloud_range=self.range,
                                              max_num_points=self.max_points_in_voxel, max_voxels=self.max_voxel_num[0])

    def __call__(self, res, info):
        vg = self.voxel_generator
        max_voxels = self.max_voxel_num[0] if res["mode"] == "train" else self.max_voxel_num[1]
        voxels, coordinates, num_points = vg.generate(res["lidar"]["points"], max_voxels=max_voxels)
        num_voxels = np.array([voxels.shape[0]], dtype=np.int64)
        res["lidar"]["voxels"] = dict(voxels=voxels, coordinates=coordinates, num_points=num_points, num_voxels=num_voxels,
                                      shape=vg.grid_size, range=vg.point_cloud_range, size=vg.voxel_size)
        return res, info


# names the shipped pipelines reference; they belong to dataset I/O / training and only need to resolve
for _name in ("LoadPointCloudAnnotations", "Preprocess", "AssignLabel", "Reformat", "DoubleFlip",
              "Empty"):
    def _make(name):
        def __init__(self, **kwargs):
            self.kwargs = kwargs

        def __call__(self, res, info):
            if name == "Empty":
                return res, info
            raise NotImplementedError("pipeline stage %s (dataset I/O / training) is outside the hot path" % name)

        return type(name, (object,), {"__init__": __init__, "__call__": __call__})
    PIPELINES.register_module(_make(_name))
