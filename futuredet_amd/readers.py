"""Readers (det3d/models/readers).  VoxelFeatureExtractorV3: per-voxel mean of the point slots."""
import torch
from torch import nn

from . import hip_ops
from .nn_utils import bn_affine, build_norm_layer, weights_version
from .registry import READERS


@READERS.register_module
class VoxelFeatureExtractorV3(nn.Module):
    """det3d/models/readers/voxel_encoder.py:8-24.  When the voxelizer already produced means (fused kernel),
    ``features`` arrives as [M, C] and is passed through."""

    def __init__(self, num_input_features=4, norm_cfg=None, name="VoxelFeatureExtractorV3"):
        super().__init__()
        self.name = name
        self.num_input_features = num_input_features

    def forward(self, features, num_voxels, coors=None):
        assert self.num_input_features == features.shape[-1]
        if features.dim() == 2:  # fused voxelizer output: already the mean
            return features
        points_mean = features[:, :, : self.num_input_features].sum(dim=1, keepdim=False) / \
            num_voxels.type_as(features).view(-1, 1)
        return points_mean.contiguous()


class PFNLayer(nn.Module):
    """Parameter holder with the state_dict keys of det3d/models/readers/pillar_encoder.py:15-55 (linear.weight,
    norm.*).  The layer itself (Linear, BatchNorm, ReLU, max over the pillar's points, concat) is evaluated inside
    fd_pillar_encode; there is no per-layer torch forward."""

    def __init__(self, in_channels, out_channels, norm_cfg=None, last_layer=False):
        super().__init__()
        self.name = "PFNLayer"
        self.last_vfe = last_layer
        if not self.last_vfe:
            out_channels = out_channels // 2
        self.units = out_channels
        if norm_cfg is None:
            norm_cfg = dict(type="BN1d", eps=1e-3, momentum=0.01)
        self.norm_cfg = norm_cfg
        self.linear = nn.Linear(in_channels, self.units, bias=False)
        self.norm = build_norm_layer(self.norm_cfg, self.units)[1]


@READERS.register_module
class PillarFeatureNet(nn.Module):
    """det3d/models/readers/pillar_encoder.py:58-164.  In eval mode on the GPU the whole reader is one HIP launch
    (fd_pillar_encode); the torch modules only define the parameters / state_dict.  Training is outside this path."""

    def __init__(self, num_input_features=4, num_filters=(64,), with_distance=False, voxel_size=(0.2, 0.2, 4),
                 pc_range=(0, -40, -3, 70.4, 40, 1), norm_cfg=None):
        super().__init__()
        self.name = "PillarFeatureNet"
        assert len(num_filters) > 0
        self.num_input = num_input_features
        num_input_features += 5
        if with_distance:
            num_input_features += 1
        self._with_distance = with_distance
        num_filters = [num_input_features] + list(num_filters)
        self.pfn_layers = nn.ModuleList([
            PFNLayer(num_filters[i], num_filters[i + 1], norm_cfg=norm_cfg, last_layer=(i >= len(num_filters) - 2))
            for i in range(len(num_filters) - 1)])
        self.vx = voxel_size[0]
        self.vy = voxel_size[1]
        self.x_offset = self.vx / 2 + pc_range[0]
        self.y_offset = self.vy / 2 + pc_range[1]
        self.compute_dtype = torch.float32
        self._packed = None
        self.register_load_state_dict_post_hook(_drop_packed)

    def _layers(self, device):
        key = (device, weights_version(self))
        if self._packed is None or self._packed[0] != key:
            if len(self.pfn_layers) > 2:
                raise NotImplementedError("fd_pillar_encode fuses one or two PFN layers (shipped configs: [64, 64])")
            layers = []
            for pfn in self.pfn_layers:
                scale, shift = bn_affine(pfn.norm)
                layers.append((pfn.linear.weight.detach().float().contiguous().to(device), scale.contiguous().to(device),
                               shift.contiguous().to(device)))
            self._packed = (key, layers)
        return self._packed[1]

    def forward(self, features, num_voxels, coors, n_dev=None):
        if self.training:
            raise NotImplementedError("PillarFeatureNet runs as one fused inference kernel (fd_pillar_encode); training is outside "
                                      "the hot path (SURVEY 2)")
        return hip_ops.pillar_encode(features, num_voxels.int(), coors.int().contiguous(), n_dev,
                                     (self.vx, self.vy, self.x_offset, self.y_offset), self._layers(features.device),
                                     with_distance=self._with_distance, out_dtype=self.compute_dtype)


def _drop_packed(module, incompatible_keys=None):
    module._packed = None
    module.__dict__.pop("_wv_tensors", None)
