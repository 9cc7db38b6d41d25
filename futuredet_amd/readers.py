"""Readers (det3d/models/readers).  VoxelFeatureExtractorV3: per-voxel mean of the point slots."""
import torch
import torch.nn.functional as F
from torch import nn

from . import hip_ops
from .nn_utils import bn_affine, build_norm_layer
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
    """det3d/models/readers/pillar_encoder.py:15-55 (state_dict: linear.weight, norm.*)."""

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

    def forward(self, inputs):
        x = self.linear(inputs)
        x = self.norm(x.permute(0, 2, 1).contiguous()).permute(0, 2, 1).contiguous()
        x = F.relu(x)
        x_max = torch.max(x, dim=1, keepdim=True)[0]
        if self.last_vfe:
            return x_max
        return torch.cat([x, x_max.repeat(1, inputs.shape[1], 1)], dim=2)


@READERS.register_module
class PillarFeatureNet(nn.Module):
    """det3d/models/readers/pillar_encoder.py:58-164.  In eval mode on the GPU the whole reader is one HIP launch
    (fd_pillar_encode); the torch modules define the parameters / state_dict and serve training."""

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

    def forward_modules(self, features, num_voxels, coors):
        dtype = features.dtype
        points_mean = features[:, :, :3].sum(dim=1, keepdim=True) / num_voxels.type_as(features).view(-1, 1, 1)
        f_cluster = features[:, :, :3] - points_mean
        f_center = torch.zeros_like(features[:, :, :2])
        f_center[:, :, 0] = features[:, :, 0] - (coors[:, 3].to(dtype).unsqueeze(1) * self.vx + self.x_offset)
        f_center[:, :, 1] = features[:, :, 1] - (coors[:, 2].to(dtype).unsqueeze(1) * self.vy + self.y_offset)
        features_ls = [features, f_cluster, f_center]
        if self._with_distance:
            features_ls.append(torch.norm(features[:, :, :3], 2, 2, keepdim=True))
        features = torch.cat(features_ls, dim=-1)
        voxel_count = features.shape[1]
        mask = (torch.arange(voxel_count, device=features.device).view(1, -1) < num_voxels.view(-1, 1).int())
        features = features * mask.unsqueeze(-1).type_as(features)
        for pfn in self.pfn_layers:
            features = pfn(features)
        return features.squeeze()

    def _layers(self, device):
        if self._packed is None or self._packed[0] != device:
            if len(self.pfn_layers) > 2:
                raise NotImplementedError("fd_pillar_encode fuses one or two PFN layers (shipped configs: [64, 64])")
            layers = []
            for pfn in self.pfn_layers:
                scale, shift = bn_affine(pfn.norm)
                layers.append((pfn.linear.weight.detach().float().contiguous().to(device), scale.contiguous().to(device),
                               shift.contiguous().to(device)))
            self._packed = (device, layers)
        return self._packed[1]

    def forward(self, features, num_voxels, coors, n_dev=None):
        if self.training:
            return self.forward_modules(features, num_voxels, coors)
        return hip_ops.pillar_encode(features, num_voxels.int(), coors.int().contiguous(), n_dev,
                                     (self.vx, self.vy, self.x_offset, self.y_offset), self._layers(features.device),
                                     with_distance=self._with_distance, out_dtype=self.compute_dtype)


def _drop_packed(module, incompatible_keys):
    module._packed = None
