"""Readers (det3d/models/readers).  VoxelFeatureExtractorV3: per-voxel mean of the point slots."""
import torch
from torch import nn

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


# The two pp configs name these types; they only need to resolve so the config files load and the registry
# lookup gives a clear message -- the PointPillars path is outside the VoxelNet hot path.
@READERS.register_module
class PillarFeatureNet(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("PillarFeatureNet (PointPillars reader) is outside the VoxelNet hot path")
