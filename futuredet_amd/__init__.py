"""futuredet_amd: MI355X (gfx950) implementation of the FutureDet LiDAR inference hot path
(voxelize -> mean reader -> SpMiddleResNetFHD sparse backbone -> RPN -> CenterHead -> decode + rotated NMS)
behind the reference's det3d registry / config surface.  HIP kernels live in csrc/ behind the C ABI in
include/futuredet_hip.h; this package is the Python host side."""
from . import registry  # noqa: F401
from .config import Config, ConfigDict  # noqa: F401
from .config_tool import get_downsample_factor  # noqa: F401
from .registry import (BACKBONES, DETECTORS, HEADS, NECKS, PIPELINES, READERS, build_backbone, build_detector,  # noqa: F401
                       build_from_cfg, build_head, build_neck, build_reader)
from . import readers, backbones, necks, heads, detectors, voxelize, loading  # noqa: F401,E402  (populate registries)

__version__ = "0.1.0"
