"""Programmatic equivalents of the reference's configs/centerpoint/*_detection.py model / test / voxel settings.

The reference's config FILES load unchanged through Config.fromfile (tests check that against the parsed
fixtures in tests/golden/configs.json); those files are not part of this repository, so benchmarks and GPU
tests build the same dictionaries here.  Values: configs/centerpoint/nusc_centerpoint_forecast_n0_detection.py
:6-15 (flags), :32-73 (model), :88-103 (test_cfg), :160-166 (voxel_generator)."""
import itertools
import logging

from .config import ConfigDict
from .config_tool import get_downsample_factor

VARIANTS = {
    # name: (timesteps, DENSE, FORECAST_FEATS, BEV_MAP)
    "forecast_n0": (1, False, False, False),
    "forecast_n3": (7, False, False, False),
    "forecast_n3dtf": (7, True, True, False),
    "forecast_n3dtfm": (7, True, True, True),
}


def centerpoint_config(variant="forecast_n0", class_name="car", voxel_size=(0.075, 0.075, 0.2),
                       pc_range=(-54, -54, -5.0, 54, 54, 3.0), max_voxel_num=(120000, 160000)):
    timesteps, dense, ff, bev = VARIANTS[variant]
    tasks = [dict(num_class=1, class_names=[class_name])]
    model = dict(
        type="VoxelNet", pretrained=None,
        reader=dict(type="VoxelFeatureExtractorV3", num_input_features=5),
        backbone=dict(type="SpMiddleResNetFHD", num_input_features=5, ds_factor=8),
        neck=dict(type="RPN", layer_nums=[5, 5], ds_layer_strides=[1, 2], ds_num_filters=[128, 256],
                  us_layer_strides=[1, 2], us_num_filters=[256, 256], num_input_features=256,
                  logger=logging.getLogger("RPN")),
        bbox_head=dict(type="CenterHead", in_channels=sum([256, 256]), tasks=tasks, dataset="nuscenes", weight=0.25,
                       code_weights=[1.0] * 10 if dense else [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2, 1.0, 1.0],
                       common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2)},
                       share_conv_channel=64, dcn_head=False, timesteps=timesteps, two_stage=False, reverse=False,
                       sparse=False, dense=dense, bev_map=bev, forecast_feature=ff, classify=False, wide_head=False))
    osf = get_downsample_factor(model)
    test_cfg = dict(post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], max_per_img=500,
                    nms=dict(use_rotate_nms=True, use_multi_class_nms=False, nms_pre_max_size=1000, nms_post_max_size=83,
                             nms_iou_threshold=0.2),
                    score_threshold=0.1, pc_range=list(pc_range[:2]), out_size_factor=osf,
                    voxel_size=list(voxel_size[:2]), double_flip=False)
    voxel_generator = dict(range=list(pc_range), voxel_size=list(voxel_size), max_points_in_voxel=10,
                           max_voxel_num=list(max_voxel_num), double_flip=False)
    return ConfigDict(timesteps=timesteps, tasks=tasks, class_names=list(itertools.chain(*[t["class_names"] for t in tasks])),
                      model=model, test_cfg=test_cfg, voxel_generator=voxel_generator, TWO_STAGE=False, DOUBLE_FLIP=False,
                      DENSE=dense, BEV_MAP=bev, FORECAST_FEATS=ff)


def pointpillars_config(class_name="car", voxel_size=(0.2, 0.2, 8), pc_range=(-51.2, -51.2, -5.0, 51.2, 51.2, 3.0),
                        max_voxel_num=(30000, 60000)):
    """configs/centerpoint/nusc_centerpoint_pp_forecast_n3dtf_detection.py (:6-15 flags, :33-79 model, :95-107 test_cfg,
    :165-170 voxel_generator) and its pedestrian twin: PointPillars reader/scatter + 3-stage RPN + the n3dtf head."""
    timesteps, dense, ff, bev = 7, True, True, False
    tasks = [dict(num_class=1, class_names=[class_name])]
    model = dict(
        type="PointPillars", pretrained=None,
        reader=dict(type="PillarFeatureNet", num_filters=[64, 64], num_input_features=5, with_distance=False,
                    voxel_size=tuple(voxel_size), pc_range=tuple(pc_range)),
        backbone=dict(type="PointPillarsScatter", ds_factor=1),
        neck=dict(type="RPN", layer_nums=[3, 5, 5], ds_layer_strides=[2, 2, 2], ds_num_filters=[64, 128, 256],
                  us_layer_strides=[0.5, 1, 2], us_num_filters=[128, 128, 128], num_input_features=64,
                  logger=logging.getLogger("RPN")),
        bbox_head=dict(type="CenterHead", in_channels=sum([128, 128, 128]), tasks=tasks, dataset="nuscenes", weight=0.25,
                       code_weights=[1.0] * 10,
                       common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2)},
                       share_conv_channel=64, dcn_head=False, timesteps=timesteps, two_stage=False, reverse=False,
                       sparse=False, dense=dense, bev_map=bev, forecast_feature=ff, classify=False, wide_head=False))
    osf = get_downsample_factor(model)
    test_cfg = dict(post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], max_per_img=500,
                    nms=dict(nms_pre_max_size=1000, nms_post_max_size=83, nms_iou_threshold=0.2),
                    score_threshold=0.1, pc_range=list(pc_range[:2]), out_size_factor=osf, voxel_size=list(voxel_size[:2]))
    voxel_generator = dict(range=list(pc_range), voxel_size=list(voxel_size), max_points_in_voxel=20,
                           max_voxel_num=list(max_voxel_num))
    return ConfigDict(timesteps=timesteps, tasks=tasks, class_names=list(itertools.chain(*[t["class_names"] for t in tasks])),
                      model=model, test_cfg=test_cfg, voxel_generator=voxel_generator, TWO_STAGE=False, DOUBLE_FLIP=False,
                      DENSE=dense, BEV_MAP=bev, FORECAST_FEATS=ff)
