"""``det3d`` import names for callers written against the reference.

The reference's config files do ``from det3d.utils.config_tool import get_downsample_factor`` and its tools do
``from det3d.torchie import Config`` / ``from det3d.models import build_detector``.  install_det3d_alias()
registers lightweight alias modules under those names that point at this package, so such code (and the
reference's ``configs/centerpoint/*.py``, unchanged) runs against the MI355X implementation.  It never
shadows a real ``det3d`` package that is already imported.
"""
import sys
import types


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__futuredet_amd_alias__ = True
    sys.modules[name] = m
    parent, _, leaf = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], leaf, m)
    return m


def install_det3d_alias(force=False):
    existing = sys.modules.get("det3d")
    if existing is not None and not getattr(existing, "__futuredet_amd_alias__", False) and not force:
        return False
    if existing is not None and getattr(existing, "__futuredet_amd_alias__", False):
        return True
    from . import apis, collate, config, config_tool, detectors, dist_infer, nms, registry, voxelize
    from . import backbones, heads, necks, readers  # noqa: F401  (populate the registries)

    _mod("det3d")
    _mod("det3d.utils", Registry=registry.Registry, build_from_cfg=registry.build_from_cfg)
    _mod("det3d.utils.config_tool", get_downsample_factor=config_tool.get_downsample_factor)
    _mod("det3d.utils.registry", Registry=registry.Registry, build_from_cfg=registry.build_from_cfg)
    _mod("det3d.torchie", Config=config.Config, ConfigDict=config.ConfigDict)
    _mod("det3d.torchie.utils", Config=config.Config, ConfigDict=config.ConfigDict)
    _mod("det3d.torchie.utils.config", Config=config.Config, ConfigDict=config.ConfigDict)
    _mod("det3d.torchie.trainer", load_checkpoint=detectors.load_checkpoint, get_dist_info=dist_infer.get_dist_info)
    _mod("det3d.torchie.trainer.utils", all_gather=dist_infer.all_gather, synchronize=dist_infer.synchronize,
         get_dist_info=dist_infer.get_dist_info)
    _mod("det3d.torchie.trainer.checkpoint", load_checkpoint=detectors.load_checkpoint)
    # tools/dist_test.py:141,177,220 reach the path through these two (det3d/torchie/apis/train.py:28-71,106-126)
    _mod("det3d.torchie.apis", batch_processor=collate.batch_processor, example_to_device=collate.example_to_device,
         get_root_logger=apis.get_root_logger, set_random_seed=apis.set_random_seed, init_dist=apis.init_dist,
         build_optimizer=apis.build_optimizer, train_detector=apis.train_detector)
    _mod("det3d.torchie.apis.train", batch_processor=collate.batch_processor, example_to_device=collate.example_to_device)
    _mod("det3d.torchie.parallel")
    _mod("det3d.torchie.parallel.collate", collate_kitti_multi=collate.collate_kitti_multi, collate_kitti=collate.collate_kitti_multi)
    names = ("READERS", "BACKBONES", "NECKS", "HEADS", "LOSSES", "DETECTORS", "SECOND_STAGE", "ROI_HEAD")
    regs = {n: getattr(registry, n) for n in names}
    builders = {n: getattr(registry, n) for n in ("build_reader", "build_backbone", "build_neck", "build_head",
                                                  "build_loss", "build_detector", "build")}
    _mod("det3d.models", **regs, **builders)
    _mod("det3d.models.registry", **regs)
    _mod("det3d.models.builder", **builders)
    _mod("det3d.datasets", PIPELINES=registry.PIPELINES, DATASETS=registry.DATASETS,
         build_dataset=apis._training_only("build_dataset (dataset / devkit code"), build_dataloader=apis._training_only("build_dataloader (dataset code"))
    _mod("det3d.datasets.registry", PIPELINES=registry.PIPELINES, DATASETS=registry.DATASETS)
    _mod("det3d.core")
    _mod("det3d.core.input")
    _mod("det3d.core.input.voxel_generator", VoxelGenerator=voxelize.VoxelGenerator)
    _mod("det3d.core.bbox")
    _mod("det3d.core.bbox.box_torch_ops", rotate_nms_pcdet=nms.rotate_nms_pcdet)
    _mod("det3d.ops")
    _mod("det3d.ops.point_cloud")
    _mod("det3d.ops.point_cloud.point_cloud_ops", points_to_voxel=voxelize.points_to_voxel)
    _mod("det3d.ops.iou3d_nms")
    _mod("det3d.ops.iou3d_nms.iou3d_nms_cuda", nms_gpu=nms.nms_gpu, boxes_iou_bev_gpu=nms.boxes_iou_bev_gpu)
    return True
