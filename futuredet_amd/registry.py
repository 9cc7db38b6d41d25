"""Registry / builder surface of the hot path's plugin API.

Mirrors the reference's registry contract (det3d/utils/registry.py:5-78,
det3d/models/registry.py:3-10, det3d/models/builder.py:16-50): a class is
registered under its ``__name__`` and built from a config dict by popping
``"type"`` and calling the class with the remaining keys.
"""
import inspect

from torch import nn


class Registry(object):
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    def __repr__(self):
        return "{}(name={}, items={})".format(
            type(self).__name__, self._name, list(self._module_dict)
        )

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def register_module(self, cls):
        if not inspect.isclass(cls):
            raise TypeError("module must be a class, but got {}".format(type(cls)))
        key = cls.__name__
        if key in self._module_dict:
            raise KeyError("{} is already registered in {}".format(key, self._name))
        self._module_dict[key] = cls
        return cls


def build_from_cfg(cfg, registry, default_args=None):
    """registry.get(cfg["type"])(**rest) with ``default_args`` as fallbacks."""
    if not (isinstance(cfg, dict) and "type" in cfg):
        raise AssertionError("cfg must be a dict with a 'type' key")
    if not (default_args is None or isinstance(default_args, dict)):
        raise AssertionError("default_args must be a dict or None")
    kwargs = dict(cfg)
    kind = kwargs.pop("type")
    if isinstance(kind, str):
        cls = registry.get(kind)
        if cls is None:
            raise KeyError("{} is not in the {} registry".format(kind, registry.name))
    elif inspect.isclass(kind):
        cls = kind
    else:
        raise TypeError("type must be a str or valid type, but got {}".format(type(kind)))
    for k, v in (default_args or {}).items():
        kwargs.setdefault(k, v)
    return cls(**kwargs)


READERS = Registry("reader")
BACKBONES = Registry("backbone")
NECKS = Registry("neck")
HEADS = Registry("head")
LOSSES = Registry("loss")
DETECTORS = Registry("detector")
SECOND_STAGE = Registry("second_stage")
ROI_HEAD = Registry("roi_head")
PIPELINES = Registry("pipeline")
DATASETS = Registry("dataset")


def build(cfg, registry, default_args=None):
    if isinstance(cfg, list):
        return nn.Sequential(*[build_from_cfg(c, registry, default_args) for c in cfg])
    return build_from_cfg(cfg, registry, default_args)


def build_reader(cfg):
    return build(cfg, READERS)


def build_backbone(cfg):
    return build(cfg, BACKBONES)


def build_neck(cfg):
    return build(cfg, NECKS)


def build_head(cfg):
    return build(cfg, HEADS)


def build_loss(cfg):
    return build(cfg, LOSSES)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    return build(cfg, DETECTORS, dict(train_cfg=train_cfg, test_cfg=test_cfg))
