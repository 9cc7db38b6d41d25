"""get_downsample_factor, imported by every centerpoint config file
(configs/centerpoint/*:4); behaviour of det3d/utils/config_tool.py:39-53."""
import numpy as np


def get_downsample_factor(model_config):
    try:
        neck_cfg = model_config["neck"]
    except Exception:
        model_config = model_config["first_stage_cfg"]
        neck_cfg = model_config["neck"]
    factor = np.prod(neck_cfg.get("ds_layer_strides", [1]))
    us = neck_cfg.get("us_layer_strides", [])
    if len(us) > 0:
        factor = factor / us[-1]
    factor = int(factor * model_config["backbone"]["ds_factor"])
    assert factor > 0
    return factor
