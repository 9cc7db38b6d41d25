"""The few det3d.torchie.apis names the inference entry point imports next to batch_processor
(tools/dist_test.py:19-32): logging, seeding and process-group set-up are real; the training-only ones resolve and
raise with the reason when called (training is outside the hot path)."""
import logging
import random

import numpy as np
import torch

from . import dist_infer
from .collate import batch_processor, example_to_device  # noqa: F401  (det3d/torchie/apis/train.py:28-71,106-126)


def get_root_logger(log_level=logging.INFO):
    """det3d/torchie/apis/env.py: root logger at ``log_level`` on rank 0, ERROR elsewhere."""
    logger = logging.getLogger()
    if not logger.hasHandlers():
        logging.basicConfig(format="%(asctime)s - %(levelname)s - %(message)s", level=log_level)
    rank, _ = dist_infer.get_dist_info()
    logger.setLevel(log_level if rank == 0 else logging.ERROR)
    return logger


def set_random_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def init_dist(launcher="pytorch", backend="nccl", **kwargs):
    """det3d/torchie/apis/env.py init_dist('pytorch'): one process per GPU from the torchrun environment
    (backend "nccl" is RCCL on ROCm)."""
    if launcher != "pytorch":
        raise NotImplementedError("only the torch.distributed launcher is supported (tools/dist_test.py uses env://)")
    return dist_infer.init_from_env(backend)


def _training_only(name):
    def fn(*args, **kwargs):
        raise NotImplementedError("%s belongs to the training loop, which is outside the inference hot path (SURVEY 2)" % name)

    fn.__name__ = name
    return fn


build_optimizer = _training_only("build_optimizer")
train_detector = _training_only("train_detector")
