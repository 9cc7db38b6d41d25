import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))
        return cache[name]

    return load


@pytest.fixture(scope="session")
def hip():
    """HIP front end; the GPU tests must run on the native library, never on a fallback."""
    import torch

    assert torch.cuda.is_available(), "gpu-marked test running without a GPU"
    from futuredet_amd import build, hip_ops, lib

    build.build()
    lib.load()
    return hip_ops
