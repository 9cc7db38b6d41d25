"""Robustness soak: many clouds of very different sizes through one detector instance (fp32 and bf16), checking that the
workspaces / plans / graphs survive changing shapes and that outputs stay finite."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from futuredet_amd import build_detector  # noqa: E402
from futuredet_amd.configs import centerpoint_config  # noqa: E402
from futuredet_amd.synth import seeded_state_dict, synthetic_cloud  # noqa: E402

cfg = centerpoint_config("forecast_n3")
net = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
net.load_state_dict(seeded_state_dict(net, 7), strict=False)
net = net.cuda().eval()
rng = np.random.default_rng(0)
sizes = [300000, 1000, 50, 120000, 500000, 7, 30000, 0, 250000, 3, 410000, 20000]
for dt in (torch.float32, torch.bfloat16):
    net.set_precision(dt)
    for i, n in enumerate(sizes):
        B = 1 + (i % 3 == 2)
        clouds = []
        for b in range(B):
            c = synthetic_cloud(seed=100 + i + b, target_points=max(n, 64))[: max(n, 0)] if n else np.zeros((0, 5), np.float32)
            clouds.append(torch.from_numpy(np.ascontiguousarray(c)).cuda())
        boxes, scores, labels, counts = net.forward_points(clouds, cfg.voxel_generator, padded=True)
        torch.cuda.synchronize()
        k = int(counts.sum())
        assert bool(torch.isfinite(scores).all()), (dt, n)
        print("%s n=%7d B=%d -> %4d detections" % (str(dt).split(".")[-1], n, B, k))
print("soak ok")
