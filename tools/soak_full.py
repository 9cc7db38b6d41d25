"""Determinism soak of FutureDet end to end (detectors.FullSweepStep: sweep assembly -> sweep -> forecast association in one graph): NS
captured steps in flight on NS streams, every replay's packed detections AND forecast blob (global-frame boxes, chains, trajectory list,
forecast ids -- float64 arithmetic) compared with the stream's first.      python tools/soak_full.py [fp32|bf16] [rounds] [in flight]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from futuredet_amd import build_detector, hip_ops  # noqa: E402
from futuredet_amd.configs import centerpoint_config  # noqa: E402
from futuredet_amd.detectors import FullSweepStep  # noqa: E402
from futuredet_amd.synth import seeded_state_dict, synthetic_sweeps, tame_box_dims  # noqa: E402

dtype = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 200
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 4
cfg = centerpoint_config("forecast_n3" if dtype == torch.bfloat16 else "forecast_n0")
net = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
net.load_state_dict(tame_box_dims(seeded_state_dict(net, 7)), strict=False)
net = net.cuda().eval()
net.set_precision(dtype)
B, N_SWEEPS = 2, 10


def sample(seed):
    raw, rows, mats, lags, close = synthetic_sweeps(seed=seed, target_points=300000, n_sweeps=N_SWEEPS)
    desc = hip_ops.sweep_descriptors(rows, mats, lags, close)
    rng = np.random.default_rng([seed, 99])
    q1, q2 = rng.normal(0, 1, 4), rng.normal(0, 1, 4)
    rec = np.concatenate([q1 / np.linalg.norm(q1), rng.normal(0, 2, 3), q2 / np.linalg.norm(q2), rng.normal(0, 300, 3)])
    return dict(raw=torch.from_numpy(raw).cuda(), desc=torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).cuda(),
                time=torch.full((6,), 0.5, dtype=torch.float64, device="cuda"), records=torch.from_numpy(rec).cuda())


samples = [[sample(10 * s + b) for b in range(B)] for s in range(NS)]
cap = max(smp["raw"].shape[0] for ss in samples for smp in ss) + 1024
streams = [torch.cuda.Stream() for _ in range(NS)]
steps = []
with torch.no_grad():
    for s, st in enumerate(streams):
        with torch.cuda.stream(st):
            step = FullSweepStep(net, cfg.voxel_generator, cap, n_sweeps=N_SWEEPS, batch_size=B, classname="car", row_caps="auto")
            step.warm_up(samples[s])
            step.capture()
            steps.append(step)
    torch.cuda.synchronize()
    first, bad_det, bad_fc = [None] * NS, 0, 0
    for r in range(rounds):
        snaps = []
        for s, st in enumerate(streams):
            with torch.cuda.stream(st):
                packed, counts = steps[s](samples[s], check=False)
                snaps.append((packed.clone(), counts.clone(), steps[s].forecast.blob.clone(), steps[s].level_counts.clone()))
        torch.cuda.synchronize()
        for s, snap in enumerate(snaps):
            assert not steps[s].overflowed(snap[3].cpu().tolist())
            if first[s] is None:
                first[s] = snap
                assert int(snap[1].sum()) > 0
            else:
                d = not (torch.equal(snap[0], first[s][0]) and torch.equal(snap[1], first[s][1]))
                f = not torch.equal(snap[2], first[s][2])
                bad_det += d
                bad_fc += f and not d
                if d or f:
                    print("round %d stream %d: detections %s, forecast blob %s (%d bytes differ)" % (r, s, "DIFFER" if d else "equal", "DIFFERS" if f else "equal",
                                                                                                       int((snap[2] != first[s][2]).sum())))
    print("%s full pipeline: %d rounds x %d in flight; replays with differing detections %d, with equal detections but a differing forecast blob %d" % (
        str(dtype).split(".")[-1], rounds, NS, bad_det, bad_fc))
    sys.exit(1 if (bad_det or bad_fc) else 0)
