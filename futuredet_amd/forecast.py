"""Forecast association consuming the head output, mirroring det3d/datasets/nuscenes/nuscenes.py:

match_boxes(ret_boxes)              : nuscenes.py:112-123
tracker(classname, time, ret_boxes) : nuscenes.py:125-257 (forward chains, constant-velocity forward, back-cast chains)

``ret_boxes`` is what the reference builds at :398-409: one list per forecast step of box objects with ``.center`` and
``.velocity`` (nuScenes-devkit ``Box`` in the reference; any object with those two array attributes works).  The
nearest-centre matchings, the chain walks and the constant-velocity extrapolation run in one HIP launch
(fd_forecast_chains, float64 like numpy); this module only moves the ≤ 83 x T centres to the device and rebuilds the
Python lists.  No CPU fallback.
"""
from copy import deepcopy

import numpy as np
import torch

from . import hip_ops


def _pack(ret_boxes):
    T = len(ret_boxes)
    n_max = max(1, max(len(b) for b in ret_boxes))
    centers = np.zeros((T, n_max, 3), np.float64)
    velocity = np.zeros((T, n_max, 3), np.float64)
    counts = np.zeros((T,), np.int32)
    for t, boxes in enumerate(ret_boxes):
        counts[t] = len(boxes)
        for j, box in enumerate(boxes):
            centers[t, j] = np.asarray(box.center, np.float64)[:3]
            v = np.asarray(box.velocity, np.float64)
            velocity[t, j, :len(v[:3])] = v[:3]
    return centers, velocity, counts


def associate(ret_boxes, time, reject_thresh, device="cuda"):
    """-> dict of host arrays: fwd_idx, fwd_ok, bwd_idx, bwd_ok, match_idx, cv_centers, status (see fd_forecast_chains)."""
    centers, velocity, counts = _pack(ret_boxes)
    out = hip_ops.forecast_chains(torch.from_numpy(centers).to(device), torch.from_numpy(velocity).to(device),
                                  torch.from_numpy(counts).to(device),
                                  torch.tensor([float(t) for t in time], dtype=torch.float64, device=device), float(reject_thresh))
    return {k: v.cpu().numpy() for k, v in out.items()}, counts


def match_boxes(ret_boxes):
    T = len(ret_boxes)
    if T < 2 or len(ret_boxes[0]) == 0:
        return [np.array(b) for b in ret_boxes]
    res, counts = associate(ret_boxes, [0.0] * (T - 1), 0.0)
    n0 = int(counts[0])
    return [np.array(box)[res["match_idx"][t, :n0]] for t, box in enumerate(ret_boxes)]


def tracker(classname, time, ret_boxes):
    reject_thresh = 2 if classname == "car" else 1          # nuscenes.py:126-132
    trajectory = []
    if classname not in ["car", "pedestrian"]:
        return trajectory
    T = len(ret_boxes)
    res, counts = associate(ret_boxes, time, reject_thresh)
    if int(res["status"][0]):                               # some step is empty: :157-158 / :219-220
        return []
    for i in range(int(counts[0])):                         # forecasting chains, :160-180
        if res["fwd_ok"][i]:
            trajectory.append([ret_boxes[t][int(res["fwd_idx"][i, t])] for t in range(T)])
    for i in range(int(counts[0])):                         # constant velocity forward, :183-193
        forecast = [ret_boxes[0][i]]
        for t in range(1, T):
            new_box = deepcopy(forecast[-1])
            new_box.center = res["cv_centers"][i, t].copy()
            forecast.append(new_box)
        trajectory.append(forecast)
    for i in range(int(counts[T - 1])):                     # back-casting chains, :196-241
        if res["bwd_ok"][i]:
            chain = [ret_boxes[T - 1 - s][int(res["bwd_idx"][i, s])] for s in range(T)]
            trajectory.append(chain[::-1])
    return trajectory
