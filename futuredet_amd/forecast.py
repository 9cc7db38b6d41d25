"""Forecast association consuming the head output, mirroring det3d/datasets/nuscenes/nuscenes.py and nusc_common.py:

match_boxes(ret_boxes)              : nuscenes.py:112-123
tracker(classname, time, ret_boxes) : nuscenes.py:125-257 (forward chains, constant-velocity forward, back-cast chains)
_second_det_to_nusc_box(detection)  : nusc_common.py:167-189   (head rows -> boxes: yaw flip, quaternion, velocity triple)
_lidar_nusc_box_to_global(...)      : nusc_common.py:192-216   (the two rigid transforms, records passed as arrays)
forecast_boxes(...)                 : nuscenes.py:384-494      (per-step split, matching / tracking, constant-velocity roll-out,
                                      jitter); the devkit token / time look-ups of :385-398 are the caller's: ``time`` and the
                                      calibrated_sensor / ego_pose records come in as arrays
multi_future(forecast_boxes, name)  : nuscenes.py:299-339      (forecast_id = connected component of the < 0.25 m graph)

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


# ------------------------------------------------------------------------------------------------ boxes from head rows
class Quat(object):
    """The little of pyquaternion.Quaternion the reference's forecasting code touches: .elements / indexing (w,x,y,z)."""

    def __init__(self, q):
        self.q = np.asarray(q, np.float64)

    @property
    def elements(self):
        return self.q

    def __getitem__(self, i):
        return self.q[i]


class Box(object):
    """Array-backed stand-in for nuscenes.utils.data_classes.Box with the attributes forecast_boxes / box_serialize read."""

    def __init__(self, center, size, orientation, label=np.nan, score=np.nan, velocity=(np.nan, np.nan, np.nan), name=None, token=None):
        self.center, self.wlh, self.orientation = np.array(center), np.array(size), orientation
        self.label, self.score, self.velocity, self.name, self.token = label, score, np.array(velocity), name, token


def _boxes_from_arrays(center, quat, vel, size, scores, labels):
    return [Box(center[i], size[i], Quat(quat[i]), label=labels[i], score=scores[i], velocity=vel[i]) for i in range(len(center))]


def det_arrays(detection, cs_record=None, pose_record=None, device="cuda"):
    """Array core of _second_det_to_nusc_box (+ _lidar_nusc_box_to_global when the records are given): one HIP launch
    (fd_det_to_global_boxes).  Returns host arrays (center f64 [n,3], quat f64 [n,4], velocity f64 [n,3], size f32 [n,3])."""
    box3d = detection["box3d_lidar"]
    box3d = box3d if isinstance(box3d, torch.Tensor) else torch.as_tensor(np.asarray(box3d))
    box3d = box3d.detach().to(device, torch.float32).contiguous()
    c, q, v, s = hip_ops.det_to_global_boxes(box3d, cs_record, pose_record)
    return c.cpu().numpy(), q.cpu().numpy(), v.cpu().numpy(), s.cpu().numpy()


def _second_det_to_nusc_box(detection, cs_record=None, pose_record=None):
    """nusc_common.py:167-189; with the two records also :192-216 (rotation = (w,x,y,z), translation = (x,y,z) each)."""
    c, q, v, s = det_arrays(detection, cs_record, pose_record)
    scores = detection["scores"].detach().cpu().numpy() if isinstance(detection["scores"], torch.Tensor) else np.asarray(detection["scores"])
    labels = detection["label_preds"].detach().cpu().numpy() if isinstance(detection["label_preds"], torch.Tensor) else np.asarray(detection["label_preds"])
    if cs_record is None and pose_record is None:
        c = c.astype(np.float32)  # the reference's Box keeps the float32 slice until its first rotation
    return _boxes_from_arrays(c, q, v, s, scores, labels)


def process_trajectories(ret_boxes, forecast, train_dist, device="cuda"):
    """nuscenes.py:341-382: every trajectory is replaced by the nearest row of the trajectory library ``train_dist``
    ([M, 2 + 4 + 3 (forecast - 1)]: start velocity xy, start orientation, offsets of the later centres from the first); the search
    (distance_matrix + argmin, :366-368) runs on the device (fd_nearest_rows).  Boxes are modified in place like the reference."""
    if len(ret_boxes) == 0:
        return []
    rows = []
    for ret_box in ret_boxes:  # :349-363
        box = ret_box[0]
        rotation = [box.orientation[0], box.orientation[1], box.orientation[2], box.orientation[3]]
        rows.append(np.array(list(box.velocity[:2]) + rotation + list(np.hstack([ret_box[i].center - box.center for i in range(1, forecast)]))))
    test_dist = np.ascontiguousarray(np.array(rows, np.float64))
    train = np.ascontiguousarray(np.asarray(train_dist, np.float64))
    idx = hip_ops.nearest_rows(torch.from_numpy(train).to(device), torch.from_numpy(test_dist).to(device)).cpu().numpy()
    out_boxes = []
    for ret_box, j in zip(ret_boxes, idx):  # :370-380
        translation = deepcopy(ret_box[0].center)
        trajectory = train[j][6:]
        for i in range(forecast - 1):
            ret_box[i + 1].center = translation + trajectory[3 * i:3 * i + 3]
        out_boxes.append(deepcopy(ret_box))
    return out_boxes


def forecast_boxes(det_forecast, time, cs_record, pose_record, forecast, forecast_mode, classname, jitter=False, K=1, C=0.0, stale=None,
                   train_dist=None, postprocess=False):
    """nuscenes.py:384-494 with the devkit look-ups factored out: ``time`` = seconds between consecutive forecast steps
    (get_time, :399-406), ``cs_record`` / ``pose_record`` = (rotation, translation) of the sample's LIDAR_TOP calibrated
    sensor and ego pose.  Returns ret_boxes: a list of trajectories, each a list of ``forecast`` boxes.
    ``postprocess`` (velocity_dense only, :465-467): trajectories are snapped to the library ``train_dist`` (process_trajectories)."""
    time = list(time)
    if stale is None:
        stale = any(t == 0 for t in time)
    labels = det_forecast["label_preds"]
    labels_np = labels.detach().cpu().numpy() if isinstance(labels, torch.Tensor) else np.asarray(labels)
    ret_boxes = []
    for t in range(forecast):  # :408-417
        mask = labels_np == t
        det = {k: det_forecast[k][torch.as_tensor(mask)] if isinstance(det_forecast[k], torch.Tensor) else np.asarray(det_forecast[k])[mask]
               for k in ("box3d_lidar", "scores", "label_preds")}
        ret_boxes.append(_second_det_to_nusc_box(det, cs_record, pose_record))
    if stale or len(ret_boxes[0]) == 0:
        return []
    if forecast_mode in ["velocity_constant", "velocity_forward", "velocity_reverse"]:
        ret_boxes = match_boxes(ret_boxes)
    elif forecast_mode in ["velocity_sparse_forward", "velocity_sparse_reverse", "velocity_sparse_match"]:
        # nuscenes.py:422-429: these modes index each Box as a [forward, reverse] pair (a two-head output that no shipped config
        # produces) and then fall into ``assert False, "Invalid Forecast Mode"`` (:468-469): the reference itself cannot execute
        # them (TypeError on the first Box, recorded in tests/golden/forecast2.npz::sparse_mode_exception).  Same outcome here.
        raise TypeError("forecast_mode %r is not executable in the reference either (nuscenes.py:422-429,468-469)" % forecast_mode)
    elif forecast_mode != "velocity_dense":
        raise AssertionError("Invalid Forecast Mode")  # :468-469
    if "dense" not in forecast_mode:  # :433-441
        trajectory_boxes = [[ret_boxes[i][j] for i in range(forecast)] for j in range(len(ret_boxes[0]))]
        if forecast_mode == "velocity_reverse":
            time = time[::-1]
        out = []
        for trajectory_box in trajectory_boxes:  # :445-463
            fboxes = [trajectory_box[0]]
            for i in range(forecast - 1):
                new_box = deepcopy(fboxes[-1])
                step = time[i] * trajectory_box[i].velocity
                new_box.center = new_box.center - step if forecast_mode == "velocity_reverse" else new_box.center + step
                fboxes.append(new_box)
            out.append(fboxes[::-1] if forecast_mode == "velocity_reverse" else fboxes)
        ret_boxes = out
    else:
        ret_boxes = tracker(classname, time, ret_boxes)  # :443,465-466
        if postprocess:
            ret_boxes = process_trajectories(ret_boxes, forecast, train_dist)
    if jitter:  # :476-492 (draws from numpy's global generator like the reference)
        jitter_boxes = []
        for trajectory_box in ret_boxes:
            for _ in range(K - 1):
                start_box = trajectory_box[0]
                vel_norm = C * np.linalg.norm(start_box.velocity)
                jittered_vel = np.random.normal(start_box.velocity, np.array([vel_norm, vel_norm, vel_norm]))
                fboxes = [start_box]
                for i in range(forecast - 1):
                    new_box = deepcopy(fboxes[-1])
                    new_box.center = new_box.center + time[i] * jittered_vel
                    fboxes.append(new_box)
                jitter_boxes.append(fboxes)
        ret_boxes = ret_boxes + jitter_boxes
    return ret_boxes


def forecast_ids(centers, match_thresh=0.25, device="cuda"):
    """Array core of multi_future: component id per box (fd_forecast_groups)."""
    centers = np.ascontiguousarray(np.asarray(centers, np.float64).reshape(-1, 3))
    if len(centers) == 0:
        return np.zeros((0,), np.int32)
    return hip_ops.forecast_groups(torch.from_numpy(centers).to(device), match_thresh).cpu().numpy()


def multi_future(forecast_boxes, classname):
    """nuscenes.py:299-339 on the serialised box dicts (keys translation / detection_name / detection_score /
    forecast_score / forecast_id / forecast_boxes), in place like the reference."""
    for sample_token in forecast_boxes.keys():
        boxes = [box for box in forecast_boxes[sample_token] if classname in box["detection_name"]]
        if len(boxes) == 0:
            continue
        ids = forecast_ids(np.array([box["translation"] for box in boxes]))
        for box, fid in zip(boxes, ids):
            box["forecast_id"] = int(fid)
            for sub in box["forecast_boxes"]:
                sub["detection_score"] = box["detection_score"]
                sub["forecast_score"] = box["forecast_score"]
                sub["forecast_id"] = int(fid)
        forecast_boxes[sample_token] = boxes
    return forecast_boxes


# ------------------------------------------------------------------------------------------------ whole batch, device-resident
def sweep_forecast(packed, counts, time, records=None, classname="car", out=None):
    """The device part of forecast_boxes(..., forecast_mode="velocity_dense") + multi_future for a batch: the head's packed output
    (``packed`` [B,T,post,11], ``counts`` [B,T], device tensors as CenterHead.predict / StaticStep return them) -> global-frame boxes ->
    chains -> trajectories and their forecast ids, all in HBM (hip_ops.ForecastOutputs; fd_forecast_from_detections, three launches,
    no synchronisation, capturable).  ``time`` [B,T-1] float64 and ``records`` [B,14] float64 (calibrated_sensor + ego_pose, see the
    header) are device tensors: the devkit look-ups that produce them are the caller's (nuscenes.py:385-406)."""
    reject = 2.0 if classname == "car" else 1.0  # nuscenes.py:126-132
    return hip_ops.forecast_from_detections(packed, counts, time, records, reject_thresh=reject, match_thresh=0.25, out=out)


def trajectories_from_arrays(h, b):
    """Rebuilds, from the host copy ``h`` of a ForecastOutputs (``.host()``), sample ``b``'s trajectories the way ``tracker`` returns them:
    a list of (kind, forecast_id, centres [T,3] float64, box index per step [T] or None for the constant-velocity roll-outs)."""
    T = h["center"].shape[1]
    out = []
    for j in range(int(h["n_traj"][b])):
        kind, src = int(h["traj_kind"][b, j]), int(h["traj_src"][b, j])
        if kind == 1:
            out.append((1, int(h["traj_group"][b, j]), h["cv_centers"][b, src].copy(), None))
            continue
        idx = h["fwd_idx"][b, src] if kind == 0 else h["bwd_idx"][b, src][::-1]
        out.append((kind, int(h["traj_group"][b, j]), np.stack([h["center"][b, t, int(idx[t])] for t in range(T)]), idx.copy()))
    return out
