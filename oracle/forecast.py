"""CPU restatement of the forecast association (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Follows det3d/datasets/nuscenes/nuscenes.py: distance_matrix :100-110, box_center / box_past_center /
box_future_center :84-98, match_boxes :112-123, tracker :125-257 -- on plain arrays instead of devkit Box objects:
centers / velocity are lists (one entry per forecast step) of [n_t, 3] float64 arrays.
"""
import numpy as np


def distance_matrix(A, B):
    M, N = A.shape[0], B.shape[0]
    A_dots = (A * A).sum(axis=1).reshape((M, 1)) * np.ones(shape=(1, N))
    B_dots = (B * B).sum(axis=1) * np.ones(shape=(M, 1))
    D = A_dots + B_dots - 2 * A.dot(B.T)
    D[np.less(D, 0.0)] = 0.0
    return np.sqrt(D)


def match_indices(centers):
    c0 = centers[0][:, :2]
    return [np.argmin(distance_matrix(c0, c[:, :2]), axis=1) for c in centers]


def _chains(centers, velocity, time, reject, sign):
    idx, dist = [], []
    for s, tm in enumerate(time):
        cur, nxt = centers[s], centers[s + 1]
        if len(cur) == 0 or len(nxt) == 0:
            continue
        moved = cur[:, :2] + sign * tm * velocity[s][:, :2]
        D = distance_matrix(moved, nxt[:, :2])
        idx.append(np.argmin(D, axis=1))
        dist.append(np.min(D, axis=1))
    if len(idx) != len(centers) - 1:
        return None
    out = []
    for i in range(idx[0].shape[0]):
        chain, void = [i], False
        for ind, dis in zip(idx, dist):
            if dis[chain[-1]] > reject:
                void = True
            chain.append(int(ind[chain[-1]]))
        if not void:
            out.append(chain)
    return out


def tracker(classname, time, centers, velocity):
    """-> (forward chains, constant-velocity centres [n0, T, 3], back-cast chains in chronological order) or None when
    the reference returns []."""
    if classname not in ("car", "pedestrian"):
        return None
    reject = 2 if classname == "car" else 1
    fwd = _chains(centers, velocity, list(time), reject, +1.0)
    if fwd is None:
        return None
    T = len(centers)
    cv = np.zeros((len(centers[0]), T, 3))
    for i in range(len(centers[0])):
        c = centers[0][i].copy()
        cv[i, 0] = c
        for s, t in enumerate(time):
            c = c + t * velocity[0][i]
            cv[i, s + 1] = c
    bwd = _chains(centers[::-1], velocity[::-1], list(time)[::-1], reject, -1.0)
    if bwd is None:
        return None
    return fwd, cv, [ch[::-1] for ch in bwd]


# ---- head rows -> boxes -> global frame (nusc_common.py:167-216) and multi_future's grouping (nuscenes.py:283-339),
#      restated on arrays.  pyquaternion / the devkit Box are absent from the image: their published arithmetic is
#      restated here (Quaternion(axis z, a) = (cos a/2, 0, 0, sin a/2); rotation_matrix = (Q.Qbar^T)[1:,1:] of the
#      normalised quaternion; q*o = Q(q).o), pinned by tests/golden/forecast2.npz, which comes from the reference's own
#      functions run over the same restatement of those two libraries.
def _qmat(q):
    w, x, y, z = q
    return np.array([[w, -x, -y, -z], [x, w, -z, y], [y, z, w, -x], [z, -y, x, w]])


def _qbar(q):
    w, x, y, z = q
    return np.array([[w, -x, -y, -z], [x, w, z, -y], [y, -z, w, x], [z, y, -x, w]])


def det_to_boxes(box3d):
    """-> center [n,3] (float32 values), quat [n,4], velocity [n,3], size [n,3]; nusc_common.py:167-189"""
    import math
    box3d = np.array(box3d, np.float32)
    yaw = -box3d[:, -1] - np.float32(np.pi / 2)
    quat = np.array([[math.cos(float(a) / 2.0), 0.0, 0.0, math.sin(float(a) / 2.0)] for a in yaw], np.float64).reshape(-1, 4)
    vel = np.concatenate([box3d[:, 6:8].astype(np.float64), np.zeros((len(box3d), 1))], axis=1)
    return box3d[:, :3].astype(np.float64), quat, vel, box3d[:, 3:6]


def boxes_to_global(center, quat, vel, records):
    """records = [(rotation wxyz, translation xyz), ...] applied in order (calibrated_sensor, ego_pose); :192-216"""
    center, quat, vel = center.copy(), quat.copy(), vel.copy()
    for rot, trans in records:
        q = np.array(rot, np.float64)
        if not abs(1.0 - np.dot(q, q)) < 1e-14:
            q = q / np.sqrt(np.dot(q, q))
        R = np.dot(_qmat(q), _qbar(q).T)[1:, 1:]
        center = center @ R.T + np.array(trans, np.float64)
        vel = vel @ R.T
        quat = quat @ _qmat(q).T
    return center, quat, vel


def forecast_ids(centers, thresh=0.25):
    """component id per box, components numbered by their smallest member (network_split over the < thresh graph)"""
    centers = np.asarray(centers, np.float64).reshape(-1, 3)
    n = len(centers)
    if n == 0:
        return np.zeros((0,), np.int64)
    A = centers
    D = (A * A).sum(axis=1).reshape((n, 1)) * np.ones((1, n)) + (A * A).sum(axis=1) * np.ones((n, 1)) - 2 * A.dot(A.T)
    D[np.less(D, 0.0)] = 0.0
    adj = np.sqrt(D) < thresh
    label = np.arange(n)
    changed = True
    while changed:
        new = np.array([label[adj[i]].min() for i in range(n)])
        changed = bool((new != label).any())
        label = new
    roots = np.sort(np.unique(label))
    return np.searchsorted(roots, label)
