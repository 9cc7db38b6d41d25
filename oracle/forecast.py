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
