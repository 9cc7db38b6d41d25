"""Helpers shared by the parity tests: element-wise tolerances with the achieved error on record, and attribution of
every NMS / detection difference to a named near-threshold pair or score (nothing is waved through by a blanket
percentage)."""
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(REPO, "gpurun_out", "parity_report.txt")

SCORE_EPS = 1e-5   # a detection may appear / disappear when its score is this close to score_threshold
IOU_EPS = 1e-4     # an NMS decision on IDENTICAL input boxes may flip when the pair's IoU is this close to the threshold
# End to end the two pipelines' boxes themselves agree only to the 1e-3 (relative, per component) of north_star, and an IoU
# moves by about the same relative amount as its boxes: a pair within 3e-3 of the threshold can legitimately flip.
IOU_EPS_E2E = 3e-3


def report(name, err, tol, note=""):
    """Prints and appends '<name> max_err tol' to gpurun_out/parity_report.txt (merged back from the GPU box; the
    round's copy is committed under profiles/)."""
    line = "%-64s err %.3e  tol %.1e  %s" % (name, err, tol, note)
    print("[parity] " + line)
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


def rel_err(got, ref):
    """max over elements of |got - ref| / max(1, |ref|)"""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    if got.size == 0:
        return 0.0
    return float((np.abs(got - ref) / np.maximum(1.0, np.abs(ref))).max())


def assert_close(name, got, ref, tol, note=""):
    """element-wise |got - ref| <= tol * max(1, |ref|); the achieved maximum goes to the parity report"""
    err = rel_err(got, ref)
    report(name, err, tol, note)
    assert err <= tol, "%s: element-wise error %.3e > %.1e" % (name, err, tol)
    return err


def greedy_violations(iou, kept, thr, eps=IOU_EPS, limit=None):
    """Checks that ``kept`` (ascending positions into a score-ordered box list with pairwise IoU matrix ``iou``) is a
    greedy-NMS result up to IoU values within ``eps`` of ``thr``: a kept box must not overlap an earlier kept box by
    more than thr + eps, a dropped box must overlap an earlier kept box by more than thr - eps.  ``limit`` = post_max
    (boxes after the limit-th kept one are not examined).  Returns a list of (position, reason)."""
    n = iou.shape[0]
    kept = [int(k) for k in kept]
    assert kept == sorted(kept), "NMS output must be in score order"
    bad = []
    cur = []
    ks = set(kept)
    for j in range(n):
        if limit is not None and len(cur) >= limit:
            break
        ov = iou[cur, j] if cur else np.zeros((0,))
        if j in ks:
            if ov.size and ov.max() > thr + eps:
                bad.append((j, "kept although IoU %.6f with kept box %d" % (ov.max(), cur[int(ov.argmax())])))
            cur.append(j)
        elif not (ov.size and ov.max() > thr - eps):
            bad.append((j, "dropped although max IoU with kept boxes is %.6f" % (ov.max() if ov.size else 0.0)))
    return bad


def nms_layout(det):
    """[x,y,z,d0,d1,d2,(vx,vy,)yaw] rows -> the layout rotate_nms_pcdet hands to the IoU routine
    (det3d/core/bbox/box_torch_ops.py:256-257): [x,y,z,d1,d0,d2,-yaw-pi/2]"""
    det = np.asarray(det, np.float32)
    out = det[:, [0, 1, 2, 4, 3, 5, det.shape[1] - 1]].copy()
    out[:, 6] = -out[:, 6] - np.float32(np.pi / 2)
    return out


def match_rows(got, want, tol=1e-3):
    """Order-insensitive matching of detection rows: returns (unmatched rows of got, unmatched rows of want)."""
    if len(got) == 0 or len(want) == 0:
        return list(range(len(got))), list(range(len(want)))
    d = (np.abs(got[:, None, :] - want[None, :, :]) / np.maximum(1.0, np.abs(want[None, :, :]))).max(-1)
    return [int(i) for i in np.nonzero(d.min(1) > tol)[0]], [int(i) for i in np.nonzero(d.min(0) > tol)[0]]


def attribute_detection_diffs(name, got, want, iou_fn, score_thr, iou_thr, n_box=9, max_frac=0.02, iou_eps=IOU_EPS_E2E):
    """got / want: [K, n_box + 2] rows (box, score, label).  Every row without a counterpart within 1e-3 must be
    explained by (a) a score within SCORE_EPS of the threshold, (b) an IoU within ``iou_eps`` of the NMS threshold with a
    box of the same label, or (c) an IoU above the threshold with another *unexplained-by-itself* unmatched row
    (the cascade of (a)/(b): its suppressor appeared or vanished).  Returns the number of unmatched rows."""
    ug, uw = match_rows(got, want)
    n_un = len(ug) + len(uw)
    report(name + " detections", float(n_un), max_frac * (len(got) + len(want)), "(unmatched rows of %d + %d)" % (len(got), len(want)))
    if n_un == 0:
        return 0
    allrows = np.concatenate([got, want], 0)
    lab = allrows[:, n_box + 1]
    unmatched = [("got", got[i]) for i in ug] + [("want", want[i]) for i in uw]
    un_boxes = np.stack([r for _, r in unmatched])
    unexplained = []
    for side, row in unmatched:
        if abs(float(row[n_box]) - score_thr) <= SCORE_EPS:
            continue
        same = allrows[lab == row[n_box + 1]]
        iou = iou_fn(nms_layout(row[None, :n_box]), nms_layout(same[:, :n_box]))[0]
        if np.any(np.abs(iou - iou_thr) <= iou_eps):
            continue
        others = un_boxes[(un_boxes[:, n_box + 1] == row[n_box + 1]) & (np.abs(un_boxes - row).max(1) > 0)]
        if len(others) and np.any(iou_fn(nms_layout(row[None, :n_box]), nms_layout(others[:, :n_box]))[0] > iou_thr - iou_eps):
            continue
        unexplained.append((side, row[:3].tolist(), float(row[n_box])))
    assert not unexplained, "%s: detections differ without a near-threshold score / IoU pair: %s" % (name, unexplained[:5])
    assert n_un <= max(2, max_frac * (len(got) + len(want))), "%s: %d unmatched rows" % (name, n_un)
    return n_un
