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
# Only the nms_pre_max best-scoring cells enter NMS: a candidate whose score is within this of the pre_max-th best score
# can be inside the cut in one pipeline and outside in the other (scores agree to ~s(1-s) * 1e-3 end to end).
TOPK_EPS = 1e-4


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


def attribute_detection_diffs(name, got, want, iou_fn, score_thr, iou_thr, n_box=9, max_frac=0.02, iou_eps=IOU_EPS_E2E, topk_cut=None,
                              row_tol=1e-3, score_eps=SCORE_EPS, topk_eps=TOPK_EPS):
    """got / want: [K, n_box + 2] rows (box, score, label).  Every row without a counterpart within 1e-3 must be
    explained by (a) a score within SCORE_EPS of the threshold or within TOPK_EPS of ``topk_cut`` (the nms_pre_max-th best
    candidate score of the oracle, when more than nms_pre_max cells pass the threshold), (b) an IoU within ``iou_eps`` of the
    NMS threshold with a box of the same label -- or an IoU that crosses the threshold when an angle moves by one ulp --, or (c) an IoU above the threshold with another unmatched row that is itself explained (the
    cascade of (a)/(b): its suppressor appeared or vanished; every chain must start at a root of kind (a) or (b)).
    ``row_tol`` / ``score_eps`` / ``topk_eps`` default to the fp32 values; the bf16 comparisons pass the eps that a 2e-2 map error implies.
    Returns the number of unmatched rows."""
    ug, uw = match_rows(got, want, row_tol)
    n_un = len(ug) + len(uw)
    report(name + " detections", float(n_un), max_frac * (len(got) + len(want)), "(unmatched rows of %d + %d)" % (len(got), len(want)))
    if n_un == 0:
        return 0
    allrows = np.concatenate([got, want], 0)
    lab = allrows[:, n_box + 1]
    unmatched = [("got", got[i]) for i in ug] + [("want", want[i]) for i in uw]
    # roots: rows that sit at a decision boundary themselves
    explained = []
    for side, row in unmatched:
        root = abs(float(row[n_box]) - score_thr) <= score_eps or (topk_cut is not None and abs(float(row[n_box]) - topk_cut) <= topk_eps)
        if not root:
            same = allrows[lab == row[n_box + 1]]
            a, b = nms_layout(row[None, :n_box]), nms_layout(same[:, :n_box])
            iou = iou_fn(a, b)[0]
            root = bool(np.any(np.abs(iou - iou_thr) <= iou_eps))
            if not root:
                # a pair whose float32 IoU flips across the threshold when an angle moves by ONE ulp is decided by how
                # sinf / cosf round on the device vs the host (boxes so large that the arithmetic has no digits left)
                for sgn in (np.float32(np.inf), np.float32(-np.inf)):
                    a2, b2 = a.copy(), b.copy()
                    a2[:, 6] = np.nextafter(a2[:, 6], sgn)
                    b2[:, 6] = np.nextafter(b2[:, 6], sgn)
                    for ia, ib in ((a2, b), (a, b2)):
                        root = root or bool(np.any((iou_fn(ia, ib)[0] > iou_thr) != (iou > iou_thr)))
        explained.append(root)
    # cascade: a row whose suppressor (an overlapping row of the same label) is itself an explained difference
    changed = True
    while changed:
        changed = False
        for i, (side, row) in enumerate(unmatched):
            if explained[i]:
                continue
            for j, (_, other) in enumerate(unmatched):
                if j != i and explained[j] and other[n_box + 1] == row[n_box + 1] and \
                        iou_fn(nms_layout(row[None, :n_box]), nms_layout(other[None, :n_box]))[0, 0] > iou_thr - iou_eps:
                    explained[i] = changed = True
                    break
    unexplained = [(side, row[:3].tolist(), float(row[n_box])) for (side, row), ok in zip(unmatched, explained) if not ok]
    assert not unexplained, "%s: detections differ without a near-threshold score / IoU pair: %s" % (name, unexplained[:5])
    assert n_un <= max(2, max_frac * (len(got) + len(want))), "%s: %d unmatched rows" % (name, n_un)
    return n_un
