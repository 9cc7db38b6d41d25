"""Rotated NMS behind the reference's two entry points, on the HIP kernels:

rotate_nms_pcdet(boxes, scores, thresh, pre_maxsize, post_max_size)  det3d/core/bbox/box_torch_ops.py:248-277
iou3d_nms_cuda.{nms_gpu, boxes_iou_bev_gpu}                          det3d/ops/iou3d_nms/src/iou3d_nms_api.cpp:11-17
"""
import numpy as np
import torch

from . import hip_ops


def nms_gpu(boxes, keep, thresh):
    """boxes [N,7] float32 device, pcdet layout, score-sorted; keep: int64 tensor [N] (host, as in the reference,
    or device).  Returns the number kept and fills keep[:num]."""
    k, count = hip_ops.rotated_nms(boxes.contiguous(), float(thresh))
    num = int(count.cpu()[0])
    keep[:num] = k[:num].to(keep.device)
    return num


def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
    ans_iou.copy_(hip_ops.boxes_iou_bev(boxes_a.contiguous(), boxes_b.contiguous()))
    return 1


def rotate_nms_pcdet(boxes, scores, thresh, pre_maxsize=None, post_max_size=None):
    """Entry point of det3d/core/bbox/box_torch_ops.py:248-277 on the device NMS: ``boxes`` [N, >= 7] in the head's layout
    (x, y, z, w, l, h, ..., yaw), ``scores`` [N]; returns the indices (into ``boxes``) of the survivors, best first, at most
    ``post_max_size`` of them.  The candidates, their order and the keep list stay on the device; the one host read is the survivor count
    that sizes the returned tensor.  (CenterHead.predict does not come through here: it runs fd_centerpoint_decode_packed.)"""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros((0,), dtype=torch.long, device=boxes.device)
    n_cand = n if pre_maxsize is None else min(n, int(pre_maxsize))
    # best n_cand candidates, descending (the reference's full sort + cut; ties have no defined order there either)
    rank = torch.topk(scores, n_cand, dim=0, largest=True, sorted=True).indices
    cand = boxes.index_select(0, rank)
    # the kernel's box layout (pcdet): (x, y, z, l, w, h, heading) with heading = -yaw - pi/2
    footprint = torch.stack([cand[:, 0], cand[:, 1], cand[:, 2], cand[:, 4], cand[:, 3], cand[:, 5], -cand[:, -1] - np.pi / 2], dim=1)
    kept, count = hip_ops.rotated_nms(footprint.float().contiguous(), float(thresh))
    limit = count[0].long() if post_max_size is None else count[0].long().clamp(max=int(post_max_size))
    survivors = rank.index_select(0, kept[:n_cand])        # fixed shape; entries past `count` are padding
    return survivors[: int(limit)].contiguous()
