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
    boxes = boxes[:, [0, 1, 2, 4, 3, 5, -1]]
    boxes[:, -1] = -boxes[:, -1] - np.pi / 2
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    boxes = boxes[order].contiguous()
    keep = torch.zeros(boxes.size(0), dtype=torch.long)
    num_out = 0 if len(boxes) == 0 else nms_gpu(boxes, keep, thresh)
    selected = order[keep[:num_out].to(order.device)].contiguous()
    if post_max_size is not None:
        selected = selected[:post_max_size]
    return selected
