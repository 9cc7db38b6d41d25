"""bf16 configuration of the oracle (BASELINE configs[2..4]).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference has no bf16 mode of its own (apex/amp is optional there and off in the shipped configs); BASELINE's bf16
configurations mean "conv features and weights in bf16, fp32 accumulate".  This module restates exactly that arithmetic over
the fp32 oracle's modules (oracle/model.py), so that the HIP bf16 path can be checked element-wise instead of against the
fp32 oracle with a blanket tolerance:

  * eval-mode BatchNorm is folded into the convolution (scale = gamma / sqrt(var + eps), shift = beta - mean * scale,
    w' = w * scale, b' = b * scale + shift, evaluated in float64 and rounded to float32 once -- as the product's fold does)
    and w' is rounded to bf16 (round to nearest even); the bias stays fp32;
  * every convolution runs on bf16-valued fp32 tensors (products are exact, the sum is fp32), adds the fp32 bias (and the
    bf16 residual), applies ReLU, and rounds its OUTPUT to bf16 once -- one rounding per fused layer, as the kernels do;
  * the voxel means are computed in fp32 and rounded to bf16 when they enter the first convolution; the head's final maps
    are bf16-rounded; decode + NMS run in fp32 on them (oracle/model.py::CenterHead.predict).

Differences that remain between this and the device are those of fp32 summation ORDER before a rounding (a few 1e-7
relative), which can move a bf16 rounding by one ulp (2^-8): the tests allow 2e-2 * max(1, |ref|) per element.

Follows (with the lines of oracle/model.py): det3d/models/backbones/scn.py:67-78,99-168, det3d/models/necks/rpn.py:124-159,
det3d/models/bbox_heads/center_head.py:129-143,336-390.
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from . import spconv_api as spconv


_TRACE = None  # list that receives one record per fused layer while tracing() is active


class tracing(object):
    """with tracing() as tr: ... -- every fused layer appends dict(kind="sparse"|"dense", x=input, residual=..., y=output (bf16-valued),
    conv=module, ...) in execution order: the tests feed each layer's INPUT to the device layer and compare the OUTPUT (teacher forcing:
    rounding differences cannot accumulate across layers)."""

    def __enter__(self):
        global _TRACE
        _TRACE = []
        return _TRACE

    def __exit__(self, *a):
        global _TRACE
        _TRACE = None


def bf(x):
    """round to bf16 (nearest even) and back to fp32"""
    if isinstance(x, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(x)).to(torch.bfloat16).float().numpy()
    return x.to(torch.bfloat16).float()


def _fold(w, b, bn, out_axis):
    """w' = w * scale, b' = b * scale + shift with scale = gamma / sqrt(var + eps), shift = beta - mean * scale, formed in float64
    and rounded to float32 once (a float32 sqrt / divide is not reproducible to the ulp across devices, and an ulp of a folded
    weight next to a bf16 rounding boundary is a whole bf16 ulp of that weight)."""
    scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    shift = bn.bias.detach().double() - bn.running_mean.detach().double() * scale
    shape = [1] * w.dim()
    shape[out_axis] = -1
    w2 = (w.detach().double() * scale.view(shape)).float()
    b2 = ((b.detach().double() * scale if b is not None else torch.zeros_like(scale)) + shift).float()
    return w2, b2


def _fold_dense(conv, bn):
    transposed = isinstance(conv, nn.ConvTranspose2d)
    w = conv.weight.detach().float()
    b = conv.bias.detach().float() if conv.bias is not None else None
    if bn is not None:
        w, b = _fold(w, b, bn, 1 if transposed else 0)
    return bf(w), b


def dense_stack(mods, x):
    """[ZeroPad2d?, Conv2d | ConvTranspose2d, BatchNorm2d?, ReLU?]* on a bf16-valued NCHW tensor"""
    mods = list(mods)
    i, pad = 0, 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.ZeroPad2d):
            pad = int(m.padding[0])
            i += 1
            continue
        assert isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)), type(m)
        j = i + 1
        bn = None
        if j < len(mods) and isinstance(mods[j], nn.BatchNorm2d):
            bn = mods[j]
            j += 1
        relu = j < len(mods) and isinstance(mods[j], nn.ReLU)
        if relu:
            j += 1
        w, b = _fold_dense(m, bn)
        if isinstance(m, nn.ConvTranspose2d):
            y = F.conv_transpose2d(x, w, None, stride=m.stride, padding=m.padding)
        else:
            y = F.conv2d(x, w, None, stride=m.stride, padding=int(m.padding[0]) + pad)
        pad = 0
        if b is not None:
            y = y + b.view(1, -1, 1, 1)
        if relu:
            y = torch.relu(y)
        y = bf(y)
        if _TRACE is not None:
            _TRACE.append(dict(kind="dense", conv=m, x=x, y=y, relu=relu))
        x = y
        i = j
    return x


def _seq_mods(seq):
    return list(seq._modules.values())


def sparse_conv(conv, bn, x, relu, residual=None):
    """one fused sparse layer: x (SparseConvTensor with bf16-valued features) -> SparseConvTensor"""
    data = x.find_indice_pair(conv.indice_key)
    if data is not None:
        out_idx, pairs, pnum, out_shape = data
    else:
        out_idx, pairs, pnum, out_shape = ops.rulebook(x.indices.numpy(), x.spatial_shape, conv.kernel_size, conv.stride, conv.padding, conv.subm)
        if conv.indice_key is not None:
            x.indice_dict[conv.indice_key] = (out_idx, pairs, pnum, out_shape)
    w = conv.weight.detach().float()
    b = conv.bias.detach().float() if conv.bias is not None else None
    if bn is not None:
        w, b = _fold(w, b, bn, 4)
    y = ops.indice_conv(np.ascontiguousarray(x.features.detach().numpy()), bf(w).numpy(), None if b is None else b.numpy(), pairs, pnum,
                        out_idx.shape[0])
    y = torch.from_numpy(y)
    if residual is not None:
        y = y + residual.features
    if relu:
        y = torch.relu(y)
    out = spconv.SparseConvTensor(bf(y), torch.from_numpy(out_idx), out_shape, x.batch_size)
    out.indice_dict = x.indice_dict
    if _TRACE is not None:
        _TRACE.append(dict(kind="sparse", conv=conv, x=x, residual=residual, y=out, relu=relu))
    return out


def _block(blk, x):  # scn.py:67-78
    y = sparse_conv(blk.conv1, blk.bn1, x, True)
    return sparse_conv(blk.conv2, blk.bn2, y, True, residual=x)


def backbone(bb, voxel_features, coors, batch_size, input_shape, levels=None):
    """oracle.model.SpMiddleResNetFHD.forward in the bf16 configuration -> [N, C*D, H, W] bf16-valued.  ``levels`` (a dict) receives
    the sparse tensors after conv_input ("input"), conv1 .. conv4 and extra_conv ("extra")."""
    sparse_shape = np.array(input_shape[::-1]) + [1, 0, 0]
    x = spconv.SparseConvTensor(bf(voxel_features.float()), coors.int(), sparse_shape, batch_size)
    keep = levels if levels is not None else {}
    x = keep["input"] = sparse_conv(bb.conv_input[0], bb.conv_input[1], x, True)
    for blk in _seq_mods(bb.conv1):
        x = _block(blk, x)
    keep["conv1"] = x
    for name, stage in (("conv2", bb.conv2), ("conv3", bb.conv3), ("conv4", bb.conv4)):
        mods = _seq_mods(stage)
        x = sparse_conv(mods[0], mods[1], x, True)
        for blk in mods[3:]:
            x = _block(blk, x)
        keep[name] = x
    mods = _seq_mods(bb.extra_conv)
    x = keep["extra"] = sparse_conv(mods[0], mods[1], x, True)
    ret = x.dense()
    N, C, D, H, W = ret.shape
    return ret.view(N, C * D, H, W)


def neck(rpn, x):  # rpn.py:124-159
    ups = []
    for i, blk in enumerate(rpn.blocks):
        x = dense_stack(_seq_mods(blk), x)
        if i - rpn._start >= 0:
            ups.append(dense_stack(_seq_mods(rpn.deblocks[i - rpn._start]), x))
    return torch.cat(ups, dim=1) if ups else x


def head(h, x):  # center_head.py:336-390 (no bev_map input in the bf16 configurations)
    assert not h.bev_map
    x = dense_stack(list(h.shared_conv), x)
    rets = []
    for i, task in enumerate(h.tasks):
        xin = torch.cat([x, rets[i - 1]["feats"]], dim=1) if (i != 0 and h.forecast_feature) else x
        ret = {}
        if task.forecast_feature:
            xin = dense_stack(list(task.forecast_conv), xin)
            ret["feats"] = xin
        for name in task.heads:
            ret[name] = dense_stack(_seq_mods(getattr(task, name)), xin)
        rets.append(ret)
    return rets


@torch.no_grad()
def run(onet, example, test_cfg):
    """-> (backbone BEV, neck BEV, head maps, detections of sample 0) of oracle.model.VoxelNet ``onet`` in the bf16 configuration"""
    feats = onet.reader(example["voxels"], example["num_points"])
    bb = backbone(onet.backbone, feats, example["coordinates"], len(example["num_voxels"]), example["shape"][0])
    bev = neck(onet.neck, bb)
    preds = head(onet.bbox_head, bev)
    det = onet.bbox_head.predict(example, [{k: v for k, v in p.items() if k != "feats"} for p in preds], test_cfg)
    return bb, bev, preds, det
