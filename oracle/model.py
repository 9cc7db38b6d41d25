"""CPU restatement of VoxelNet (reader -> backbone -> neck -> head -> predict).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Plain, unfused torch-CPU modules whose
attribute names reproduce the reference's state_dict keys, so one seeded state_dict drives the
reference classes (when generating tests/golden), this oracle, and the HIP product path.

Follows:
  reader    det3d/models/readers/voxel_encoder.py:17-24
  backbone  det3d/models/backbones/scn.py:37-176 (over oracle/spconv_api.py)
  neck      det3d/models/necks/rpn.py:22-159
  head      det3d/models/bbox_heads/center_head.py:81-174 (SepHead), :232-390 (CenterHead)
  predict   center_head.py:542-747, det3d/core/bbox/box_torch_ops.py:248-277,
            det3d/ops/iou3d_nms/src/iou3d_nms.cpp:90-135
  detector  det3d/models/detectors/voxelnet.py:23-56
"""
import copy

import numpy as np
import torch
from torch import nn

from . import ops
from . import spconv_api as spconv


def _bn1d(c):
    return nn.BatchNorm1d(c, eps=1e-3, momentum=0.01)  # scn.py:52,95


def _bn2d(c):
    return nn.BatchNorm2d(c, eps=1e-3, momentum=0.01)  # rpn.py:46


class Seq(nn.Module):
    """det3d Sequential with .add(): children named "0","1",... (models/utils/misc.py:22-95)."""

    def __init__(self, *mods):
        super().__init__()
        for i, m in enumerate(mods):
            self.add_module(str(i), m)

    def add(self, m):
        self.add_module(str(len(self._modules)), m)

    def __getitem__(self, i):
        return list(self._modules.values())[i]

    def forward(self, x):
        for m in self._modules.values():
            x = m(x)
        return x


class VFE(nn.Module):
    def __init__(self, num_input_features=4):
        super().__init__()
        self.num_input_features = num_input_features

    def forward(self, features, num_voxels, coors=None):
        assert self.num_input_features == features.shape[-1]
        mean = features[:, :, : self.num_input_features].sum(dim=1) / num_voxels.type_as(features).view(-1, 1)
        return mean.contiguous()


class SparseBasicBlock(spconv.SparseModule):  # scn.py:37-80
    def __init__(self, planes, indice_key):
        super().__init__()
        self.conv1 = spconv.SubMConv3d(planes, planes, 3, stride=1, padding=1, bias=True, indice_key=indice_key)
        self.bn1 = _bn1d(planes)
        self.relu = nn.ReLU()
        self.conv2 = spconv.SubMConv3d(planes, planes, 3, stride=1, padding=1, bias=True, indice_key=indice_key)
        self.bn2 = _bn1d(planes)

    def forward(self, x):
        identity = x
        out = self.conv1(x)
        out.features = self.relu(self.bn1(out.features))
        out = self.conv2(out)
        out.features = self.bn2(out.features)
        out.features = self.relu(out.features + identity.features)
        return out


class SpMiddleResNetFHD(nn.Module):  # scn.py:83-176
    def __init__(self, num_input_features=128):
        super().__init__()
        S = spconv.SparseSequential
        self.conv_input = S(spconv.SubMConv3d(num_input_features, 16, 3, bias=False, indice_key="res0"),
                            _bn1d(16), nn.ReLU())
        self.conv1 = S(SparseBasicBlock(16, "res0"), SparseBasicBlock(16, "res0"))
        self.conv2 = S(spconv.SparseConv3d(16, 32, 3, 2, padding=1, bias=False), _bn1d(32), nn.ReLU(),
                       SparseBasicBlock(32, "res1"), SparseBasicBlock(32, "res1"))
        self.conv3 = S(spconv.SparseConv3d(32, 64, 3, 2, padding=1, bias=False), _bn1d(64), nn.ReLU(),
                       SparseBasicBlock(64, "res2"), SparseBasicBlock(64, "res2"))
        self.conv4 = S(spconv.SparseConv3d(64, 128, 3, 2, padding=[0, 1, 1], bias=False), _bn1d(128), nn.ReLU(),
                       SparseBasicBlock(128, "res3"), SparseBasicBlock(128, "res3"))
        self.extra_conv = S(spconv.SparseConv3d(128, 128, (3, 1, 1), (2, 1, 1), bias=False), _bn1d(128), nn.ReLU())

    def forward(self, voxel_features, coors, batch_size, input_shape):
        sparse_shape = np.array(input_shape[::-1]) + [1, 0, 0]  # scn.py:151
        ret = spconv.SparseConvTensor(voxel_features, coors.int(), sparse_shape, batch_size)
        x = self.conv_input(ret)
        c1 = self.conv1(x)
        c2 = self.conv2(c1)
        c3 = self.conv3(c2)
        c4 = self.conv4(c3)
        ret = self.extra_conv(c4).dense()
        N, C, D, H, W = ret.shape
        return ret.view(N, C * D, H, W), {"conv1": c1, "conv2": c2, "conv3": c3, "conv4": c4}


class RPN(nn.Module):  # rpn.py:22-159
    def __init__(self, layer_nums, ds_layer_strides, ds_num_filters, us_layer_strides, us_num_filters,
                 num_input_features, **kw):
        super().__init__()
        start = len(layer_nums) - len(us_layer_strides)
        in_f = [num_input_features, *ds_num_filters[:-1]]
        blocks, deblocks = [], []
        for i, nl in enumerate(layer_nums):
            blk = Seq(nn.ZeroPad2d(1), nn.Conv2d(in_f[i], ds_num_filters[i], 3, stride=ds_layer_strides[i], bias=False),
                      _bn2d(ds_num_filters[i]), nn.ReLU())
            for _ in range(nl):
                blk.add(nn.Conv2d(ds_num_filters[i], ds_num_filters[i], 3, padding=1, bias=False))
                blk.add(_bn2d(ds_num_filters[i]))
                blk.add(nn.ReLU())
            blocks.append(blk)
            if i - start >= 0:
                s = us_layer_strides[i - start]
                cout = us_num_filters[i - start]
                if s > 1:
                    conv = nn.ConvTranspose2d(ds_num_filters[i], cout, s, stride=s, bias=False)
                else:
                    s = int(np.round(1 / s))
                    conv = nn.Conv2d(ds_num_filters[i], cout, s, stride=s, bias=False)
                deblocks.append(Seq(conv, _bn2d(cout), nn.ReLU()))
        self.blocks = nn.ModuleList(blocks)
        self.deblocks = nn.ModuleList(deblocks)
        self._start = start

    def forward(self, x):
        ups = []
        for i, blk in enumerate(self.blocks):
            x = blk(x)
            if i - self._start >= 0:
                ups.append(self.deblocks[i - self._start](x))
        return torch.cat(ups, dim=1) if ups else x


class SepHead(nn.Module):  # center_head.py:81-174 (bn=True, final_kernel=3 as built at :361-368)
    def __init__(self, in_channels, heads, head_conv=64, forecast_feature=False, wide_head=False):
        super().__init__()
        self.heads = heads
        self.forecast_feature = forecast_feature
        if forecast_feature:
            self.forecast_conv = nn.Sequential(
                nn.Conv2d(in_channels, head_conv, 3, padding=1), nn.BatchNorm2d(head_conv), nn.ReLU(),
                nn.Conv2d(head_conv, head_conv, 3, padding=1), nn.BatchNorm2d(head_conv), nn.ReLU())
        if wide_head:  # :127-128
            head_conv = in_channels
        for head, (classes, num_conv) in heads.items():
            fc = Seq()
            for _ in range(num_conv - 1):
                fc.add(nn.Conv2d(head_conv, head_conv, 3, padding=1))
                fc.add(nn.BatchNorm2d(head_conv))
                fc.add(nn.ReLU())
            fc.add(nn.Conv2d(head_conv, classes, 3, padding=1))
            setattr(self, head, fc)

    def forward(self, x):
        ret = {}
        if self.forecast_feature:
            x = self.forecast_conv(x)
            ret["feats"] = x
        for head in self.heads:
            ret[head] = getattr(self, head)(x)
        return ret


class CenterHead(nn.Module):  # center_head.py:232-390 (branches reachable from the shipped configs)
    def __init__(self, in_channels, tasks, common_heads, share_conv_channel=64, num_hm_conv=2, timesteps=1,
                 dense=False, bev_map=False, forecast_feature=False, classify=False, reverse=False, sparse=False, wide_head=False, **kw):
        super().__init__()
        for flag in ("two_stage", "dcn_head"):
            assert not kw.get(flag, False), flag
        self.dense, self.bev_map, self.forecast_feature, self.classify = dense, bev_map, forecast_feature, classify
        self.reverse, self.sparse, self.wide_head = reverse, sparse, wide_head
        self.standard = not (reverse or sparse or dense or classify or wide_head)  # :268-271
        self.timesteps = timesteps
        self.target_timesteps = 7
        self.num_classes = [len(t["class_names"]) for t in tasks]
        if sparse:  # :322-324
            self.num_classes = 2 * [1]
        if dense:
            self.num_classes = timesteps * [1]
        if classify:  # :329-330
            self.num_classes = timesteps * [3]
        if wide_head:  # :332-334
            self.num_classes = [7]
            share_conv_channel = 512
        if bev_map:
            c = share_conv_channel
            self.bev_conv = nn.Sequential(
                nn.Conv2d(6, 16, 3, padding=1), nn.BatchNorm2d(16), nn.ReLU(),
                nn.Conv2d(16, 32, 3, padding=1), nn.BatchNorm2d(32), nn.ReLU(),
                nn.Conv2d(32, c, 3, padding=1), nn.BatchNorm2d(c), nn.ReLU())
        self.shared_conv = nn.Sequential(nn.Conv2d(in_channels, share_conv_channel, 3, padding=1),
                                         nn.BatchNorm2d(share_conv_channel), nn.ReLU())
        self.tasks = nn.ModuleList()
        for i, num_cls in enumerate(self.num_classes):
            heads = copy.deepcopy(dict(common_heads))
            if not (dense or classify or wide_head) and "vel" in heads:  # :355 (standard, reverse, sparse)
                heads["vel"] = (timesteps * heads["vel"][0], heads["vel"][1])
            heads.update(dict(hm=(num_cls, num_hm_conv)))
            cin = 2 * share_conv_channel if (i != 0 and forecast_feature) else share_conv_channel
            self.tasks.append(SepHead(cin, heads, forecast_feature=forecast_feature, wide_head=wide_head))

    def forward(self, x, bev_map=None):
        rets = []
        x = self.shared_conv(x)
        if self.bev_map:
            x = x + self.bev_conv(bev_map)
        for i, task in enumerate(self.tasks):
            if i != 0 and self.forecast_feature:
                rets.append(task(torch.cat([x, rets[i - 1]["feats"]], dim=1)))
            else:
                rets.append(task(x))
        return rets

    @torch.no_grad()
    def predict(self, example, preds_dicts, test_cfg):
        post_range = torch.tensor(test_cfg["post_center_limit_range"], dtype=preds_dicts[0]["hm"].dtype)
        steps = []
        if self.sparse:  # :572-587: forward task's steps, then the reverse task's
            num_classes = [1, 1] * self.target_timesteps
            for src in (preds_dicts[0], preds_dicts[1]):
                for i in range(self.timesteps):
                    d = dict(src)
                    d["vel"] = src["vel"][:, 2 * i:2 * i + 2]
                    steps.append(d)
        elif self.standard or self.reverse:  # :559-570
            pd = preds_dicts[0]
            vels = [pd["vel"][:, 2 * i:2 * i + 2] for i in range(self.timesteps)]
            if len(vels) == 1:
                vels = self.target_timesteps * vels
            num_classes = [1] * self.target_timesteps
            for v in vels:
                d = dict(pd)
                d["vel"] = v
                steps.append(d)
        elif self.wide_head:  # :597-604: step s = heat-map channel s of the one task
            pd = preds_dicts[0]
            steps = [dict(pd, hm=pd["hm"][:, i].unsqueeze(1)) for i in range(self.timesteps)]
            num_classes = self.timesteps * [1]
        elif self.classify:  # :589-595: the class channels collapse to their maximum, one class per step afterwards
            steps = [dict(d, hm=torch.max(d["hm"], dim=1)[0].unsqueeze(1)) for d in preds_dicts]
            num_classes = self.timesteps * [1]
        else:  # :606-607
            steps = [dict(d) for d in preds_dicts]
            num_classes = self.num_classes
        rets = []
        for pd in steps:
            pd = {k: v.permute(0, 2, 3, 1).contiguous() for k, v in pd.items()}
            hm = torch.sigmoid(pd["hm"])
            dim = torch.exp(pd["dim"])
            rot = torch.atan2(pd["rot"][..., 0:1], pd["rot"][..., 1:2])
            B, H, W, ncls = hm.shape
            reg = pd["reg"].reshape(B, H * W, 2)
            hei = pd["height"].reshape(B, H * W, 1)
            rot = rot.reshape(B, H * W, 1)
            dim = dim.reshape(B, H * W, 3)
            hm = hm.reshape(B, H * W, ncls)
            ys, xs = torch.meshgrid([torch.arange(0, H), torch.arange(0, W)], indexing="ij")
            ys = ys.view(1, H, W).repeat(B, 1, 1).to(hm)
            xs = xs.view(1, H, W).repeat(B, 1, 1).to(hm)
            xs = xs.view(B, -1, 1) + reg[:, :, 0:1]
            ys = ys.view(B, -1, 1) + reg[:, :, 1:2]
            xs = xs * test_cfg["out_size_factor"] * test_cfg["voxel_size"][0] + test_cfg["pc_range"][0]
            ys = ys * test_cfg["out_size_factor"] * test_cfg["voxel_size"][1] + test_cfg["pc_range"][1]
            vel = pd["vel"].reshape(B, H * W, 2)
            boxes = torch.cat([xs, ys, hei, dim, vel, rot], dim=2)
            rets.append(self.post_processing(boxes, hm, test_cfg, post_range, len(rets)))
        out = []
        for i in range(len(rets[0])):
            flag = 0
            for j, nc in enumerate(num_classes):
                rets[j][i]["label_preds"] = rets[j][i]["label_preds"] + flag
                flag += nc
            ret = {k: torch.cat([r[i][k] for r in rets]) for k in ("box3d_lidar", "scores", "label_preds")}
            meta = example.get("metadata") if isinstance(example, dict) else None
            ret["metadata"] = meta[i] if meta else None
            out.append(ret)
        return out

    @staticmethod
    def post_processing(batch_box_preds, batch_hm, test_cfg, post_range, task_id=0):  # :699-747
        nms_cfg = test_cfg["nms"]
        res = []
        for i in range(len(batch_hm)):
            box_preds = batch_box_preds[i]
            scores, labels = torch.max(batch_hm[i], dim=-1)
            mask = (scores > test_cfg["score_threshold"]) \
                & (box_preds[..., :3] >= post_range[:3]).all(1) & (box_preds[..., :3] <= post_range[3:]).all(1)
            box_preds, scores, labels = box_preds[mask], scores[mask], labels[mask]
            if test_cfg.get("circular_nms", False):  # :722-725, _circle_nms :750-758: no pre-NMS cut in this mode
                dets = torch.cat([box_preds[:, [0, 1]], scores.view(-1, 1)], dim=1).numpy()
                sel = torch.from_numpy(np.asarray(circle_nms(dets, test_cfg["min_radius"][task_id])[:nms_cfg["nms_post_max_size"]], np.int64))
            else:
                sel = rotate_nms_pcdet(box_preds[:, [0, 1, 2, 3, 4, 5, -1]].float(), scores.float(),
                                       nms_cfg["nms_iou_threshold"], nms_cfg["nms_pre_max_size"],
                                       nms_cfg["nms_post_max_size"])
            res.append({"box3d_lidar": box_preds[sel], "scores": scores[sel], "label_preds": labels[sel]})
        return res


def circle_nms(dets, thresh):  # core/utils/circle_nms_jit.py:5-27, dets [n,3] float32 = x, y, score
    """Greedy by descending score; a later box is suppressed when its SQUARED centre distance to a kept one is <= thresh (float32
    products and sum, as numpy evaluates the reference's expression on float32 scalars when numba is an identity decorator -- which is how
    tests/golden/make_golden.py runs it).  Vectorised per kept box; the visiting order and every comparison are the loop's."""
    x, y = dets[:, 0], dets[:, 1]
    order = dets[:, 2].argsort()[::-1]
    alive = np.ones(len(dets), bool)
    thr = np.float32(thresh)
    keep = []
    for pos, i in enumerate(order):
        if not alive[i]:
            continue
        keep.append(int(i))
        rest = order[pos + 1:]
        dx, dy = x[i] - x[rest], y[i] - y[rest]
        alive[rest[dx * dx + dy * dy <= thr]] = False
    return keep


def rotate_nms_pcdet(boxes, scores, thresh, pre_maxsize=None, post_max_size=None):  # box_torch_ops.py:248-277
    boxes = boxes[:, [0, 1, 2, 4, 3, 5, -1]]
    boxes[:, -1] = -boxes[:, -1] - np.pi / 2
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    boxes = boxes[order].contiguous()
    if len(boxes) == 0:
        keep = torch.zeros(0, dtype=torch.long)
    else:
        keep = torch.from_numpy(ops.nms(boxes.numpy(), float(thresh)))
    sel = order[keep].contiguous()
    if post_max_size is not None:
        sel = sel[:post_max_size]
    return sel


class VoxelNet(nn.Module):  # voxelnet.py:23-56, single_stage.py:23-27
    def __init__(self, reader, backbone, neck, bbox_head, test_cfg=None, **kw):
        super().__init__()
        self.reader = VFE(reader["num_input_features"])
        self.backbone = SpMiddleResNetFHD(backbone["num_input_features"])
        nk = {k: v for k, v in neck.items() if k not in ("type", "logger")}
        self.neck = RPN(**nk)
        hk = {k: v for k, v in bbox_head.items() if k not in ("type", "logger")}
        self.bbox_head = CenterHead(**hk)
        self.test_cfg = test_cfg

    def extract_feat(self, example):
        feats = self.reader(example["voxels"], example["num_points"])
        x, _ = self.backbone(feats, example["coordinates"], len(example["num_voxels"]), example["shape"][0])
        return self.neck(x)

    @torch.no_grad()
    def forward(self, example, return_loss=False):
        x = self.extract_feat(example)
        bev = None
        if self.bbox_head.bev_map:
            bev = torch.stack(example["bev_map"], dim=1).float()
        preds = self.bbox_head(x, bev)
        return self.bbox_head.predict(example, preds, self.test_cfg)


# ------------------------------------------------------------------------------------------------ PointPillars
class PFNLayer(nn.Module):  # pillar_encoder.py:15-55
    def __init__(self, cin, cout, last_layer):
        super().__init__()
        self.last_vfe = last_layer
        self.units = cout if last_layer else cout // 2
        self.linear = nn.Linear(cin, self.units, bias=False)
        self.norm = _bn1d(self.units)

    def forward(self, x):
        x = self.linear(x)
        x = torch.relu(self.norm(x.permute(0, 2, 1)).permute(0, 2, 1))
        x_max = x.max(dim=1, keepdim=True)[0]
        if self.last_vfe:
            return x_max
        return torch.cat([x, x_max.expand(-1, x.shape[1], -1)], dim=2)


class PillarFeatureNet(nn.Module):  # pillar_encoder.py:58-164
    def __init__(self, num_input_features=4, num_filters=(64,), with_distance=False, voxel_size=(0.2, 0.2, 4),
                 pc_range=(0, -40, -3, 70.4, 40, 1), **kw):
        super().__init__()
        fin = num_input_features + 5 + (1 if with_distance else 0)
        nf = [fin] + list(num_filters)
        self.pfn_layers = nn.ModuleList([PFNLayer(nf[i], nf[i + 1], i >= len(nf) - 2) for i in range(len(nf) - 1)])
        self.with_distance = with_distance
        self.vx, self.vy = voxel_size[0], voxel_size[1]
        self.x_offset = self.vx / 2 + pc_range[0]
        self.y_offset = self.vy / 2 + pc_range[1]

    def forward(self, features, num_voxels, coors):
        mean = features[:, :, :3].sum(dim=1, keepdim=True) / num_voxels.type_as(features).view(-1, 1, 1)   # :120-122
        f_cluster = features[:, :, :3] - mean
        cx = coors[:, 3].to(features.dtype).unsqueeze(1) * self.vx + self.x_offset                          # :128-133
        cy = coors[:, 2].to(features.dtype).unsqueeze(1) * self.vy + self.y_offset
        f_center = torch.stack([features[:, :, 0] - cx, features[:, :, 1] - cy], dim=-1)
        parts = [features, f_cluster, f_center]
        if self.with_distance:
            parts.append(torch.norm(features[:, :, :3], 2, 2, keepdim=True))
        x = torch.cat(parts, dim=-1)
        mask = torch.arange(x.shape[1]).view(1, -1) < num_voxels.view(-1, 1)                                # :146-149
        x = x * mask.unsqueeze(-1).type_as(x)
        for pfn in self.pfn_layers:
            x = pfn(x)
        return x.squeeze()


def pillars_scatter(voxel_features, coords, batch_size, input_shape):  # pillar_encoder.py:186-221
    nx, ny = int(input_shape[0]), int(input_shape[1])
    C = voxel_features.shape[1]
    out = torch.zeros((batch_size, C, ny * nx), dtype=voxel_features.dtype)
    for b in range(batch_size):
        m = coords[:, 0] == b
        idx = (coords[m, 2] * nx + coords[m, 3]).long()
        out[b][:, idx] = voxel_features[m].t()
    return out.view(batch_size, C, ny, nx)


class PointPillars(nn.Module):  # point_pillars.py:5-50
    def __init__(self, reader, backbone, neck, bbox_head, test_cfg=None, **kw):
        super().__init__()
        self.reader = PillarFeatureNet(**{k: v for k, v in reader.items() if k != "type"})
        self.neck = RPN(**{k: v for k, v in neck.items() if k not in ("type", "logger")})
        self.bbox_head = CenterHead(**{k: v for k, v in bbox_head.items() if k not in ("type", "logger")})
        self.test_cfg = test_cfg

    @torch.no_grad()
    def forward(self, example, return_loss=False):
        feats = self.reader(example["voxels"], example["num_points"], example["coordinates"])
        x = pillars_scatter(feats, example["coordinates"], len(example["num_voxels"]), example["shape"][0])
        preds = self.bbox_head(self.neck(x))
        return self.bbox_head.predict(example, preds, self.test_cfg)
