"""CenterHead (det3d/models/bbox_heads/center_head.py:81-174 SepHead, :232-390 CenterHead, :542-747 predict).

Constructor signature and state_dict keys follow the reference (shared_conv.{0,1}.*, tasks.{i}.{reg,height,dim,
rot,vel,hm}.{0,1,3}.*, tasks.{i}.forecast_conv.*, bev_conv.*).  Branches: "standard" (n0 / n3: one task, velocity split
per timestep), "dense" (n3dtf / n3dtfm: one task per timestep, optional chained forecast features and BEV-map branch) and
"classify" (the reference constructor's DEFAULT, center_head.py:253,329-330,589-595: one task per timestep with a three-class
heat-map whose channel maximum is the score map); ``reverse`` (center_head.py:559: decoded exactly like the standard head -- the mode differs in
the training targets only) and ``sparse`` (:322-324,572-587: a forward and a reverse task, each with a velocity pair per timestep; the forward
task's steps first, then the reverse task's) and ``wide_head`` (:332-334,597-604: ONE task on a 512-channel shared convolution whose branches keep that
width and whose heat-map has a channel per timestep; step s decodes channel s with the shared regression maps); ``dcn_head`` / ``two_stage`` are
False in every shipped config and raise.  In eval mode on the device the head runs on the convolution plan of dense_bf16.py
(the only device path; a head it cannot take raises); predict() runs the HIP decode + rotated NMS (fd_centerpoint_decode) for
all (sample, heat-map) groups in one call; the loss is training-only and out of scope of this path.
"""
import copy
import logging

import torch
from torch import nn

from . import hip_ops
from .nn_utils import Sequential, kaiming_init, weights_version
from .registry import HEADS


class SepHead(nn.Module):
    def __init__(self, in_channels, heads, head_conv=64, final_kernel=1, bn=False, init_bias=-2.19, two_stage=False,
                 forecast_feature=False, wide_head=False, **kwargs):
        super().__init__(**kwargs)
        assert not two_stage, "two_stage is False in every shipped config"
        self.heads = heads
        self.forecast_feature = forecast_feature
        if self.forecast_feature:
            self.forecast_conv = nn.Sequential(
                nn.Conv2d(in_channels, head_conv, kernel_size=3, padding=1, bias=True), nn.BatchNorm2d(head_conv),
                nn.ReLU(inplace=True),
                nn.Conv2d(head_conv, head_conv, kernel_size=3, padding=1, bias=True), nn.BatchNorm2d(head_conv),
                nn.ReLU(inplace=True))
        if wide_head:  # center_head.py:127-128: the branches keep the width of the shared convolution
            head_conv = in_channels
        for head in self.heads:
            classes, num_conv = self.heads[head]
            fc = Sequential()
            for _ in range(num_conv - 1):
                fc.add(nn.Conv2d(head_conv, head_conv, kernel_size=final_kernel, stride=1, padding=final_kernel // 2,
                                 bias=True))
                if bn:
                    fc.add(nn.BatchNorm2d(head_conv))
                fc.add(nn.ReLU())
            fc.add(nn.Conv2d(head_conv, classes, kernel_size=final_kernel, stride=1, padding=final_kernel // 2, bias=True))
            if "hm" in head:
                fc[-1].bias.data.fill_(init_bias)
            else:
                for m in fc.modules():
                    if isinstance(m, nn.Conv2d):
                        kaiming_init(m)
            self.__setattr__(head, fc)

    def forward_modules(self, x):
        ret = {}
        if self.forecast_feature:
            x = self.forecast_conv(x)
            ret["feats"] = x
        for head in self.heads:
            ret[head] = self.__getattr__(head)(x)
        return ret

    def forward(self, x):
        return self.forward_modules(x)


def _drop_caches(module, incompatible_keys=None):
    module._plan = None
    module.__dict__.pop("_wv_tensors", None)


@HEADS.register_module
class CenterHead(nn.Module):
    def __init__(self, in_channels=[128, ], tasks=[], dataset="nuscenes", weight=0.25, code_weights=[], common_heads=dict(),
                 logger=None, init_bias=-2.19, share_conv_channel=64, num_hm_conv=2, dcn_head=False, timesteps=1,
                 two_stage=False, reverse=False, sparse=False, dense=False, bev_map=False, forecast_feature=False,
                 classify=True, wide_head=False):
        super().__init__()
        unsupported = dict(dcn_head=dcn_head, two_stage=two_stage)
        on = [k for k, v in unsupported.items() if v]
        if on:
            raise NotImplementedError("CenterHead options %s are False in every shipped centerpoint config and are not "
                                      "part of the inference hot path" % on)
        self.two_stage, self.reverse, self.sparse, self.dense = two_stage, reverse, sparse, dense
        self.bev_map, self.forecast_feature, self.classify, self.wide_head = bev_map, forecast_feature, classify, wide_head
        self.target_timesteps = 7
        self.standard = not (reverse or sparse or dense or classify or wide_head)  # center_head.py:268-271
        num_classes = [len(t["class_names"]) for t in tasks]
        self.class_names = [t["class_names"] for t in tasks]
        self.code_weights = code_weights
        self.box_n_dim = 9 if ("vel" in common_heads and "rot" in common_heads) else 7
        self.weight = weight
        self.dataset = dataset
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.use_direction_classifier = False
        self.timesteps = timesteps
        self.logger = logger or logging.getLogger("CenterHead")
        self.logger.info(f"num_classes: {num_classes}")
        self.tasks = nn.ModuleList()
        if self.sparse:    # center_head.py:322-324: a forward and a reverse task
            self.num_classes = 2 * [1]
        if self.dense:
            self.num_classes = self.timesteps * [1]
        if self.classify:  # center_head.py:329-330 (after the dense rule, as there)
            self.num_classes = self.timesteps * [3]
        if self.wide_head:  # center_head.py:332-334
            self.num_classes = [7]
            share_conv_channel = 512
        if self.bev_map:
            c = share_conv_channel
            self.bev_conv = nn.Sequential(
                nn.Conv2d(6, 16, kernel_size=3, padding=1, bias=True), nn.BatchNorm2d(16), nn.ReLU(inplace=True),
                nn.Conv2d(16, 32, kernel_size=3, padding=1, bias=True), nn.BatchNorm2d(32), nn.ReLU(inplace=True),
                nn.Conv2d(32, c, kernel_size=3, padding=1, bias=True), nn.BatchNorm2d(c), nn.ReLU(inplace=True))
        self.shared_conv = nn.Sequential(nn.Conv2d(in_channels, share_conv_channel, kernel_size=3, padding=1, bias=True),
                                         nn.BatchNorm2d(share_conv_channel), nn.ReLU(inplace=True))
        for i, num_cls in enumerate(self.num_classes):
            heads = copy.deepcopy(dict(common_heads))
            for head in heads.keys():
                if not (self.dense or self.classify or self.wide_head) and head in ["vel", "rvel"]:  # center_head.py:355 (standard, reverse, sparse)
                    heads[head] = (self.timesteps * heads[head][0], heads[head][1])
            heads.update(dict(hm=(num_cls, num_hm_conv)))
            cin = 2 * share_conv_channel if (i != 0 and self.forecast_feature) else share_conv_channel
            self.tasks.append(SepHead(cin, heads, bn=True, init_bias=init_bias, final_kernel=3, two_stage=self.two_stage,
                                      forecast_feature=self.forecast_feature, wide_head=self.wide_head))
        self.compute_dtype = torch.float32
        self._plan = None
        self.register_load_state_dict_post_hook(_drop_caches)
        self.logger.info("Finish CenterHead Initialization")

    invalidate_caches = _drop_caches

    def _apply(self, fn, *a, **kw):
        _drop_caches(self)
        return super()._apply(fn, *a, **kw)

    # ----------------------------------------------------------------------------------------------- forward
    def forward_modules(self, x, bev_map=None):
        ret_dicts = []
        x = self.shared_conv(x)
        if self.bev_map:
            x = x + self.bev_conv(bev_map)
        for i, task in enumerate(self.tasks):
            if i != 0 and self.forecast_feature:
                ret_dicts.append(task(torch.cat([x, ret_dicts[i - 1]["feats"]], dim=1)))
            else:
                ret_dicts.append(task(x))
        return ret_dicts

    def forward(self, x, bev_map=None, *kwargs):
        if self.training or not x.is_cuda:
            return self.forward_modules(x, bev_map)
        if self.compute_dtype not in (torch.bfloat16, torch.float32):
            raise ValueError("CenterHead: compute_dtype must be float32 or bfloat16, got %s" % (self.compute_dtype,))
        ver = (weights_version(self), self.compute_dtype)
        if self._plan is None or self._plan[0] != ver:
            from .dense_bf16 import HeadPlan

            self._plan = (ver, HeadPlan(self, self.compute_dtype))  # raises ValueError for a head the kernels do not take
        return self._plan[1](x.to(self.compute_dtype).permute(0, 2, 3, 1).contiguous(), bev_map)

    def loss(self, example, preds_dicts, **kwargs):
        raise NotImplementedError("training losses (center_head.py:396-539) are outside the inference hot path")

    # ----------------------------------------------------------------------------------------------- predict
    def _groups(self, preds_dicts):
        """-> (list of per-group source dicts, vel tensor per output step, step->group map, num_classes per step)."""
        if self.standard or self.reverse:  # center_head.py:559-570
            pd = preds_dicts[0]
            vels = [pd["vel"][:, 2 * i:2 * i + 2] for i in range(self.timesteps)]
            if len(vels) == 1:
                vels = self.target_timesteps * vels
            return [pd], vels, [0] * len(vels), [1] * len(vels)  # (the reference writes [1] * target_timesteps and can only run 1 or 7 steps)
        if self.sparse:  # center_head.py:572-587: the forward task's steps, then the reverse task's
            fwd, rev = preds_dicts[0], preds_dicts[1]
            vels = [fwd["vel"][:, 2 * i:2 * i + 2] for i in range(self.timesteps)] + [rev["vel"][:, 2 * i:2 * i + 2] for i in range(self.timesteps)]
            return [fwd, rev], vels, [0] * self.timesteps + [1] * self.timesteps, [1] * (2 * self.timesteps)
        if self.wide_head:  # center_head.py:597-604: step s = heat-map channel s of the one task, every other map shared
            pd = preds_dicts[0]
            srcs = [dict(pd, hm=pd["hm"][:, i:i + 1]) for i in range(self.timesteps)]
            return srcs, [pd["vel"]] * self.timesteps, list(range(self.timesteps)), [1] * self.timesteps
        vels = [pd["vel"] for pd in preds_dicts]  # center_head.py:606-607 (dense), :589-595 (classify: one class per step after the channel max)
        return list(preds_dicts), vels, list(range(len(preds_dicts))), [1] * len(preds_dicts) if self.classify else list(self.num_classes)

    @staticmethod
    def _circular(test_cfg):
        """test_cfg.circular_nms (center_head.py:722-725).  ``per_class_nms`` has nothing to reproduce: the reference's branch is ``pass``
        (:668-669), no task's result is collected and predict fails on ``rets[0]`` two lines later."""
        if test_cfg.get("per_class_nms", False):
            raise NotImplementedError("test_cfg.per_class_nms: the reference's branch is empty (center_head.py:668-669) and its predict fails on rets[0]")
        return bool(test_cfg.get("circular_nms", False))

    @staticmethod
    def _group_radius(test_cfg, step_group, G):
        """``test_cfg.min_radius[task_id]`` (center_head.py:724; task_id = output step, :609) per decode group; None when two steps of one
        group differ.  A scalar ``min_radius`` (as every shipped config writes it) fails here as it does in the reference."""
        radii = test_cfg["min_radius"]
        per_step = [float(radii[s]) for s in range(len(step_group))]  # TypeError for a scalar, IndexError for a short list: the reference's own
        out = [None] * G
        for s, g in enumerate(step_group):
            if out[g] is None:
                out[g] = per_step[s]
            elif out[g] != per_step[s]:
                return None
        return [0.0 if r is None else r for r in out]

    @torch.no_grad()
    def predict_packed(self, preds_dicts, test_cfg):
        """Device-only decode straight from the convolution plan's NHWC output buffer: (packed [B,S,post,11] float32 rows
        x y z w l h vx vy yaw score label, counts [B,S] int32) -- five launches (keys + histogram, selection, rank + box decode, IoU mask,
        sweep + gather + assembly: fd_centerpoint_decode_packed), no torch kernel in between.  None when the maps did not come from the plan (torch path, bev_map head)."""
        if self.wide_head:  # (its T groups differ in the heat-map channel only: decoded through the per-group path below)
            return None
        raws = [getattr(pd, "raw", None) for pd in preds_dicts]
        if any(r is None for r in raws) or any(r[0] is not raws[0][0] or r[2] != raws[0][2] for r in raws):
            return None
        circular = self._circular(test_cfg)
        zbuf, _, where = raws[0]
        T, B, H, W, C = zbuf.shape
        hm_channels = where["hm"][1]
        assert hm_channels == 1 or self.classify, "multi-class heat-maps are decoded as their channel maximum only in the classify mode (center_head.py:589-595)"
        if self.sparse:  # center_head.py:572-587: two tasks (forward, reverse), a velocity pair per timestep each; 2 T output steps
            G = 2
            S = 2 * self.timesteps
            assert 2 * self.timesteps <= where["vel"][1], "velocity channels 2 s, 2 s + 1 must exist for every step"
            step_group = [0] * self.timesteps + [1] * self.timesteps
            step_vel = [2 * s for s in range(self.timesteps)] * 2
            num_classes = [1] * S
        elif self.standard or self.reverse:  # center_head.py:559-570: one task, step s = its boxes + velocity channels 2s, 2s+1 (all steps share them when timesteps == 1)
            G = 1
            # center_head.py:559-565 emits one step per velocity pair: ``timesteps`` of them when the head forecasts, else
            # ``target_timesteps`` copies of the single pair (the same rule as _groups above)
            S = self.timesteps if self.timesteps > 1 else self.target_timesteps
            assert self.timesteps <= 1 or 2 * S <= where["vel"][1], "velocity channels 2 s, 2 s + 1 must exist for every step"
            step_group = [0] * S
            step_vel = [2 * s if self.timesteps > 1 else 0 for s in range(S)]
            num_classes = [1] * S
        else:              # :606-607: one task per step
            G = T
            S = T
            step_group = list(range(T))
            step_vel = [0] * T
            num_classes = [1] * T if self.classify else list(self.num_classes)
        labels, acc = [], 0
        for ncls in num_classes:
            labels.append(acc)
            acc += ncls
        cache = self.__dict__.setdefault("_lab_cache", {})
        ck = (zbuf.device, B, S, int(test_cfg["nms"]["nms_post_max_size"]))
        if ck not in cache:  # labels do not depend on the data (built in the eager set-up pass, before any graph capture)
            cache[ck] = torch.as_tensor(labels, dtype=torch.int64, device=zbuf.device).view(1, S, 1).expand(B, S, ck[3]).contiguous()
        group_radius = None
        if circular:
            group_radius = self._group_radius(test_cfg, step_group, G)
            if group_radius is None:  # steps that share a decode group ask for different radii: one group per step (predict_padded)
                return None
        flat = zbuf[:G].reshape(G * B, H, W, C)
        cfg = hip_ops.make_decode_cfg(H, W, test_cfg, hm_channels=hm_channels, group_radius=group_radius)
        views = [hip_ops.nhwc_channel_view(flat, where[k][0]) for k in ("hm", "reg", "height", "dim", "rot")]
        return hip_ops.centerpoint_decode_packed(views, hip_ops.nhwc_channel_view(flat, where["vel"][0]), G * B, B, cfg, zbuf.device, step_group, step_vel, labels)

    @torch.no_grad()
    def predict_padded(self, preds_dicts, test_cfg):
        """Device-only decode: (boxes [B,S,post,9], scores [B,S,post], labels [B,S,post] int64, counts [B,S] int32)
        with S output steps; entries k >= counts[b,s] are padding.  No host synchronisation."""
        fused = self.predict_packed(preds_dicts, test_cfg)
        if fused is not None:
            packed, counts = fused
            B, S, post, _ = packed.shape
            return packed[..., :9], packed[..., 9], self._lab_cache[(packed.device, B, S, post)], counts
        srcs, vels, step_group, num_classes = self._groups(preds_dicts)
        group_radius = None
        if self._circular(test_cfg):
            group_radius = self._group_radius(test_cfg, step_group, len(srcs))
            if group_radius is None:  # a decode group per output step, each with its step's radius
                srcs, step_group = [srcs[g] for g in step_group], list(range(len(step_group)))
                group_radius = self._group_radius(test_cfg, step_group, len(srcs))
        B, hm_channels, H, W = srcs[0]["hm"].shape
        assert all(s["hm"].shape[1] == hm_channels for s in srcs) and (hm_channels == 1 or self.classify), \
            "multi-class heat-maps are decoded as their channel maximum only in the classify mode (center_head.py:589-595)"
        f = lambda k: torch.cat([s[k].float() for s in srcs], 0).contiguous() if len(srcs) > 1 else srcs[0][k].float().contiguous()  # noqa: E731
        cfg = hip_ops.make_decode_cfg(H, W, test_cfg, hm_channels=hm_channels, group_radius=group_radius)
        boxes7, scores, cell, count = hip_ops.centerpoint_decode(f("hm"), f("reg"), f("height"), f("dim"), f("rot"), cfg)
        post = cfg.nms_post_max
        G = len(srcs)
        # group-major [G*B, ...] -> [B, G, ...]
        boxes7 = boxes7.view(G, B, post, 7).transpose(0, 1)
        scores = scores.view(G, B, post).transpose(0, 1)
        cell = cell.view(G, B, post).transpose(0, 1)
        count = count.view(G, B).transpose(0, 1)
        # assemble all S output steps at once (one gather for every step's velocity instead of a Python loop)
        S = len(vels)
        dev = boxes7.device
        ck = (dev, tuple(step_group), tuple(num_classes), B, post)
        cache = self.__dict__.setdefault("_dec_cache", {})
        if ck not in cache:  # small index / label tensors, built once (no per-step host->device copies)
            offs = [0]
            for ncls in num_classes[:-1]:
                offs.append(offs[-1] + ncls)
            cache[ck] = (torch.as_tensor(step_group, device=dev),
                         torch.as_tensor(offs, dtype=torch.int64, device=dev).view(1, S, 1).expand(B, S, post).contiguous())
        gsel, labels = cache[ck]
        b7 = boxes7.index_select(1, gsel)            # [B, S, post, 7]
        idx = cell.index_select(1, gsel).long().clamp_(min=0)  # [B, S, post]
        if all(v is vels[0] for v in vels):          # timesteps == 1: the same two channels for every step
            vstack = vels[0].float().reshape(B, 1, 2, H * W).expand(B, S, 2, H * W)
        else:
            vstack = torch.stack([v.float().reshape(B, 2, H * W) for v in vels], 1)  # [B, S, 2, HW]
        v = torch.gather(vstack, 3, idx.unsqueeze(2).expand(B, S, 2, post)).transpose(2, 3)  # [B, S, post, 2]
        boxes = torch.cat([b7[..., :6], v, b7[..., 6:7]], dim=-1)
        return boxes, scores.index_select(1, gsel), labels, count.index_select(1, gsel)

    @torch.no_grad()
    def predict(self, example, preds_dicts, test_cfg, **kwargs):
        boxes, scores, labels, counts = self.predict_padded(preds_dicts, test_cfg)
        B, S, post, _ = boxes.shape
        if test_cfg.get("circular_nms", False) and bool((counts < 0).any()):
            raise RuntimeError("circular NMS: a group holds more than %d candidates above the score threshold and keeps fewer than nms_post_max_size of "
                               "them -- the reference's uncut result cannot be reproduced from the candidates taken (raise score_threshold)" % hip_ops.CIRCLE_PRE_MAX)
        valid = torch.arange(post, device=boxes.device).view(1, 1, post) < counts.unsqueeze(-1)
        metas = example.get("metadata") if isinstance(example, dict) else None
        ret = []
        for b in range(B):
            m = valid[b].reshape(-1)
            ret.append({"box3d_lidar": boxes[b].reshape(-1, 9)[m], "scores": scores[b].reshape(-1)[m],
                        "label_preds": labels[b].reshape(-1)[m],
                        "metadata": metas[b] if metas is not None and len(metas) > b else None})
        return ret
