"""CenterHead (det3d/models/bbox_heads/center_head.py:81-174 SepHead, :232-390 CenterHead, :542-747 predict).

Constructor signature and state_dict keys follow the reference (shared_conv.{0,1}.*, tasks.{i}.{reg,height,dim,
rot,vel,hm}.{0,1,3}.*, tasks.{i}.forecast_conv.*, bev_conv.*).  Only the branches reachable from the shipped
configs exist: "standard" (n0 / n3: one task, velocity split per timestep) and "dense" (n3dtf / n3dtfm: one task
per timestep, optional chained forecast features and BEV-map branch).  predict() runs the HIP decode + rotated
NMS (fd_centerpoint_decode) for all (sample, heat-map) groups in one call; the loss is training-only and out of
scope of this path.
"""
import copy
import logging

import torch
from torch import nn

from . import hip_ops
from .nn_utils import FoldedConv, Sequential, fold_stack, kaiming_init, weights_version
from .registry import HEADS


class SepHead(nn.Module):
    def __init__(self, in_channels, heads, head_conv=64, final_kernel=1, bn=False, init_bias=-2.19, two_stage=False,
                 forecast_feature=False, wide_head=False, **kwargs):
        super().__init__(**kwargs)
        assert not two_stage and not wide_head, "two_stage / wide_head are False in every shipped config"
        self.heads = heads
        self.forecast_feature = forecast_feature
        if self.forecast_feature:
            self.forecast_conv = nn.Sequential(
                nn.Conv2d(in_channels, head_conv, kernel_size=3, padding=1, bias=True), nn.BatchNorm2d(head_conv),
                nn.ReLU(inplace=True),
                nn.Conv2d(head_conv, head_conv, kernel_size=3, padding=1, bias=True), nn.BatchNorm2d(head_conv),
                nn.ReLU(inplace=True))
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
        self._fused = None
        self.register_load_state_dict_post_hook(lambda m, k: m.invalidate_caches())

    def invalidate_caches(self):
        self._fused = None
        self.__dict__.pop("_wv_tensors", None)

    def _apply(self, fn, *a, **kw):
        self.invalidate_caches()
        return super()._apply(fn, *a, **kw)

    def forward_modules(self, x):
        ret = {}
        if self.forecast_feature:
            x = self.forecast_conv(x)
            ret["feats"] = x
        for head in self.heads:
            ret[head] = self.__getattr__(head)(x)
        return ret

    def _fuse(self, dtype, channels_last):
        """All heads read the same map: their first convs become one conv (Cout = 64*nheads, BN folded, ReLU) and
        their final convs one block-diagonal conv, so a task costs 2 launches instead of 12."""
        key = (dtype, channels_last, next(self.parameters()).device, weights_version(self))
        if self._fused is not None and self._fused[0] == key:
            return self._fused[1:]
        names = list(self.heads)
        pre = fold_stack(self.forecast_conv, dtype, channels_last) if self.forecast_feature else []
        firsts, finals = [], []
        for h in names:
            mods = list(getattr(self, h)._modules.values())
            st = fold_stack(mods, torch.float32, False)
            assert len(st) == 2, "fusion assumes num_conv == 2 (all shipped configs)"
            firsts.append(st[0])
            finals.append(st[1])
        w1 = torch.cat([f.weight for f in firsts], 0)
        b1 = torch.cat([f.bias for f in firsts], 0)
        hc = firsts[0].weight.shape[0]
        couts = [f.weight.shape[0] for f in finals]
        k = finals[0].weight.shape[-1]
        w2 = torch.zeros((sum(couts), hc * len(names), k, k), dtype=torch.float32, device=w1.device)
        o = 0
        for i, f in enumerate(finals):
            w2[o:o + couts[i], i * hc:(i + 1) * hc] = f.weight
            o += couts[i]
        b2 = torch.cat([f.bias for f in finals], 0)
        mf = torch.channels_last if channels_last else torch.contiguous_format
        conv1 = (w1.to(dtype).contiguous(memory_format=mf), b1.to(dtype), firsts[0].padding)
        conv2 = (w2.to(dtype).contiguous(memory_format=mf), b2.to(dtype), finals[0].padding)
        self._fused = (key, pre, conv1, conv2, names, couts)
        return self._fused[1:]

    def forward_fused(self, x, dtype, channels_last):
        pre, (w1, b1, p1), (w2, b2, p2), names, couts = self._fuse(dtype, channels_last)
        ret = {}
        for conv in pre:
            x = conv(x)
        if self.forecast_feature:
            ret["feats"] = x
        y = FoldedConv(w1, b1, 1, p1, True)(x)
        z = FoldedConv(w2, b2, 1, p2, False)(y)
        o = 0
        for name, c in zip(names, couts):
            ret[name] = z[:, o:o + c]
            o += c
        return ret

    def forward(self, x):
        return self.forward_modules(x)



def _drop_caches(module, incompatible_keys=None):
    module._folded = None
    module._plan = None
    module.__dict__.pop("_wv_tensors", None)


@HEADS.register_module
class CenterHead(nn.Module):
    def __init__(self, in_channels=[128, ], tasks=[], dataset="nuscenes", weight=0.25, code_weights=[], common_heads=dict(),
                 logger=None, init_bias=-2.19, share_conv_channel=64, num_hm_conv=2, dcn_head=False, timesteps=1,
                 two_stage=False, reverse=False, sparse=False, dense=False, bev_map=False, forecast_feature=False,
                 classify=True, wide_head=False):
        super().__init__()
        unsupported = dict(dcn_head=dcn_head, two_stage=two_stage, reverse=reverse, sparse=sparse, classify=classify,
                           wide_head=wide_head)
        on = [k for k, v in unsupported.items() if v]
        if on:
            raise NotImplementedError("CenterHead options %s are False in every shipped centerpoint config and are not "
                                      "part of the inference hot path" % on)
        self.two_stage, self.reverse, self.sparse, self.dense = two_stage, reverse, sparse, dense
        self.bev_map, self.forecast_feature, self.classify, self.wide_head = bev_map, forecast_feature, classify, wide_head
        self.target_timesteps = 7
        self.standard = not dense
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
        if self.dense:
            self.num_classes = self.timesteps * [1]
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
                if not self.dense and head in ["vel", "rvel"]:
                    heads[head] = (self.timesteps * heads[head][0], heads[head][1])
            heads.update(dict(hm=(num_cls, num_hm_conv)))
            cin = 2 * share_conv_channel if (i != 0 and self.forecast_feature) else share_conv_channel
            self.tasks.append(SepHead(cin, heads, bn=True, init_bias=init_bias, final_kernel=3, two_stage=self.two_stage,
                                      forecast_feature=self.forecast_feature, wide_head=self.wide_head))
        self.compute_dtype = torch.float32
        self.channels_last = False
        self._folded = None
        self._plan = None
        self.use_hip_conv = True
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
        if self.training:
            return self.forward_modules(x, bev_map)
        if x.is_cuda and self.use_hip_conv and self.compute_dtype in (torch.bfloat16, torch.float32):
            ver = (weights_version(self), self.compute_dtype)
            if self._plan is None or self._plan[0] != ver:
                from .dense_bf16 import HeadPlan

                try:
                    self._plan = (ver, HeadPlan(self, self.compute_dtype))
                except ValueError:
                    self._plan = (ver, None)
            if self._plan[1] is not None:
                return self._plan[1](x.to(self.compute_dtype).permute(0, 2, 3, 1).contiguous(), bev_map)
        dt, cl = self.compute_dtype, self.channels_last
        key = (dt, cl, next(self.parameters()).device, weights_version(self.shared_conv) + (weights_version(self.bev_conv) if self.bev_map else 0))
        if self._folded is None or self._folded[0] != key:
            shared = fold_stack(self.shared_conv, dt, cl)
            bev = fold_stack(self.bev_conv, dt, cl) if self.bev_map else None
            self._folded = (key, shared, bev)
        _, shared, bev = self._folded
        x = x.to(dt)
        if cl:
            x = x.contiguous(memory_format=torch.channels_last)
        for conv in shared:
            x = conv(x)
        if self.bev_map:
            y = bev_map.to(dt)
            for conv in bev:
                y = conv(y)
            x = x + y
        rets = []
        for i, task in enumerate(self.tasks):
            inp = torch.cat([x, rets[i - 1]["feats"]], dim=1) if (i != 0 and self.forecast_feature) else x
            rets.append(task.forward_fused(inp, dt, cl))
        return rets

    def loss(self, example, preds_dicts, **kwargs):
        raise NotImplementedError("training losses (center_head.py:396-539) are outside the inference hot path")

    # ----------------------------------------------------------------------------------------------- predict
    def _groups(self, preds_dicts):
        """-> (list of per-group source dicts, vel tensor per output step, step->group map, num_classes per step)."""
        if self.standard:  # center_head.py:559-570
            pd = preds_dicts[0]
            vels = [pd["vel"][:, 2 * i:2 * i + 2] for i in range(self.timesteps)]
            if len(vels) == 1:
                vels = self.target_timesteps * vels
            return [pd], vels, [0] * len(vels), [1] * self.target_timesteps
        vels = [pd["vel"] for pd in preds_dicts]  # center_head.py:606-607
        return list(preds_dicts), vels, list(range(len(preds_dicts))), list(self.num_classes)

    @torch.no_grad()
    def predict_packed(self, preds_dicts, test_cfg):
        """Device-only decode straight from the convolution plan's NHWC output buffer: (packed [B,S,post,11] float32 rows
        x y z w l h vx vy yaw score label, counts [B,S] int32) -- seven launches (keys, select, footprints, IoU mask, sweep, gather,
        assembly), no torch kernel in between.  None when the maps did not come from the plan (torch path, bev_map head)."""
        raws = [getattr(pd, "raw", None) for pd in preds_dicts]
        if any(r is None for r in raws) or any(r[0] is not raws[0][0] or r[2] != raws[0][2] for r in raws):
            return None
        if test_cfg.get("per_class_nms", False) or test_cfg.get("circular_nms", False):
            raise NotImplementedError("only rotated NMS is configured in the shipped test_cfg")
        zbuf, _, where = raws[0]
        T, B, H, W, C = zbuf.shape
        assert where["hm"][1] == 1, "single-class heat-maps (every shipped task has one class)"
        if self.standard:  # center_head.py:559-570: one task, step s = its boxes + velocity channels 2s, 2s+1 (all steps share them when timesteps == 1)
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
            num_classes = list(self.num_classes)
        labels, acc = [], 0
        for ncls in num_classes:
            labels.append(acc)
            acc += ncls
        cache = self.__dict__.setdefault("_lab_cache", {})
        ck = (zbuf.device, B, S, int(test_cfg["nms"]["nms_post_max_size"]))
        if ck not in cache:  # labels do not depend on the data (built in the eager set-up pass, before any graph capture)
            cache[ck] = torch.as_tensor(labels, dtype=torch.int64, device=zbuf.device).view(1, S, 1).expand(B, S, ck[3]).contiguous()
        flat = zbuf[:G].reshape(G * B, H, W, C)
        cfg = hip_ops.make_decode_cfg(H, W, test_cfg)
        views = [hip_ops.nhwc_channel_view(flat, where[k][0]) for k in ("hm", "reg", "height", "dim", "rot")]
        boxes7, scores, cell, count = hip_ops.centerpoint_decode_views(views, G * B, cfg, zbuf.device)
        return hip_ops.assemble_detections(boxes7, scores, cell, count, hip_ops.nhwc_channel_view(flat, where["vel"][0]), B, cfg.nms_post_max,
                                           step_group, step_vel, labels)

    @torch.no_grad()
    def predict_padded(self, preds_dicts, test_cfg):
        """Device-only decode: (boxes [B,S,post,9], scores [B,S,post], labels [B,S,post] int64, counts [B,S] int32)
        with S output steps; entries k >= counts[b,s] are padding.  No host synchronisation."""
        fused = self.predict_packed(preds_dicts, test_cfg)
        if fused is not None:
            packed, counts = fused
            B, S, post, _ = packed.shape
            return packed[..., :9], packed[..., 9], self._lab_cache[(packed.device, B, S, post)], counts
        if test_cfg.get("per_class_nms", False) or test_cfg.get("circular_nms", False):
            raise NotImplementedError("only rotated NMS is configured in the shipped test_cfg")
        srcs, vels, step_group, num_classes = self._groups(preds_dicts)
        B, _, H, W = srcs[0]["hm"].shape
        assert all(s["hm"].shape[1] == 1 for s in srcs), "single-class heat-maps (every shipped task has one class)"
        f = lambda k: torch.cat([s[k].float() for s in srcs], 0).contiguous() if len(srcs) > 1 else srcs[0][k].float().contiguous()  # noqa: E731
        cfg = hip_ops.make_decode_cfg(H, W, test_cfg)
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
        valid = torch.arange(post, device=boxes.device).view(1, 1, post) < counts.unsqueeze(-1)
        metas = example.get("metadata") if isinstance(example, dict) else None
        ret = []
        for b in range(B):
            m = valid[b].reshape(-1)
            ret.append({"box3d_lidar": boxes[b].reshape(-1, 9)[m], "scores": scores[b].reshape(-1)[m],
                        "label_preds": labels[b].reshape(-1)[m],
                        "metadata": metas[b] if metas is not None and len(metas) > b else None})
        return ret
