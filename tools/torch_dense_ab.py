#!/usr/bin/env python
"""A/B timing of the RPN neck + CenterHead: the hand-written MFMA convolution plan (the product's only eval path) against the same
folded stacks run through PyTorch-ROCm (F.conv2d -> MIOpen).  The PyTorch executor lives HERE, not in the package: the product has one
dense path (round-4 review).  Usage: python tools/torch_dense_ab.py [--variant forecast_n0] [--dtype fp32|bf16] [--size 180]
Round-4 figure for the whole sweep with this executor in place of the plan: 241.9 vs 270.2 sweeps/s (profiles/round4_measure_round_final.txt)."""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from futuredet_amd import build_detector  # noqa: E402
from futuredet_amd.configs import centerpoint_config  # noqa: E402
from futuredet_amd.nn_utils import fold_stack  # noqa: E402
from futuredet_amd.synth import seeded_state_dict, tame_box_dims  # noqa: E402


def run_folded(f, x):
    if f.transposed:
        y = F.conv_transpose2d(x, f.weight, f.bias, stride=f.stride, padding=f.padding)
    else:
        y = F.conv2d(x, f.weight, f.bias, stride=f.stride, padding=f.padding)
    return F.relu_(y) if f.relu else y


def torch_rpn(rpn, dtype):
    blocks = [fold_stack(b._modules.values(), dtype, False) for b in rpn.blocks]
    deblocks = [fold_stack(d._modules.values(), dtype, False) for d in rpn.deblocks]

    def fwd(x):
        ups = []
        for i, stack in enumerate(blocks):
            for f in stack:
                x = run_folded(f, x)
            if i >= rpn.first_up:
                y = x
                for f in deblocks[i - rpn.first_up]:
                    y = run_folded(f, y)
                ups.append(y)
        return torch.cat(ups, 1) if ups else x
    return fwd


def torch_head(head, dtype):
    shared = fold_stack(head.shared_conv, dtype, False)
    tasks = []
    for t in head.tasks:
        pre = fold_stack(t.forecast_conv, dtype, False) if t.forecast_feature else []
        tasks.append((pre, {h: fold_stack(list(getattr(t, h)._modules.values()), dtype, False) for h in t.heads}))

    def fwd(x):
        for f in shared:
            x = run_folded(f, x)
        rets = []
        for i, (pre, hs) in enumerate(tasks):
            y = torch.cat([x, rets[i - 1]["feats"]], 1) if (i and head.forecast_feature) else x
            d = {}
            for f in pre:
                y = run_folded(f, y)
            if pre:
                d["feats"] = y
            for h, st in hs.items():
                z = y
                for f in st:
                    z = run_folded(f, z)
                d[h] = z
            rets.append(d)
        return rets
    return fwd


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="forecast_n0")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--size", type=int, default=180)
    a = ap.parse_args()
    dt = torch.float32 if a.dtype == "fp32" else torch.bfloat16
    cfg = centerpoint_config(a.variant, "car")
    net = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    net.load_state_dict(tame_box_dims(seeded_state_dict(net, 7)), strict=False)
    net = net.cuda().eval().set_precision(dt)
    x = torch.randn((1, 256, a.size, a.size), device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        t_rpn, t_head = torch_rpn(net.neck, dt), torch_head(net.bbox_head, dt)
        ms_plan = timed(lambda: net.bbox_head(net.neck(x)))
        ms_torch = timed(lambda: t_head(t_rpn(x)))
        y0, y1 = net.neck(x).float(), t_rpn(x).float()
    print("neck + head, %s, %dx%d: plan %.3f ms, PyTorch/MIOpen %.3f ms; neck outputs differ by %.2e (max abs)" % (
        a.dtype, a.size, a.size, ms_plan, ms_torch, float((y0 - y1).abs().max())))


if __name__ == "__main__":
    main()
