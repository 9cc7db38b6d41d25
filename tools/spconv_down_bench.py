"""Micro-benchmark of the three strided sparse convolutions (16 -> 32, 32 -> 64, 64 -> 128; scn.py:110,120,130) on the real rulebooks of
synthetic clouds: pairs, fill statistics of the rulebook and the launch time per tuning variant.
usage: python tools/spconv_down_bench.py [--dtype fp32|bf16] [--batch 1] [--depth 0,2,3,4] [--tm 0,64,128]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from futuredet_amd import build_backbone, hip_ops  # noqa: E402
from futuredet_amd.synth import seeded_state_dict, synthetic_cloud  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="fp32")
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--depth", default="0")
ap.add_argument("--tm", default="0")
args = ap.parse_args()
dev = torch.device("cuda")
dt = torch.float32 if args.dtype == "fp32" else torch.bfloat16
coors = []
for b in range(args.batch):
    pts = torch.from_numpy(synthetic_cloud(b, 300000)).to(dev)
    out = hip_ops.voxelize(pts, [0.075, 0.075, 0.2], [-54, -54, -5.0, 54, 54, 3.0], 10, 160000, want_voxels=False, want_mean=True, mean_stride=16, coor_cols=4)
    m = int(out["num_voxels"].cpu()[0])
    c = out["coors"][:m].clone()
    c[:, 0] = b
    coors.append(c)
coors = torch.cat(coors).contiguous()
bb = build_backbone(dict(type="SpMiddleResNetFHD", num_input_features=5, ds_factor=8))
bb.load_state_dict(seeded_state_dict(bb, 7), strict=False)
bb = bb.to(dev).eval()
idx = bb.build_indexes(lambda i0: i0.mark(coors), args.batch, [1440, 1440, 40], dev)
for lvl in (0, 1, 2):
    cin, cout = [16, 32, 64][lvl], [32, 64, 128][lvl]
    src, dst = idx[lvl], idx[lvl + 1]
    pad = [1, 1, 1] if lvl < 2 else [0, 1, 1]
    nbr = src.rulebook(dst, [3, 3, 3], [2, 2, 2], pad)
    n = dst.n
    v = nbr[:, :n] >= 0
    pairs = int(v.sum())
    items16 = v[:, : n // 16 * 16].reshape(27, -1, 16).any(dim=2).float().mean().item()
    x = torch.randn((src.n, cin), device=dev).to(dt)
    wpk = hip_ops.pack_spconv_weight(torch.randn((27, cin, cout)) * (2.0 / (27 * cin)) ** 0.5, dt).to(dev)
    bias = torch.zeros(cout, device=dev)
    print("level %d -> %d: %d -> %d channels, %d input rows, %d output rows, %d pairs (%.1f per output row, fill %.2f of the (row, tap) slots; "
          "%.2f of the (16-row, tap) items hold a pair)" % (lvl, lvl + 1, cin, cout, src.n, n, pairs, pairs / n, pairs / (27.0 * n), items16), flush=True)
    for depth in [int(t) for t in args.depth.split(",")]:
        for tm in [int(t) for t in args.tm.split(",")]:
            hip_ops.set_tuning("v2_depth", depth)
            hip_ops.set_tuning("v2_tm", tm)
            f = lambda: hip_ops.spconv_apply(x, wpk, bias, nbr, n, cout, relu=True)  # noqa: E731
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / args.iters
            print("   %s depth=%d tm=%d: %.1f us, %.1f TFLOP/s by pairs" % (args.dtype, depth, tm, us, 2.0 * pairs * cin * cout / us * 1e-6), flush=True)
hip_ops.set_tuning("v2_depth", 0)
hip_ops.set_tuning("v2_tm", 0)
