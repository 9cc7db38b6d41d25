"""Micro-benchmark of fd_spconv_apply on the real rulebooks of one synthetic cloud (for rocprofv3 / tuning).
usage: python tools/spconv_bench.py [--level 3] [--dtype fp32] [--iters 20] [--points 300000]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from futuredet_amd import build_backbone, hip_ops, sparse  # noqa: E402
from futuredet_amd.synth import seeded_state_dict, synthetic_cloud  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--level", type=int, default=3)
ap.add_argument("--dtype", default="fp32")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--points", type=int, default=300000)
args = ap.parse_args()
dev = torch.device("cuda")
dt = torch.float32 if args.dtype == "fp32" else torch.bfloat16
pts = torch.from_numpy(synthetic_cloud(0, args.points)).to(dev)
out = hip_ops.voxelize(pts, [0.075, 0.075, 0.2], [-54, -54, -5.0, 54, 54, 3.0], 10, 160000, want_voxels=False, want_mean=True,
                       mean_stride=16, coor_cols=4)
m = int(out["num_voxels"].cpu()[0])
bb = build_backbone(dict(type="SpMiddleResNetFHD", num_input_features=5, ds_factor=8))
bb.load_state_dict(seeded_state_dict(bb, 7), strict=False)
bb = bb.to(dev).eval()
idx = bb.build_indexes(lambda i0: i0.mark(out["coors"][:m].contiguous()), 1, [1440, 1440, 40], dev)
lvl = args.level
C = [16, 32, 64, 128][lvl]
ix = idx[lvl]
nbr = ix.rulebook(ix, [3, 3, 3], [1, 1, 1], [1, 1, 1])
pairs = int((nbr[:, :ix.n] >= 0).sum())
x = torch.randn((ix.n, C), device=dev).to(dt)
w = torch.randn((27, C, C)) * (2.0 / (27 * C)) ** 0.5
wpk = hip_ops.pack_spconv_weight(w, dt).to(dev)
bias = torch.zeros(C, device=dev)
for _ in range(3):
    y = hip_ops.spconv_apply(x, wpk, bias, nbr, ix.n, C, residual=x, relu=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.iters):
    y = hip_ops.spconv_apply(x, wpk, bias, nbr, ix.n, C, residual=x, relu=True)
e1.record()
torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / args.iters
s = 4 if dt == torch.float32 else 2
bgs = s * pairs * 2 * C + 8 * pairs + s * 27 * C * C
print("level %d C=%d n=%d pairs=%d (%.1f/row) %s: %.1f us  pair-GFLOP/s %.0f  B_gs %.0f GB/s" %
      (lvl, C, ix.n, pairs, pairs / ix.n, args.dtype, us, 2.0 * pairs * C * C / us / 1e3, bgs / us / 1e3))
# ---- work distribution over 128-row tiles (load-balance analysis)
valid = (nbr[:, :ix.n] >= 0)
nt = (ix.n + 127) // 128
pad = nt * 128 - ix.n
v = torch.nn.functional.pad(valid, (0, pad)).view(27, nt, 128).sum(2)      # [27, nt] pairs per tap per tile
groups = ((v + 15) // 16).sum(0).cpu().numpy()                              # MFMA groups per tile
print("tiles %d: groups/tile mean %.1f max %d min %d; padded-slot efficiency %.3f" %
      (nt, groups.mean(), groups.max(), groups.min(), pairs / (groups.sum() * 16.0)))
for slots in (256, 512, 768):
    # blocks resident at once, contiguous chunk per XCD (xcd_swizzle) vs round-robin
    import numpy as np
    per = np.zeros(slots)
    order = np.arange(nt)
    for name, assign in (("round-robin", order % slots), ("xcd-chunk", (order * slots // nt))):
        load = np.bincount(assign, weights=groups, minlength=slots)
        print("  %d slots %-11s: max load %.0f vs mean %.1f -> efficiency %.2f" % (slots, name, load.max(), groups.sum() / slots, groups.sum() / slots / load.max()))
