"""How local are the rulebooks?  For the SubM levels of one synthetic cloud: the share of pairs whose input row lies inside the
window [tile0 - H, tile0 + TM + H) of their output row's TM-row workgroup tile, and the share of (16-row group, tap) items
that have at least one pair outside it (those items need the global gather path in an LDS-window kernel).
usage: python tools/window_stats.py [--tm 128,256] [--halo 32,64,128]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from futuredet_amd import build_backbone, hip_ops  # noqa: E402
from futuredet_amd.synth import seeded_state_dict, synthetic_cloud  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tm", default="128,256")
ap.add_argument("--halo", default="32,64,128")
ap.add_argument("--points", type=int, default=300000)
ap.add_argument("--scene", default="dense")
args = ap.parse_args()
dev = torch.device("cuda")
pts = torch.from_numpy(synthetic_cloud(0, args.points, profile=args.scene)).to(dev)
out = hip_ops.voxelize(pts, [0.075, 0.075, 0.2], [-54, -54, -5.0, 54, 54, 3.0], 10, 160000, want_voxels=False, want_mean=True, mean_stride=16, coor_cols=4)
m = int(out["num_voxels"].cpu()[0])
bb = build_backbone(dict(type="SpMiddleResNetFHD", num_input_features=5, ds_factor=8))
bb.load_state_dict(seeded_state_dict(bb, 7), strict=False)
bb = bb.to(dev).eval()
idx = bb.build_indexes(lambda i0: i0.mark(out["coors"][:m].contiguous()), 1, [1440, 1440, 40], dev)
for lvl in range(4):
    ix = idx[lvl]
    nbr = ix.rulebook(ix, [3, 3, 3], [1, 1, 1], [1, 1, 1])[:, :ix.n].long()
    n = ix.n
    rows = torch.arange(n, device=dev)
    valid = nbr >= 0
    for tm in [int(v) for v in args.tm.split(",")]:
        t0 = (rows // tm) * tm
        for h in [int(v) for v in args.halo.split(",")]:
            inside = (nbr >= (t0 - h)[None, :]) & (nbr < (t0 + tm + h)[None, :])
            far = valid & ~inside
            pair_share = float(far.sum()) / float(valid.sum())
            g = (n + 15) // 16
            pad = g * 16 - n
            f = torch.nn.functional.pad(far, (0, pad)).view(27, g, 16).any(-1)
            v = torch.nn.functional.pad(valid, (0, pad)).view(27, g, 16).any(-1)
            print("level %d rows %6d  TM %3d halo %3d: pairs outside %.3f, (group, tap) items with a far pair %.3f of the non-empty items (%.3f of all), window rows %d"
                  % (lvl, n, tm, h, pair_share, float(f.sum()) / float(v.sum()), float(f.sum()) / float(27 * g), tm + 2 * h))
