"""Locality of the SubM rulebooks in index order, as the LDS-window kernel (fd_spconv_bf16win.hip) sees it: for a workgroup tile of TM rows
with a window of HALO rows on either side, the fraction of (32-row group, tap) items with at least one neighbour outside the window (those
take the global gather), the fraction of pairs outside, and the distribution of the distance |neighbour row - own row| of the outside pairs.
usage: python tools/window_stats.py [--points 300000] [--scene dense]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from futuredet_amd import build_backbone, hip_ops  # noqa: E402
from futuredet_amd.synth import seeded_state_dict, synthetic_cloud  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=300000)
ap.add_argument("--scene", default="dense")
ap.add_argument("--strips", default="1", help="comma list of S: evaluate the index order 'super-strips of S tile rows, x-major inside a strip' (1 = the shipped order: "
                "8 x 8 column tiles, tile rows major) by re-ranking the rows offline -- VERDICT r5 #4")
ap.add_argument("--shapes", default="", help="comma list of TM:HALO pairs to print (default: the sweep of round 5)")
args = ap.parse_args()
dev = torch.device("cuda")
pts = torch.from_numpy(synthetic_cloud(0, args.points, profile=args.scene)).to(dev)
out = hip_ops.voxelize(pts, [0.075, 0.075, 0.2], [-54, -54, -5.0, 54, 54, 3.0], 10, 160000, want_voxels=False, want_mean=True, mean_stride=16, coor_cols=4)
m = int(out["num_voxels"].cpu()[0])
bb = build_backbone(dict(type="SpMiddleResNetFHD", num_input_features=5, ds_factor=8)).to(dev).eval()
idx = bb.build_indexes(lambda i0: i0.mark(out["coors"][:m].contiguous()), 1, [1440, 1440, 40], dev)
def reorder(ix, nbr, S):
    """rows re-ranked by (b, tile row // S, tile column, tile row % S, y % 8, x % 8, z): returns the rulebook in the new order"""
    if S == 1:
        return nbr
    c = ix.coords[: ix.n].long()  # (b, z, y, x)
    b, z, y, x = c[:, 0], c[:, 1], c[:, 2], c[:, 3]
    ty, tx = y >> 3, x >> 3
    key = ((((((b * 4096 + ty // S) * 4096 + tx) * S + ty % S) * 8 + (y & 7)) * 8 + (x & 7)) * 64 + z)
    order = torch.argsort(key)                      # new position -> old row
    rank = torch.empty_like(order)
    rank[order] = torch.arange(ix.n, device=order.device)   # old row -> new position
    out = nbr[:, order]
    return torch.where(out >= 0, rank[out.clamp(min=0)], out)


for lvl in (1, 2, 3):
    ix = idx[lvl]
    n = ix.n
    nbr0 = ix.rulebook(ix, [3, 3, 3], [1, 1, 1], [1, 1, 1])[:, :n].long()
    for S in [int(v) for v in args.strips.split(",")]:
        nbr = reorder(ix, nbr0, S)
        rows = torch.arange(n, device=dev)[None, :].expand(27, n)
        have = nbr >= 0
        dist = (nbr - rows).abs()[have]
        qs = [50, 80, 90, 95, 98, 99]
        print("level %d S=%d rows %d pairs %d: |neighbour - row| percentiles %s = %s" % (lvl, S, n, int(have.sum()), qs, [int(np.percentile(dist.cpu().numpy(), q)) for q in qs]), flush=True)
        per_cu = (n + 255) // 256
        shapes = [tuple(int(v) for v in sh.split(":")) for sh in args.shapes.split(",") if sh] or \
                 [(tm, halo) for tm in sorted({128, 256, 384, 512, ((per_cu + 127) // 128) * 128}) for halo in (32, 64, 128, 192, 256)]
        for tm, halo in shapes:
            tile0 = (rows // tm) * tm
            outside = have & ((nbr < tile0 - halo) | (nbr >= tile0 + tm + halo))
            pad = (-n) % 32
            o = torch.nn.functional.pad(outside, (0, pad)).view(27, -1, 32).any(-1)
            print("  S=%d TM %4d HALO %3d: pairs outside %.3f, (32-row group, tap) items with an outside pair %.3f" % (
                S, tm, halo, float(outside.sum()) / float(have.sum()), float(o.float().mean())), flush=True)
