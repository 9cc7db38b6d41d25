"""The compressed-rulebook sparse convolution (fd_spconv_tiles.hip) against the dense-table kernels it replaces, on the real rulebooks of
synthetic clouds: results must be bit-identical; run under `rocprofv3 --kernel-trace --stats` for the kernel durations (eager event timing of
the 16-channel launches is host-bound).  usage: python tools/tiles_bench.py [--batch 2] [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from futuredet_amd import build_backbone, hip_ops  # noqa: E402
from futuredet_amd.synth import seeded_state_dict, synthetic_cloud  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--scene", default="dense")
args = ap.parse_args()
dev = torch.device("cuda")
coors = []
for b in range(args.batch):
    pts = torch.from_numpy(synthetic_cloud(b, 300000, profile=args.scene)).to(dev)
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


def timed(f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / args.iters


cases = [("16->16 SubM level 0", 0, 0, 16, 16, torch.float32, True), ("16->32 strided 0->1", 0, 1, 16, 32, torch.float32, False),
         ("32->32 SubM level 1 bf16", 1, 1, 32, 32, torch.bfloat16, True)]
for name, ls, ld, cin, cout, dt, subm in cases:
    src, dst = idx[ls], idx[ld]
    nbr = src.rulebook(dst, [3, 3, 3], [1, 1, 1] if subm else [2, 2, 2], [1, 1, 1])
    n = dst.n
    pairs = int((nbr[:, :n] >= 0).sum())
    x = torch.randn((src.n, cin), device=dev).to(dt)
    wpk = hip_ops.pack_spconv_weight(torch.randn((27, cin, cout)) * (2.0 / (27 * cin)) ** 0.5, dt).to(dev)
    bias = torch.randn(cout, device=dev) * 0.1
    res = x if subm and cin == cout else None
    f = lambda: hip_ops.spconv_apply(x, wpk, bias, nbr, n, cout, residual=res, relu=True)  # noqa: E731
    hip_ops.set_tuning("spconv_tiles", -1)
    want = f().clone()
    us_old = timed(f)
    hip_ops.set_tuning("spconv_tiles", 0)
    got = f().clone()
    us_new = timed(f)
    for e in ([47, 175, 128] if os.environ.get("TILES_ABLATE") else []):
        hip_ops.set_tuning("spconv_tiles", e)
        print("   ablation %d (1: no gathers, 2: no tap loop, 4: no list loads, 8: no vmcnt(0), 16: no segment scan, 32: no store): %.1f us" % (e, timed(f)), flush=True)
    hip_ops.set_tuning("spconv_tiles", 0)
    recs, packed, cursor = nbr.tiles
    same = torch.equal(want, got)
    print("%s: %d rows, %d pairs (%.1f per row); packed list %d entries; dense-table kernel %.1f us, tiles kernel %.1f us (event-timed, eager); bit-identical: %s; max |diff| %.3g"
          % (name, n, pairs, pairs / n, int(cursor.cpu()[0]), us_old, us_new, same, float((want.float() - got.float()).abs().max())), flush=True)
    assert int(cursor.cpu()[0]) == pairs
