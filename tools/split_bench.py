"""fd_spconv_apply on the real rulebooks of one synthetic cloud: native fp32 kernel vs the split-operand (3 x bf16) kernel per
level, with the split kernel's row-group variants.  usage: python tools/split_bench.py [--levels 1,2,3] [--rg 0,1,2] [--iters 20]
Also times the strided convolutions between levels (32->64, 64->128) and the 3x1x1 extra convolution.  Leaving parts of the kernel out (what
does a part cost): tools/probes/build_split_exp.sh <mask>... then FD_LIB_PATH=tools/probes/libfd_split_exp<mask>.so python tools/split_bench.py"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from futuredet_amd import build_backbone, hip_ops  # noqa: E402
from futuredet_amd.synth import seeded_state_dict, synthetic_cloud  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--levels", default="1,2,3")
ap.add_argument("--rg", default="0,1,2")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--points", type=int, default=300000)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--strided", type=int, default=1)
args = ap.parse_args()
dev = torch.device("cuda")
pts = torch.from_numpy(synthetic_cloud(args.seed, args.points)).to(dev)
out = hip_ops.voxelize(pts, [0.075, 0.075, 0.2], [-54, -54, -5.0, 54, 54, 3.0], 10, 160000, want_voxels=False, want_mean=True,
                       mean_stride=16, coor_cols=4)
m = int(out["num_voxels"].cpu()[0])
bb = build_backbone(dict(type="SpMiddleResNetFHD", num_input_features=5, ds_factor=8))
bb.load_state_dict(seeded_state_dict(bb, 7), strict=False)
bb = bb.to(dev).eval()
idx = bb.build_indexes(lambda i0: i0.mark(out["coors"][:m].contiguous()), 1, [1440, 1440, 40], dev)


def timed(fn, iters):
    for _ in range(3):
        y = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        y = fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters, y


def case(tag, src, dst, ks, st, pd, cin, cout, residual):
    K = ks[0] * ks[1] * ks[2]
    x = torch.randn((src.n, cin), device=dev)
    w = torch.randn((K, cin, cout)) * (2.0 / (K * cin)) ** 0.5
    bias = torch.zeros(cout, device=dev)
    nbr = src.rulebook(dst, ks, st, pd)
    pairs = int((nbr[:, :dst.n] >= 0).sum())
    res = torch.randn((dst.n, cout), device=dev) if residual else None
    wn = hip_ops.pack_spconv_weight(w).to(dev)
    ws = hip_ops.pack_spconv_weight(w, hip_ops.F32_SPLIT).to(dev)
    us_n, y_n = timed(lambda: hip_ops.spconv_apply(x, wn, bias, nbr, dst.n, cout, residual=res, relu=True), args.iters)
    xp = hip_ops.rows_to_planes(x)
    resp = hip_ops.rows_to_planes(res) if residual else None
    print("%s %3d->%3d n_out=%6d pairs=%7d (%.1f/row, fill %.2f) native fp32      : %7.1f us  %6.1f TFLOP/s" %
          (tag, cin, cout, dst.n, pairs, pairs / max(dst.n, 1), pairs / max(K * dst.n, 1), us_n, 2.0 * pairs * cin * cout / us_n / 1e6), flush=True)
    for rg in [int(v) for v in args.rg.split(",")]:
        hip_ops.set_tuning("split_rg", rg)
        us_s, y_s = timed(lambda: hip_ops.spconv_apply(xp, ws, bias, nbr, dst.n, cout, residual=resp, relu=True, mode="p2p"), args.iters)
        d = float((hip_ops.planes_to_rows(y_s) - y_n).abs().max() / y_n.abs().max().clamp_min(1.0))
        print("%s %3d->%3d                                                   split rg=%d        : %7.1f us  %6.1f TFLOP/s  (x%.2f)  max|d| vs native %.2e" %
              (tag, cin, cout, rg, us_s, 2.0 * pairs * cin * cout / us_s / 1e6, us_n / us_s, d), flush=True)
    hip_ops.set_tuning("split_rg", 0)


C = [16, 32, 64, 128]
for lvl in [int(v) for v in args.levels.split(",")]:
    case("level %d subm" % lvl, idx[lvl], idx[lvl], [3, 3, 3], [1, 1, 1], [1, 1, 1], C[lvl], C[lvl], True)
if args.strided:
    stages = bb._stages()
    for lvl in (2, 3, 4):
        conv = stages[lvl][0]
        ks, st, pd = conv.geometry()
        case("level %d->%d   " % (lvl - 1, lvl), idx[lvl - 1], idx[lvl], list(ks), list(st), list(pd), conv.in_channels, conv.out_channels, False)
