"""Per-layer timing of the RPN / CenterHead convolutions in fp32: hand-written MFMA conv (fd_conv2d_nhwc_f32, every tile
candidate) vs torch / MIOpen.  usage: python tools/dense_fp32_bench.py [--tiles 0,1,2,...]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from futuredet_amd import hip_ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tiles", default="0")
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
LAYERS = [("256->128 @180 3x3", 256, 128, 180, 3, 1), ("128->128 @180 3x3", 128, 128, 180, 3, 1), ("128->256 @180 3x3 s2", 128, 256, 180, 3, 2),
          ("256->256 @90 3x3", 256, 256, 90, 3, 1), ("128->256 @180 1x1", 128, 256, 180, 1, 1), ("256->256 @90 1x1", 256, 256, 90, 1, 1),
          ("512->64 @180 3x3", 512, 64, 180, 3, 1), ("64->384 @180 3x3", 64, 384, 180, 3, 1), ("384->11 @180 3x3", 384, 11, 180, 3, 1),
          ("384->23 @180 3x3", 384, 23, 180, 3, 1)]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


for name, cin, cout, hw, ks, st in LAYERS:
    x = torch.randn(1, cin, hw, hw, device="cuda")
    w = torch.randn(cout, cin, ks, ks, device="cuda") * 0.02
    b = torch.randn(cout, device="cuda")
    pad = 1 if ks == 3 else 0
    ho = (hw + 2 * pad - ks) // st + 1
    fl = 2.0 * ho * ho * cout * cin * ks * ks
    us = timeit(lambda: F.conv2d(x, w, None, stride=st, padding=pad), args.iters)
    line = "%-22s %6.2f GF | MIOpen %7.1f us %6.1f TF" % (name, fl / 1e9, us, fl / us / 1e6)
    xn = x.permute(0, 2, 3, 1).contiguous()
    wpk = hip_ops.pack_conv2d_weight_f32(w.cpu()).cuda()
    out = torch.empty((1, ho, ho, cout), device="cuda")
    for t in [int(v) for v in args.tiles.split(",")]:
        us = timeit(lambda: hip_ops.conv2d_nhwc_f32(xn, wpk, b, cout, ks, st, True, out=out, tile=t), args.iters)
        line += " | t%d %7.1f us %6.1f TF" % (t, us, fl / us / 1e6)
    if ks == 1:  # pointwise GEMM variants (tile ids after the direct-kernel tiles) and the direct kernel with one tap (tile 1)
        nt = hip_ops.conv2d_f32_num_tiles()
        for t in [1] + list(range(nt - 3, nt + 1)):
            us = timeit(lambda: hip_ops.conv2d_nhwc_f32(xn, wpk, b, cout, ks, st, True, out=out, tile=t), args.iters)
            line += " | %s %7.1f us %6.1f TF" % ("direct" if t == 1 else "p%d" % (t - nt + 4), us, fl / us / 1e6)
    if ks == 3 and st == 1:
        wpw = hip_ops.pack_conv2d_weight_wino(w.cpu()).cuda()
        for t in range(1, hip_ops.conv2d_wino_f32_num_tiles() + 1):
            us = timeit(lambda: hip_ops.conv2d_wino_nhwc_f32(xn, wpw, b, cout, True, out=out, tile=t), args.iters)
            line += " | w%d %7.1f us %6.1f TF" % (t, us, fl / us / 1e6)
    print(line, flush=True)
