"""Micro-benchmark of fd_conv2d_nhwc_bf16 on the RPN / head layer shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from futuredet_amd import hip_ops  # noqa: E402

dev = torch.device("cuda")
shapes = [("rpn.b0.0 256->128", 3, 1, 256, 128, 180), ("rpn.b0.k 128->128", 3, 1, 128, 128, 180), ("rpn.b1.0 128->256 s2", 3, 2, 128, 256, 180),
          ("rpn.b1.k 256->256", 3, 1, 256, 256, 90), ("deblock0 1x1 128->256", 1, 1, 128, 256, 180), ("up 1x1 256->256", 1, 1, 256, 256, 90),
          ("head.shared 512->64", 3, 1, 512, 64, 180), ("head.first 64->384", 3, 1, 64, 384, 180), ("head.final 384->11", 3, 1, 384, 11, 180)]
for name, ks, st, cin, cout, hw in shapes:
    x = torch.randn((1, hw, hw, cin), device=dev).bfloat16()
    w = torch.randn((cout, cin, ks, ks)) * (2.0 / (cin * ks * ks)) ** 0.5
    wpk = hip_ops.pack_conv2d_weight(w).to(dev)
    b = torch.zeros(cout, device=dev)
    for _ in range(3):
        y = hip_ops.conv2d_nhwc_bf16(x, wpk, b, cout, ks, st, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = hip_ops.conv2d_nhwc_bf16(x, wpk, b, cout, ks, st, True)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 20
    ho = y.shape[1]
    gf = 2.0 * ho * ho * cout * cin * ks * ks / 1e9
    print("%-26s %7.1f us  %6.1f GFLOP  %6.1f TFLOP/s" % (name, us, gf, gf / us * 1e3 / 1e3))
