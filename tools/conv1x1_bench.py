"""The two pointwise convolutions of the fp32 RPN as the sweep runs them (two maps per pass, output into a channel window of the 512-channel
concat buffer): every pointwise tile variant of fd_conv2d_nhwc_f32 / fd_conv2d_shuffle_nhwc_f32, event-timed.  usage: python tools/conv1x1_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from futuredet_amd import hip_ops  # noqa: E402


def timeit(fn, iters=30):
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


nt = hip_ops.conv2d_f32_num_tiles()
tiles = list(range(nt - 7, nt + 1))
B = 2
# deblock 0: 1x1 128 -> 256 at 180 x 180 into channels [0, 256) of the concat buffer
x = torch.randn(B, 180, 180, 128, device="cuda")
w = torch.randn(256, 128, 1, 1) * 0.05
wpk = hip_ops.pack_conv2d_weight_f32(w).cuda()
b = torch.randn(256, device="cuda")
cat = torch.empty((B, 180, 180, 512), device="cuda")
fl = 2.0 * B * 180 * 180 * 128 * 256
line = "128->256 @180 x %d maps, into a 512-channel buffer:" % B
for t in tiles:
    try:
        us = timeit(lambda: hip_ops.conv2d_nhwc_f32(x, wpk, b, 256, 1, 1, True, out=cat, co_off=0, tile=t))
        line += " | t%d %6.1f us %5.1f TF" % (t, us, fl / us / 1e6)
    except hip_ops.FutureDetHipError:
        line += " | t%d n/a" % t
print(line, flush=True)
# deblock 1: ConvTranspose2d(256 -> 256, k 2, s 2) at 90 x 90 as one 1x1 conv to 1024 virtual channels + pixel shuffle into channels [256, 512)
x2 = torch.randn(B, 90, 90, 256, device="cuda")
w2 = torch.randn(1024, 256, 1, 1) * 0.05
wpk2 = hip_ops.pack_conv2d_weight_f32(w2).cuda()
fl2 = 2.0 * B * 90 * 90 * 256 * 1024
line = "256->4x256 @90 x %d maps (pixel shuffle), into the same buffer:" % B
for t in tiles:
    try:
        us = timeit(lambda: hip_ops.conv2d_shuffle_nhwc_f32(x2, wpk2, b, 256, 2, True, out=cat, co_off=256, tile=t))
        line += " | t%d %6.1f us %5.1f TF" % (t, us, fl2 / us / 1e6)
    except hip_ops.FutureDetHipError:
        line += " | t%d n/a" % t
print(line, flush=True)
