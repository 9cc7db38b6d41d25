"""Graph-timed bf16 3x3 / 1x1 conv layers of the RPN / head (use FD_LIB_PATH to compare two builds on one box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from futuredet_amd import hip_ops
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 0  # conv_nt override: 0 = heuristic, 64 / 32 = narrower channel blocks
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 1     # maps per launch
hip_ops.set_tuning("conv_nt", nt)
line = "conv_nt=%d B=%d: " % (nt, NB)
for cin, cout, hw, ks, st in [(128,128,180,3,1),(256,128,180,3,1),(256,256,90,3,1),(512,64,180,3,1),(64,384,180,3,1),(128,256,180,3,2),(128,256,180,1,1)]:
    x = torch.randn(NB, hw, hw, cin, device="cuda").bfloat16(); w = torch.randn(cout, cin, ks, ks) * 0.02; b = torch.randn(cout, device="cuda")
    wp = hip_ops.pack_conv2d_weight(w).cuda()
    us = timeit(lambda: hip_ops.conv2d_nhwc_bf16(x, wp, b, cout, ks, st, True))
    line += "%d->%d@%d k%ds%d %.1f us | " % (cin, cout, hw, ks, st, us)
print(line, flush=True)
