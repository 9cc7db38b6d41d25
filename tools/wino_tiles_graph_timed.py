"""Graph-timed (no host time between launches) comparison of the producer/consumer Winograd tile (7) with the best one-role tile (6)."""
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
for cin, cout, hw in [(128,128,180),(256,128,180),(256,256,90),(64,384,180),(512,64,180)]:
    x = torch.randn(1, hw, hw, cin, device="cuda"); w = torch.randn(cout, cin, 3, 3) * 0.02; b = torch.randn(cout, device="cuda")
    wp = hip_ops.pack_conv2d_weight_wino(w).cuda(); out = torch.empty(1, hw, hw, cout, device="cuda")
    line = "%d->%d@%d" % (cin, cout, hw)
    for dbg in (0,):
        line += " | tile7 %.1f" % (timeit(lambda: hip_ops.conv2d_wino_nhwc_f32(x, wp, b, cout, True, out=out, tile=7)))
    line += " | w6 %.1f" % timeit(lambda: hip_ops.conv2d_wino_nhwc_f32(x, wp, b, cout, True, out=out, tile=6))
    print(line, flush=True)
