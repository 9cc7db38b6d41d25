import os, sys, torch
sys.path.insert(0, "/root/repo")
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
line = os.environ.get("FD_LIB_PATH", "product")[-12:] + ": "
for nt in (0, 64, 32):
    hip_ops.set_tuning("conv_nt", nt)
    for cin, cout, hw in [(256,256,16),(32,256,16),(256,256,64)]:
        x = torch.randn(1, hw, hw, cin, device="cuda").bfloat16(); w = torch.randn(cout, cin, 3, 3) * 0.02; b = torch.randn(cout, device="cuda")
        wp = hip_ops.pack_conv2d_weight(w).cuda()
        us = timeit(lambda: hip_ops.conv2d_nhwc_bf16(x, wp, b, cout, 3, 1, True))
        line += "nt%d %d@%d %.1f | " % (nt, cin, hw, us)
print(line, flush=True)
