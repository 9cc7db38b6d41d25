"""Phase accounting of the fp32 32 -> 32 sparse conv (experiment copy tools/probes/c32_ablation_experiment.hip, tuning build):
   tools/probes/build_exp.sh fd_spconv_c32 trace -DFD_C32_TRACE --src tools/probes/c32_ablation_experiment.hip
   FD_LIB_PATH=tools/probes/libfd_fd_spconv_c32_trace.so python tools/c32_trace.py [--mode uniform|product]
Thread 0 of every workgroup accumulates shader-clock cycles per phase over all of its chunks."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from futuredet_amd import build_backbone, hip_ops, lib  # noqa: E402
from futuredet_amd.synth import seeded_state_dict, synthetic_cloud  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="product")
args = ap.parse_args()
L = lib.load()
L.fd_debug_set_c32_trace.restype = ctypes.c_int
L.fd_debug_set_c32_trace.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda")
pts = torch.from_numpy(synthetic_cloud(0, 300000)).to(dev)
out = hip_ops.voxelize(pts, [0.075, 0.075, 0.2], [-54, -54, -5.0, 54, 54, 3.0], 10, 160000, want_voxels=False, want_mean=True, mean_stride=16, coor_cols=4)
m = int(out["num_voxels"].cpu()[0])
bb = build_backbone(dict(type="SpMiddleResNetFHD", num_input_features=5, ds_factor=8))
bb.load_state_dict(seeded_state_dict(bb, 7), strict=False)
bb = bb.to(dev).eval()
idx = bb.build_indexes(lambda i0: i0.mark(out["coors"][:m].contiguous()), 1, [1440, 1440, 40], dev)
ix = idx[1]
C = 32
x = torch.randn((ix.n, C), device=dev)
wpk = hip_ops.pack_spconv_weight(torch.randn((27, C, C)) * 0.05).to(dev)
nbr = ix.rulebook(ix, [3, 3, 3], [1, 1, 1], [1, 1, 1])
kw = dict(residual=x, relu=True)
if args.mode == "uniform":
    kw["balanced"] = False
trace = torch.zeros((8192 * 16,), dtype=torch.int64, device=dev)
for _ in range(3):
    hip_ops.spconv_apply(x, wpk, None, nbr, ix.n, C, **kw)
torch.cuda.synchronize()
assert L.fd_debug_set_c32_trace(trace.data_ptr()) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
hip_ops.spconv_apply(x, wpk, None, nbr, ix.n, C, **kw)
e1.record()
torch.cuda.synchronize()
L.fd_debug_set_c32_trace(None)
t = trace.cpu().numpy().reshape(-1, 16).astype(np.float64)
t = t[t[:, 8] > 0]
names = ["stage + accumulator init", "barrier 1 wait", "compaction", "barrier 2 wait", "item list + first gathers", "item loop", "loop-end barrier wait", "epilogue"]
span = t[:, 11].max() - t[:, 10].min()
us = 1e3 * e0.elapsed_time(e1)
print("32->32, %d rows: %.1f us, %d workgroups, %.2f chunks each; kernel span %.0f cycles (%.2f GHz); workgroup life mean %.0f max %.0f"
      % (ix.n, us, len(t), t[:, 9].mean(), span, span / us / 1e3, t[:, 8].mean(), t[:, 8].max()))
for i, n in enumerate(names):
    print("   %-28s %8.0f cycles per workgroup  %7.0f per chunk  %5.1f %% of its life" % (n, t[:, i].mean(), (t[:, i] / t[:, 9]).mean(), 100 * t[:, i].mean() / t[:, 8].mean()))
