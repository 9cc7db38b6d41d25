"""Phase trace of the windowed bf16 sparse conv (tools/probes/build_win_trace.sh first; FD_LIB_PATH=tools/probes/libfd_win_trace.so)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from futuredet_amd import build_backbone, hip_ops, lib  # noqa: E402
from futuredet_amd.synth import seeded_state_dict, synthetic_cloud  # noqa: E402

dev = torch.device("cuda")
L = ctypes.CDLL(lib.LIB_PATH)
pts = torch.from_numpy(synthetic_cloud(0, 300000)).to(dev)
out = hip_ops.voxelize(pts, [0.075, 0.075, 0.2], [-54, -54, -5.0, 54, 54, 3.0], 10, 160000, want_voxels=False, want_mean=True, mean_stride=16, coor_cols=4)
m = int(out["num_voxels"].cpu()[0])
bb = build_backbone(dict(type="SpMiddleResNetFHD", num_input_features=5, ds_factor=8))
bb.load_state_dict(seeded_state_dict(bb, 7), strict=False)
bb = bb.to(dev).eval()
idx = bb.build_indexes(lambda i0: i0.mark(out["coors"][:m].contiguous()), 1, [1440, 1440, 40], dev)
names = ["window+slice staging", "operand issue + idx fetch", "MFMA block", "barrier wait", "epilogue", "whole kernel", "weight ring moves", "passes"]
for lvl in (2, 3):
    C = [16, 32, 64, 128][lvl]
    ix = idx[lvl]
    x = torch.randn((ix.n, C), device=dev).bfloat16()
    w = torch.randn((27, C, C)) * (2.0 / (27 * C)) ** 0.5
    wpk = hip_ops.pack_spconv_weight(w, torch.bfloat16).to(dev)
    nbr = ix.rulebook(ix, [3, 3, 3], [1, 1, 1], [1, 1, 1])
    bias = torch.zeros(C, device=dev)
    tr = torch.zeros((512, 8), dtype=torch.int64, device=dev)
    for _ in range(3):
        hip_ops.spconv_apply(x, wpk, bias, nbr, ix.n, C, residual=x, relu=True)
    torch.cuda.synchronize()
    assert L.fd_debug_set_win_trace(ctypes.c_void_p(tr.data_ptr())) == 0
    hip_ops.spconv_apply(x, wpk, bias, nbr, ix.n, C, residual=x, relu=True)
    torch.cuda.synchronize()
    L.fd_debug_set_win_trace(None)
    t = tr.cpu().double()
    t = t[t[:, 5] > 0]
    print("level %d C=%d: %d workgroups" % (lvl, C, len(t)))
    for i, nm in enumerate(names):
        print("   %-28s mean %10.0f  max %10.0f  (%.1f %% of the kernel)" % (nm, t[:, i].mean(), t[:, i].max(), 100 * t[:, i].mean() / t[:, 5].mean()))
