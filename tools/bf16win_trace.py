"""Phase trace of the bf16 LDS-window sparse conv (fd_spconv_bf16win.hip) on the real rulebooks of one synthetic cloud.
   tools/probes/build_exp.sh fd_spconv_bf16win trace -DFD_WIN_TRACE && FD_LIB_PATH=tools/probes/libfd_fd_spconv_bf16win_trace.so python tools/bf16win_trace.py
Per workgroup, thread 0 accumulates shader cycles: [0] pass prologue (window + W DMA, first entries, barrier), [1] tap loop, [2] epilogue,
[3] whole kernel, [4] passes."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from futuredet_amd import build_backbone, hip_ops, lib  # noqa: E402
from futuredet_amd.synth import seeded_state_dict, synthetic_cloud  # noqa: E402

dev = torch.device("cuda")
L = lib.load()
raw = ctypes.CDLL(lib.LIB_PATH)
raw.fd_debug_set_wintrace.restype = ctypes.c_int
raw.fd_debug_set_wintrace.argtypes = [ctypes.c_void_p]
pts = torch.from_numpy(synthetic_cloud(0, 300000)).to(dev)
out = hip_ops.voxelize(pts, [0.075, 0.075, 0.2], [-54, -54, -5.0, 54, 54, 3.0], 10, 160000, want_voxels=False, want_mean=True, mean_stride=16, coor_cols=4)
m = int(out["num_voxels"].cpu()[0])
bb = build_backbone(dict(type="SpMiddleResNetFHD", num_input_features=5, ds_factor=8))
bb.load_state_dict(seeded_state_dict(bb, 7), strict=False)
bb = bb.to(dev).eval()
idx = bb.build_indexes(lambda i0: i0.mark(out["coors"][:m].contiguous()), 1, [1440, 1440, 40], dev)
for lvl in (1, 2, 3):
    C = [16, 32, 64, 128][lvl]
    ix = idx[lvl]
    x = torch.randn((ix.n, C), device=dev).bfloat16()
    wpk = hip_ops.pack_spconv_weight(torch.randn((27, C, C)) * (2.0 / (27 * C)) ** 0.5, torch.bfloat16).to(dev)
    nbr = ix.rulebook(ix, [3, 3, 3], [1, 1, 1], [1, 1, 1])
    for rg in [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0"])]:
        hip_ops.set_tuning("bf16_rg", rg)
        for _ in range(3):
            hip_ops.spconv_apply(x, wpk, torch.zeros(C, device=dev), nbr, ix.n, C, residual=x, relu=True)
        trace = torch.zeros((1024 * 8,), dtype=torch.int64, device=dev)
        assert raw.fd_debug_set_wintrace(trace.data_ptr()) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip_ops.spconv_apply(x, wpk, torch.zeros(C, device=dev), nbr, ix.n, C, residual=x, relu=True)
        e1.record()
        torch.cuda.synchronize()
        raw.fd_debug_set_wintrace(None)
        t = trace.cpu().numpy().reshape(-1, 8)
        t = t[t[:, 3] > 0]
        if len(t) == 0:
            print("level %d C=%d rows %d: not on the window kernel of this build" % (lvl, C, ix.n), flush=True)
            continue
        print("level %d C=%d rows %d rg=%d: %d workgroups, kernel %.1f us; cycles per workgroup (median / max): prologue %d / %d, tap loop %d / %d, epilogue %d / %d, "
              "whole %d / %d, passes %d; inside the tap loop (median): requests %d, MFMA phase %d, hand-over wait + barrier %d" % (
                  lvl, C, ix.n, rg, len(t), 1e3 * e0.elapsed_time(e1), np.median(t[:, 0]), t[:, 0].max(), np.median(t[:, 1]), t[:, 1].max(),
                  np.median(t[:, 2]), t[:, 2].max(), np.median(t[:, 3]), t[:, 3].max(), t[:, 4].max(), np.median(t[:, 5]), np.median(t[:, 6]), np.median(t[:, 7])), flush=True)
    hip_ops.set_tuning("bf16_rg", 0)
